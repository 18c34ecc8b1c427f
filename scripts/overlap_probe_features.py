"""Companion of overlap_probe.py for the FEATURE path (backbone, FPN, RoIAlign, heat-map heads, encoders, cross modules: Engine.features):
N feature passes of the 64-image batch on ONE stream against the same N split over two / three streams (an execution plan per stream,
driven by its own host thread like the pipelined evaluator's slots).  Wall clock per pass."""
import os, sys, time, threading, torch
sys.argv = ['x']; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.engine import Engine
dev = torch.device('cuda', 0)
assets = synthetic_assets(0); model = vpho_net(assets); model.load_state_dict(bench_state_dict(model, seed=1)); model = model.to(dev).eval()
b = synth_batch(64, assets, seed=206, rank=0)
batch = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in b.items()}
engines = [Engine(model) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
N = 24
def drive(j, n):
    with torch.cuda.device(dev), torch.cuda.stream(streams[j]), torch.no_grad():
        for _ in range(n):
            engines[j].features(batch)
for j in range(3):
    drive(j, 2)
torch.cuda.synchronize()
for rep in range(2):
    for k in (1, 2, 3):
        torch.cuda.synchronize(); t = time.perf_counter()
        th = [threading.Thread(target=drive, args=(j, N // k)) for j in range(k)]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t) / N
        print(f'{k} stream(s): {t * 1e3:.2f} ms per feature pass of 64 images', flush=True)
