"""Experiment: capture Engine.features into a HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture) and compare
replay time / CPU enqueue time / outputs with the eager path."""
import sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.engine import Engine
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, a).items()}
eng = Engine(m)
for _ in range(3): ref = eng.features(data)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); f = eng.features(data); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'eager   : CPU enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms')
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.features(data)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.graph(g, stream=s):
    fg = eng.features(data)
torch.cuda.synchronize()
print(f'capture+instantiate {1e3*(time.perf_counter()-t0):.1f} ms')
for _ in range(5):
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'replay  : CPU enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms')
for k, v in ref.items():
    if torch.is_tensor(v):
        print(k, 'max abs diff', float((v.float() - fg[k].float()).abs().max()))
