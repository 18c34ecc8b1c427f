"""Is the training step launch-bound?  forward + backward (DiffusionTrainStep.loss_and_grads, ~2 500 launches) eagerly against ONE HIP-graph
replay of the same launches (both with the weight gradients on their own stream), ms per call.  python scripts/train_graph_probe.py [bs]"""
import os, sys, time
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.train_step import DiffusionTrainStep
from vpho_amd.trainer import synthetic_mano_targets
dev = torch.device('cuda', 0)
torch.manual_seed(206); torch.cuda.manual_seed(206)
assets = synthetic_assets(0)
sd = synth_state_dict(vpho_net(assets), seed=1)
step = DiffusionTrainStep(sd, dev, assets=assets)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=11).items()}
g = torch.Generator().manual_seed(100)
data['hm_hand'] = (torch.rand(bs, 21, 64, 64, generator=g) * 0.2).to(dev)
data['hm_obj'] = (torch.rand(bs, 27, 64, 64, generator=g) * 0.2).to(dev)
gt_h = (torch.randn(bs, 96, generator=g) * 0.5).to(dev) + torch.tensor([1., 0, 0, 0, 1, 0], device=dev).repeat(16)
gt_o = (torch.randn(bs, 9, generator=g) * 0.5).to(dev)
data.update(synthetic_mano_targets(step.mano_head.mano, gt_h, (torch.randn(bs, 10, generator=g) * 0.5).to(dev), data['is_right']))
data['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(dev)
reps = 20
u = lambda: torch.rand(reps, bs, device=dev) * (1. - 1e-5) + 1e-5
draws = dict(t_h=u(), z_h=torch.randn(reps, bs, 96, device=dev), t_o=u(), z_o=torch.randn(reps, bs, 9, device=dev))
for _ in range(2):
    step.step(data, gt_h, gt_o, draws=draws)
torch.cuda.synchronize()
def fb():
    step.buckets.begin()
    L, G = step.loss_and_grads(data, gt_h, gt_o, draws, sink=step.buckets)
    step.buckets.finish()
    return L
def timeit(f, n=8):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
t_step = timeit(lambda: step.step(data, gt_h, gt_o, draws=draws))
t_eager = timeit(fb)
cpu0 = time.process_time(); fb(); cpu_issue = (time.process_time() - cpu0) * 1e3; torch.cuda.synchronize()
print(f'bs {bs}: whole step {t_step:.1f} ms; forward + backward eager {t_eager:.1f} ms (CPU time to issue one: {cpu_issue:.1f} ms)', flush=True)
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
        Lg = fb()
    torch.cuda.current_stream().wait_stream(side)
    t_graph = timeit(graph.replay)
    print(f'forward + backward as ONE graph replay {t_graph:.1f} ms  (loss {float(Lg["total_loss"]):.6f})', flush=True)
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:600], flush=True)
