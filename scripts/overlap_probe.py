"""Do two chip-filling kernels of different streams fill each other's ramps and tails?  N score evaluations (time embedding + pose encoder +
score head, the sampler's per-stage chain) on ONE stream against the same N split over TWO streams (two score networks with their own
workspaces), wall clock.  Perfect filling would bring the two-stream time towards the matrix-bound time; no filling leaves it at the
one-stream time."""
import os, sys, time, torch
sys.argv = ['x']; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict
from vpho_amd.assets import synthetic_assets
from vpho_amd import ops
a = synthetic_assets(0); m = vpho_net(a); sd = synth_state_dict(m, 1)
dev = 'cuda'
bs, S, N = 64, 100, 60
nets = [ops.ScoreNet(sd, 'denoiser_hand', dev) for _ in range(3)]
feat = torch.randn(bs, 1024, device=dev) * 0.3
xs = [torch.randn(bs * S, 96, device=dev) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
for n, x in zip(nets, xs):
    for _ in range(3): n.score(feat, x, 0.3, S)
torch.cuda.synchronize()
def run(k):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(N):
        j = i % k
        with torch.cuda.stream(streams[j]):
            nets[j].score(feat, xs[j], 0.3, S)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N
for rep in range(2):
    for k in (1, 2, 3):
        t = run(k)
        print(f'{k} stream(s): {t * 1e6:.1f} us per evaluation (time embedding + pose encoder + hand score head); 27.1 + 1.15 GFLOP -> {28.3e9 / t / 1e12:.1f} TF/s', flush=True)
