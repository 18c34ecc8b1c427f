"""times of the Winograd kernel (two layer shapes) for the library named by VPHO_HIP_LIB; see scripts/wino_ablate.sh"""
import os, sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd import ops
from vpho_amd.model.pack import winograd_weights
def timeit(f, n=20, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best
out = []
for (N, H, Cin, Cout) in [(64, 16, 256, 256), (64, 64, 256, 256)]:
    x = torch.randn(N, H, H, Cin, device='cuda'); w = torch.randn(Cout, 9 * Cin, device='cuda') * 0.02; b = torch.randn(Cout, device='cuda')
    u = winograd_weights(w)
    t = timeit(lambda: ops.conv3x3_winograd(x, u, b, out_slope=0.01))
    out.append(f'{t*1e6:8.1f} us ({2.0*N*H*H*Cin*Cout*4/t/1e12:5.1f} TF/s executed)')
print(os.environ.get('VPHO_HIP_LIB', 'product'), ' | '.join(out), flush=True)
