"""Weight-gradient kernel on the training step's main shapes (time per launch incl. the slice sum), for the ablation builds of
scripts/kernel_ablate.sh conv_wgrad WGRAD_ABLATE 0 1 2 4 7  (VPHO_HIP_LIB selects the build)."""
import os, sys, time, torch
sys.argv = ['x']; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import conv_backward as CB
dev = 'cuda'
def timeit(f, n=10, reps=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best
out = []
for (N, H, cin, cout, k, st) in [(64, 32, 128, 128, 3, 1), (64, 16, 256, 256, 3, 1), (64, 16, 1024, 256, 1, 1), (64, 16, 256, 1024, 1, 1), (64, 32, 128, 512, 1, 1), (64, 64, 64, 64, 3, 1), (64, 64, 64, 256, 1, 1)]:
    pad = k // 2
    OH = (H + 2 * pad - k) // st + 1
    x = torch.randn(N, H, H, cin, device=dev); dy = torch.randn(N, OH, OH, cout, device=dev)
    fl = 2.0 * N * OH * OH * cin * cout * k * k
    tw = timeit(lambda: CB.conv2d_wgrad(x, dy, k, k, st, pad))
    out.append(f'{cin}->{cout}k{k}@{H}: {tw * 1e6:.0f} us {fl / tw / 1e12:.0f} TF/s')
print(' | '.join(out), flush=True)
