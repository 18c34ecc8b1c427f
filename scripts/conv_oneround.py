"""The 128 x 128 class's launches of AT MOST one round of tiles (<= 2 workgroups per CU), per tile choice: run once per value of VPHO_CONV_TILE
(read once per process):   for t in 0 12864; do VPHO_CONV_TILE=$t python scripts/conv_oneround.py; done
Shapes = the predict step's one-round layers (scripts/conv_layers.py): N, H, Cin, Cout, k, stride, launches per step."""
import os, sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
dev = 'cuda'
def timeit(f, n=20, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best
shapes = [(64, 16, 1024, 256, 1, 1, 10), (64, 32, 512, 128, 1, 1, 6), (64, 32, 512, 256, 1, 1, 2), (128, 8, 2048, 512, 1, 1, 2), (128, 16, 1024, 512, 1, 1, 1),
          (64, 64, 128, 128, 3, 2, 2), (64, 32, 256, 256, 3, 2, 2), (128, 16, 512, 512, 3, 2, 1), (64, 64, 256, 128, 1, 1, 2)]
tot = 0.0
for (N, H, Cin, Cout, k, st, n) in shapes:
    x = torch.randn(N, H, H, Cin, device=dev); w = torch.randn(Cout, Cin * k * k, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    t = timeit(lambda: ops.conv2d_nhwc(x, w, b, kh=k, kw=k, stride=st, pad=k // 2, out_slope=0.01))
    fl = 2 * N * (H // st) ** 2 * Cin * Cout * k * k
    tot += n * t
    print(f'tile {os.environ.get("VPHO_CONV_TILE", "default")}: N{N} H{H} {Cin}->{Cout} k{k} s{st}: {t*1e6:.1f} us  {fl/t/1e12:.1f} TF/s  x{n}', flush=True)
print(f'tile {os.environ.get("VPHO_CONV_TILE", "default")}: {tot*1e3:.3f} ms per step over these layers')
