"""The 1x1 layers of the predict step's 128 x 128 tile class (+ residual / second input + LeakyReLU), stand-alone: round 4's one-tile
kernel (VPHO_CONV_PERS=0) beside the persistent multi-tile kernel (default).  VPHO_CONV_TILE=1288|12864|64 forces a tile."""
import os, sys, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
# N, H, Cin, Cout, residual, C2 of a strided second input
shapes = [(64, 64, 64, 256, True, 0), (64, 32, 128, 512, True, 0), (64, 16, 256, 1024, True, 0), (128, 8, 512, 2048, True, 0), (64, 32, 128, 256, True, 0),
          (64, 64, 64, 256, False, 64), (64, 32, 128, 512, False, 256), (64, 16, 256, 1024, False, 512), (128, 8, 512, 2048, False, 1024),
          (64, 64, 256, 128, False, 0), (128, 16, 1024, 512, False, 0), (64, 32, 512, 256, False, 0), (64, 32, 512, 128, False, 0), (64, 16, 1024, 256, False, 0)]


def timed(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tot = {'0': 0.0, '1': 0.0}
for (N, H, Cin, Cout, with_res, C2) in shapes:
    x = torch.randn(N, H, H, Cin, device='cuda'); w = torch.randn(Cout, Cin + C2, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
    res = torch.randn(N, H, H, Cout, device='cuda') if with_res else None
    kw = dict(x2=torch.randn(N, H if C2 == 64 else 2 * H, H if C2 == 64 else 2 * H, C2, device='cuda'), stride2=1 if C2 == 64 else 2) if C2 else {}
    f = lambda: ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=res, **kw)
    t = {}
    for mode in ('0', '1', '0', '1'):
        os.environ['VPHO_CONV_PERS'] = mode
        t[mode] = min(t.get(mode, 1.0), timed(f))
    os.environ.pop('VPHO_CONV_PERS')
    fl = 2.0 * N * H * H * (Cin + C2) * Cout
    by = 4.0 * N * H * H * (Cin + C2 + Cout * (2 if with_res else 1))
    tiles = (N * H * H + 127) // 128 * (Cout // 128)
    for m in t: tot[m] += t[m]
    print(f'N{N} H{H} {Cin}{"+" + str(C2) if C2 else ""}->{Cout} res={int(with_res)} ({tiles} tiles): one-tile {t["0"] * 1e6:7.1f} us {fl / t["0"] / 1e12:6.1f} TF/s | persistent {t["1"] * 1e6:7.1f} us '
          f'{fl / t["1"] / 1e12:6.1f} TF/s {by / t["1"] / 1e12:5.2f} TB/s  ({100 * (t["1"] / t["0"] - 1):+.1f} %)', flush=True)
print(f'sum over the shapes: one-tile {tot["0"] * 1e3:.3f} ms, persistent {tot["1"] * 1e3:.3f} ms')
