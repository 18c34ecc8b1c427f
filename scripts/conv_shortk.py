"""The short-K 1x1 layers (+ residual + LeakyReLU) under a forced tile (VPHO_CONV_TILE=1288|12864|64; unset: the plan's own choice)."""
import os, sys, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
shapes = [(64, 64, 64, 256, True), (64, 32, 128, 512, True), (64, 16, 256, 1024, True), (64, 32, 128, 256, False), (64, 64, 256, 64, False),
          (64, 32, 512, 128, False), (64, 16, 1024, 256, False), (128, 8, 512, 2048, True)]
for (N, H, Cin, Cout, with_res) in shapes:
    x = torch.randn(N, H, H, Cin, device='cuda'); w = torch.randn(Cout, Cin, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
    res = torch.randn(N, H, H, Cout, device='cuda') if with_res else None
    f = lambda: ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=res)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 2.0 * N * H * H * Cin * Cout
    by = 4.0 * N * H * H * (Cin + Cout * (2 if with_res else 1))
    print(f'tile {os.environ.get("VPHO_CONV_TILE", "plan"):>5s}  N{N} H{H} {Cin}->{Cout} res={int(with_res)}: {t * 1e6:7.1f} us  {fl / t / 1e12:6.1f} TF/s  {by / t / 1e12:5.2f} TB/s')
