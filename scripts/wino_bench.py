"""Winograd F(2x2,3x3) kernel against the direct implicit-GEMM kernel on the feature path's 3x3 / stride-1 layers: error and time."""
import sys, time, torch
import os; sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
from vpho_amd.model.pack import winograd_weights
dev = 'cuda'
def timeit(f, n=20, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best
SHAPES = [(64, 32, 128, 128), (64, 16, 256, 256), (128, 8, 512, 512), (64, 32, 256, 128), (64, 64, 64, 64), (64, 64, 256, 256)]
if os.environ.get('WINO_SMALL'):                       # the encoders' small maps (WINO_SMALL=1): is the direct kernel the better choice there?
    SHAPES = [(64, 8, 256, 256), (64, 16, 128, 128), (64, 8, 128, 128), (64, 4, 128, 128), (64, 4, 256, 256), (64, 2, 256, 256)]
for (N, H, Cin, Cout) in SHAPES:
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(N, H, H, Cin, generator=g).to(dev); w = (torch.randn(Cout, 9 * Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev); b = torch.randn(Cout, generator=g).to(dev)
    u = winograd_weights(w)
    yd = ops.conv2d_nhwc(x, w, b, kh=3, kw=3, pad=1, out_slope=0.01)
    yw = ops.conv3x3_winograd(x, u, b, out_slope=0.01)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x[:4].permute(0, 3, 1, 2).double().cpu(), w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double().cpu(), b.double().cpu(), 1, 1), 0.01).permute(0, 2, 3, 1)
    sc = ref.abs().max().item()
    ed, ew = (yd[:4].double().cpu() - ref).abs().max().item() / sc, (yw[:4].double().cpu() - ref).abs().max().item() / sc
    td = timeit(lambda: ops.conv2d_nhwc(x, w, b, kh=3, kw=3, pad=1, out_slope=0.01)); tw = timeit(lambda: ops.conv3x3_winograd(x, u, b, out_slope=0.01))
    fl = 2.0 * N * H * H * Cin * Cout * 9
    print(f'N{N} H{H} {Cin}->{Cout}: direct {td*1e6:.1f} us ({fl/td/1e12:.1f} TF/s, err {ed:.2e})  winograd {tw*1e6:.1f} us ({fl/tw/1e12:.1f} direct-equivalent TF/s, err {ew:.2e})  x{td/tw:.2f}', flush=True)
