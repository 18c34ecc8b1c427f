"""Timing of the convolution backward building blocks at ResNet-50 layer sizes (bs=64)."""
import sys, time, torch
sys.argv = ['x']; sys.path.insert(0, '.')
from vpho_amd import ops, conv_backward as CB
dev = 'cuda'
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
for (N, H, cin, cout, k, st) in [(64, 64, 256, 256, 3, 1), (64, 32, 128, 128, 3, 1), (64, 16, 1024, 256, 1, 1), (64, 64, 128, 128, 3, 2), (64, 16, 256, 1024, 1, 1),
                                 (64, 64, 64, 64, 3, 1), (64, 64, 64, 256, 1, 1), (64, 8, 512, 512, 3, 1), (64, 8, 2048, 512, 1, 1), (64, 16, 256, 256, 3, 1),
                                 (64, 256, 4, 64, 7, 2)]:
    pad = k // 2
    OH = (H + 2 * pad - k) // st + 1
    x = torch.randn(N, H, H, cin, device=dev); w = torch.randn(cout, k * k * cin, device=dev) * 0.05; dy = torch.randn(N, OH, OH, cout, device=dev)
    fl = 2.0 * N * OH * OH * cin * cout * k * k
    tf = timeit(lambda: ops.conv2d_nhwc(x, w, None, kh=k, kw=k, stride=st, pad=pad))
    td = timeit(lambda: CB.conv2d_dgrad(dy, w, (H, H), k, k, st, pad))
    tw = timeit(lambda: CB.conv2d_wgrad(x, dy, k, k, st, pad))
    print(f'N{N} H{H} {cin}->{cout} k{k} s{st}: fwd {tf*1e3:.3f} ms ({fl/tf/1e12:.0f} TF/s)  dgrad {td*1e3:.3f} ms ({fl/td/1e12:.0f} TF/s)  wgrad {tw*1e3:.3f} ms ({fl/tw/1e12:.0f} TF/s)')
