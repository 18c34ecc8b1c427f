import sys, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd import ops
def timeit(name, f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    print(f'{name:40s} {a.elapsed_time(b) / n * 1e3:9.1f} us')
N, H = 64, 16
x = torch.randn(N, H, H, 256, device='cuda'); w = torch.randn(1024, 256, device='cuda') * 0.05; b = torch.randn(1024, device='cuda')
r = torch.randn(N, H, H, 1024, device='cuda')
big = torch.empty(2 * N, H, H, 1024, device='cuda')
timeit('1x1 256->1024 + res', lambda: ops.conv2d_nhwc(x, w, b, res=r, out_slope=0.01))
timeit('same, out = upper half of 2N buffer', lambda: ops.conv2d_nhwc(x, w, b, res=r, out_slope=0.01, out=big[N:]))
timeit('same, out = lower half', lambda: ops.conv2d_nhwc(x, w, b, res=r, out_slope=0.01, out=big[:N]))
