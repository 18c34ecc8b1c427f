"""Times one convolution shape (for VPHO_CONV_DBG ablations of the GLDS kernel)."""
import sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd import ops
for (N, H, Cin, Cout, k) in [(64, 64, 256, 256, 3), (64, 16, 1024, 256, 1), (64, 64, 256, 128, 1)]:
    x = torch.randn(N, H, H, Cin, device='cuda'); w = torch.randn(Cout, Cin * k * k, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
    f = lambda: ops.conv2d_nhwc(x, w, b, kh=k, kw=k, pad=k // 2, out_slope=0.01)
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print(f'N{N} H{H} {Cin}->{Cout} k{k}: {dt*1e3:.3f} ms  {2.0*N*H*H*Cin*Cout*k*k/dt/1e12:.1f} TF/s')
