"""One 1x1 convolution (+ bias + residual + LeakyReLU) launched a few times: the target of scripts/pmc_conv.sh.  argv: N H Cin Cout res(0|1)"""
import sys, torch
a = [int(v) for v in sys.argv[1:6]]
import os; sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
N, H, Cin, Cout, with_res = a
x = torch.randn(N, H, H, Cin, device='cuda'); w = torch.randn(Cout, Cin, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
res = torch.randn(N, H, H, Cout, device='cuda') if with_res else None
for _ in range(6):
    ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=res)
torch.cuda.synchronize()
