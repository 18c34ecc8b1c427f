"""One Winograd launch, a few times: target of PMC passes.  argv: N H Cin Cout"""
import os, sys, torch
a = [int(v) for v in sys.argv[1:5]]
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
from vpho_amd.model.pack import winograd_weights
N, H, Cin, Cout = a
g = torch.Generator().manual_seed(1)
x = torch.randn(N, H, H, Cin, generator=g).cuda(); w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.02).cuda(); b = torch.randn(Cout, generator=g).cuda()
u = winograd_weights(w)
for _ in range(5):
    ops.conv3x3_winograd(x, u, b, out_slope=0.01)
torch.cuda.synchronize()
