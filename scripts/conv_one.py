import sys, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from vpho_amd import ops
N,H,Cin,Cout,k=64,64,256,256,3
x=torch.randn(N,H,H,Cin,device='cuda'); w=torch.randn(Cout,Cin*k*k,device='cuda')*0.05; b=torch.randn(Cout,device='cuda')
for _ in range(5): ops.conv2d_nhwc(x,w,b,kh=k,kw=k,pad=1,out_slope=0.01)
torch.cuda.synchronize()
