import sys, time, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from vpho_amd import ops
N,H,Cin,Cout,k=64,64,256,256,3
x=torch.randn(N,H,H,Cin,device='cuda'); w=torch.randn(Cout,Cin*k*k,device='cuda')*0.05; b=torch.randn(Cout,device='cuda')
f=lambda: ops.conv2d_nhwc(x,w,b,kh=k,kw=k,pad=1,out_slope=0.01)
f(); torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); t=(time.perf_counter()-t)/10
print(f'{t*1e3:.3f} ms {2*N*H*H*Cin*Cout*9/t/1e12:.1f} TF/s')
