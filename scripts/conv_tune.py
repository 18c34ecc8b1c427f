import sys, time, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from vpho_amd import ops
dev='cuda'
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
shapes=[(64,64,256,256,3,1),(64,16,256,256,3,1),(64,32,128,128,3,1),(64,32,128,512,1,1),(64,16,256,1024,1,1),(64,16,1024,256,1,1),(64,64,64,256,1,1),(64,8,512,512,3,1),(64,8,512,2048,1,1),(64,8,2048,512,1,1),(64,64,256,64,1,1),(64,32,512,128,1,1)]
for (N,H,Cin,Cout,k,st) in shapes:
    x=torch.randn(N,H,H,Cin,device=dev); w=torch.randn(Cout,Cin*k*k,device=dev)*0.05; b=torch.randn(Cout,device=dev)
    res=torch.randn(N,H,H,Cout,device=dev)
    t=timeit(lambda: ops.conv2d_nhwc(x,w,b,kh=k,kw=k,stride=st,pad=k//2,out_slope=0.01,res=res))
    fl=2*N*H*H*Cin*Cout*k*k/(st*st)
    print(f'conv N{N} H{H} Cin{Cin} Cout{Cout} k{k}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s', flush=True)
