import sys, time, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from vpho_amd import ops
dev='cuda'
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
N,H,Cin,Cout=64,64,64,256
x=torch.randn(N,H,H,Cin,device=dev); w=torch.randn(Cout,Cin,device=dev)*0.05; b=torch.randn(Cout,device=dev)
res=torch.randn(N,H,H,Cout,device=dev); out=torch.empty(N,H,H,Cout,device=dev)
for name,kw in (('bias+res+lrelu',dict(res=res,out_slope=0.01)),('bias only',dict()),('no bias',dict(nob=True))):
    nob=kw.pop('nob',False)
    t=timeit(lambda: ops.conv2d_nhwc(x,w,None if nob else b,out=out,**kw))
    byt=x.numel()*4+out.numel()*4*(2 if 'res' in kw else 1)
    print(f'{name}: {t*1e6:.1f} us  {byt/t/1e12:.2f} TB/s', flush=True)
# pure copy reference
y=torch.empty_like(res)
t=timeit(lambda: y.copy_(res)); print(f'torch copy 268MB: {t*1e6:.1f} us {2*res.numel()*4/t/1e12:.2f} TB/s')
t=timeit(lambda: torch.add(res, out, out=y)); print(f'torch add: {t*1e6:.1f} us {3*res.numel()*4/t/1e12:.2f} TB/s')
