import sys, time, torch
sys.argv=['x']; sys.path.insert(0,'.')
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict
from vpho_amd.assets import synthetic_assets
from vpho_amd import ops
from vpho_amd.model.pack import pack_conv
import numpy as np
a=synthetic_assets(0); m=vpho_net(a); sd=synth_state_dict(m,1)
dev='cuda'
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n
# conv layers at bs=64
for (N,H,Cin,Cout,k,st) in [(64,64,64,64,1,1),(64,64,64,64,3,1),(64,64,256,256,3,1),(64,32,128,128,3,1),(64,16,256,256,3,1),(64,8,512,512,3,1),(64,16,1024,256,1,1),(64,8,2048,512,1,1),(64,32,256,256,3,1)]:
    x=torch.randn(N,H,H,Cin,device=dev); w=torch.randn(Cout,Cin*k*k,device=dev)*0.05; b=torch.randn(Cout,device=dev)
    t=timeit(lambda: ops.conv2d_nhwc(x,w,b,kh=k,kw=k,stride=st,pad=k//2,out_slope=0.01))
    fl=2*N*H*H*Cin*Cout*k*k/(st*st)
    print(f'conv N{N} H{H} Cin{Cin} Cout{Cout} k{k}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TF/s')
for name,D in (('hand',96),('obj',9)):
    net=ops.ScoreNet(sd,f'denoiser_{name}',dev)
    bs,S=64,100
    feat=torch.randn(bs,1024,device=dev)*0.3
    x=torch.randn(bs*S,D,device=dev)
    t=timeit(lambda: net.score(feat,x,0.3,S),10)
    R=bs*S; fl=R*(2*net.Dp*256+2*256*256+net.nheads*(2*256*256+2*256*3))
    print(f'score {name}: {t*1e3:.3f} ms/eval  {fl/t/1e12:.1f} TF/s (restructured flops)')
    init=torch.randn(bs*S,D)*2.5
    torch.cuda.synchronize(); t0=time.time()
    xs,xf,st=net.sample(feat,init.to(dev),S,0.65,50,xs_f64=(name=='obj'))
    torch.cuda.synchronize(); dt=time.time()-t0
    print(f'sample {name}: {dt*1e3:.1f} ms nfev {st["nfev"]} acc {st["n_accepted"]} rej {st["n_rejected"]} -> {dt/st["nfev"]*1e3:.3f} ms/eval')
