"""One score-network evaluation: error of the HIP kernels and of the oracle's fp32 (torch CPU) arithmetic against a float64 evaluation,
relative to max |score|; per output dimension for the object network.  python scripts/score_error.py"""
import os, sys, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict
from oracle import nets as N
sd = bench_state_dict(vpho_net(synthetic_assets(0)), seed=1)
g = torch.Generator().manual_seed(3)
bs, S = 16, 100
for name, D, xs in (('hand', 96, 1.5), ('obj', 9, 1.0)):
    p = f'denoiser_{name}'
    net = ops.ScoreNet(sd, p, 'cuda')
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith(p)}
    feat = torch.randn(bs, 1024, generator=g) * 0.3
    x = torch.randn(bs * S, D, generator=g) * xs
    fr = feat[:, None].repeat(1, S, 1).reshape(-1, 1024)
    for t in (0.65, 0.3, 0.05, 1e-5):
        tt = torch.full((bs * S, 1), t)
        ref = N.denoiser(sd64, p, fr.double(), x.double(), tt.double())
        o32 = N.denoiser(sd, p, fr, x, tt).double()
        hip = net.score(feat.cuda(), x.cuda(), t, S).cpu().double()
        sc = float(ref.abs().max())
        f = lambda e: f'{float(e.abs().max()) / sc:.2e} / {float(e.pow(2).mean().sqrt()) / sc:.2e}'
        print(f'{name} t={t}: max / rms error relative to max|score| {sc:.3g}: HIP {f(hip - ref)}   oracle fp32 {f(o32 - ref)}   HIP - oracle {f(hip - o32)}')
        if name == 'obj':
            print('    per dimension rms (HIP | oracle):', ' '.join(f'{float((hip - ref)[:, d].pow(2).mean().sqrt()) / sc:.1e}|{float((o32 - ref)[:, d].pow(2).mean().sqrt()) / sc:.1e}' for d in range(D)))
