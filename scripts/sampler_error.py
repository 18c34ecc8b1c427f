"""ODE solve of one batch (README sizes, bs 16): error of the HIP sampler and of the oracle's fp32 solve against the float64 solve of the same
accepted step sequence (oracle/sampler_fp64.py), per network and per output dimension; device controller and VPHO_RK_HOST=1.
python scripts/sampler_error.py"""
import os, sys, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict
from oracle import nets as N, sampler_fp64 as SF
sd = bench_state_dict(vpho_net(synthetic_assets(0)), seed=1)
g = torch.Generator().manual_seed(5)
bs, S, T0, steps = 16, 100, 0.65, 50
for name, D in (('hand', 96), ('obj', 9)):
    p = f'denoiser_{name}'
    feat = torch.randn(bs, 1024, generator=g) * 0.3
    init = torch.randn(bs * S, D, generator=g) * N.ve_prior_sigma(T0)
    fr = feat[:, None].repeat(1, S, 1).reshape(-1, 1024)
    xs_o, x_o, info = N.ode_sample(sd, p, fr, init, T0, steps)
    rows = torch.arange(0, bs * S, 4)
    x64 = SF.solve_on_steps(sd, p, fr, init, info['steps'], steps, rows)
    eo = x_o[rows].double() - x64
    print(f'{name}: oracle fp32 vs fp64: max {float(eo.abs().max()):.2e} rms {float(eo.pow(2).mean().sqrt()):.2e}; nfev {info["nfev"]}')
    for host in ('0', '1'):
        for f64 in (True, False):
            os.environ['VPHO_RK_HOST'] = host
            net = ops.ScoreNet(sd, p, 'cuda')
            xs, x, st = net.sample(feat.cuda(), init.cuda(), S, T0, steps, xs_f64=f64, x_f64=f64)
            eh = x.cpu()[rows].double() - x64
            print(f'   HIP (VPHO_RK_HOST={host}, f64 outputs {f64}): nfev {st["nfev"]}  vs fp64: max {float(eh.abs().max()):.2e} rms {float(eh.pow(2).mean().sqrt()):.2e}; vs oracle max {float((x.cpu().double() - x_o.double()).abs().max()):.2e}'
                  + ('   per dim rms: ' + ' '.join(f'{float(eh[:, d].pow(2).mean().sqrt()):.1e}' for d in range(D)) if D == 9 else ''))
os.environ.pop('VPHO_RK_HOST')
# ---- where along the trajectory does each fp32 solve of the object network leave the exact scheme?  (dense stamps, all float64)
p, D = 'denoiser_obj', 9
for seed in (6, 7):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(bs, 1024, generator=g) * 0.3
    init = torch.randn(bs * S, D, generator=g) * N.ve_prior_sigma(T0)
    fr = feat[:, None].repeat(1, S, 1).reshape(-1, 1024)
    xs_o, x_o, info = N.ode_sample(sd, p, fr, init, T0, steps)
    net = ops.ScoreNet(sd, p, 'cuda')
    xs, x, st = net.sample(feat.cuda(), init.cuda(), S, T0, steps, xs_f64=True, x_f64=True)
    rows = torch.arange(0, bs * S, 4)
    x64, xs64 = SF.solve_on_steps(sd, p, fr, init, info['steps'], steps, rows, dense=True, T0=T0)
    dh, do = (xs.cpu()[rows] - xs64).abs(), (xs_o[rows] - xs64).abs()
    fmt = lambda d: ' '.join(f'{float(d[:, i].max()):.1e}' for i in range(0, steps, 3))
    print(f'seed {seed}: obj |x - x_fp64| at the dense stamps 0, 3, 6, ... (max over rows and dimensions)')
    print('   HIP   :', fmt(dh))
    print('   oracle:', fmt(do))
    print('   rms at the last stamp: HIP', f'{float(dh[:, -1].pow(2).mean().sqrt()):.2e}', 'oracle', f'{float(do[:, -1].pow(2).mean().sqrt()):.2e}',
          '| after the denoise step: HIP', f'{float((x.cpu()[rows] - x64).pow(2).mean().sqrt()):.2e}', 'oracle', f'{float((x_o[rows] - x64).pow(2).mean().sqrt()):.2e}')
# ---- the same through the whole engine (real encodings of the conditioned benchmark weights), bs 8
from vpho_amd.configs.args import cfg
from vpho_amd.synth import synth_batch
from vpho_amd.assets import ANCHOR_SKELETON
from oracle import vpho as OV
assets = synthetic_assets(0)
m = vpho_net(assets); m.load_state_dict(sd); m = m.cuda().eval()
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, steps, 30, 10, T0
n = 8
data = synth_batch(n, assets, seed=777)
torch.manual_seed(99)
nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
ref, info = OV.predict(sd, assets, ANCHOR_SKELETON, data, sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=30, topk_obj=10, noise_hand=nh, noise_obj=no)
gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
m(gdata, mode='predict')
for serial in (False, True):
    m._engine.serial_samplers = serial
    out = m._engine.predict(gdata, noise_hand=nh, noise_obj=no)
    torch.cuda.synchronize()
    gi = m._engine.last_info
    sig = N.ve_prior_sigma(T0)
    rep = lambda e: e.detach().cpu()[:, None].repeat(1, S, 1).reshape(-1, 1024)
    r = SF.compare(sd, 'denoiser_obj', rep(info['features']['encoding_obj']), no * sig, info['obj_ode']['steps'], steps, out['diff_final_obj_6d'].reshape(-1, 9),
                   ref['diff_final_obj_6d'].reshape(-1, 9), feat_hip=rep(gi['features']['encoding_obj']), steps_hip=gi['obj_ode']['steps'], stride=4)
    print(f'engine (serial samplers {serial}) obj:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in r.items()})
    rows = torch.arange(0, n * S, 4)
    x64h = SF.solve_on_steps(sd, 'denoiser_obj', rep(gi['features']['encoding_obj']), no * sig, gi['obj_ode']['steps'], steps, rows)
    e = out['diff_final_obj_6d'].reshape(-1, 9).cpu()[rows] - x64h
    print('   per dim rms:', ' '.join(f'{float(e[:, d].pow(2).mean().sqrt()):.1e}' for d in range(9)), '| per image rms:', ' '.join(f'{float(e.view(n, -1, 9)[i].pow(2).mean().sqrt()):.1e}' for i in range(n)))
    print('   HIP steps:   ', [(round(float(s[0]), 7), round(float(s[1]), 7)) for s in gi['obj_ode']['steps'] if s[3]])
    print('   oracle steps:', [(round(float(s[0]), 7), round(float(s[1]), 7)) for s in info['obj_ode']['steps'] if s[3]])
