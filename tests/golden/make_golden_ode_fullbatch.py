"""Golden vectors of the reference's ``cond_ode_sampler`` (lib/model/score_based_model.py:45-105, through
``ScoreBasedModelAgent.sample`` :130-146 and the installed scipy ``solve_ivp``) at the README batch: R = 64 images x 100
hypotheses = 6400 rows per solve, sampling_steps=50, T0=0.65 -- ONE RK45 controller over the whole 6400 x D state (quirk Q5).
Stored: the RHS-evaluation times (scipy's accepted / rejected step sequence), nfev, every 16th row of the final sample and
every 128th row of the dense-output trajectory (as float32).  Inputs are regenerated from seeds by the tests (encodings: seeded normal x 0.3;
prior: torch.manual_seed + torch.randn like sde.py:26-28).  Weights: vpho_amd.synth.bench_state_dict(seed=1).
Run in the build container only (needs /root/reference)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

BS, S, STEPS, T0 = 64, 100, 50, 0.65
FEAT_SEED = {'hand': 31, 'obj': 32}
DRAW_SEED = {'hand': 41, 'obj': 42}


def main():
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import bench_state_dict
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_ode_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['main.py', '--mode', 'eval', '--sample_num', str(S), '--sampling_steps', str(STEPS), '--sample_T0', str(T0)]
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    import torch.utils.model_zoo as zoo
    import lib.model.backbone_FPN_HFL as ref_fpn
    zoo.load_url = lambda url, **kw: ref_fpn.ResNet(ref_fpn.Bottleneck, [3, 4, 6, 3]).state_dict()
    import lib.model.VPHO as ref_vpho
    torch.manual_seed(0)
    ref = ref_vpho.vpho_net().eval()
    sys.argv = ['x']
    from vpho_amd.model.VPHO import vpho_net
    sd = bench_state_dict(vpho_net(assets), seed=1)
    missing, _ = ref.load_state_dict(sd, strict=False)
    assert not missing
    ref.cfg.sampling_steps, ref.cfg.sample_num = STEPS, S
    P = dict(cfg=np.array([BS, S, STEPS]), T0=np.array(T0))
    with torch.no_grad():
        for name, den, D in (('hand', ref.denoiser_hand, 96), ('obj', ref.denoiser_obj, 9)):
            enc = MG.seeded((BS, 1024), FEAT_SEED[name], 0.3)
            feat = enc[:, None].repeat(1, S, 1).reshape(-1, 1024)                 # VPHO.py:238-239
            calls = []
            orig = den.forward
            den.forward = lambda d, _o=orig, _c=calls: (_c.append(float(d['t'][0, 0])), _o(d))[1]
            torch.manual_seed(DRAW_SEED[name])
            t0 = time.time()
            xs, x = ref.score_agent.sample({'feat': feat}, den, T0)
            den.forward = orig
            print(name, 'nfev', len(calls), f'{time.time() - t0:.1f} s', tuple(xs.shape), xs.dtype)
            P[f'{name}_tcalls'] = np.array(calls)
            P[f'{name}_x'] = x.numpy()[::16]
            P[f'{name}_xs'] = xs.numpy()[::128].astype(np.float32)
            P[f'{name}_feat_seed'], P[f'{name}_draw_seed'] = np.array(FEAT_SEED[name]), np.array(DRAW_SEED[name])
    path = os.path.join(HERE, 'golden_ode_fullbatch.npz')
    np.savez_compressed(path, **P)
    print({k: v.shape for k, v in P.items()}, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
