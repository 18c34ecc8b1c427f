"""Golden vectors that settle the NaN guard of the sampler (lib/model/score_based_model.py:65-72): the reference's
``cond_ode_sampler`` run with a score network that returns NaN in the three dimensions of one head and +inf / -inf in those of
another at EVERY evaluation (planted in ``head.head.2.bias``).  Inside the solve the reference's ``score_eval_wrapper`` sees a NaN,
prints its warning and zeroes NaN, +inf and -inf alike (``nan_to_num_(nan=0, posinf=0, neginf=0)``): those dimensions get a zero
right-hand side and keep their start value through the whole trajectory, the controller runs on the finite dimensions.  The final
denoise evaluation (:95-102) is NOT guarded: the returned sample is NaN / -inf / +inf there.  Stored: RHS-evaluation times, xs, x.
Object network (3 heads: head 0 NaN, head 1 finite, head 2 +-inf) and hand network (32 heads: head 5 NaN, head 9 +-inf).
Run in the build container only (needs /root/reference)."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

BS, S, STEPS, T0 = 3, 5, 7, 0.65
PLANT = {'obj': dict(nan_head=0, inf_head=2), 'hand': dict(nan_head=5, inf_head=9)}
FEAT_SEED = {'hand': 131, 'obj': 132}
DRAW_SEED = {'hand': 141, 'obj': 142}


def plant(sd, name):
    """the state_dict entries with the planted biases (also used by the tests to build the same weights)"""
    b = sd[f'denoiser_{name}.head.head.2.bias'].clone()
    b[PLANT[name]['nan_head']] = float('nan')
    b[PLANT[name]['inf_head']] = torch.tensor([float('inf'), float('-inf'), float('inf')])
    return b


def main():
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import bench_state_dict
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_nan_')
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
    for name in ('hand', 'obj'):
        sd[f'denoiser_{name}.head.head.2.bias'] = plant(sd, name)
    missing, _ = ref.load_state_dict(sd, strict=False)
    assert not missing
    ref.cfg.sampling_steps, ref.cfg.sample_num = STEPS, S
    P = dict(cfg=np.array([BS, S, STEPS]), T0=np.array(T0))
    with torch.no_grad():
        for name, den, D in (('hand', ref.denoiser_hand, 96), ('obj', ref.denoiser_obj, 9)):
            enc = MG.seeded((BS, 1024), FEAT_SEED[name], 0.3)
            feat = enc[:, None].repeat(1, S, 1).reshape(-1, 1024)
            calls = []
            orig = den.forward
            den.forward = lambda d, _o=orig, _c=calls: (_c.append(float(d['t'][0, 0])), _o(d))[1]
            torch.manual_seed(DRAW_SEED[name])
            xs, x = ref.score_agent.sample({'feat': feat}, den, T0)
            den.forward = orig
            nh, ih = PLANT[name]['nan_head'], PLANT[name]['inf_head']
            print(name, 'nfev', len(calls), 'x planted dims:', x[0, 3 * nh:3 * nh + 3].tolist(), x[0, 3 * ih:3 * ih + 3].tolist(),
                  'xs finite:', bool(torch.isfinite(xs).all()))
            P[f'{name}_tcalls'], P[f'{name}_x'], P[f'{name}_xs'] = np.array(calls), x.numpy(), xs.numpy()
            P[f'{name}_feat_seed'], P[f'{name}_draw_seed'] = np.array(FEAT_SEED[name]), np.array(DRAW_SEED[name])
            P[f'{name}_plant'] = np.array([nh, ih])
    path = os.path.join(HERE, 'golden_nan_guard.npz')
    np.savez_compressed(path, **P)
    print({k: v.shape for k, v in P.items()}, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
