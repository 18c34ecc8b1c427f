"""Generate the golden fixtures in this directory by IMPORTING the reference's own Python modules.

Run in the build container only (``python tests/golden/make_golden.py``); needs ``/root/reference``.  Nothing here
travels to the GPU box except the resulting ``*.npz`` files.

What is real reference code and what is stubbed:
* imported unchanged: lib.configs.args, lib.model.{VPHO, backbone_FPN_HFL, head_inplane, encoding, denoiser,
  parallel_linear, cross_module, physics, sde, score_based_model, aggregation, head_mano, head_object},
  lib.utils.{hand_fn, physics_fn, transform_fn}; scipy.integrate.solve_ivp is the installed scipy.
* not installed here, bound to the oracle's restatements: pytorch3d.transforms.rotation_conversions,
  manopth.manolayer.ManoLayer, torchvision.ops.roi_align.  (timm / ipdb / pytorch3d.ops.knn are imported by the
  reference but never called on this path: empty stubs.)
* assets: the synthetic tables of vpho_amd.assets written in the reference's on-disk formats under a temp CWD;
  ``lib.dataset.base`` (which would open the DexYCB model directory at import) is replaced by a stub exposing
  ``YCB_MESHES``.
Weights: vpho_amd.synth.synth_state_dict(seed) loaded into the reference module; inputs: vpho_amd.synth.synth_batch.
Neither is stored -- the tests regenerate them from the same seeds.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = '/root/reference'

# cfg1 sizes; T0=0.2 keeps the random-weight object hypotheses inside the crop so that the heat-map / physics top-k are
# not all-tied zeros (the T0=0.65 sampler is pinned by the ode_* block fixtures)
CFG1 = dict(bs=2, sample_num=4, sampling_steps=5, topk_hand=8, topk_obj=3, sample_T0=0.2)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs(assets):
    from oracle import rotations as R, mano as M, roi_align as RA

    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops', roi_align=lambda f, b, output_size, spatial_scale=1.0, sampling_ratio=-1,
                   aligned=False: RA.roi_align_fast(f, b, output_size, spatial_scale))
    _stub('timm')
    _stub('timm.models', create_model=None)
    _stub('timm.utils', ModelEmaV3=None)
    _stub('ipdb', set_trace=None)
    _stub('pytorch3d')
    _stub('pytorch3d.transforms', **{k: getattr(R, k) for k in dir(R) if not k.startswith('_')})
    _stub('pytorch3d.transforms.rotation_conversions', **{k: getattr(R, k) for k in dir(R) if not k.startswith('_')})
    _stub('pytorch3d.ops')
    _stub('pytorch3d.ops.knn', knn_points=None)

    class ManoLayer(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            assert kw['ncomps'] == 45 and kw['center_idx'] == 0 and kw['flat_hand_mean'] and not kw['use_pca']
            self.assets = {k: torch.as_tensor(v) for k, v in assets['mano'].items()}

        def forward(self, th_pose_coeffs, th_betas):
            return M.mano_forward(self.assets, th_pose_coeffs, th_betas)

    _stub('manopth')
    _stub('manopth.manolayer', ManoLayer=ManoLayer)
    ycb = {k: dict(kpt3d=v['kpt3d'], verts_sampled=v['verts_sampled'], verts=v['verts'], CoM=v['CoM'],
                   shift=np.eye(4)) for k, v in assets['ycb'].items()}
    _stub('lib.dataset')
    _stub('lib.dataset.base', YCB_MESHES=ycb, YCB_CLASSES={i + 1: n for i, n in enumerate(ycb)},
          YCB_ID={n: i + 1 for i, n in enumerate(ycb)})


def write_assets(root, assets):
    a = assets['anchor']
    os.makedirs(os.path.join(root, 'asset/ours'), exist_ok=True)
    os.makedirs(os.path.join(root, 'asset/2021_CVPR_CPF/anchor'), exist_ok=True)
    with open(os.path.join(root, 'asset/ours/vert2joint.pkl'), 'wb') as f:
        pickle.dump({'vert2joint': a['vert2joint']}, f)
    d = os.path.join(root, 'asset/2021_CVPR_CPF/anchor')
    np.savetxt(os.path.join(d, 'face_vertex_idx.txt'), a['face_vert_idx'], fmt='%d')
    np.savetxt(os.path.join(d, 'anchor_weight.txt'), a['anchor_weight'], fmt='%.9e')
    np.savetxt(os.path.join(d, 'merged_vertex_assignment.txt'), np.zeros(778, dtype=np.int32), fmt='%d')
    with open(os.path.join(d, 'anchor_mapping_path.pkl'), 'wb') as f:
        pickle.dump({}, f)


def seeded(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(size=shape) * scale).astype(np.float32))


def main():
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import synth_state_dict, synth_batch

    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_')
    write_assets(tmp, assets)
    os.chdir(tmp)
    c = CFG1
    sys.argv = ['main.py', '--mode', 'eval', '--sample_num', str(c['sample_num']), '--sampling_steps',
                str(c['sampling_steps']), '--topk_hand', str(c['topk_hand']), '--topk_obj', str(c['topk_obj']),
                '--sample_T0', str(c['sample_T0'])]
    sys.path.insert(0, REF)
    install_stubs(assets)
    import torch.utils.model_zoo as zoo
    import lib.model.backbone_FPN_HFL as ref_fpn
    zoo.load_url = lambda url, **kw: ref_fpn.ResNet(ref_fpn.Bottleneck, [3, 4, 6, 3]).state_dict()
    import lib.model.VPHO as ref_vpho
    from lib.utils.transform_fn import average_quaternion as ref_avgq
    from lib.utils.physics_fn import VERT2ANCHOR
    import lib.model.score_based_model as ref_sbm
    from lib.engine.test import TesterHand

    torch.manual_seed(0)
    ref = ref_vpho.vpho_net().eval()
    # our container module only provides the key layout + seeded values
    sys.argv = ['x']
    from vpho_amd.model.VPHO import vpho_net
    sd = synth_state_dict(vpho_net(assets), seed=1)
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing, missing
    print('unexpected keys (our extra manopth buffers):', unexpected)
    G = {}

    with torch.no_grad():
        # ---- blocks -------------------------------------------------------------------------------------------
        x = seeded((1, 3, 64, 64), 11)
        ph, po = ref.feature_extractor(x)
        G['fpn_h'], G['fpn_o'] = ph.numpy(), po.numpy()
        x = seeded((1, 256, 32, 32), 12)
        G['hm_hand'] = ref.head_hm_hand(x)[:, :, ::2, ::2].numpy()
        x = seeded((1, 277, 32, 32), 13, 0.3)
        e, st = ref.encoder_hand(x)
        G['enc_hand'], G['enc_hand_stage1'] = e.numpy(), st[1].numpy()
        for bs in (1, 3):
            xh, xo, g = seeded((bs, 256, 8, 8), 14, 0.2), seeded((bs, 256, 8, 8), 15, 0.2), seeded((bs, 1, 3), 16)
            yh, yo, yg = ref.cross_hand(xh, xo, g)
            G[f'cross_hand_bs{bs}'] = torch.cat([yh, yo, yg], 1).numpy()
        th, to = seeded((2, 32, 512), 17), seeded((2, 32, 512), 18)
        G['force_local'] = ref.head_physics(th, to)['force_local'].numpy()
        enc = seeded((3, 1024), 19, 0.3)
        pose, shape = ref.head_mano(enc)
        G['mano_pose'], G['mano_shape'] = pose.numpy(), shape.numpy()
        for name, den, D in (('hand', ref.denoiser_hand, 96), ('obj', ref.denoiser_obj, 9)):
            feat, xx = seeded((6, 1024), 20, 0.3), seeded((6, D), 21, 1.5)
            tt = torch.linspace(0.05, 0.65, 6)[:, None]
            G[f'score_{name}'] = den({'feat': feat, 'sampled_pose': xx, 't': tt}).numpy()
            # full cond_ode_sampler run with the reference's own prior draw
            torch.manual_seed(5)
            feat = seeded((8, 1024), 22, 0.3)
            calls = []
            orig = den.forward
            den.forward = lambda d, _o=orig, _c=calls: (_c.append(float(d['t'][0, 0])), _o(d))[1]
            ref.cfg.sampling_steps = 5
            xs, xfin = ref.score_agent.sample({'feat': feat}, den, 0.65)
            den.forward = orig
            G[f'ode_{name}_xs'], G[f'ode_{name}_x'] = xs.numpy(), xfin.numpy()
            G[f'ode_{name}_nfev'] = np.array(len(calls))
            G[f'ode_{name}_tcalls'] = np.array(calls)
        # ---- geometry helpers ---------------------------------------------------------------------------------
        Q = torch.nn.functional.normalize(seeded((3, 5, 7, 4), 23), dim=-1)
        W = seeded((3, 5, 7), 24).abs() + 0.1
        G['avgq_w'], G['avgq'] = ref_avgq(Q, W).numpy(), ref_avgq(Q).numpy()
        v = torch.as_tensor(assets['mano']['v_template'])[None] + seeded((2, 778, 3), 25, 0.002)
        pts, frame = VERT2ANCHOR(v)
        G['anchor_pts'], G['anchor_frame'] = pts.numpy(), frame.numpy()

        # ---- whole forward at cfg1 sizes ----------------------------------------------------------------------
        data = synth_batch(c['bs'], assets, seed=206)
        rec = {}
        ha, oa = ref.hoi_aggregator.hand_aggregator, ref.hoi_aggregator.obj_aggregator
        o1 = ha.select_topk_hand_by_observed_heatmap_and_fuse_by_index
        o2, o3 = oa.select_topk_object_by_heatmap, oa.select_topk_object_by_physics3

        def w1(**kw):
            r = o1(**kw)
            rec.setdefault('hand_topk', []).append(r['topk'].clone())
            rec.setdefault('hand_val', []).append(r['val'].clone())
            return r

        def w2(**kw):
            r = o2(**kw)
            rec.setdefault('obj_heat_topk', []).append(r[0].clone())
            return r

        def w3(**kw):
            r = o3(**kw)
            rec.setdefault('obj_phys_topk', []).append(r[0].clone())
            return r

        ha.select_topk_hand_by_observed_heatmap_and_fuse_by_index = w1
        oa.select_topk_object_by_heatmap, oa.select_topk_object_by_physics3 = w2, w3
        hp0 = ha.select_by_physics

        def w4(**kw):
            rec['hand_phys_pose_in'] = kw['pose'].clone()
            return hp0(**kw)

        ha.select_by_physics = w4
        torch.manual_seed(7)
        out = ref(dict(data), mode='predict')
    P = {}
    for k, v in out.items():
        a = v.numpy()
        if k in ('hand_heatmap', 'obj_heatmap'):
            a = a[:, :, ::2, ::2]
        P[k] = a
    for lvl in range(4):
        P[f'hand_topk_l{lvl}'] = rec['hand_topk'][lvl].numpy()
        P[f'hand_val_l{lvl}'] = rec['hand_val'][lvl].numpy()
    for i, nm in enumerate(['transl', 'rot', 'final']):
        P[f'obj_heat_topk_{nm}'] = rec['obj_heat_topk'][i].numpy()
    P['obj_phys_topk'] = rec['obj_phys_topk'][0].numpy()
    P['hand_phys_pose_in'] = rec['hand_phys_pose_in'].numpy()
    P['cfg'] = np.array([c['bs'], c['sample_num'], c['sampling_steps'], c['topk_hand'], c['topk_obj']])
    P['sample_T0'] = np.array(c['sample_T0'])
    # ---- contact detection (lib/utils/physics_fn.py:47-117,201-221) on a synthetic hand next to a box -------------------
    from lib.utils.physics_fn import detect_hand_and_object_contact
    rng = np.random.default_rng(41)
    hv = (assets['mano']['v_template'] + rng.normal(size=(778, 3)) * 0.001).astype(np.float64)
    hn = rng.normal(size=(778, 3)); hn /= np.linalg.norm(hn, axis=-1, keepdims=True)
    ov = (assets['ycb']['003_cracker_box']['verts'] * 0.6 + np.array([0.06, 0.0, 0.0])).astype(np.float64)
    on = rng.normal(size=ov.shape); on /= np.linalg.norm(on, axis=-1, keepdims=True)
    hc, oc, o2h = detect_hand_and_object_contact(hv, hn, ov, on, normal_distance_thresh=[-0.01, 0.01], vertical_distance_thresh=0.005)
    G['contact_hand'], G['contact_obj'], G['contact_o2h'] = hc, oc, o2h
    fcg = VERT2ANCHOR.get_force_contact(hc)
    G['contact_force'] = fcg
    G['contact_is_grasped'] = np.array(bool(VERT2ANCHOR.check_is_grasped(fcg)))
    # ---- TesterHand (lib/engine/test.py:585-680) on seeded joints / vertices -------------------------------------------
    rng = np.random.default_rng(31)
    gtj, gtv = rng.normal(size=(6, 21, 3)).astype(np.float32) * 0.05, rng.normal(size=(6, 778, 3)).astype(np.float32) * 0.05
    pdj = (gtj + rng.normal(size=gtj.shape) * 0.01).astype(np.float32)
    pdv = (gtv + rng.normal(size=gtv.shape) * 0.01).astype(np.float32)
    res = TesterHand()({'is_right': np.array([True, False, True, True, False, False]), 'gt_joint': gtj, 'pd_joint': pdj,
                        'gt_vert': gtv, 'pd_vert': pdv})
    for k in ('MJE', 'PA_MJE', 'MVE', 'PAMVE'):
        G['tester_' + k] = np.asarray(res[k]['both'], dtype=np.float64)
    G['tester_JE'] = np.stack([res[f'MJE_{i}']['both'] for i in range(21)], -1).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'golden_blocks.npz'), **G)
    np.savez_compressed(os.path.join(HERE, 'golden_predict.npz'), **P)
    for n in ('golden_blocks.npz', 'golden_predict.npz'):
        print(n, os.path.getsize(os.path.join(HERE, n)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
