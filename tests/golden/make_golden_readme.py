"""Golden vectors of the reference's whole ``vpho_net.forward(mode='predict')`` at the README evaluation config
(sample_num=100, sampling_steps=50, topk_hand=30, topk_obj=10, sample_T0=0.65 -- README.md:61-71) on ONE batch of 8 synthetic
images: every output the parity bar names, the index tensor AND the complete score vector of every top-k the aggregation makes
(torch.Tensor.topk is wrapped while the reference's forward runs), the 31 physics candidates, and the RHS-evaluation times of
both ODE solves (= scipy's accepted / rejected step sequence).

Weights: vpho_amd.synth.bench_state_dict(seed=1) -- heat-maps with the contrast of a trained head (candidate scores spread out:
rank ties below fp32 resolution are rare) and conditioned score networks (hypotheses contract towards a mode like a trained
model's: with purely random weights the T0=0.65 object hypotheses stay metres outside the crop, every heat score is exactly 0
and torch.topk's unspecified order among equal values decides the reference's own result).

Run in the build container only (needs /root/reference); same stubs / assets as make_golden.py.  The in-process trajectories
are not stored (size).

    python make_golden_readme.py             -> golden_predict_readme.npz    (8 images; also the CPU pin of the oracle)
    python make_golden_readme.py --bs 64     -> golden_predict_readme64.npz  (the benchmark's batch: the batch-coupled quirks Q3 / Q5 see
                                                64 images; hypotheses stored for every 4th sample, heat-maps every 8th pixel)
    python make_golden_readme.py --bs 64 --variant default,threads1,mkldnn_off
                                             -> golden_predict_readme64_selfcheck.npz: the REFERENCE's forward run again on the same
                                                inputs and the same prior draws under other execution settings of the same fp32
                                                arithmetic (1 intra-op thread instead of all; oneDNN convolutions / matmuls off = ATen's
                                                native kernels); only the 13 top-k index tensors and the three aggregated outputs are
                                                stored.  It answers "does the reference reproduce its own top-k lists?" with data."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

CFG = dict(bs=8, sample_num=100, sampling_steps=50, topk_hand=30, topk_obj=10, sample_T0=0.65, data_seed=4242, draw_seed=5)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=CFG['bs'])
    ap.add_argument('--variant', default=None, help="comma-separated: default, threads1, threads2, mkldnn_off -> the self-check fixture")
    args = ap.parse_args()
    big = args.bs != CFG['bs']
    CFG['bs'] = args.bs
    if big:
        CFG.update(data_seed=4264, draw_seed=6)
    hyp_stride, hm_stride = (4, 8) if big else (1, 4)
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import bench_state_dict, synth_batch
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_readme_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    c = CFG
    sys.argv = ['main.py', '--mode', 'eval', '--sample_num', str(c['sample_num']), '--sampling_steps', str(c['sampling_steps']),
                '--topk_hand', str(c['topk_hand']), '--topk_obj', str(c['topk_obj']), '--sample_T0', str(c['sample_T0'])]
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
    assert not missing, missing
    data = synth_batch(c['bs'], assets, seed=c['data_seed'])

    # ---- recorders ---------------------------------------------------------------------------------------------------
    topk_calls = []
    orig_topk = torch.Tensor.topk

    def rec_topk(self, *a, **kw):
        r = orig_topk(self, *a, **kw)
        topk_calls.append((self.detach().clone(), r[0].clone(), r[1].clone()))
        return r

    rec = {}
    ha = ref.hoi_aggregator.hand_aggregator
    orig_phys = ha.select_by_physics

    def phys(**kw):
        rec['phys_cand'] = kw['pose'].clone()
        return orig_phys(**kw)

    ha.select_by_physics = phys
    tcalls = {'hand': [], 'obj': []}
    for name, den in (('hand', ref.denoiser_hand), ('obj', ref.denoiser_obj)):
        den.forward = lambda d, _o=den.forward, _c=tcalls[name]: (_c.append(float(d['t'][0, 0])), _o(d))[1]

    if args.variant:
        return self_check(ref, data, c, args.variant.split(','), topk_calls, rec_topk, orig_topk)
    torch.manual_seed(c['draw_seed'])
    state = torch.get_rng_state()
    torch.Tensor.topk = rec_topk
    try:
        with torch.no_grad():
            out = ref(dict(data), mode='predict')
    finally:
        torch.Tensor.topk = orig_topk
    torch.set_rng_state(state)                                            # the prior draws the forward made (sde.py:26-28)
    nh = torch.randn(c['bs'] * c['sample_num'], 96)
    no = torch.randn(c['bs'] * c['sample_num'], 9)

    # order of the reference's topk calls (aggregation.py): 4 hand levels (:217,:246), object transl / rot (:777), physics3
    # (:987), final heat-map list (:777), then 5 fingers of select_by_physics (:596)
    assert len(topk_calls) == 4 + 2 + 2 + 5, len(topk_calls)
    P = dict(cfg=np.array([c['bs'], c['sample_num'], c['sampling_steps'], c['topk_hand'], c['topk_obj']]), sample_T0=np.array(c['sample_T0']),
             data_seed=np.array(c['data_seed']), noise_hand_crc=np.array(float(nh.double().sum())), noise_obj_crc=np.array(float(no.double().sum())),
             draw_seed=np.array(c['draw_seed']), hyp_stride=np.array(hyp_stride), hm_stride=np.array(hm_stride))
    for k in ('reg_hand_joint', 'force_local', 'agg_obj_6d', 'agg_hand_mano', 'agg_hand_joint', 'agg_hand_vert'):
        P[k] = out[k].numpy()
    for k in ('diff_final_hand_mano', 'diff_final_obj_6d'):
        P[k] = out[k].numpy()[:, ::hyp_stride]
    for k in ('hand_heatmap', 'obj_heatmap'):
        P[k] = out[k].numpy()[:, :, ::hm_stride, ::hm_stride]
    for lvl in range(4):
        sc, val, idx = topk_calls[lvl]
        P[f'hand_score_l{lvl}'], P[f'hand_val_l{lvl}'], P[f'hand_topk_l{lvl}'] = sc.numpy(), val.numpy(), idx.numpy()
    for i, nm in ((4, 'transl'), (5, 'rot'), (6, 'phys'), (7, 'heat')):
        sc, val, idx = topk_calls[i]
        P[f'obj_{nm}_score'], P[f'obj_{nm}_topk'] = sc.numpy(), idx.numpy()
    P['hand_phys_score'] = torch.stack([topk_calls[8 + f][0] for f in range(5)], 1).numpy()      # (bs,5,31)
    P['hand_phys_topk'] = torch.stack([topk_calls[8 + f][2] for f in range(5)], 1).numpy()       # (bs,5,5)
    P['hand_phys_cand'] = rec['phys_cand'].numpy()                                               # (bs,31,58)
    P['tcalls_hand'], P['tcalls_obj'] = np.array(tcalls['hand']), np.array(tcalls['obj'])
    path = os.path.join(HERE, 'golden_predict_readme64.npz' if big else 'golden_predict_readme.npz')
    np.savez_compressed(path, **P)
    print({k: v.shape for k, v in P.items()})
    print('nfev hand/obj', len(tcalls['hand']), len(tcalls['obj']), '|', os.path.getsize(path) // 1024, 'KiB')


TOPK_NAMES = ['hand_topk_l0', 'hand_topk_l1', 'hand_topk_l2', 'hand_topk_l3', 'obj_transl_topk', 'obj_rot_topk', 'obj_phys_topk',
              'obj_heat_topk'] + [f'hand_phys_topk_f{f}' for f in range(5)]


def self_check(ref, data, c, variants, topk_calls, rec_topk, orig_topk):
    """Same module, same inputs, same prior draws; only HOW the fp32 arithmetic is executed changes."""
    import contextlib
    import time
    n_default = torch.get_num_threads()
    P = dict(cfg=np.array([c['bs'], c['sample_num'], c['sampling_steps'], c['topk_hand'], c['topk_obj']]), sample_T0=np.array(c['sample_T0']),
             data_seed=np.array(c['data_seed']), draw_seed=np.array(c['draw_seed']), variants=np.array(len(variants)),
             threads_default=np.array(n_default))
    for vi, v in enumerate(variants):
        torch.set_num_threads({'threads1': 1, 'threads2': 2}.get(v, n_default))
        ctx = torch.backends.mkldnn.flags(enabled=False) if v == 'mkldnn_off' else contextlib.nullcontext()
        del topk_calls[:]
        torch.manual_seed(c['draw_seed'])
        torch.Tensor.topk = rec_topk
        t0 = time.time()
        try:
            with torch.no_grad(), ctx:
                out = ref(dict(data), mode='predict')
        finally:
            torch.Tensor.topk = orig_topk
        assert len(topk_calls) == 13, len(topk_calls)
        for nm, call in zip(TOPK_NAMES, topk_calls):
            P[f'v{vi}_{nm}'] = call[2].numpy().astype(np.int16)
        for k in ('agg_obj_6d', 'agg_hand_joint', 'agg_hand_vert'):
            P[f'v{vi}_{k}'] = out[k].numpy()
        P[f'v{vi}_code'] = np.array({'default': 0, 'threads1': 1, 'threads2': 2, 'mkldnn_off': 3}[v])
        print(v, 'done in %.0f s' % (time.time() - t0), flush=True)
    path = os.path.join(HERE, 'golden_predict_readme64_selfcheck.npz' if c['bs'] == 64 else 'golden_predict_readme_selfcheck.npz')
    np.savez_compressed(path, **P)
    print(sorted(P), os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
