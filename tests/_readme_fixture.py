"""Shared comparison against the reference's own forward at the README config (tests/golden/golden_predict_readme.npz, written by
tests/golden/make_golden_readme.py: 8 images in one batch, sample_num=100, sampling_steps=50, top-k 30/10, sample_T0=0.65).

Continuous outputs to a tolerance; every selected index list against the reference's through oracle/compare.py: indices equal
(every list the two sides' own score vectors force to be identical -- guaranteed_identical -- must be, when the tested side supplies
its score vectors), except where the two candidates' scores -- both read from the REFERENCE's own score vector -- differ by less than a FIXED bound
(oracle/compare.py: E2E_TIE_REL when each side ranks its own hypotheses).  The aggregated poses are asserted on every image whose
selections are identical, and -- for the HIP path -- on ALL images."""
import os

import numpy as np
import torch

from oracle.compare import parity_summary, TIE_REL, E2E_TIE_REL

R = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_predict_readme.npz'))
BS, S, STEPS, KH, KO = (int(v) for v in R['cfg'])
CFG = dict(sample_num=S, sampling_steps=STEPS, topk_hand=KH, topk_obj=KO, sample_T0=float(R['sample_T0']))


def use(name):
    """switch the module to another fixture of the same generator: 'golden_predict_readme.npz' (8 images, default) or
    'golden_predict_readme64.npz' (make_golden_readme.py --bs 64: the benchmark's batch; hypotheses / heat-maps stored strided)"""
    global R, BS, S, STEPS, KH, KO, CFG
    R = np.load(os.path.join(os.path.dirname(__file__), 'golden', name))
    BS, S, STEPS, KH, KO = (int(v) for v in R['cfg'])
    CFG = dict(sample_num=S, sampling_steps=STEPS, topk_hand=KH, topk_obj=KO, sample_T0=float(R['sample_T0']))


def inputs(assets):
    """The batch and the prior draws the reference's forward made (sde.py:26-28: hand first, then object)."""
    from vpho_amd.synth import synth_batch
    data = synth_batch(BS, assets, seed=int(R['data_seed']))
    state = torch.get_rng_state()
    torch.manual_seed(int(R['draw_seed']))
    nh, no = torch.randn(BS * S, 96), torch.randn(BS * S, 9)
    torch.set_rng_state(state)
    assert float(nh.double().sum()) == float(R['noise_hand_crc']) and float(no.double().sum()) == float(R['noise_obj_crc'])
    return data, nh, no


def reference_dbg():
    """The reference's selections in the layout of the oracle's dbg dict (oracle.aggregation.hoi_aggregate)."""
    t = lambda k: torch.as_tensor(np.asarray(R[k]))
    hand = dict(topk=[t(f'hand_topk_l{l}') for l in range(4)], val=[t(f'hand_val_l{l}') for l in range(4)],
                score=[t(f'hand_score_l{l}') for l in range(4)])
    return dict(hand=hand, hand_phys=dict(topk=t('hand_phys_topk'), score=t('hand_phys_score'), cand=t('hand_phys_cand')),
                transl_topk=t('obj_transl_topk'), rot_topk=t('obj_rot_topk'), phys_topk=t('obj_phys_topk'), heat_topk=t('obj_heat_topk'),
                transl_score=t('obj_transl_score'), rot_score=t('obj_rot_score'), phys_score=t('obj_phys_score'), heat_score=t('obj_heat_score'))


def oracle_as_tested(od):
    """The oracle's dbg dict in the layout of the HIP path's ``Engine.last_info['agg']`` (index tensors only)."""
    h = od['hand']
    return dict(hand_topk=[(i[:, :, None] if i.dim() == 2 else i).transpose(1, 2).int().contiguous() for i in h['topk']],
                hand_phys_topk=od['hand_phys']['topk'].int(), transl_topk=od['transl_topk'].int(), rot_topk=od['rot_topk'].int(),
                phys_topk=od['phys_topk'].int(), heat_topk=od['heat_topk'].int())


def compare(out, gd, upstream_tol, nfev=None, agg_tol=2e-4, min_identical=None, bound=E2E_TIE_REL, all_images_agg_tol=None):
    """out: output dict of the side under test (CPU or device tensors); gd: its selections (HIP layout).  Returns the summary.
    ``bound``: E2E_TIE_REL (each side ranks its own hypotheses) -- see oracle/compare.py; ``all_images_agg_tol``: when given, the
    aggregated outputs of EVERY image must agree to it (no waiver on the outputs the north star names)."""
    t = lambda a: torch.as_tensor(np.asarray(a))
    c = lambda v: v.detach().cpu()
    hs = int(R['hyp_stride']) if 'hyp_stride' in R else 1
    ms = int(R['hm_stride']) if 'hm_stride' in R else 4
    for k in ('reg_hand_joint', 'force_local', 'diff_final_hand_mano', 'diff_final_obj_6d'):
        got = c(out[k]).double()
        got = got.reshape(BS, S, -1)[:, ::hs] if k.startswith('diff_final') else got
        err = float((got.reshape(R[k].shape) - t(R[k]).double()).abs().max())
        # hypotheses whose rot6d columns are nearly parallel amplify the solver's rounding differences in Gram-Schmidt: one of the
        # 12 800 joint rotations of the fixture differs by 1.4e-4 (median 2e-7); 5e-4 stays inside the 1e-3 bar
        assert err < (5e-4 if k == 'diff_final_hand_mano' else upstream_tol), (k, err)
    for k in ('hand_heatmap', 'obj_heatmap'):
        assert float((c(out[k])[:, :, ::ms, ::ms].double() - t(R[k]).double()).abs().max()) < upstream_tol, k
    if nfev is not None:                                     # scipy's step sequence: 2 start-up + 6 per attempt + 1 denoise call
        assert tuple(nfev) == (len(R['tcalls_hand']), len(R['tcalls_obj'])), (nfev, len(R['tcalls_hand']), len(R['tcalls_obj']))
    ref = {k: t(R[k]) for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_hand_mano', 'agg_obj_6d')}
    res, rep = parity_summary(out, ref, gd, reference_dbg(), S, bound=bound)
    assert res['images_with_gap_above_tie_bound'] == 0 and res['max_rel_score_gap_at_first_differences'] <= bound, res
    if min_identical is not None:
        assert res['images_all_selections_identical'] >= min_identical, res
    for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_hand_mano', 'agg_obj_6d'):
        e = res[f'max_abs_{k}_where_identical']
        assert e is not None and e < agg_tol, (k, e, res)
        if all_images_agg_tol is not None and k != 'agg_hand_mano':      # axis-angle near pi flips sign: joints / vertices carry the bar
            assert res[f'max_abs_{k}_all'] < all_images_agg_tol, (k, res)
    return res
