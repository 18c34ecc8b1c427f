"""Shared comparison against the reference's README-size forward (tests/golden/golden_predict_readme.npz): continuous outputs
to a tolerance, selection indices exactly except where two neighbouring candidates tie within fp32 resolution."""
import os

import numpy as np
import torch

R = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_predict_readme.npz'))
CFG = dict(sample_num=100, sampling_steps=50, topk_hand=30, topk_obj=10, sample_T0=0.2)


def compare(out, hand_topk, obj_topk, upstream_tol, obj_scores=None, hand_val=None):
    """out: dict of CPU tensors; hand_topk: list of 4 index tensors in the reference's layout (bs,k[,5]);
    obj_topk: dict transl / rot / final / phys; obj_scores: the same keys -> (bs, n) score vectors of the side under test
    (torch.topk leaves the order among EQUAL scores unspecified -- with random weights at T0=0.65 most object hypotheses
    project outside the crop and score exactly 0 -- so object selections are compared rank by rank through their scores).
    hand_val: the tested side's top-k values per level (same layout as hand_topk), used to measure the score noise between the
    two sides.  Returns the number of images whose hand cascade selected a different index somewhere."""
    t = lambda a: torch.as_tensor(np.asarray(a))
    for k in ('reg_hand_joint', 'force_local', 'diff_final_hand_mano', 'diff_final_obj_6d'):
        err = float((out[k].double() - t(R[k]).double()).abs().max())
        assert err < upstream_tol, (k, err)
    for k in ('hand_heatmap', 'obj_heatmap'):
        assert float((out[k][:, :, ::4, ::4].double() - t(R[k]).double()).abs().max()) < upstream_tol, k
    bs = R['agg_obj_6d'].shape[0]
    swaps = torch.zeros(bs, dtype=torch.long)
    for lvl in range(4):
        want, val = t(R[f'hand_topk_l{lvl}']).long(), t(R[f'hand_val_l{lvl}'])
        got = hand_topk[lvl].long().reshape(want.shape)
        ne = got != want
        # What changes the fused pose: candidates sample_num .. 2*sample_num-1 are identical copies of the regression pose, so
        # they count as one; at levels 0-2 only the selected SET matters (a weighted mean); at level 3 the rank matters too
        # (rank i of every finger forms physics candidate i)
        S = CFG['sample_num']
        g2, w2 = got.clamp(max=S), want.clamp(max=S)
        if lvl < 3:
            g2, w2 = g2.sort(dim=1).values, w2.sort(dim=1).values
        swaps += (g2 != w2).reshape(bs, -1).sum(1)
        # top-k VALUES must agree rank by rank; an index may differ only inside a run of tied values (e.g. the 100 identical
        # regression candidates: torch.topk's order among equal scores is unspecified) -- tied = closer to a neighbouring rank
        # than 4x the score noise between the two sides; the last rank may tie with the first unselected candidate
        if hand_val is not None:
            mine = hand_val[lvl].reshape(want.shape).to(val.dtype)
            noise = float((mine - val).abs().max())
            assert noise < 2e-3 * float(val.abs().max()), ('hand level', lvl, noise)
        else:
            noise = 2.5e-6 * float(val.abs().max())
        inf = torch.full_like(val[:, :1], float('inf'))
        prev_gap = torch.cat([inf, (val[:, 1:] - val[:, :-1]).abs()], 1)
        next_gap = torch.cat([(val[:, 1:] - val[:, :-1]).abs(), torch.zeros_like(val[:, :1])], 1)
        distinct = torch.minimum(prev_gap, next_gap) > 4 * noise
        assert not (ne & distinct).any(), ('hand level', lvl, (ne & distinct).nonzero().tolist()[:5])
    for k, name in (('transl', 'obj_heat_topk_transl'), ('rot', 'obj_heat_topk_rot'), ('final', 'obj_heat_topk_final'), ('phys', 'obj_phys_topk')):
        got, want = obj_topk[k].long().reshape(R[name].shape), t(R[name]).long()
        if obj_scores is None:
            assert torch.equal(got, want), k
        else:
            sc = obj_scores[k].double().reshape(bs, -1)
            a, b = torch.gather(sc, 1, got), torch.gather(sc, 1, want)
            assert float((a - b).abs().max()) <= 1e-6 * float(sc.abs().max()) + 1e-12, (k, got.tolist(), want.tolist())
    assert float((out['agg_obj_6d'].double() - t(R['agg_obj_6d']).double()).abs().max()) < 2e-5
    # a swap between two DIFFERENT candidates whose scores tie within the noise (accepted above) changes the fused pose, and
    # which way such a tie falls depends on the summation order of the platform's fp32 kernels -- the reference's own result is
    # not reproducible across machines there; the aggregated poses are compared on the images without such a swap
    clean = swaps == 0
    for k in ('agg_hand_mano', 'agg_hand_joint', 'agg_hand_vert'):
        if clean.any():
            err = float((out[k].double() - t(R[k]).double())[clean].abs().max())
            assert err < 2e-4, (k, err)
    return int((~clean).sum())
