"""fp64 referee of the selection chain (TEST INFRASTRUCTURE -- see oracle/__init__.py; used by tests/ and bench.py's parity block).

The aggregation (aggregation.py:1167-1353) is a chain of nine top-k selections over fp32 scores.  Two fp32 implementations of the
same score differ in their last bits, so whenever two candidates around rank k score within that rounding noise the two sides may pick
different lists -- and because every later stage is fed the fused pick, "identical lists" cannot be asked of ANY second implementation
(the reference on another BLAS differs the same way).  What CAN be asked, and is decidable without comparing two fp32 roundings with
each other, is that a pick is as good as the reference's own arithmetic allows.  This module decides that with a kernel-independent
judge:

* For every stage the candidates the side under test actually scored (its hypotheses, its fused joints of the level before, its
  forces) are taken as exact inputs, and the stage's score function (aggregation.py:196-218,242-248 hand levels; :753-777 object
  heat; :958-987 object physics; :553-592 hand physics) is evaluated on them in **fp64** -> s64 -- FK, projection, bicubic
  look-up, sums all in double -- and in the reference's **fp32** arithmetic (the oracle's restatement) -> s32.
* regret(list) = how much worse, in fp64, the worst pick is than the true k-th best:  max(0, s64_(k) - min_{i in list} s64_i);
  for the one list that is consumed by RANK (hand level 3: rank i of every finger forms physics candidate i,
  aggregation.py:1297-1312) max_r |s64_(r) - s64_{list[r]}|.  A list is *optimal* when its regret is 0 (exact copies and exactly
  equal scores are interchangeable by construction).
* eps32[b, f] = max over the candidates of ONE (image, finger) score vector of |s32 - s64|: the rounding noise of the reference's own
  arithmetic on THAT vector (round 3 used the maximum over the whole batch for every image: a mis-selection on a well-resolved image
  then passed if another image of the batch was noisy -- ADVICE r3).  ANY top-k taken from scores that are within eps of the truth has
  regret <= 2 eps (order statistics of two vectors that differ by <= eps differ by <= eps).  eps_own[b, f] is the same quantity for
  the score vector the side under test really ranked.
* Per (image, finger): `regret <= 2 max(eps32, eps_own)` and, where the list under test differs from the oracle's fp32 list on the
  same candidates, the fp64 scores of the exchanged candidates within `2 max(eps32, eps_own)` of each other -- both are theorems for
  a correct top-k of scores with those errors, so a violation is a selection bug, whatever the data.  That the tested side's error is
  itself of the size of the reference's is asserted separately (`eps_own_over_eps32`), and `within_reference_noise` reports the
  stricter statement `regret <= 2 eps32[b, f]` (the pick is one the reference's own arithmetic could have produced on this vector).

Everything is reported relative to the score scale of its (image, finger): scale = max_c |s64|.
"""
import torch

from . import aggregation as A

STAGES = ('hand_level0', 'hand_level1', 'hand_level2', 'hand_level3', 'obj_transl', 'obj_rot', 'obj_heat', 'obj_physics', 'hand_physics')
RANKED = ('hand_level3',)


def _c(t, dtype=None):
    t = t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)
    return t if dtype is None else t.to(dtype)


def record_from_oracle(out, info, data):
    """selection record of an oracle run (oracle.vpho.predict): what every stage scored and what it picked"""
    f, d = info['features'], info['agg']
    h, hp = d['hand'], d['hand_phys']
    lists = {f'hand_level{l}': (h['topk'][l] if h['topk'][l].dim() == 3 else h['topk'][l][:, :, None]).long() for l in range(4)}
    lists.update(obj_transl=d['transl_topk'][:, :, None].long(), obj_rot=d['rot_topk'][:, :, None].long(),
                 obj_heat=d['heat_topk'][:, :, None].long(), obj_physics=d['phys_topk'][:, :, None].long(),
                 hand_physics=hp['topk'].permute(0, 2, 1).long())
    sc = {f'hand_level{l}': (h['score'][l] if h['score'][l].dim() == 3 else h['score'][l][:, :, None]) for l in range(4)}
    sc.update(obj_transl=d['transl_score'][:, :, None], obj_rot=d['rot_score'][:, :, None], obj_heat=d['heat_score'][:, :, None],
              obj_physics=d['phys_score'][:, :, None], hand_physics=hp['score'].permute(0, 2, 1))
    return dict(lists=lists, scores=sc, cascade_state=[_c(s) for s in h['state']], betas=_c(f['mano_shape']), hand_heatmap=_c(f['hand_heatmap']),
                obj_heatmap=_c(f['obj_heatmap']), force_local=_c(f['force_local']), obj_pose=_c(out['diff_final_obj_6d']),
                transl=_c(d['transl']), cand=_c(d['pose6d_candidate']), force_point=_c(d['force_point']), force_global=_c(d['force_global']),
                cand58=_c(hp['cand']), obj_vert=_c(d['obj_vert']), data=data)


def record_from_hip(out, info, data):
    """selection record of a HIP run: ``Engine.last_info`` with ``Engine.keep_states = True`` + the output dict"""
    f, g = info['features'], info['agg']
    assert 'cascade_state' in g, 'run the engine with keep_states = True (the candidates of every cascade level are kept)'
    bs = g['transl_topk'].shape[0]
    lists = {f'hand_level{l}': _c(g['hand_topk'][l]).long().permute(0, 2, 1) for l in range(4)}              # (bs,F,k) -> (bs,k,F)
    for st, key in (('obj_transl', 'transl_topk'), ('obj_rot', 'rot_topk'), ('obj_heat', 'heat_topk'), ('obj_physics', 'phys_topk')):
        lists[st] = _c(g[key]).long().reshape(bs, -1)[:, :, None]
    lists['hand_physics'] = _c(g['hand_phys_topk']).long().reshape(bs, 5, -1).permute(0, 2, 1)
    cpu_data = {k: (_c(v) if torch.is_tensor(v) else v) for k, v in data.items()}
    sc = {f'hand_level{l}': _c(g['hand_score'][l]) for l in range(4)}                                          # (bs,C,F) as ranked by the kernel
    sc.update(obj_transl=_c(g['transl_score'])[:, :, None], obj_rot=_c(g['rot_score'])[:, :, None], obj_heat=_c(g['heat_score'])[:, :, None],
              obj_physics=_c(g['phys_score'])[:, :, None], hand_physics=_c(g['hand_phys_score']))
    return dict(lists=lists, scores=sc, cascade_state=[_c(s) for s in g['cascade_state']], betas=_c(f['mano_shape']), hand_heatmap=_c(out['hand_heatmap']),
                obj_heatmap=_c(out['obj_heatmap']), force_local=_c(out['force_local']), obj_pose=_c(out['diff_final_obj_6d']),
                transl=_c(g['transl']), cand=_c(g['pose6d_candidate']), force_point=_c(g['force_point']), force_global=_c(g['force_global']),
                cand58=_c(g['cand58']), obj_vert=_c(g['obj_vert']), data=cpu_data)


def stage_scores(assets, anchor_skeleton, rec, stage, dtype, chunk=8):
    """scores of ALL candidates of ``stage`` on the record's own candidates, arithmetic in ``dtype`` -> (bs, C, F)"""
    d = rec['data']
    t = lambda x: _c(x, dtype)
    bs = rec['betas'].shape[0]
    outs = []
    for b0 in range(0, bs, chunk):
        sl = slice(b0, min(b0 + chunk, bs))
        K, isr, names = t(d['cam_intr_crop_flip'])[sl], _c(d['is_right']).bool()[sl], list(d['obj_name'])[sl]
        if stage.startswith('hand_level'):
            lvl = int(stage[-1])
            s = A.hand_level_scores(assets['mano'], t(rec['cascade_state'][lvl])[sl], t(rec['betas'])[sl], t(d['root_joint_flip'])[sl], K,
                                    t(rec['hand_heatmap'])[sl], t(d['bbox_hand'])[sl], lvl)
            s = s[:, :, None] if s.dim() == 2 else s
        elif stage in ('obj_transl', 'obj_rot', 'obj_heat'):
            if stage == 'obj_heat':
                pose = rec['cand'][sl]
            else:
                pose = rec['obj_pose'][sl].clone()
                if stage == 'obj_rot':
                    pose[..., 6:] = rec['transl'][sl][:, None].to(pose.dtype)
            s = A.obj_heat_scores(assets['ycb'], pose, t(d['root_joint'])[sl], names, isr, K, t(rec['obj_heatmap'])[sl],
                                  t(d['bbox_obj_rect'])[sl], dtype=dtype)[:, :, None]
        elif stage == 'obj_physics':
            s = A.obj_physics_scores(assets['ycb'], rec['cand'][sl], t(d['root_joint'])[sl], names, isr, t(rec['force_point'])[sl],
                                     t(rec['force_global'])[sl], dtype=dtype)[:, :, None]
        elif stage == 'hand_physics':
            s = A.hand_physics_scores(assets['mano'], assets['anchor'], anchor_skeleton, t(rec['cand58'])[sl], t(d['root_joint_flip'])[sl],
                                      t(rec['force_local'])[sl], t(rec['obj_vert'])[sl]).permute(0, 2, 1)
        else:
            raise KeyError(stage)
        outs.append(s)
    return torch.cat(outs, 0)


def _regret(s64, lst, ranked):
    """s64 (C,), lst (k,) -> regret >= 0 (absolute)"""
    k = lst.numel()
    top = torch.sort(s64, descending=True).values[:k]
    picked = s64[lst]
    if ranked:
        return float((top - picked).abs().max())
    return float((top[-1] - picked.min()).clamp(min=0))


def _exchange_gap(s64, a, b):
    """largest fp64 score distance between the candidates two lists exchanged (paired in score order); 0 for equal multisets"""
    from collections import Counter
    ca, cb = Counter(a.tolist()), Counter(b.tolist())
    oa = sorted((ca - cb).elements(), key=lambda i: -float(s64[i]))
    ob = sorted((cb - ca).elements(), key=lambda i: -float(s64[i]))
    return max([abs(float(s64[x]) - float(s64[y])) for x, y in zip(oa, ob)], default=0.0)


def referee(assets, anchor_skeleton, rec, stages=STAGES):
    """-> {stage: dict(eps32_rel, bound_rel, regret_rel (bs,), regret32_rel (bs,), optimal (bs,) bool, optimal32 (bs,) bool,
    differs_from_o32 (bs,) bool, exchange_gap_rel (bs,))}: regrets of the record's lists and of the fp32 oracle's lists on the SAME
    candidates, both judged by the fp64 scores."""
    rep = {}
    for st in stages:
        s64 = stage_scores(assets, anchor_skeleton, rec, st, torch.float64)
        s32 = stage_scores(assets, anchor_skeleton, rec, st, torch.float32)
        lst = rec['lists'][st]                                              # (bs,k,F)
        bs, k, F = lst.shape
        assert s64.shape[0] == bs and s64.shape[2] == F, (st, s64.shape, lst.shape)
        scale = s64.abs().amax(1).clamp(min=1e-300)                         # (bs,F)
        eps_rel = ((s32.double() - s64).abs().amax(1) / scale)              # (bs,F)
        reg, reg32 = torch.zeros(bs, F, dtype=torch.float64), torch.zeros(bs, F, dtype=torch.float64)
        gap = torch.zeros(bs, F, dtype=torch.float64)
        differs = torch.zeros(bs, dtype=torch.bool)
        _, l32 = A.topk_stable(s32, k, dim=1)
        ranked = st in RANKED
        for b in range(bs):
            for f in range(F):
                reg[b, f] = _regret(s64[b, :, f], lst[b, :, f], ranked)
                reg32[b, f] = _regret(s64[b, :, f], l32[b, :, f], ranked)
                same = torch.equal(lst[b, :, f], l32[b, :, f]) if ranked else sorted(lst[b, :, f].tolist()) == sorted(l32[b, :, f].tolist())
                if not same:
                    differs[b] = True
                    if ranked:
                        ne = lst[b, :, f] != l32[b, :, f]
                        gap[b, f] = (s64[b, lst[b, ne, f], f] - s64[b, l32[b, ne, f], f]).abs().max()
                    else:
                        gap[b, f] = _exchange_gap(s64[b, :, f], lst[b, :, f], l32[b, :, f])
        eps = float(eps_rel.max())
        own = rec.get('scores', {}).get(st)
        eps_own, own_topk, eps_own_bf = None, None, None
        if own is not None:                    # the side under test's own score vectors: their fp64 error, and list == top-k of them
            own = own.reshape(s64.shape)
            eps_own_bf = (own.double() - s64).abs().amax(1) / scale
            eps_own = float(eps_own_bf.max())
            o = torch.where(torch.isnan(own), torch.full_like(own, float('inf')), own)
            own_topk = bool(torch.equal(A.topk_stable(o, k, dim=1)[1], lst))
        reg_bf, gap_bf = reg / scale, gap / scale
        noise_bf = eps_rel if eps_own_bf is None else torch.maximum(eps_rel, eps_own_bf)
        rep[st] = dict(eps32_rel=eps, bound_rel=2 * eps, eps_own_rel=eps_own, list_is_topk_of_own_scores=own_topk, regret_rel=reg_bf.amax(1), regret32_rel=(reg32 / scale).amax(1),
                       optimal=(reg == 0).all(1), optimal32=(reg32 == 0).all(1), differs_from_o32=differs,
                       exchange_gap_rel=gap_bf.amax(1),
                       # per (image, finger), relative to that vector's own score scale
                       eps32_bf=eps_rel, eps_own_bf=eps_own_bf, regret_bf=reg_bf, exchange_gap_bf=gap_bf,
                       within_own_and_reference_noise_bf=(reg_bf <= 2 * noise_bf) & (gap_bf <= 2 * noise_bf),
                       within_reference_noise_bf=(reg_bf <= 2 * eps_rel) & (gap_bf <= 2 * eps_rel))
    return rep


def summary(rep):
    """bench.py / test print-out: per stage and overall"""
    bs = next(iter(rep.values()))['optimal'].shape[0]
    all_opt, all_opt32 = torch.ones(bs, dtype=torch.bool), torch.ones(bs, dtype=torch.bool)
    per = {}
    ok = True
    for st, r in rep.items():
        all_opt &= r['optimal']
        all_opt32 &= r['optimal32']
        within = bool(r['within_own_and_reference_noise_bf'].all())
        ok &= within
        ratio = None if r['eps_own_bf'] is None else (r['eps_own_bf'] / r['eps32_bf'].clamp(min=1e-300))
        per[st] = dict(eps32_rel=r['eps32_rel'], eps_tested_rel=r.get('eps_own_rel'), regret_max_rel=float(r['regret_rel'].max()), regret32_max_rel=float(r['regret32_rel'].max()),
                       vectors=int(r['eps32_bf'].numel()), vectors_within_2eps32_of_their_own_vector=int(r['within_reference_noise_bf'].sum()),
                       eps_tested_over_eps32_median_max=None if ratio is None else [float(ratio.median()), float(ratio.max())],
                       images_optimal=int(r['optimal'].sum()), images_optimal_fp32_reference=int(r['optimal32'].sum()),
                       images_list_differs_from_fp32_reference=int(r['differs_from_o32'].sum()),
                       exchange_gap_max_rel=float(r['exchange_gap_rel'].max()), within_reference_noise=within)
    return dict(images=bs, images_identical_to_fp64_order=int(all_opt.sum()), images_identical_to_fp64_order_fp32_reference=int(all_opt32.sum()),
                regret_max_rel=max(p['regret_max_rel'] for p in per.values()),
                regret_max_rel_fp32_reference=max(p['regret32_max_rel'] for p in per.values()),
                all_within_reference_noise=ok, per_stage=per)
