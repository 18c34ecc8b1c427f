"""Selection-by-selection comparison of the HIP aggregation with the oracle's (TEST INFRASTRUCTURE -- see oracle/__init__.py;
used by tests/ and by bench.py's parity block only).

The aggregation (aggregation.py:1167-1353) is a chain of top-k selections over scores that are fp32 sums; the north star asks for
"top-k indices bit-exact".  Two candidates whose scores agree to the last bits of fp32 are ordered by the summation order of the
platform's kernels -- in the reference as well (torch.topk does not even define the order of EQUAL scores).  This module makes
that precise with a FIXED bound instead of a measured one:

* every selected index list of the side under test is compared with the oracle's;
* candidates that are exact copies of each other (the S regression candidates at cascade levels 1-3, aggregation.py:120-126 and
  :235-236; equal rows among the 31 physics candidates) count as one candidate;
* where the fused result only depends on the selected SET (weighted / unweighted means: levels 0-2, object lists, hand physics)
  the lists are compared as multisets; at level 3 rank i of every finger forms physics candidate i, so ranks are compared;
* for every remaining difference the GAP is reported: |score(got) - score(want)| / |score(want)| with BOTH scores taken from
  the oracle's own score vector.

Two FIXED bounds, by what the two sides were given:
* ``TIE_REL`` = 1e-6 -- both sides scored IDENTICAL candidates (the oracle's aggregation is fed the HIP path's own hypotheses,
  heat-maps and forces): only the fp32 rounding of FK / projection / bicubic sums differs (fp32 has 24 bits = 6e-8, a score is a
  sum of <= 60 products).  A gap above it is a wrong selection by the aggregation kernels.
* ``E2E_TIE_REL`` = 1e-3 -- end to end, each side scored its OWN hypotheses, which are reproduced to ~1e-5 (asserted <= 5e-4 by the
  callers): a score is an O(1) function of the hypothesis with slope <~ 10 / rad, so candidates closer than ~1e-3 relative can
  legitimately change order.  A top-k selection is discontinuous -- there is no end-to-end bound on its OUTPUT other than through
  this decomposition: (hypotheses agree to tolerance) x (selection exact on identical hypotheses).
"""
import torch

TIE_REL = 1e-6
E2E_TIE_REL = 1e-3
# Identical-candidate comparison on batches other than the pinned one (tests/test_gpu_seed_probe.py).  The hand cascade re-runs MANO
# FK on every candidate at every level and feeds each level's fused rotation to the next, so the level scores of the two sides see
# joints that differ by fp32 rounding (~2e-7 m -> ~1e-4 px after projection) on heat-maps with O(0.1)/px slopes: score differences of
# 1e-6 ... 1e-4 relative by level 3.  1e-6 holds on the pinned batch; over further batches the first differences reach 7.6e-5.
CASCADE_TIE_REL = 2e-4


def _c(t):
    return t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)


def _list_diff(got, want, score, ranked):
    """got, want: 1-d long tensors (aliased ids); score: oracle score per aliased id lookup (callable id -> float).
    Returns (n_diff, max_gap)."""
    if ranked:
        ne = got != want
        if not bool(ne.any()):
            return 0, 0.0
        # only the first differing rank can be a tie (a swap of two neighbours shows there; an insertion shifts every later rank)
        r = int(ne.nonzero()[0])
        a, b = int(got[r]), int(want[r])
        return int(ne.sum()), abs(score(a) - score(b)) / max(abs(score(b)), 1e-30)
    a, b = sorted(got.tolist()), sorted(want.tolist())
    if a == b:
        return 0, 0.0
    # multiset difference, paired in score order
    from collections import Counter
    ca, cb = Counter(a), Counter(b)
    only_a = sorted((ca - cb).elements(), key=score, reverse=True)
    only_b = sorted((cb - ca).elements(), key=score, reverse=True)
    gaps = [abs(score(x) - score(y)) / max(abs(score(y)), 1e-30) for x, y in zip(only_a, only_b)]
    return len(only_a), max(gaps)


def _margin(sc, lst, ranked):
    """smallest score distance a perturbation has to bridge to change the list: sc (C,), lst (k,).  Sets: min over (picked i, unpicked
    j, sc_i != sc_j) of |sc_i - sc_j|; ranked: the smallest positive gap between neighbours of the k + 1 best."""
    if ranked:
        d = torch.sort(sc, descending=True).values[:lst.numel() + 1]
        g = d[:-1] - d[1:]
        g = g[g > 0]
    else:
        m = torch.zeros(sc.numel(), dtype=torch.bool)
        m[lst] = True
        g = (sc[m][:, None] - sc[~m][None, :]).abs().reshape(-1)
        g = g[g > 0]
    return float(g.min()) if g.numel() else float('inf')


def guaranteed_identical(gd, od):
    """Which lists MUST come out identical on the two sides, from their own score vectors: if every candidate's score differs by at most d
    between the sides and the reference's list has a margin above 2 d, both top-k are the same list (order statistics of vectors that
    differ by <= d differ by <= d) -- no bound is chosen, d is the measured deviation of that (image, stage).  gd needs the tested
    side's complete score vectors (Engine.keep_states).  -> {stage: (bs,) bool}; a stage only counts when its dependencies do."""
    h, hp = od['hand'], od['hand_phys']
    bs = h['topk'][0].shape[0]
    pairs = {}
    for lvl in range(4):
        want, sc = h['topk'][lvl].long(), h['score'][lvl].double()
        if want.dim() == 2:
            want, sc = want[:, :, None], sc[:, :, None]
        pairs[f'hand_level{lvl}'] = (_c(gd['hand_score'][lvl]).double().reshape(sc.shape), sc, want, lvl == 3)
    for st, key, skey in (('obj_transl', 'transl_topk', 'transl_score'), ('obj_rot', 'rot_topk', 'rot_score'),
                          ('obj_physics', 'phys_topk', 'phys_score'), ('obj_heat', 'heat_topk', 'heat_score')):
        sc = od[skey].double()[:, :, None]
        pairs[st] = (_c(gd[skey]).double().reshape(sc.shape), sc, od[key].long()[:, :, None], False)
    sc = hp['score'].double().permute(0, 2, 1)
    pairs['hand_physics'] = (_c(gd['hand_phys_score']).double().reshape(sc.shape), sc, hp['topk'].long().permute(0, 2, 1), False)
    ok = {}
    for st, (got, ref, lst, ranked) in pairs.items():
        g = torch.ones(bs, dtype=torch.bool)
        for b in range(bs):
            for f in range(ref.shape[2]):
                d = float((got[b, :, f] - ref[b, :, f]).abs().max())
                g[b] &= _margin(ref[b, :, f], lst[b, :, f], ranked) > 2 * d
        ok[st] = g
    hand_chain = ['hand_level0', 'hand_level1', 'hand_level2', 'hand_level3']
    obj_chain = ['obj_transl', 'obj_rot', 'obj_heat']
    deps = {s_: hand_chain[:i] for i, s_ in enumerate(hand_chain)}
    deps.update({s_: obj_chain[:i] for i, s_ in enumerate(obj_chain)})
    deps['obj_physics'] = hand_chain + ['obj_transl', 'obj_rot']
    deps['hand_physics'] = hand_chain + obj_chain + ['obj_physics']
    out = {}
    for st in hand_chain + obj_chain + ['obj_physics', 'hand_physics']:
        g = ok[st].clone()
        for d_ in deps[st]:
            g &= out[d_]
        out[st] = g
    return out


def selection_report(gd, od, S):
    """gd: ``Engine.last_info['agg']`` of the HIP path (hand_topk[lvl] (bs,F,k) int32, hand_phys_topk (bs,5,kp), transl_topk ...);
    od: the oracle's ``dbg`` (oracle.aggregation.hoi_aggregate) for the SAME images.

    The selections form a chain: cascade level l+1 scores candidates that carry level l's fused joints; the object physics list
    scores against the fused hand; the hand physics list against the fused object.  Only the FIRST difference of an image along
    that chain can be a tie -- everything after it compares different candidate sets -- so the gap is evaluated there
    (``primary``), and later differences of the same image are counted as ``consequent``."""
    h = od['hand']
    bs = h['topk'][0].shape[0]
    stages = ['hand_level0', 'hand_level1', 'hand_level2', 'hand_level3', 'obj_transl', 'obj_rot', 'obj_heat', 'obj_physics', 'hand_physics']
    n_diff = {s: torch.zeros(bs, dtype=torch.long) for s in stages}
    gap = {s: torch.zeros(bs, dtype=torch.float64) for s in stages}
    for lvl in range(4):
        want = h['topk'][lvl].long()                                   # (bs,k) | (bs,k,5)
        sc = h['score'][lvl].double()                                  # (bs,2S) | (bs,2S,5)
        got = _c(gd['hand_topk'][lvl]).long()                          # (bs,F,k)
        if want.dim() == 2:
            want, sc = want[:, :, None], sc[:, :, None]
        got = got.transpose(1, 2)                                      # (bs,k,F)
        assert got.shape == want.shape, (lvl, got.shape, want.shape)
        st = f'hand_level{lvl}'
        for b in range(bs):
            for f in range(want.shape[2]):
                # the S regression candidates are exact copies of each other once level 0 has written the fused wrist into them
                alias = (lambda i: min(i, S)) if lvl >= 1 else (lambda i: i)
                gg = torch.tensor([alias(int(i)) for i in got[b, :, f]])
                ww = torch.tensor([alias(int(i)) for i in want[b, :, f]])
                n, g = _list_diff(gg, ww, lambda i, b=b, f=f: float(sc[b, i, f]), ranked=(lvl == 3))
                n_diff[st][b] += n
                gap[st][b] = max(float(gap[st][b]), g)
    # hand physics: top-kp of the 31 candidates per finger, fused by an unweighted mean -> set comparison; equal rows alias
    hp = od['hand_phys']
    want, sc, cand = hp['topk'].long(), hp['score'].double(), hp['cand']       # (bs,5,kp), (bs,5,31), (bs,31,58)
    got = _c(gd['hand_phys_topk']).long().reshape(want.shape)
    for b in range(bs):
        first, ali = {}, []
        for c in range(cand.shape[1]):
            ali.append(first.setdefault(cand[b, c, :48].numpy().tobytes(), c))
        for f in range(5):
            gg = torch.tensor([ali[int(i)] for i in got[b, f]])
            ww = torch.tensor([ali[int(i)] for i in want[b, f]])
            n, g = _list_diff(gg, ww, lambda i, b=b, f=f: float(sc[b, f, i]), ranked=False)
            n_diff['hand_physics'][b] += n
            gap['hand_physics'][b] = max(float(gap['hand_physics'][b]), g)
    # object lists: torch.topk leaves the order among EQUAL scores open (hypotheses outside the crop score exactly 0), and every
    # list is fused by a mean -> compare as sets, candidates with exactly equal scores count as one
    ko = od['transl_topk'].shape[1]
    g_t, g_r = _c(gd['transl_topk']).long().reshape(bs, ko), _c(gd['rot_topk']).long().reshape(bs, ko)
    w_t, w_r = od['transl_topk'].long(), od['rot_topk'].long()
    for st, key, skey in (('obj_transl', 'transl_topk', 'transl_score'), ('obj_rot', 'rot_topk', 'rot_score'),
                          ('obj_physics', 'phys_topk', 'phys_score'), ('obj_heat', 'heat_topk', 'heat_score')):
        want, sc = od[key].long(), od[skey].double()
        got = _c(gd[key]).long().reshape(want.shape)
        for b in range(bs):
            if st in ('obj_physics', 'obj_heat'):
                # candidate c of the 10 x 10 cross product is (translation list[c // 10], rotation list[c % 10]) (aggregation.py:
                # 1235-1242): the label depends on the ORDER of the two lists, the candidate does not -> relabel the tested side's
                # picks in the oracle's numbering (possible when both lists hold the same hypotheses; otherwise the difference is
                # consequent and the labels are compared as they are)
                pos_t = {int(v): i for i, v in enumerate(w_t[b])}
                pos_r = {int(v): i for i, v in enumerate(w_r[b])}
                rel = []
                for cidx in got[b].tolist():
                    ti, ri = int(g_t[b, cidx // ko]), int(g_r[b, cidx % ko])
                    rel.append(pos_t[ti] * ko + pos_r[ri] if (ti in pos_t and ri in pos_r) else cidx)
                got_b = torch.tensor(rel)
            else:
                got_b = got[b]
            first = {}
            ali = [first.setdefault(float(sc[b, i]), i) for i in range(sc.shape[1])]
            gg = torch.tensor([ali[int(i)] for i in got_b])
            ww = torch.tensor([ali[int(i)] for i in want[b]])
            n, g = _list_diff(gg, ww, lambda i, b=b: float(sc[b, i]), ranked=False)
            n_diff[st][b] += n
            gap[st][b] = g
    # dependency chains: which earlier stages must be identical for a stage's gap to mean anything
    hand_chain = ['hand_level0', 'hand_level1', 'hand_level2', 'hand_level3']
    obj_chain = ['obj_transl', 'obj_rot', 'obj_heat']
    deps = {s: hand_chain[:i] for i, s in enumerate(hand_chain)}
    deps.update({s: obj_chain[:i] for i, s in enumerate(obj_chain)})
    deps['obj_physics'] = hand_chain + ['obj_transl', 'obj_rot']
    deps['hand_physics'] = hand_chain + obj_chain + ['obj_physics']
    primary_gap = torch.zeros(bs, dtype=torch.float64)
    detail = {}
    for s in stages:
        clean_before = torch.ones(bs, dtype=torch.bool)
        for d_ in deps[s]:
            clean_before &= n_diff[d_] == 0
        prim = clean_before & (n_diff[s] > 0)
        primary_gap = torch.where(prim, torch.maximum(primary_gap, gap[s]), primary_gap)
        detail[s] = dict(images_primary=int(prim.sum()), max_gap_primary=float(gap[s][prim].max()) if bool(prim.any()) else 0.0,
                         images_consequent=int(((n_diff[s] > 0) & ~clean_before).sum()))
    hand_diff = sum(n_diff[s] for s in hand_chain + ['hand_physics'])
    obj_diff = sum(n_diff[s] for s in obj_chain + ['obj_physics'])
    rep = dict(hand_differences_per_image=hand_diff, object_differences_per_image=obj_diff, primary_gap_per_image=primary_gap,
               detail=detail)
    if 'hand_score' in gd:
        # lists the two sides' own scores force to be identical (guaranteed_identical): a difference there is a selection error
        g = guaranteed_identical(gd, od)
        rep['guaranteed_lists'] = int(sum(int(v.sum()) for v in g.values()))
        rep['guaranteed_images'] = int(torch.stack(list(g.values())).all(0).sum())
        rep['guaranteed_but_different'] = {st: int((g[st] & (n_diff[st] > 0)).sum()) for st in stages if bool((g[st] & (n_diff[st] > 0)).any())}
    return rep


def parity_summary(out, ref, gd, od, S, bound=TIE_REL):
    """Numbers for bench.py's ``parity`` block and the README-size tests: ``out`` outputs of the side under test, ``ref`` the
    oracle's (or the reference fixture's) for the same images / prior draws; gd / od their selections; ``bound``: TIE_REL when
    both sides were given identical candidates, E2E_TIE_REL end to end."""
    TIE_REL = bound
    rep = selection_report(gd, od, S)
    hand_clean = rep['hand_differences_per_image'] == 0
    obj_clean = rep['object_differences_per_image'] == 0
    all_clean = hand_clean & obj_clean
    pg = rep['primary_gap_per_image']
    d = lambda k: (_c(out[k]).double() - _c(ref[k]).double()).abs().reshape(hand_clean.shape[0], -1).amax(1)
    res = dict(images=int(hand_clean.shape[0]), images_all_selections_identical=int(all_clean.sum()),
               images_hand_selection_identical=int(hand_clean.sum()), images_object_selection_identical=int(obj_clean.sum()),
               images_first_difference_is_a_tie=int(((~all_clean) & (pg <= TIE_REL)).sum()),
               # first difference with a relative score gap above the bound.  Given identical candidates the bound is 1e-6 -- BELOW the fp32
               # rounding noise of the scores themselves (hand_level3: 1e-5 of the score scale) --, so a count here is
               # not a wrong pick: whether a pick lies within the scores' own noise is what the fp64 referee decides (oracle/referee.py,
               # tests/test_gpu_referee.py).  (Named images_with_wrong_selection until round 4.)
               images_with_gap_above_tie_bound=int((pg > TIE_REL).sum()),
               max_rel_score_gap_at_first_differences=float(pg.max()), tie_bound=TIE_REL, per_stage=rep['detail'])
    for k in ('guaranteed_lists', 'guaranteed_images', 'guaranteed_but_different'):
        if k in rep:
            res[k] = rep[k]
    for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_hand_mano'):
        e = d(k)
        res[f'max_abs_{k}_where_identical'] = float(e[hand_clean].max()) if bool(hand_clean.any()) else None
        res[f'max_abs_{k}_all'] = float(e.max())
    e = d('agg_obj_6d')
    res['max_abs_agg_obj_6d_where_identical'] = float(e[obj_clean].max()) if bool(obj_clean.any()) else None
    res['max_abs_agg_obj_6d_all'] = float(e.max())
    worst = torch.stack([d(k) for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_obj_6d')]).amax(0)
    res['images_within_1e-3_on_joints_vertices_6dof'] = int((worst <= 1e-3).sum())
    mp = (_c(out['agg_hand_joint']).double() - _c(ref['agg_hand_joint']).double()).norm(dim=-1).mean(-1) * 1000
    res['mpjpe_delta_mm_all'] = float(mp.mean())
    res['mpjpe_delta_mm_where_identical'] = float(mp[hand_clean].mean()) if bool(hand_clean.any()) else None
    return res, rep


SELFCHECK_VARIANTS = {0: 'default (all intra-op threads, oneDNN on)', 1: '1 intra-op thread', 2: '2 intra-op threads', 3: 'oneDNN off (ATen native kernels)'}
SELFCHECK_LISTS = ['hand_topk_l0', 'hand_topk_l1', 'hand_topk_l2', 'hand_topk_l3', 'obj_transl_topk', 'obj_rot_topk', 'obj_phys_topk',
                   'obj_heat_topk'] + [f'hand_phys_topk_f{f}' for f in range(5)]


def reference_self_agreement(path):
    """How well the REFERENCE reproduces its own result: tests/golden/golden_predict_readme64_selfcheck.npz holds the 13 top-k index
    tensors and the three aggregated outputs of the reference's own ``forward('predict')`` (README config, the 64-image batch of
    golden_predict_readme64.npz, identical inputs and prior draws) under several execution settings of the same fp32 arithmetic
    (tests/golden/make_golden_readme.py --variant ...).  Returns, per variant against the default run: images on which every one of the
    13 lists is identical, images within 1e-3 on joints / vertices / 6-DoF, largest output difference, per-list counts."""
    import numpy as np
    P = np.load(path)
    n = int(P['cfg'][0])
    rep = {'images': n, 'what': "the reference's own forward('predict') re-run on identical inputs and prior draws; each variant compared "
                                "with its default run (variant 0)", 'variants': {}}
    for v in range(1, int(P['variants'])):
        same = np.ones(n, bool)
        per = {}
        for nm in SELFCHECK_LISTS:
            eq = (P[f'v0_{nm}'].reshape(n, -1) == P[f'v{v}_{nm}'].reshape(n, -1)).all(1)
            per[nm] = int(eq.sum())
            same &= eq
        d = {k: np.abs(P[f'v0_{k}'].astype(np.float64) - P[f'v{v}_{k}'].astype(np.float64)).reshape(n, -1).max(1)
             for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_obj_6d')}
        worst = np.max(np.stack(list(d.values())), 0)
        mp = np.linalg.norm(P['v0_agg_hand_joint'].astype(np.float64) - P[f'v{v}_agg_hand_joint'].astype(np.float64), axis=-1).mean(-1) * 1000
        rep['variants'][SELFCHECK_VARIANTS[int(P[f'v{v}_code'])]] = dict(
            images_all_selections_identical=int(same.sum()), **{'images_within_1e-3_on_joints_vertices_6dof': int((worst <= 1e-3).sum())},
            max_abs={k: float(x.max()) for k, x in d.items()}, mpjpe_delta_mm_all=float(mp.mean()), images_identical_per_list=per)
    return rep
