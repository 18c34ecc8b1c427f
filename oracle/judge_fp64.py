"""End-to-end fp64 judge (TEST INFRASTRUCTURE -- see oracle/__init__.py; used by tests/ and scripts/e2e_fp64.py only).

``vpho_net.forward(mode='predict')`` (VPHO.py:90-304) ends in nine top-k selections (aggregation.py:1167-1353), so two fp32
implementations of it cannot be asked for identical outputs on every image: where two candidates score within rounding of each other the
lists differ, and the fused result with them.  oracle/referee.py judges every selection GIVEN its candidates, oracle/sampler_fp64.py the
hypotheses given the encodings, bench.py's features_vs_fp64 the encodings -- this module chains the three into ONE float64 ``predict``:
feature path, both probability-flow solves, rot6d -> axis-angle, the whole aggregation, all in double.  The only thing taken from an fp32
run is the ACCEPTED STEP SEQUENCE of its two RK45 solves (the controller is part of the algorithm; its decisions are discrete; rows are
independent on a fixed sequence -- oracle/sampler_fp64.py), so every fp32 side is held against the exact evaluation of the very scheme
it ran:

    truth(side) = predict in float64 on side's step sequences;     side is "within 1e-3 of the truth" on an image when its 21 joints,
    778 vertices and object 6-DoF all are.

Reported next to each other for the HIP path and for the fp32 oracle (= the reference's arithmetic) this says which of the two is closer
to what the algorithm computes in exact arithmetic, image by image -- the question "the two fp32 sides differ on 2-6 images of 64: whose
rounding is it?" has no other decidable form.  For every image on which the two fp32 sides differ by more than 1e-3 the first selection
that differs is located, with the fp64 margin of that list (how far the scores would have to move to change it) and which side's list
the fp64 order agrees with.
"""
import torch

from . import aggregation as A
from . import nets as N
from . import sampler_fp64 as SF
from . import vpho as OV
from .compare import selection_report, _margin

OUT_KEYS = ('agg_hand_joint', 'agg_hand_vert', 'agg_obj_6d')
STAGES = ['hand_level0', 'hand_level1', 'hand_level2', 'hand_level3', 'obj_transl', 'obj_rot', 'obj_heat', 'obj_physics', 'hand_physics']


def _to64(d):
    return {k: (v.double() if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in d.items()}


def features64(sd, assets, data):
    """the oracle's feature path (VPHO.py:112-172) in float64 on the whole batch (the cross modules attend over the batch axis, quirk Q3)"""
    with torch.no_grad():
        return OV.features(_to64(sd), assets, _to64(data))


def predict_fp64(sd, assets, anchor_skeleton, data, *, sample_num, sample_T0, sampling_steps, topk_hand, topk_obj, noise_hand, noise_obj,
                 steps_hand, steps_obj, feat64=None, chunk=8):
    """float64 ``predict`` on the accepted step sequences ``steps_hand`` / ``steps_obj`` of an fp32 run ([(t, h, err, accepted)], the
    ``steps`` log of oracle.nets.ode_sample / Engine.last_info).  noise_*: the standard-normal prior draws of that run (fp32).
    -> (out, dbg): out holds OUT_KEYS + agg_hand_mano + the final hypotheses, dbg the aggregation's lists and scores (oracle format)."""
    bs, S = data['rgb'].shape[0], sample_num
    f = features64(sd, assets, data) if feat64 is None else feat64
    d64 = _to64(data)
    sig = N.ve_prior_sigma(sample_T0)
    rep = lambda e: e[:, None].repeat(1, S, 1).reshape(-1, e.shape[-1])
    # the prior draw scaled in fp32, as both fp32 sides start from it (sde.py:26-28); everything after it in double
    x_h = SF.solve_on_steps(sd, 'denoiser_hand', rep(f['encoding_hand']), noise_hand * sig, steps_hand, sampling_steps)
    x_o = SF.solve_on_steps(sd, 'denoiser_obj', rep(f['encoding_obj']), noise_obj * sig, steps_obj, sampling_steps)
    final = OV.postprocess_diffusion_hand(x_h.reshape(bs, S, 96), f['mano_shape'])
    fl = final.reshape(bs, S, 58)
    obj = x_o.reshape(bs, S, 9)
    outs, dbgs = [], []
    with torch.no_grad():
        for b0 in range(0, bs, chunk):                       # the aggregation is per image: chunks bound the (candidates x points) tensors
            sl = slice(b0, min(b0 + chunk, bs))
            a = A.hoi_aggregate(assets, anchor_skeleton, cam_intrinsic=d64['cam_intr_crop_flip'][sl], root_joint_flip=d64['root_joint_flip'][sl],
                                root_joint=d64['root_joint'][sl], is_right=data['is_right'][sl], force_local=f['force_local'][sl],
                                is_grasped=data['is_grasped'][sl], hand_pose_diff=fl[sl].reshape(-1, 58)[:, :48].clone(),
                                hand_pose_regression=f['mano_pose'][sl], hand_shape=fl[sl].reshape(-1, 58)[:, 48:], hand_heatmap=f['hand_heatmap'][sl],
                                hand_bbox=d64['bbox_hand'][sl], hand_topk=topk_hand, obj_pose6d=obj[sl], obj_heatmap=f['obj_heatmap'][sl],
                                obj_bbox=d64['bbox_obj_rect'][sl], obj_topk=topk_obj, obj_name=list(data['obj_name'])[sl], dtype=torch.float64)
            outs.append(dict(agg_hand_joint=a['hand_agg_joint'], agg_hand_vert=a['hand_agg_vert'], agg_hand_mano=a['hand_agg_mano'], agg_obj_6d=a['obj_agg_6d']))
            dbgs.append(a['dbg'])
    out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    out['diff_final_hand_mano'], out['diff_final_obj_6d'], out['hand_x6d'] = final, obj, x_h
    return out, _cat_dbg(dbgs)


def _cat_dbg(dbgs):
    """concatenate the per-chunk ``dbg`` dicts of oracle.aggregation.hoi_aggregate along the image axis (the keys the comparisons read)"""
    cat = lambda xs: torch.cat(xs, 0)
    h = {k: [cat([d['hand'][k][l] for d in dbgs]) for l in range(4)] for k in ('topk', 'score', 'val')}
    hp = {k: cat([d['hand_phys'][k] for d in dbgs]) for k in ('topk', 'score', 'cand')}
    out = dict(hand=h, hand_phys=hp)
    if 'state' in dbgs[0]['hand']:
        h['state'] = [cat([d['hand']['state'][l] for d in dbgs]) for l in range(4)]
    h['fused_pose'] = cat([d['hand']['fused_pose'] for d in dbgs])
    for k in ('transl_topk', 'rot_topk', 'phys_topk', 'heat_topk', 'phys_score', 'transl_score', 'rot_score', 'heat_score'):
        out[k] = cat([d[k] for d in dbgs])
    return out


def as_tested(dbg):
    """the oracle-format ``dbg`` of a run in the layout oracle.compare.selection_report expects of the side under test
    (Engine.last_info['agg']: hand lists (bs, F, k))"""
    h = dbg['hand']
    lst = [(t if t.dim() == 3 else t[:, :, None]).permute(0, 2, 1) for t in h['topk']]
    return dict(hand_topk=lst, hand_phys_topk=dbg['hand_phys']['topk'], transl_topk=dbg['transl_topk'], rot_topk=dbg['rot_topk'],
                phys_topk=dbg['phys_topk'], heat_topk=dbg['heat_topk'])


def _worst(a, b):
    """(bs,) largest |a - b| over joints, vertices and object 6-DoF"""
    c = lambda t: t.detach().cpu().double()
    n = c(a[OUT_KEYS[0]]).shape[0]
    return torch.stack([(c(a[k]) - c(b[k])).abs().reshape(n, -1).amax(1) for k in OUT_KEYS]).amax(0)


def first_flip(rep):
    """per image: index into STAGES of the first selection that differs along the dependency chain (-1: none), from a selection_report"""
    bs = rep['primary_gap_per_image'].shape[0]
    first = torch.full((bs,), -1, dtype=torch.long)
    for b in range(bs):
        for i, st in enumerate(STAGES):
            if int(rep['_n_diff'][st][b]) > 0:
                first[b] = i
                break
    return first


def _report(gd, od, S):
    """selection_report + the per-stage difference counts it computes internally (recomputed here stage by stage)"""
    rep = selection_report(gd, od, S)
    # per-stage counts: selection_report only returns hand / object totals; a stage differs on an image when it is primary or consequent
    # there -- recover them by comparing the lists directly (sets; ranks at level 3), exact-copy aliasing as in selection_report is not
    # needed to LOCATE the first difference because aliases only make lists equal that differ in label
    bs = rep['primary_gap_per_image'].shape[0]
    nd = {st: torch.zeros(bs, dtype=torch.long) for st in STAGES}
    h = od['hand']
    for lvl in range(4):
        want = h['topk'][lvl].long()
        want = want if want.dim() == 3 else want[:, :, None]
        got = torch.as_tensor(gd['hand_topk'][lvl]).cpu().long().transpose(1, 2)
        al = (lambda t: t.clamp(max=S)) if lvl >= 1 else (lambda t: t)               # regression copies are one candidate from level 1 on
        for b in range(bs):
            for f in range(want.shape[2]):
                a_, w_ = al(got[b, :, f]), al(want[b, :, f])
                same = torch.equal(a_, w_) if lvl == 3 else sorted(a_.tolist()) == sorted(w_.tolist())
                nd[f'hand_level{lvl}'][b] += 0 if same else 1
    hd, od_ = rep['hand_differences_per_image'], rep['object_differences_per_image']
    for st, key in (('obj_transl', 'transl_topk'), ('obj_rot', 'rot_topk'), ('obj_heat', 'heat_topk'), ('obj_physics', 'phys_topk')):
        want = od[key].long()
        got = torch.as_tensor(gd[key]).cpu().long().reshape(want.shape)
        for b in range(bs):
            nd[st][b] = 0 if sorted(got[b].tolist()) == sorted(want[b].tolist()) else 1
    want = od['hand_phys']['topk'].long()
    got = torch.as_tensor(gd['hand_phys_topk']).cpu().long().reshape(want.shape)
    for b in range(bs):
        nd['hand_physics'][b] = sum(0 if sorted(got[b, f].tolist()) == sorted(want[b, f].tolist()) else 1 for f in range(5))
    # images selection_report calls clean (aliases, equal scores) are clean here too
    for b in range(bs):
        if int(hd[b]) == 0:
            for st in STAGES[:4] + ['hand_physics']:
                nd[st][b] = 0
        if int(od_[b]) == 0:
            for st in STAGES[4:8]:
                nd[st][b] = 0
    rep['_n_diff'] = nd
    return rep


def _stage_margin(dbg, stage, b):
    """fp64 margin of the list of ``stage`` on image b, relative to the score scale of its vector(s): the smallest score distance a
    perturbation has to bridge to change the list (min over the fingers)"""
    if stage.startswith('hand_level'):
        lvl = int(stage[-1])
        sc, lst = dbg['hand']['score'][lvl][b].double(), dbg['hand']['topk'][lvl][b].long()
        sc, lst = (sc, lst) if sc.dim() == 2 else (sc[:, None], lst[:, None])
        ranked = lvl == 3
    elif stage == 'hand_physics':
        sc, lst, ranked = dbg['hand_phys']['score'][b].double().T, dbg['hand_phys']['topk'][b].long().T, False
    else:
        key = {'obj_transl': 'transl', 'obj_rot': 'rot', 'obj_heat': 'heat', 'obj_physics': 'phys'}[stage]
        sc, lst, ranked = dbg[key + '_score'][b].double()[:, None], dbg[key + '_topk'][b].long()[:, None], False
    m = [(_margin(sc[:, f], lst[:, f], ranked) / max(float(sc[:, f].abs().max()), 1e-300)) for f in range(sc.shape[1])]
    return min(m)


def fused_rotation_errors(states_side, states64, rep, S):
    """How exactly a side FUSES: the rotation a cascade level writes into every candidate (the weighted quaternion mean of its top-k,
    aggregation.py:222-236,250-269 -- torch.linalg.eigh in the reference, an in-kernel Jacobi solve on the HIP path) against the float64
    fusion, on the images whose lists up to and including that level are the float64 lists (same picks, same weights up to rounding).
    states: the candidates each level scored, [(bs, 2S, 48)] x 4 -- level l + 1's candidate 0 carries the rotations fused at levels <= l.
    -> {level: (images compared, max |axis-angle difference|, median over the images)}"""
    out = {}
    bs = states64[0].shape[0]
    ok = torch.ones(bs, dtype=torch.bool)
    for lvl in range(3):
        ok &= rep['_n_diff'][f'hand_level{lvl}'] == 0
        a, b = states_side[lvl + 1][:, 0].detach().cpu().double(), states64[lvl + 1][:, 0].double()
        e = (a - b).abs().amax(1)
        out[f'level{lvl}'] = dict(images=int(ok.sum()), max=float(e[ok].max()) if bool(ok.any()) else None, median=float(e[ok].median()) if bool(ok.any()) else None)
    return out


def topk_value_errors(vals_side, vals64, rep):
    """the scores of the k picks of every cascade level (they become the fusion weights, aggregation.py:219-221,249) against the float64
    scores of the same picks, on the images whose lists up to that level are the float64 lists: relative to the largest score of the
    vector -> {level: (images, rms, max)}.  vals_side: [(bs, F, k)] x 4 (HIP layout) ; vals64: [(bs, k) | (bs, k, F)] (oracle layout)"""
    out = {}
    bs = vals64[0].shape[0]
    ok = torch.ones(bs, dtype=torch.bool)
    for lvl in range(4):
        ok &= rep['_n_diff'][f'hand_level{lvl}'] == 0
        b = vals64[lvl].double()
        b = b[:, :, None] if b.dim() == 2 else b                                  # (bs, k, F)
        a = torch.as_tensor(vals_side[lvl]).detach().cpu().double()
        a = a.permute(0, 2, 1) if a.shape != b.shape else a
        if a.shape != b.shape or not bool(ok.any()):
            out[f'level{lvl}'] = None
            continue
        # lists are equal as sets (levels 0-2) -- sort both by the float64 value so that picks pair up
        e = ((torch.sort(a, dim=1).values - torch.sort(b, dim=1).values).abs() / b.abs().amax(1, keepdim=True).clamp(min=1e-300))[ok]
        out[f'level{lvl}'] = dict(images=int(ok.sum()), rms=float(e.pow(2).mean().sqrt()), max=float(e.max()))
    return out


def judge(out_hip, agg_hip, out_or, dbg_or, out64_hip, dbg64_hip, out64_or, dbg64_or, S, bar=1e-3):
    """The table of one batch.  out_* / *_dbg: outputs and selections of the HIP path (agg_hip = Engine.last_info['agg']), of the fp32
    oracle, and of the float64 predicts on either side's step sequences."""
    w_hip, w_or = _worst(out_hip, out64_hip), _worst(out_or, out64_or)
    w_sides, w_truths = _worst(out_hip, out_or), _worst(out64_hip, out64_or)
    r_hip = _report(agg_hip, dbg64_hip, S)                 # HIP's lists against the fp64 lists of its own scheme
    r_or = _report(as_tested(dbg_or), dbg64_or, S)         # the fp32 oracle's against the fp64 lists of ITS scheme
    r_sides = _report(agg_hip, dbg_or, S)                  # the two fp32 sides against each other
    f_hip, f_or, f_sides = first_flip(r_hip), first_flip(r_or), first_flip(r_sides)
    bs = w_hip.shape[0]
    rows = []
    for b in range(bs):
        if float(w_sides[b]) <= bar:
            continue
        st = STAGES[int(f_sides[b])] if int(f_sides[b]) >= 0 else None
        # at the first stage the two fp32 sides differ: does each side's list there equal the fp64 list of its own scheme?
        hip_ok = st is not None and int(r_hip['_n_diff'][st][b]) == 0 and all(int(r_hip['_n_diff'][s_][b]) == 0 for s_ in STAGES[:STAGES.index(st)] if _dep(s_, st))
        or_ok = st is not None and int(r_or['_n_diff'][st][b]) == 0 and all(int(r_or['_n_diff'][s_][b]) == 0 for s_ in STAGES[:STAGES.index(st)] if _dep(s_, st))
        rows.append(dict(image=b, first_stage_the_fp32_sides_differ=st, rel_score_gap_there=float(r_sides['primary_gap_per_image'][b]),
                         fp64_margin_rel=None if st is None else _stage_margin(dbg64_or, st, b),
                         fp64_order_agrees_with=('both' if hip_ok and or_ok else 'hip' if hip_ok else 'oracle' if or_ok else 'neither'),
                         max_abs_hip_vs_oracle=float(w_sides[b]), max_abs_hip_vs_fp64=float(w_hip[b]), max_abs_oracle_vs_fp64=float(w_or[b])))
    per_stage = {}
    for r in rows:
        d = per_stage.setdefault(str(r['first_stage_the_fp32_sides_differ']), dict(images=0, fp64_agrees_with_hip=0, fp64_agrees_with_oracle=0, both=0, neither=0))
        d['images'] += 1
        d[{'hip': 'fp64_agrees_with_hip', 'oracle': 'fp64_agrees_with_oracle', 'both': 'both', 'neither': 'neither'}[r['fp64_order_agrees_with']]] += 1
    clean = lambda r: int(((r['hand_differences_per_image'] == 0) & (r['object_differences_per_image'] == 0)).sum())
    hist = lambda f: {STAGES[i]: int((f == i).sum()) for i in range(len(STAGES)) if int((f == i).sum())}
    fused = None
    if 'cascade_state' in agg_hip and 'state' in dbg_or['hand'] and 'state' in dbg64_hip['hand']:
        fused = dict(hip=fused_rotation_errors(agg_hip['cascade_state'], dbg64_hip['hand']['state'], r_hip, S),
                     oracle=fused_rotation_errors(dbg_or['hand']['state'], dbg64_or['hand']['state'], r_or, S))
    cands = None
    if fused is not None:
        def cand_err(st, st64):
            a, b = torch.as_tensor(st[0]).detach().cpu().double(), st64[0].double()          # (bs, 2S, 48): [S diffusion | S regression copies]
            e = (a - b).abs()
            return {'diffusion_hypotheses': {'rms': float(e[:, :S].pow(2).mean().sqrt()), 'max': float(e[:, :S].max())},
                    'regression_copies_joints_1_15': {'rms': float(e[:, S:, 3:].pow(2).mean().sqrt()), 'max': float(e[:, S:, 3:].max())}}
        cands = dict(hip=cand_err(agg_hip['cascade_state'], dbg64_hip['hand']['state']), oracle=cand_err(dbg_or['hand']['state'], dbg64_or['hand']['state']))
    vals = None
    if 'hand_val' in agg_hip and 'val' in dbg64_hip['hand']:
        vals = dict(hip=topk_value_errors(agg_hip['hand_val'], dbg64_hip['hand']['val'], r_hip),
                    oracle=topk_value_errors([v if v.dim() == 3 else v[:, :, None] for v in dbg_or['hand']['val']], dbg64_or['hand']['val'], r_or))
    return dict(images=bs, bar=bar, fused_rotation_abs_err_vs_fp64=fused, topk_value_rel_err_vs_fp64=vals, candidate_pose_abs_err_vs_fp64=cands,
                images_within_1e3_of_fp64={'hip': int((w_hip <= bar).sum()), 'oracle': int((w_or <= bar).sum())},
                images_lists_identical_to_fp64={'hip': clean(r_hip), 'oracle': clean(r_or)},
                first_differing_stage_vs_fp64={'hip': hist(f_hip), 'oracle': hist(f_or)},
                hip_vs_oracle={'images_within_1e3': int((w_sides <= bar).sum()), 'images_lists_identical': clean(r_sides), 'first_differing_stage': hist(f_sides)},
                the_two_fp64_truths_within_1e3_of_each_other=int((w_truths <= bar).sum()),
                max_abs_vs_fp64_where_lists_identical={'hip': _max_where(w_hip, r_hip), 'oracle': _max_where(w_or, r_or)},
                images_outside_1e3_between_the_fp32_sides=rows, by_first_stage=per_stage)


def _max_where(w, rep):
    m = (rep['hand_differences_per_image'] == 0) & (rep['object_differences_per_image'] == 0)
    return float(w[m].max()) if bool(m.any()) else None


def _dep(earlier, stage):
    """does ``stage`` consume the result of ``earlier`` (the dependency chains of oracle.compare.selection_report)"""
    hand, obj = STAGES[:4], STAGES[4:7]
    deps = {s: hand[:i] for i, s in enumerate(hand)}
    deps.update({s: obj[:i] for i, s in enumerate(obj)})
    deps['obj_physics'] = hand + ['obj_transl', 'obj_rot']
    deps['hand_physics'] = hand + obj + ['obj_physics']
    return earlier in deps[stage]
