"""Device-side hand metrics against the reference's TesterHand fixture and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_blocks.npz'))


def _data():
    rng = np.random.default_rng(31)
    gtj, gtv = rng.normal(size=(6, 21, 3)).astype(np.float32) * 0.05, rng.normal(size=(6, 778, 3)).astype(np.float32) * 0.05
    pdj = (gtj + rng.normal(size=gtj.shape) * 0.01).astype(np.float32)
    pdv = (gtv + rng.normal(size=gtv.shape) * 0.01).astype(np.float32)
    return gtj, gtv, pdj, pdv


def test_matches_tester_hand_fixture():
    from vpho_amd import ops
    gtj, gtv, pdj, pdv = _data()
    t = lambda a: torch.from_numpy(a).cuda()
    mje, pa, je = ops.hand_metrics(t(pdj), t(gtj), per_point=True)
    mve, pav = ops.hand_metrics(t(pdv), t(gtv))
    np.testing.assert_allclose(mje.cpu().numpy(), G['tester_MJE'], rtol=1e-5)
    np.testing.assert_allclose(pa.cpu().numpy(), G['tester_PA_MJE'], rtol=2e-5)
    np.testing.assert_allclose(je.cpu().numpy(), G['tester_JE'], rtol=1e-5)
    np.testing.assert_allclose(mve.cpu().numpy(), G['tester_MVE'], rtol=1e-5)
    np.testing.assert_allclose(pav.cpu().numpy(), G['tester_PAMVE'], rtol=2e-5)


def test_procrustes_invariances_and_reflection_branch():
    """PA error is invariant to any similarity transform of the prediction and ~0 when pd is a similarity copy of gt;
    a mirrored prediction exercises the det(R) < 0 branch (transform_fn.py:51-54) and must agree with the oracle."""
    from oracle import metrics as OM
    from vpho_amd import ops
    rng = np.random.default_rng(5)
    gt = (rng.normal(size=(4, 21, 3)) * 0.05).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.linalg.det(q))
    pd = (1.7 * gt @ q.T + np.array([0.1, -0.2, 0.3])).astype(np.float32)
    _, pa = ops.hand_metrics(torch.from_numpy(pd).cuda(), torch.from_numpy(gt).cuda())
    assert pa.abs().max().item() < 1e-6
    noisy = (gt + rng.normal(size=gt.shape) * 0.01).astype(np.float32)
    moved = (0.6 * noisy @ q.T - 0.05).astype(np.float32)
    pa1 = ops.hand_metrics(torch.from_numpy(noisy).cuda(), torch.from_numpy(gt).cuda())[1]
    pa2 = ops.hand_metrics(torch.from_numpy(moved).cuda(), torch.from_numpy(gt).cuda())[1]
    assert (pa1 - pa2).abs().max().item() < 1e-6
    mirrored = noisy.copy()
    mirrored[..., 0] *= -1
    pa3 = ops.hand_metrics(torch.from_numpy(mirrored).cuda(), torch.from_numpy(gt).cuda())[1].cpu().numpy()
    ref = np.array([OM.mje_pamje(gt[i].astype(np.float64), mirrored[i].astype(np.float64))[1] for i in range(4)])
    np.testing.assert_allclose(pa3, ref, rtol=1e-4)


# ------------------------------------------------------------------------------------------------ object metrics
def _obj_cols():
    from oracle import metrics as OM
    return {k: i for i, k in enumerate(OM.OBJ_METRIC_NAMES)}, OM


def _check_obj(got, ref, nn_atol, f_atol):
    col, OM = _obj_cols()
    for k in ('MCE', 'OCE', 'REP'):
        np.testing.assert_allclose(got[:, col[k]], ref[:, col[k]], rtol=1e-6, err_msg=k)
    for k in ('MCE2', 'ADD'):
        np.testing.assert_allclose(got[:, col[k]], ref[:, col[k]], rtol=5e-6, err_msg=k)
    for k in ('ADDS', 'CD'):
        np.testing.assert_allclose(got[:, col[k]], ref[:, col[k]], atol=nn_atol, rtol=2e-6, err_msg=k)
    for k in ('ADD01d', 'ADDS01d', 'REP5'):
        np.testing.assert_array_equal(got[:, col[k]], ref[:, col[k]], err_msg=k)
    for k in OM.OBJ_METRIC_NAMES[10:]:
        np.testing.assert_allclose(got[:, col[k]], ref[:, col[k]], atol=f_atol, err_msg=k)


def test_object_metrics_match_tester_object_fixture(assets):
    """vpho_obj_metrics_f64 vs the reference's own TesterObject (lib/engine/test.py:240-503) on the committed fixture, and
    vs the oracle, which computes the same exact nearest-neighbour distances (fp32, direct differences)."""
    from vpho_amd import ops
    col, OM = _obj_cols()
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_objmetrics.npz'))
    names = list(assets['ycb'].keys())
    M = ops.ObjectMetrics(assets['ycb'], 'cuda')
    d = lambda a: torch.from_numpy(np.asarray(a, np.float64)).cuda()
    got = M(d(g['pd_rt']), d(g['gt_rt']), d(g['cam_intr']), torch.from_numpy(g['obj_idx'].astype(np.int32)).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    _check_obj(got, g['metrics'], nn_atol=2e-5, f_atol=3e-3)          # the reference's cdist expansion is only that accurate
    orc = np.stack([OM.object_metrics(assets['ycb'][names[int(o)]], g['pd_rt'][i], g['gt_rt'][i], g['cam_intr'][i])
                    for i, o in enumerate(g['obj_idx'])])
    _check_obj(got, orc, nn_atol=1e-9, f_atol=1e-6)                    # same arithmetic: distances and counts agree exactly


def test_object_metrics_properties(assets):
    """Identity prediction -> every distance 0 and every hit 1; a pure translation d -> MCE = OCE = ADD = |d|, MCE2 = |d|;
    ragged vertex counts (objects with fewer full vertices than max_verts) and a single image."""
    from vpho_amd import ops
    col, OM = _obj_cols()
    ycb = {k: dict(v) for k, v in assets['ycb'].items()}
    names = list(ycb.keys())
    ycb[names[1]]['verts'] = ycb[names[1]]['verts'][:777]             # ragged
    ycb[names[2]]['verts'] = ycb[names[2]]['verts'][:300]
    M = ops.ObjectMetrics(ycb, 'cuda')
    rng = np.random.default_rng(3)
    from oracle import rotations as R
    n = 5
    Rm = R.axis_angle_to_matrix(torch.from_numpy(rng.normal(size=(n, 3)))).numpy()
    t = rng.normal(size=(n, 3)) * 0.05 + np.array([0, 0, 0.7])
    gt = np.concatenate([Rm, t[:, :, None]], -1)
    cam = np.tile(np.array([[500.0, 0, 128], [0, 500.0, 128], [0, 0, 1]]), (n, 1, 1))
    oid = torch.tensor([0, 1, 2, 1, 2], dtype=torch.int32).cuda()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    same = M(d(gt), d(gt), d(cam), oid).cpu().numpy()
    for k in ('MCE', 'OCE', 'MCE2', 'ADD', 'ADDS', 'REP', 'CD'):
        assert np.abs(same[:, col[k]]).max() < 1e-12, k
    for k in ('ADD01d', 'ADDS01d', 'REP5') + OM.OBJ_METRIC_NAMES[10:]:
        np.testing.assert_allclose(same[:, col[k]], 1.0, atol=1e-6, err_msg=k)
    shift = gt.copy()
    dvec = np.array([0.003, -0.004, 0.012])
    shift[:, :, 3] += dvec
    mv = M(d(shift), d(gt), d(cam), oid).cpu().numpy()
    for k in ('MCE', 'OCE', 'MCE2', 'ADD'):
        np.testing.assert_allclose(mv[:, col[k]], np.linalg.norm(dvec), rtol=2e-5, err_msg=k)
    orc = np.stack([OM.object_metrics(ycb[names[int(o)]], shift[i], gt[i], cam[i]) for i, o in enumerate(oid.cpu().numpy())])
    _check_obj(mv, orc, nn_atol=1e-9, f_atol=1e-6)
    one = M(d(shift[:1]), d(gt[:1]), d(cam[:1]), oid[:1]).cpu().numpy()
    np.testing.assert_array_equal(one, mv[:1])


def test_object_metric_block_of_the_evaluation_rows(assets):
    """evaluate.object_metric_block = obj_9D_to_mat + root joint (transform_fn.py:85-90, train_diff_hand_obj.py:594-597) +
    TesterObject, all on the device, vs the oracle on a synthetic batch."""
    from vpho_amd import evaluate as E
    from vpho_amd.synth import synth_batch
    col, OM = _obj_cols()
    bs = 7
    data = synth_batch(bs, assets, seed=11)
    rng = np.random.default_rng(12)
    gt9 = np.concatenate([data['gt_obj_rt'][:, :2, :3].reshape(bs, 6).numpy(), (data['gt_obj_rt'][:, :, 3] - data['root_joint']).numpy()], -1)
    pose9 = gt9.astype(np.float64) + rng.normal(size=(bs, 9)) * np.linspace(1e-4, 0.2, bs)[:, None]
    out = {'agg_obj_6d': torch.from_numpy(pose9).cuda()}
    gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    got = E.object_metric_block(out, gdata, assets).cpu().numpy()
    rt = OM.obj_9d_to_rt(pose9, data['root_joint'].numpy())
    orc = np.stack([OM.object_metrics(assets['ycb'][n], rt[i], data['gt_obj_rt'][i].numpy(), data['cam_intr'][i].numpy())
                    for i, n in enumerate(data['obj_name'])])
    _check_obj(got, orc, nn_atol=1e-7, f_atol=1e-3)
    assert got[0, col['ADD']] < 1e-3 < got[-1, col['ADD']]          # the sweep really goes from near-exact to far off


def test_one_object_metrics_instance_serves_concurrent_streams(assets):
    """The evaluator's slots share one ObjectMetrics object and call it on their own streams: every call must own its workspace
    (a workspace kept on the object was shared by kernels of different streams)."""
    from vpho_amd import ops
    from oracle import rotations as R
    M = ops.ObjectMetrics(assets['ycb'], 'cuda')
    rng = np.random.default_rng(5)
    n, sets = 64, []
    for s in range(6):
        Rm = R.axis_angle_to_matrix(torch.from_numpy(rng.normal(size=(n, 3)))).numpy()
        t = rng.normal(size=(n, 3)) * 0.05 + np.array([0, 0, 0.7])
        gt = np.concatenate([Rm, t[:, :, None]], -1)
        pd = gt.copy(); pd[:, :, 3] += rng.normal(size=(n, 3)) * 0.01 * (s + 1)
        cam = np.tile(np.array([[500.0, 0, 128], [0, 500.0, 128], [0, 0, 1]]), (n, 1, 1))
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        sets.append((d(pd), d(gt), d(cam), torch.from_numpy(rng.integers(0, len(M.names), n).astype(np.int32)).cuda()))
    want = [M(*a).clone() for a in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in sets]
    for rep in range(5):
        got = []
        for st, a in zip(streams, sets):
            with torch.cuda.stream(st):
                got.append(M(*a))
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g, w)
