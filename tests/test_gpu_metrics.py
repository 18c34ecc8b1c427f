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
