"""Contact detection kernels against the reference fixture (sklearn ball tree) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_blocks.npz'))


def _inputs(assets):
    rng = np.random.default_rng(41)
    hv = (assets['mano']['v_template'] + rng.normal(size=(778, 3)) * 0.001).astype(np.float64)
    hn = rng.normal(size=(778, 3)); hn /= np.linalg.norm(hn, axis=-1, keepdims=True)
    ov = (assets['ycb']['003_cracker_box']['verts'] * 0.6 + np.array([0.06, 0.0, 0.0])).astype(np.float64)
    on = rng.normal(size=ov.shape); on /= np.linalg.norm(on, axis=-1, keepdims=True)
    return hv, hn, ov, on


def test_matches_reference_fixture(assets):
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    t = lambda a: torch.from_numpy(a.astype(np.float32))[None].cuda().contiguous()
    hv, hn, ov, on = _inputs(assets)
    hc, oc, o2h = agg.contact_detect(t(hv), t(hn), t(ov), t(on), normal_thresh=(-0.01, 0.01), vertical_thresh=0.005)
    hc, oc, o2h = hc[0].cpu().numpy(), oc[0].cpu().numpy(), o2h[0].cpu().numpy()
    # fp32 coordinates vs the reference's fp64: the weight's exp(1600 x) amplifies a 6e-8 m rounding to ~1e-4 relative, and a
    # point within that distance of a gate may flip: compare where the reference is not within 1e-6 m of a threshold
    ref_h, ref_o = G['contact_hand'], G['contact_obj']
    assert np.abs(hc - ref_h).max() < 2e-3 and np.abs(oc - ref_o).max() < 2e-3
    assert ((hc > 0) == (ref_h > 0)).mean() > 0.995 and ((oc > 0) == (ref_o > 0)).mean() > 0.995
    same = (o2h >= 0) == (G['contact_o2h'] >= 0)
    assert same.mean() > 0.995
    both = (o2h >= 0) & (G['contact_o2h'] >= 0)
    assert np.array_equal(o2h[both], G['contact_o2h'][both])
    fc, gr = agg.force_contact(torch.from_numpy(ref_h.astype(np.float32))[None].cuda().contiguous())
    np.testing.assert_allclose(fc[0].cpu().numpy(), G['contact_force'], rtol=1e-5, atol=1e-7)
    assert bool(gr[0].item()) == bool(G['contact_is_grasped'])


def test_batched_ragged_sizes_match_oracle(assets):
    """3 samples, 1080 'gap-filled' hand points (not a multiple of the 256-thread block) against 2500 object points (not a
    multiple of the 1024-point LDS tile)."""
    from oracle import contact as OC
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    rng = np.random.default_rng(7)
    n, nh, no = 3, 1080, 2500
    hv = rng.uniform(-0.05, 0.05, size=(n, nh, 3)).astype(np.float32)
    ov = rng.uniform(-0.05, 0.05, size=(n, no, 3)).astype(np.float32)
    unit = lambda a: (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
    hn, on = unit(rng.normal(size=hv.shape)), unit(rng.normal(size=ov.shape))
    t = lambda a: torch.from_numpy(a).cuda().contiguous()
    hc, oc, o2h = agg.contact_detect(t(hv), t(hn), t(ov), t(on))
    for i in range(n):
        rh, ro, r2h = OC.detect(hv[i].astype(np.float64), hn[i].astype(np.float64), ov[i].astype(np.float64), on[i].astype(np.float64))
        assert np.abs(hc[i].cpu().numpy() - rh).max() < 2e-3
        assert np.abs(oc[i].cpu().numpy() - ro).max() < 2e-3
        g = o2h[i].cpu().numpy()
        both = (g >= 0) & (r2h >= 0)
        assert both.sum() > 10 and np.array_equal(g[both], r2h[both])
        assert ((g >= 0) == (r2h >= 0)).mean() > 0.995
