"""Pseudo-force label optimisation: the persistent AdamW kernel against the oracle's autograd loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(assets, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    v = torch.as_tensor(assets['mano']['v_template'])[None] + torch.randn(B, 778, 3, generator=g) * 0.002 + torch.tensor([0.0, 0.0, 0.7])
    grav = torch.nn.functional.normalize(torch.randn(B, 1, 3, generator=g), dim=-1)
    com = torch.tensor([0.05, 0.0, 0.7]) + torch.randn(B, 1, 3, generator=g) * 0.02
    fc = torch.rand(B, 32, generator=g)
    grasped = torch.rand(B, generator=g) < 0.8
    return v.contiguous(), grav.contiguous(), com.contiguous(), fc.contiguous(), grasped


def test_anchor_frames_match_oracle(assets):
    from oracle.aggregation import vert2anchor
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    v = _inputs(assets, 3)[0]
    pts, frame = vert2anchor(assets['anchor'], ANCHOR_SKELETON, v)
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    p, f = agg.anchor_frames(v.cuda())
    assert (p.cpu() - pts).abs().max().item() < 1e-6
    assert (f.cpu() - frame).abs().max().item() < 1e-5


@pytest.mark.parametrize('iters,phase1,tol', [(40, 25, 2e-5), (400, 300, 5e-4)])
def test_force_optimize_matches_autograd_loop(assets, iters, phase1, tol):
    """Same parameters after `iters` AdamW steps (both phases exercised).  Tolerance grows with the iteration count: both sides
    are fp32 with different reduction orders, and Adam's 1/sqrt(v) amplifies rounding of tiny gradients."""
    from oracle import force_optim as FO
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    B = 6
    v, grav, com, fc, grasped = _inputs(assets, B, seed=1)
    ref = FO.optimize(assets['anchor'], ANCHOR_SKELETON, v, grav, com, fc, grasped, iters=iters, phase1=phase1)
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    out = agg.force_optimize(v.cuda(), grav.view(B, 3).cuda(), com.view(B, 3).cuda(), fc.cuda(), grasped.to(torch.uint8).cuda(), B,
                             iters=iters, phase1=phase1)
    torch.cuda.synchronize()
    assert (out['scale'].cpu() - ref['scale']).abs().max().item() < tol
    assert (out['weight'].cpu() - ref['weight']).abs().max().item() < tol
    assert (out['force_local'].cpu() - ref['force_local']).abs().max().item() < tol
    assert (out['force_global'].cpu() - ref['force_global']).abs().max().item() < tol
    np.testing.assert_allclose(out['losses'].cpu().numpy()[0], np.array(ref['losses']), rtol=max(50 * tol, 1e-3), atol=1e-6)
    assert (out['force_local'].cpu()[~grasped] == 0).all() and (out['force_global'].cpu()[~grasped] == 0).all()


def test_batches_are_independent_and_full_loop_reduces_losses(assets):
    """Two batches in one launch equal two separate launches (the batch-mean coupling stays inside a batch); the full
    3000-iteration run drives force-balance and gravity-alignment losses down."""
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    B = 8
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    ins = [_inputs(assets, B, seed=s) for s in (2, 3)]
    cat = [torch.cat([a[i] for a in ins]) for i in range(5)]
    call = lambda v, g, c, f, m, **kw: agg.force_optimize(v.cuda(), g.view(-1, 3).cuda(), c.view(-1, 3).cuda(), f.cuda(), m.to(torch.uint8).cuda(), B, **kw)
    both = call(*cat, iters=60, phase1=30)
    one = call(*ins[1], iters=60, phase1=30)
    assert torch.equal(both['scale'][B:], one['scale']) and torch.equal(both['weight'][B:], one['weight'])
    short = call(*ins[0], iters=2, phase1=1)
    full = call(*ins[0])                                   # 3000 iterations, 300 in phase 1
    torch.cuda.synchronize()
    ls, lf = short['losses'].cpu().numpy()[0], full['losses'].cpu().numpy()[0]
    assert np.isfinite(lf).all()
    assert lf[0] < 0.5 * ls[0] and lf[1] < ls[1]
