"""Physics branch of the training step (lib/model/VPHO.py:170-172,205-212) against the reference's own
CrossModule x 2 + HeadPhysics + get_loss under autograd (tests/golden/make_golden_physics_train.py; dropout sites at 0):
forward tokens / forces, the five weighted losses, the gradient of all 48 parameter tensors and of the two stage maps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_physics_train.npz'))
BS, STRIDE = 6, 997
W = dict(force_loss=1.0, gravity_loss=1.0, torque_loss=30.0, supervised_loss=10.0, CoM_loss=100.0)


def inputs(assets):
    """the generator's inputs (tests/golden/make_golden_physics_train.py::inputs)"""
    g = np.random.default_rng(123)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    grav = g.normal(size=(BS, 1, 3))
    grav /= np.linalg.norm(grav, axis=-1, keepdims=True)
    vert = np.asarray(assets['mano']['v_template'])[None] + g.normal(size=(BS, 778, 3)) * 0.002 + np.array([0.02, -0.01, 0.7])
    return dict(st_h=f32(g.normal(size=(BS, 256, 8, 8)) * 0.2), st_o=f32(g.normal(size=(BS, 256, 8, 8)) * 0.2), gravity=f32(grav),
                gt_vert=f32(vert), gt_CoM=f32(np.array([0.05, 0.0, 0.7]) + g.normal(size=(BS, 1, 3)) * 0.02),
                gt_force_local=f32(g.normal(size=(BS, 32, 3)) * 0.1), is_grasped=torch.from_numpy(g.random(BS) < 0.7))


@pytest.fixture(scope='module')
def run(sd, assets):
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.train_blocks import PhysicsTrain
    d = inputs(assets)
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    pt = PhysicsTrain(sd, agg, torch.device('cuda'))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    out = pt.forward_backward(nhwc(d['st_h']), nhwc(d['st_o']), d['gravity'].cuda(), d['gt_CoM'].cuda(), torch.ones(BS, dtype=torch.bool).cuda(),
                              d['gt_vert'].cuda(), d['gt_force_local'].cuda(), d['is_grasped'].cuda(), tuple(W.values()))
    torch.cuda.synchronize()
    return out


def test_forward_tokens_forces_and_losses(run):
    L, d_sth, d_sto, grads, fl, aux = run
    c = lambda t: t.detach().cpu().numpy()
    assert np.abs(c(aux['tok_hand'])[:, ::8, ::16] - G['tok_hand']).max() < 2e-5
    assert np.abs(c(aux['tok_obj'])[:, ::8, ::16] - G['tok_obj']).max() < 2e-5
    assert np.abs(c(aux['scale']) - G['scale']).max() < 1e-5          # the raw fc_scale output (abs is taken inside get_local_force)
    assert np.abs(c(aux['com']) - G['CoM']).max() < 1e-5
    assert np.abs(c(fl) - G['force_local']).max() < 1e-5
    for k in W:
        np.testing.assert_allclose(float(L[k]), float(G[k]), rtol=2e-5, err_msg=k)


def test_parameter_gradients(run):
    L, d_sth, d_sto, grads, fl, aux = run
    names = [k[len('gnorm_'):] for k in G.files if k.startswith('gnorm_') and not k.startswith('gnorm_st_')]
    assert len(names) == 48 and set(names) == set(grads), sorted(set(names) ^ set(grads))[:6]
    worst = 0.0
    for n in names:
        g = grads[n].detach().cpu().reshape(-1)
        nrm = float(G['gnorm_' + n])
        np.testing.assert_allclose(float(g.double().norm()), nrm, rtol=1e-4, atol=1e-9, err_msg=n)
        ref = G['g_' + n]
        got = (g if g.numel() <= 4096 else g[::STRIDE]).numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        worst = max(worst, err)
        assert err < 2e-4, (n, err)
    print('worst relative gradient error', worst)


def test_stage_map_gradients(run):
    """d loss / d stage maps: what the two encoders' backward receives (the detached streams contribute nothing, VPHO.py:170-171)"""
    L, d_sth, d_sto, grads, fl, aux = run
    for k, t in (('st_h', d_sth), ('st_o', d_sto)):
        g = t.permute(0, 3, 1, 2).contiguous().cpu().reshape(-1)            # NHWC -> the reference's NCHW order
        np.testing.assert_allclose(float(g.double().norm()), float(G['gnorm_' + k]), rtol=1e-4)
        ref = G['g_' + k]
        assert np.abs(g[::101].numpy() - ref).max() < 2e-4 * np.abs(ref).max(), k


@pytest.mark.parametrize('S,p', [(6, 0.0), (32, 0.1), (64, 0.3), (65, 0.0), (100, 0.2), (192, 0.1)])
def test_attention_forward_backward_with_dropout_mask_matches_autograd(S, p):
    """vpho_mha_dropout_f32 / vpho_mha_bwd_f32 (sequence axis = batch, 65 token slots x 2 heads) vs fp64 autograd of the same
    attention written out with an explicit keep-mask / (1 - p) on the probabilities (nn.MultiheadAttention's dropout site).  Above 64
    positions (a per-rank batch above 64 images) the backward is the three-launch form over a workspace (vpho_mha_bwd_ws_f32)."""
    from vpho_amd import ops
    B, E, H = 65, 512, 2
    hd = E // H
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(S * B, 3 * E, generator=g) * 0.5
    d_out = torch.randn(S * B, E, generator=g)
    mask = None if p == 0.0 else ((torch.rand(B * H, S, S, generator=g) >= p).float() / (1.0 - p))
    x = qkv.double().requires_grad_(True)
    t = x.view(S, B, 3, H, hd)
    q, k, v = t[:, :, 0].permute(1, 2, 0, 3), t[:, :, 1].permute(1, 2, 0, 3), t[:, :, 2].permute(1, 2, 0, 3)     # (B,H,S,hd)
    P = torch.softmax((q / hd ** 0.5) @ k.transpose(-1, -2), dim=-1)
    if mask is not None:
        P = P * mask.double().view(B, H, S, S)
    out = (P @ v).permute(2, 0, 1, 3).reshape(S * B, E)
    (out * d_out.double()).sum().backward()
    dev = lambda a: None if a is None else a.cuda().contiguous()
    got = ops.mha(dev(qkv), S, B, E, H, drop=dev(mask)).view(S * B, E)
    dq = ops.mha_bwd(dev(qkv), dev(d_out), S, B, E, H, drop=dev(mask))
    torch.cuda.synchronize()
    assert float((got.cpu().double() - out.detach()).abs().max()) < 2e-5 * float(out.detach().abs().max())
    assert float((dq.cpu().double() - x.grad).abs().max()) < 2e-5 * float(x.grad.abs().max())


def test_physics_branch_with_dropout_runs_and_differs(sd, assets):
    """cfg.cross_dropout > 0: all five dropout sites active (fresh Bernoulli masks per call); finite losses / gradients, and two calls
    differ from each other and from the p = 0 result"""
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.train_blocks import PhysicsTrain
    d = inputs(assets)
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    args = (nhwc(d['st_h']), nhwc(d['st_o']), d['gravity'].cuda(), d['gt_CoM'].cuda(), torch.ones(BS, dtype=torch.bool).cuda(),
            d['gt_vert'].cuda(), d['gt_force_local'].cuda(), d['is_grasped'].cuda(), tuple(W.values()))
    torch.manual_seed(3)
    pt = PhysicsTrain(sd, agg, torch.device('cuda'), p_drop=0.1)
    a = pt.forward_backward(*args)
    b = pt.forward_backward(*args)
    for L, dh, do, grads, fl, aux in (a, b):
        assert all(torch.isfinite(v).all() for v in L.values()) and torch.isfinite(dh).all() and torch.isfinite(do).all()
        assert all(torch.isfinite(v).all() for v in grads.values()) and len(grads) == 48
    assert float(a[0]['CoM_loss']) != float(b[0]['CoM_loss'])
    assert abs(float(a[0]['CoM_loss']) - float(G['CoM_loss'])) > 1e-6
