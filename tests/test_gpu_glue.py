"""Glue kernels against a plain torch fp32 restatement of the same op: attention (both kernels: the wave-per-4-queries one
and the generic fallback) and the batched MANO forward kinematics at every hands-per-block variant."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mha_ref(qkv, S, B, E, nhead):
    hd = E // nhead
    x = qkv.double().view(S, B, 3, nhead, hd)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]                     # (S, B, nhead, hd)
    att = torch.einsum('sbhd,tbhd->bhst', q, k) / math.sqrt(hd)
    return torch.einsum('bhst,tbhd->sbhd', att.softmax(-1), v).reshape(S, B, E).float()


# (S, B, E, nhead): the cross module's shape (sequence = batch axis, quirk Q3) at several batch sizes incl. ragged query
# chunks and >64 keys; a small-head fast-path shape; two shapes only the generic kernel takes (hd % 4 != 0, hd > 256)
@pytest.mark.parametrize('S,B,E,nhead', [(64, 65, 512, 2), (1, 65, 512, 2), (5, 65, 512, 2), (70, 65, 512, 2), (200, 3, 64, 4),
                                        (256, 2, 32, 2), (7, 3, 12, 2), (9, 2, 640, 2)])
def test_mha_matches_torch(S, B, E, nhead):
    from vpho_amd import ops
    g = torch.Generator().manual_seed(S * 1000 + E)
    qkv = torch.randn(S * B, 3 * E, generator=g)
    out = ops.mha(qkv.cuda(), S, B, E, nhead).cpu()
    ref = _mha_ref(qkv, S, B, E, nhead)
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('n_img,per_img', [(2, 1), (3, 50), (5, 300), (1, 1030)])     # -> 1, 4 and 16 hands per block, ragged tails
def test_mano_fk_batched_matches_oracle(assets, n_img, per_img):
    from oracle import mano as omano
    from vpho_amd import ops
    g = torch.Generator().manual_seed(n_img * 7 + per_img)
    n = n_img * per_img
    pose = torch.randn(n, 48, generator=g) * 0.4
    betas = torch.randn(n_img, 10, generator=g) * 0.5
    M = ops.Mano(assets['mano'], 'cuda')
    ctx = M.shape(betas.cuda())
    verts, joints = M.fk(pose.cuda(), ctx, per_img, True)
    _, joints_only = M.fk(pose.cuda(), ctx, per_img, False)
    pick = torch.linspace(0, n - 1, min(n, 40)).long()
    rv, rj = omano.get_hand_verts(assets['mano'], pose[pick], betas[pick // per_img])
    np.testing.assert_allclose(verts.cpu()[pick].numpy(), np.asarray(rv), atol=2e-6)
    np.testing.assert_allclose(joints.cpu()[pick].numpy(), np.asarray(rj), atol=2e-6)
    assert torch.equal(joints, joints_only)


@pytest.mark.parametrize('n,F,k', [(5, 1, 5), (200, 5, 30), (512, 1, 30), (513, 5, 64), (1024, 1, 30)])
def test_wavefront_topk_matches_a_stable_descending_sort(assets, n, F, k):
    """vpho_topk_f32 (wavefront butterfly arg-max; 8 value slots per lane up to 512 candidates, 16 up to 1024): values descending, ties
    by ascending index (torch.topk leaves that order open; the oracle defines it the same way), NaN first, -inf never picked twice."""
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    g = torch.Generator().manual_seed(n * 7 + F)
    x = torch.randn(3, n, F, generator=g)
    x[0, : n // 2] = x[0, 0]                                # a long run of exact ties
    if n > 40:
        x[1, 7, 0] = float('nan')
        x[2, 3:20] = float('-inf')
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    val, idx = agg.topk(x.cuda().contiguous() if F > 1 else x[..., 0].cuda().contiguous(), k, F)
    val, idx = val.cpu(), idx.cpu().long()
    xs = torch.where(torch.isnan(x), torch.full_like(x, float('inf')), x)
    sv, si = torch.sort(xs, dim=1, descending=True, stable=True)
    assert torch.equal(idx, si[:, :k].permute(0, 2, 1))
    assert torch.equal(val, sv[:, :k].permute(0, 2, 1))
    assert idx.min() >= 0 and idx.max() < n
