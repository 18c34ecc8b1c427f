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


@pytest.mark.parametrize('n_img,per_img', [(2, 1), (3, 50), (5, 300), (1, 1030), (41, 100), (130, 33), (140, 30), (260, 16), (280, 15)])
def test_mano_fk_batched_matches_oracle(assets, n_img, per_img):
    """1, 4 and 16 hands per block of the packed-FMA kernel, ragged tails; from 4 096 hands with vertices on (>= 16 per image) the matrix-core
    kernel: 32-hand blocks that straddle two or three images, a ragged last block, the 10-vertex last tile"""
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
    if n >= 4096 and per_img >= 16:                         # the same launch on the packed-FMA kernel (no tiled table): joints bit-identical
        M2 = ops.Mano(assets['mano'], 'cuda')
        M2.c.posedirs_mfma = None
        v2, j2 = M2.fk(pose.cuda(), ctx, per_img, True)
        assert torch.equal(j2, joints)
        # observed bit-identical: the fp32 MFMA accumulates its k steps as the same chain of fused multiply-adds; not relied upon
        assert float((v2 - verts).abs().max()) < 3e-7


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
    ev = torch.gather(x, 1, si[:, :k]).permute(0, 2, 1)      # a NaN ranks as +inf and is returned as itself (torch.topk's values)
    assert torch.equal(torch.isnan(val), torch.isnan(ev)) and torch.equal(val[~torch.isnan(val)], ev[~torch.isnan(ev)])
    assert (val[~torch.isnan(val)] == sv[:, :k].permute(0, 2, 1)[~torch.isnan(ev)]).all()
    assert idx.min() >= 0 and idx.max() < n


def _window_boxes(g, n, size):
    """Boxes in image coordinates (the RoIAligns use spatial_scale 1/4): ordinary ones plus the corner cases of the window rule --
    thinner than one map pixel, beyond the map on either side, covering the whole image, integer and almost-integer edges."""
    c = torch.rand(n, 2, generator=g) * size
    h = torch.rand(n, 2, generator=g) * size * 0.45 + 2
    b = torch.cat([c - h, c + h], 1)
    special = torch.tensor([[10.0, 10.0, 10.5, 200.0], [-50.0, -30.0, 20.0, 40.0], [200.0, 180.0, 400.0, 300.0], [0.0, 0.0, size, size],
                            [40.0, 80.0, 160.0, 159.99999], [300.0, 300.0, 320.0, 320.0], [-40.0, -40.0, -10.0, -10.0], [64.0, 64.0, 64.0, 64.0]])
    b[:special.shape[0]] = special
    return b


@pytest.mark.parametrize('C,k', [(256, 3), (64, 1), (36, 3)])
def test_roi_windows_give_bit_identical_roi_align(C, k):
    """Demand-driven FPN output (vpho_roi_windows_i32 + pixel-list convolution + vpho_roi_align_window_nhwc_f32): the convolution
    computed only on the window pixels and read through the window table gives exactly the RoIAlign outputs of the full map, for both
    boxes of an image, with and without the W flip -- and every window pixel holds the full map's value."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(C + k)
    N, H, W, R = 12, 64, 64, 32
    x = torch.randn(N, H, W, 32, generator=g).cuda()
    w = (torch.randn(C, k * k * 32, generator=g) * 0.1).cuda()
    b = torch.randn(C, generator=g).cuda()
    ba, bb = _window_boxes(g, N, 256.0).cuda(), _window_boxes(torch.Generator().manual_seed(5), N, 256.0).roll(3, 0).cuda()
    flip = (torch.arange(N) % 2).to(torch.uint8).cuda()
    full = ops.conv2d_nhwc(x, w, b, kh=k, kw=k, pad=k // 2)
    for boxes_b in (bb, None):
        win = ops.roi_windows(ba, boxes_b, N, H, W, 0.25)
        rows = ops.conv2d_nhwc(x, w, b, kh=k, kw=k, pad=k // 2, rows=win)
        scat, mask = win.to_map(rows)
        assert torch.equal(scat[mask], full[mask])
        n_rows = int(win.count)
        wins = win.wins.cpu()
        assert n_rows == int((wins[:, 3] * wins[:, 4]).sum()) and n_rows == int(mask.sum()) and 0 < n_rows < N * H * W
        assert torch.equal(wins[:, 0], torch.cumsum(wins[:, 3] * wins[:, 4], 0) - wins[:, 3] * wins[:, 4])
        # the same windows dilated by the 3x3 halo, results stored in place (lateral convolution + top-down add of that level)
        halo = ops.roi_windows(ba, boxes_b, N, H, W, 0.25, dilate=1)
        _, hmask = halo.to_map(torch.zeros(N * H * W, 1).cuda())
        assert bool((torch.nn.functional.max_pool2d(mask.float()[:, None], 3, 1, 1)[:, 0].bool() <= hmask).all())
        inplace = torch.full((N, H, W, C), -7.0).cuda()
        ops.conv2d_nhwc(x, w, b, kh=k, kw=k, pad=k // 2, rows=halo, rows_scatter=True, out=inplace)
        assert torch.equal(inplace[hmask], full[hmask]) and bool((inplace[~hmask] == -7.0).all())
        small = torch.randn(N, H // 2, W // 2, C, generator=torch.Generator().manual_seed(3)).cuda()
        up_full = ops.resize_bilinear_nhwc(small, H, W, out=full.clone(), accumulate=True)
        up_rows = ops.resize_bilinear_nhwc(small, H, W, out=full.clone(), accumulate=True, rows=halo)
        assert torch.equal(up_rows[hmask], up_full[hmask]) and torch.equal(up_rows[~hmask], full[~hmask])
        for boxes in ((ba,) if boxes_b is None else (ba, boxes_b)):
            for fl in (None, flip):
                ref = ops.roi_align_nhwc(full, boxes, R, 0.25, flip_w=fl)
                got = ops.roi_align_nhwc(rows, boxes, R, 0.25, flip_w=fl, win=win)
                assert torch.equal(got, ref)
        # one pooling pass, two destinations (the object branch: plain crop + flipped crop into a wider buffer at a channel offset)
        wide = torch.full((N, R, R, C + 8), -3.0).cuda()
        plain = ops.roi_align_dual_nhwc(rows, ba, R, 0.25, win, wide, flip_w2=flip, c_off2=4)
        assert torch.equal(plain, ops.roi_align_nhwc(full, ba, R, 0.25))
        assert torch.equal(wide[..., 4:4 + C], ops.roi_align_nhwc(full, ba, R, 0.25, flip_w=flip))
        assert bool((wide[..., :4] == -3.0).all()) and bool((wide[..., 4 + C:] == -3.0).all())
    # a box whose window is the whole map: every pixel listed once, in order
    whole = torch.tensor([[0.0, 0.0, 256.0, 256.0]] * N).cuda()
    win = ops.roi_windows(whole, None, N, H, W, 0.25)
    assert int(win.count) == N * H * W and torch.equal(win.row_map.cpu(), torch.arange(N * H * W, dtype=torch.int32))


@pytest.mark.parametrize('Cin,Cout', [(32, 64), (64, 256)])
def test_winograd_on_roi_windows_is_bit_identical_to_the_full_map(Cin, Cout):
    """vpho_conv3x3_winograd_rows_nhwc_f32: the tiles of every image's even grid that touch its window, written as the compact window
    matrix, equal the full-map Winograd launch bit for bit on every window pixel -- also when everything outside the window dilated by
    the 3x3 halo is NaN (a tile row / column that only serves pixels outside the window never mixes into the ones inside) -- and
    RoIAlign through the window table gives the full map's result."""
    from vpho_amd import ops
    from vpho_amd.model.pack import winograd_weights
    g = torch.Generator().manual_seed(Cin + Cout)
    N, H, W, R = 10, 64, 64, 32
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.1).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    u = winograd_weights(w)
    full = ops.conv3x3_winograd(x, u, b, out_slope=0.3)
    ba, bb = _window_boxes(g, N, 256.0), _window_boxes(torch.Generator().manual_seed(9), N, 256.0).roll(2, 0).cuda()
    ba[8], ba[9] = torch.tensor([44.0, 28.0, 150.0, 121.0]), torch.tensor([101.0, 36.5, 149.0, 200.0])      # windows that start on odd pixels
    ba = ba.cuda()
    for boxes_b in (bb, None):
        win = ops.roi_windows(ba, boxes_b, N, H, W, 0.25)
        halo = ops.roi_windows(ba, boxes_b, N, H, W, 0.25, dilate=1)
        _, hmask = halo.to_map(torch.zeros(N * H * W, 1).cuda())
        xn = torch.where(hmask[..., None], x, torch.full_like(x, float('nan')))
        rows = torch.full((N * H * W, Cout), -7.0).cuda()
        ops.conv3x3_winograd(xn, u, b, out_slope=0.3, rows=win, out=rows)
        n_rows = int(win.count)
        scat, mask = win.to_map(rows)
        assert torch.equal(scat[mask], full[mask]) and bool((rows[n_rows:] == -7.0).all())
        wins, tiles = win.wins.cpu(), win.tiles().cpu()
        per = (((wins[:, 1] + wins[:, 4] - 1) >> 1) - (wins[:, 1] >> 1) + 1) * (((wins[:, 2] + wins[:, 3] - 1) >> 1) - (wins[:, 2] >> 1) + 1)
        assert torch.equal(tiles[1:], torch.cumsum(per, 0).int()) and int(tiles[0]) == 0
        assert boxes_b is not None or bool((wins[8:, 1:3] % 2 == 1).all())          # the odd-origin windows really are (no second box there)
        got = ops.roi_align_nhwc(rows, ba, R, 0.25, win=win)
        assert torch.equal(got, ops.roi_align_nhwc(full, ba, R, 0.25))
    # through ops.conv3x3 (weights transformed on first use); shapes the Winograd kernel does not take fall back to the direct pixel list
    win = ops.roi_windows(ba, None, N, H, W, 0.25)
    assert torch.equal(ops.conv3x3(x, w, b, out_slope=0.3, rows=win)[:int(win.count)], win_rows(full, win))
    n = int(win.count)
    assert torch.equal(ops.conv3x3(x, w, b, winograd=False, rows=win)[:n], ops.conv2d_nhwc(x, w, b, kh=3, kw=3, pad=1, rows=win)[:n])


def win_rows(full, win):
    n = int(win.count)
    return full.reshape(-1, full.shape[-1])[win.row_map[:n].long()]


def test_features_with_and_without_roi_windows_are_identical(model_cpu, assets):
    """Engine.features with the FPN outputs restricted to the RoI windows (default) against VPHO_ROI_WINDOW=0 (full maps): every
    tensor downstream of the RoIAligns is bit-identical."""
    import copy
    from vpho_amd.model.engine import Engine
    from vpho_amd.synth import synth_batch
    m = copy.deepcopy(model_cpu).cuda().eval()
    data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(6, assets, seed=11).items()}
    eng = Engine(m)
    assert eng.roi_window
    with torch.no_grad():
        a = eng.features(data)
        eng.roi_window = False
        b = eng.features(data)
    assert a['roi_win_hand'] is not None and b['roi_win_hand'] is None and b['hand_feat'].dim() == 4 and a['hand_feat'].dim() == 2
    frac = int(a['roi_win_hand'].count) / (6 * 64 * 64)
    assert 0.1 < frac < 1.0
    for k in ('hf_hr', 'enc_in_hand', 'enc_in_obj', 'hm_hand_nhwc', 'hm_obj_nhwc', 'encoding_hand', 'encoding_obj', 'mano_pose', 'mano_shape',
              'reg_hand_vert', 'tok_hand', 'tok_obj', 'force_local'):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('rows,cin,cout,slope', [(64, 1024, 1024, 0.01), (5, 1024, 512, 0.01), (64, 512, 96, 1.0), (3, 512, 10, 1.0), (130, 100, 7, 0.0)])
def test_linear_with_fp64_accumulation_is_the_correctly_rounded_product(rows, cin, cout, slope):
    """vpho_linear_acc64_f32 (round 6, the regression head): products and sum in double, one rounding -- equal to torch's float64 result
    rounded to fp32 up to one ulp of the output, batch-invariant, and closer to float64 than the fp32-MFMA GEMM of the same layer."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g).cuda()
    w = (torch.randn(cout, cin, generator=g) * (1.0 / cin) ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    y = ops.linear(x, w, b, out_slope=slope, acc64=True)
    ref = torch.nn.functional.leaky_relu(x.double() @ w.double().t() + b.double(), slope)
    err = (y.double() - ref).abs().max().item()
    assert err <= 1.2e-7 * max(1.0, ref.abs().max().item()), err
    assert torch.equal(y[:1], ops.linear(x[:1].contiguous(), w, b, out_slope=slope, acc64=True))            # a row does not depend on the batch
    if cin % 4 == 0:
        y32 = ops.linear(x, w, b, out_slope=slope)
        assert err <= (y32.double() - ref).abs().max().item()
