"""conv_igemm (fp32 MFMA implicit GEMM) against torch CPU fp32 convolutions.  Tolerance: fp32 accumulation-order
differences only -> 2e-5 relative to the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(size=shape) * scale).astype(np.float32))


def _close(got, ref, tol=2e-5):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'max abs err {err:.3e} vs scale {ref.abs().max().item():.3e}'


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 1, 1, 0),
    (2, 16, 16, 64, 128, 3, 1, 1),
    (3, 17, 13, 32, 21, 3, 2, 1),        # odd sizes, Cout tail, small tile
    (1, 32, 32, 4, 64, 7, 2, 3),         # stem: Cin padded 3->4, K=196 (K tail)
    (8, 32, 32, 128, 256, 3, 1, 1),      # big-tile path (>=192 tiles of 128x128)
    (4, 64, 64, 256, 128, 1, 1, 0),      # big-tile, 1x1
    (5, 9, 9, 280, 256, 1, 1, 0),        # Cin not multiple of 32
]


@pytest.mark.parametrize('case', CASES)
def test_conv_matches_torch(case):
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    N, H, W, Cin, Cout, k, st, pad = case
    x = _rand((N, Cin, H, W), 1)
    w = _rand((Cout, Cin, k, k), 2, (2.0 / (Cin * k * k)) ** 0.5)
    b = _rand((Cout,), 3)
    ref = F.conv2d(x, w, b, st, pad)
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda(), kh=k, kw=k, stride=st, pad=pad)
    _close(y.permute(0, 3, 1, 2), ref)


def test_conv_epilogue_residual_lrelu_and_prologue_affine():
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    N, H, W, Cin, Cout = 2, 16, 16, 128, 256
    x, w, b = _rand((N, Cin, H, W), 4), _rand((Cout, Cin, 1, 1), 5, 0.1), _rand((Cout,), 6)
    res = _rand((N, Cout, H, W), 7)
    sc, sh = _rand((Cin,), 8).abs() + 0.5, _rand((Cin,), 9)
    pre = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.01)
    ref = F.leaky_relu(F.conv2d(pre, w, b) + res, 0.01)
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda(),
                        res=res.permute(0, 2, 3, 1).contiguous().cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                        in_slope=0.01, out_slope=0.01)
    _close(y.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('N,H,Cin,Cout', [(16, 32, 64, 256), (16, 32, 32, 256), (8, 16, 128, 512), (5, 13, 96, 192), (3, 9, 256, 64), (64, 8, 512, 2048)])
def test_residual_requested_before_the_last_k_stage(N, H, Cin, Cout):
    """round 4: the residual tile of a 1x1 expansion (conv3 of every bottleneck, backbone_FPN_HFL.py:311-350) is loaded in front of the last
    k stage instead of after it.  K = 32 (one stage: requested before the only stage), 64, 96 ... 512 in all three tile classes, ragged
    pixel counts and Cout tails; against torch, and bit-identical to the round-3 order of the same kernel (VPHO_CONV_DBG=8)."""
    import os
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    x, w, b = _rand((N, Cin, H, H), 50), _rand((Cout, Cin, 1, 1), 51, (2.0 / Cin) ** 0.5), _rand((Cout,), 52)
    res = _rand((N, Cout, H, H), 53)
    ref = F.leaky_relu(F.conv2d(x, w, b) + res, 0.01)
    xg, wg, bg, rg = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda(), res.permute(0, 2, 3, 1).contiguous().cuda()
    y = ops.conv2d_nhwc(xg, wg, bg, res=rg, out_slope=0.01)
    _close(y.permute(0, 3, 1, 2), ref)
    os.environ['VPHO_CONV_DBG'] = '8'
    try:
        y_late = ops.conv2d_nhwc(xg, wg, bg, res=rg, out_slope=0.01)
    finally:
        os.environ.pop('VPHO_CONV_DBG', None)
    assert torch.equal(y, y_late)


@pytest.mark.parametrize('N,H,Cin,Cout', [(16, 32, 256, 256), (9, 31, 128, 64), (3, 17, 512, 128), (2, 8, 544, 64)])
def test_prologue_affine_on_the_direct_to_lds_kernel(N, H, Cin, Cout):
    """Pre-activation BN + LeakyReLU applied to the fragments as they are read from LDS (1x1, Cin % 32 == 0, Cin <= 512) in all
    three tile classes incl. ragged pixel counts; Cin = 544 exceeds the table and takes the register-staged kernel."""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    x, w, b = _rand((N, Cin, H, H), 40), _rand((Cout, Cin, 1, 1), 41, (2.0 / Cin) ** 0.5), _rand((Cout,), 42)
    sc, sh = _rand((Cin,), 43).abs() + 0.5, _rand((Cin,), 44)
    pre = F.leaky_relu(x * sc[None, :, None, None] + sh[None, :, None, None], 0.01)
    ref = F.leaky_relu(F.conv2d(pre, w, b), 0.01)
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda(), in_scale=sc.cuda(), in_shift=sh.cuda(),
                        in_slope=0.01, out_slope=0.01)
    _close(y.permute(0, 3, 1, 2), ref)


def test_prologue_affine_pads_with_zero_after_activation():
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    x, w = _rand((1, 8, 6, 6), 10), _rand((16, 8, 3, 3), 11, 0.2)
    sc, sh = _rand((8,), 12).abs() + 0.5, _rand((8,), 13) + 1.0
    pre = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    ref = F.conv2d(pre, w, None, 1, 1)
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), None, kh=3, kw=3, pad=1,
                        in_scale=sc.cuda(), in_shift=sh.cuda(), in_slope=0.0)
    _close(y.permute(0, 3, 1, 2), ref)


def test_transposed_conv_as_four_phase_convs():
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_deconv4x4s2
    N, Cin, Cout, H = 2, 32, 16, 8
    x, w = _rand((N, Cin, H, H), 14), _rand((Cin, Cout, 4, 4), 15, 0.1)
    ref = F.conv_transpose2d(x, w, None, 2, 1)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.empty((N, 2 * H, 2 * H, Cout), device='cuda')
    for (py, px), (wp, pady, padx) in pack_deconv4x4s2(w).items():
        ops.conv2d_nhwc(xg, wp.cuda(), None, kh=2, kw=2, pad_y=pady, pad_x=padx, out_hw=(H, H),
                        out_view=(y, 2 * H * 2 * H * Cout, 2 * 2 * H * Cout, 2 * Cout, (py * 2 * H + px) * Cout))
    _close(y.permute(0, 3, 1, 2), ref)


def test_linear_and_small_rows():
    from vpho_amd import ops
    x, w, b = _rand((3, 1024), 16), _rand((96, 1024), 17, 0.03), _rand((96,), 18)
    _close(ops.linear(x.cuda(), w.cuda(), b.cuda(), out_slope=0.01), F.leaky_relu(F.linear(x, w, b), 0.01))


def test_bad_arguments_raise():
    from vpho_amd import ops
    x = torch.zeros((1, 4, 4, 6), device='cuda')
    with pytest.raises(ops.VphoError):
        ops.conv2d_nhwc(x, torch.zeros((8, 6), device='cuda'))           # Cin % 4 != 0
    with pytest.raises(ops.VphoError):
        ops.conv2d_nhwc(torch.zeros((1, 4, 4, 8)), torch.zeros((8, 8)))  # CPU tensors: no fallback


def test_tensors_beyond_4gb_are_refused():
    """The kernels address x / w with 32-bit byte offsets: an operand of 4.3 GB is refused with an error before any launch"""
    from vpho_amd import ops
    x = torch.empty(2, 8192, 8192, 8, device='cuda')
    assert x.numel() * 4 > 4.2e9
    w = torch.zeros(8, 8, device='cuda')
    with pytest.raises(ops.VphoError, match='3.9 GB'):
        ops.conv2d_nhwc(x, w, None)


def test_mixed_uniform_and_generic_tap_paths_agree():
    """Cin % 32 == 0 takes the wave-uniform tap path, the same layer with VPHO-independent channel padding (Cin = 36 of ld 64) the
    generic one: both against torch, with padding on all four sides, stride 2 and a ragged pixel count."""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    for cin in (64, 36):
        x = _rand((3, cin, 15, 11), 7)
        w = _rand((40, cin, 3, 3), 8, (2.0 / (cin * 9)) ** 0.5)
        ref = F.conv2d(x, w, None, 2, 1)
        y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), None, kh=3, kw=3, stride=2, pad=1)
        _close(y.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(8, 32, 32, 128, 128), (4, 16, 16, 256, 256), (3, 8, 8, 512, 64), (2, 4, 6, 16, 64), (5, 64, 64, 64, 64)])
def test_winograd_3x3_matches_fp64_and_the_direct_kernel(N, H, W, Cin, Cout, capsys):
    """Winograd F(2x2,3x3) kernel (csrc/conv_winograd.hip; the default for the 3x3 / stride-1 layers of the plan): bias + LeakyReLU epilogue, zero padding at the borders, tile counts
    that do not fill a workgroup; error against an fp64 convolution no larger than 2x the direct kernel's (observed: smaller)."""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv, winograd_weights
    g = torch.Generator().manual_seed(N * H + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.01)
    xg, wg, bg = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda()
    direct = ops.conv2d_nhwc(xg, wg, bg, kh=3, kw=3, pad=1, out_slope=0.01).permute(0, 3, 1, 2).cpu().double()
    wino = ops.conv3x3_winograd(xg, winograd_weights(wg), bg, out_slope=0.01).permute(0, 3, 1, 2).cpu().double()
    auto = ops.conv3x3(xg, wg, bg, out_slope=0.01, winograd=True).permute(0, 3, 1, 2).cpu().double()
    assert torch.equal(auto, wino)
    sc = ref.abs().max().item()
    ed, ew = (direct - ref).abs().max().item() / sc, (wino - ref).abs().max().item() / sc
    assert ew <= 2 * ed + 1e-9 and ew < 5e-6, (ed, ew)
    with capsys.disabled():
        print(f'\n[winograd] {(N, H, W, Cin, Cout)}: max error vs fp64: direct {ed:.2e}, winograd {ew:.2e}', end='')


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(4, 32, 32, 128, 128), (3, 8, 8, 512, 64), (2, 4, 6, 16, 64), (5, 64, 64, 64, 64), (2, 16, 16, 48, 192), (9, 16, 16, 256, 256)])
def test_two_waves_per_simd_winograd_is_bit_identical_to_the_one_wave_kernel(N, H, W, Cin, Cout):
    """round 4: conv_winograd8_kernel (8 waves, each frequency half of a 32 x 32 block in its own wave, producer roles alternating between
    the two waves of a SIMD, output-transform exchange through LDS; VPHO_WINO8=1) against conv_winograd_kernel (VPHO_WINO8=0): same k order,
    same transform expressions -> bit-identical, incl. Cin = 16 (one super-stage), an odd number of super-stages (Cin = 48), ragged tile
    blocks and Cout = 192 (three channel blocks: the non-XCD block order)."""
    import os
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv, winograd_weights
    x = _rand((N, H, W, Cin), 70).cuda()
    w = _rand((Cout, Cin, 3, 3), 71, (2.0 / (9 * Cin)) ** 0.5)
    b = _rand((Cout,), 72).cuda()
    u = winograd_weights(pack_conv(w).cuda())
    try:
        os.environ['VPHO_WINO8'] = '1'
        y8 = ops.conv3x3_winograd(x, u, b, 0.01)
        os.environ['VPHO_WINO8'] = '0'
        y4 = ops.conv3x3_winograd(x, u, b, 0.01)
    finally:
        os.environ.pop('VPHO_WINO8', None)
    assert torch.isfinite(y8).all() and float(y8.abs().max()) > 0.1
    assert torch.equal(y8, y4)
    ref = F.leaky_relu(F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), w.double(), b.cpu().double(), 1, 1), 0.01)
    _close(y8.permute(0, 3, 1, 2), ref, tol=3e-6)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(4, 32, 32, 128, 128), (9, 16, 16, 256, 256), (130, 8, 8, 512, 64), (5, 64, 64, 128, 64), (3, 64, 64, 16, 64), (2, 32, 16, 48, 192),
                                            (7, 8, 8, 128, 128), (1, 16, 64, 32, 64)])
def test_winograd_with_the_input_staged_through_lds_is_bit_identical(N, H, W, Cin, Cout):
    """round 5: conv_winograd_kernel<1> moves the block's unique input pixels of a super-stage into a third LDS region by
    `buffer_load ... lds` and lets every lane read its 4 x 4 patch from there (one 64-register patch set instead of two) -- against the
    register path (VPHO_WINO_STAGED=0).  Same transforms on the same values: bit-identical.  Maps of 64 / 32 / 16 / 8 columns (2 / 4 / 8 tile
    rows per block of one image, four whole images per block), image counts that leave the last block ragged or with absent images, one and
    three super-stages, Cout = 192 (the non-XCD block order), non-square maps."""
    import os
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv, winograd_weights
    x = _rand((N, H, W, Cin), 80).cuda()
    w = _rand((Cout, Cin, 3, 3), 81, (2.0 / (9 * Cin)) ** 0.5)
    b = _rand((Cout,), 82).cuda()
    u = winograd_weights(pack_conv(w).cuda())
    try:
        os.environ['VPHO_WINO8'] = '0'
        os.environ['VPHO_WINO_STAGED'] = '0'
        y_reg = ops.conv3x3_winograd(x, u, b, 0.01)
        os.environ['VPHO_WINO_STAGED'] = '1'
        y_lds = ops.conv3x3_winograd(x, u, b, 0.01)
        y_lds2 = ops.conv3x3_winograd(x, u, b, 0.01)
    finally:
        os.environ.pop('VPHO_WINO8', None)
        os.environ.pop('VPHO_WINO_STAGED', None)
    assert torch.isfinite(y_lds).all() and float(y_lds.abs().max()) > 0.1
    assert torch.equal(y_lds, y_reg) and torch.equal(y_lds2, y_reg)
    ref = F.leaky_relu(F.conv2d(x[:4].cpu().permute(0, 3, 1, 2).double(), w.double(), b.cpu().double(), 1, 1), 0.01)
    _close(y_lds[:4].permute(0, 3, 1, 2), ref, tol=3e-6)


@pytest.mark.parametrize('N,H,Cin,Cout', [(64, 16, 1024, 256), (8, 32, 512, 256), (3, 10, 64, 128), (2, 6, 96, 64)])
def test_upsampled_residual_in_the_epilogue_equals_the_separate_top_down_pass(N, H, Cin, Cout):
    """round 4: `_upsample_add(p, lateral(c))` (backbone_FPN_HFL.py:66-68,98-104) inside the lateral 1x1 convolution's epilogue
    (vpho_conv_desc.res_up): bit-identical to the convolution followed by resize_bilinear_nhwc(accumulate=True), in all three tile
    classes, odd coarse sizes (10 <- 5, 6 <- 3 and a non-2x ratio 10 <- 4), and against torch (F.interpolate, align_corners=False)."""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    x, w, b = _rand((N, Cin, H, H), 80), _rand((Cout, Cin, 1, 1), 81, (2.0 / Cin) ** 0.5), _rand((Cout,), 82)
    xg, wg, bg = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda()
    for h in ((H // 2, H // 2), (4, 4) if H == 10 else (H // 2, H // 2)):
        p = _rand((N, Cout, h[0], h[1]), 83)
        pg = p.permute(0, 2, 3, 1).contiguous().cuda()
        fused = ops.conv2d_nhwc(xg, wg, bg, res_up=pg)
        q = ops.conv2d_nhwc(xg, wg, bg)
        ops.resize_bilinear_nhwc(pg, H, H, out=q, accumulate=True)
        assert torch.equal(fused, q)
        ref = F.conv2d(x, w, b) + F.interpolate(p, size=(H, H), mode='bilinear', align_corners=False)
        _close(fused.permute(0, 3, 1, 2), ref)


def test_upsampled_residual_on_scattered_roi_windows():
    """the stride-4 FPN level: lateral convolution + top-down add only on the pixels of the dilated RoI windows, written in place"""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    N, H, Cin, Cout = 6, 64, 256, 256
    x, w, b = _rand((N, Cin, H, H), 90), _rand((Cout, Cin, 1, 1), 91, (2.0 / Cin) ** 0.5), _rand((Cout,), 92)
    xg, wg, bg = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda()
    pg = _rand((N, H // 2, H // 2, Cout), 93).cuda()
    g = torch.Generator().manual_seed(5)
    c = torch.rand(N, 2, generator=g) * 160 + 48
    hw = torch.rand(N, 2, generator=g) * 60 + 20
    boxes = torch.cat([c - hw, c + hw], 1).cuda()
    win = ops.roi_windows(boxes, None, N, H, H, 0.25, dilate=1)
    base = torch.full((N, H, H, Cout), 7.0, device='cuda')
    fused = ops.conv2d_nhwc(xg, wg, bg, rows=win, rows_scatter=True, res_up=pg, out=base.clone())
    sep = ops.conv2d_nhwc(xg, wg, bg, rows=win, rows_scatter=True, out=base.clone())
    ops.resize_bilinear_nhwc(pg, H, H, out=sep, accumulate=True, rows=win)
    assert torch.equal(fused, sep)
    assert float((fused == 7.0).float().mean()) > 0.2            # pixels outside the windows are untouched


@pytest.mark.parametrize('N,H,C1,C2,Cout,stride2', [(64, 64, 64, 64, 256, 1), (16, 32, 128, 256, 512, 2), (5, 9, 96, 64, 192, 2), (128, 8, 512, 1024, 2048, 2), (2, 6, 32, 32, 64, 1)])
def test_second_input_of_a_1x1_convolution_is_the_merged_projection_shortcut(N, H, C1, C2, Cout, stride2):
    """round 4 (vpho_conv_desc.x2): conv3 and the projection shortcut of a stage-opening bottleneck (backbone_FPN_HFL.py:311-350: 1x1
    convolution + BatchNorm of the block input, stride 1 or 2) as ONE convolution over [conv2 output | block input].  Against torch
    (two convolutions + add + LeakyReLU) and against the two-launch form of the same kernels (one accumulation chain instead of sum,
    bias, add: fp32 rounding apart); all tile classes, odd sizes, strided second input."""
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    H2 = H * stride2 - (1 if stride2 == 2 and H % 2 else 0)            # an odd strided input size too: the last pixel read is (H - 1) * 2
    y, x = _rand((N, C1, H, H), 100), _rand((N, C2, H2, H2), 101)
    w3, wd = _rand((Cout, C1, 1, 1), 102, (1.0 / C1) ** 0.5), _rand((Cout, C2, 1, 1), 103, (1.0 / C2) ** 0.5)
    b3, bd = _rand((Cout,), 104), _rand((Cout,), 105)
    ref = F.leaky_relu(F.conv2d(y, w3, b3) + F.conv2d(x, wd, bd, stride=stride2), 0.01)
    yg, xg = y.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda()
    w_cat = torch.cat([pack_conv(w3), pack_conv(wd)], 1).contiguous().cuda()
    got = ops.conv2d_nhwc(yg, w_cat, (b3 + bd).cuda(), x2=xg, stride2=stride2, out_slope=0.01)
    _close(got.permute(0, 3, 1, 2), ref)
    r = ops.conv2d_nhwc(xg, pack_conv(wd).cuda(), bd.cuda(), stride=stride2)
    two = ops.conv2d_nhwc(yg, pack_conv(w3).cuda(), b3.cuda(), res=r, out_slope=0.01)
    assert float((got - two).abs().max()) <= 4e-6 * max(1.0, float(two.abs().max()))


@pytest.mark.parametrize('N,H,Cin,Cout,res,bias,x2', [
    (64, 32, 128, 512, True, True, 0),       # 2048 tiles on 512 slots: four tiles per workgroup, residual (conv3 of a bottleneck)
    (64, 16, 256, 1024, True, True, 0),      # two tiles per workgroup
    (64, 32, 128, 512, False, True, 256),    # merged projection shortcut (second input, strided)
    (33, 31, 64, 256, True, False, 0),       # ragged pixel tail (31713 pixels), no bias, two k stages
    (40, 32, 32, 256, False, False, 0),      # ONE k stage per tile
    (17, 16, 96, 128, True, True, 0),        # a single column of tiles, three k stages, ragged tail, at most one round (VPHO_CONV_PERS=2)
    (64, 32, 512, 128, False, True, 0),      # exactly one round of 512 tiles
])
def test_persistent_multi_tile_convolution_is_bit_identical_to_the_one_tile_kernel(N, H, Cin, Cout, res, bias, x2):
    """round 5 (conv_igemm_pers_kernel): a workgroup walks several 128 x 128 tiles, requests the next tile's first two k stages behind
    the current tile's last one and runs its epilogue through a wave-private LDS slice while they land; stores / residual loads through
    buffer resources.  The k order of every output element is conv_igemm_glds_kernel's: VPHO_CONV_PERS=0 (round 4's kernel) and = 2
    (the persistent kernel for every launch it serves) must agree bit for bit; and both agree with torch."""
    import os
    from vpho_amd import ops
    g = torch.Generator().manual_seed(N * 7 + Cin)
    x = torch.randn(N, H, H, Cin, generator=g).cuda()
    kw = {}
    K = Cin
    if x2:
        kw = dict(x2=torch.randn(N, 2 * H, 2 * H, x2, generator=g).cuda(), stride2=2)
        K = Cin + x2
    w = (torch.randn(Cout, K, generator=g) * (1.0 / K) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda() if bias else None
    r = torch.randn(N, H, H, Cout, generator=g).cuda() if res else None
    out = {}
    for mode in ('0', '2'):
        os.environ['VPHO_CONV_PERS'] = mode
        try:
            out[mode] = ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=r, **kw).clone()
            torch.cuda.synchronize()
        finally:
            os.environ.pop('VPHO_CONV_PERS', None)
    assert torch.isfinite(out['2']).all() and float(out['2'].abs().max()) > 0
    assert torch.equal(out['0'], out['2']), float((out['0'] - out['2']).abs().max())
    # against an fp64 product of the same operands
    xs = x.reshape(-1, Cin).double()
    ref = xs @ w[:, :Cin].double().t()
    if x2:
        ref = ref + kw['x2'][:, ::2, ::2][:, :H, :H].reshape(-1, x2).double() @ w[:, Cin:].double().t()
    if bias:
        ref = ref + b.double()
    if res:
        ref = ref + r.reshape(-1, Cout).double()
    ref = torch.where(ref > 0, ref, 0.01 * ref)
    assert float((out['2'].reshape(-1, Cout).double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # into a channel slice of a wider buffer (y_sx != Cout: concatenation targets), written in place
    wide = torch.full((N, H, H, Cout + 64), 7.0, device='cuda')
    os.environ['VPHO_CONV_PERS'] = '2'
    try:
        ld = Cout + 64
        ops.conv2d_nhwc(x, w, b, out_slope=0.01, res=r, out_view=(wide, H * H * ld, H * ld, ld, 32), **kw)
    finally:
        os.environ.pop('VPHO_CONV_PERS', None)
    assert torch.equal(wide[..., 32:32 + Cout], out['0']) and float(wide[..., :32].min()) == 7.0 and float(wide[..., 32 + Cout:].max()) == 7.0


@pytest.mark.parametrize('N,H,Cin,Cout,k,stride,opt', [
    (64, 16, 256, 1024, 1, 1, 'res'),          # conv3 of a layer3 bottleneck: both branches = 2048 tiles (persistent walk per group)
    (64, 16, 1024, 256, 1, 1, ''),             # conv1: 256 tiles per branch -- half the slots alone, one full round together
    (64, 32, 256, 128, 1, 1, 'shared'),        # the first block of layer2: both branches read the SAME block input
    (64, 32, 128, 512, 1, 2, 'x2shared'),      # ... and its conv3 with the merged projection shortcut of that shared input
    (64, 32, 128, 128, 3, 2, ''),              # the strided 3x3 of a stage-opening block (direct kernel)
    (64, 8, 2048, 256, 1, 1, 'res_up'),        # FPN lateral with the top-down add in its epilogue
    (5, 32, 284, 256, 1, 1, ''),               # encoder projection: 284 input channels (not a multiple of 32), odd batch
    (5, 16, 128, 64, 1, 1, 'pre'),             # encoder block conv1: pre-activation affine per group, Cout 64
    (3, 2, 128, 256, 1, 1, 'res'),             # a 2 x 2 map: 12 pixels per group, one ragged tile each
])
def test_grouped_launch_is_bit_identical_to_one_launch_per_group(N, H, Cin, Cout, k, stride, opt):
    """vpho_conv_desc.groups (round 6): the twin hand / object branches as ONE launch, blockIdx.y = group.  Whatever tile class the doubled
    tile count selects, every output element's k order is unchanged: the grouped result equals the two single launches bit for bit."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(N * 11 + Cin + Cout)
    G = 2
    shared = opt == 'shared'
    x = torch.randn((N if shared else G * N), H, H, Cin, generator=g).cuda()
    K = k * k * Cin
    kw, kw_g = {}, [{}, {}]
    OH = (H + 2 * (k // 2) - k) // stride + 1
    if opt == 'x2shared':
        x2 = torch.randn(N, stride * OH, stride * OH, 256, generator=g).cuda()
        K += 256
        kw = dict(x2=x2, stride2=stride, x2_shared=True)
        kw_g = [dict(x2=x2, stride2=stride)] * 2
    w = (torch.randn(G, Cout, K, generator=g) * (1.0 / K) ** 0.5).cuda()
    b = torch.randn(G, Cout, generator=g).cuda()
    if opt == 'res':
        r = torch.randn(G * N, OH, OH, Cout, generator=g).cuda()
        kw = dict(res=r)
        kw_g = [dict(res=r[:N]), dict(res=r[N:])]
    if opt == 'res_up':
        ru = torch.randn(G * N, H // 2, H // 2, Cout, generator=g).cuda()
        kw = dict(res_up=ru)
        kw_g = [dict(res_up=ru[:N].contiguous()), dict(res_up=ru[N:].contiguous())]
    if opt == 'pre':
        sc, sh = torch.rand(G, Cin, generator=g).cuda() + 0.5, torch.randn(G, Cin, generator=g).cuda()
        kw = dict(in_scale=sc, in_shift=sh, in_slope=0.01)
        kw_g = [dict(in_scale=sc[i].contiguous(), in_shift=sh[i].contiguous(), in_slope=0.01) for i in range(G)]
    conv = dict(kh=k, kw=k, stride=(1 if opt == 'x2shared' else stride), pad=k // 2, out_slope=0.01)
    if opt == 'x2shared':
        x = torch.randn(G * N, OH, OH, Cin, generator=g).cuda()       # conv3's own input is already at the block's output resolution
    got = ops.conv2d_nhwc(x, w, b, groups=G, x_shared=shared, **conv, **kw)
    torch.cuda.synchronize()
    assert got.shape[0] == G * N and torch.isfinite(got).all() and float(got.abs().max()) > 0
    for i in range(G):
        xi = x if shared else x[i * N:(i + 1) * N]
        one = ops.conv2d_nhwc(xi, w[i].contiguous(), b[i].contiguous(), **conv, **kw_g[i])
        assert torch.equal(got[i * N:(i + 1) * N], one), (i, float((got[i * N:(i + 1) * N] - one).abs().max()))


@pytest.mark.parametrize('N,H,W,Cin,Cout,shared', [(64, 32, 32, 128, 128, False), (64, 16, 16, 256, 256, False), (5, 8, 8, 256, 256, False), (3, 4, 6, 128, 128, False),
                                                   (5, 2, 2, 128, 128, False), (4, 32, 32, 256, 256, True)])
def test_grouped_winograd_is_bit_identical_to_one_launch_per_group(N, H, W, Cin, Cout, shared):
    """the same for the Winograd kernel (vpho_conv3x3_winograd_grouped_nhwc_f32): tile blocks of a group never hold another group's tiles"""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(N + H + Cin)
    G = 2
    x = torch.randn((N if shared else G * N), H, W, Cin, generator=g).cuda()
    w = (torch.randn(G, Cout, 9 * Cin, generator=g) * (1.0 / (9 * Cin)) ** 0.5).cuda()
    b = torch.randn(G, Cout, generator=g).cuda()
    got = ops.conv3x3(x, w, b, out_slope=0.01, groups=G, x_shared=shared)
    torch.cuda.synchronize()
    for i in range(G):
        xi = x if shared else x[i * N:(i + 1) * N]
        one = ops.conv3x3(xi, w[i].contiguous(), b[i].contiguous(), out_slope=0.01)
        assert torch.equal(got[i * N:(i + 1) * N], one), (i, float((got[i * N:(i + 1) * N] - one).abs().max()))

