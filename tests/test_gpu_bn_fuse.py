"""BatchNorm reductions in the convolution epilogues (ABI 12: vpho_conv_desc.stats / bn_x, vpho_conv3x3_winograd_stats_nhwc_f32,
vpho_bn_train_*_stats_f32; the training path's Bottleneck / Residual / HeadHeatmap2 blocks, backbone_FPN_HFL.py:330-350,
encoding.py:21-36, head_inplane.py:40-58 under model.train()).  The fused path must give the stand-alone path's results to
rounding (the reductions are re-associated: per-tile fp32 sums, then fp64), its convolution outputs bit for bit, and must be
bit-reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).cuda()


# (N, H, W, cin, cout, k): 128x128 / 128x64 / 64x64 tile classes of the direct kernel, ragged pixel counts and channel tails, and the Winograd shapes
SHAPES = [(64, 32, 32, 64, 256, 1), (16, 16, 16, 256, 64, 1), (2, 9, 7, 32, 40, 1), (3, 8, 8, 36, 132, 1), (64, 16, 16, 128, 128, 3),
          (4, 32, 32, 64, 64, 3), (2, 10, 6, 16, 64, 3), (2, 8, 8, 12, 20, 3),
          (3, 12, 12, 64, 128, 3), (2, 20, 12, 128, 64, 3)]      # 6 tiles per row: the Winograd kernel's register-path instantiations, ragged tile blocks


@pytest.mark.parametrize('N,H,W,cin,cout,k', SHAPES)
def test_forward_partial_sums_match_the_column_sums_of_the_output(N, H, W, cin, cout, k):
    from vpho_amd import ops
    x = _rand((N, H, W, cin), 1)
    w = _rand((cout, k * k * cin), 2, (1.0 / (k * k * cin)) ** 0.5)
    b = _rand((cout,), 3)
    f = ops.BnFuse()
    if k == 1:
        y = ops.conv2d_nhwc(x, w, b, bn=f)
        want = ops.conv2d_nhwc(x, w, b)
    else:
        y = ops.conv3x3_train(x, w, b, bn=f)
        want = ops.conv3x3_train(x, w, b)
    assert f.live(), 'the shape is one the fused epilogue serves'
    assert torch.equal(y, want)                                     # the epilogue's extra work does not touch the stored values
    part = f.stats[:f.rows].double()
    y2 = y.reshape(-1, cout).double()
    s, ss = y2.sum(0), (y2 * y2).sum(0)
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().numpy(), s.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(y2.abs().sum(0).max()))
    np.testing.assert_allclose(part[:, 1].sum(0).cpu().numpy(), ss.cpu().numpy(), rtol=2e-5)
    # bit-reproducible
    f2 = ops.BnFuse()
    (ops.conv2d_nhwc(x, w, b, bn=f2) if k == 1 else ops.conv3x3_train(x, w, b, bn=f2))
    assert f2.rows == f.rows and torch.equal(f2.stats[:f.rows], f.stats[:f.rows])
    # and the BatchNorm that consumes them agrees with the one that runs its own pass
    gamma, beta = _rand((cout,), 4).abs() + 0.5, _rand((cout,), 5)
    rm1, rv1, rm2, rv2 = (torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros(cout).cuda(), torch.ones(cout).cuda())
    a1, (m1, i1) = ops.bn_train_forward(y, gamma, beta, rm1, rv1, slope=0.01, partials=f)
    a2, (m2, i2) = ops.bn_train_forward(y, gamma, beta, rm2, rv2, slope=0.01)
    np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i1.cpu().numpy(), i2.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(a1.cpu().numpy(), a2.cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(rv1.cpu().numpy(), rv2.cpu().numpy(), rtol=2e-5)


@pytest.mark.parametrize('N,H,W,cin,cout,k', SHAPES)
def test_input_gradient_with_recomputed_gate_and_backward_sums(N, H, W, cin, cout, k):
    """y = conv(a), a = lrelu(bn(c)): the input-gradient convolution of dy gated by the sign recomputed from c must store exactly what the
    gate read from the stored activation stores, and leave sum da, sum da * xhat"""
    from vpho_amd import ops, conv_backward as CB
    c = _rand((N, H, W, cin), 11)
    gamma, beta = _rand((cin,), 12).abs() + 0.5, _rand((cin,), 13) * 0.3
    a, saved = ops.bn_train_forward(c, gamma, beta, slope=0.01)
    w = _rand((cout, k * k * cin), 14, (1.0 / (k * k * cin)) ** 0.5)
    dy = _rand((N, H, W, cout), 15)
    pad = 1 if k == 3 else 0
    want = CB.conv2d_dgrad(dy, w, (H, W), k, k, 1, pad, gate=(a, 0.01))
    f = ops.BnFuse(c, saved, gamma, beta)
    got = CB.conv2d_dgrad(dy, w, (H, W), k, k, 1, pad, gate=(a, 0.01), bn=f)
    if cin % 4:
        assert not f.live()
        return
    assert f.live()
    assert torch.equal(got, want)
    xh = ((c - saved[0]) * saved[1]).reshape(-1, cin).double()
    d2 = got.reshape(-1, cin).double()
    part = f.stats[:f.rows].double()
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().numpy(), d2.sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float(d2.abs().sum(0).max()))
    np.testing.assert_allclose(part[:, 1].sum(0).cpu().numpy(), (d2 * xh).sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float((d2 * xh).abs().sum(0).max()))
    dx1, dg1, db1 = ops.bn_train_backward(c, got, gamma, saved, partials=f)
    dx2, dg2, db2 = ops.bn_train_backward(c, got, gamma, saved)
    tol = lambda t: 2e-5 * float(t.abs().max()) + 1e-7
    np.testing.assert_allclose(dg1.cpu().numpy(), dg2.cpu().numpy(), rtol=1e-4, atol=tol(dg2))
    np.testing.assert_allclose(db1.cpu().numpy(), db2.cpu().numpy(), rtol=1e-4, atol=tol(db2))
    np.testing.assert_allclose(dx1.cpu().numpy(), dx2.cpu().numpy(), rtol=1e-3, atol=tol(dx2))


def test_many_partial_rows_go_through_the_two_level_finish():
    """more than 256 partial rows: the partial matrix itself is column-reduced first (vpho_bn_train_forward_stats_f32)"""
    from vpho_amd import ops
    x = _rand((64, 64, 64, 64), 21)
    w = _rand((64, 64), 22, 0.125)
    f = ops.BnFuse()
    y = ops.conv2d_nhwc(x, w, bn=f)
    assert f.live() and f.rows > 256
    gamma, beta = torch.ones(64).cuda(), torch.zeros(64).cuda()
    a1, (m1, i1) = ops.bn_train_forward(y, gamma, beta, slope=1.0, partials=f)
    a2, (m2, i2) = ops.bn_train_forward(y, gamma, beta, slope=1.0)
    np.testing.assert_allclose(m1.cpu().numpy(), m2.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i1.cpu().numpy(), i2.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(a1.cpu().numpy(), a2.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_switch_off_runs_the_stand_alone_passes(monkeypatch):
    from vpho_amd import ops
    monkeypatch.setattr(ops, 'FUSE_BN', False)
    f = ops.BnFuse()
    ops.conv2d_nhwc(_rand((2, 8, 8, 32), 1), _rand((64, 32), 2), bn=f)
    assert not f.live()


@pytest.mark.parametrize('N,H,W,cin,cout', [(64, 16, 16, 256, 64), (2, 9, 7, 40, 32), (16, 32, 32, 128, 32)])
def test_residual_block_gradient_with_stored_gate_and_bn3_sums(N, H, W, cin, cout):
    """the identity-shortcut bottleneck's dx = dgrad(dc1, conv1) + dsum, gated by the previous block's output (= this block's input,
    out = lrelu(bn3(c3) + shortcut): the sign cannot come from c3 alone), with the sums of that block's bn3 backward in the same epilogue"""
    from vpho_amd import ops, conv_backward as CB
    c3 = _rand((N, H, W, cin), 31)
    short = _rand((N, H, W, cin), 32)
    gamma, beta = _rand((cin,), 33).abs() + 0.5, _rand((cin,), 34) * 0.3
    out_prev, saved = ops.bn_train_forward(c3, gamma, beta, slope=0.01, res=short)
    w = _rand((cout, cin), 35, (1.0 / cin) ** 0.5)
    dc1, dsum = _rand((N, H, W, cout), 36), _rand((N, H, W, cin), 37)
    want = CB.conv2d_dgrad(dc1, w, (H, W), 1, 1, res=dsum, gate=(out_prev, 0.01))
    f = ops.BnFuse(c3, saved, gamma, beta, stored_gate=True)
    got = CB.conv2d_dgrad(dc1, w, (H, W), 1, 1, res=dsum, gate=(out_prev, 0.01), bn=f)
    assert f.live() and torch.equal(got, want)
    xh = ((c3 - saved[0]) * saved[1]).reshape(-1, cin).double()
    d2 = got.reshape(-1, cin).double()
    part = f.stats[:f.rows].double()
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().numpy(), d2.sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float(d2.abs().sum(0).max()))
    np.testing.assert_allclose(part[:, 1].sum(0).cpu().numpy(), (d2 * xh).sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float((d2 * xh).abs().sum(0).max()))


def test_layer_backward_is_the_same_with_and_without_the_fused_reductions(monkeypatch):
    """three identity bottlenecks behind a projection one (a ResNet stage in miniature): outputs, input gradient and every parameter
    gradient with the reductions in the convolution epilogues against the stand-alone passes"""
    from vpho_amd import ops, train_blocks as TB
    from vpho_amd.model.pack import pack_conv
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g)

    def block(cin, planes, down):
        bn = lambda c: dict(gamma=(rnd(c).abs() + 0.5).cuda(), beta=(rnd(c) * 0.2).cuda(), running_mean=torch.zeros(c).cuda(), running_var=torch.ones(c).cuda())
        p = dict(conv1=pack_conv(rnd(planes, cin, 1, 1) * cin ** -0.5).cuda(), conv2=pack_conv(rnd(planes, planes, 3, 3) * (9 * planes) ** -0.5).cuda(),
                 conv3=pack_conv(rnd(4 * planes, planes, 1, 1) * planes ** -0.5).cuda(), bn1=bn(planes), bn2=bn(planes), bn3=bn(4 * planes))
        if down:
            p['down'] = pack_conv(rnd(4 * planes, cin, 1, 1) * cin ** -0.5).cuda()
            p['bnd'] = bn(4 * planes)
        return p

    params = [block(64, 64, True)] + [block(256, 64, False) for _ in range(3)]
    x, dout = rnd(8, 16, 16, 64).cuda(), rnd(8, 16, 16, 256).cuda()

    def run():
        net = TB.FPNTrain.__new__(TB.FPNTrain)
        net.blocks = {'L': [(f'L.0.{i}', p, 1) for i, p in enumerate(params)]}
        net.calls = {}
        net.shapes = {}
        y = net._run_layer('L', x, 't')
        grads = {}
        seq = net.calls[('L', 't')]
        gated, f3, dy = False, None, dout
        outs = {}
        for i in range(len(seq) - 1, -1, -1):
            k, b = seq[i]
            fuse = TB.FUSE_LRELU_BWD and i > 0 and 'down' not in b.p
            f_prev = seq[i - 1][1].bn3_fuse() if fuse else None
            dy, gr = b.backward(dy, gated=gated, gate_input=fuse, bn_prev=f_prev, bn3=f3)
            gated, f3 = fuse, f_prev
            outs.update({f'{k}.{n}': v for n, v in gr.items()})
        return y, dy, outs

    y1, dx1, g1 = run()
    monkeypatch.setattr(ops, 'FUSE_BN', False)
    monkeypatch.setattr(TB, 'FUSE_LRELU_BWD', False)
    y0, dx0, g0 = run()
    # The two paths differ in the association of the batch sums, i.e. by ~1e-7 in every activation -- enough to flip the LeakyReLU gate of
    # an element that sits within 1e-7 of zero, and ONE flipped gate moves a channel's gradient sums by that element's whole gradient
    # (seen here: one flip in the second block, 6e-3 of a d beta, everything upstream of it shifted by ~1e-3).  The comparison is therefore
    # in the L2 norm: a flip is a 1e-3 effect, a wrong partial sum (a missed tile, a wrong lane) is a 1e-1 effect.
    rel = lambda a, b: float((a - b).double().norm() / b.double().norm().clamp_min(1e-30))
    assert rel(y1, y0) < 1e-5, rel(y1, y0)
    assert rel(dx1, dx0) < 1e-2, rel(dx1, dx0)
    assert g1.keys() == g0.keys()
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-2, (k, rel(g1[k], g0[k]))
    # the last block's gradients come before any possible flip upstream: there the two paths agree to rounding
    for k in g0:
        if k.startswith('L.0.3.'):
            assert rel(g1[k], g0[k]) < 1e-4, (k, rel(g1[k], g0[k]))


@pytest.mark.parametrize('rows,C', [(64 * 32 * 32, 128), (300, 36), (5000, 7)])
def test_batchnorm_backward_with_shortcut_add_and_column_sums(rows, C):
    """encoding.Residual's backward in one pass: dx = BatchNorm backward + d shortcut, and the column sums of dx (the bias gradient of the
    convolution in front) -- bit for bit the element-wise kernel + add, column sums to fp64 rounding"""
    from vpho_amd import ops
    x, dy, res = _rand((rows, C), 41), _rand((rows, C), 42), _rand((rows, C), 43)
    gamma, beta = _rand((C,), 44).abs() + 0.5, _rand((C,), 45)
    _, saved = ops.bn_train_forward(x, gamma, beta, slope=0.01)
    dx0, dg0, db0 = ops.bn_train_backward(x, dy, gamma, saved)
    dx1, dg1, db1, cs = ops.bn_train_backward(x, dy, gamma, saved, res=res, want_colsum=True)
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dx1, dx0 + res)
    want = dx1.double().sum(0)
    np.testing.assert_allclose(cs.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6 * float(dx1.abs().sum(0).max()))
    dx2, _, _, cs2 = ops.bn_train_backward(x, dy, gamma, saved, want_colsum=True)
    assert torch.equal(dx2, dx0) and torch.equal(ops.bn_train_backward(x, dy, gamma, saved, want_colsum=True)[3], cs2)


def test_phase_convolutions_append_their_sums():
    """maps written by several strided convolutions: the four phases of a stride-2 input gradient (gate recomputed from the BatchNorm input,
    backward sums) and the four phases of a transposed convolution (forward sums over the 2H x 2W map)"""
    from vpho_amd import ops, conv_backward as CB
    N, H, W, cin, cout = 4, 16, 16, 32, 48
    c = _rand((N, H, W, cin), 51)
    gamma, beta = _rand((cin,), 52).abs() + 0.5, _rand((cin,), 53) * 0.3
    a, saved = ops.bn_train_forward(c, gamma, beta, slope=0.01)
    w = _rand((cout, 9 * cin), 54, (9 * cin) ** -0.5)
    dy = _rand((N, H // 2, W // 2, cout), 55)
    want = CB.conv2d_dgrad(dy, w, (H, W), 3, 3, 2, 1, gate=(a, 0.01))
    f = ops.BnFuse(c, saved, gamma, beta)
    got = CB.conv2d_dgrad(dy, w, (H, W), 3, 3, 2, 1, gate=(a, 0.01), bn=f)
    assert f.live() and f.parts == 4 and torch.equal(got, want)
    xh = ((c - saved[0]) * saved[1]).reshape(-1, cin).double()
    d2 = got.reshape(-1, cin).double()
    part = f.stats[:f.rows].double()
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().numpy(), d2.sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float(d2.abs().sum(0).max()))
    np.testing.assert_allclose(part[:, 1].sum(0).cpu().numpy(), (d2 * xh).sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float((d2 * xh).abs().sum(0).max()))
    # forward: four 2x2 phase convolutions into one (N, 2H, 2W, co) map
    co = 64
    up = torch.empty((N, 2 * H, 2 * W, co), device='cuda')
    f2 = ops.BnFuse(parts=4)
    for py in (0, 1):
        for px in (0, 1):
            wp = _rand((co, 4 * cin), 60 + 2 * py + px, (4 * cin) ** -0.5)
            ops.conv2d_nhwc(c, wp, None, kh=2, kw=2, pad_y=1 - py, pad_x=1 - px, out_hw=(H, W),
                            out_view=(up, 4 * H * W * co, 4 * W * co, 2 * co, (py * 2 * W + px) * co), bn=f2)
    assert f2.live()
    u2 = up.reshape(-1, co).double()
    part = f2.stats[:f2.rows].double()
    np.testing.assert_allclose(part[:, 0].sum(0).cpu().numpy(), u2.sum(0).cpu().numpy(), rtol=2e-5, atol=2e-5 * float(u2.abs().sum(0).max()))
    np.testing.assert_allclose(part[:, 1].sum(0).cpu().numpy(), (u2 * u2).sum(0).cpu().numpy(), rtol=2e-5)
