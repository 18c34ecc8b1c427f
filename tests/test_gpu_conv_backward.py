"""Convolution backward building blocks against torch autograd of torch.nn.functional.conv2d (fp32, CPU)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(N, H, W, cin, cout, k, stride, pad, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    return x.detach(), w.detach(), dy, x.grad, w.grad


# ResNet / FPN / head shapes in miniature: 3x3 s1 p1, 1x1 s1, 3x3 s2 p1 (bottleneck conv2 of a down-sampling block),
# 1x1 s2 (its shortcut), 2x2 p0 and 7x7 s1 p3 for generality; channel counts not multiples of 32, ragged pixel counts
@pytest.mark.parametrize('N,H,W,cin,cout,k,stride,pad', [(2, 16, 16, 8, 12, 3, 1, 1), (3, 10, 14, 16, 8, 1, 1, 0), (2, 16, 12, 8, 16, 3, 2, 1),
                                                         (2, 8, 8, 12, 20, 1, 2, 0), (1, 9, 9, 4, 8, 2, 1, 0), (1, 12, 12, 4, 8, 7, 1, 3),
                                                         (4, 32, 32, 64, 64, 3, 1, 1), (2, 32, 32, 128, 64, 3, 2, 1),
                                                         # implicit TN weight-gradient kernel (Cin, Cout multiples of 4; the rest take the im2col path): ragged channel tails, both tile
                                                         # sizes, strides, split pixel ranges (8x64x64 pixels), a 1-pixel-row tail
                                                         (2, 9, 11, 36, 44, 3, 1, 1), (1, 16, 16, 128, 256, 1, 1, 0), (2, 14, 14, 256, 128, 3, 1, 1),
                                                         (2, 16, 16, 160, 136, 3, 2, 1), (3, 6, 10, 64, 32, 1, 2, 0), (8, 64, 64, 64, 64, 3, 1, 1),
                                                         (1, 1, 1, 32, 4, 1, 1, 0), (2, 8, 8, 284, 256, 1, 1, 0), (2, 32, 32, 4, 64, 7, 2, 3), (2, 8, 8, 6, 10, 3, 1, 1)])
def test_dgrad_and_wgrad_match_autograd(N, H, W, cin, cout, k, stride, pad):
    from vpho_amd import conv_backward as CB
    from vpho_amd.model.pack import pack_conv
    x, w, dy, dx_ref, dw_ref = _case(N, H, W, cin, cout, k, stride, pad, seed=N * 1000 + H * 10 + k)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyg = dy.permute(0, 2, 3, 1).contiguous().cuda()
    wp = pack_conv(w).cuda()
    dx = CB.conv2d_dgrad(dyg, wp, (H, W), k, k, stride, pad).permute(0, 3, 1, 2).cpu()
    dw = CB.conv2d_wgrad(xg, dyg, k, k, stride, pad).cpu().view(cout, k, k, cin).permute(0, 3, 1, 2)
    np.testing.assert_allclose(dx.numpy(), dx_ref.numpy(), atol=2e-5 * float(dx_ref.abs().max()), rtol=1e-4)
    np.testing.assert_allclose(dw.numpy(), dw_ref.numpy(), atol=2e-5 * float(dw_ref.abs().max()), rtol=1e-4)
    db = CB.conv2d_bias_grad(dyg).cpu()
    np.testing.assert_allclose(db.numpy(), dy.sum(dim=(0, 2, 3)).numpy(), atol=1e-4 * float(dy.abs().sum(dim=(0, 2, 3)).max()), rtol=1e-4)


def test_dgrad_through_the_weight_cache_follows_an_in_place_weight_write():
    """inside ``with DgradWeightCache():`` the input gradients read derived weight layouts.  A conv weight mutated in place with
    nobody calling refresh() must give the input gradient of the NEW weights (1x1 stride 1, 3x3 stride 2 phases), bit for bit the
    uncached path's."""
    from vpho_amd import conv_backward as CB
    g = torch.Generator().manual_seed(11)
    cache = CB.DgradWeightCache()
    cases = []
    for (cin, cout, k, stride, pad) in ((32, 64, 1, 1, 0), (32, 48, 3, 2, 1), (16, 32, 1, 2, 0)):
        w = (torch.randn(cout, k * k * cin, generator=g) * 0.1).cuda()
        dy = torch.randn(2, 8 // stride, 8 // stride, cout, generator=g).cuda()
        cases.append((w, dy, k, stride, pad))
    with cache:
        for w, dy, k, stride, pad in cases:
            CB.conv2d_dgrad(dy, w, (8, 8), k, k, stride, pad)
    cache.refresh()
    for w, *_ in cases:
        w.mul_(-0.5).add_(0.03)                              # an in-place write nobody announces
    with cache:
        got = [CB.conv2d_dgrad(dy, w, (8, 8), k, k, stride, pad) for w, dy, k, stride, pad in cases]
    assert cache.rebuilds >= 1
    want = [CB.conv2d_dgrad(dy, w, (8, 8), k, k, stride, pad) for w, dy, k, stride, pad in cases]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize('N,H,W,C,slope', [(2, 8, 8, 16, 1.0), (4, 16, 12, 33, 0.01), (64, 32, 32, 64, 0.01), (3, 1, 1, 8, 0.0)])
def test_batchnorm_training_mode_matches_autograd(N, H, W, C, slope):
    """nn.BatchNorm2d.train() (+ LeakyReLU) forward, running statistics and backward vs torch autograd"""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(N * 100 + C)
    x = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = torch.nn.functional.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    y = torch.nn.functional.leaky_relu(y, slope) if slope != 1.0 else y
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    rm_g, rv_g = rm.cuda(), rv.cuda()
    yg, saved = ops.bn_train_forward(nhwc(x), gamma.detach().cuda(), beta.detach().cuda(), rm_g, rv_g, slope=slope)
    np.testing.assert_allclose(yg.permute(0, 3, 1, 2).cpu().numpy(), y.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(rm_g.cpu().numpy(), rm_ref.numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(rv_g.cpu().numpy(), rv_ref.numpy(), atol=1e-6, rtol=1e-5)
    dyg = nhwc(dy)
    if slope != 1.0:
        dyg = ops.lrelu_bwd(dyg, yg, slope)
    dx, dgam, dbet = ops.bn_train_backward(nhwc(x), dyg, gamma.detach().cuda(), saved)
    tol = lambda ref: 3e-5 * float(ref.abs().max()) + 1e-7
    np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy(), atol=tol(x.grad), rtol=1e-4)
    np.testing.assert_allclose(dgam.cpu().numpy(), gamma.grad.numpy(), atol=tol(gamma.grad), rtol=1e-4)
    np.testing.assert_allclose(dbet.cpu().numpy(), beta.grad.numpy(), atol=tol(beta.grad), rtol=1e-4)


@pytest.mark.parametrize('rows,C', [(64 * 32 * 32, 64), (64 * 64 * 64, 256), (4 * 16 * 12, 33), (3, 8), (64 * 16 * 16, 1024)])
def test_in_launch_finish_of_the_column_reductions_equals_the_finishing_kernels(rows, C, monkeypatch):
    """The workgroup that draws the last arrival ticket of its column block finishes the block's channels inside the reduction launch
    (train_score.hip::col_reduce_kernel, VPHO_COL_FINISH=fused; the default stays the separate finishing kernels, which measured
    faster inside the training step).  Same chunk order as the finishing kernels: bit for bit the same
    statistics / gradients / column sums -- on the first launch of a stream, on repeated launches (the tickets go back to zero) and on
    two streams whose launches overlap."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(rows, C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    rm0, rv0 = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()

    def run():
        rm, rv = rm0.clone(), rv0.clone()
        y, saved = ops.bn_train_forward(x, gamma, beta, rm, rv, slope=0.01)
        dx, dgam, dbet = ops.bn_train_backward(x, dy, gamma, saved)
        return [y, saved[0], saved[1], rm, rv, dx, dgam, dbet, ops.colsum(dy)]

    monkeypatch.setenv('VPHO_COL_FINISH', 'separate')
    want = run()
    monkeypatch.setenv('VPHO_COL_FINISH', 'fused')
    for _ in range(3):
        got = run()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                outs.append(run())
    torch.cuda.synchronize()
    for got in outs:
        for a, b in zip(got, want):
            assert torch.equal(a, b)


@pytest.mark.parametrize('N,H,W,cin,cout', [(64, 64, 64, 64, 256), (64, 32, 32, 128, 512), (16, 32, 32, 64, 64), (64, 16, 16, 1024, 256)])
def test_weight_gradient_with_many_pixel_slices_is_deterministic_and_exact_to_rounding(N, H, W, cin, cout):
    """Small dW, many pixel slices (up to 256): the slices are summed by 4 or 16 threads per quad (each ascending over its slices, then
    ascending over the threads; conv_wgrad.hip::wgrad_reduce_grouped_kernel).  A fixed order: repeated launches are bit-identical, and
    the result is the fp64 product to fp32 rounding."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(N + cin)
    x = torch.randn(N, H, W, cin, generator=g).cuda()
    dy = torch.randn(N, H, W, cout, generator=g).cuda()
    got = ops.conv2d_wgrad_nhwc(x, dy, 1, 1, 1, 0, 0)
    for _ in range(3):
        assert torch.equal(ops.conv2d_wgrad_nhwc(x, dy, 1, 1, 1, 0, 0), got)
    ref = dy.reshape(-1, cout).double().t() @ x.reshape(-1, cin).double()
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize('tag,stride', [('id', 1), ('down', 2)])
def test_bottleneck_training_step_matches_reference_module(tag, stride):
    """Training-mode forward + backward of a whole Bottleneck (conv/BN(train)/LeakyReLU x3, shortcut, residual) vs the
    reference's own module under autograd (fixture by tests/golden/make_golden_bottleneck.py)."""
    import os
    from vpho_amd.model.pack import pack_conv
    from vpho_amd.train_blocks import BottleneckTrain
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_bottleneck.npz'))
    t = lambda k: torch.from_numpy(G[f'{tag}_{k}'])
    bn = lambda n: dict(gamma=t(f'init_{n}.weight').cuda(), beta=t(f'init_{n}.bias').cuda(),
                        running_mean=t(f'init_{n}.running_mean').cuda(), running_var=t(f'init_{n}.running_var').cuda())
    params = dict(conv1=pack_conv(t('init_conv1.weight')).cuda(), conv2=pack_conv(t('init_conv2.weight')).cuda(),
                  conv3=pack_conv(t('init_conv3.weight')).cuda(), bn1=bn('bn1'), bn2=bn('bn2'), bn3=bn('bn3'))
    names = {'conv1': 'conv1.weight', 'conv2': 'conv2.weight', 'conv3': 'conv3.weight'}
    bns = {'bn1': 'bn1', 'bn2': 'bn2', 'bn3': 'bn3'}
    if tag == 'down':
        params['down'] = pack_conv(t('init_downsample.0.weight')).cuda()
        params['bnd'] = bn('downsample.1')
        names['down'] = 'downsample.0.weight'
        bns['bnd'] = 'downsample.1'
    blk = BottleneckTrain(params, stride)
    nhwc = lambda a: a.permute(0, 2, 3, 1).contiguous().cuda()
    out = blk.forward(nhwc(t('x')))
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), G[f'{tag}_out'], atol=3e-5, rtol=1e-4)
    dx, grads = blk.backward(nhwc(t('dout')))
    tol = lambda ref: 1e-4 * float(np.abs(ref).max()) + 1e-7
    np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), G[f'{tag}_dx'], atol=tol(G[f'{tag}_dx']), rtol=1e-3)
    for k, ref_name in names.items():
        ref = G[f'{tag}_grad_{ref_name}']
        cout, cin, kh, kw = ref.shape
        got = grads[k].cpu().view(cout, kh, kw, cin).permute(0, 3, 1, 2).numpy()
        np.testing.assert_allclose(got, ref, atol=tol(ref), rtol=1e-3, err_msg=k)
    for k, ref_name in bns.items():
        for ours, theirs in (('gamma', 'weight'), ('beta', 'bias')):
            ref = G[f'{tag}_grad_{ref_name}.{theirs}']
            np.testing.assert_allclose(grads[f'{k}.{ours}'].cpu().numpy(), ref, atol=tol(ref), rtol=1e-3, err_msg=f'{k}.{ours}')
        for stat in ('running_mean', 'running_var'):
            np.testing.assert_allclose(params[k][stat].cpu().numpy(), G[f'{tag}_after_{ref_name}.{stat}'], atol=2e-6, rtol=2e-5, err_msg=f'{k}.{stat}')


@pytest.mark.parametrize('N,H,W,C,k,stride,pad', [(2, 16, 16, 8, 3, 2, 1), (1, 9, 11, 4, 3, 2, 1), (2, 8, 8, 16, 2, 2, 0)])
def test_maxpool_backward_matches_autograd(N, H, W, C, k, stride, pad):
    from vpho_amd import ops
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
    y = torch.nn.functional.max_pool2d(x, k, stride, pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    dx = ops.maxpool_bwd(nhwc(x), nhwc(dy), k, stride, pad).permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(dx.numpy(), x.grad.numpy(), atol=1e-6)
    # overlapping windows take the two-pass form (arg-max bytes, then position codes); the one-pass gather must give the same bits,
    # also with ties inside a window (first maximum in row-major order)
    xt = torch.randint(0, 3, (N, H, W, C), generator=g).float().cuda()
    dyt = nhwc(dy)
    one = torch.empty_like(xt)
    ops._call('vpho_maxpool_bwd_nhwc_f32', ops._f32(xt), ops._f32(dyt), ops.I(N), ops.I(H), ops.I(W), ops.I(C), ops.I(k), ops.I(stride), ops.I(pad), ops._f32(one))
    assert torch.equal(ops.maxpool_bwd(xt, dyt, k, stride, pad), one)


@pytest.mark.parametrize('N,H,W,C,OH,OW', [(2, 8, 8, 16, 16, 16), (1, 4, 6, 8, 8, 12), (2, 16, 16, 4, 8, 8), (1, 5, 7, 4, 13, 9)])
def test_bilinear_resize_backward_matches_autograd(N, H, W, C, OH, OW):
    from vpho_amd import ops
    g = torch.Generator().manual_seed(H * OW)
    x = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
    y = torch.nn.functional.interpolate(x, size=(OH, OW), mode='bilinear', align_corners=False)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = ops.resize_bilinear_bwd(dy.permute(0, 2, 3, 1).contiguous().cuda(), H, W).permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(dx.numpy(), x.grad.numpy(), atol=2e-6 * float(x.grad.abs().max()) + 1e-7, rtol=1e-5)


def test_backbone_training_step_matches_reference_fpn(sd):
    """Training-mode forward + backward of the whole two-branch ResNet-50 / FPN backbone vs the reference's own FPN module under
    autograd (fixture by tests/golden/make_golden_fpn_train.py): outputs, all 275 parameter gradients (norm + strided sample),
    running statistics (incl. the shared layer4, updated by both branch calls)."""
    import os
    from vpho_amd.train_blocks import FPNTrain
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_fpn_train.npz'))
    net = FPNTrain(sd, 'feature_extractor', 'cuda')
    ph, po = net.forward(torch.from_numpy(G['x']).cuda())
    for got, key in ((ph, 'p2_h'), (po, 'p2_o')):
        ref = G[key]
        np.testing.assert_allclose(got.permute(0, 3, 1, 2)[:, ::8].cpu().numpy(), ref, atol=2e-4 * float(np.abs(ref).max()), rtol=1e-3)
    nhwc = lambda a: torch.from_numpy(a).permute(0, 2, 3, 1).contiguous().cuda()
    grads = net.backward(nhwc(G['A']), nhwc(G['B']))
    names = [k[len('gnorm_'):] for k in G.files if k.startswith('gnorm_')]
    assert set(names) == set(grads), (sorted(set(names) ^ set(grads))[:10])
    worst = 0.0
    for k in names:
        g = grads[k].reshape(-1).cpu()
        nrm = float(G['gnorm_' + k])
        assert abs(float(g.double().norm()) - nrm) <= 2e-3 * nrm + 1e-9, (k, float(g.double().norm()), nrm)
        err = float(np.abs(g[::997].numpy() - G['gsample_' + k]).max()) / (nrm / max(1.0, g.numel() ** 0.5) + 1e-30)
        worst = max(worst, err)
        assert err < 0.05, (k, err)                       # sampled entries within 5 % of the RMS entry (deep net, fp32 both sides)
    for k in ('layer0_h.1', 'layer4_h.0.2.bn3', 'layer2_o.0.0.downsample.1'):
        pass
    st = {'layer0_h.1': net.stem['bn']}
    st['layer4_h.0.2.bn3'] = net.blocks['layer4_h'][2][1]['bn3']
    st['layer2_o.0.0.downsample.1'] = net.blocks['layer2_o'][0][1]['bnd']
    for k, b in st.items():
        np.testing.assert_allclose(b['running_mean'].cpu().numpy(), G['rm_' + k], atol=1e-5, rtol=1e-4, err_msg=k)
        np.testing.assert_allclose(b['running_var'].cpu().numpy(), G['rv_' + k], atol=1e-5, rtol=1e-4, err_msg=k)


def test_encoder_training_step_matches_reference_module(sd):
    """Training-mode forward + backward of the residual Encoder (8 pre-activation bottlenecks, 4 max-pools) vs the reference's
    own module under autograd, with gradients entering at the encoding and at the second stage map (the cross modules' input)."""
    import os
    from vpho_amd.train_blocks import EncoderTrain
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_encoder_train.npz'))
    net = EncoderTrain(sd, 'encoder_hand', 'cuda')
    g = np.random.default_rng(21)
    NB = 12
    x = torch.from_numpy((g.normal(size=(NB, net.cin, 32, 32)) * 0.3).astype(np.float32))
    xin = torch.zeros(NB, 32, 32, net.cin_pad)
    xin[..., :net.cin] = x.permute(0, 2, 3, 1)
    enc, stages = net.forward(xin.cuda())
    A = torch.from_numpy(g.normal(size=tuple(enc.shape)).astype(np.float32))
    B = torch.from_numpy(g.normal(size=(NB, stages[1].shape[3], stages[1].shape[1], stages[1].shape[2])).astype(np.float32))
    np.testing.assert_allclose(enc.cpu().numpy(), G['enc'], atol=3e-4 * float(np.abs(G['enc']).max()), rtol=1e-3)
    np.testing.assert_allclose(stages[1].permute(0, 3, 1, 2).reshape(-1)[::499].cpu().numpy(), G['stage1_sample'], atol=3e-4 * float(np.abs(G['stage1_sample']).max()), rtol=1e-3)
    dx, grads = net.backward(A.cuda(), B.permute(0, 2, 3, 1).contiguous().cuda())
    dxn = dx[..., :net.cin].permute(0, 3, 1, 2).reshape(-1).cpu()
    assert abs(float(dxn.double().norm()) - float(G['dx_norm'])) <= 2e-3 * float(G['dx_norm'])
    # ReLU kinks and pooling arg-maxes are not smooth: where a pre-activation or two pooled values sit within rounding of each
    # other the two implementations route the gradient differently, so single entries may differ -- the bulk must not
    rms = float(G['dx_norm']) / dxn.numel() ** 0.5
    err = np.abs(dxn[::499].numpy() - G['dx_sample'])
    assert float(np.median(err)) < 1e-3 * rms and float(np.mean(err > 0.05 * rms)) < 0.01, (float(np.median(err)) / rms, float(np.mean(err > 0.05 * rms)))
    names = [k[len('gnorm_'):] for k in G.files if k.startswith('gnorm_')]
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:10]
    top = max(float(G['gnorm_' + k]) for k in names)
    for k in names:
        gr = grads[k].reshape(-1).cpu()
        nrm = float(G['gnorm_' + k])
        if k.endswith(('conv1.bias', 'conv2.bias')):      # a bias in front of a BatchNorm has zero gradient: both sides hold rounding noise
            assert nrm < 1e-4 * top and float(gr.double().norm()) < 1e-4 * top, k
            continue
        assert abs(float(gr.double().norm()) - nrm) <= 3e-3 * nrm + 1e-9, (k, float(gr.double().norm()), nrm)
        e = np.abs(gr[::499].numpy() - G['gsample_' + k])
        rms_k = nrm / max(1.0, gr.numel() ** 0.5)
        assert float(e.max()) < 0.15 * rms_k + 1e-9 and (e.size < 16 or float(np.median(e)) < 0.02 * rms_k + 1e-9), (k, float(e.max()) / rms_k, float(np.median(e)) / rms_k)
    np.testing.assert_allclose(net.blocks[7][1]['bn2']['running_mean'].cpu().numpy(), G['rm_reg.7.bn2'], atol=1e-5, rtol=1e-4)


def test_roi_align_backward_matches_oracle_autograd():
    """RoIAlign backward (incl. the W-flip of left hands and accumulation of two RoIs into one gradient) vs autograd through the
    oracle's differentiable restatement of torchvision.ops.roi_align"""
    from oracle.roi_align import roi_align_fast
    from vpho_amd import ops
    g = torch.Generator().manual_seed(9)
    N, C, H, W, P = 3, 8, 20, 24, 8
    feat = torch.randn(N, C, H, W, generator=g).requires_grad_(True)
    boxes = torch.tensor([[8.0, 6.0, 70.0, 60.0], [0.0, 0.0, 95.9, 79.0], [30.0, 20.0, 34.0, 90.0]])
    boxes2 = boxes * 0.7 + 3.0
    rois = lambda b: torch.cat([torch.arange(N, dtype=torch.float32)[:, None], b], 1)
    flip = torch.tensor([True, False, True])
    y1 = roi_align_fast(feat, rois(boxes), P, 0.25)
    y2 = roi_align_fast(feat, rois(boxes2), P, 0.25)
    y2 = torch.where(flip[:, None, None, None], y2.flip(-1), y2)
    d1, d2 = torch.randn(y1.shape, generator=g), torch.randn(y2.shape, generator=g)
    ((y1 * d1).sum() + (y2 * d2).sum()).backward()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    df = ops.roi_align_bwd(nhwc(d1), boxes.cuda(), (H, W), C, 0.25)
    df = ops.roi_align_bwd(nhwc(d2), boxes2.cuda(), (H, W), C, 0.25, flip_w=flip.to(torch.uint8).cuda(), into=df)
    np.testing.assert_allclose(df.permute(0, 3, 1, 2).cpu().numpy(), feat.grad.numpy(), atol=2e-5 * float(feat.grad.abs().max()), rtol=1e-4)


def _check_sampled_grads(G, grads, skip_zero=()):
    names = [k[len('gnorm_'):] for k in G.files if k.startswith('gnorm_')]
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:10]
    top = max(float(G['gnorm_' + k]) for k in names)
    for k in names:
        gr = grads[k].reshape(-1).cpu()
        nrm = float(G['gnorm_' + k])
        if k in skip_zero:                                     # a bias in front of a BatchNorm: zero gradient, rounding noise on both sides
            assert nrm < 1e-4 * top and float(gr.double().norm()) < 1e-4 * top, k
            continue
        assert abs(float(gr.double().norm()) - nrm) <= 3e-3 * nrm + 1e-9, (k, float(gr.double().norm()), nrm)
        e = np.abs(gr[::499].numpy() - G['gsample_' + k])
        rms_k = nrm / max(1.0, gr.numel() ** 0.5)
        assert float(e.max()) < 0.15 * rms_k + 1e-9 and (e.size < 16 or float(np.median(e)) < 0.02 * rms_k + 1e-9), (k, float(e.max()) / rms_k)


def test_heatmap_head_training_step_matches_reference_module(sd):
    """Training-mode forward + backward of HeadHeatmap2 incl. the transposed convolution (four phase convolutions and their
    gradients reassembled into the (Cin, Cout, 4, 4) weight gradient) vs the reference's own module under autograd."""
    import os
    from vpho_amd.train_blocks import HeatmapHeadTrain
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_hmhead_train.npz'))
    net = HeatmapHeadTrain(sd, 'head_hm_hand', 'cuda')
    g = np.random.default_rng(33)
    x = torch.from_numpy((g.normal(size=(4, 256, 16, 16)) * 0.2).astype(np.float32))
    out = net.forward(x.permute(0, 2, 3, 1).contiguous().cuda())
    A = torch.from_numpy(g.normal(size=(4, out.shape[3], out.shape[1], out.shape[2])).astype(np.float32))
    o = out.permute(0, 3, 1, 2).reshape(-1).cpu()
    assert abs(float(o.double().norm()) - float(G['out_norm'])) < 1e-4 * float(G['out_norm'])
    np.testing.assert_allclose(o[::499].numpy(), G['out_sample'], atol=2e-4 * float(np.abs(G['out_sample']).max()), rtol=1e-3)
    dx, grads = net.backward(A.permute(0, 2, 3, 1).contiguous().cuda())
    dxn = dx.permute(0, 3, 1, 2).reshape(-1).cpu()
    assert abs(float(dxn.double().norm()) - float(G['dx_norm'])) <= 2e-3 * float(G['dx_norm'])
    rms = float(G['dx_norm']) / dxn.numel() ** 0.5
    err = np.abs(dxn[::499].numpy() - G['dx_sample'])
    assert float(np.median(err)) < 1e-3 * rms and float(np.mean(err > 0.05 * rms)) < 0.01
    _check_sampled_grads(G, grads, skip_zero=('conv_layers.1.bias',))


def test_align_heatmap_backward_matches_oracle_autograd():
    """align_hm_to_bbox_rectangle (transposing bilinear resample, zero padding) + W-flip backward vs autograd on the oracle"""
    from oracle.vpho import align_hm_to_bbox_rectangle, flip_w
    from vpho_amd import ops
    g = torch.Generator().manual_seed(4)
    N, C, S = 3, 5, 16
    hm = torch.randn(N, C, S, S, generator=g).requires_grad_(True)
    bbox = torch.tensor([[40.0, 50.0, 180.0, 200.0], [10.0, 20.0, 250.0, 190.0], [90.0, 60.0, 160.0, 210.0]])
    c = (bbox[:, :2] + bbox[:, 2:]) / 2
    m = (bbox[:, 2:] - bbox[:, :2]).max(-1, keepdim=True).values
    rect = torch.cat([c - m / 2, c + m / 2], -1)
    flip = torch.tensor([True, False, True])
    y = flip_w(align_hm_to_bbox_rectangle(hm, bbox, rect, S), flip)
    dy = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward()
    d = ops.align_heatmap_bwd(dy.permute(0, 2, 3, 1).contiguous().cuda(), bbox.cuda(), rect.cuda(), flip.to(torch.uint8).cuda())
    np.testing.assert_allclose(d.permute(0, 3, 1, 2).cpu().numpy(), hm.grad.numpy(), atol=2e-5 * float(hm.grad.abs().max()), rtol=1e-4)


@pytest.mark.parametrize('N,H,W,cin,cout', [(4, 16, 16, 64, 128), (3, 8, 12, 128, 64)])
def test_training_winograd_forward_and_input_gradient(N, H, W, cin, cout):
    """Training path of the 3x3 / stride-1 convolutions (weights change every step): u = G g G^T made on the DEVICE
    (vpho_winograd_weights_f32) equals the host transform of pack.winograd_weights, for the forward convolution and -- on the flipped,
    channel-transposed weights -- for its input gradient; the Winograd input gradient with the fused LeakyReLU backward (gate) equals
    torch autograd of conv2d(leaky_relu(x)) to fp32 rounding."""
    import torch.nn.functional as F
    from vpho_amd import ops, conv_backward as CB
    from vpho_amd.model.pack import pack_conv, winograd_weights
    g = torch.Generator().manual_seed(N * H + cin)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5)
    wp = pack_conv(w).cuda()
    u_dev, u_host = ops.winograd_weights_device(wp), winograd_weights(wp)
    assert u_dev.shape == u_host.shape and float((u_dev - u_host).abs().max()) <= 1e-7 * float(u_host.abs().max())
    wt = CB._flip_transpose(wp, cout, cin, 3, 3).contiguous()                                  # (Cin, 9*Cout): the input-gradient convolution's weights
    ut_dev, ut_host = ops.winograd_weights_device(wp, for_input_gradient=True), winograd_weights(wt)
    assert ut_dev.shape == ut_host.shape and float((ut_dev - ut_host).abs().max()) <= 1e-7 * float(ut_host.abs().max())
    # in-place weight update -> the cached transform follows the tensor's version
    before = u_dev.clone()                                                                     # the transform is updated in place
    wp.mul_(2.0)
    assert torch.allclose(ops.winograd_weights_device(wp), 2 * before)
    wp.mul_(0.5)
    # forward + input gradient through LeakyReLU against autograd (fp64 reference)
    x = torch.randn(N, cin, H, W, generator=g).double().requires_grad_(True)
    a = F.leaky_relu(x, 0.01)
    y = F.conv2d(a, w.double(), None, 1, 1)
    dy = torch.randn(N, cout, H, W, generator=g).double()
    dx, = torch.autograd.grad(y, x, dy)
    a_g = a.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
    y_g = ops.conv3x3_train(a_g, wp)
    assert float((y_g.permute(0, 3, 1, 2).cpu().double() - y.detach()).abs().max()) < 2e-6 * float(y.detach().abs().max())
    dy_g = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dx_g = CB.conv2d_dgrad(dy_g, wp, (H, W), 3, 3, 1, 1, gate=(a_g, 0.01))
    assert float((dx_g.permute(0, 3, 1, 2).cpu().double() - dx).abs().max()) < 2e-6 * float(dx.abs().max())
    da_g = CB.conv2d_dgrad(dy_g, wp, (H, W), 3, 3, 1, 1)                                       # no gate: gradient w.r.t. the convolution's input
    a2 = a.detach().requires_grad_(True)
    da, = torch.autograd.grad(F.conv2d(a2, w.double(), None, 1, 1), a2, dy)
    assert float((da_g.permute(0, 3, 1, 2).cpu().double() - da).abs().max()) < 2e-6 * float(da.abs().max())


def test_weight_gradient_over_roi_window_groups():
    """vpho_conv2d_wgrad_groups_nhwc_f32: a gradient that came back through RoIAlign is zero outside the RoI windows; reducing over the
    live 32-pixel groups only (vpho_window_groups_i32) gives the full reduction's weight gradient (to fp32 summation order), and the
    group list is exactly the groups that touch a window, ascending."""
    from vpho_amd import ops, conv_backward as CB
    g = torch.Generator().manual_seed(5)
    N, H, W, cin, cout = 6, 64, 64, 64, 128
    boxes = torch.tensor([[20.0, 30.0, 200.0, 180.0], [0.0, 0.0, 256.0, 256.0], [100.0, 10.0, 130.0, 250.0], [-20.0, 40.0, 90.0, 300.0],
                          [200.0, 200.0, 255.0, 255.0], [64.0, 64.0, 192.0, 96.0]]).cuda()
    win = ops.roi_windows(boxes, None, N, H, W, 0.25)
    lst, cnt = ops.window_groups(win)
    _, mask = win.to_map(torch.zeros(N * H * W, 1).cuda())
    live = mask.reshape(-1, 32).any(1).nonzero().flatten().int()
    n = int(cnt)
    assert n == live.numel() and torch.equal(lst[:n], live) and 0 < n < N * H * W // 32
    x = torch.randn(N, H, W, cin, generator=g).cuda()
    dy = torch.randn(N, H, W, cout, generator=g).cuda() * mask[..., None]
    # input gradient of a 3x3 convolution on the windows dilated by the halo only: equals the full map's, zeros included
    w3 = (torch.randn(cout, 9 * cin, generator=g) * 0.05).cuda()
    halo = ops.roi_windows(boxes, None, N, H, W, 0.25, dilate=1)
    dfull, dwin = CB.conv2d_dgrad(dy, w3, (H, W), 3, 3, 1, 1), CB.conv2d_dgrad(dy, w3, (H, W), 3, 3, 1, 1, rows=halo)
    _, hmask = halo.to_map(torch.zeros(N * H * W, 1).cuda())
    assert torch.equal(dwin[hmask], dfull[hmask]) and bool((dwin[~hmask] == 0).all()) and float(dfull[~hmask].abs().max()) == 0.0
    for k, pad in ((3, 1), (1, 0)):
        full = CB.conv2d_wgrad(x, dy, k, k, 1, pad)
        part = CB.conv2d_wgrad(x, dy, k, k, 1, pad, groups=(lst, cnt))
        assert float((full - part).abs().max()) < 2e-5 * float(full.abs().max()), k


def test_winograd_weight_batch_equals_the_single_transforms():
    """ops.WinogradWeightBatch.refresh(): one launch for the forward AND input-gradient transforms of several 3x3 weights == the
    per-tensor launches, bit for bit, after the weights changed in place without torch noticing (a kernel wrote through raw pointers)."""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(12)
    ws = [torch.randn(co, 9 * ci, generator=g).cuda() for co, ci in ((64, 64), (128, 32), (32, 128), (256, 16))]
    batch = ops.WinogradWeightBatch()
    with batch:
        us = [(ops.winograd_weights_device(w, False), ops.winograd_weights_device(w, True)) for w in ws]
    assert len(batch.items) == 8
    fresh = [torch.randn(w.shape, generator=g).cuda() for w in ws]
    for w, f in zip(ws, fresh):
        w.copy_(f)                                                  # new values in place
    want = [(ops.winograd_weights_device(f, False).clone(), ops.winograd_weights_device(f, True).clone()) for f in fresh]
    for (uf, ub), (wf, wb) in zip(us, want):
        ub.zero_(); uf.zero_()
    batch.refresh()
    torch.cuda.synchronize()
    for w, (uf, ub), (wf, wb) in zip(ws, us, want):
        assert torch.equal(uf, wf) and torch.equal(ub, wb)
        assert ops.winograd_weights_device(w, True) is ub           # marked current: no further launch
