"""Opt-in split-bf16 score head (VPHO_SCORE_MFMA=bf16x6 | bf16x9; csrc/score_ode.hip::head_tile_split) against the default fp32-MFMA
kernel and an fp64 evaluation of the oracle's denoiser: the error study the switch rests on.  Every fp32 operand is split into three
bf16 pieces exactly; bf16x9 multiplies all nine pairs (nothing dropped), bf16x6 drops the three smallest.  Bars: the split kernels'
error against fp64 may not exceed the fp32-MFMA kernel's own by more than 25 % (x9) / 50 % (x6), on the README row count (6 400 rows:
ordinary and tail tiles) at three noise levels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_split_bf16_head_is_as_accurate_as_the_fp32_mfma_head(sd, name, D, capsys):
    from oracle import nets as N
    from vpho_amd import ops
    net = ops.ScoreNet(sd, f'denoiser_{name}', 'cuda')
    sd64 = {k: v.double() for k, v in sd.items() if k.startswith(f'denoiser_{name}.')}
    bs, S = 64, 100
    g = torch.Generator().manual_seed(77)
    feat = torch.randn(bs, 1024, generator=g) * 0.3
    pick = torch.cat([torch.arange(0, 128), torch.arange(3000, 3128), torch.arange(6144 - 64, 6400)])     # ordinary tiles + every tail tile
    report = []
    for t in (0.65, 0.3, 0.01):
        x = torch.randn(bs * S, D, generator=g) * float(N.ve_prior_sigma(t))
        ref = N.denoiser(sd64, f'denoiser_{name}', feat.double()[pick // S], x.double()[pick], torch.full((len(pick), 1), t, dtype=torch.float64))
        scale = ref.abs().max().item()
        out = {}
        for mode in ('f32', 'bf16x9', 'bf16x6'):
            net.set_split(mode)
            out[mode] = net.score(feat.cuda(), x.cuda(), t, S).cpu()
            assert torch.isfinite(out[mode]).all()
        err = {m: ((out[m][pick].double() - ref).abs().max().item() / scale, (out[m][pick].double() - ref).pow(2).mean().sqrt().item() / scale) for m in out}
        dev = {m: (out[m] - out['f32']).abs().max().item() / scale for m in ('bf16x9', 'bf16x6')}
        report.append((t, err, dev))
        assert err['bf16x9'][0] <= 1.25 * err['f32'][0] + 1e-9 and err['bf16x9'][1] <= 1.25 * err['f32'][1] + 1e-10, (t, err)
        assert err['bf16x6'][0] <= 1.5 * err['f32'][0] + 1e-9 and err['bf16x6'][1] <= 1.5 * err['f32'][1] + 1e-10, (t, err)
        assert max(dev.values()) <= 2e-5                       # the same bar the fp32 kernel is held to against the fp32 oracle
    net.set_split('f32')
    with capsys.disabled():
        for t, err, dev in report:
            print(f'\n[split study] {name} t={t}: max/rms error vs fp64 (relative to max|score|): ' +
                  ', '.join(f'{m} {e[0]:.2e}/{e[1]:.2e}' for m, e in err.items()) +
                  f'; max deviation from the fp32 kernel: x9 {dev["bf16x9"]:.2e}, x6 {dev["bf16x6"]:.2e}', end='')


def test_predict_with_split_products_agrees_with_the_default(model_cpu, assets):
    """vpho_net.forward(mode='predict') with split-bf16 score heads AND convolutions against the default engine on the same inputs and prior draws:
    same step sequence of both solves, outputs within the north-star 1e-3 (observed ~1e-5)."""
    import copy
    from vpho_amd.configs.args import cfg
    from vpho_amd.model.engine import Engine
    from vpho_amd.synth import synth_batch
    saved = (cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0)
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 16, 10, 8, 4, 0.65
    try:
        m = copy.deepcopy(model_cpu).cuda().eval()
        data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(5, assets, seed=3).items()}
        eng = Engine(m)
        g = torch.Generator().manual_seed(9)
        nh, no = torch.randn(5 * 16, 96, generator=g), torch.randn(5 * 16, 9, generator=g)
        ref = {k: v.clone() for k, v in eng.predict(data, nh, no).items() if torch.is_tensor(v)}
        info = {k: (eng.last_info[k]['nfev'], [s[3] for s in eng.last_info[k]['steps']]) for k in ('hand_ode', 'obj_ode')}
        for mode in ('bf16x6', 'bf16x9'):
            eng.score_hand.set_split(mode)
            eng.score_obj.set_split(mode)
            eng.conv_terms = int(mode[-1])                    # the convolutions of the feature path too
            out = eng.predict(data, nh, no)
            for k in ('hand_heatmap', 'obj_heatmap', 'reg_hand_joint', 'force_local'):
                d = (out[k].double() - ref[k].double()).abs().max().item()
                assert 0 < d < 1e-4 * max(1.0, ref[k].abs().max().item()), (mode, k, d)      # the split kernels did run, and agree
            for k in ('hand_ode', 'obj_ode'):
                assert (eng.last_info[k]['nfev'], [s[3] for s in eng.last_info[k]['steps']]) == info[k]
            for k in ('diff_final_hand_mano', 'diff_final_obj_6d', 'diff_final_hand_joint'):
                assert (out[k].double() - ref[k].double()).abs().max().item() < 1e-3, (mode, k)
    finally:
        cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = saved


CONV_CASES = [  # N, H, W, Cin, Cout, k, stride, pad  -> all three tile classes, padding, stride 2, ragged pixel counts, K = 16
    (16, 32, 32, 128, 128, 3, 1, 1), (8, 64, 64, 64, 256, 1, 1, 0), (64, 16, 16, 256, 256, 3, 1, 1), (4, 31, 29, 32, 64, 3, 2, 1),
    (2, 8, 8, 512, 128, 1, 1, 0), (3, 17, 17, 16, 48, 3, 1, 1), (64, 8, 8, 2048, 256, 1, 1, 0), (5, 9, 9, 48, 32, 1, 1, 0)]


@pytest.mark.parametrize('case', CONV_CASES)
def test_split_bf16_conv_is_as_accurate_as_the_fp32_mfma_conv(case, capsys):
    """ops.conv_split(6 | 9): the convolution with split-bf16 products against the fp32-MFMA kernel and an fp64 torch convolution, with
    bias, residual and LeakyReLU epilogue.  Bar: error against fp64 at most 1.25x (x9) / 1.5x (x6) the fp32 kernel's."""
    import torch.nn.functional as F
    from vpho_amd import ops
    from vpho_amd.model.pack import pack_conv
    N, H, W, Cin, Cout, k, st, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g)
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    res = torch.randn(N, Cout, OH, OW, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), st, pad) + res.double(), 0.01)
    xg, wg, bg, rg = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv(w).cuda(), b.cuda(), res.permute(0, 2, 3, 1).contiguous().cuda()
    out = {}
    for name, terms in (('f32', 0), ('bf16x9', 9), ('bf16x6', 6)):
        with ops.conv_split(terms):
            out[name] = ops.conv2d_nhwc(xg, wg, bg, kh=k, kw=k, stride=st, pad=pad, res=rg, out_slope=0.01).permute(0, 3, 1, 2).cpu().double()
    scale = ref.abs().max().item()
    err = {m: ((o - ref).abs().max().item() / scale, (o - ref).pow(2).mean().sqrt().item() / scale) for m, o in out.items()}
    assert err['bf16x9'][0] <= 1.25 * err['f32'][0] + 1e-9 and err['bf16x9'][1] <= 1.25 * err['f32'][1] + 1e-10, err
    assert err['bf16x6'][0] <= 1.5 * err['f32'][0] + 1e-9 and err['bf16x6'][1] <= 1.5 * err['f32'][1] + 1e-10, err
    with capsys.disabled():
        print(f'\n[split study] conv {case}: max/rms error vs fp64: ' + ', '.join(f'{m} {e[0]:.2e}/{e[1]:.2e}' for m, e in err.items()), end='')


def test_split_bf16_conv_on_a_pixel_list():
    """the RoI-window launch (row_map / row_count) through the split kernel: the listed pixels equal the full split convolution's"""
    from vpho_amd import ops
    g = torch.Generator().manual_seed(4)
    N, H, W = 6, 64, 64
    x = torch.randn(N, H, W, 64, generator=g).cuda()
    w = (torch.randn(128, 9 * 64, generator=g) * 0.05).cuda()
    b = torch.randn(128, generator=g).cuda()
    boxes = torch.tensor([[20.0, 30.0, 200.0, 180.0]] * N).cuda()
    win = ops.roi_windows(boxes, None, N, H, W, 0.25)
    with ops.conv_split(6):
        full = ops.conv2d_nhwc(x, w, b, kh=3, kw=3, pad=1)
        rows = ops.conv2d_nhwc(x, w, b, kh=3, kw=3, pad=1, rows=win)
    scat, mask = win.to_map(rows)
    assert 0 < int(mask.sum()) < N * H * W and torch.equal(scat[mask], full[mask])


def test_predict_with_winograd_agrees_with_the_default(model_cpu, assets):
    """the default plan (3x3 / stride-1 convolutions of the feature path in Winograd form) against VPHO_WINOGRAD=0 (the direct implicit
    GEMM everywhere): same step sequences of both solves, feature-path outputs to 1e-5 of their range, samples within the north-star 1e-3 (observed ~1e-5)."""
    import copy
    from vpho_amd.configs.args import cfg
    from vpho_amd.model.engine import Engine
    from vpho_amd.synth import synth_batch
    saved = (cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0)
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 16, 10, 8, 4, 0.65
    try:
        m = copy.deepcopy(model_cpu).cuda().eval()
        data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(5, assets, seed=3).items()}
        eng = Engine(m)
        eng.winograd = False                                 # direct kernels; set explicitly so the test does not depend on VPHO_WINOGRAD
        g = torch.Generator().manual_seed(9)
        nh, no = torch.randn(5 * 16, 96, generator=g), torch.randn(5 * 16, 9, generator=g)
        ref = {k: v.clone() for k, v in eng.predict(data, nh, no).items() if torch.is_tensor(v)}
        info = {k: (eng.last_info[k]['nfev'], [s[3] for s in eng.last_info[k]['steps']]) for k in ('hand_ode', 'obj_ode')}
        eng.winograd = True
        out = eng.predict(data, nh, no)
        for k in ('hand_ode', 'obj_ode'):
            assert (eng.last_info[k]['nfev'], [s[3] for s in eng.last_info[k]['steps']]) == info[k]
        for k in ('hand_heatmap', 'obj_heatmap', 'reg_hand_joint', 'force_local'):
            d = (out[k].double() - ref[k].double()).abs().max().item()
            assert 0 < d < 1e-5 * max(1.0, ref[k].abs().max().item()), (k, d)
        for k in ('diff_final_hand_mano', 'diff_final_obj_6d', 'diff_final_hand_joint'):
            assert (out[k].double() - ref[k].double()).abs().max().item() < 1e-3, k
    finally:
        cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = saved
