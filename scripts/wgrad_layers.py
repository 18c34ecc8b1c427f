#!/usr/bin/env python
"""Weight-gradient launches of one training step, by shape: the step is run once with conv_backward.conv2d_wgrad recording its argument
shapes, then every distinct shape is timed stand-alone (kernel + slice reduction, HIP events).  Sorted by time per step.
VPHO_WGRAD_TILE / VPHO_WGRAD_WANT (conv_wgrad.hip's tuning aids) apply.      python scripts/wgrad_layers.py > gpurun_out/wgrad_layers.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['VPHO_WGRAD_STREAM'] = '0'


def main():
    import torch
    from vpho_amd import conv_backward as CB
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict, synth_batch
    from vpho_amd.train_step import DiffusionTrainStep
    from vpho_amd.trainer import synthetic_mano_targets
    dev = torch.device('cuda', 0)
    torch.manual_seed(206)
    assets = synthetic_assets(0)
    step = DiffusionTrainStep(synth_state_dict(vpho_net(assets), seed=1), dev, assets=assets)
    bs = 64
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=11, rank=0).items()}
    g = torch.Generator().manual_seed(100)
    data['hm_hand'] = (torch.rand(bs, 21, 64, 64, generator=g) * 0.2).to(dev)
    data['hm_obj'] = (torch.rand(bs, 27, 64, 64, generator=g) * 0.2).to(dev)
    gt_h = (torch.randn(bs, 96, generator=g) * 0.5).to(dev) + torch.tensor([1., 0, 0, 0, 1, 0], device=dev).repeat(16)
    gt_o = (torch.randn(bs, 9, generator=g) * 0.5).to(dev)
    data.update(synthetic_mano_targets(step.mano_head.mano, gt_h, (torch.randn(bs, 10, generator=g) * 0.5).to(dev), data['is_right']))
    data['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(dev)
    seen = collections.Counter()
    grouped_calls = []
    orig = CB.conv2d_wgrad

    def rec(x, dy, kh, kw, stride=1, pad=0, cin=None, pad_y=None, pad_x=None, groups=None):
        seen[(tuple(x.shape), tuple(dy.shape), kh, kw, stride, pad, cin, pad_y, pad_x, groups is not None)] += 1
        if groups is not None:
            grouped_calls.append((x, dy, kh, kw, stride, pad, groups))
        return orig(x, dy, kh, kw, stride, pad, cin, pad_y, pad_x, groups)
    CB.conv2d_wgrad = rec
    step.step(data, gt_h, gt_o, repeat_num=20)
    torch.cuda.synchronize()
    CB.conv2d_wgrad = orig
    rows = []
    for (xs, ys, kh, kw, stride, pad, cin, pad_y, pad_x, grouped), n in seen.items():
        if grouped:
            continue                                        # the window launches reduce over device-side lists: not reproducible from shapes
        x, dy = torch.randn(xs, device=dev), torch.randn(ys, device=dev)
        f = lambda: orig(x, dy, kh, kw, stride, pad, cin, pad_y, pad_x)
        f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        c_in = xs[-1] if cin is None else cin
        fl = 2.0 * ys[0] * ys[1] * ys[2] * ys[3] * c_in * kh * kw
        rows.append((n * ms, n, ms, fl / ms / 1e9, xs, ys[-1], kh, stride))
    for x, dy, kh, kw, stride, pad, groups in grouped_calls:       # the window launches: the step's own tensors and pixel-group lists
        live = int(groups[1].item()) * 32 / (dy.shape[0] * dy.shape[1] * dy.shape[2])
        for tag, gr in (('window list', groups), ('dense', None)):
            f = lambda: orig(x, dy, kh, kw, stride, pad, groups=gr)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2.0 * dy.numel() * x.shape[-1] * kh * kw
            print(f'# window launch {tuple(x.shape)} -> {dy.shape[-1]}, k{kh}: {tag}: {ms:.3f} ms; live pixel groups {live:.3f} of the map; '
                  f'{fl * (live if gr is not None else 1.0) / ms / 1e9:.1f} TF/s on the pixels it reduces')
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f'# weight gradients of one 64-image training step by shape (window launches left out): {tot:.2f} ms per step, {sum(r[1] for r in rows)} launches')
    print(f'{"ms/step":>8s} {"n":>3s} {"ms":>7s} {"TF/s":>6s}  x shape -> Cout, k, stride')
    for t, n, ms, tf, xs, co, k, st in rows:
        print(f'{t:8.3f} {n:3d} {ms:7.3f} {tf:6.1f}  {xs} -> {co}, k{k}, s{st}')


if __name__ == '__main__':
    main()
