#!/usr/bin/env python
"""Convolution launches of one predict step (README config, bs 64), by shape: the step is run once (plain launches, VPHO_GRAPHS=0) with
ops.conv2d_nhwc / ops.conv3x3_winograd recording their calls, then every distinct call is replayed stand-alone on the step's own
tensors (HIP events, 5 repeats).  Columns: time per step, launches, time per launch, TFLOP/s of the direct form, the launch's
algorithmic bytes (input + second input + residual + output, weights once) and the time those bytes take at 5 TB/s.
python scripts/conv_layers.py > gpurun_out/conv_layers.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['VPHO_GRAPHS'] = '0'


def main():
    import torch
    from vpho_amd import ops
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.configs.args import cfg
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import bench_state_dict, synth_batch
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
    assets = synthetic_assets(0)
    m = vpho_net(assets)
    m.load_state_dict(bench_state_dict(m))
    m = m.cuda().eval()
    data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, assets, seed=11).items()}
    m(data, mode='predict')
    torch.cuda.synchronize()
    calls = []
    o_conv, o_wino = ops.conv2d_nhwc, ops.conv3x3_winograd

    def rec_conv(x, w, bias=None, **kw):
        calls.append(('igemm', (x, w, bias), kw))
        return o_conv(x, w, bias, **kw)

    def rec_wino(x, u, bias=None, out_slope=1.0, **kw):
        kw = dict(kw, out_slope=out_slope)
        calls.append(('wino', (x, u, bias), kw))
        return o_wino(x, u, bias, **kw)
    ops.conv2d_nhwc, ops.conv3x3_winograd = rec_conv, rec_wino
    import vpho_amd.model.engine as E
    m._engine.predict(data)
    torch.cuda.synchronize()
    ops.conv2d_nhwc, ops.conv3x3_winograd = o_conv, o_wino

    def shape(t):
        return None if t is None else tuple(t.shape)
    groups = collections.OrderedDict()
    for kind, (x, w, b), kw in calls:
        key = (kind, shape(x), shape(w), tuple(sorted((k, shape(v) if torch.is_tensor(v) else (type(v).__name__ if not isinstance(v, (int, float, bool, tuple, type(None))) else v))
                                                         for k, v in kw.items() if k != 'out')))
        groups.setdefault(key, []).append((kind, (x, w, b), kw))
    rows = []
    for key, lst in groups.items():
        kind, (x, w, b), kw = lst[0]
        kw = {k: v for k, v in kw.items() if k != 'out'}
        f = (lambda: o_conv(x, w, b, **kw)) if kind == 'igemm' else (lambda: o_wino(x, w, b, **kw))
        y = f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        N, H, W, ld = x.shape
        if kind == 'igemm':
            cout, K = w.shape
            kh, kwd = kw.get('kh', 1), kw.get('kw', 1)
        else:
            cout, K, kh, kwd = w.shape[2], 9 * w.shape[0] * 8, 3, 3
        rows_obj = kw.get('rows')
        live = 1.0
        npix_out = y.numel() / cout
        if kw.get('out_view') is not None or kw.get('out_hw') is not None:
            # the launch writes INTO a larger view (the four phase convolutions of the transposed convolution each fill a quarter of the
            # 64 x 64 map): count the pixels the launch computes, not the pixels of the tensor it returns (370 "TFLOP/s")
            st_, pd_ = kw.get('stride', 1), kw.get('pad', 0)
            py_ = kw.get('pad_y') if kw.get('pad_y') is not None else pd_
            px_ = kw.get('pad_x') if kw.get('pad_x') is not None else pd_
            oh, ow = kw['out_hw'] if kw.get('out_hw') is not None else ((H + 2 * py_ - kh) // st_ + 1, (W + 2 * px_ - kwd) // st_ + 1)
            npix_out = N * oh * ow
        if rows_obj is not None:
            live = float(rows_obj.count.item()) / (rows_obj.shape[0] * rows_obj.shape[1] * rows_obj.shape[2])
            npix_out = rows_obj.shape[0] * rows_obj.shape[1] * rows_obj.shape[2] * live
        fl = 2.0 * npix_out * cout * K
        by = 4.0 * (x.numel() * live + npix_out * cout + w.numel() + sum(t.numel() for t in (kw.get('res'), kw.get('x2'), kw.get('res_up')) if torch.is_tensor(t)))
        extras = ','.join(k for k in ('res', 'x2', 'res_up', 'rows', 'gate', 'out_view') if kw.get(k) is not None)
        rows.append((len(lst) * ms, len(lst), ms, fl / ms / 1e9, by / 1e6, by / 5e9, kind, (N, H, W, ld), cout, kh, kw.get('stride', 1), extras, live))
    rows.sort(reverse=True)
    print(f'# convolution launches of one 64-image predict step by shape: {sum(r[0] for r in rows):.2f} ms stand-alone, {sum(r[1] for r in rows)} launches')
    print(f'{"ms/step":>8s} {"n":>3s} {"ms":>7s} {"TF/s":>6s} {"MB":>7s} {"ms@5TB/s":>8s}  kernel  x shape -> Cout, k, stride  [fused inputs] live')
    for t, n, ms, tf, mb, tb, kind, xs, co, k, st, ex, live in rows:
        print(f'{t:8.3f} {n:3d} {ms:7.3f} {tf:6.1f} {mb:7.1f} {tb:8.3f}  {kind:5s}  {xs} -> {co}, k{k}, s{st}  [{ex}] {live:.2f}')


if __name__ == '__main__':
    main()
