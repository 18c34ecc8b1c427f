"""What the BatchNorm reductions cost where: conv -> BatchNorm(train) forward and dgrad -> BatchNorm backward on the training step's
layer shapes (bs 64), each timed as (a) the stand-alone reduction pass, (b) the reductions in the convolution's epilogue.
    python scripts/bn_fuse_bench.py            (on the GPU box; one line per shape)
Per shape: conv alone | conv with the epilogue sums | BatchNorm with its own pass | BatchNorm from the partial sums | fused - unfused total."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vpho_amd import ops, conv_backward as CB


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3       # us


# (N, H, W, cin, cout, k): bottleneck conv1 / conv2 / conv3 of layers 1-4, encoder blocks
SHAPES = [(64, 64, 64, 64, 64, 1), (64, 64, 64, 64, 64, 3), (64, 64, 64, 64, 256, 1), (64, 64, 64, 256, 64, 1),
          (64, 32, 32, 512, 128, 1), (64, 32, 32, 128, 128, 3), (64, 32, 32, 128, 512, 1),
          (64, 16, 16, 1024, 256, 1), (64, 16, 16, 256, 256, 3), (64, 16, 16, 256, 1024, 1),
          (64, 8, 8, 2048, 512, 1), (64, 8, 8, 512, 512, 3), (64, 8, 8, 512, 2048, 1)]
g = torch.Generator().manual_seed(0)
print(f"{'shape':34s} {'conv':>8s} {'conv+sums':>10s} {'bn own':>8s} {'bn part':>8s} {'fwd diff':>9s} | {'dgrad':>8s} {'dgrad+sums':>11s} {'bnb own':>8s} {'bnb part':>9s} {'bwd diff':>9s}   (us)")
for N, H, W, cin, cout, k in SHAPES:
    x = torch.randn(N, H, W, cin, generator=g).cuda()
    w = (torch.randn(cout, k * k * cin, generator=g) * (1.0 / (k * k * cin)) ** 0.5).cuda()
    gamma, beta = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
    conv = (lambda bn=None: ops.conv2d_nhwc(x, w, bn=bn)) if k == 1 else (lambda bn=None: ops.conv3x3_train(x, w, bn=bn))
    f = ops.BnFuse()
    y = conv(f)
    t_conv = timed(lambda: conv())
    t_conv_s = timed(lambda: conv(ops.BnFuse()))
    t_bn = timed(lambda: ops.bn_train_forward(y, gamma, beta, slope=0.01))
    t_bn_p = timed(lambda: ops.bn_train_forward(y, gamma, beta, slope=0.01, partials=f))
    # backward: the input of this convolution is a = lrelu(bn(c)); dy arrives at the convolution's output
    c = torch.randn(N, H, W, cin, generator=g).cuda()
    gi, bi = torch.ones(cin).cuda(), torch.zeros(cin).cuda()
    a, saved = ops.bn_train_forward(c, gi, bi, slope=0.01)
    dy = torch.randn(N, H, W, cout, generator=g).cuda()
    pad = 1 if k == 3 else 0
    dg = lambda bn=None: CB.conv2d_dgrad(dy, w, (H, W), k, k, 1, pad, gate=(a, 0.01), bn=bn)
    fb = ops.BnFuse(c, saved, gi, bi)
    da = dg(fb)
    t_dg = timed(lambda: dg())
    t_dg_s = timed(lambda: dg(ops.BnFuse(c, saved, gi, bi)))
    t_bb = timed(lambda: ops.bn_train_backward(c, da, gi, saved))
    t_bb_p = timed(lambda: ops.bn_train_backward(c, da, gi, saved, partials=fb))
    print(f'{str((N, H, W, cin, cout, k)):34s} {t_conv:8.1f} {t_conv_s:10.1f} {t_bn:8.1f} {t_bn_p:8.1f} {t_conv_s + t_bn_p - t_conv - t_bn:9.1f} | '
          f'{t_dg:8.1f} {t_dg_s:11.1f} {t_bb:8.1f} {t_bb_p:9.1f} {t_dg_s + t_bb_p - t_dg - t_bb:9.1f}', flush=True)
