"""The RoI-window Winograd launch of the predict step (FPN smoothing convolution 256 -> 256 on the 64 x 64 maps of 64 images, only the tiles that touch
an image's RoI window): time per launch.  VPHO_WINO_STAGED=0 / 1 (read per call) switches the input staging of the blocks that allow it."""
import os, sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd import ops
from vpho_amd.assets import synthetic_assets
from vpho_amd.synth import synth_batch
from vpho_amd.model.pack import winograd_weights
dev = 'cuda'
b = synth_batch(64, synthetic_assets(0), seed=206, rank=0)
bh, bhr, bo = (b[k].float().contiguous().to(dev) for k in ('bbox_hand', 'bbox_hand_rect', 'bbox_obj_rect'))
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 64, 64, 256, generator=g).to(dev)
w = (torch.randn(256, 9 * 256, generator=g) * (2.0 / (9 * 256)) ** 0.5).to(dev)
bias = torch.randn(256, generator=g).to(dev)
u = winograd_weights(w)
def timeit(f, n=20, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / n)
    return best
for name, win in (('hand', ops.roi_windows(bh, bhr, 64, 64, 64, 0.25)), ('obj', ops.roi_windows(bo, None, 64, 64, 64, 0.25))):
    res = {}
    for st in ('0', '1', '0', '1'):
        os.environ['VPHO_WINO_STAGED'] = st
        y = ops.conv3x3_winograd(x, u, bias, 0.01, rows=win)
        res.setdefault(st, []).append(timeit(lambda: ops.conv3x3_winograd(x, u, bias, 0.01, rows=win)))
        res.setdefault('y' + st, y)
    os.environ.pop('VPHO_WINO_STAGED')
    print(f"{name} windows ({int(win.count) / (64 * 4096):.3f} of the pixels): registers {min(res['0']) * 1e6:.1f} us, staged {min(res['1']) * 1e6:.1f} us; bit-identical {torch.equal(res['y0'], res['y1'])}", flush=True)
