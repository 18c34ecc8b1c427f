"""Per-layer timing of every conv_igemm launch of one feature pass at bs=64 (events around each call)."""
import sys, time, collections
import torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.engine import Engine
from vpho_amd import ops

a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, a).items()}
eng = Engine(m)
eng.features(data); torch.cuda.synchronize()
orig = ops.conv2d_nhwc
rec = []
def timed(x, w, bias=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, w, bias, **kw); e1.record()
    N, H, W, ld = x.shape
    rows = kw.get('rows')                                     # RoI-window launches: the pixels really computed (synchronises; timing aid only)
    ys = y.shape if rows is None else (int(rows.count), 1, 1, y.shape[-1])
    rec.append((e0, e1, (N, H, W, kw.get('cin', ld), w.shape[0], kw.get('kh', 1), kw.get('stride', 1), 'win' if rows is not None else ''), ys))
    return y
ops.conv2d_nhwc = timed
orig_w = ops.conv3x3_winograd
def timed_w(x, u, bias=None, out_slope=1.0, out=None, rows=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_w(x, u, bias, out_slope, out, rows); e1.record()
    N, H, W, ld = x.shape
    cout, cin = u.shape[2], u.shape[0] * 8
    ys = (N, H, W, cout) if rows is None else (int(rows.count), 1, 1, cout)
    rec.append((e0, e1, (N, H, W, cin, cout, 3, 1, 'wino-win' if rows is not None else 'wino'), ys))
    return y
ops.conv3x3_winograd = timed_w
eng.features(data); torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, key, ys in rec:
    t = e0.elapsed_time(e1)
    M = ys[0] * ys[1] * ys[2]
    fl = 2.0 * M * key[4] * key[3] * key[5] * key[5]
    d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += t; d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f'total conv time {tot:.2f} ms in {len(rec)} launches')
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'N{key[0]} H{key[1]} Cin{key[3]} Cout{key[4]} k{key[5]} s{key[6]} {key[7]}: calls {n} time {t:.3f} ms ({100*t/tot:.1f}%)  {fl/t/1e9:.1f} TF/s')
