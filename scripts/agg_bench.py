"""Exclusive times of the HBM/latency-bound kernels of the aggregation at the README sizes (bs 64, S 100, top-k 30 / 10):
   python scripts/agg_bench.py            (torch events around 20 back-to-back launches each)"""
import sys, torch
sys.argv = ['x']; sys.path.insert(0, '.')
from vpho_amd import ops
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from vpho_amd.synth import synth_batch

dev = 'cuda'
assets = synthetic_assets(0)
bs, S, KH, KO = 64, 100, 30, 10
g = torch.Generator().manual_seed(0)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=1).items()}
mano = ops.Mano(assets['mano'], dev)
agg = ops.Aggregation(assets, ANCHOR_SKELETON, dev)


def timeit(name, f, n=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    print(f'{name:34s} {a.elapsed_time(b) / n * 1e3:9.1f} us')


betas = (torch.randn(bs, 10, generator=g) * 0.3).to(dev)
ctx = mano.shape(betas)
pose = (torch.randn(bs * 2 * S, 48, generator=g) * 0.3).to(dev)
timeit('mano_fk 12800 joints-only', lambda: mano.fk(pose, ctx, 2 * S, False))
p64 = pose[:bs * S].contiguous()
timeit('mano_fk 6400 with vertices', lambda: mano.fk(p64, ctx, S, True))
p31 = pose[:bs * 31].contiguous()
timeit('mano_fk 1984 with vertices', lambda: mano.fk(p31, ctx, 31, True))
p1 = pose[:bs].contiguous()
timeit('mano_fk 64 with vertices', lambda: mano.fk(p1, ctx, 1, True))
for level, n_obs in ((0, 20), (1, 15), (3, 5)):
    hv = torch.rand(bs, 2 * S, n_obs, generator=g).to(dev)
    pp = pose.view(bs, 2 * S, 48).clone()
    timeit(f'hand_fuse level {level}', lambda: agg.hand_fuse_level(hv, pp, KH, level, want_topk_pose=(level == 3)))
sc = torch.rand(bs, S, generator=g).to(dev)
timeit('topk 64 x 100 -> 10', lambda: agg.topk(sc, KO))
sc5 = torch.rand(bs, 31, 5, generator=g).to(dev)
timeit('topk 64 x 31 x 5 -> 5', lambda: agg.topk(sc5, 5, 5))
cand = torch.randn(bs, KO * KO, 9, generator=g).double().to(dev)
cand[..., 6:] *= 0.05
root = data['root_joint'].float().contiguous()
oid = agg.obj_ids(data['obj_name'])
isr = data['is_right'].to(torch.uint8).contiguous()
fp = (torch.randn(bs, 32, 3, generator=g) * 0.05).to(dev) + root[:, None]
fg = torch.randn(bs, 32, 3, generator=g).to(dev)
timeit('obj_physics 64 x 100', lambda: agg.obj_physics_score(cand, root, oid, isr, fp, fg))
K = data['cam_intr_crop_flip'].float().view(bs, 9).contiguous()
bb = data['bbox_obj_rect'].float().contiguous()
hm = torch.rand(bs, 27, 64, 64, generator=g).to(dev)
timeit('obj_heat 64 x 100', lambda: agg.obj_heat_score(cand, root, oid, isr, K, bb, hm))
idx = torch.stack([torch.randperm(KO * KO, generator=g)[:5] for _ in range(bs)]).int().to(dev)
timeit('obj_fuse 64 x 5', lambda: agg.obj_fuse(cand, idx, None, idx, torch.full((bs, 5), 0.2, device=dev), isr))
verts = torch.randn(bs * 31, 778, 3, generator=g).to(dev) * 0.05
fl = torch.randn(bs, 32, 3, generator=g).to(dev)
timeit('force_anchor 1984', lambda: agg.force_anchor(verts, data['root_joint_flip'].float().contiguous(), fl, 31))
fp2, fg2 = agg.force_anchor(verts, data['root_joint_flip'].float().contiguous(), fl, 31)
ov = torch.randn(bs, 2048, 3, generator=g).to(dev) * 0.05
timeit('hand_phys_score 64 x 31', lambda: agg.hand_phys_score(fp2, fg2, ov, bs, 31))
fs = agg.hand_phys_score(fp2, fg2, ov, bs, 31)
_, fidx = agg.topk(fs, 5, 5)
c58 = torch.randn(bs, 31, 58, generator=g).to(dev) * 0.3
timeit('hand_phys_fuse', lambda: agg.hand_phys_fuse(c58, fidx))
