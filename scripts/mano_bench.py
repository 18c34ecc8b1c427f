"""MANO FK launch timing: 6 400 hypotheses with vertices (100 per image), 12 800 joints-only.  VPHO_MANO_MFMA=0: the packed-FMA kernel."""
import sys, torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd import ops
from vpho_amd.assets import synthetic_assets
a = synthetic_assets(0)
M = ops.Mano(a['mano'], 'cuda')
g = torch.Generator().manual_seed(0)
ctx = M.shape((torch.randn(64, 10, generator=g) * 0.5).cuda())
for n, per, verts in ((6400, 100, True), (12800, 200, False), (64 * 30, 30, True)):
    pose = (torch.randn(n, 48, generator=g) * 0.4).cuda()
    for _ in range(3):
        M.fk(pose, ctx, per, verts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        M.fk(pose, ctx, per, verts)
    e1.record(); torch.cuda.synchronize()
    print(f'{n} hands, {per} per image, verts={verts}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
