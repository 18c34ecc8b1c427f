"""Whose MANO forward kinematics is closer to float64?  The joints-only HIP kernels (the heat-map cascade's scores are look-ups at their
projected joints) and the reference arithmetic (oracle/mano.py in torch fp32) against oracle/mano.py in float64, on the same poses.
python scripts/fk_error.py  -> rms / max joint position error in metres, by joint level"""
import os, sys
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vpho_amd.assets import synthetic_assets
from vpho_amd import ops
from oracle.mano import get_hand_verts
assets = synthetic_assets(0)
mano = ops.Mano(assets['mano'], 'cuda')
g = torch.Generator().manual_seed(0)
bs, S = 16, 200
pose = torch.randn(bs, S, 48, generator=g) * 0.35
betas = torch.randn(bs, 10, generator=g) * 0.5
ctx = mano.shape(betas.cuda())
_, j_hip = mano.fk(pose.view(-1, 48).cuda(), ctx, S, False)
v_hip, j_hip_v = mano.fk(pose.view(-1, 48).cuda(), ctx, S, True)
shape = betas[:, None].expand(bs, S, 10).reshape(-1, 10)
v32, j32 = get_hand_verts(assets['mano'], pose.view(-1, 48), shape)
v64, j64 = get_hand_verts(assets['mano'], pose.view(-1, 48).double(), shape.double())
LEVEL = {0: [0], 1: [1, 5, 9, 13, 17], 2: [2, 6, 10, 14, 18], 3: [3, 7, 11, 15, 19], 4: [4, 8, 12, 16, 20]}
for name, j in (('HIP joints-only', j_hip.cpu().double()), ('HIP with vertices', j_hip_v.cpu().double()), ('torch fp32 (reference arithmetic)', j32.double())):
    e = (j.view(-1, 21, 3) - j64.view(-1, 21, 3)).norm(dim=-1)
    print(f'{name:36s} ' + '  '.join(f'L{l}: rms {e[:, idx].pow(2).mean().sqrt():.2e} max {e[:, idx].max():.2e}' for l, idx in LEVEL.items()))
ev_h, ev_o = (v_hip.cpu().double().view(-1, 778, 3) - v64.view(-1, 778, 3)).norm(dim=-1), (v32.double().view(-1, 778, 3) - v64.view(-1, 778, 3)).norm(dim=-1)
print(f'vertices: HIP rms {ev_h.pow(2).mean().sqrt():.2e} max {ev_h.max():.2e};  torch fp32 rms {ev_o.pow(2).mean().sqrt():.2e} max {ev_o.max():.2e}')
# common-mode part: the error of the per-image mean over the 200 hands of an image (what a shared J / v_shaped error looks like)
for name, j in (('HIP joints-only', j_hip.cpu().double()), ('torch fp32', j32.double())):
    d = (j.view(bs, S, 21, 3) - j64.view(bs, S, 21, 3))
    cm = d.mean(1, keepdim=True)
    print(f'{name:18s} common-mode per image rms {cm.norm(dim=-1).pow(2).mean().sqrt():.2e}; candidate-specific rms {(d - cm).norm(dim=-1).pow(2).mean().sqrt():.2e}')
