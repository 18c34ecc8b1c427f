"""Exclusive timing of score_head_kernel (HIP events around each launch) for the hand and object score networks."""
import sys, torch
import os; sys.argv = ['x']; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict
from vpho_amd.assets import synthetic_assets
from vpho_amd import ops
a = synthetic_assets(0); m = vpho_net(a); sd = synth_state_dict(m, 1)
dev = 'cuda'
key = [k for k in ops.PROF_CLASSES if 'head' in k][0]
import os
for name, D in (('hand', 96), ('obj', 9)):
    net = ops.ScoreNet(sd, f'denoiser_{name}', dev)       # VPHO_SCORE_MFMA=bf16x6|bf16x9 selects the split-bf16 head
    bs, S = 64, 100
    feat = torch.randn(bs, 1024, device=dev) * 0.3
    x = torch.randn(bs * S, D, device=dev)
    for _ in range(3): net.score(feat, x, 0.3, S)
    torch.cuda.synchronize()
    ops.prof_enable(key, True)
    for _ in range(30): net.score(feat, x, 0.3, S)
    torch.cuda.synchronize()
    r = ops.prof_collect(key)
    ops.prof_enable(key, False)
    print(f"head {name}: {r['total_ms'] / r['launches'] * 1e3:.1f} us/launch  {r['flops'] / r['total_ms'] / 1e9:.1f} TF/s  ({r['launches']} launches)")
