"""CPU-only study: how close are adjacent ranks in the hand cascade / physics selections at the README config, and which
kinds of candidates (diffusion vs regression copy) sit at near-ties?  Uses the oracle only (test infrastructure)."""
import sys, time
import torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from oracle import vpho as OV

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = 100
a = synthetic_assets(0)
import os
from vpho_amd.synth import HM_GAIN_CONTRAST
m = vpho_net(a); sd = synth_state_dict(m, 1, hm_gain=float(os.environ.get("HM_GAIN", HM_GAIN_CONTRAST)))
data = synth_batch(n, a, seed=777)
torch.manual_seed(99)
nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
t0 = time.time()
ref, info = OV.predict(sd, a, ANCHOR_SKELETON, data, sample_num=S, sample_T0=0.65, sampling_steps=50, topk_hand=30, topk_obj=10, noise_hand=nh, noise_obj=no)
print('oracle', time.time() - t0, 's')
torch.save(dict(ref=ref, info=info, data=data), '/tmp/tie_oracle.pt')
h = info['agg']['hand']
for lvl in range(4):
    v, i = h['val'][lvl], h['topk'][lvl]
    # v: (bs,k) or (bs,k,5)
    gap = (v[:, :-1] - v[:, 1:]).abs() / v[:, :-1].abs().clamp_min(1e-30)
    isreg = i >= S
    print(f'level {lvl}: val range [{float(v.min()):.4g},{float(v.max()):.4g}] reg-copies in topk: {int(isreg.sum())}/{isreg.numel()}',
          f'gaps<1e-6: {int((gap < 1e-6).sum())}, of which exact 0: {int((gap == 0).sum())}; <1e-5: {int((gap<1e-5).sum())}, <1e-4: {int((gap<1e-4).sum())}; total {gap.numel()}')
    nz = gap[(gap < 1e-5)]
    print('    small gaps:', sorted(nz.flatten().tolist())[:20])
hp = info['agg']['hand_phys']
sc = hp['score']  # (bs,5,31)
v, _ = torch.sort(sc, dim=-1, descending=True)
gap = (v[..., :-1] - v[..., 1:]).abs() / v[..., :-1].abs().clamp_min(1e-30)
print('hand_phys: scores', tuple(sc.shape), 'gaps among top-6 <1e-6:', int((gap[..., :6] < 1e-6).sum()), 'exact 0:', int((gap[..., :6] == 0).sum()), '<1e-4', int((gap[..., :6] < 1e-4).sum()))
print('rank5/6 gap per (img,finger):', gap[..., 4].tolist())
