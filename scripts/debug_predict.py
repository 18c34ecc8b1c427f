"""Stage-by-stage comparison of the HIP path with the oracle at cfg1 sizes (run on the GPU box)."""
import sys
import numpy as np
import torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from oracle import vpho as OV

bs, S, steps, kh, ko, T0 = 2, 4, 5, 8, 3, 0.2
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, steps, kh, ko, T0
a = synthetic_assets(0)
m = vpho_net(a)
sd = synth_state_dict(m, 1)
m.load_state_dict(sd)
data = synth_batch(bs, a, seed=206)
torch.manual_seed(7)
nh, no = torch.randn(bs * S, 96), torch.randn(bs * S, 9)
ref, rinfo = OV.predict(sd, a, ANCHOR_SKELETON, data, sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=kh, topk_obj=ko,
                        noise_hand=nh, noise_obj=no)
m = m.cuda().eval()
gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
from vpho_amd.model.engine import Engine
eng = Engine(m)
out = eng.predict(gdata, noise_hand=nh, noise_obj=no)
torch.cuda.synchronize()
f, rf = eng.last_info['features'], rinfo['features']
def rep(name, got, want):
    got, want = got.detach().double().cpu(), want.detach().double()
    print(f'{name:28s} shape {tuple(got.shape)} max|d| {float((got-want).abs().max()):.3e}  ref max {float(want.abs().max()):.3e}')
nchw = lambda t: t.permute(0, 3, 1, 2)
rep('hand_feat', nchw(f['hand_feat']), rf['hand_feat']); rep('obj_feat', nchw(f['obj_feat']), rf['obj_feat'])
rep('hf_hr', nchw(f['hf_hr']), rf['hf_hr'])
rep('enc_in_hand[:256]', nchw(f['enc_in_hand'])[:, :256], rf['hf_hr_rect'])
rep('enc_in_obj[:256]', nchw(f['enc_in_obj'])[:, :256], rf['of_or_rect'])
for k in ('hand_heatmap', 'obj_heatmap', 'encoding_hand', 'encoding_obj', 'mano_pose', 'mano_shape', 'reg_hand_vert', 'reg_hand_joint', 'force_local'):
    rep(k, f[k], rf[k])
rep('stage_hand', nchw(f['stage_hand']), rf['stage_hand'])
rep('tok_hand', f['tok_hand'].view(bs, 65, 512)[:, :32], rf['tok_hand'])
rep('tok_obj', f['tok_obj'].view(bs, 65, 512)[:, 32:64], rf['tok_obj'])
print('nfev', eng.last_info['hand_ode']['nfev'], rinfo['hand_ode']['nfev'], eng.last_info['obj_ode']['nfev'], rinfo['obj_ode']['nfev'])
for k in ref:
    rep(k, out[k], ref[k])
ga, ra = eng.last_info['agg'], rinfo['agg']
for l in range(4):
    g = ga['hand_topk'][l].cpu().permute(0, 2, 1).squeeze(-1) if l else ga['hand_topk'][l].cpu()[:, 0]
    print('hand topk level', l, 'equal' if np.array_equal(g.numpy(), ra['hand']['topk'][l].numpy()) else ('DIFF', g.numpy().tolist(), ra['hand']['topk'][l].numpy().tolist()))
for k, rk in (('transl_topk', 'transl_topk'), ('rot_topk', 'rot_topk'), ('phys_topk', 'phys_topk'), ('heat_topk', 'heat_topk')):
    g = ga[k].cpu().view(bs, -1).numpy()
    print(k, 'equal' if np.array_equal(g, ra[rk].numpy()) else ('DIFF', g.tolist(), ra[rk].numpy().tolist()))
g = ga['hand_phys_topk'].cpu().numpy()
print('hand_phys_topk', 'equal' if np.array_equal(g, ra['hand_phys']['topk'].numpy()) else ('DIFF', g.tolist(), ra['hand_phys']['topk'].numpy().tolist()))
rep('phys_score', ga['phys_score'], ra['phys_score'])
rep('cascade_pose', ga['cascade_pose'], ra['cascade_mano'][:, :48])
rep('force_point', ga['force_point'], ra['force_point']); rep('force_global', ga['force_global'], ra['force_global'])
