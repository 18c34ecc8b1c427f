import sys, copy, torch, numpy as np
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from tests._readme_fixture import R, CFG
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = (CFG[k] for k in ('sample_num', 'sampling_steps', 'topk_hand', 'topk_obj', 'sample_T0'))
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(2, a, seed=4242).items()}
m(data, mode='predict')
out = m._engine.predict(data, noise_hand=torch.from_numpy(R['noise_hand']), noise_obj=torch.from_numpy(R['noise_obj']))
d = m._engine.last_info['agg']
c = lambda t: t.detach().cpu()
print('diff_final_hand_mano max abs', float((c(out['diff_final_hand_mano']) - torch.from_numpy(R['diff_final_hand_mano'])).abs().max()))
for lvl in range(4):
    want, val = torch.from_numpy(R[f'hand_topk_l{lvl}']).long(), torch.from_numpy(R[f'hand_val_l{lvl}'])
    gi, gv = c(d['hand_topk'][lvl]).long(), c(d['hand_val'][lvl])
    if lvl: gi, gv = gi.reshape(2, 5, -1).transpose(1, 2), gv.reshape(2, 5, -1).transpose(1, 2)
    gi, gv = gi.reshape(want.shape), gv.reshape(want.shape)
    ne = (gi != want)
    print('level', lvl, 'mismatches', int(ne.sum()), 'val noise on agreeing', float((gv - val)[~ne].abs().max()), 'all', float((gv - val).abs().max()))
    for pos in ne.nonzero().tolist()[:8]:
        b, r = pos[0], pos[1]
        f = pos[2] if len(pos) > 2 else None
        sl = (b, slice(max(0, r - 1), r + 3)) + ((f,) if f is not None else ())
        print('   img', b, 'rank', r, 'finger', f, 'ref idx', want[sl].tolist(), 'ref val', [round(x, 6) for x in val[sl].tolist()], '| gpu idx', gi[sl].tolist(), 'gpu val', [round(x, 6) for x in gv[sl].tolist()])
dm = (c(out['agg_hand_mano']) - torch.from_numpy(R['agg_hand_mano'])).abs()
print('agg_hand_mano abs diff per image/joint (48 pose = 16x3):', [[round(float(x), 4) for x in dm[b, :48].reshape(16, 3).amax(1)] for b in range(2)])
print('agg_hand_joint max abs diff', float((c(out['agg_hand_joint']) - torch.from_numpy(R['agg_hand_joint'])).abs().max()))
print('gpu hand_phys_topk', c(d['hand_phys_topk']).tolist())
sc = c(d['hand_phys_score'])
for b in range(2):
    s0 = sc[b, :, 0]
    order = s0.argsort(descending=True)[:8]
    print('img', b, 'finger0 top scores', [(int(i), round(float(s0[i]), 6)) for i in order])
from oracle import vpho as OV
from vpho_amd.assets import ANCHOR_SKELETON
data_c = synth_batch(2, a, seed=4242)
sdc = {k: v.cpu() for k, v in m.state_dict().items()}
ref, info = OV.predict(sdc, a, ANCHOR_SKELETON, data_c, noise_hand=torch.from_numpy(R['noise_hand']), noise_obj=torch.from_numpy(R['noise_obj']), **CFG)
print('oracle hand_phys topk', info['agg']['hand_phys']['topk'].tolist())
print('oracle finger0 scores img0', [(int(i), round(float(info['agg']['hand_phys']['score'][0, 0, i]), 6)) for i in info['agg']['hand_phys']['score'][0, 0].argsort(descending=True)[:8]])
print('oracle vs ref agg_hand_mano', float((ref['agg_hand_mano'] - torch.from_numpy(R['agg_hand_mano'])).abs().max()))
print('--- real (non-copy) mismatches')
for lvl in range(4):
    want = torch.from_numpy(R[f'hand_topk_l{lvl}']).long(); val = torch.from_numpy(R[f'hand_val_l{lvl}'])
    gi, gv = c(d['hand_topk'][lvl]).long(), c(d['hand_val'][lvl])
    if lvl: gi, gv = gi.reshape(2, 5, -1).transpose(1, 2), gv.reshape(2, 5, -1).transpose(1, 2)
    gi, gv = gi.reshape(want.shape), gv.reshape(want.shape)
    ne = gi != want
    real = ne & ~((gi >= 100) & (want >= 100))
    for pos in real.nonzero().tolist()[:10]:
        tp = tuple(pos)
        print('  level', lvl, 'pos', pos, 'ref idx', int(want[tp]), 'val', float(val[tp]), 'gpu idx', int(gi[tp]), 'val', float(gv[tp]))
