"""Stage-by-stage comparison of the HIP aggregation with the oracle's, both fed the HIP path's own candidates (README config)."""
import sys
import torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from oracle.aggregation import hoi_aggregate

cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
n = 2
data = synth_batch(n, a, seed=777)
torch.manual_seed(99)
nh, no = torch.randn(n * 100, 96), torch.randn(n * 100, 9)
gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
out = m._engine.predict(gdata, noise_hand=nh, noise_obj=no) if hasattr(m, '_engine') and m._engine else None
if out is None:
    m(gdata, mode='predict'); out = m._engine.predict(gdata, noise_hand=nh, noise_obj=no)
torch.cuda.synchronize()
info = m._engine.last_info
gf, gd = info['features'], info['agg']
c = lambda t: t.detach().cpu()
fl = c(out['diff_final_hand_mano']).reshape(-1, 58)
same = hoi_aggregate(a, ANCHOR_SKELETON, cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                     root_joint=data['root_joint'], is_right=data['is_right'], force_local=c(gf['force_local']),
                     is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(), hand_pose_regression=c(gf['mano_pose']),
                     hand_shape=fl[:, 48:], hand_heatmap=c(gf['hand_heatmap']), hand_bbox=data['bbox_hand'], hand_topk=30,
                     obj_pose6d=c(out['diff_final_obj_6d']), obj_heatmap=c(gf['obj_heatmap']), obj_bbox=data['bbox_obj_rect'],
                     obj_topk=10, obj_name=data['obj_name'])
od = same['dbg']
print('is_right', data['is_right'], 'is_grasped', data['is_grasped'])
for lvl in range(4):
    gi, oi = c(gd['hand_topk'][lvl]).long(), od['hand']['topk'][lvl].long()
    gv, ov = c(gd['hand_val'][lvl]), od['hand']['val'][lvl]
    print(f'level {lvl}: idx equal {bool((gi.reshape(oi.shape) == oi).all())}  n_diff {(gi.reshape(oi.shape) != oi).sum().item()}  val max abs {float((gv.reshape(ov.shape) - ov).abs().max()):.3e}')
print('cascade pose', float((c(gd['cascade_pose']) - od['cascade_mano'][:, :48]).abs().max()))
print('force_point', float((c(gd['force_point']) - od['force_point']).abs().max()), 'force_global', float((c(gd['force_global']) - od['force_global']).abs().max()))
for k, ok in (('transl_topk', 'transl_topk'), ('rot_topk', 'rot_topk'), ('phys_topk', 'phys_topk'), ('heat_topk', 'heat_topk')):
    gi, oi = c(gd[k]).long(), od[ok].long()
    print(k, 'equal', bool((gi.reshape(oi.shape) == oi).all()), gi.reshape(oi.shape).tolist() if gi.numel() < 30 else '', oi.tolist() if oi.numel() < 30 else '')
print('phys_score max abs', float((c(gd['phys_score']).reshape(od['phys_score'].shape) - od['phys_score']).abs().max()))
print('obj_vert', float((c(gd['obj_vert']) - same['agg_obj_vert']).abs().max()))
hp = od['hand_phys']
print('hand_phys keys', list(hp.keys()))
gi = c(gd['hand_phys_topk']).long()
for k in hp:
    v = hp[k]
    if torch.is_tensor(v): print('  ', k, tuple(v.shape))
print('gpu hand_phys_topk', tuple(gi.shape), gi.tolist())
for k in ('topk', 'idx', 'index'):
    if k in hp: print('oracle', k, hp[k].tolist())
print('cand58', float((c(gd['cand58']).reshape(-1) - hp['cand'].reshape(-1)).abs().max()) if 'cand' in hp else 'n/a')
print('agg joint', float((c(out['agg_hand_joint']) - same['hand_agg_joint']).abs().max()), 'agg mano', float((c(out['agg_hand_mano']) - same['hand_agg_mano']).abs().max()))
print('per-joint abs diff (mm):', ((c(out['agg_hand_joint']) - same['hand_agg_joint']).norm(dim=-1) * 1000).round().tolist())
sc = hp['score']
print('oracle score img1 finger0 sorted desc (idx,val):', [(int(i), float(sc[1, 0, i])) for i in sc[1, 0].argsort(descending=True)[:8]])
print('oracle score img1 cand 18 vs 21 per finger:', sc[1, :, 18].tolist(), sc[1, :, 21].tolist())
import vpho_amd.model.engine as EM
gs = c(gd['hand_phys_score'])
print('gpu score shape', tuple(gs.shape))
gs = gs.reshape(2, -1, 5) if gs.numel() == 2 * 31 * 5 else gs
print('gpu score img1 cand 18 vs 21:', gs[1, 18].tolist() if gs.dim() == 3 else gs[1].tolist()[:10], gs[1, 21].tolist() if gs.dim() == 3 else '')
from oracle.aggregation import local_to_global, nearest
from oracle.mano import get_hand_verts
cand = c(gd['cand58']).reshape(2, 31, 58)
print('cand58 vs oracle-built: n/a; candidate 21 pose img1:', cand[1, 21, :6].tolist())
vert, _ = get_hand_verts(a['mano'], cand.reshape(-1, 58)[:, :48], cand.reshape(-1, 58)[:, 48:])
vert = vert.reshape(2, 31, 778, 3)
gv = c(gd['cand_vert']).reshape(2, 31, 778, 3)
dv = (gv - vert).abs().amax(dim=(2, 3))
print('cand verts max abs per candidate img1:', [f'{x:.1e}' for x in dv[1].tolist()])
vert_abs = gv + data['root_joint_flip'][:, None, None]
fl = c(gf['force_local'])[:, None].expand(2, 31, 32, 3)
fp, fg = local_to_global(a['anchor'], ANCHOR_SKELETON, fl, vert_abs)
gfp, gfg = c(gd['cand_force_point']).reshape(2, 31, 32, 3), c(gd['cand_force_global']).reshape(2, 31, 32, 3)
print('force_point diff per cand img1:', [f'{x:.1e}' for x in (gfp - fp).abs().amax(dim=(2, 3))[1].tolist()])
print('force_global diff per cand img1:', [f'{x:.1e}' for x in (gfg - fg).abs().amax(dim=(2, 3))[1].tolist()])
ov = c(gd['obj_vert'])
dmin, amin = nearest(gfp, ov[:, None])
fn = gfg.norm(dim=-1); fw = fn / fn.sum(dim=-1, keepdim=True)
I = (gfg / fn[..., None]).sum(-2).norm(dim=-1)
score = -(fw * dmin * I[:, :, None])
from oracle.aggregation import FINGER_FORCE_LEVEL
fs = torch.stack([score[:, :, FINGER_FORCE_LEVEL[f]].sum(-1) for f in range(5)], -1)
print('oracle score from GPU force anchors, cand 21:', fs[1, 21].tolist())
print('I cand 18/21:', I[1, 18].item(), I[1, 21].item(), ' dmin sum', dmin[1, 18].sum().item(), dmin[1, 21].sum().item())
print('--- level-3 top-k detail')
for lvl in range(1, 4):
    gi, oi = c(gd['hand_topk'][lvl]).long(), od['hand']['topk'][lvl].long()
    gv, ov_ = c(gd['hand_val'][lvl]), od['hand']['val'][lvl]
    print(lvl, 'gpu idx', tuple(gi.shape), 'oracle idx', tuple(oi.shape), 'gpu val', tuple(gv.shape), 'oracle val', tuple(ov_.shape))
    if gi.shape != oi.shape:
        gi2 = gi.reshape(2, 5, -1).transpose(1, 2); gv2 = gv.reshape(2, 5, -1).transpose(1, 2)
    else:
        gi2, gv2 = gi, gv
    ne = (gi2 != oi)
    print('   transposed compare: n_diff', int(ne.sum()), 'val max abs', float((gv2 - ov_).abs().max()))
    if ne.any():
        w = ne.nonzero()[:6]
        for b, k, f in w.tolist():
            print('   img', b, 'rank', k, 'finger', f, 'gpu idx', int(gi2[b, k, f]), 'val', float(gv2[b, k, f]), '| oracle idx', int(oi[b, k, f]), 'val', float(ov_[b, k, f]))
