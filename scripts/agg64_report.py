"""GPU: README config, one batch of N images; HIP aggregation vs the oracle's aggregation fed the HIP path's own candidates.
Dumps everything needed to analyse selection differences offline to gpurun_out/agg64.pt."""
import sys, os, torch
sys.argv = sys.argv[:1]
sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.model.engine import Engine
from vpho_amd.synth import bench_state_dict, synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from oracle.aggregation import hoi_aggregate
from oracle.compare import parity_summary

n = int(os.environ.get('N', '64'))
S = 100
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a)
sd = bench_state_dict(m, 1) if os.environ.get('W', 'c') == 'c' else synth_state_dict(m, 1)
m.load_state_dict(sd); m = m.cuda().eval()
eng = Engine(m)
eng.use_graphs = False
data = synth_batch(n, a, seed=777)
torch.manual_seed(99)
nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
out = eng.predict(gdata, noise_hand=nh, noise_obj=no)
torch.cuda.synchronize()
info = eng.last_info
gf, gd = info['features'], info['agg']
c = lambda t: t.detach().cpu()
fl = c(out['diff_final_hand_mano']).reshape(-1, 58)
same = hoi_aggregate(a, ANCHOR_SKELETON, cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                     root_joint=data['root_joint'], is_right=data['is_right'], force_local=c(gf['force_local']),
                     is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(), hand_pose_regression=c(gf['mano_pose']),
                     hand_shape=fl[:, 48:], hand_heatmap=c(gf['hand_heatmap']), hand_bbox=data['bbox_hand'], hand_topk=30,
                     obj_pose6d=c(out['diff_final_obj_6d']), obj_heatmap=c(gf['obj_heatmap']), obj_bbox=data['bbox_obj_rect'],
                     obj_topk=10, obj_name=data['obj_name'])
so = dict(agg_hand_joint=same['hand_agg_joint'], agg_hand_vert=same['hand_agg_vert'], agg_hand_mano=same['hand_agg_mano'], agg_obj_6d=same['obj_agg_6d'])
res, rep = parity_summary(out, so, gd, same['dbg'], S)
print({k: v for k, v in res.items() if k != 'per_stage'})
print(res['per_stage'])
print('hand diffs per image', rep['hand_differences_per_image'].tolist())
# second run of the HIP aggregation on the same inputs: determinism
agg2, gd2 = eng.aggregate(gf, gdata, out['diff_final_hand_mano'].reshape(-1, 58).contiguous(), out['diff_final_obj_6d'], S, 30, 10)
torch.cuda.synchronize()
for lvl in range(4):
    print('rerun level', lvl, 'idx equal', torch.equal(gd2['hand_topk'][lvl], gd['hand_topk'][lvl]), 'val equal', torch.equal(gd2['hand_val'][lvl], gd['hand_val'][lvl]))
print('rerun agg equal', torch.equal(agg2['hand_agg_mano'], out['agg_hand_mano']))
# per-image (bs=1) HIP aggregation vs the batched one
dump = dict(gd={k: ([c(t) for t in v] if isinstance(v, list) else c(v)) for k, v in gd.items()}, od=same['dbg'], rep=rep,
            out={k: c(out[k]) for k in ('agg_hand_mano', 'agg_hand_joint', 'agg_obj_6d')}, same={k: v for k, v in so.items()})
os.makedirs('gpurun_out', exist_ok=True)
torch.save(dump, 'gpurun_out/agg64.pt')
