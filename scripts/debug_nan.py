import sys, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a=synthetic_assets(0); m=vpho_net(a); m.load_state_dict(synth_state_dict(m,1)); m=m.cuda().eval()
data={k:(v.cuda() if torch.is_tensor(v) else v) for k,v in synth_batch(64,a,seed=206).items()}
torch.manual_seed(11)
out=m(data,mode='predict'); torch.cuda.synchronize()
info=m._engine.last_info; ag=info['agg']
bad=(~torch.isfinite(out['agg_hand_mano'])).any(dim=1).nonzero().flatten().tolist()
print('bad images', bad)
for k in ('cascade_pose','cand58','force_point','force_global','obj_vert','phys_score','pose6d_candidate'):
    t=ag[k]; nb=(~torch.isfinite(t.reshape(t.shape[0],-1))).any(dim=1).nonzero().flatten().tolist(); print(k, 'nonfinite images', nb)
for k in ('diff_final_hand_mano','diff_final_obj_6d','agg_obj_6d','force_local','reg_hand_joint'):
    t=out[k]; nb=(~torch.isfinite(t.reshape(t.shape[0],-1))).any(dim=1).nonzero().flatten().tolist(); print(k, nb)
b=bad[0] if bad else 0
print('agg_hand_mano[b]', out['agg_hand_mano'][b])
print('hand_phys_topk[b]', ag['hand_phys_topk'][b])
for l in range(4): print('lvl',l,'val', ag['hand_val'][l][b].flatten()[:8].tolist())
