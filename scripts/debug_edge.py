import sys, copy, numpy as np, torch
sys.argv=sys.argv[:1]; sys.path.insert(0,'.')
from oracle import vpho as OV
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from vpho_amd.configs.args import cfg
from vpho_amd.synth import synth_batch, synth_state_dict
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.model.engine import Engine
a=synthetic_assets(0); m=vpho_net(a); sd=synth_state_dict(m,1); m.load_state_dict(sd)
bs,S,steps,kh,ko,T0=3,6,4,5,3,0.2
data=synth_batch(bs,a,seed=300+bs)
data['is_right']=torch.tensor([False]*bs); data['is_grasped']=torch.tensor([False]*bs)
data['root_joint']=data['root_joint_flip'].clone(); data['root_joint'][~data['is_right'],0]*=-1
cfg.sample_num,cfg.sampling_steps,cfg.topk_hand,cfg.topk_obj,cfg.sample_T0=S,steps,kh,ko,T0
torch.manual_seed(21); nh,no=torch.randn(bs*S,96),torch.randn(bs*S,9)
ref,ri=OV.predict(sd,a,ANCHOR_SKELETON,data,sample_num=S,sample_T0=T0,sampling_steps=steps,topk_hand=kh,topk_obj=ko,noise_hand=nh,noise_obj=no)
m=m.cuda().eval(); g={k:(v.cuda() if torch.is_tensor(v) else v) for k,v in data.items()}
eng=Engine(m); out=eng.predict(g,noise_hand=nh,noise_obj=no); torch.cuda.synchronize()
ga,ra=eng.last_info['agg'],ri['agg']
for l in range(4):
    got=ga['hand_topk'][l].cpu(); got=got[:,0] if l==0 else got.permute(0,2,1)
    gv=ga['hand_val'][l].cpu(); gv=gv[:,0] if l==0 else gv.permute(0,2,1)
    eq=np.array_equal(got.numpy(), ra['hand']['topk'][l].numpy())
    print('level',l,'equal',eq)
    if not eq:
        for b in range(bs):
            print(' b',b,'got idx',got[b].numpy().T.tolist()); print('     ref idx',ra['hand']['topk'][l][b].numpy().T.tolist())
            print('     got val',np.round(gv[b].numpy().T,6).tolist()); print('     ref val',np.round(ra['hand']['val'][l][b].numpy().T,6).tolist())
