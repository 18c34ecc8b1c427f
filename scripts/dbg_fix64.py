import sys, numpy as np, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.model.engine import Engine
from vpho_amd.synth import bench_state_dict, synth_batch
from vpho_amd import ops
from oracle import vpho as OV, rotations as R, nets as N
from oracle.mano import get_hand_verts
Rz = np.load('tests/golden/golden_predict_readme64.npz')
assets = synthetic_assets(0); m = vpho_net(assets); sd = bench_state_dict(m, 1); m.load_state_dict(sd); m = m.cuda().eval()
cdata = synth_batch(64, assets, seed=int(Rz['data_seed']))
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cdata.items()}
eng = Engine(m)
with torch.no_grad():
    f = eng.features(data)
torch.cuda.synchronize()
b = 6
enc = f['encoding_hand'].cpu()
F = torch.nn.functional
h = F.leaky_relu(F.linear(enc, sd['head_mano.base_layer.0.weight'], sd['head_mano.base_layer.0.bias']), 0.01)
h = F.leaky_relu(F.linear(h, sd['head_mano.base_layer.2.weight'], sd['head_mano.base_layer.2.bias']), 0.01)
x6 = F.linear(h, sd['head_mano.fc_pose.weight'], sd['head_mano.fc_pose.bias']).reshape(64, 16, 6)      # from HIP's encoding, CPU arithmetic
aa_cpu = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(x6)).reshape(64, 48)
aa_hip = f['mano_pose'].cpu()
d = (aa_cpu - aa_hip).abs().reshape(64, 16, 3).amax(-1)
print('axis-angle HIP vs CPU conversion of the same encoding: max', float(d.max()), 'at', divmod(int(d.argmax()), 16))
print('image 6 per-joint |d aa|:', [f'{v:.1e}' for v in d[b].tolist()])
j = int(d[b].argmax())
print('joint', j, 'x6', x6[b, j].tolist(), 'aa cpu', aa_cpu[b, 3*j:3*j+3].tolist(), 'aa hip', aa_hip[b, 3*j:3*j+3].tolist(), 'angle', float(aa_cpu[b, 3*j:3*j+3].norm()), float(aa_hip[b, 3*j:3*j+3].norm()))
Rm_cpu = R.axis_angle_to_matrix(aa_cpu[b, 3*j:3*j+3]); Rm_hip = R.axis_angle_to_matrix(aa_hip[b, 3*j:3*j+3])
print('rotation matrices differ by', float((Rm_cpu - Rm_hip).abs().max()), ' vs rot6d->matrix', float((R.rotation_6d_to_matrix(x6[b, j]) - Rm_hip).abs().max()))
# FK of the SAME axis-angle on both sides
betas = f['mano_shape'].cpu()
v_cpu, j_cpu = get_hand_verts(assets['mano'], aa_hip, betas)
print('FK (HIP kernel) vs oracle FK on HIP pose: joints', float((f['reg_hand_joint'].cpu() - j_cpu).abs().max()), 'verts', float((f['reg_hand_vert'].cpu() - v_cpu).abs().max()))
with torch.no_grad():
    of = OV.features(sd, assets, cdata)
c = lambda t: t.detach().cpu()
for k in ('hf_hr', 'encoding_hand', 'encoding_obj', 'mano_pose', 'mano_shape', 'reg_hand_joint', 'stage_hand', 'force_local'):
    a_, b_ = c(f[k]), of[k]
    if a_.dim() == 4 and a_.shape != b_.shape:
        a_ = a_.permute(0, 3, 1, 2)
    e = (a_.reshape(64, -1) - b_.reshape(64, -1)).abs().amax(1)
    print(f'{k:16s} max err {float(e.max()):.2e} at image {int(e.argmax())}; image 6: {float(e[6]):.2e}; scale {float(b_.abs().max()):.2e}')
ein = c(f['enc_in_hand']).permute(0, 3, 1, 2)[:, :277]
oin = torch.cat((of['hf_hr_rect'], torch.nn.functional.interpolate(OV.align_hm_to_bbox_rectangle(of['hand_heatmap'], cdata['bbox_hand'], cdata['bbox_hand_rect'], 64), size=(32, 32), mode='bilinear', align_corners=False)), 1)
e = (ein - oin).abs()
print('encoder input: per image max err', [f'{v:.1e}' for v in e.amax((1, 2, 3)).tolist()[:10]], ' image 6: feat part', float(e[6, :256].max()), 'heat-map part', float(e[6, 256:].max()))
w = e[6].amax(0)
print('image 6 error by output column (max over rows/channels):', [f'{v:.0e}' for v in w.amax(0).tolist()])
