import sys, os, numpy as np, torch
sys.argv = ['x']; sys.path.insert(0, '.')
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict
from vpho_amd.train_blocks import EncoderTrain
G = np.load('tests/golden/golden_encoder_train.npz')
sd = synth_state_dict(vpho_net(synthetic_assets(0)), seed=1)
net = EncoderTrain(sd, 'encoder_hand', 'cuda')
g = np.random.default_rng(21)
x = torch.from_numpy((g.normal(size=(12, net.cin, 32, 32)) * 0.3).astype(np.float32))
xin = torch.zeros(12, 32, 32, net.cin_pad); xin[..., :net.cin] = x.permute(0, 2, 3, 1)
enc, stages = net.forward(xin.cuda())
A = torch.from_numpy(g.normal(size=tuple(enc.shape)).astype(np.float32))
B = torch.from_numpy(g.normal(size=(12, stages[1].shape[3], stages[1].shape[1], stages[1].shape[2])).astype(np.float32))
dx, grads = net.backward(A.cuda(), B.permute(0, 2, 3, 1).contiguous().cuda())
for k in [k[len('gnorm_'):] for k in G.files if k.startswith('gnorm_')]:
    gr = grads[k].reshape(-1).cpu(); nrm = float(G['gnorm_' + k])
    rms = nrm / max(1.0, gr.numel() ** 0.5)
    print(f'{k:28s} norm ratio {float(gr.double().norm())/nrm:.5f}  max sample err / rms {float(np.abs(gr[::499].numpy() - G["gsample_" + k]).max())/(rms+1e-30):.4f}')
