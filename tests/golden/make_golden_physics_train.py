"""Golden vectors for the physics branch of the training step (lib/model/VPHO.py:170-172,205-212): the reference's own
``CrossModule`` x 2 (lib/model/cross_module.py:91-137: 3x3 projections, NeRF gravity embedding, positional code, one post-norm
``nn.TransformerEncoderLayer`` attending over the BATCH axis -- quirk Q3), ``HeadPhysics`` (lib/model/physics.py:648-721) and its
five losses (``get_loss`` :456-500: force, gravity, torque, supervised, CoM) under autograd, weighted as VPHO.py:214-219 does.
Modules are in train() mode with every dropout probability set to 0 (PositionalEncoding, the encoder layer's three Dropout
modules and the attention dropout): dropout masks come from the global RNG stream and cannot be part of a fixture.
Stored: losses, forward outputs, gradients of every parameter (whole tensors up to 4096 entries, else norm + strided sample) and of
the two stage maps.  Weights: vpho_amd.synth.synth_state_dict(seed=1).  Run in the build container only."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

BS, STRIDE = 6, 997
W = dict(force_loss=1.0, gravity_loss=1.0, torque_loss=30.0, supervised_loss=10.0, CoM_loss=100.0)      # lib/configs/args.py:213-219


def inputs(assets):
    g = np.random.default_rng(123)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    grav = g.normal(size=(BS, 1, 3))
    grav /= np.linalg.norm(grav, axis=-1, keepdims=True)
    vert = np.asarray(assets['mano']['v_template'])[None] + g.normal(size=(BS, 778, 3)) * 0.002 + np.array([0.02, -0.01, 0.7])
    return dict(st_h=f32(g.normal(size=(BS, 256, 8, 8)) * 0.2), st_o=f32(g.normal(size=(BS, 256, 8, 8)) * 0.2), gravity=f32(grav),
                gt_vert=f32(vert), gt_CoM=f32(np.array([0.05, 0.0, 0.7]) + g.normal(size=(BS, 1, 3)) * 0.02),
                gt_force_local=f32(g.normal(size=(BS, 32, 3)) * 0.1), is_grasped=torch.from_numpy(g.random(BS) < 0.7))


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def main():
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import synth_state_dict
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_phys_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['main.py', '--mode', 'train']
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    import torch.utils.model_zoo as zoo
    import lib.model.backbone_FPN_HFL as ref_fpn
    zoo.load_url = lambda url, **kw: ref_fpn.ResNet(ref_fpn.Bottleneck, [3, 4, 6, 3]).state_dict()
    import lib.model.VPHO as ref_vpho
    torch.manual_seed(0)
    ref = ref_vpho.vpho_net()
    sys.argv = ['x']
    from vpho_amd.model.VPHO import vpho_net
    sd = synth_state_dict(vpho_net(assets), seed=1)
    missing, _ = ref.load_state_dict(sd, strict=False)
    assert not missing
    ref.train()
    zero_dropout(ref)
    d = inputs(assets)
    st_h, st_o = d['st_h'].clone().requires_grad_(True), d['st_o'].clone().requires_grad_(True)
    # VPHO.py:170-172
    enc_phy_hand, _, _ = ref.cross_hand(st_h, st_o.detach(), d['gravity'])
    _, enc_phy_obj, _ = ref.cross_obj(st_h.detach(), st_o, d['gravity'])
    pd = ref.head_physics(enc_phy_hand, enc_phy_obj)
    # VPHO.py:205-212
    gt_force_point, pd_force_global = ref.head_physics.from_local_to_global(force_local=pd['force_local'], hand_vert=d['gt_vert'])
    losses = ref.head_physics.get_loss(gt_force_point=gt_force_point, pd_force_global=pd_force_global, gt_CoM=d['gt_CoM'], pd_CoM=pd['CoM'],
                                       gt_force_local=d['gt_force_local'], pd_force_local=pd['force_local'], gt_gravity=d['gravity'],
                                       is_grasped=d['is_grasped'])
    assert set(losses) == set(W), losses.keys()
    total = sum(losses[k] * W[k] for k in W)
    total.backward()
    G = {k: np.float64((losses[k] * W[k]).item()) for k in W}
    G['force_local'], G['CoM'], G['scale'] = pd['force_local'].detach().numpy(), pd['CoM'].detach().numpy(), pd['scale'].detach().numpy()
    G['tok_hand'] = enc_phy_hand.detach().numpy()[:, ::8, ::16]
    G['tok_obj'] = enc_phy_obj.detach().numpy()[:, ::8, ::16]
    G['force_point'], G['force_global'] = gt_force_point.detach().numpy(), pd_force_global.detach().numpy()
    n = 0
    for name, p in ref.named_parameters():
        if not name.startswith(('cross_hand.', 'cross_obj.', 'head_physics.')):
            continue
        if p.grad is None:
            G['nograd_' + name] = np.array(1)
            continue
        gf = p.grad.reshape(-1)
        G['gnorm_' + name] = np.float64(gf.double().norm().item())
        G['g_' + name] = gf.numpy().copy() if gf.numel() <= 4096 else gf[::STRIDE].numpy().copy()
        n += 1
    for k, t in (('st_h', st_h), ('st_o', st_o)):
        G['gnorm_' + k] = np.float64(t.grad.double().norm().item())
        G['g_' + k] = t.grad.reshape(-1)[::101].numpy().copy()
    out = os.path.join(HERE, 'golden_physics_train.npz')
    np.savez_compressed(out, **G)
    print(n, 'parameter gradients;', [k for k in G if k.startswith('nograd_')], os.path.getsize(out) // 1024, 'KiB;', {k: float(G[k]) for k in W})


if __name__ == '__main__':
    main()
