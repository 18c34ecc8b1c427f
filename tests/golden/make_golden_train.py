"""Golden vectors for the DSM training step by running the reference's own BaseDenoiser, ve_marginal_prob and loss_fn
(lib/model/denoiser.py, lib/model/sde.py, lib/model/score_based_model.py) with autograd, then torch.optim.AdamW as the
reference's trainer configures it (train_diff_hand_obj.py:49-52).  Run in the build container only.

``loss_fn`` draws t and z itself (torch.rand / torch.randn_like on the CPU generator): the same draws are re-derived here from
the same seed, in the same order, and stored with the fixture.  Gradients of the large first ParallelLinear are stored as a
strided sample + their norm (the full tensor is 46 MB)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

STRIDE = 9973


def main():
    import tempfile
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.synth import synth_state_dict
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_train_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['main.py', '--mode', 'train']
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    from lib.model.denoiser import BaseDenoiser
    from lib.model.sde import init_sde
    from lib.model.score_based_model import loss_fn
    from vpho_amd.model.VPHO import vpho_net
    sd_all = synth_state_dict(vpho_net(assets), seed=1)
    _, marginal, sde_fn, eps, _ = init_sde('ve')
    G = {}
    for name, head, D in (('hand', 'mano_pose', 96), ('obj', 'obj', 9)):
        net = BaseDenoiser(marginal, head=head)
        pre = f'denoiser_{name}.'
        missing, unexpected = net.load_state_dict({k[len(pre):]: v for k, v in sd_all.items() if k.startswith(pre)}, strict=True)
        net.train()
        bs, R = 6, 3
        g = np.random.default_rng(41 + D)
        feat = torch.from_numpy(g.normal(size=(bs, 1024)).astype(np.float32) * 0.3)
        gt = torch.from_numpy(g.normal(size=(bs, D)).astype(np.float32) * 0.5)
        feat_req = feat.clone().requires_grad_(True)
        opt = torch.optim.AdamW(net.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=2e-4)
        torch.manual_seed(1000 + D)
        state = torch.get_rng_state()
        total = 0
        for _ in range(R):                                         # ScoreBasedModelAgent.get_score_loss (:117-128)
            total = total + loss_fn(model=net, data={'feat': feat_req, 'gt_pose': gt}, marginal_prob_func=marginal, sde_fn=sde_fn, eps=eps)
        total = total / R
        total.backward()
        torch.set_rng_state(state)                                 # the draws loss_fn made, in its order
        ts, zs = [], []
        for _ in range(R):
            ts.append(torch.rand(bs) * (1. - eps) + eps)
            zs.append(torch.randn_like(gt))
        G[f'{name}_feat'], G[f'{name}_gt'] = feat.numpy(), gt.numpy()
        G[f'{name}_t'], G[f'{name}_z'] = torch.stack(ts).numpy(), torch.stack(zs).numpy()
        G[f'{name}_loss'] = np.float64(total.item())
        G[f'{name}_dfeat'] = feat_req.grad.numpy()
        before = {k: v.detach().clone() for k, v in net.named_parameters()}
        for k, v in net.named_parameters():
            gflat = v.grad.reshape(-1)
            G[f'{name}_gnorm_{k}'] = np.float64(gflat.double().norm().item())
            G[f'{name}_gsample_{k}'] = gflat[::STRIDE].numpy().copy()
        opt.step()
        for k, v in net.named_parameters():
            G[f'{name}_psample_{k}'] = v.detach().reshape(-1)[::STRIDE].numpy().copy()
            G[f'{name}_dnorm_{k}'] = np.float64((v.detach() - before[k]).double().norm().item())
        print(name, 'loss', total.item(), 'params', [k for k, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(HERE, 'golden_train_score.npz'), **G)
    print(os.path.getsize(os.path.join(HERE, 'golden_train_score.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
