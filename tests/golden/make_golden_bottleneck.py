"""Golden vectors for a training-mode Bottleneck by running the reference's own module
(lib/model/backbone_FPN_HFL.py:311-350) with autograd.  Run in the build container only; same stubs as make_golden.py."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    from vpho_amd.assets import synthetic_assets
    assets = synthetic_assets(0)
    tmp = tempfile.mkdtemp(prefix='vpho_golden_bn_')
    MG.write_assets(tmp, assets)
    os.chdir(tmp)
    sys.argv = ['main.py']
    sys.path.insert(0, MG.REF)
    MG.install_stubs(assets)
    import torch.nn as nn
    from lib.model.backbone_FPN_HFL import Bottleneck
    G = {}
    for tag, inpl, planes, stride in (('id', 64, 16, 1), ('down', 32, 16, 2)):
        torch.manual_seed(3 + stride)
        down = None
        if stride != 1 or inpl != planes * 4:
            down = nn.Sequential(nn.Conv2d(inpl, planes * 4, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        blk = Bottleneck(inpl, planes, stride, down).train()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
        for k, v in blk.state_dict().items():
            G[f'{tag}_init_{k}'] = v.detach().numpy().copy()
        x = torch.randn(3, inpl, 12, 12).requires_grad_(True)
        out = blk(x)
        dout = torch.randn(out.shape)
        out.backward(dout)
        G[f'{tag}_x'], G[f'{tag}_dout'], G[f'{tag}_out'], G[f'{tag}_dx'] = x.detach().numpy(), dout.numpy(), out.detach().numpy(), x.grad.numpy()
        for k, v in blk.named_parameters():
            G[f'{tag}_grad_{k}'] = v.grad.numpy()
        for k, v in blk.state_dict().items():
            if 'running' in k:
                G[f'{tag}_after_{k}'] = v.detach().numpy().copy()
        print(tag, out.shape, float(out.abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'golden_bottleneck.npz'), **G)
    print(os.path.getsize(os.path.join(HERE, 'golden_bottleneck.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
