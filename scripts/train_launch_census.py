#!/usr/bin/env python
"""Which lines of the training step launch torch's own kernels (copies, fills, element-wise glue)?  One step of train.py's workload under
torch.profiler (CPU activity only: no tracing library is attached to the GPU), every aten operator that launches device work attributed
to the innermost vpho_amd source line on its Python stack.  Output: a table sorted by count.

    python scripts/train_launch_census.py [--bs 64] > gpurun_out/census.txt
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--repeat_num', type=int, default=20)
    args = ap.parse_args()
    import torch
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict, synth_batch
    from vpho_amd.train_step import DiffusionTrainStep
    from vpho_amd.trainer import synthetic_mano_targets
    dev = torch.device('cuda', 0)
    torch.manual_seed(206)
    assets = synthetic_assets(0)
    step = DiffusionTrainStep(synth_state_dict(vpho_net(assets), seed=1), dev, assets=assets)
    bs = args.bs
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=11, rank=0).items()}
    g = torch.Generator().manual_seed(100)
    data['hm_hand'] = (torch.rand(bs, 21, 64, 64, generator=g) * 0.2).to(dev)
    data['hm_obj'] = (torch.rand(bs, 27, 64, 64, generator=g) * 0.2).to(dev)
    gt_h = (torch.randn(bs, 96, generator=g) * 0.5).to(dev) + torch.tensor([1., 0, 0, 0, 1, 0], device=dev).repeat(16)
    gt_o = (torch.randn(bs, 9, generator=g) * 0.5).to(dev)
    data.update(synthetic_mano_targets(step.mano_head.mano, gt_h, (torch.randn(bs, 10, generator=g) * 0.5).to(dev), data['is_right']))
    data['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(dev)
    for _ in range(2):
        step.step(data, gt_h, gt_o, repeat_num=args.repeat_num)
    torch.cuda.synchronize()
    # every aten operator of one step, seen at the Python dispatch level, attributed to the innermost vpho_amd frame of the Python stack
    from torch.utils._python_dispatch import TorchDispatchMode
    root = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    by_line = collections.Counter()
    by_op = collections.Counter()
    VIEWS = ('view', 'reshape', 'permute', 'transpose', 't.', 'slice', 'select', 'expand', 'unsqueeze', 'squeeze', 'as_strided', 'alias', 'detach',
             'empty', 'unbind', 'split', 'narrow', 'lift_fresh', '_unsafe_view', 'is_', 'sym_', 'stride', 'size', 'numel', 'dim', 'item', '_local_scalar')

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace('aten.', '')
            if not any(name.startswith(v) for v in VIEWS):
                f = sys._getframe(1)
                where = '?'
                while f is not None:
                    fn = f.f_code.co_filename
                    if ('vpho_amd' in fn or fn.endswith('train.py')) and 'train_launch_census' not in fn:
                        where = f'{fn.replace(root + "/", "")}:{f.f_lineno} {f.f_code.co_name}'
                        break
                    f = f.f_back
                by_line[(where, name)] += 1
                by_op[name] += 1
            return func(*args, **(kwargs or {}))

    with Census():
        step.step(data, gt_h, gt_o, repeat_num=args.repeat_num)
    torch.cuda.synchronize()
    print(f'# one training step, bs {bs}: aten leaf operators by innermost vpho_amd source line')
    for (where, op), n in by_line.most_common(120):
        print(f'{n:6d}  {op:24s} {where}')
    print('# totals by operator')
    for op, n in by_op.most_common():
        print(f'{n:6d}  {op}')


if __name__ == '__main__':
    main()
