#!/usr/bin/env python
"""torch operators that launch device work in one predict step of the README config (the HIP graphs of the feature path and the aggregation
replay captured launches: what shows up here is what runs OUTSIDE them), by vpho_amd source line.  python scripts/predict_launch_census.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.configs.args import cfg
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import bench_state_dict, synth_batch
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
    assets = synthetic_assets(0)
    m = vpho_net(assets)
    m.load_state_dict(bench_state_dict(m))
    m = m.cuda().eval()
    data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, assets, seed=11).items()}
    for _ in range(3):
        m(data, mode='predict')
    torch.cuda.synchronize()
    root = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    by_line = collections.Counter()
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
                    if 'vpho_amd' in fn and 'census' not in fn:
                        where = f'{fn.replace(root + "/", "")}:{f.f_lineno} {f.f_code.co_name}'
                        break
                    f = f.f_back
                by_line[(where, name)] += 1
            return func(*args, **(kwargs or {}))

    with Census():
        m(data, mode='predict')
    torch.cuda.synchronize()
    print('# one predict step, bs 64, README config: torch operators outside the HIP graphs, by source line')
    for (where, op), n in by_line.most_common(60):
        print(f'{n:5d}  {op:24s} {where}')


if __name__ == '__main__':
    main()
