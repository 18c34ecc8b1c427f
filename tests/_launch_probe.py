"""Helper of tests/test_launch_cpu.py: an entry point shaped like bench.py (``--gpus N``), on gloo, no GPU."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--tag', default='x')
    args = ap.parse_args()
    from vpho_amd.launch import maybe_spawn, world_from_env
    maybe_spawn(args.gpus)
    import torch
    import torch.distributed as dist
    world, rank, local = world_from_env(args.gpus)
    if world > 1:
        from vpho_amd.launch import init_process_group
        init_process_group(None)                          # no device: gloo; loud on failure
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({'n_gpus': world, 'sum': float(t), 'tag': args.tag, 'omp': os.environ.get('OMP_NUM_THREADS'),
                          'pid_is_child': 'TORCHELASTIC_RUN_ID' in os.environ}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
