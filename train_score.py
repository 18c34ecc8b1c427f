#!/usr/bin/env python
"""Score-network training on frozen encodings (first slice): DSM steps/s on synthetic batches.

    python train_score.py --steps 50 [--bs 64 --repeat_num 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P train_score.py --gpus N

One step = what the reference's training forward + backward + optimiser step do for the two denoisers on one batch
(lib/model/VPHO.py:190-191, lib/model/score_based_model.py:117-128, lib/engine/train_diff_hand_obj.py:169-199): repeat_num
DSM draws each for denoiser_hand (96-d) and denoiser_obj (9-d), gradients to all their parameters and to the two encodings,
data-parallel gradient averaging (one RCCL all-reduce per network) and AdamW.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--repeat_num', type=int, default=20)
    args = ap.parse_args()
    from vpho_amd.launch import maybe_spawn, world_from_env
    maybe_spawn(args.gpus)             # N > 1 from a bare shell: start the N rank processes (before any GPU call)
    sys.argv = sys.argv[:1]
    import torch
    import torch.distributed as dist
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict
    from vpho_amd.train_score import ScoreTrainer
    world, rank, local = world_from_env(args.gpus)
    torch.manual_seed(206 + rank * 100000000)             # base_trainer.py:39-50
    torch.cuda.manual_seed(206 + rank * 100000000)
    if os.environ.get('VPHO_REHEARSE_ONE_GPU') == '1':      # every rank on cuda:0 over gloo (1-GPU box; timings meaningless)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from vpho_amd.launch import init_process_group
    init_process_group(dev)                               # loud on failure: bounded timeout, expected vs observed world, first collective
    sd = synth_state_dict(vpho_net(synthetic_assets(0)), seed=1)
    hand, obj = ScoreTrainer(sd, 'denoiser_hand', dev), ScoreTrainer(sd, 'denoiser_obj', dev)
    g = torch.Generator().manual_seed(100 + rank)
    feat_h, feat_o = (torch.randn(args.bs, 1024, generator=g) * 0.3).to(dev), (torch.randn(args.bs, 1024, generator=g) * 0.3).to(dev)
    gt_h, gt_o = (torch.randn(args.bs, 96, generator=g) * 0.5).to(dev), (torch.randn(args.bs, 9, generator=g) * 0.5).to(dev)

    def step():
        lh, _ = hand.step(feat_h, gt_h, repeat_num=args.repeat_num)
        lo, _ = obj.step(feat_o, gt_o, repeat_num=args.repeat_num)
        return lh, lo

    for _ in range(args.warmup):
        l0 = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l1 = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    rows = args.repeat_num * args.bs
    flop = sum(3 * 2.0 * rows * (1408 * t.nheads * 256 + 128 * 128 + t.Dp * 256 + 256 * 256 + t.nheads * 256 * 3) for t in (hand, obj))
    if rank == 0:
        print(json.dumps({'metric': 'score-network DSM training steps/s (hand + object denoiser, frozen encodings)',
                          'value': world * args.steps / dt, 'unit': 'steps/s', 'images_per_s': world * args.steps * args.bs / dt,
                          'n_gpus': world, 'steps': args.steps, 'ms_per_step': 1e3 * dt / args.steps, 'dtype': 'f32',
                          'config': {'per_gpu_batch': args.bs, 'repeat_num': args.repeat_num, 'rows_per_step': rows},
                          'gemm_tflops': flop / (dt / args.steps) / 1e12,
                          'loss_hand_first_last': [float(l0[0]), float(l1[0])], 'loss_obj_first_last': [float(l0[1]), float(l1[1])]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
