#!/usr/bin/env python
"""End-to-end training step: images/s of forward + backward + AdamW on synthetic batches.

    python train.py --steps 10 [--bs 64 --repeat_num 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P train.py --gpus N

One step = the reference's ``accel.backward(loss) + optimizer.step()`` (lib/engine/train_diff_hand_obj.py:169-199) with all 13 losses
of lib/model/VPHO.py:190-212: training-mode two-branch ResNet-50/FPN, RoIAlign, heat-map heads and re-alignment, the two encoders,
repeat_num DSM draws per score network, head_mano + MANO layer, both cross modules + head_physics, the whole backward, bucketed
all-reduces of the flat gradient buffer (RCCL) overlapped with it under data parallelism, AdamW on all 569 tensors.  One JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--bs', type=int, default=64)
    ap.add_argument('--repeat_num', type=int, default=20)
    ap.add_argument('--breakdown', action='store_true', help='also time forward+backward and the optimiser separately')
    ap.add_argument('--seed', type=int, default=206)
    ap.add_argument('--no_roofline', action='store_true', help='skip the instrumented repeat that feeds the roofline block (profiler passes)')
    args = ap.parse_args()
    from vpho_amd.launch import maybe_spawn, world_from_env
    maybe_spawn(args.gpus)             # N > 1 from a bare shell: start the N rank processes (before any GPU call)
    sys.argv = sys.argv[:1]
    import torch
    import torch.distributed as dist
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict, synth_batch
    from vpho_amd.train_step import DiffusionTrainStep
    world, rank, local = world_from_env(args.gpus)
    # rehearsal aid for a 1-GPU box: VPHO_REHEARSE_ONE_GPU=1 puts every rank on cuda:0 and uses gloo (timings meaningless)
    rehearse = os.environ.get('VPHO_REHEARSE_ONE_GPU') == '1'
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from vpho_amd.launch import init_process_group
    init_process_group(dev)                               # loud on failure: bounded timeout, expected vs observed world, first collective
    # base_trainer.py:39-50: seed + rank * 1e8 -- every rank draws its own DSM times / noise from the device generator
    torch.manual_seed(args.seed + rank * 100000000)
    torch.cuda.manual_seed(args.seed + rank * 100000000)
    assets = synthetic_assets(0)
    sd = synth_state_dict(vpho_net(assets), seed=1)
    step = DiffusionTrainStep(sd, dev, assets=assets)
    bs = args.bs
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=11, rank=rank).items()}
    g = torch.Generator().manual_seed(100 + rank)
    data['hm_hand'] = (torch.rand(bs, 21, 64, 64, generator=g) * 0.2).to(dev)
    data['hm_obj'] = (torch.rand(bs, 27, 64, 64, generator=g) * 0.2).to(dev)
    # ground truth: 16 x rot6d spread around the identity (one pose for the DSM target and the MANO losses), object rot6d + translation
    gt_h = (torch.randn(bs, 96, generator=g) * 0.5).to(dev) + torch.tensor([1., 0, 0, 0, 1, 0], device=dev).repeat(16)
    gt_o = (torch.randn(bs, 9, generator=g) * 0.5).to(dev)
    from vpho_amd.trainer import synthetic_mano_targets
    data.update(synthetic_mano_targets(step.mano_head.mano, gt_h, (torch.randn(bs, 10, generator=g) * 0.5).to(dev), data['is_right']))
    data['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(dev)          # pseudo-force labels: the physics losses are part of the step

    first = None
    for _ in range(args.warmup):
        L = step.step(data, gt_h, gt_o, repeat_num=args.repeat_num)
        first = first or {k: float(v) for k, v in L.items()}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L = step.step(data, gt_h, gt_o, repeat_num=args.repeat_num)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    res = {'metric': 'end-to-end training images/s (all 13 losses of vpho_net.forward(mode=train), every module trained)',
           'value': world * args.steps * bs / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'ms_per_step': 1e3 * dt / args.steps,
           'dtype': 'f32', 'config': {'per_gpu_batch': bs, 'repeat_num': args.repeat_num, 'patch': 256},
           # parameters as the reference counts them: convolution weights by their (cout, cin, kh, kw) shape, not by the packed layout's padded channels
           'trained_tensors': len(step.names), 'trained_parameters': int(sum((math.prod(step._conv_meta[k]) if k in step._conv_meta else step.master[k].numel()) for k in step.names)),
           'loss_first': first, 'loss_last': {k: float(v) for k, v in L.items()}}
    if not args.no_roofline:
        # roofline of the step's dominant kernel, measured live: HIP events around every weight-gradient launch (conv_wgrad.hip) in a
        # separate instrumented repeat of the step (the events serialise nothing, but they are kept out of the timed region).  EVERY rank
        # runs the repeat -- the step's gradient exchange is collective --, rank 0 reports its own kernels.  In the timed step the weight
        # gradients run on a stream of their own, overlapping the rest of the backward (conv_backward.WgradStream): an event pair would
        # span the other stream's kernels, so the instrumented repeat keeps everything on one stream -- EXCLUSIVE durations
        from vpho_amd import ops
        ws_before = os.environ.get('VPHO_WGRAD_STREAM')
        os.environ['VPHO_WGRAD_STREAM'] = '0'                       # read by WgradStream.__enter__ on the Python side, once per step
        names = ('conv_wgrad_64x64', 'conv_wgrad_128x128')
        for nm in names:
            ops.prof_enable(nm, True)
        n_prof = 2
        for _ in range(n_prof):
            step.step(data, gt_h, gt_o, repeat_num=args.repeat_num)
        torch.cuda.synchronize()
        prof = {nm: ops.prof_collect(nm) for nm in names}
        for nm in names:
            ops.prof_enable(nm, False)
        os.environ.pop('VPHO_WGRAD_STREAM')
        if ws_before is not None:
            os.environ['VPHO_WGRAD_STREAM'] = ws_before
        PEAK = 157.3                                                   # fp32 MFMA, dense (MI355X_MICROARCH.md)
        blk = {}
        for nm, r in prof.items():
            if r['launches']:
                blk[nm] = {'TFLOP/s': r['flops'] / r['total_ms'] / 1e9, 'frac': r['flops'] / r['total_ms'] / 1e9 / PEAK, 'launches_per_step': r['launches'] / n_prof,
                           'kernel_ms_per_step': r['total_ms'] / n_prof, 'avg_launch_us': 1e3 * r['total_ms'] / r['launches'], 'flop_per_launch_avg': r['flops'] / r['launches'],
                           'bytes_per_launch_avg': r.get('bytes', 0.0) / r['launches']}
        dom = max(blk, key=lambda k: blk[k]['kernel_ms_per_step']) if blk else None
        # the 8-wave class = the 128 x 128 and the 128 x 64 tile (both instantiations' rows of the PMC table, launch-weighted)
        kname = {'conv_wgrad_64x64': 'conv_wgrad_tn_kernel<64, 64, 2, 2>', 'conv_wgrad_128x128': 'conv_wgrad_tn_kernel<128,'}
        res['roofline'] = None if dom is None else {'bound': 'mfma', 'kernel': {'conv_wgrad_64x64': 'conv_wgrad_tn_kernel<64,64,2,2>', 'conv_wgrad_128x128': 'conv_wgrad_tn_kernel<128,128,4,2> and <128,64,4,2> (the 8-wave tiles)'}[dom]
                           + ' (weight gradient dW = dY^T . im2col(x) as an implicit TN GEMM on fp32 MFMA)',
                           'achieved': blk[dom]['TFLOP/s'], 'peak': PEAK, 'unit': 'TFLOP/s', 'frac': blk[dom]['frac'],
                           # the committed counter pass is of the default batch (64 per GPU): no figure for another batch size
                           'traffic': pmc_traffic(kname[dom]) if bs == 64 else None, 'traffic_source': pmc_traffic(kname[dom], source=True) if bs == 64 else None,
                           'algorithmic_bytes_per_launch': blk[dom].get('bytes_per_launch_avg'),
                           'share_of_step': blk[dom]['kernel_ms_per_step'] / res['ms_per_step'],
                           'timing': 'HIP events around every launch on the launch stream, in an instrumented repeat of the step with the weight gradients kept on '
                                     'the main stream (exclusive durations; the timed steps overlap them with the rest of the backward)', 'classes': blk}
    if args.breakdown:
        draws = dict(t_h=torch.rand(args.repeat_num, bs, device=dev) * 0.99 + 0.01, z_h=torch.randn(args.repeat_num, bs, 96, device=dev),
                     t_o=torch.rand(args.repeat_num, bs, device=dev) * 0.99 + 0.01, z_o=torch.randn(args.repeat_num, bs, 9, device=dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step.loss_and_grads(data, gt_h, gt_o, draws)
        torch.cuda.synchronize()
        res['ms_forward_backward'] = 1e3 * (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            step.fpn.forward(data['rgb'])
        torch.cuda.synchronize()
        res['ms_backbone_forward'] = 1e3 * (time.perf_counter() - t0) / 3
    if world > 1:                                       # data parallelism keeps the replicas identical: compare parameter checksums
        chk = torch.stack([step.master[k].double().sum() for k in step.names]).sum().reshape(1)
        chk = chk.cpu() if rehearse else chk
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        res['replicas_in_sync'] = bool(all(float(c) == float(allc[0]) for c in allc))
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel, source=False):
    """HBM bytes per launch of `kernel` (launch-weighted over its instantiations' rows) from the newest committed rocprofv3 --pmc summary of
    the training step (profiles/r0N_train_pmc_hbm_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, scripts/profile_round.sh PART=train); None when absent"""
    root = os.path.dirname(os.path.abspath(__file__))
    for rnd in ('r06', 'r05', 'r04'):
        path = os.path.join(root, 'profiles', f'{rnd}_train_pmc_hbm_traffic.json')
        try:
            with open(path) as f:
                tab = json.load(f)
        except Exception:
            continue
        hit = {n: v for n, v in tab.items() if kernel.replace(' ', '') in n.replace(' ', '')}
        if hit:
            if source:
                return os.path.relpath(path, root)
            n = sum(v['launches'] for v in hit.values())
            return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hit.values()) / max(n, 1)
    return None


if __name__ == '__main__':
    main()
