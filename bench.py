#!/usr/bin/env python
"""Headline benchmark: eval images/s of ``vpho_net.forward(mode='predict')`` at the README config
(bs=64 per GPU, sample_num=100, sampling_steps=50, topk_hand=30, topk_obj=10, sample_T0=0.65) on synthetic 256x256 crops.

    python bench.py --gpus N --steps 10 --warmup 2        # N > 1: starts its N rank processes itself (vpho_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one forward pass over one per-rank batch (inputs resident in HBM), incl. the CPU prior draw the reference
makes (sde.py:26-28); after the K timed steps the ranks exchange their per-image metric rows with ONE all-gather.
Prints one JSON line (rank 0) with `roofline` (dominant kernel: conv_igemm 128x128 tile, timed with HIP events on its
launch stream during the timed steps) and `cpu_baseline` (the oracle on the host cores, bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# OpenMP workers that spin after every parallel CPU region burn the host's CPU quota for nothing (the GPU boxes give a rank
# 16 CPUs; a throttled cgroup stalls the launch threads in 100 ms periods).  Must be set before torch loads libgomp.
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (16 x the fp32 rate)
FEATURE_GFLOP_PER_IMAGE = 37.09
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_LIMITS = {'mano_fk': 'matrix cores / vector ALU, not HBM: pose blend [hands x 135] x [135 x 2334] + 16-joint transform blend per vertex (6 GFLOP per 6400-hand launch) -- fp32 MFMA in mano_fk_mfma_kernel (the one launch with vertices for all hypotheses: 105 us, and the class\'s HBM bytes: 12-byte vertex stores reach HBM as 1.9 x their size), packed fp32 FMAs in the joints-only mano_fk_kernel<HB> launches; the tables stream from L2',
              'obj_physics': 'vector ALU: 65 536 squared distances per candidate from an LDS-resident point cloud',
              'hand_fuse': 'latency: one workgroup per (image, finger) -- ranking by counting, 30 quaternions, a 4x4 Jacobi eigen-solve',
              'roi_align': 'HBM / L2 gather: window rows read once, pooled output written once',
              'resize_bilinear': 'HBM: read-modify-write of the finer map (top-down add of the FPN)'}
CONV_CLASS_NAME = ('conv_igemm 128x128 tile class = conv_igemm_glds_kernel<128,128,4,2,false> (one tile per workgroup) + '
                   'conv_igemm_pers_kernel<128,128,4,2> (persistent tile walk); fp32 MFMA implicit GEMM, 8 waves, direct-to-LDS tiles')
CONV_CLASS_KERNELS = ('conv_igemm_glds_kernel<128, 128, 4, 2, false>', 'conv_igemm_pers_kernel<128, 128, 4, 2>')
HBM_KERNEL_NAMES = {'mano_fk': 'mano_fk', 'obj_physics': 'obj_physics_kernel', 'hand_fuse': 'hand_fuse_kernel',
                    'roi_align': 'roi_align_nhwc_kernel', 'resize_bilinear': 'resize_bilinear_nhwc_kernel'}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--bs', type=int, default=64)
    p.add_argument('--sample_num', type=int, default=100)
    p.add_argument('--sampling_steps', type=int, default=50)
    p.add_argument('--topk_hand', type=int, default=30)
    p.add_argument('--topk_obj', type=int, default=10)
    p.add_argument('--sample_T0', type=float, default=0.65)
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_kernel_timing', action='store_true')
    p.add_argument('--no_fp64_judge', action='store_true', help='skip the end-to-end float64 judge of the parity leg (~75 s of CPU on 16 cores)')
    p.add_argument('--no_opt_in', action='store_true', help='skip the secondary measurement of the opt-in split-bf16 product path (N=1 only)')
    p.add_argument('--cpu_images', type=int, default=64, help='images of the oracle / parity leg (one batch)')
    p.add_argument('--weights', choices=('conditioned', 'random'), default='conditioned', help='synthetic weight set (vpho_amd.synth)')
    p.add_argument('--pipeline', type=int, default=3, help='evaluation batches kept in flight (1 = sequential loop)')
    p.add_argument('--score_mfma', choices=('f32', 'bf16x6', 'bf16x9'), default='f32',
                   help='score head products: f32 = fp32 MFMA (default, the path parity is stated on); bf16x6 / bf16x9 = opt-in split-bf16 products with fp32 accumulation')
    p.add_argument('--conv_mfma', choices=('f32', 'bf16x6', 'bf16x9'), default='f32', help='the same switch for the convolutions of the feature path (opt-in)')
    p.add_argument('--winograd', action='store_true', help='(default since round 3; kept so old command lines still parse)')
    p.add_argument('--no_winograd', action='store_true', help='A/B aid: the direct implicit GEMM for the 3x3 / stride-1 convolutions too (VPHO_WINOGRAD=0)')
    p.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                   help='weak (default): every rank its own batch of --bs images, like accelerate\'s prepared DataLoader (train_diff_hand_obj.py:121-124); '
                        'strong: ONE global batch of --bs images per step, rank r takes images [r*bs/N, (r+1)*bs/N) (SURVEY 8e); the batch-coupled '
                        'quirks Q3 / Q5 then see the LOCAL batch of bs/N images')
    p.add_argument('--device_prior', action='store_true', help='opt-in: the sampler prior from the device generator (Philox) instead of the CPU default generator (sde.py:26-28); a different random stream')
    p.add_argument('--no_roi_window', action='store_true', help='compute the full stride-4 FPN maps instead of the pixels the RoIAligns read (same results; A/B aid)')
    return p.parse_args()


def main():
    args = parse()
    from vpho_amd.launch import maybe_spawn, world_from_env
    maybe_spawn(args.gpus)             # N > 1 from a bare shell: start the N rank processes (before any GPU call) and exit with their code
    sys.argv = sys.argv[:1]
    import torch
    import torch.distributed as dist
    from vpho_amd.configs.args import cfg
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict, synth_batch
    from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
    from vpho_amd import ops, evaluate as E

    world, rank, local_rank = world_from_env(args.gpus)
    from vpho_amd.hostcpu import usable_cpus
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // max(1, world))))      # the ranks of a node share its CPUs
    # rehearsal aid for a 1-GPU box: VPHO_REHEARSE_ONE_GPU=1 puts every rank on cuda:0 and uses gloo (timings meaningless)
    rehearse = os.environ.get('VPHO_REHEARSE_ONE_GPU') == '1'
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    from vpho_amd.launch import init_process_group
    init_process_group(dev)                               # loud on failure: bounded timeout, expected vs observed world, first collective

    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = \
        args.sample_num, args.sampling_steps, args.topk_hand, args.topk_obj, args.sample_T0
    assets = synthetic_assets(0)
    model = vpho_net(assets)
    from vpho_amd.synth import HM_GAIN_CONTRAST, HM_GAIN_FLAT, bench_state_dict
    # default: heat-maps with contrast + conditioned score networks (vpho_amd.synth.bench_state_dict); `--weights random` is the
    # round-1 set (flat heat-maps, unconditioned score networks: nfev 57/57 instead of 51/51)
    sd = bench_state_dict(model, seed=1) if args.weights == 'conditioned' else synth_state_dict(model, seed=1, hm_gain=HM_GAIN_FLAT)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    torch.manual_seed(206 + rank * 100000000)            # base_trainer.py:39-50 (seed + rank*1e8)
    batches = []
    strong = args.scaling == 'strong'
    lo, hi = E.shard_range(args.bs, rank, world) if strong else (0, args.bs)
    lbs = hi - lo                                         # images of THIS rank per step
    if strong and (lbs <= 0 or args.bs % world):
        raise SystemExit(f'--scaling strong needs --bs ({args.bs}) to be a multiple of the rank count ({world})')
    for i in range(2):                                    # two resident batches, alternated
        # strong scaling: every rank builds the SAME global batch and keeps its contiguous shard
        b = synth_batch(args.bs, assets, seed=206 + i, rank=0 if strong else rank)
        b = {k: v[lo:hi] for k, v in b.items()}
        batches.append({k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in b.items()})
    # synthetic ground truth for the metric rows: MANO FK of a seeded small pose (through the HIP FK)
    from vpho_amd.model.engine import Engine
    if args.no_roi_window:
        os.environ['VPHO_ROI_WINDOW'] = '0'                # read by every execution plan (this one and the pipeline slots')
    score_mfma = os.environ.get('VPHO_SCORE_MFMA', 'f32') if args.score_mfma == 'f32' else args.score_mfma
    os.environ['VPHO_SCORE_MFMA'] = score_mfma             # read when an execution plan packs its score networks
    if args.no_winograd:
        os.environ['VPHO_WINOGRAD'] = '0'
    if args.device_prior:
        os.environ['VPHO_DEVICE_PRIOR'] = '1'
    conv_mfma = os.environ.get('VPHO_CONV_MFMA', 'f32') if args.conv_mfma == 'f32' else args.conv_mfma
    os.environ['VPHO_CONV_MFMA'] = conv_mfma
    model._engine = Engine(model)
    eng = model._engine
    # share of the stride-4 FPN pixels the RoIAligns of a batch can read (= what the two smoothing convolutions compute)
    roi_frac = []
    for b in batches:
        fh, fw = b['rgb'].shape[2] // 4, b['rgb'].shape[3] // 4
        wh = ops.roi_windows(b['bbox_hand'].float().contiguous(), b['bbox_hand_rect'].float().contiguous(), lbs, fh, fw, 0.25)
        wo = ops.roi_windows(b['bbox_obj_rect'].float().contiguous(), None, lbs, fh, fw, 0.25)
        roi_frac.append((int(wh.count) / (lbs * fh * fw), int(wo.count) / (lbs * fh * fw)))
    g = torch.Generator().manual_seed(1234 + rank)
    gt_pose = (torch.randn(lbs, 48, generator=g) * 0.2).to(dev)
    gt_ctx = eng.mano.shape((torch.randn(lbs, 10, generator=g) * 0.5).to(dev))
    gt_vert, gt_joint = eng.mano.fk(gt_pose, gt_ctx, 1, True)
    gt_vert = gt_vert + batches[0]['root_joint'][:, None]
    gt_joint = gt_joint + batches[0]['root_joint'][:, None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):                           # warm-up runs the complete step (incl. the metric rows)
        out = model(batches[i % 2], mode='predict')
        E.metric_rows(out, batches[i % 2], gt_joint, gt_vert, 0, assets)
    if args.warmup:
        E.gather_rows(E.metric_rows(out, batches[0], gt_joint, gt_vert, 0, assets))
        if args.pipeline > 1:
            torch.cuda.synchronize()
    barrier()
    step_events = []

    pipe = E.PipelinedPredictor(model, args.pipeline) if args.pipeline > 1 else None
    if pipe is not None and args.warmup:
        # every slot of the pipelined evaluator has its own execution plan (packed weights, captured HIP graphs, sampler
        # workspaces): the untimed warm-up has to go through each of them as well
        for f in [pipe.submit(batches[i % 2], lambda out, batch, engine: E.metric_rows(out, batch, gt_joint, gt_vert, 0, assets))
                  for i in range(max(args.warmup, 2 * args.pipeline))]:
            f.result()
        barrier()

    def run_steps(k, pipelined=True):
        rows, nfev = [], []
        if pipe is not None and pipelined:
            def post(out, batch, engine, i=0):
                return None
            futs = []
            for i in range(k):
                first = (rank * k + i) * lbs
                futs.append(pipe.submit(batches[i % 2], lambda out, batch, engine, first=first: (
                    E.metric_rows(out, batch, gt_joint, gt_vert, first, assets),
                    (engine.last_info['hand_ode']['nfev'], engine.last_info['obj_ode']['nfev']))))
            for f in futs:
                r, n = f.result()
                rows.append(r)
                nfev.append(n)
        else:
            for i in range(k):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                step_events.append(ev)
                out = model(batches[i % 2], mode='predict')
                rows.append(E.metric_rows(out, batches[i % 2], gt_joint, gt_vert, (rank * k + i) * lbs, assets))
                nfev.append((eng.last_info['hand_ode']['nfev'], eng.last_info['obj_ode']['nfev']))
        return E.gather_rows(torch.cat(rows, 0)), nfev     # the ONE collective of the evaluation

    # ---- timed region: K steps, no instrumentation -----------------------------------------------------------------
    def thread_cpu():
        """{native thread id: user + system CPU seconds} of this process (where the host's busy threads are: VERDICT r5, Host)"""
        try:
            import psutil
            return {t.id: t.user_time + t.system_time for t in psutil.Process().threads()}
        except Exception:
            return {}
    thr0 = thread_cpu()
    cpu0 = sum(os.times()[:2])
    t0 = time.perf_counter()
    all_rows, nfev = run_steps(args.steps)
    ev_end = torch.cuda.Event(enable_timing=True)
    ev_end.record()
    barrier()
    dt = time.perf_counter() - t0
    host_cpu_s = sum(os.times()[:2]) - cpu0                 # user+system CPU seconds of this process over the timed region
    thr1 = thread_cpu()
    import threading
    names = {t.native_id: t.name for t in threading.enumerate()}
    by_thread = sorted(((names.get(i, 'native thread (HIP runtime / OpenMP / RCCL)'), thr1[i] - thr0.get(i, 0.0)) for i in thr1), key=lambda kv: -kv[1])
    if pipe is not None:                                   # per-step latencies: a short sequential run after the timed region
        step_events.clear()
        run_steps(min(args.steps, 5), pipelined=False)
        ev_end = torch.cuda.Event(enable_timing=True)
        ev_end.record()
        torch.cuda.synchronize()
    ev_all = step_events + [ev_end]
    step_ms_seq = [a.elapsed_time(b) for a, b in zip(ev_all[:-1], ev_all[1:])]
    step_ms = sorted(step_ms_seq)
    if os.environ.get('VPHO_BENCH_VERBOSE') and rank == 0:
        print('per-step ms:', ' '.join(f'{t:.1f}' for t in step_ms_seq), file=sys.stderr)
    # ---- roofline leg: the same K steps again with HIP events recorded around every launch of the timed kernel
    # classes on their launch streams (kept out of the timed region: ~500 event pairs per step perturb it by 10-15 %)
    hbm_classes = ('mano_fk', 'obj_physics', 'hand_fuse', 'roi_align', 'resize_bilinear')
    timed_classes = ('conv_igemm_128x128', 'conv_igemm_128x64', 'conv_igemm_64x64', 'score_head', 'conv_winograd', 'pose_encoder') + hbm_classes
    prof = {c: dict(total_ms=0.0, launches=0, flops=0.0, bytes=0.0) for c in timed_classes}
    if not args.no_kernel_timing:
        for c in timed_classes:
            ops.prof_enable(c, True)
        graphs_were = eng.use_graphs
        eng.use_graphs = False                             # events cannot be recorded inside a graph replay: same kernels, plain launches
        run_steps(args.steps, pipelined=False)
        barrier()
        for c in timed_classes:
            ops.prof_enable(c, False)
            prof[c] = ops.prof_collect(c)
        # the score head once more with the two samplers one after the other: in the pass above the hand and the object solve
        # run concurrently (as in the timed region), so an event pair around a head launch also spans the other solve's kernels
        ops.prof_enable('score_head', True)
        ops.prof_enable('pose_encoder', True)
        eng.serial_samplers = True
        run_steps(min(args.steps, 5), pipelined=False)
        barrier()
        eng.serial_samplers = False
        ops.prof_enable('score_head', False)
        ops.prof_enable('pose_encoder', False)
        head_excl = ops.prof_collect('score_head')
        pe_excl = ops.prof_collect('pose_encoder')
        eng.use_graphs = graphs_were
    conv, head = prof['conv_igemm_128x128'], prof['score_head']
    if args.no_kernel_timing:
        head_excl = dict(total_ms=0.0, launches=0, flops=0.0, bytes=0.0)
        pe_excl = dict(total_ms=0.0, launches=0, flops=0.0, bytes=0.0)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    rank_dt = [dt]
    if world > 1:
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)                       # the per-rank clocks, so that a first multi-GPU run can check itself
        rank_dt = [float(t.item()) for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert all_rows.shape[0] == world * args.steps * lbs
    fabric = {'world_size_reported': dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
              'backend': dist.get_backend() if (dist.is_available() and dist.is_initialized()) else 'none',
              'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, 'nccl') else None,
              'rank_ms_per_step_min_max': [min(rank_dt) / args.steps * 1e3, max(rank_dt) / args.steps * 1e3]}

    result = None
    wl_key = workload_key(args.bs, args.sample_num, args.sampling_steps)
    if rank == 0:
        images = world * args.steps * lbs                  # strong: world * lbs = --bs per step
        host_cpu = {'cpu_seconds_per_step': host_cpu_s / args.steps, 'busy_threads_equivalent': host_cpu_s / dt,
                    # the eight busiest threads over the timed region, CPU ms per step (psutil; Python threads by name, the rest are the runtime's)
                    'cpu_ms_per_step_by_thread': [[n_, 1e3 * c_ / args.steps] for n_, c_ in by_thread[:8] if c_ > 0]}
        conv_tf = conv['flops'] / (conv['total_ms'] * 1e-3) / 1e12 if conv['total_ms'] > 0 else 0.0
        head_tf = head['flops'] / (head['total_ms'] * 1e-3) / 1e12 if head['total_ms'] > 0 else 0.0
        head_excl_tf = head_excl['flops'] / (head_excl['total_ms'] * 1e-3) / 1e12 if head_excl['total_ms'] > 0 else 0.0
        # fp32-equivalent peak of the score head: the fp32 MFMA peak, or the dense bf16 peak over the 6 / 9 products per fp32 product
        head_peak = FP32_MFMA_PEAK_TFLOPS if score_mfma == 'f32' else BF16_MFMA_PEAK_TFLOPS / int(score_mfma[-1])
        conv_peak = FP32_MFMA_PEAK_TFLOPS if conv_mfma == 'f32' else BF16_MFMA_PEAK_TFLOPS / int(conv_mfma[-1])
        result = {
            # BASELINE.json's metric string at the default arguments; another --bs / --sample_num / --sampling_steps names itself
            'metric': f'eval images/sec (bs={args.bs}, sample_num={args.sample_num}, steps={args.sampling_steps}); MPJPE delta vs ref',
            'value': images / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'step_ms_min_median_max': [step_ms[0], step_ms[len(step_ms) // 2], step_ms[-1]], 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32' if (score_mfma, conv_mfma) == ('f32', 'f32') else f'f32 storage and accumulation; products as split-bf16 (score head {score_mfma}, convolutions {conv_mfma}; opt-in, NOT the default)',
            'data': 'synthetic',
            'config': {'workload': 'vpho_net.forward(mode=predict), ' + ('README eval config (BASELINE.json configs[1])' if
                                    (args.bs, args.sample_num, args.sampling_steps, args.topk_hand, args.topk_obj) == (64, 100, 50, 30, 10)
                                    else 'non-default config (see the keys below)'),
                       'per_gpu_batch': lbs, 'global_batch_per_step': world * lbs,
                       'batch_coupling': 'CrossModule attention over the batch axis (Q3) and the one RK45 controller per solve (Q5) act on each rank\'s LOCAL batch, as per DDP rank in the reference',
                       'sample_num': args.sample_num, 'sampling_steps': args.sampling_steps,
                       'topk_hand': args.topk_hand, 'topk_obj': args.topk_obj, 'sample_T0': args.sample_T0, 'crop': '256x256',
                       'pipeline_depth': args.pipeline, 'score_mfma': score_mfma, 'conv_mfma': conv_mfma, 'winograd_3x3': bool(eng.winograd),
                       'fpn_roi_window': {'enabled': bool(eng.roi_window), 'what': 'the last convolution of each FPN branch is computed only on the pixels its '
                                          'RoIAligns read (VPHO.py:126-129 are the maps\' only readers); bit-identical results, --no_roi_window computes the full maps',
                                          'pixel_share_hand_obj_per_batch': roi_frac,
                                          'boxes': 'synth_batch: hand / object half-extent = focal * 0.11 / depth x U(0.75,1.15) / U(0.6,1.1), unchanged since round 1'},
                       'weights': ('vpho_amd.synth.bench_state_dict(seed=1): seeded, heat-map contrast 0.7, conditioned score networks' if args.weights == 'conditioned' else 'vpho_amd.synth.synth_state_dict(seed=1): round-1 random set') + '; synthetic MANO/YCB tables', 'parallelism': f'dp{world}',
                       'nfev_hand_obj_per_step': nfev[-1],
                       'prior_draw': 'device generator (Philox), opt-in: NOT the reference RNG stream' if eng.device_prior else 'CPU generator inside the timed step (sde.py:26-28)'},
            'roofline': {'bound': 'mfma', 'kernel': CONV_CLASS_NAME if conv_mfma == 'f32'
                         else f'conv_igemm_split_kernel<128,128,4,2,{conv_mfma[-1]}> (split-bf16 products, opt-in)', 'achieved': conv_tf,
                         'peak': conv_peak, 'unit': 'TFLOP/s', 'frac': conv_tf / conv_peak,
                         # launch-weighted over BOTH kernels of the class (one-tile and persistent multi-tile), and only from a counter
                         # pass taken at this batch size / sample count
                         'traffic': pmc_traffic(CONV_CLASS_KERNELS, workload=wl_key),
                         'traffic_source': (pmc_traffic(CONV_CLASS_KERNELS, per_kernel=True, workload=wl_key) or {}).get('source'),
                         'timing': 'HIP events around every launch, in a separate instrumented repeat of the K steps',
                         'algorithmic_bytes_per_launch': conv['bytes'] / max(conv['launches'], 1),
                         'pose_encoder': ({'TFLOP/s': prof['pose_encoder']['flops'] / (prof['pose_encoder']['total_ms'] * 1e-3) / 1e12, 'frac': prof['pose_encoder']['flops'] / (prof['pose_encoder']['total_ms'] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                           'avg_launch_us': prof['pose_encoder']['total_ms'] * 1e3 / max(prof['pose_encoder']['launches'], 1), 'launches_per_step': prof['pose_encoder']['launches'] / max(args.steps, 1),
                                           'note': 'durations span the other solve\'s kernels, like score_head',
                                           'samplers_serialised': ({'TFLOP/s': pe_excl['flops'] / (pe_excl['total_ms'] * 1e-3) / 1e12, 'frac': pe_excl['flops'] / (pe_excl['total_ms'] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                                                    'avg_launch_us': pe_excl['total_ms'] * 1e3 / max(pe_excl['launches'], 1),
                                                                    'what': 'same kernels, object solve after the hand solve: exclusive durations (flops as executed: the object state padded to 32 inputs)'}
                                                                   if pe_excl['total_ms'] > 0 else None)} if prof.get('pose_encoder', {}).get('total_ms', 0) > 0 else None),
                         'other_kernels': {k: {'TFLOP/s': (v['flops'] / (v['total_ms'] * 1e-3) / 1e12 if v['total_ms'] > 0 else 0.0),
                                               'kernel_ms_per_step': v['total_ms'] / max(args.steps, 1),
                                               'launches_per_step': v['launches'] / max(args.steps, 1)}
                                           for k, v in prof.items() if k in ('conv_igemm_128x64', 'conv_igemm_64x64', 'conv_winograd')},
                         'launches_per_step': conv['launches'] / max(args.steps, 1),
                         'avg_launch_us': conv['total_ms'] * 1e3 / max(conv['launches'], 1),
                         'flop_per_launch_avg': conv['flops'] / max(conv['launches'], 1),
                         'kernel_ms_per_step': conv['total_ms'] / max(args.steps, 1),
                         'score_head': {'achieved': head_tf, 'frac': head_tf / head_peak, 'peak': head_peak,
                                        'kernel_ms_per_step': head['total_ms'] / max(args.steps, 1),
                                        'launches_per_step': head['launches'] / max(args.steps, 1),
                                        'note': 'hand and object solves run concurrently: these durations span the other solve\'s kernels',
                                        'samplers_serialised': {'achieved': head_excl_tf, 'frac': head_excl_tf / head_peak,
                                                                'avg_launch_us': head_excl['total_ms'] * 1e3 / max(head_excl['launches'], 1),
                                                                'what': 'same kernels, object solve after the hand solve: exclusive durations (hand 32 heads + object 3 heads, averaged over launches by time)'}},
                         # the convolutions of the step as ONE family, in ALGORITHMIC flops (2 x pixels x Cout x taps x Cin of the direct form,
                         # whatever algorithm ran): the three direct tile classes + the Winograd launches, whose 2.25 x fewer multiply-adds
                         # make their algorithmic rate exceed the executed one
                         'conv_family_algorithmic': (lambda fam, ms: {
                             'achieved': fam / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, 'frac': (fam / (ms * 1e-3) / 1e12 / conv_peak) if ms > 0 else 0.0,
                             'kernel_ms_per_step': ms / max(args.steps, 1), 'unit': 'TFLOP/s',
                             'what': 'direct-form flops of every convolution launch / summed kernel time: conv_igemm 128x128 + 128x64 + 64x64 (executed = algorithmic) '
                                     '+ conv_winograd (algorithmic = 2.25 x executed)'})(
                             sum(prof[k]['flops'] for k in ('conv_igemm_128x128', 'conv_igemm_128x64', 'conv_igemm_64x64')) + 2.25 * prof['conv_winograd']['flops'],
                             sum(prof[k]['total_ms'] for k in ('conv_igemm_128x128', 'conv_igemm_128x64', 'conv_igemm_64x64', 'conv_winograd'))),
                         'note': 'kernel = the direct-convolution tile class with the most kernel time.  Since round 3 its most efficient layers (the 3x3 / stride-1 '
                                 'convolutions, 113-135 TFLOP/s) run as conv_winograd launches instead, so the class average is lower than in round 2 although no '
                                 'launch got slower; by summed exclusive time the score head (score_head below, samplers_serialised) is the largest single kernel of the step',
                         'feature_path_gflop_per_image_ref': FEATURE_GFLOP_PER_IMAGE},
            # HBM-bound kernels of the path (north star: MANO skinning, distance kernels, top-k as GB/s against the chip's HBM peak):
            # achieved = ALGORITHMIC bytes (operands read once + results written once, stated at the launch site) / HIP-event kernel time
            'hbm': {'peak_GBps': HBM_PEAK_GBPS, 'note': 'achieved = algorithmic bytes / kernel time (HIP events, instrumented repeat of the K steps); '
                                                          'traffic = HBM bytes per launch from the newest committed rocprofv3 --pmc summary (2 x FETCH_SIZE + WRITE_SIZE; traffic_by_kernel names the file and every kernel of the class, traffic is their launch-weighted mean), null when absent',
                    'kernels': {k: {'GB/s': (prof[k]['bytes'] / (prof[k]['total_ms'] * 1e-3) / 1e9 if prof[k]['total_ms'] > 0 else 0.0),
                                    'frac': (prof[k]['bytes'] / (prof[k]['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBPS if prof[k]['total_ms'] > 0 else 0.0),
                                    'avg_launch_us': prof[k]['total_ms'] * 1e3 / max(prof[k]['launches'], 1),
                                    'launches_per_step': prof[k]['launches'] / max(args.steps, 1),
                                    'kernel_ms_per_step': prof[k]['total_ms'] / max(args.steps, 1),
                                    'algorithmic_bytes_per_launch': prof[k]['bytes'] / max(prof[k]['launches'], 1),
                                    'traffic': pmc_traffic(HBM_KERNEL_NAMES[k], workload=wl_key), 'traffic_by_kernel': pmc_traffic(HBM_KERNEL_NAMES[k], per_kernel=True, workload=wl_key),
                                    'limited_by': HBM_LIMITS[k]} for k in hbm_classes}},
            'metrics_rows_gathered': int(all_rows.shape[0]),
            'host_cpu': host_cpu,
            'fabric': fabric,
        }
        post_rows = lambda out, batch, engine: E.metric_rows(out, batch, gt_joint, gt_vert, 0, assets)
        if world == 1 and not args.no_opt_in and (score_mfma, conv_mfma) == ('f32', 'f32'):
            result['opt_in'] = {'split_bf16x6': secondary_leg(args, model, batches, E, post_rows, {'VPHO_SCORE_MFMA': 'bf16x6', 'VPHO_CONV_MFMA': 'bf16x6'},
                'same step, same evaluator; fp32 products of the score head and of the direct convolutions as 6 exact bf16 products each '
                '(--score_mfma bf16x6 --conv_mfma bf16x6), fp32 storage and accumulation; opt-in, not `value`'),
                               # all nine cross products: every fp32 product exact (nothing dropped), fp32 accumulation -- VERDICT r5 item 10: reported beside x6
                               'split_bf16x9': secondary_leg(args, model, batches, E, post_rows, {'VPHO_SCORE_MFMA': 'bf16x9', 'VPHO_CONV_MFMA': 'bf16x9'},
                'same step with all nine bf16 products per fp32 product (--score_mfma bf16x9 --conv_mfma bf16x9): no term dropped; opt-in, not `value`')}
        if world == 1 and not args.no_opt_in and eng.roi_window and not args.no_roi_window:
            # the RoI-window saving is data dependent (boxes that span the crop lose it): the same step with the full stride-4 maps
            result['value_full_maps'] = secondary_leg(args, model, batches, E, post_rows, {'VPHO_ROI_WINDOW': '0'},
                'same step with --no_roi_window: the last convolution of each FPN branch on every pixel (what boxes spanning the crop would cost)')
        if world == 1 and not args.no_cpu_baseline:
            result.update(cpu_baseline_leg(args, cfg, model, sd, assets, ANCHOR_SKELETON, dev))
        emit(result)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


DETAIL_PATH = os.path.join('gpurun_out', 'bench_detail.json')
LINE_LIMIT = 6144                  # bytes; the driver's parser read 19.5 KB in round 4 and not 22.3 KB in round 5 -- stay far away


def emit(full, stream=None):
    """The full record (every diagnostic block) goes to a side file and, prefixed, to stderr; the LAST stdout line is the compact result."""
    stream = stream or sys.stdout
    line = compact_line(full)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, DETAIL_PATH), 'w') as f:
            json.dump(full, f, indent=1)
        line['detail'] = DETAIL_PATH
    except OSError:
        line['detail'] = None
    text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    assert len(text) < LINE_LIMIT and '\n' not in text, len(text)
    sys.stdout.flush()
    print(text, file=stream, flush=True)
    return text


def _r(x, sig=5):
    """numbers of the result line at `sig` significant digits (the side file keeps them in full); None for non-finite values"""
    if isinstance(x, str):
        return x if len(x) <= 240 else x[:237] + '...'       # no free-text field may grow the line
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float('inf'), float('-inf')):
            return None
        return float(f'{x:.{sig}g}')
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    return x


def compact_line(full):
    """The one JSON line the driver parses: the contract keys, `roofline`, `cpu_baseline` and a short numeric `parity` summary --
    a pure function of the full record (tests/test_bench_line_cpu.py holds it under LINE_LIMIT)."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if (isinstance(d, dict) and len(ks) > 1) else (d.get(ks[0]) if isinstance(d, dict) else None))
    cfgf, rf = full.get('config', {}), full.get('roofline', {})
    line = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                     'vs_baseline', 'dtype', 'data')}
    line['config'] = {k: cfgf.get(k) for k in ('workload', 'per_gpu_batch', 'global_batch_per_step', 'sample_num', 'sampling_steps', 'topk_hand',
                                               'topk_obj', 'sample_T0', 'crop', 'pipeline_depth', 'parallelism', 'score_mfma', 'conv_mfma',
                                               'winograd_3x3', 'nfev_hand_obj_per_step')}
    line['config']['roi_window_pixel_share'] = g(cfgf, 'fpn_roi_window', 'pixel_share_hand_obj_per_batch') if g(cfgf, 'fpn_roi_window', 'enabled') else None
    ser = g(rf, 'score_head', 'samplers_serialised') or {}
    line['roofline'] = {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                               'algorithmic_bytes_per_launch', 'flop_per_launch_avg', 'avg_launch_us', 'launches_per_step', 'kernel_ms_per_step')}
    line['roofline']['timing'] = 'HIP events per launch on the launch stream, instrumented repeat of the K steps'
    # the largest kernel BY TIME is the score head; `frac` there is from exclusive durations (the object solve after the hand solve)
    line['roofline']['dominant_by_time'] = {'kernel': 'score_head_kernel<true>', 'frac': ser.get('frac'), 'achieved': ser.get('achieved'),
                                            'avg_launch_us': ser.get('avg_launch_us'), 'kernel_ms_per_step': g(rf, 'score_head', 'kernel_ms_per_step'),
                                            'launches_per_step': g(rf, 'score_head', 'launches_per_step'), 'frac_overlapped': g(rf, 'score_head', 'frac')}
    line['roofline']['classes_TFLOPs_ms'] = {k: [v.get('TFLOP/s'), v.get('kernel_ms_per_step')] for k, v in (rf.get('other_kernels') or {}).items()}
    pe = g(rf, 'pose_encoder', 'samplers_serialised') or {}
    line['roofline']['classes_TFLOPs_ms']['pose_encoder'] = [pe.get('TFLOP/s'), None]
    line['roofline']['conv_family_algorithmic_frac'] = g(rf, 'conv_family_algorithmic', 'frac')
    hb = g(full, 'hbm', 'kernels') or {}
    line['hbm_GBps_frac'] = {k: [v.get('GB/s'), v.get('frac')] for k, v in hb.items()}
    if full.get('cpu_baseline') is not None:
        line['cpu_baseline'] = full['cpu_baseline']
    par = full.get('parity')
    if par is not None:
        e2e, same = par.get('end_to_end_vs_oracle') or {}, par.get('aggregation_given_identical_candidates') or {}
        within = 'images_within_1e-3_on_joints_vertices_6dof'
        selfv = g(par, 'reference_self_agreement', 'variants') or {}
        s64, f64j = par.get('sampler_vs_fp64') or {}, par.get('fp64_judge') or {}
        line['parity'] = {
            'images': e2e.get('images'), 'nfev_equal': par.get('nfev_equal'),
            'upstream_max_abs': max((par.get('upstream_max_abs') or {'-': None}).values(), key=lambda v: -1.0 if v is None else v),
            'e2e_lists_identical': e2e.get('images_all_selections_identical'), 'e2e_within_1e-3': e2e.get(within),
            'e2e_gap_above_tie_bound': e2e.get('images_with_gap_above_tie_bound'),
            'reference_vs_itself_lists_identical': [v.get('images_all_selections_identical') for v in selfv.values()],
            'reference_vs_itself_within_1e-3': [v.get(within) for v in selfv.values()],
            'same_candidates_lists_identical': same.get('images_all_selections_identical'), 'same_candidates_within_1e-3': same.get(within),
            'same_candidates_gap_above_tie_bound': same.get('images_with_gap_above_tie_bound'),
            'same_candidates_max_rel_gap': same.get('max_rel_score_gap_at_first_differences'),
            'same_candidates_gaps_all_within_2eps32_in_fp64': (all(g_['within_2_eps32'] for g_ in par['identical_candidates_gaps_above_tie_bound'])
                                                               if par.get('identical_candidates_gaps_above_tie_bound') is not None else None),
            'mpjpe_delta_mm': e2e.get('mpjpe_delta_mm_all'),
            'fp64_referee_lists_within_reference_noise': g(par, 'fp64_referee', 'all_within_reference_noise'),
            'fp64_referee_images_identical_hip_ref': [g(par, 'fp64_referee', 'images_identical_to_fp64_order'), g(par, 'fp64_referee', 'images_identical_to_fp64_order_fp32_reference')],
            'sampler_err_ratio_hip_over_ref_max': [g(s64, 'hand', 'ratio_max'), g(s64, 'obj', 'ratio_max')],
            'fp64_judge_within_1e-3_hip_ref': [f64j.get('images_within_1e-3_hip'), f64j.get('images_within_1e-3_oracle')] if f64j else None,
            'fp64_judge_lists_identical_hip_ref': [g(f64j, 'images_lists_identical_to_fp64', 'hip'), g(f64j, 'images_lists_identical_to_fp64', 'oracle')] if f64j else None,
        }
    line['value_full_maps'] = g(full, 'value_full_maps', 'value')
    line['opt_in_split_bf16x6_value'] = g(full, 'opt_in', 'split_bf16x6', 'value')
    line['opt_in_split_bf16x9_value'] = g(full, 'opt_in', 'split_bf16x9', 'value')
    line['step_ms_min_median_max'] = full.get('step_ms_min_median_max')
    line['host_busy_threads'] = g(full, 'host_cpu', 'busy_threads_equivalent')
    line['metrics_rows_gathered'] = full.get('metrics_rows_gathered')
    line['fabric'] = full.get('fabric')
    line = _r(line)
    line['value'], line['ms_per_step'] = full.get('value'), full.get('ms_per_step')       # the contract's two numbers in full: value = images / (steps x ms_per_step)
    return line


def secondary_leg(args, model, batches, E, post, env, what):
    """A secondary measurement, clearly separate from `value`: the same step through a fresh pipelined evaluator whose execution plans
    are built under the environment switches `env` (read when a plan is built).  Used for the opt-in split-bf16 products (every fp32
    product of the score head and of the direct convolutions from six exact bf16 x bf16 products, fp32 storage and accumulation:
    DESIGN 4b, tests/test_gpu_split_head.py) and for the full-map FPN (`--no_roi_window`).  NOT the headline."""
    import torch
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        pipe = E.PipelinedPredictor(model, max(args.pipeline, 1))
        for f in [pipe.submit(batches[i % 2], post) for i in range(2 * max(args.pipeline, 1) + 1)]:
            f.result()
        torch.cuda.synchronize()
        k = max(4, min(args.steps, 12))
        t0 = time.perf_counter()
        for f in [pipe.submit(batches[i % 2], post) for i in range(k)]:
            f.result()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pipe.close()
    finally:
        for key, v in saved.items():
            if v is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = v
    return {'value': k * args.bs / dt, 'unit': 'images/s', 'ms_per_step': dt / k * 1e3, 'steps': k, 'what': what}


PMC_ROUNDS = ('r06', 'r05', 'r04', 'r03', 'r02', 'r01')
# counter passes are per workload: the README config (BASELINE cfg2) and the stress config (cfg4) each have their own file
PMC_FILES = {'cfg2': '{rnd}_pmc_hbm_traffic.json', 'cfg4': '{rnd}_pmc_hbm_traffic_cfg4.json'}


def workload_key(bs, sample_num, sampling_steps):
    return {(64, 100, 50): 'cfg2', (128, 256, 100): 'cfg4'}.get((bs, sample_num, sampling_steps))


def pmc_traffic(kernel, per_kernel=False, workload='cfg2', root=None):
    """HBM bytes per launch of the kernels whose name starts with `kernel` (a name or a tuple of names), LAUNCH-WEIGHTED over them (a
    profiling class covers several kernels: the one-tile and the persistent convolution, the matrix-core FK kernel and the packed-FMA
    ones), from the newest committed rocprofv3 --pmc summary OF THIS WORKLOAD (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE;
    profiles/rNN_pmc_hbm_traffic[_cfg4].json by scripts/pmc_summary.py); None when there is no pass for the workload.
    per_kernel: {'source': file, 'kernels': {name: bytes per launch, launches}} instead."""
    if workload not in PMC_FILES:
        return None
    names = (kernel,) if isinstance(kernel, str) else tuple(kernel)
    for rnd in PMC_ROUNDS:
        path = os.path.join(root or ROOT, 'profiles', PMC_FILES[workload].format(rnd=rnd))
        try:
            with open(path) as f:
                tab = json.load(f)
        except Exception:
            continue
        hit = {name: v for name, v in tab.items() if any(name.startswith(k) or k in name.split('(')[0] for k in names)}
        if not hit:
            continue
        if per_kernel:
            return {'source': os.path.relpath(path, root or ROOT), 'kernels': {n: {'hbm_bytes_per_launch': v['hbm_bytes_per_launch'], 'launches_in_pass': v['launches']} for n, v in hit.items()}}
        n = sum(v['launches'] for v in hit.values())
        return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hit.values()) / max(n, 1)
    return None


def cpu_baseline_leg(args, cfg, model, sd, assets, skeleton, dev):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload -- ONE batch of `cpu_images` images
    (default 64 = the per-GPU batch, so the batch-coupled quirks Q3/Q5 see the benchmark's own batch size) -- and the parity of
    the HIP path against it on identical inputs and identical prior draws, reported selection by selection (oracle/compare.py)."""
    import torch
    from oracle import vpho as OV
    from oracle.compare import parity_summary, TIE_REL, E2E_TIE_REL
    from oracle.aggregation import hoi_aggregate
    from vpho_amd.synth import synth_batch
    n = args.cpu_images
    data = synth_batch(n, assets, seed=777)
    torch.manual_seed(99)
    nh, no = torch.randn(n * args.sample_num, 96), torch.randn(n * args.sample_num, 9)
    from vpho_amd.hostcpu import usable_cpus
    threads_before = torch.get_num_threads()
    cores = min(threads_before, usable_cpus())             # the CPUs this process may really use (affinity AND cgroup quota)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    ref, info = OV.predict(sd, assets, skeleton, data, sample_num=args.sample_num, sample_T0=args.sample_T0,
                           sampling_steps=args.sampling_steps, topk_hand=args.topk_hand, topk_obj=args.topk_obj,
                           noise_hand=nh, noise_obj=no)
    t_cpu = time.perf_counter() - t0
    torch.set_num_threads(threads_before)
    gdata = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
    model._engine.keep_states = True                        # the candidates and score vectors of every selection stage, for the referee
    out = model._engine.predict(gdata, noise_hand=nh, noise_obj=no)
    torch.cuda.synchronize()
    eng_info = model._engine.last_info
    model._engine.keep_states = False
    mx = lambda a, b: float((a.double().cpu() - b.double()).abs().max())
    upstream = {k: mx(out[k], ref[k]) for k in ('hand_heatmap', 'obj_heatmap', 'force_local', 'reg_hand_joint',
                                                'diff_final_hand_mano', 'diff_final_obj_6d')}
    end_to_end, _ = parity_summary(out, ref, eng_info['agg'], info['agg'], args.sample_num, bound=E2E_TIE_REL)
    # second comparison: the oracle's aggregation fed the HIP path's OWN candidates, heat-maps and forces -- isolates the
    # aggregation kernels (identical inputs on both sides; what remains is the fp32 rounding of FK / projection / bicubic sums)
    gf = eng_info['features']
    c = lambda t: t.detach().cpu()
    fl = c(out['diff_final_hand_mano']).reshape(-1, 58)
    same = hoi_aggregate(assets, skeleton, cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                         root_joint=data['root_joint'], is_right=data['is_right'], force_local=c(gf['force_local']),
                         is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(), hand_pose_regression=c(gf['mano_pose']),
                         hand_shape=fl[:, 48:], hand_heatmap=c(gf['hand_heatmap']), hand_bbox=data['bbox_hand'],
                         hand_topk=args.topk_hand, obj_pose6d=c(out['diff_final_obj_6d']), obj_heatmap=c(gf['obj_heatmap']),
                         obj_bbox=data['bbox_obj_rect'], obj_topk=args.topk_obj, obj_name=data['obj_name'])
    same_out = dict(agg_hand_joint=same['hand_agg_joint'], agg_hand_vert=same['hand_agg_vert'], agg_hand_mano=same['hand_agg_mano'],
                    agg_obj_6d=same['obj_agg_6d'])
    given_same, _ = parity_summary(out, same_out, eng_info['agg'], same['dbg'], args.sample_num, bound=TIE_REL)
    # the judge of the top-k chain: every list of the HIP path re-scored in fp64 on the HIP path's own candidates (oracle/referee.py)
    from oracle import referee as RFE
    rf = RFE.referee(assets, skeleton, RFE.record_from_hip(out, eng_info, data))
    ref_sum = RFE.summary(rf)
    # every first difference on identical candidates whose fp32 score gap is above TIE_REL, judged in float64: the distance of the exchanged
    # candidates' fp64 scores against the rounding noise eps32 of the reference's own arithmetic on that very score vector -- within
    # 2 eps32 both picks are picks the reference's arithmetic could have made (any top-k of scores within eps of the truth)
    from oracle import judge_fp64 as JF
    rsame = JF._report(eng_info['agg'], same['dbg'], args.sample_num)
    fsame = JF.first_flip(rsame)
    gaps_above = []
    for b in (rsame['primary_gap_per_image'] > TIE_REL).nonzero().reshape(-1).tolist():
        st = JF.STAGES[int(fsame[b])]
        gb, eb = rf[st]['exchange_gap_bf'][b], rf[st]['eps32_bf'][b]
        fi = int(gb.argmax())
        gaps_above.append({'image': b, 'stage': st, 'rel_gap_of_the_fp32_scores': float(rsame['primary_gap_per_image'][b]), 'fp64_gap_of_the_exchanged_candidates_rel': float(gb[fi]),
                           'eps32_of_that_score_vector_rel': float(eb[fi]), 'within_2_eps32': bool(gb[fi] <= 2 * eb[fi])})
    # the sampler's own referee: both ODEs solved in fp64 on the accepted step sequence (rows are independent on a fixed sequence: every
    # 4th hypothesis), each side against the exact solution of the same scheme on its own encoding (oracle/sampler_fp64.py)
    from oracle import sampler_fp64 as SF
    from oracle import nets as ON
    S_ = args.sample_num
    sig = ON.ve_prior_sigma(args.sample_T0)
    rep = lambda e: e.detach().cpu()[:, None].repeat(1, S_, 1).reshape(-1, 1024)
    sampler64 = {}
    for name, key, noise, x_hip, x_or in (('hand', 'denoiser_hand', nh, eng_info['hand_x6d'], info['hand_x6d']),
                                          ('obj', 'denoiser_obj', no, out['diff_final_obj_6d'].reshape(-1, 9), ref['diff_final_obj_6d'].reshape(-1, 9))):
        sampler64[name] = SF.compare(sd, key, rep(info['features'][f'encoding_{name}']), noise * sig, info[f'{name}_ode']['steps'], args.sampling_steps,
                                     x_hip, x_or, feat_hip=rep(gf[f'encoding_{name}']), steps_hip=eng_info[f'{name}_ode']['steps'], stride=4)
    # the feature path's own referee: the oracle's feature path in FLOAT64 on the first 8 images (heat-maps and encodings are per-image
    # quantities: eval-mode BatchNorm, no batch coupling upstream of the cross modules), against which the HIP kernels and the oracle's fp32
    # arithmetic are both held -- "close to the oracle" is not "close to the truth"
    n64 = min(8, n)
    dbl = lambda d_: {k: (v[:n64].double() if (torch.is_tensor(v) and v.is_floating_point()) else (v[:n64] if (torch.is_tensor(v) or isinstance(v, list)) else v)) for k, v in d_.items()}
    f64 = OV.features({k: (v.double() if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in sd.items()}, assets, dbl(data))
    feat64 = {}
    for k in ('hand_heatmap', 'obj_heatmap', 'encoding_hand', 'encoding_obj'):
        eh = (gf[k][:n64].detach().cpu().double() - f64[k]).abs()
        eo = (info['features'][k][:n64].double() - f64[k]).abs()
        feat64[k] = {'err_hip_max': float(eh.max()), 'err_oracle_max': float(eo.max()), 'err_hip_rms': float(eh.pow(2).mean().sqrt()),
                     'err_oracle_rms': float(eo.pow(2).mean().sqrt()), 'scale': float(f64[k].abs().max())}
    # the end-to-end float64 judge (oracle/judge_fp64.py): ONE predict in double on either side's accepted step sequences -- which fp32 side is
    # within 1e-3 of what the algorithm computes in exact arithmetic, image by image
    judge64 = None
    if not args.no_fp64_judge:
        from oracle import judge_fp64 as J64
        kw64 = dict(sample_num=args.sample_num, sample_T0=args.sample_T0, sampling_steps=args.sampling_steps, topk_hand=args.topk_hand, topk_obj=args.topk_obj,
                    noise_hand=nh, noise_obj=no)
        tj = time.perf_counter()
        f64j = J64.features64(sd, assets, data)
        o64h, d64h = J64.predict_fp64(sd, assets, skeleton, data, steps_hand=eng_info['hand_ode']['steps'], steps_obj=eng_info['obj_ode']['steps'], feat64=f64j, **kw64)
        o64o, d64o = J64.predict_fp64(sd, assets, skeleton, data, steps_hand=info['hand_ode']['steps'], steps_obj=info['obj_ode']['steps'], feat64=f64j, **kw64)
        judge64 = J64.judge({k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in out.items()}, eng_info['agg'], ref, info['agg'], o64h, d64h, o64o, d64o, args.sample_num)
        judge64['images_within_1e-3_hip'], judge64['images_within_1e-3_oracle'] = judge64['images_within_1e3_of_fp64']['hip'], judge64['images_within_1e3_of_fp64']['oracle']
        judge64['seconds'] = time.perf_counter() - tj
    # how well the REFERENCE reproduces itself (committed fixture written by the reference's own forward under other thread counts /
    # oneDNN off, tests/golden/make_golden_readme.py --variant): the yardstick for end_to_end_vs_oracle's list counts
    from oracle.compare import reference_self_agreement
    selfcheck = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'golden_predict_readme64_selfcheck.npz')
    self_rep = reference_self_agreement(selfcheck) if os.path.exists(selfcheck) else None
    return {'cpu_baseline': {'value': n / t_cpu, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
                             'sample': f'one batch of {n} images at the same config (S={args.sample_num}, steps={args.sampling_steps}), oracle '
                                       f'(torch-CPU + host RK45, {cores} threads = this process\'s CPU quota), {t_cpu:.1f} s; nfev hand/obj {info["hand_ode"]["nfev"]}/{info["obj_ode"]["nfev"]}'},
            'parity': {'sample': f'{n} images in one batch, identical inputs and prior draws; bar: 1e-3 on joints / vertices / 6-DoF, selected '
                                 'indices equal.  A top-k chain is discontinuous, so parity = (everything upstream of the aggregation agrees: '
                                 'upstream_max_abs) x (every selection list, judged on the HIP path\'s own candidates by an fp64 evaluation of '
                                 'the stage score, has a regret within twice the rounding noise of the reference\'s fp32 arithmetic on those '
                                 'candidates: fp64_referee -- regret_max_rel / images_identical_to_fp64_order for the HIP lists and, beside them, '
                                 'for the fp32 oracle\'s lists on the same candidates); aggregation_given_identical_candidates and '
                                 'end_to_end_vs_oracle report list-by-list equality with the oracle (tie_bound there is a reported number, '
                                 'not a criterion)',
                       'nfev_equal': [eng_info['hand_ode']['nfev'] == info['hand_ode']['nfev'],
                                      eng_info['obj_ode']['nfev'] == info['obj_ode']['nfev']],
                       'upstream_max_abs': upstream,
                       'fp64_referee': ref_sum,
                       'features_vs_fp64': dict(feat64, what=f'heat-maps and encodings of the first {n64} images against the oracle\'s feature path run in float64: err_hip = HIP kernels '
                                                '(fp32 MFMA, one accumulation chain per output, Winograd for the 3x3 / stride-1 layers), err_oracle = the reference arithmetic (torch CPU fp32, blocked sums)'),
                       'sampler_vs_fp64': dict(sampler64, what='final hypotheses of both ODE solves against a float64 solve (score network, stage algebra, denoise step '
                                               'in double) of its OWN accepted step sequence (the controller turns 1e-7 of the stages into 2e-5 of the next step size: step_size_rel_diff_max), every 4th hypothesis, each side on its own encoding: err_* = max / rms '
                                               '|x - x_fp64|, ratio = HIP / oracle (<= 1: the kernels are at least as close to the exact scheme as the reference\'s fp32 arithmetic)'),
                       'reference_self_agreement': self_rep,
                       'end_to_end_vs_oracle': end_to_end,
                       'aggregation_given_identical_candidates': given_same,
                       'identical_candidates_gaps_above_tie_bound': gaps_above,
                       'fp64_judge': judge64}}


if __name__ == '__main__':
    main()
