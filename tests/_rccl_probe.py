"""Child process of tests/test_gpu_rccl.py: executes every RCCL call site of the package on a 'nccl' process group.

Started BEFORE anything touches the GPU with WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* in the environment and VPHO_FORCE_NCCL=1, under which
``launch.init_process_group`` builds the 'nccl' group also at world size 1 and ``gather_rows`` / ``GradBuckets`` / ``allreduce_mean_scale``
take their collective branches instead of returning early (launch.group_active).  A one-rank 'nccl' group is a real RCCL communicator: the
library is loaded, the device bound (device_id=), the collectives run on RCCL's stream.
Counterpart in the reference: accelerate's process group (lib/engine/base_trainer.py:22), the prepared loaders and the metric gather
(lib/engine/train_diff_hand_obj.py:121-124, 333-335), DDP's bucketed gradient all-reduce (:180).

Modes: ``one``  -- world 1: metric-row gather (incl. an empty shard), ScoreTrainer step and one full DiffusionTrainStep.step with the bucketed
                   all-reduces, each compared BIT FOR BIT with the same step made before the group existed; prints one JSON line.
       ``dup``  -- every rank on cuda:0 over 'nccl' (what does RCCL say to two ranks on one GPU?): reports, never asserts.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _train_fixture(dev, bs=4):
    import torch
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict, synth_batch
    from vpho_amd.trainer import synthetic_mano_targets
    from vpho_amd.train_step import DiffusionTrainStep
    assets = synthetic_assets(0)
    sd = synth_state_dict(vpho_net(assets), seed=1)

    def make():
        step = DiffusionTrainStep(sd, dev, assets=assets)
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth_batch(bs, assets, seed=11).items()}
        g = torch.Generator().manual_seed(100)
        data['hm_hand'] = (torch.rand(bs, 21, 64, 64, generator=g) * 0.2).to(dev)
        data['hm_obj'] = (torch.rand(bs, 27, 64, 64, generator=g) * 0.2).to(dev)
        gt_h = (torch.randn(bs, 96, generator=g) * 0.5).to(dev) + torch.tensor([1., 0, 0, 0, 1, 0], device=dev).repeat(16)
        gt_o = (torch.randn(bs, 9, generator=g) * 0.5).to(dev)
        data.update(synthetic_mano_targets(step.mano_head.mano, gt_h, (torch.randn(bs, 10, generator=g) * 0.5).to(dev), data['is_right']))
        data['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(dev)
        reps = 3
        draws = dict(t_h=torch.rand(reps, bs, generator=g).to(dev) * 0.99 + 0.01, z_h=torch.randn(reps, bs, 96, generator=g).to(dev),
                     t_o=torch.rand(reps, bs, generator=g).to(dev) * 0.99 + 0.01, z_o=torch.randn(reps, bs, 9, generator=g).to(dev))
        return step, data, gt_h, gt_o, draws

    return sd, make


def _full_step(make, n=2):
    import torch
    torch.manual_seed(1234)                                  # the cross modules' dropout masks come from the device generator
    torch.cuda.manual_seed(1234)
    step, data, gt_h, gt_o, draws = make()
    for _ in range(n):
        losses = step.step(data, gt_h, gt_o, draws=draws, repeat_num=3)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in step.state_dict().items()}, {k: float(v) for k, v in losses.items()}


def _score_step(sd, dev, n=3):
    import torch
    from vpho_amd.train_score import ScoreTrainer
    tr = ScoreTrainer(sd, 'denoiser_obj', dev)
    g = torch.Generator().manual_seed(5)
    feat, gt = (torch.randn(8, 1024, generator=g) * 0.3).to(dev), (torch.randn(8, 9, generator=g) * 0.5).to(dev)
    ts, zs = (torch.rand(4, 8, generator=g) * 0.99 + 0.01).to(dev), torch.randn(4, 8, 9, generator=g).to(dev)
    for _ in range(n):
        tr.step(feat, gt, ts=ts, zs=zs)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in tr.state_dict().items()}


def mode_one():
    import torch
    import torch.distributed as dist
    from vpho_amd import launch
    from vpho_amd.evaluate import gather_rows, ROW
    assert launch.force_group() and int(os.environ['WORLD_SIZE']) == 1
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    res = {}
    # --- before the group exists: the no-group steps (their collective call sites return early)
    assert not launch.group_active()
    sd, make = _train_fixture(dev)
    ref_full, ref_losses = _full_step(make)
    ref_score = _score_step(sd, dev)
    # --- the RCCL communicator
    backend = launch.init_process_group(dev)
    assert backend == 'nccl' and dist.get_backend() == 'nccl' and dist.get_world_size() == 1 and launch.group_active()
    res['backend'] = backend
    res['rccl_version'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
    # --- the metric-row gather (evaluate.py; train_diff_hand_obj.py:333-335): rows out = rows in, also for a rank without rows
    rows = torch.arange(5 * ROW, device=dev, dtype=torch.float32).view(5, ROW)
    out = gather_rows(rows)
    assert out.data_ptr() != rows.data_ptr() and torch.equal(out, rows), 'gather_rows over RCCL changed the rows'
    empty = gather_rows(torch.zeros((0, ROW), device=dev))
    assert tuple(empty.shape) == (0, ROW)
    res['gather_rows'] = 'rows out == rows in (5 rows; 0 rows)'
    # --- other collectives the entry points use: barrier, all_gather of the replica checksums (train.py)
    dist.barrier()
    chk = torch.ones(1, device=dev, dtype=torch.float64) * 3.25
    got = [torch.zeros_like(chk)]
    dist.all_gather(got, chk)
    assert float(got[0]) == 3.25
    # --- the bucketed gradient exchange under the backward (grad_buckets.py): asynchronous all-reduces on RCCL's stream
    from vpho_amd import grad_buckets
    calls = {'n': 0}
    orig = dist.all_reduce

    def counting(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)

    dist.all_reduce = counting
    try:
        got_full, got_losses = _full_step(make)
        n_bucket_calls = calls['n']
        got_score = _score_step(sd, dev)
    finally:
        dist.all_reduce = orig
    assert n_bucket_calls == 2 * len(grad_buckets.BUCKETS), n_bucket_calls         # two steps, one all-reduce per milestone
    assert calls['n'] == n_bucket_calls + 3, calls                                  # three ScoreTrainer steps, one flat all-reduce each
    bad = [k for k in ref_full if not torch.equal(ref_full[k], got_full[k])]
    assert not bad, ('full training step differs with the RCCL exchange', bad[:5], len(bad))
    assert ref_losses == got_losses, (ref_losses, got_losses)
    bad = [k for k in ref_score if not torch.equal(ref_score[k], got_score[k])]
    assert not bad, ('score-network step differs with the RCCL all-reduce', bad)
    res['train_step'] = f'{len(ref_full)} tensors bit-identical to the no-group step after 2 steps, {n_bucket_calls} bucket all-reduces'
    res['score_step'] = f'{len(ref_score)} tensors bit-identical after 3 steps'
    dist.destroy_process_group()
    assert not dist.is_initialized()
    res['destroyed'] = True
    print(json.dumps(res), flush=True)


def mode_dup():
    """two ranks, both on cuda:0, backend 'nccl' -- RCCL is expected to refuse a duplicate GPU; whatever happens is printed"""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    res = {'rank': rank, 'world': world}
    try:
        dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=60))
        t = torch.ones(4, device=dev) * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        res['all_reduce'] = t.tolist()
        dist.destroy_process_group()
        res['outcome'] = 'ran'
    except Exception as e:                                   # noqa: BLE001 -- the point is to record what RCCL says
        res['outcome'] = 'refused'
        res['error'] = f'{type(e).__name__}: {str(e)[:600]}'
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    {'one': mode_one, 'dup': mode_dup}[sys.argv[1]]()
