"""Evaluation loop of the hot path: per-rank batches, per-image metric rows, ONE fixed-layout all-gather.

Counterpart of the reference's ``Trainer.evaluate`` / ``postprocess`` (lib/engine/train_diff_hand_obj.py:202-357,578-602)
for synthetic batches.  The reference gathers pickled nested dicts with ``gather_for_metrics(use_gather_object=True)``
(:333-335); here every rank fills a ``(n_images, ROW)`` fp32 tensor on its GPU and the ranks exchange it with a single
``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI with backend 'nccl', gloo on CPU for tests).
The forward pass itself needs no collective: the image batch is the only sharded dimension.
"""
import torch
import torch.distributed as dist

# row layout: [global image index, MJE(regression), MJE(first hypothesis), MJE(aggregated), MVE(aggregated),
#              |agg - regression| mean joint distance (mm), object translation norm (m), is_right,
#              PA-MJE(regression), PA-MJE(aggregated), PA-MVE(aggregated), 0,
#              then the 16 object metrics of the aggregated pose in the order of ops.OBJ_METRIC_NAMES (TesterObject,
#              test.py:240-503: MCE, OCE, MCE2, ADD, ADD-S, ADD<0.1d, ADD-S<0.1d, REP, REP<5px, CD, F-score x6)]
ROW = 28
OBJ_COL = 12


def mje_mm(pd, gt):
    """TesterHand MJE (test.py:657-668): mean over joints of the L2 distance, millimetres."""
    return (pd - gt).norm(dim=-1).mean(dim=-1) * 1000.0


def postprocess(out, root_joint, is_right):
    """train_diff_hand_obj.py:578-602: un-flip left hands along x and add the root joint (hand outputs only)."""
    sgn = torch.where(is_right.bool(), 1.0, -1.0).to(out['agg_hand_joint'].dtype)[:, None, None]
    res = {}
    for k in ('reg_hand_joint', 'agg_hand_joint', 'reg_hand_vert', 'agg_hand_vert'):
        v = out[k].clone()
        v[..., 0] = v[..., 0] * sgn[..., 0]
        res[k] = v + root_joint[:, None]
    first = out['diff_final_hand_joint'][:, 0].clone()
    first[..., 0] = first[..., 0] * sgn[..., 0]
    res['first_hand_joint'] = first + root_joint[:, None]
    return res


_OBJ_METRICS = {}


def object_metric_block(out, data, assets):
    """(bs,16) fp64: TesterObject on the aggregated object pose ('mean_candidate_pose', train_diff_hand_obj.py:249-258,
    498-501) -- obj_9D_to_mat + root joint, then the device metric kernels.  Needs data['gt_obj_rt'], data['cam_intr']."""
    from . import ops
    dev = out['agg_obj_6d'].device
    key = (id(assets), str(dev))
    if key not in _OBJ_METRICS:
        _OBJ_METRICS[key] = ops.ObjectMetrics(assets['ycb'], dev)
    M = _OBJ_METRICS[key]
    pd_rt = ops.obj_9d_to_rt(out['agg_obj_6d'].double().contiguous(), data['root_joint'].float().contiguous())
    return M(pd_rt, data['gt_obj_rt'].double().contiguous(), data['cam_intr'].double().contiguous(), M.obj_ids(data['obj_name']))


def metric_rows(out, data, gt_joint, gt_vert, first_index, assets=None):
    """(bs, ROW) fp32 on the model's device."""
    pp = postprocess(out, data['root_joint'], data['is_right'])
    bs = gt_joint.shape[0]
    rows = torch.empty((bs, ROW), device=gt_joint.device, dtype=torch.float32)
    if torch.is_tensor(first_index):                 # per-image ids (a loader batch that is not a run of the data set)
        rows[:, 0] = first_index.to(device=rows.device, dtype=torch.float32).reshape(bs)
    else:
        rows[:, 0] = torch.arange(first_index, first_index + bs, device=rows.device, dtype=torch.float32)
    rows[:, 1] = mje_mm(pp['reg_hand_joint'], gt_joint)
    rows[:, 2] = mje_mm(pp['first_hand_joint'], gt_joint)
    rows[:, 3] = mje_mm(pp['agg_hand_joint'], gt_joint)
    rows[:, 4] = mje_mm(pp['agg_hand_vert'], gt_vert)
    rows[:, 5] = mje_mm(pp['agg_hand_joint'], pp['reg_hand_joint'])
    rows[:, 6] = out['agg_obj_6d'][:, 6:].float().norm(dim=-1)
    rows[:, 7] = data['is_right'].float()
    rows[:, 8:] = 0.0
    if gt_joint.is_cuda:                 # Procrustes-aligned metrics by the HIP kernel (test.py:657-680 on the device)
        from . import ops
        c = lambda t: t.float().contiguous()
        rows[:, 8] = ops.hand_metrics(c(pp['reg_hand_joint']), c(gt_joint))[1] * 1000.0
        rows[:, 9] = ops.hand_metrics(c(pp['agg_hand_joint']), c(gt_joint))[1] * 1000.0
        rows[:, 10] = ops.hand_metrics(c(pp['agg_hand_vert']), c(gt_vert))[1] * 1000.0
        if assets is not None and 'gt_obj_rt' in data:
            rows[:, OBJ_COL:OBJ_COL + 16] = object_metric_block(out, data, assets).float()
    return rows


class _Pending:
    """Future of one pipelined batch: the worker's future + the HIP event behind the batch's last kernel"""

    def __init__(self, fut):
        self._fut = fut

    def result(self, timeout=None):
        res, done = self._fut.result(timeout)
        done.synchronize()
        return res

    def done(self):
        return self._fut.done() and self._fut.result()[1].query()


class PipelinedPredictor:
    """Evaluation batches are independent, so `depth` of them are kept in flight: each on its own HIP stream, driven by its own
    host thread and execution plan (packed weights are per plan).  One batch alone leaves the GPU idle at the sampler's
    per-attempt host syncs and at the tails of its small launches; a second batch fills those gaps.
    The CPU prior draws are made by the submitting thread, in submission order (hand then object per batch), so a seeded
    run draws exactly what the sequential loop of the reference would (sde.py:26-28)."""

    def __init__(self, model, depth=3):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from .model.engine import Engine
        self.model, self.depth = model, depth
        self.engines = [Engine(model) for _ in range(depth)]
        self.dev = self.engines[0].dev
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(depth)]
        self.locks = [threading.Lock() for _ in range(depth)]
        # ONE worker thread and queue per slot: with a shared pool a thread that has finished its slot's batch takes the next task in line,
        # which may belong to ANOTHER slot, and sleeps on that slot's lock while its own slot sits idle
        self.pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f'vpho-predict-{i}') for i in range(depth)]
        self.n = 0
        import os
        self.sync_in_worker = os.environ.get('VPHO_PIPE_SYNC_IN_WORKER', '0') == '1'      # A/B aid: the round-1 behaviour

    def submit(self, batch, post=None):
        """Returns a future of post(out, batch, engine) (or of the output dict).  The worker thread only ENQUEUES the batch and
        moves on to its next one; ``.result()`` waits (in the caller's thread, sleeping) for the event recorded behind the batch's
        last kernel, so the result can be consumed from any stream."""
        from .configs.args import cfg
        bs = batch['rgb'].shape[0]
        noise_h = torch.randn(bs * cfg.sample_num, 96)
        noise_o = torch.randn(bs * cfg.sample_num, 9)
        slot = self.n % self.depth
        self.n += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.dev))

        def work():
            with self.locks[slot], torch.cuda.device(self.dev), torch.cuda.stream(self.streams[slot]), torch.no_grad():
                self.streams[slot].wait_event(ready)
                eng = self.engines[slot]
                out = eng.predict(batch, noise_hand=noise_h, noise_obj=noise_o)
                res = post(out, batch, eng) if post is not None else out
                done = torch.cuda.Event(blocking=True)      # sleep, do not spin: the slot threads share the rank's CPU quota
                done.record(self.streams[slot])
                if self.sync_in_worker:
                    done.synchronize()
                return res, done

        return _Pending(self.pools[slot].submit(work))

    def close(self):
        for p in self.pools:
            p.shutdown(wait=True)


def gather_rows(rows):
    """All ranks' rows, concatenated in rank order.  Ranks may hold different numbers of rows (a data loader's ragged last batches, like
    accelerate's ``gather_for_metrics``, train_diff_hand_obj.py:333-335): the counts travel first (one all-gather of a single integer per
    rank), the rows padded to the largest count, the padding dropped again.  No-op without a process group."""
    from .launch import group_active
    if not group_active():
        return rows
    world = dist.get_world_size()
    try:
        host_stage = dist.get_backend() == 'gloo' and rows.is_cuda       # CPU rehearsal backend: stage through host memory
        dev = torch.device('cpu') if host_stage else rows.device
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev))
        counts = counts.tolist()
        cap = max(counts)
        mine = torch.zeros((cap, rows.shape[1]), device=dev, dtype=rows.dtype)
        mine[:rows.shape[0]] = rows.to(dev)
        out = torch.empty((world * cap, rows.shape[1]), device=dev, dtype=rows.dtype)
        dist.all_gather_into_tensor(out, mine)
        if rows.is_cuda and not host_stage:
            torch.cuda.current_stream(rows.device).synchronize()          # an RCCL failure surfaces HERE, with the context below
        out = torch.cat([out[r * cap:r * cap + counts[r]] for r in range(world)], 0) if any(c != cap for c in counts) else out
        return out.to(rows.device)
    except Exception as e:
        raise RuntimeError(f'gather_rows: the all-gather of the metric rows failed on rank {dist.get_rank()} of {world} (backend '
                           f'{dist.get_backend()}, {tuple(rows.shape)} rows on {rows.device}): {type(e).__name__}: {e}') from e


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for strong-scaling splits of one global batch."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def summarize(rows):
    """Table like train_diff_hand_obj.py:336-357 for right / left / both hands."""
    res = {}
    for name, mask in (('right', rows[:, 7] > 0.5), ('left', rows[:, 7] < 0.5), ('both', torch.ones_like(rows[:, 7], dtype=torch.bool))):
        sel = rows[mask]
        if sel.shape[0] == 0:
            continue
        res[name] = dict(n=int(sel.shape[0]), MJE_reg=float(sel[:, 1].mean()), MJE_first=float(sel[:, 2].mean()),
                         MJE_agg=float(sel[:, 3].mean()), MVE_agg=float(sel[:, 4].mean()), PA_MJE_reg=float(sel[:, 8].mean()),
                         PA_MJE_agg=float(sel[:, 9].mean()), PA_MVE_agg=float(sel[:, 10].mean()))
    # object table (test.py:521-584 'average_instance' column: distances in mm, hit rates / F-scores in percent, REP in pixels)
    from .ops_names import OBJ_METRIC_NAMES
    obj = rows[:, OBJ_COL:OBJ_COL + 16].double().mean(0)
    res['object'] = {k: float(obj[i] * (1000.0 if k in ('MCE', 'OCE', 'MCE2', 'ADD', 'ADDS', 'CD') else (1.0 if k == 'REP' else 100.0)))
                     for i, k in enumerate(OBJ_METRIC_NAMES)}
    return res
