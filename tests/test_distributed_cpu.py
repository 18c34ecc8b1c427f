"""N>1 path on CPU: world_size-2 gloo process group, the single all-gather of fixed-layout metric rows, sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vpho_amd import evaluate as E
    n = 5
    rows = torch.zeros((n, E.ROW))
    rows[:, 0] = torch.arange(rank * n, rank * n + n)
    rows[:, 3] = 10.0 * (rank + 1)
    rows[:, 7] = rank % 2
    g = E.gather_rows(rows)
    q.put((rank, g.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    from vpho_amd import evaluate as E
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        g = got[r]
        assert g.shape == (10, E.ROW)
        assert torch.equal(g[:, 0], torch.arange(10, dtype=torch.float32))      # rank order, every image exactly once
        assert torch.equal(g[:5, 3], torch.full((5,), 10.0)) and torch.equal(g[5:, 3], torch.full((5,), 20.0))
    s = E.summarize(got[0])
    assert s['both']['n'] == 10 and s['right']['n'] == 5 and s['left']['n'] == 5
    assert s['both']['MJE_agg'] == pytest.approx(15.0)


def _worker_ragged(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vpho_amd import evaluate as E
    n = 5 if rank == 0 else 3                             # a data loader's ragged last batch on one rank
    rows = torch.zeros((n, E.ROW))
    rows[:, 0] = torch.arange(n) + 100 * rank
    q.put((rank, E.gather_rows(rows).clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_takes_ragged_counts_like_gather_for_metrics():
    """ranks with different numbers of images (train_diff_hand_obj.py:333-335 gathers pickled lists of any length): counts first,
    rows padded to the largest count, padding dropped -- every image exactly once, rank order"""
    from vpho_amd import evaluate as E
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert got[r].shape == (8, E.ROW)
        assert got[r][:, 0].tolist() == [0, 1, 2, 3, 4, 100, 101, 102]


def test_gather_rows_without_process_group_is_identity():
    from vpho_amd import evaluate as E
    rows = torch.randn(3, E.ROW)
    assert E.gather_rows(rows) is rows


@pytest.mark.parametrize('n,world', [(64, 8), (64, 3), (5, 8), (0, 4)])
def test_shard_range_partitions_exactly(n, world):
    from vpho_amd.evaluate import shard_range
    seen = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        assert 0 <= lo <= hi <= n
        seen += list(range(lo, hi))
    assert seen == list(range(n))


def test_metric_rows_and_postprocess_cpu():
    """postprocess un-flips left hands and adds the root (train_diff_hand_obj.py:578-602); MJE in millimetres."""
    from vpho_amd import evaluate as E
    bs = 4
    g = torch.Generator().manual_seed(0)
    out = dict(reg_hand_joint=torch.randn(bs, 21, 3, generator=g) * 0.05, agg_hand_joint=torch.randn(bs, 21, 3, generator=g) * 0.05,
               reg_hand_vert=torch.randn(bs, 778, 3, generator=g) * 0.05, agg_hand_vert=torch.randn(bs, 778, 3, generator=g) * 0.05,
               diff_final_hand_joint=torch.randn(bs, 6, 21, 3, generator=g) * 0.05, agg_obj_6d=torch.randn(bs, 9, generator=g).double())
    data = dict(root_joint=torch.randn(bs, 3, generator=g), is_right=torch.tensor([True, False, True, False]))
    pp = E.postprocess(out, data['root_joint'], data['is_right'])
    exp = out['agg_hand_joint'].clone()
    exp[1, :, 0] *= -1
    exp[3, :, 0] *= -1
    assert torch.allclose(pp['agg_hand_joint'], exp + data['root_joint'][:, None])
    rows = E.metric_rows(out, data, pp['agg_hand_joint'], pp['agg_hand_vert'], first_index=8)
    assert rows.shape == (bs, E.ROW)
    assert torch.equal(rows[:, 0], torch.tensor([8., 9., 10., 11.]))
    assert torch.allclose(rows[:, 3], torch.zeros(bs), atol=1e-4) and torch.allclose(rows[:, 4], torch.zeros(bs), atol=1e-4)
    assert torch.all(rows[:, 1] > 1.0)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.argv = sys.argv[:1]
    from oracle import train_score as OT
    from vpho_amd.assets import synthetic_assets
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict
    from vpho_amd.train_score import allreduce_mean_scale, SUFFIXES
    sd = synth_state_dict(vpho_net(synthetic_assets(0)), seed=1)
    p, D, bs, reps = 'denoiser_obj', 9, 4, 2
    g = torch.Generator().manual_seed(7)                                   # every rank draws the GLOBAL batch, then takes its shard
    feat, gt = torch.randn(world * bs, 1024, generator=g) * 0.3, torch.randn(world * bs, D, generator=g) * 0.5
    ts = torch.rand(reps, world * bs, generator=g) * (1 - 1e-5) + 1e-5
    zs = torch.randn(reps, world * bs, D, generator=g)
    sl = slice(rank * bs, (rank + 1) * bs)
    _, grads, _ = OT.loss_and_grads(sd, p, feat[sl], gt[sl], ts[:, sl, None], zs[:, sl])
    flat = torch.cat([grads[s].reshape(-1) for s in SUFFIXES])
    scale = allreduce_mean_scale(flat)                                     # what ScoreTrainer.step does with its flat buffer
    mean = flat * scale
    _, full, _ = OT.loss_and_grads(sd, p, feat, gt, ts[:, :, None], zs)     # the same step as ONE batch of world*bs images
    want = torch.cat([full[s].reshape(-1) for s in SUFFIXES])
    q.put((rank, scale, float((mean - want).abs().max()), float(want.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_score_training_gradient_average_world2_gloo():
    """Data-parallel DSM step: per-rank gradients, one SUM all-reduce of the flat buffer and the 1/world factor equal the
    gradient of the global batch (the loss is a mean over images) -- the DDP semantics of the reference's accelerate setup."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for rank, scale, err, mag in got:
        assert scale == 0.5
        assert err <= 2e-6 * mag + 1e-9, (rank, err, mag)


# ---------------------------------------------------------------------------------------------------------------------------------
# bucketed gradient exchange of the training step (vpho_amd/grad_buckets.py), world 2 on gloo
_SHAPES = {'denoiser_hand.head.head.0.weight': (4, 7, 5), 'head_mano.fc_pose.bias': (6,), 'cross_obj.gravity_proj.weight': (3, 5),
           'head_physics.fc_CoM.2.bias': (3,), 'encoder_hand.reg.0.conv1.weight': (4, 4, 1, 1), 'head_hm_hand.final_layer.bias': (5,),
           'encoder_obj.project.weight': (4, 6, 1, 1), 'head_hm_obj.deconv_layers.0.weight': (3, 2, 4, 4),
           'feature_extractor.smooth3_h.weight': (2, 2, 3, 3), 'feature_extractor.latlayer1_o.bias': (2,),
           'feature_extractor.layer4_h.0.0.conv1.weight': (8, 4, 1, 1), 'feature_extractor.layer2_o.0.1.bn2.weight': (4,),
           'feature_extractor.layer1_h.0.0.conv2.weight': (2, 2, 3, 3), 'feature_extractor.layer0_h.0.weight': (2, 3, 7, 7),
           'feature_extractor.layer0_h.1.bias': (2,)}
_NEVER = 'head_hm_obj.deconv_layers.0.weight'            # no loss reaches it on this "batch": keeps a zero gradient


def _grad(name, rank, step):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(f'{name}|{rank}|{step}'.encode()))       # not hash(): str hashes differ per process
    return torch.randn(_SHAPES[name], generator=g)


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vpho_amd.grad_buckets import GradBuckets, BUCKETS, bucket_of
    B = GradBuckets(_SHAPES, 'cpu')
    params = {k: torch.zeros(s) for k, s in _SHAPES.items()}
    trace = []
    for step in range(3):
        B.begin()
        for b in BUCKETS:                                 # the backward's milestone order; the last milestone is left to finish()
            B.put({k: _grad(k, rank, step) for k in _SHAPES if bucket_of(k) == b and k != _NEVER})
            if b != 'fpn_end':
                B.flush(b)
        scale = B.finish()
        for k in B.names:
            params[k] -= 0.1 * scale * B.view[k]
        trace.append(B.flat.clone())
    # numpy payloads: torch tensors travel through the queue as shared-memory handles that die with this process
    q.put((rank, {k: v.numpy().copy() for k, v in params.items()}, [t.numpy().copy() for t in trace], B.names, dict(B.range)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_exchange_world2_gloo():
    """three steps on two ranks with different gradients: every bucket's asynchronous all-reduce delivers the SAME sums to both
    ranks (replicas bit-identical), equal to the sum of the two ranks' gradients; a tensor no loss reached stays at zero; the flat
    buffer is laid out in the backward's milestone order"""
    from vpho_amd.grad_buckets import BUCKETS, bucket_of
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, params, trace, names, rng = q.get(timeout=180)
        got[r] = ({k: torch.from_numpy(v) for k, v in params.items()}, [torch.from_numpy(t) for t in trace], names, rng)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (p0, t0, names, rng), (p1, t1, _, _) = got[0], got[1]
    order = [bucket_of(k) for k in names]
    assert order == sorted(order, key=BUCKETS.index)                       # buckets are contiguous and in milestone order
    assert [b for b in BUCKETS if b in rng] == list(BUCKETS) and all(rng[a][1] == rng[b][0] for a, b in zip(BUCKETS[:-1], BUCKETS[1:]))
    for step in range(3):
        assert torch.equal(t0[step], t1[step])                              # identical reduced gradients on both ranks
    for k in _SHAPES:
        assert torch.equal(p0[k], p1[k]), k                                 # replicas stay bit-identical
        want = torch.zeros(_SHAPES[k])
        for step in range(3):
            if k != _NEVER:
                want -= 0.1 * 0.5 * (_grad(k, 0, step) + _grad(k, 1, step))
        assert torch.allclose(p0[k], want, atol=1e-6), k
    assert float(p0[_NEVER].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ vpho_net.forward('train') under DDP
class _Toy(torch.nn.Module):
    """stand-in with the structure of vpho_net._forward_train: the step computes loss and gradients itself and returns the loss
    through vpho_amd.model.VPHO._DepositGrads"""
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.zeros(3))
        self.b = torch.nn.Parameter(torch.zeros(2, 2))
        self.unreached = torch.nn.Parameter(torch.zeros(4))          # no loss of the batch reaches it: explicit zero gradient

    def forward(self, scale):
        from vpho_amd.model.VPHO import _DepositGrads
        grads = {'a': torch.full((3,), float(scale)), 'b': torch.full((4,), 2.0 * float(scale))}          # flat, like the step's buffers
        named = [(k, p) for k, p in self.named_parameters() if p.requires_grad]
        return _DepositGrads.apply(torch.tensor(10.0 * float(scale)), [grads.get(k) for k, _ in named], *[p for _, p in named])


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel as DDP
    m = DDP(_Toy())
    out = []
    for it in range(2):                                              # two iterations: DDP's "finished reduction" bookkeeping holds
        m.zero_grad()
        loss = m(rank + 1.0)
        (loss * 3.0).backward()                                      # the incoming gradient scales the deposited ones
        out.append({k: p.grad.tolist() for k, p in m.module.named_parameters()})
    with m.no_sync():                                                # gradient accumulation: local gradients, added to .grad
        m(rank + 1.0).backward()
    out.append({k: p.grad.tolist() for k, p in m.module.named_parameters()})
    q.put((rank, float(loss.detach()), out))
    dist.barrier()
    dist.destroy_process_group()


def test_train_forward_loss_node_takes_part_in_ddp_gradient_averaging():
    """the loss returned by forward(mode='train') must let DistributedDataParallel average the gradients --
    parameters are real inputs of the autograd node, so the per-parameter hooks fire; rank gradients 1 and 2 average to 1.5"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (l, o) for r, l, o in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        loss, outs = got[r]
        assert loss == 10.0 * (r + 1)
        for it in range(2):
            assert outs[it]['a'] == [4.5] * 3 and outs[it]['b'] == [[9.0, 9.0], [9.0, 9.0]] and outs[it]['unreached'] == [0.0] * 4
        # no_sync: the averaged gradient of the second iteration + this rank's own
        assert outs[2]['a'] == [4.5 + (r + 1.0)] * 3
