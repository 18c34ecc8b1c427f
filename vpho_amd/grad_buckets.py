"""Bucketed data-parallel gradient exchange, overlapped with the backward pass.

The reference trains under HF accelerate = ``DistributedDataParallel`` (lib/engine/train_diff_hand_obj.py:121-124,180): gradients are
all-reduced bucket by bucket while autograd is still running.  The HIP training step has no autograd, but its analytic backward
finishes the modules in a fixed order -- score networks / head_mano / physics branch first, then per branch encoder + heat-map
head, then the FPN heads, the residual layers 4-2 and finally layer1 + stem -- so the flat gradient buffer is laid out in THAT
order, cut into one bucket per milestone, and each bucket's all-reduce (RCCL over xGMI under backend 'nccl', asynchronous on
RCCL's own stream) is issued the moment the milestone is reached.  The 125 MB of layers 4-2 travel while layer1 and the stem (the
64 x 64-resolution, most expensive part of the backward) are still computing; only the last, 0.9 MB bucket is exposed.
Every rank receives the same sums, so replicas stay bit-identical.  gloo on device tensors (the CPU rehearsal backend) stages a
bucket through host memory and is synchronous.
"""
import torch
import torch.distributed as dist

# milestone order of DiffusionTrainStep.loss_and_grads
BUCKETS = ('heads', 'branch_hand', 'branch_obj', 'fpn_top', 'fpn_mid', 'fpn_end')


def bucket_of(name):
    """bucket (milestone) at which the gradient of the reference parameter `name` is final"""
    if name.startswith(('denoiser_', 'head_mano.', 'head_physics.', 'cross_')):
        return 'heads'
    if name.startswith(('encoder_hand.', 'head_hm_hand.')):
        return 'branch_hand'
    if name.startswith(('encoder_obj.', 'head_hm_obj.')):
        return 'branch_obj'
    if name.startswith('feature_extractor.'):
        rest = name[len('feature_extractor.'):]
        if rest.startswith(('smooth', 'latlayer', 'toplayer')):
            return 'fpn_top'
        if rest.startswith(('layer4', 'layer3', 'layer2')):
            return 'fpn_mid'
        return 'fpn_end'                                  # layer1_h, layer0_h (stem)
    raise KeyError(name)


class GradBuckets:
    def __init__(self, shapes, device, dtype=torch.float32, conv_meta=None):
        """shapes: {reference parameter name: shape of its slot}.  The flat buffer holds the tensors grouped by bucket (names sorted
        inside).  conv_meta {name: (cout, cin, kh, kw)}: these slots hold the gradient in the kernels' PACKED layout
        (cout, kh*kw*cin_pad) -- the layout the weight-gradient kernels write and AdamW steps the packed weights in, so that neither
        the hand-over nor the optimiser needs a permuting copy per tensor; an element-wise all-reduce does not care."""
        self.conv_meta = dict(conv_meta or {})
        order = {b: i for i, b in enumerate(BUCKETS)}
        self.names = sorted(shapes, key=lambda k: (order[bucket_of(k)], k))
        sizes = [int(torch.Size(shapes[k]).numel()) for k in self.names]
        self.flat = torch.zeros(sum(sizes), device=device, dtype=dtype)
        self.view, self.range, off = {}, {}, 0
        for k, n in zip(self.names, sizes):
            self.view[k] = self.flat[off:off + n].view(shapes[k])
            b = bucket_of(k)
            lo, hi = self.range.get(b, (off, off))
            self.range[b] = (min(lo, off), off + n)
            off += n
        self._work, self._flushed = [], set()

    # ---- one step -------------------------------------------------------------------------------------------------------
    def begin(self):
        """tensors no loss of this batch reaches keep a zero gradient"""
        self.flat.zero_()
        self._work, self._flushed = [], set()

    def put(self, grads):
        """copy a group of finished gradients into their slots of the flat buffer.  torch._foreach_copy_ is ONE multi-tensor launch
        only while every tensor of the list is contiguous (a single strided source sends the whole list down the per-tensor path:
        569 copy launches per step, measured) -- convolution gradients therefore travel in the packed layout their kernels produced
        (`packed_grad`, attached by train_blocks._unpack_grad to the reference-layout view it returns)."""
        if not grads:
            return
        dst, src = [], []
        for k, g in grads.items():
            slot = self.view[k]
            pk = getattr(g, 'packed_grad', None)
            if pk is not None and pk.shape == slot.shape:
                g = pk
            elif g.shape != slot.shape and k in self.conv_meta:       # a reference-layout gradient from elsewhere: one strided copy
                cout, cin, kh, kw = self.conv_meta[k]
                slot.view(cout, kh, kw, -1)[..., :cin].copy_(g.reshape(cout, cin, kh, kw).permute(0, 2, 3, 1))
                continue
            else:
                g = g.reshape(slot.shape)
            dst.append(slot)
            src.append(g if g.is_contiguous() else g.contiguous())
        if dst:
            torch._foreach_copy_(dst, src)

    def flush(self, bucket):
        """the gradients of `bucket` are final on this rank: start its all-reduce (sum) without waiting for it"""
        assert bucket not in self._flushed, bucket
        self._flushed.add(bucket)
        from .launch import group_active
        if bucket not in self.range or not group_active():
            return
        lo, hi = self.range[bucket]
        seg = self.flat[lo:hi]
        if dist.get_backend() == 'gloo' and seg.is_cuda:           # CPU rehearsal backend: through host memory, synchronous
            host = seg.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            seg.copy_(host)
        else:
            self._work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """every bucket flushed (missing milestones are flushed now) and every exchange complete -> 1 / world_size (the averaging
        factor the optimiser applies) -- 1.0 without a process group"""
        for b in BUCKETS:
            if b not in self._flushed:
                self.flush(b)
        for w in self._work:
            w.wait()
        self._work = []
        return 1.0 / dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1.0
