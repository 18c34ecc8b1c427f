"""Training of the two score networks on frozen image encodings (first slice).

Counterpart of ``ScoreBasedModelAgent.get_score_loss`` (lib/model/score_based_model.py:117-128; ``repeat_num`` draws of the
denoising-score-matching loss ``loss_fn`` :11-42) as the reference's training forward calls it for ``denoiser_hand`` and
``denoiser_obj`` (lib/model/VPHO.py:190-191), of ``loss.backward()`` restricted to the denoiser parameters and the encoding,
and of the AdamW step (lib/engine/train_diff_hand_obj.py:49-52,169-199).  The backbone/encoder backward (the rest of row 4)
is not part of this slice: ``step`` returns d loss / d encoding for it.

All draws of a step are one batch of rows = repeat_num * batch: every matrix product is ONE fp32-MFMA GEMM launch of
``vpho_conv2d_nhwc_f32`` (forward 1408 -> nheads*256, its input- and weight-gradient products, the encoders), the rest are
the HBM-bound kernels of ``csrc/train_score.hip``.  torch allocates, changes layouts (permute/contiguous) and, under
``torch.distributed``, all-reduces the flat gradient buffer over RCCL -- one collective per step and network.
Parameters are kept in the reference's ``state_dict`` layout (``state_dict()`` round-trips with ``vpho_net``).
"""
import torch
import torch.distributed as dist

from . import ops

SUFFIXES = ('t_encoder.1.weight', 't_encoder.1.bias', 'pose_encoder.0.weight', 'pose_encoder.0.bias',
            'pose_encoder.2.weight', 'pose_encoder.2.bias', 'head.head.0.weight', 'head.head.0.bias',
            'head.head.2.weight', 'head.head.2.bias')


def allreduce_mean_scale(flat_grad):
    """DDP gradient averaging: SUM all-reduce of the flat buffer (RCCL under backend 'nccl'), the 1/world factor is returned
    and folded into the optimiser kernel.  No-op (scale 1) without a process group."""
    from .launch import group_active
    if group_active():
        if dist.get_backend() == 'gloo' and flat_grad.is_cuda:      # one-GPU rehearsal backend: through host memory
            host = flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat_grad.copy_(host)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


def _wgrad(x, dy):
    """weight gradient of y = x W^T: dy^T x (out, in) -- the implicit TN GEMM of csrc/conv_wgrad.hip on a 1x1 'image' per row"""
    return ops.conv2d_wgrad_nhwc(x.view(x.shape[0], 1, 1, x.shape[1]), dy.view(dy.shape[0], 1, 1, dy.shape[1]), 1, 1, 1, 0, 0)


class ScoreTrainer:
    def __init__(self, state_dict, prefix, device, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.prefix, self.dev = prefix, torch.device(device)
        d = lambda t: t.detach().float().to(self.dev).contiguous().clone()
        self.fourier_W = d(state_dict[f'{prefix}.t_encoder.0.W'])                       # fixed buffer (denoiser.py:27)
        self.params = {s: d(state_dict[f'{prefix}.{s}']) for s in SUFFIXES}
        self.nheads = self.params['head.head.0.weight'].shape[0]
        self.D = 3 * self.nheads
        self.Dp = (self.D + 3) // 4 * 4
        # one flat gradient buffer (a single all-reduce) with per-parameter views; AdamW moments likewise
        sizes = [self.params[s].numel() for s in SUFFIXES]
        self.flat_grad = torch.zeros(sum(sizes), device=self.dev)
        self.grads, off = {}, 0
        for s, n in zip(SUFFIXES, sizes):
            self.grads[s] = self.flat_grad[off:off + n].view(self.params[s].shape)
            off += n
        self.m = {s: torch.zeros_like(p) for s, p in self.params.items()}
        self.v = {s: torch.zeros_like(p) for s, p in self.params.items()}
        self.hyper = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay)
        self.steps = 0
        # forward + backward of a step are ~70 sync-free launches on fixed buffers (parameters are updated in place, gradients
        # land in the flat buffer): replayed as one HIP graph per (batch, repeat_num) shape
        import os
        from .model.graphs import GraphedCall
        self.use_graphs = os.environ.get('VPHO_GRAPHS', '1') != '0'
        self._graph = GraphedCall(lambda t: self._loss_and_grads(t['feat'], t['gt'], t['ts'], t['zs']), self.dev)

    # ---------------------------------------------------------------------------------------------- loss + gradients
    @torch.no_grad()
    def loss_and_grads(self, feat, gt_pose, ts, zs):
        """feat (bs,1024), gt_pose (bs,D), ts (reps,bs) in [1e-5,1], zs (reps,bs,D) standard normal.
        Fills self.grads; returns (loss: 0-d fp64 device tensor, d loss / d feat (bs,1024))."""
        f32 = lambda t: t.float().contiguous()
        if not self.use_graphs:
            return self._loss_and_grads(f32(feat), f32(gt_pose), f32(ts), f32(zs))
        with torch.cuda.device(self.dev):
            loss, dfeat = self._graph(dict(feat=f32(feat), gt=f32(gt_pose), ts=f32(ts), zs=f32(zs)))
            return loss.clone(), dfeat.clone()

    @torch.no_grad()
    def _loss_and_grads(self, feat, gt_pose, ts, zs):
        P, n, D, Dp = self.params, self.nheads, self.D, self.Dp
        reps, bs = ts.shape
        M = reps * bs
        with torch.cuda.device(self.dev):
            xt, emb, std = ops.dsm_prepare(gt_pose, ts, zs, self.fourier_W, Dp)
            # ---- forward (denoiser.py:68-82), activations kept
            total = torch.empty((M, 1408), device=self.dev)
            ops.linear_into(emb, P['t_encoder.1.weight'], P['t_encoder.1.bias'], total, 0, out_slope=0.0)          # t_feat
            w0 = torch.nn.functional.pad(P['pose_encoder.0.weight'], (0, Dp - D)).contiguous()
            p1 = ops.linear(xt, w0, P['pose_encoder.0.bias'], out_slope=0.0)
            ops.linear_into(p1, P['pose_encoder.2.weight'], P['pose_encoder.2.bias'], total, 128, out_slope=0.0)   # pose_feat
            total.view(reps, bs, 1408)[:, :, 384:] = feat                                                          # the encoding, per draw
            W1 = P['head.head.0.weight']                                                                           # (n, 1408, 256)
            w1_fwd = W1.permute(0, 2, 1).reshape(n * 256, 1408).contiguous()
            h = ops.linear(total, w1_fwd, P['head.head.0.bias'].reshape(n * 256).contiguous(), out_slope=0.0)      # (M, n*256)
            score = ops.plinear2_fwd(h, P['head.head.2.weight'], P['head.head.2.bias'], std, n)
            loss, dout = ops.dsm_loss(score, zs.view(M, D), std, M)
            # ---- backward
            G = self.grads
            dpre, dw2, db2 = ops.plinear2_bwd(h, dout, P['head.head.2.weight'], n)
            G['head.head.2.weight'].copy_(dw2)
            G['head.head.2.bias'].copy_(db2)
            G['head.head.0.bias'].copy_(ops.colsum(dpre).view(n, 256))
            dw1 = _wgrad(total, dpre)                                                                              # (n*256, 1408)
            G['head.head.0.weight'].copy_(dw1.view(n, 256, 1408).permute(0, 2, 1))
            w1_bwd = W1.permute(1, 0, 2).reshape(1408, n * 256).contiguous()
            dtotal = ops.linear(dpre, w1_bwd)                                                                      # (M, 1408)
            dfeat = ops.sum_repeats(dtotal, 384, bs, reps, 1024)
            dtf = ops.relu_bwd(dtotal, 0, 1408, total, 0, 1408, M, 128)
            G['t_encoder.1.weight'].copy_(_wgrad(emb, dtf))
            G['t_encoder.1.bias'].copy_(ops.colsum(dtf))
            dq2 = ops.relu_bwd(dtotal, 128, 1408, total, 128, 1408, M, 256)
            G['pose_encoder.2.weight'].copy_(_wgrad(p1, dq2))
            G['pose_encoder.2.bias'].copy_(ops.colsum(dq2))
            dp1 = ops.linear(dq2, P['pose_encoder.2.weight'].t().contiguous())
            dq1 = ops.relu_bwd(dp1, 0, 256, p1, 0, 256, M, 256)
            G['pose_encoder.0.weight'].copy_(_wgrad(xt, dq1)[:, :D])
            G['pose_encoder.0.bias'].copy_(ops.colsum(dq1))
        return loss, dfeat

    # ---------------------------------------------------------------------------------------------- one training step
    @torch.no_grad()
    def step(self, feat, gt_pose, ts=None, zs=None, repeat_num=20, eps=1e-5):
        """One optimiser step on a batch.  Without ts/zs the draws are made like loss_fn's (torch.rand / torch.randn on the
        encoding's device, score_based_model.py:24,31).  Under torch.distributed the gradients are averaged over the ranks
        (DDP semantics) with one all-reduce of the flat buffer.  Returns (loss, d loss / d feat) of this rank's batch."""
        bs = feat.shape[0]
        if ts is None:
            ts = torch.rand(repeat_num, bs, device=self.dev) * (1. - eps) + eps
            zs = torch.randn(repeat_num, bs, self.D, device=self.dev)
        loss, dfeat = self.loss_and_grads(feat, gt_pose, ts, zs)
        scale = allreduce_mean_scale(self.flat_grad)
        self.steps += 1
        with torch.cuda.device(self.dev):
            for s in SUFFIXES:
                ops.adamw_(self.params[s], self.grads[s].contiguous(), self.m[s], self.v[s], self.steps, grad_scale=scale, **self.hyper)
        return loss, dfeat

    def state_dict(self):
        out = {f'{self.prefix}.{s}': p.clone() for s, p in self.params.items()}
        out[f'{self.prefix}.t_encoder.0.W'] = self.fourier_W.clone()
        return out
