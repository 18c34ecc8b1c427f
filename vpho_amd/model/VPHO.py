"""``vpho_net`` -- drop-in for the reference's lib/model/VPHO.py:48-304 (same constructor contract: no arguments, reads
the module-global ``cfg``; same ``forward(data, mode)`` contract and output dict; same checkpoint key layout).
The arithmetic of ``forward(mode='predict')`` runs in the hand-written HIP kernels of ``vpho_amd/csrc`` through the
C-ABI in ``include/vpho_hip.h``; there is no CPU / eager fallback.
"""
import torch
import torch.nn as nn

from ..configs.args import cfg
from ..assets import load_assets, ANCHOR_SKELETON
from . import layers as L


class vpho_net(nn.Module):
    def __init__(self, assets=None):
        super().__init__()
        self.cfg = cfg
        self.assets = assets if assets is not None else load_assets(getattr(cfg, 'asset_root', 'asset'))
        self.anchor_skeleton = ANCHOR_SKELETON
        self.feature_extractor = L.FPN()
        self.denoiser_hand = L.BaseDenoiser(head='mano_pose')
        self.denoiser_obj = L.BaseDenoiser(head='obj')
        self.head_hm_hand = L.HeadHeatmap2(256, 21, 128)
        self.head_hm_obj = L.HeadHeatmap2(256, 27, 128)
        self.encoder_hand = L.Encoder(256 + 21, 256)
        self.encoder_obj = L.Encoder(256 + 27, 256)
        self.head_mano = L.HeadMano(self.assets['mano'], in_dim=1024)
        self.head_obj = L.HeadObject(self.assets['ycb'])
        self.cross_hand = L.CrossModule(8, 512)
        self.cross_obj = L.CrossModule(8, 512)
        self.head_physics = L.HeadPhysics(hid_dim=512)
        self._engine = None

    def forward(self, data, mode='predict'):
        assert mode in ['train', 'score', 'sample', 'predict']
        if mode == 'train':
            return self._forward_train(data)
        if mode != 'predict':
            # the reference asserts these two names and implements neither (VPHO.py:113,175,227: only 'train' / 'predict' branches)
            raise NotImplementedError(f"mode='{mode}' has no implementation in the reference either (lib/model/VPHO.py:113-304)")
        from .engine import Engine
        if self._engine is None or self._engine.stale(self):
            self._engine = Engine(self)
        return self._engine.predict(data)

    # ------------------------------------------------------------------------------------------------------------ training
    def _forward_train(self, data):
        """``forward(data, mode='train')`` (lib/model/VPHO.py:175-226): returns ``(loss_dt, pd_dt)`` -- the 13 weighted losses +
        ``total_loss``, and reg_hand_vert / reg_hand_joint / hand_heatmap / obj_heatmap.  The whole step runs on the HIP kernels of
        ``train_step.DiffusionTrainStep`` (forward AND analytic backward in one pass; there is no autograd graph over them).
        ``loss_dt['total_loss']`` carries a one-node graph instead: ``total_loss.backward()`` / ``accel.backward(total_loss)``
        (train_diff_hand_obj.py:180) deposits the gradients into ``.grad`` of this module's parameters, scaled by the incoming
        gradient and accumulated like autograd does, so the reference's loop -- backward, clip, ``optimizer.step()`` -- works
        unchanged with any torch optimiser.  Module parameters are re-packed into the kernels' layouts when they have changed;
        BatchNorm running statistics are written back to the module's buffers.
        Batch keys: those of predict + hm_hand, hm_obj, gt_mano (bs,58 axis-angle + betas), gt_obj (bs,9), gt_hand_vert_flip,
        gt_hand_jt3d_flip, force_local, is_grasped, gravity, obj_CoM (lib/dataset/dexycb6.py:471-509).  ``data['_draws']`` (optional,
        tests): fixed DSM draws instead of fresh ones from the device generator."""
        from ..train_step import DiffusionTrainStep
        from .engine import _signature
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise ops_error("vpho_net.forward(mode='train') runs on the GPU only: move the module with .to('cuda')")
        ts = getattr(self, '_train_step', None)
        sig = _signature(self)
        with torch.no_grad():
            if ts is None:
                # cloned: the step's master copies must not alias this module's parameters (an external optimiser owns those)
                ts = DiffusionTrainStep({k: v.detach().clone() for k, v in self.state_dict().items()}, dev, assets=self.assets)
                object.__setattr__(self, '_train_step', ts)
            elif sig != self._train_sig:
                ts.load_params(self.state_dict())
            gt_hand = data['gt_hand6d'] if 'gt_hand6d' in data else _aa_to_rot6d(data['gt_mano'][:, :48].float())
            bs = data['rgb'].shape[0]
            draws = data.get('_draws')
            if draws is None:
                eps, reps = 1e-5, cfg.repeat_num
                u = lambda: torch.rand(reps, bs, device=dev) * (1. - eps) + eps
                draws = dict(t_h=u(), z_h=torch.randn(reps, bs, 96, device=dev), t_o=u(), z_o=torch.randn(reps, bs, 9, device=dev))
            losses, grads, pd_dt = ts.loss_and_grads(data, gt_hand, data['gt_obj'].float(), draws, want_outputs=True)
            # running statistics live in the step's own buffers: mirror them into the module (nn.BatchNorm2d updates them in forward)
            own = dict(self.named_buffers())
            for k, v in ts._running_stats().items():                  # the live tensors: no clone of the 569 masters per forward
                if k in own:
                    own[k].copy_(v)
        object.__setattr__(self, '_train_sig', _signature(self))
        # every trainable parameter is a real input of the loss node, so autograd itself accumulates into .grad: DistributedDataParallel's
        # per-parameter hooks fire (bucketed all-reduce), no_sync / gradient accumulation work, find_unused_parameters is not needed --
        # a parameter no loss of this batch reaches gets an explicit zero, as it does under the reference's autograd
        named = [(k, p) for k, p in self.named_parameters() if p.requires_grad]
        total = _DepositGrads.apply(losses['total_loss'].detach().float(), [grads.get(k) for k, _ in named], *[p for _, p in named])
        loss_dt = {k: v.detach().float() for k, v in losses.items()}
        loss_dt['total_loss'] = total
        return loss_dt, pd_dt


def _aa_to_rot6d(aa):
    """mano_aa_to_6D(gt_mano)[..., :96] (lib/model/head_mano.py:10-18): pytorch3d's axis_angle_to_matrix (through the quaternion, with
    its small-angle branch) and matrix_to_rotation_6d (first two rows) -- ground-truth preprocessing, plain tensor arithmetic"""
    a = aa.reshape(-1, 3)
    ang = a.norm(dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), a * s], -1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    rows = [1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r)]
    return torch.stack(rows, -1).reshape(aa.shape[0], -1).contiguous()


def ops_error(msg):
    from .. import ops
    return ops.VphoError(msg)


class _DepositGrads(torch.autograd.Function):
    """total_loss as a function of the module's parameters whose backward hands autograd the step's analytic gradients (scaled by the
    incoming gradient).  The parameters are genuine inputs of the node: accumulation into ``.grad``, DistributedDataParallel's
    gradient hooks, ``no_sync`` and ``retain_grad`` behave as for any other autograd function."""

    @staticmethod
    def forward(ctx, value, grads, *params):
        ctx.grads = grads
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        # a parameter no loss of this batch reached: under a process group an explicit zero, so that DistributedDataParallel's hook for it
        # fires and every rank steps every tensor; single-process None, like the reference's autograd -- torch optimisers skip
        # such a parameter (no weight decay, no momentum update), as DiffusionTrainStep.step() does
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        out = []
        for gr, (shape, dtype, device) in zip(ctx.grads, ctx.meta):
            if gr is None:
                out.append(torch.zeros(shape, dtype=dtype, device=device) if ddp else None)
            else:
                out.append(gr.reshape(shape).to(dtype) * g)
        return (None, None, *out)
