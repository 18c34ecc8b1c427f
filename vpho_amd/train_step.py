"""End-to-end training step: ``vpho_net.forward(mode='train')`` (lib/model/VPHO.py:115-226) with every module
in training mode and ``loss.backward()`` through all of it, data-parallel gradient averaging and AdamW
(lib/engine/train_diff_hand_obj.py:49-52,169-199).

Losses (all 13 of VPHO.py:190-212, weighted as :214-219): ``diff_hand`` / ``diff_obj`` (DSM draws through the score networks),
``hm_hand`` / ``hm_obj`` (heat-map heads), ``vert`` / ``joint`` / ``mano_pose`` / ``mano_shape`` (``head_mano`` + MANO layer; need the
MANO tables in ``assets`` and the hand ground truth in the batch) and ``force`` / ``gravity`` / ``torque`` / ``supervised`` / ``CoM``
(cross modules + ``head_physics``; need the anchor tables in ``assets`` and gravity / obj_CoM / force_local / is_grasped in the batch).
Backward: score networks -> encoders (at the encoding AND, from the physics branch, at the second stage map) -> resize /
re-alignment / heat-map heads -> RoIAlign -> the two-branch backbone.  ``cfg.gradient_clip`` > 0 clips the global gradient norm
(off by default, as in the reference).
Composition of ``train_blocks`` / ``train_score``; torch allocates, slices and, under ``torch.distributed``, all-reduces ONE
flat gradient buffer (RCCL under backend 'nccl').
"""
import torch
import torch.distributed as dist

from . import ops
from .configs.args import cfg
from .model.pack import pack_conv, pack_deconv4x4s2
from .train_blocks import FPNTrain, EncoderTrain, HeatmapHeadTrain, HeadManoTrain, PhysicsTrain
from .train_score import ScoreTrainer, SUFFIXES


class DiffusionTrainStep:
    def __init__(self, state_dict, device, lr=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, loss_weights=None, assets=None,
                 cross_dropout=None):
        self.dev = torch.device(device)
        sd = state_dict
        lr = cfg.base_learning_rate if lr is None else lr
        self.fpn = FPNTrain(sd, 'feature_extractor', self.dev)
        self.hm = dict(h=HeatmapHeadTrain(sd, 'head_hm_hand', self.dev), o=HeatmapHeadTrain(sd, 'head_hm_obj', self.dev))
        self.enc = dict(h=EncoderTrain(sd, 'encoder_hand', self.dev), o=EncoderTrain(sd, 'encoder_obj', self.dev))
        self.score = dict(h=ScoreTrainer(sd, 'denoiser_hand', self.dev, lr, betas, eps, weight_decay),
                          o=ScoreTrainer(sd, 'denoiser_obj', self.dev, lr, betas, eps, weight_decay))
        # MANO losses (head_mano + the MANO layer) need the MANO tables: enabled when `assets` is given
        self.mano_head = HeadManoTrain(sd, 'head_mano', ops.Mano(assets['mano'], self.dev), self.dev) if assets is not None else None
        # physics losses (cross modules + head_physics) need the CPF anchor tables
        self.phys = None
        if assets is not None and 'anchor' in assets and 'cross_hand.proj_hand.weight' in sd:
            from .assets import ANCHOR_SKELETON
            self.phys = PhysicsTrain(sd, ops.Aggregation(assets, ANCHOR_SKELETON, self.dev), self.dev,
                                     cfg.cross_dropout if cross_dropout is None else cross_dropout)
        self.w = dict(force=cfg.weight_force_loss, gravity=cfg.weight_gravity_loss, torque=cfg.weight_torque_loss,
                      supervised=cfg.weight_supervised_loss, CoM=cfg.weight_CoM_loss)
        self.w.update(diff_hand=cfg.weight_diff_hand_loss, diff_obj=cfg.weight_diff_obj_loss, hm_hand=cfg.weight_hm_hand_loss,
                      hm_obj=cfg.weight_hm_obj_loss, vert=cfg.weight_vert_loss, joint=cfg.weight_joint_loss, mano_pose=cfg.weight_mano_pose_loss,
                      mano_shape=cfg.weight_mano_shape_loss)
        self.w.update(loss_weights or {})
        self.hyper = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay)
        self.steps = 0
        # optimiser state: AdamW (element-wise) steps every tensor in the layout the kernels read it in -- the live tensor IS the master.
        # Convolution weights stay packed (cout, kh*kw*cin_pad; padded input channels are zero and stay zero: zero gradient, decoupled
        # decay of zero), their gradients arrive packed from the weight-gradient kernels, and the reference's (cout, cin, kh, kw) is
        # produced only where it is asked for (state_dict / load_params / the gradients handed to callers).  Round 3 kept masters in
        # the reference layout and re-packed 155 tensors after every step (one strided copy launch each).  Only the two ConvTranspose
        # weights keep a reference-layout master (four phase kernels are cut from it).
        self.master, self._repack, self._conv_meta = {}, {}, {}
        self._register()
        from .conv_backward import DgradWeightCache
        self.dgrad_weights = DgradWeightCache()          # the input-gradient layouts of every convolution weight, rebuilt once per step
        self.wino_weights = ops.WinogradWeightBatch()    # ... and the Winograd transforms of the 3x3 weights (forward + input gradient)

    # ------------------------------------------------------------------------------------------------------------------
    def _register(self):
        def add(name, tensor, repack):
            self.master[name] = tensor
            self._repack[name] = repack

        def conv(name, live, shape, cin_pad=None):
            assert live.is_contiguous() and live.shape[0] == shape[0] and live.shape[1] % (shape[2] * shape[3]) == 0, (name, live.shape, shape)
            self._conv_meta[name] = tuple(shape)
            add(name, live, None)

        def vec(name, live):
            add(name, live, None)

        f = self.fpn
        P = 'feature_extractor.'
        conv(P + 'layer0_h.0.weight', f.stem['conv'], f.shapes['layer0_h.0.weight'], 4)
        vec(P + 'layer0_h.1.weight', f.stem['bn']['gamma'])
        vec(P + 'layer0_h.1.bias', f.stem['bn']['beta'])
        for blks in f.blocks.values():
            for k, p, _ in blks:
                for c, b in (('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3')):
                    conv(f'{P}{k}.{c}.weight', p[c], f.shapes[f'{k}.{c}.weight'])
                    vec(f'{P}{k}.{b}.weight', p[b]['gamma'])
                    vec(f'{P}{k}.{b}.bias', p[b]['beta'])
                if 'down' in p:
                    conv(f'{P}{k}.downsample.0.weight', p['down'], f.shapes[f'{k}.downsample.0.weight'])
                    vec(f'{P}{k}.downsample.1.weight', p['bnd']['gamma'])
                    vec(f'{P}{k}.downsample.1.bias', p['bnd']['beta'])
        for hk, (wp, b) in f.heads.items():
            conv(f'{P}{hk}.weight', wp, f.shapes[f'{hk}.weight'])
            vec(f'{P}{hk}.bias', b)
        for br, mod in (('hand', self.hm['h']), ('obj', self.hm['o'])):
            P = f'head_hm_{br}.'
            for (wp, b), key in ((mod.c0, 'conv_layers.0'), (mod.c1, 'conv_layers.1'), (mod.final, 'final_layer')):
                conv(f'{P}{key}.weight', wp, mod.shapes[f'{key}.weight'])
                vec(f'{P}{key}.bias', b)
            vec(P + 'conv_layers.2.weight', mod.bn1['gamma'])
            vec(P + 'conv_layers.2.bias', mod.bn1['beta'])
            vec(P + 'deconv_layers.1.weight', mod.bn2['gamma'])
            vec(P + 'deconv_layers.1.bias', mod.bn2['beta'])
            cin, co = mod.shapes['deconv_layers.0.weight'][:2]
            w = torch.zeros(mod.shapes['deconv_layers.0.weight'], device=self.dev)
            for (py, px), (wp, _, _) in mod.deconv.items():          # ConvTranspose2d weight back from its four phase convolutions
                sub = wp.view(co, 2, 2, cin)
                for dy_ in (0, 1):
                    for dx_ in (0, 1):
                        w[:, :, mod.TAP[py][dy_], mod.TAP[px][dx_]] = sub[:, dy_, dx_, :].t()

            def put(wnew, mod=mod):
                for k, (wp, _, _) in pack_deconv4x4s2(wnew).items():
                    mod.deconv[k][0].copy_(wp)
            add(P + 'deconv_layers.0.weight', w, put)
        for br, mod in (('hand', self.enc['h']), ('obj', self.enc['o'])):
            P = f'encoder_{br}.'
            conv(P + 'project.weight', mod.project[0], mod.shapes['project.weight'], mod.cin_pad)
            vec(P + 'project.bias', mod.project[1])
            for k, p in mod.blocks:
                for c in ('conv1', 'conv2', 'conv3'):
                    conv(f'{P}{k}.{c}.weight', p[c][0], mod.shapes[f'{k}.{c}.weight'])
                    vec(f'{P}{k}.{c}.bias', p[c][1])
                for b in ('bn', 'bn1', 'bn2'):
                    vec(f'{P}{k}.{b}.weight', p[b]['gamma'])
                    vec(f'{P}{k}.{b}.bias', p[b]['beta'])
        for tr in self.score.values():
            for s in SUFFIXES:
                vec(f'{tr.prefix}.{s}', tr.params[s])
        if self.mano_head is not None:
            for k, v in self.mano_head.p.items():
                vec(f'head_mano.{k}', v)
        if self.phys is not None:
            for br, c in self.phys.cross.items():
                for k, v in c.p.items():
                    vec(f'cross_{br}.{k}', v)
                for k, (wp, b) in c.conv.items():
                    conv(f'cross_{br}.{k}.weight', wp, c.shapes[k])
                    vec(f'cross_{br}.{k}.bias', b)
            for k, v in self.phys.p.items():
                vec(f'head_physics.{k}', v)
        # ONE flat gradient buffer, laid out in the order in which the backward finishes the modules and cut into buckets whose
        # all-reduce overlaps the rest of the backward (grad_buckets.py)
        from .grad_buckets import GradBuckets
        self.buckets = GradBuckets({k: tuple(v.shape) for k, v in self.master.items()}, self.dev, conv_meta=self._conv_meta)
        self.names = self.buckets.names
        self.flat_grad, self.grad_view = self.buckets.flat, self.buckets.view
        self.m = {k: torch.zeros_like(v) for k, v in self.master.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.master.items()}

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def loss_and_grads(self, data, gt_hand, gt_obj, draws, want_outputs=False, sink=None):
        """data: the batch dict of ``vpho_net`` (device tensors: rgb, bbox_hand, bbox_obj, bbox_hand_rect, bbox_obj_rect, is_right,
        hm_hand (bs,21,64,64), hm_obj (bs,27,64,64)); gt_hand (bs,96) = mano_aa_to_6D(gt_mano), gt_obj (bs,9); draws: dict t_h, z_h,
        t_o, z_o (reps,bs[,D]).  -> losses {name: 0-d fp64 tensor, weighted}, grads {reference parameter name: gradient}.
        ``sink`` (grad_buckets.GradBuckets): gradients are handed over as soon as they are final and each milestone is flushed, so that
        the bucket's all-reduce runs under the remaining backward."""
        R, HM = cfg.roi_size, cfg.heatmap_size
        f32 = lambda t: t.float().contiguous()
        bb_h, bb_o, bb_hr, bb_or = (f32(data[k]) for k in ('bbox_hand', 'bbox_obj', 'bbox_hand_rect', 'bbox_obj_rect'))
        left = (~data['is_right'].bool()).to(torch.uint8).contiguous()
        eh, eo = self.enc['h'], self.enc['o']
        from .conv_backward import WgradStream
        with torch.cuda.device(self.dev), self.dgrad_weights, self.wino_weights, WgradStream(self.dev) as wstream:
            # ---- forward (VPHO.py:115-150)
            # The FPN outputs are read only through the RoIAligns below (VPHO.py:126-129), so the two smoothing convolutions -- forward,
            # input gradient and weight gradient, the largest launches of the step -- work on the RoI windows only (window = every pixel
            # a bilinear sample of the image's boxes can weigh, vpho_roi_windows_i32; same values on those pixels, zeros elsewhere)
            bs, H, W = data['rgb'].shape[0], data['rgb'].shape[2] // 4, data['rgb'].shape[3] // 4
            fwin = dict(h=ops.roi_windows(bb_h, bb_hr, bs, H, W, 0.25), o=ops.roi_windows(bb_or, None, bs, H, W, 0.25)) if W % 32 == 0 else None
            ph, po = self.fpn.forward(data['rgb'], windows=fwin)
            assert ph.shape[:3] == (bs, H, W)
            in_h = torch.zeros((bs, R, R, eh.cin_pad), device=self.dev)
            in_o = torch.zeros((bs, R, R, eo.cin_pad), device=self.dev)
            hf_hr = ops.roi_align_nhwc(ph, bb_h, R, 0.25)                          # tight box: the hand heat-map head's input
            ops.roi_align_nhwc(ph, bb_hr, R, 0.25, out=in_h)                       # rectangular box: the encoder's input
            of_or = ops.roi_align_nhwc(po, bb_or, R, 0.25)                         # un-flipped: the object heat-map head's input
            ops.roi_align_nhwc(po, bb_or, R, 0.25, flip_w=left, out=in_o)
            hm_h = self.hm['h'].forward(hf_hr)
            hm_o = self.hm['o'].forward(of_or)
            ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_h, bb_h, bb_hr), R, R, out=in_h, c_off=256)
            ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_o, bb_o, bb_or, flip_w=left), R, R, out=in_o, c_off=256)
            enc_h, st_h = eh.forward(in_h)
            enc_o, st_o = eo.forward(in_o)
            # ---- losses (VPHO.py:190-195,214-220) and their gradients at the encodings / heat maps
            L = {}
            L['diff_hand_loss'], d_enc_h = self.score['h']._loss_and_grads(enc_h, f32(gt_hand), f32(draws['t_h']), f32(draws['z_h']))
            L['diff_obj_loss'], d_enc_o = self.score['o']._loss_and_grads(enc_o, f32(gt_obj), f32(draws['t_o']), f32(draws['z_o']))
            L['hm_hand_loss'], d_hm_h = ops.mse_loss(hm_h, ops.nchw_to_nhwc(f32(data['hm_hand'])), self.w['hm_hand'])
            L['hm_obj_loss'], d_hm_o = ops.mse_loss(hm_o, ops.nchw_to_nhwc(f32(data['hm_obj'])), self.w['hm_obj'])
            L['diff_hand_loss'] = L['diff_hand_loss'] * self.w['diff_hand']
            L['diff_obj_loss'] = L['diff_obj_loss'] * self.w['diff_obj']
            d_enc_mano = None
            if self.mano_head is not None and 'gt_hand_vert_flip' in data:
                # head_mano -> MANO -> vert / joint / mano_pose / mano_shape losses (VPHO.py:147-148,197-204); gt_hand is
                # mano_aa_to_6D(gt_mano)[..., :96], the shape ground truth are the last 10 entries of gt_mano
                # HO3D images: the regressed joints enter the joint loss in HO3D's convention (get_joint_aligned_with_HO3D, VPHO.py:154-157)
                ho3d = data['is_ho3d'].to(torch.uint8).contiguous() if 'is_ho3d' in data else None
                Lm, d_enc_mano, gm = self.mano_head.forward_backward(
                    enc_h, f32(data['gt_hand_vert_flip']), f32(data['gt_hand_jt3d_flip']), f32(gt_hand), f32(data['gt_mano'][:, 48:]),
                    data['is_right'].to(torch.uint8).contiguous(), (self.w['vert'], self.w['joint'], self.w['mano_pose'], self.w['mano_shape']),
                    is_ho3d=ho3d)
                L.update(Lm)
            # physics branch (VPHO.py:164-172,205-212): cross modules on the encoders' second stage maps (8x8), head_physics, 5 losses
            d_stage = dict(h=None, o=None)
            gp = None
            if self.phys is not None and 'force_local' in data and 'gt_hand_vert_flip' in data:
                Lp, d_stage['h'], d_stage['o'], gp, _, _ = self.phys.forward_backward(
                    st_h[1], st_o[1], f32(data['gravity']), f32(data['obj_CoM']), data['is_right'], f32(data['gt_hand_vert_flip']),
                    f32(data['force_local']), data['is_grasped'],
                    (self.w['force'], self.w['gravity'], self.w['torque'], self.w['supervised'], self.w['CoM']))
                L.update(Lp)
            # ---- backward
            G = {}
            dfeat = {}
            for tr, w_diff in ((self.score['h'], self.w['diff_hand']), (self.score['o'], self.w['diff_obj'])):
                G.update({f'{tr.prefix}.{s}': (tr.grads[s] if w_diff == 1.0 else tr.grads[s] * w_diff) for s in SUFFIXES})
            if d_enc_mano is not None:
                G.update({f'head_mano.{k}': v for k, v in gm.items()})
            if gp is not None:
                G.update(gp)
            if sink is not None:
                wstream.join()                                 # weight gradients come from their own stream
                sink.put(G)
                sink.flush('heads')
            for br, long_, enc, head, d_enc, d_hm_loss, w_diff, box_rect, box_head, box_tight, flip in (
                    ('h', 'hand', eh, self.hm['h'], d_enc_h, d_hm_h, self.w['diff_hand'], bb_hr, bb_h, bb_h, None),
                    ('o', 'obj', eo, self.hm['o'], d_enc_o, d_hm_o, self.w['diff_obj'], bb_or, bb_or, bb_o, left)):
                if w_diff != 1.0:
                    d_enc = d_enc * w_diff
                if br == 'h' and d_enc_mano is not None:
                    d_enc = ops.add_lrelu(d_enc, d_enc_mano)
                d_in, g = enc.backward(d_enc, d_stage1=d_stage[br])
                G.update({f'encoder_{long_}.{k}': v for k, v in g.items()})
                nj = enc.cin - 256
                # feature channels of the encoder input: RoIAlign of the rectangular box (W-flipped for the object branch)
                df = ops.roi_align_bwd(d_in, box_rect, (H, W), 256, 0.25, flip_w=flip, c_off=0)
                # heat-map channels: 64 -> 32 resize, re-alignment (+ flip), joined by the heat-map loss's own gradient
                d_al = ops.resize_bilinear_bwd(d_in[..., 256:256 + nj].contiguous(), HM, HM)
                d_hm = ops.add_lrelu(ops.align_heatmap_bwd(d_al, box_tight, box_rect, flip), d_hm_loss)
                d_head_in, gh = head.backward(d_hm)
                G.update({f'head_hm_{long_}.{k}': v for k, v in gh.items()})
                dfeat[br] = ops.roi_align_bwd(d_head_in, box_head, (H, W), 256, 0.25, into=df)
                if sink is not None:
                    wstream.join()
                    sink.put({f'encoder_{long_}.{k}': v for k, v in g.items()})
                    sink.put({f'head_hm_{long_}.{k}': v for k, v in gh.items()})
                    sink.flush('branch_hand' if br == 'h' else 'branch_obj')

            def fpn_ready(part, milestone):                    # FPNTrain.backward reports its three milestones
                if sink is not None:
                    wstream.join()
                    sink.put({f'feature_extractor.{k}': v for k, v in part.items()})
                    sink.flush(milestone)
            # the FPN outputs are read only through the RoIAligns above, so their gradients live in the RoI windows: the weight gradients
            # of the two smoothing convolutions (the largest of the step) reduce over those pixels only
            # ... and their input gradients are computed on the windows dilated by the 3x3 halo only
            groups = halo = None
            if W % 32 == 0:
                groups = dict(h=ops.window_groups(fwin['h']), o=ops.window_groups(fwin['o']))
                halo = dict(h=ops.roi_windows(bb_h, bb_hr, bs, H, W, 0.25, dilate=1), o=ops.roi_windows(bb_or, None, bs, H, W, 0.25, dilate=1))
            G.update({f'feature_extractor.{k}': v for k, v in self.fpn.backward(dfeat['h'], dfeat['o'], on_ready=fpn_ready, groups=groups, halo=halo).items()})
            L['total_loss'] = sum(L.values())
            if want_outputs:                                   # pd_dt of VPHO.py:221-225
                pd = dict(hand_heatmap=ops.nhwc_to_nchw(hm_h), obj_heatmap=ops.nhwc_to_nchw(hm_o))
                if self.mano_head is not None and self.mano_head.last_outputs is not None:
                    pd['reg_hand_vert'], pd['reg_hand_joint'] = self.mano_head.last_outputs
                return L, G, pd
        return L, G

    @torch.no_grad()
    def load_params(self, state_dict):
        """refresh every trained tensor (masters + the kernels' packed copies) and the BatchNorm running statistics from a state_dict
        in the reference's layout -- used by vpho_net.forward(mode='train') when an external optimiser has updated the module"""
        for k in self.names:
            src = state_dict[k].to(self.dev)
            if k in self._conv_meta:
                self._ref_view(k).copy_(src.reshape(self._conv_meta[k]))
                continue
            self.master[k].copy_(src.reshape(self.master[k].shape))
            if self._repack[k] is not None:
                self._repack[k](self.master[k])
        for k, v in self._running_stats().items():
            v.copy_(state_dict[k].to(self.dev))
        self.dgrad_weights.refresh()
        self.wino_weights.refresh()

    def _ref_view(self, name):
        """the packed convolution weight `name` seen in the reference's (cout, cin, kh, kw) layout (a strided view of the live tensor)"""
        cout, cin, kh, kw = self._conv_meta[name]
        return self.master[name].view(cout, kh, kw, -1)[..., :cin].permute(0, 3, 1, 2)

    def _running_stats(self):
        """{reference name: live running_mean / running_var tensor}"""
        out = {}

        def bn(name, p):
            out[name + '.running_mean'], out[name + '.running_var'] = p['running_mean'], p['running_var']

        f = self.fpn
        bn('feature_extractor.layer0_h.1', f.stem['bn'])
        for blks in f.blocks.values():
            for k, p, _ in blks:
                for b, nm in (('bn1', 'bn1'), ('bn2', 'bn2'), ('bn3', 'bn3'), ('bnd', 'downsample.1')):
                    if b in p:
                        bn(f'feature_extractor.{k}.{nm}', p[b])
        for br, mod in (('hand', self.hm['h']), ('obj', self.hm['o'])):
            bn(f'head_hm_{br}.conv_layers.2', mod.bn1)
            bn(f'head_hm_{br}.deconv_layers.1', mod.bn2)
        for br, mod in (('hand', self.enc['h']), ('obj', self.enc['o'])):
            for k, p in mod.blocks:
                for b in ('bn', 'bn1', 'bn2'):
                    bn(f'encoder_{br}.{k}.{b}', p[b])
        return out

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, data, gt_hand, gt_obj, draws=None, repeat_num=None, eps=1e-5, lr=None, gradient_clip=None):
        """loss_and_grads + gradient average over the ranks (bucketed all-reduces issued DURING the backward, grad_buckets.py) + AdamW on
        every tensor.
        Without `draws` they are made like loss_fn's (torch.rand / torch.randn on the device, score_based_model.py:24,31)."""
        bs = data['rgb'].shape[0]
        reps = cfg.repeat_num if repeat_num is None else repeat_num
        if draws is None:
            u = lambda: torch.rand(reps, bs, device=self.dev) * (1. - eps) + eps
            draws = dict(t_h=u(), z_h=torch.randn(reps, bs, 96, device=self.dev), t_o=u(), z_o=torch.randn(reps, bs, 9, device=self.dev))
        self.buckets.begin()
        losses, grads = self.loss_and_grads(data, gt_hand, gt_obj, draws, sink=self.buckets)
        unknown = set(grads) - set(self.names)
        assert not unknown, sorted(unknown)[:5]
        # single process: tensors no loss of this batch reached keep their value (grad None, like torch.optim skips them).  Data
        # parallel: WHICH tensors a batch reaches is decided per rank (e.g. the physics keys of a batch), but the bucket exchange
        # delivers the other ranks' gradients for them all the same, and under DDP every parameter has a (possibly zero) gradient on
        # every rank -- so every tensor is stepped on every rank and the replicas cannot drift apart
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        live = list(self.names) if multi else [k for k in self.names if k in grads]
        scale = self.buckets.finish()                            # waits for the bucket exchanges still in flight
        clip = cfg.gradient_clip if gradient_clip is None else gradient_clip
        if clip > 0:                                       # accel.clip_grad_norm_ (train_diff_hand_obj.py:182-183): global L2 norm of the averaged gradients
            if scale != 1.0:
                self.flat_grad.mul_(scale)
                scale = 1.0
            coef = (clip / (self.flat_grad.norm(2) + 1e-6)).clamp(max=1.0)            # torch.nn.utils.clip_grad_norm_, no host sync
            self.flat_grad.mul_(coef)
        self.steps += 1
        hyper = dict(self.hyper, lr=self.hyper['lr'] if lr is None else lr)
        with torch.cuda.device(self.dev):
            key = tuple(live)
            if getattr(self, '_adam_key', None) != key:                  # the list is the same every step unless a loss is switched off
                self._adam = ops.AdamWList([(self.master[k], self.grad_view[k], self.m[k], self.v[k]) for k in live])
                self._adam_key = key
            self._adam.step(self.steps, grad_scale=scale, **hyper)
            for k in live:
                if self._repack[k] is not None:
                    self._repack[k](self.master[k])
            self.dgrad_weights.refresh()
            self.wino_weights.refresh()
        return losses

    def state_dict(self):
        """trained tensors + BatchNorm running statistics under the reference's names"""
        out = {k: (self._ref_view(k).clone(memory_format=torch.contiguous_format) if k in self._conv_meta else v.clone()) for k, v in self.master.items()}
        for tr in self.score.values():
            out[f'{tr.prefix}.t_encoder.0.W'] = tr.fourier_W.clone()

        def bn(name, p):
            out[name + '.running_mean'] = p['running_mean'].clone()
            out[name + '.running_var'] = p['running_var'].clone()

        f = self.fpn
        bn('feature_extractor.layer0_h.1', f.stem['bn'])
        for blks in f.blocks.values():
            for k, p, _ in blks:
                for b, nm in (('bn1', 'bn1'), ('bn2', 'bn2'), ('bn3', 'bn3'), ('bnd', 'downsample.1')):
                    if b in p:
                        bn(f'feature_extractor.{k}.{nm}', p[b])
        for br, mod in (('hand', self.hm['h']), ('obj', self.hm['o'])):
            bn(f'head_hm_{br}.conv_layers.2', mod.bn1)
            bn(f'head_hm_{br}.deconv_layers.1', mod.bn2)
        for br, mod in (('hand', self.enc['h']), ('obj', self.enc['o'])):
            for k, p in mod.blocks:
                for b in ('bn', 'bn1', 'bn2'):
                    bn(f'encoder_{br}.{k}.{b}', p[b])
        return out
