"""Execution plan of ``vpho_net.forward(mode='predict')`` on the HIP kernels.

``Engine`` packs the module's parameters once (BatchNorm folded, NHWC implicit-GEMM layout, score-net restructuring) and
replays the reference's data flow (VPHO.py:112-304, aggregation.py:1167-1353) as a sequence of C-ABI calls on the current
HIP stream.  Torch is used for allocation, views and dtype/flag conversion only.
"""
import os

import torch

from .. import ops
from ..configs.args import cfg
from . import pack as P

MANO_JOINT_LEVEL = {0: [0], 1: [1, 5, 9, 13, 17], 2: [2, 6, 10, 14, 18], 3: [3, 7, 11, 15, 19], 4: [4, 8, 12, 16, 20]}
PHY_TOPK = 5          # aggregation.py:1246


def _signature(model):
    """Cheap change detector for the packed weights: in-place version counters + storage addresses of all tensors."""
    import itertools
    v = p = 0
    for t in itertools.chain(model.parameters(), model.buffers()):
        v += t._version
        p ^= t.data_ptr()
    return v, p, str(next(model.parameters()).device)


class Engine:
    def __init__(self, model):
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise ops.VphoError("vpho_net.forward(mode='predict') runs on the GPU only: move the module with .to('cuda')")
        self.dev = dev
        self.sig = _signature(model)
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        d = lambda t: t.contiguous().to(dev)
        fold = lambda conv, bn=None, cin_pad=None: P.fold_conv_bn(sd, conv, bn, cin_pad, dev)

        # ---- backbone ------------------------------------------------------------------------------------------
        fe = 'feature_extractor'
        self.stem = fold(f'{fe}.layer0_h.0', f'{fe}.layer0_h.1', cin_pad=4)

        def res_layer(name, blocks, stride):
            out = []
            for i in range(blocks):
                p = f'{fe}.{name}.0.{i}'
                blk = dict(c1=fold(p + '.conv1', p + '.bn1'), c2=fold(p + '.conv2', p + '.bn2'),
                           c3=fold(p + '.conv3', p + '.bn3'), stride=stride if i == 0 else 1, down=None)
                if (p + '.downsample.0.weight') in sd:
                    blk['down'] = fold(p + '.downsample.0', p + '.downsample.1')
                    # conv3 and the projection shortcut as ONE 1x1 convolution over the concatenated inputs [conv2 output | block input]
                    # (vpho_conv_desc.x2): weights side by side along k, biases added
                    blk['c3_down'] = (torch.cat([blk['c3'][0], blk['down'][0]], 1).contiguous(), (blk['c3'][1] + blk['down'][1]).contiguous())
                out.append(blk)
            return out

        self.layers = dict(layer1_h=res_layer('layer1_h', 3, 1), layer2_h=res_layer('layer2_h', 4, 2),
                           layer3_h=res_layer('layer3_h', 6, 2), layer4_h=res_layer('layer4_h', 3, 2),
                           layer2_o=res_layer('layer2_o', 4, 2), layer3_o=res_layer('layer3_o', 6, 2))
        self.fpn = {k: fold(f'{fe}.{k}') for k in ('toplayer_h', 'toplayer_o', 'smooth3_h', 'smooth3_o', 'latlayer1_h',
                                                   'latlayer2_h', 'latlayer3_h', 'latlayer1_o', 'latlayer2_o', 'latlayer3_o')}

        # ---- heat-map heads, encoders ------------------------------------------------------------------------------
        def hm_head(p):
            s, t = P.bn_affine(sd, p + '.deconv_layers.1')
            wd = sd[p + '.deconv_layers.0.weight'] * s[None, :, None, None]
            phases = {k: (d(w), py, px) for k, (w, py, px) in P.pack_deconv4x4s2(wd).items()}
            return dict(c0=fold(p + '.conv_layers.0'), c1=fold(p + '.conv_layers.1', p + '.conv_layers.2'), deconv=phases,
                        deconv_b=d(t), final=fold(p + '.final_layer'))

        self.hm = dict(hand=hm_head('head_hm_hand'), obj=hm_head('head_hm_obj'))

        def encoder(p, cin):
            cin_pad = (cin + 3) // 4 * 4
            blocks = []
            for i in range(8):
                q = f'{p}.reg.{i}'
                s, t = P.bn_affine(sd, q + '.bn')
                blocks.append(dict(pre=(d(s), d(t)), c1=fold(q + '.conv1', q + '.bn1'), c2=fold(q + '.conv2', q + '.bn2'), c3=fold(q + '.conv3')))
            return dict(project=fold(p + '.project', cin_pad=cin_pad), blocks=blocks, cin_pad=cin_pad)

        self.enc = dict(hand=encoder('encoder_hand', 277), obj=encoder('encoder_obj', 283))

        # ---- the twin branches as GROUPS (vpho_conv_desc.groups): hand | object layer2 / layer3, FPN top layer and the laterals above the
        # stride-4 level, heat-map heads up to the last 1x1 (21 / 27 output maps), both encoders (277 / 283 inputs, both padded to 284): same
        # shapes, different weights (backbone_FPN_HFL.py:79-109, VPHO.py:131-149, encoding.py:58-73) -> stacked (2, Cout, K) and run as ONE
        # launch each.  VPHO_GROUPED=0: one launch per branch (A/B aid; bit-identical, tests/test_gpu_predict.py)
        self.grouped = os.environ.get('VPHO_GROUPED', '1') != '0'
        # regression head (head_mano's four linear layers on one row per image) with fp64 products and sums: its rounding noise is what the
        # cascade's regression copies inherit, amplified by the 6-D normalisation (VPHO_HEAD_F64=0: the fp32-MFMA GEMM kernel; A/B aid)
        self.head64 = os.environ.get('VPHO_HEAD_F64', '1') != '0'
        if self.grouped:
            pair = lambda a, b: (torch.stack([a[0], b[0]]).contiguous(), torch.stack([a[1], b[1]]).contiguous())

            def pair_blocks(name_h, name_o):
                out = []
                for bh, bo in zip(self.layers[name_h], self.layers[name_o]):
                    blk = dict(c1=pair(bh['c1'], bo['c1']), c2=pair(bh['c2'], bo['c2']), c3=pair(bh['c3'], bo['c3']), stride=bh['stride'], down=None)
                    if bh['down'] is not None:
                        blk['down'] = pair(bh['down'], bo['down'])
                        blk['c3_down'] = pair(bh['c3_down'], bo['c3_down'])
                    out.append(blk)
                return out

            self.g_layers = dict(layer2=pair_blocks('layer2_h', 'layer2_o'), layer3=pair_blocks('layer3_h', 'layer3_o'))
            self.g_fpn = {k: pair(self.fpn[k + '_h'], self.fpn[k + '_o']) for k in ('toplayer', 'latlayer1', 'latlayer2')}
            hh, ho = self.hm['hand'], self.hm['obj']
            self.g_hm = dict(c0=pair(hh['c0'], ho['c0']), c1=pair(hh['c1'], ho['c1']), deconv_b=torch.stack([hh['deconv_b'], ho['deconv_b']]).contiguous(),
                             deconv={k: (torch.stack([hh['deconv'][k][0], ho['deconv'][k][0]]).contiguous(), hh['deconv'][k][1], hh['deconv'][k][2]) for k in hh['deconv']})
            e2 = dict(hand=encoder('encoder_hand', 283), obj=self.enc['obj'])          # the hand encoder's 277 inputs padded to 284 like the object's 283
            assert e2['hand']['cin_pad'] == e2['obj']['cin_pad']
            self.g_enc = dict(project=pair(e2['hand']['project'], e2['obj']['project']), cin_pad=e2['obj']['cin_pad'],
                              blocks=[dict(pre=(torch.stack([a['pre'][0], b['pre'][0]]).contiguous(), torch.stack([a['pre'][1], b['pre'][1]]).contiguous()),
                                           c1=pair(a['c1'], b['c1']), c2=pair(a['c2'], b['c2']), c3=pair(a['c3'], b['c3']))
                                      for a, b in zip(e2['hand']['blocks'], e2['obj']['blocks'])])

        lin = lambda p: (d(sd[p + '.weight']), d(sd[p + '.bias']))
        self.head_mano = dict(l0=lin('head_mano.base_layer.0'), l2=lin('head_mano.base_layer.2'), pose=lin('head_mano.fc_pose'),
                              shape=lin('head_mano.fc_shape'))

        def cross(p):
            a = p + '.attn.layers.0'
            return dict(proj_hand=fold(p + '.proj_hand'), proj_obj=fold(p + '.proj_obj'),
                        grav=(d(torch.nn.functional.pad(sd[p + '.gravity_proj.weight'], (0, 1))), d(sd[p + '.gravity_proj.bias'])),
                        pe=d(sd[p + '.pose_embedder.pe'][:, 0, :]),
                        in_proj=(d(sd[a + '.self_attn.in_proj_weight']), d(sd[a + '.self_attn.in_proj_bias'])),
                        out_proj=lin(a + '.self_attn.out_proj'), l1=lin(a + '.linear1'), l2=lin(a + '.linear2'),
                        n1=lin(a + '.norm1'), n2=lin(a + '.norm2'))

        self.cross = dict(hand=cross('cross_hand'), obj=cross('cross_obj'))
        self.phys = dict(s0=lin('head_physics.fc_scale.0'), s2=lin('head_physics.fc_scale.2'), w0=lin('head_physics.fc_weight.0'),
                         w2=lin('head_physics.fc_weight.2'), anchor=d(sd['head_physics.anchor']))

        # ---- sampler, MANO, aggregation tables ---------------------------------------------------------------------
        self.score_hand = ops.ScoreNet(sd, 'denoiser_hand', dev)
        self.score_obj = ops.ScoreNet(sd, 'denoiser_obj', dev)
        mano = dict(v_template=sd['head_mano.mano_layer.th_v_template'][0], shapedirs=sd['head_mano.mano_layer.th_shapedirs'],
                    posedirs=sd['head_mano.mano_layer.th_posedirs'], J_regressor=sd['head_mano.mano_layer.th_J_regressor'],
                    weights=sd['head_mano.mano_layer.th_weights'])
        self.mano = ops.Mano(mano, dev)
        names = model.head_obj.names
        ycb = {n: dict(kpt3d=sd[f'head_obj.point_{n}'], verts_sampled=sd[f'head_obj.vert_{n}'], CoM=sd[f'head_obj.CoM_{n}'][0]) for n in names}
        self.agg = ops.Aggregation(dict(ycb=ycb, anchor=model.assets['anchor']), model.anchor_skeleton, dev)
        self.last_info = {}
        self._obj_stream = None
        self._pin = {}
        # ONE persistent host thread drives the object sampler of every predict() call: the C side keeps its pinned
        # staging block and blocking event per thread (score_ode.hip), so a thread per call would leak both
        from concurrent.futures import ThreadPoolExecutor
        self._obj_worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix='vpho-obj-sampler')
        # launch-bound, sync-free phases are replayed as HIP graphs (VPHO_GRAPHS=0: plain launches, same kernels)
        self.use_graphs = os.environ.get('VPHO_GRAPHS', '1') != '0'
        # FPN outputs only where RoIAlign reads them (VPHO_ROI_WINDOW=0: the full 64 x 64 maps; same results)
        self.roi_window = os.environ.get('VPHO_ROI_WINDOW', '1') != '0'
        self.feature_streams = int(os.environ.get('VPHO_FEATURE_STREAMS', '1'))
        # the FPN's three top-down adds (F.interpolate + add, backbone_FPN_HFL.py:66-68) ride in the lateral 1x1 convolutions' epilogues
        # (vpho_conv_desc.res_up; VPHO_FPN_FUSE=0: separate read-modify-write passes; bit-identical)
        self.fpn_fuse = os.environ.get('VPHO_FPN_FUSE', '1') != '0'
        # projection shortcuts of the six stage-opening bottlenecks merged into their conv3 (VPHO_DOWN_FUSE=0: two launches + residual add)
        self.down_fuse = os.environ.get('VPHO_DOWN_FUSE', '1') != '0'
        # 3x3 / stride-1 convolutions in Winograd F(2x2,3x3) form on the fp32 matrix cores (2.25 x fewer multiply-adds, smaller error
        # against fp64 than the direct kernel).  VPHO_WINOGRAD=0: the direct implicit GEMM everywhere
        self.winograd = os.environ.get('VPHO_WINOGRAD', '1') != '0'
        # opt-in split-bf16 convolution products (default: fp32 MFMA); see ops.conv_split
        self.conv_terms = {'f32': 0, 'bf16x6': 6, 'bf16x9': 9}[os.environ.get('VPHO_CONV_MFMA', 'f32')]
        self.serial_samplers = False            # True: object sampler after the hand sampler on one stream (exclusive kernel timings)
        # VPHO_DEVICE_PRIOR=1: draw the sampler's prior with torch's DEVICE generator (Philox) instead of the CPU default generator.
        # NOT the default: the order of draws from the CPU generator is part of the reference's RNG contract (sde.py:26-28: hand
        # (bs*S, 96) first, then object (bs*S, 9)); behind the switch a seeded run is reproducible but is a different random stream
        self.device_prior = os.environ.get('VPHO_DEVICE_PRIOR', '0') == '1'
        # True: last_info['agg'] also keeps the candidates every cascade level scored (4 copies of (bs, 2S, 48)); what the fp64 referee
        # of the selection chain is given (tests / bench parity block)
        self.keep_states = False
        self._feat_side = None
        from .graphs import GraphedCall
        self._features_graph = GraphedCall(self.features, dev)
        self._aggregate_graph = GraphedCall(self._aggregate_from_tensors, dev)

    def stale(self, model):
        return _signature(model) != self.sig

    # ------------------------------------------------------------------------------------------------ feature path
    def _bottleneck(self, x, b, out=None):
        y = ops.conv2d_nhwc(x, *b['c1'], out_slope=0.01)
        if b['stride'] == 1:
            y = ops.conv3x3(y, *b['c2'], out_slope=0.01, winograd=self.winograd)
        else:
            y = ops.conv2d_nhwc(y, *b['c2'], kh=3, kw=3, stride=b['stride'], pad=1, out_slope=0.01)
        if b['down'] is not None and self.down_fuse and x.shape[-1] % 32 == 0 and y.shape[-1] % 32 == 0:
            # the first block of a stage: its projection shortcut (1x1 convolution + BatchNorm of the block input, stride 1 or 2) rides in
            # conv3 as a second input: one launch, and the 4C-wide shortcut map is neither written nor re-read
            return ops.conv2d_nhwc(y, *b['c3_down'], x2=x, stride2=b['stride'], out_slope=0.01, out=out)
        r = x if b['down'] is None else ops.conv2d_nhwc(x, *b['down'], stride=b['stride'])
        return ops.conv2d_nhwc(y, *b['c3'], res=r, out_slope=0.01, out=out)

    def _layer(self, x, name, out=None):
        """``out``: where the layer's LAST block writes its result (a slice of a larger buffer)"""
        blocks = self.layers[name]
        for i, b in enumerate(blocks):
            x = self._bottleneck(x, b, out if i == len(blocks) - 1 else None)
        return x

    def _fpn(self, rgb, windows=None):
        """FPN.forward.  ``windows`` = {'h': (RoiWindows, the same dilated by 1), 'o': ...}: the two stride-4 outputs are produced
        only on the pixels the RoIAligns read, as compact (rows, 256) matrices (vpho_roi_windows_i32); the lateral convolution and
        the top-down add of that level run on the dilated windows."""
        tr = self._fpn_trunk(rgb)
        with self._side():
            obj = self._fpn_branch('o', tr, windows)
        hand = self._fpn_branch('h', tr, windows)
        self._join()
        return [hand, obj]

    # The hand and the object branch are independent between the shared stem / layer1 and the shared layer4, and again from the
    # top-down path to the cross modules.  With VPHO_FEATURE_STREAMS=2 the object branch is issued on a second stream (a parallel
    # branch of the captured HIP graph), so the many sub-chip launches of the two branches (16 x 16 and 8 x 8 maps, 32 x 32 RoI
    # crops) overlap instead of queueing; same kernels, same results.
    def _side(self):
        import contextlib
        if self.feature_streams < 2:
            return contextlib.nullcontext()
        if self._feat_side is None:
            self._feat_side = torch.cuda.Stream(self.dev)
        self._feat_side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._feat_side)

    def _join(self):
        if self.feature_streams >= 2 and self._feat_side is not None:
            torch.cuda.current_stream().wait_stream(self._feat_side)

    def _fpn_trunk(self, rgb):
        x = ops.nchw_to_nhwc(rgb, 4)
        c1 = ops.maxpool_nhwc(ops.conv2d_nhwc(x, *self.stem, kh=7, kw=7, stride=2, pad=3, out_slope=0.01), 3, 2, 1)
        c2 = self._layer(c1, 'layer1_h')
        # shared layer4 (quirk Q6): both branches go through the same weights, so they run as ONE batch of 2N images
        # (twice the tiles per launch at 8x8 resolution, half the launches); convolutions are per-image, results unchanged.  The two
        # layer3 stacks write their last block straight into the halves of that batch: no concatenation copy
        n = c2.shape[0]
        c4 = torch.empty((2 * n, c2.shape[1] // 4, c2.shape[2] // 4, self.layers['layer3_h'][-1]['c3'][0].shape[0]), device=c2.device)
        with self._side():
            c3o = self._layer(c2, 'layer2_o')
            c4o = self._layer(c3o, 'layer3_o', out=c4[n:])
        c3h = self._layer(c2, 'layer2_h')
        c4h = self._layer(c3h, 'layer3_h', out=c4[:n])
        self._join()
        c5 = self._layer(c4, 'layer4_h')
        return dict(c2=c2, h=(c5[:n], c4h, c3h), o=(c5[n:], c4o, c3o))

    def _fpn_branch(self, br, tr, windows):
        c5, c4, c3 = tr[br]
        c2 = tr['c2']
        p = ops.conv2d_nhwc(c5, *self.fpn[f'toplayer_{br}'])
        for lat, c in ((f'latlayer1_{br}', c4), (f'latlayer2_{br}', c3), (f'latlayer3_{br}', c2)):
            # the stride-4 level: lateral convolution and top-down add only inside the windows dilated by the 3x3 halo
            halo = windows[br][1] if (windows is not None and c is c2) else None
            if self.fpn_fuse:       # top-down add inside the lateral convolution's epilogue: the finer map is written once
                p = ops.conv2d_nhwc(c, *self.fpn[lat], rows=halo, rows_scatter=halo is not None, res_up=p)
                continue
            q = ops.conv2d_nhwc(c, *self.fpn[lat], rows=halo, rows_scatter=halo is not None)
            p = ops.resize_bilinear_nhwc(p, q.shape[1], q.shape[2], out=q, accumulate=True, rows=halo)
        return ops.conv3x3(p, *self.fpn[f'smooth3_{br}'], winograd=self.winograd, rows=None if windows is None else windows[br][0])

    def _hm_head(self, x, h):
        y = ops.conv3x3(x, *h['c0'], winograd=self.winograd)
        y = ops.conv3x3(y, *h['c1'], winograd=self.winograd)                             # BN folded; LeakyReLU(1.0) = identity (Q1)
        N, H, W, _ = y.shape
        co = h['deconv_b'].shape[0]
        up = torch.empty((N, 2 * H, 2 * W, co), device=y.device)
        for (py, px), (w, pady, padx) in h['deconv'].items():
            ops.conv2d_nhwc(y, w, h['deconv_b'], kh=2, kw=2, pad_y=pady, pad_x=padx, out_hw=(H, W), out_slope=0.0,
                            out_view=(up, 4 * H * W * co, 4 * W * co, 2 * co, (py * 2 * W + px) * co))
        return ops.conv2d_nhwc(up, *h['final'])

    def _encoder(self, x, e):
        x = ops.conv2d_nhwc(x, *e['project'])
        stages = []
        for i, b in enumerate(e['blocks']):
            y = ops.conv2d_nhwc(x, *b['c1'], in_scale=b['pre'][0], in_shift=b['pre'][1], in_slope=0.01, out_slope=0.01)
            y = ops.conv3x3(y, *b['c2'], out_slope=0.01, winograd=self.winograd)
            x = ops.conv2d_nhwc(y, *b['c3'], res=x)
            if i % 2 == 1:
                x = ops.maxpool_nhwc(x, 2, 2, 0)
                stages.append(x)
        N = x.shape[0]
        return ops.nhwc_to_nchw(x).view(N, -1), stages

    def _cross(self, c, st_h, st_o, grav, flip_u8):
        bs = st_h.shape[0]
        ph = ops.conv3x3(st_h, *c['proj_hand'], winograd=self.winograd)
        po = ops.conv3x3(st_o, *c['proj_obj'], winograd=self.winograd)
        ge = ops.linear(ops.nerf_embed(grav, flip_u8), *c['grav'])
        x = ops.cross_tokens(ph, po, ge, c['pe']).view(bs * 65, 512)
        qkv = ops.linear(x, *c['in_proj'])
        o = ops.linear(ops.mha(qkv, bs, 65, 512, 2).view(bs * 65, 512), *c['out_proj'])
        x = ops.add_layernorm(x, o, *c['n1'])
        ff = ops.linear(ops.linear(x, *c['l1'], out_slope=0.0), *c['l2'])
        return ops.add_layernorm(x, ff, *c['n2'])                                       # (bs*65, 512)

    # ---- the same blocks on the two branches at once: tensors hold the hand images [0, N) and the object images [N, 2N)
    def _bottleneck_g(self, x, b, out=None, x_shared=False):
        y = ops.conv2d_nhwc(x, *b['c1'], out_slope=0.01, groups=2, x_shared=x_shared)
        if b['stride'] == 1:
            y = ops.conv3x3(y, *b['c2'], out_slope=0.01, winograd=self.winograd, groups=2)
        else:
            y = ops.conv2d_nhwc(y, *b['c2'], kh=3, kw=3, stride=b['stride'], pad=1, out_slope=0.01, groups=2)
        if b['down'] is not None and self.down_fuse and x.shape[-1] % 32 == 0 and y.shape[-1] % 32 == 0:
            return ops.conv2d_nhwc(y, *b['c3_down'], x2=x, stride2=b['stride'], out_slope=0.01, out=out, groups=2, x2_shared=x_shared)
        r = x if b['down'] is None else ops.conv2d_nhwc(x, *b['down'], stride=b['stride'], groups=2, x_shared=x_shared)
        return ops.conv2d_nhwc(y, *b['c3'], res=r, out_slope=0.01, out=out, groups=2)

    def _layer_g(self, x, name, x_shared=False):
        for i, b in enumerate(self.g_layers[name]):
            x = self._bottleneck_g(x, b, x_shared=x_shared and i == 0)
        return x

    def _hm_head_g(self, x):
        h = self.g_hm
        y = ops.conv3x3(x, *h['c0'], winograd=self.winograd, groups=2)
        y = ops.conv3x3(y, *h['c1'], winograd=self.winograd, groups=2)                     # BN folded; LeakyReLU(1.0) = identity (Q1)
        N2, H, W, _ = y.shape
        co = h['deconv_b'].shape[1]
        up = torch.empty((N2, 2 * H, 2 * W, co), device=y.device)
        for (py, px), (w, pady, padx) in h['deconv'].items():
            ops.conv2d_nhwc(y, w, h['deconv_b'], kh=2, kw=2, pad_y=pady, pad_x=padx, out_hw=(H, W), out_slope=0.0, groups=2,
                            out_view=(up, 4 * H * W * co, 4 * W * co, 2 * co, (py * 2 * W + px) * co))
        n = N2 // 2                                                                          # the last 1x1 has 21 / 27 output maps: one launch per branch
        return ops.conv2d_nhwc(up[:n], *self.hm['hand']['final']), ops.conv2d_nhwc(up[n:], *self.hm['obj']['final'])

    def _encoder_g(self, x):
        e = self.g_enc
        x = ops.conv2d_nhwc(x, *e['project'], groups=2)
        stages = []
        for i, b in enumerate(e['blocks']):
            y = ops.conv2d_nhwc(x, *b['c1'], in_scale=b['pre'][0], in_shift=b['pre'][1], in_slope=0.01, out_slope=0.01, groups=2)
            y = ops.conv3x3(y, *b['c2'], out_slope=0.01, winograd=self.winograd, groups=2)
            x = ops.conv2d_nhwc(y, *b['c3'], res=x, groups=2)
            if i % 2 == 1:
                x = ops.maxpool_nhwc(x, 2, 2, 0)
                stages.append(x)
        N2 = x.shape[0]
        return ops.nhwc_to_nchw(x).view(N2, -1), stages

    def features(self, data):
        """VPHO.py:112-172.  Returns a dict of device tensors (NHWC unless noted)."""
        with ops.conv_split(self.conv_terms):
            return self._features_grouped(data) if (self.grouped and self.conv_terms == 0) else self._features(data)

    def _features_grouped(self, data):
        """_features with the twin branches as grouped launches (one stream; results bit-identical to _features)"""
        rgb = data['rgb'].float().contiguous()
        bs = rgb.shape[0]
        f32 = lambda k: data[k].float().contiguous()
        is_right = data['is_right'].bool()
        left_u8 = (~is_right).to(torch.uint8).contiguous()
        R = cfg.roi_size
        bb_h, bb_o, bb_hr, bb_or = f32('bbox_hand'), f32('bbox_obj'), f32('bbox_hand_rect'), f32('bbox_obj_rect')
        win_h = win_o = halo_h = halo_o = None
        if self.roi_window:
            fh, fw = rgb.shape[2] // 4, rgb.shape[3] // 4
            win_h = ops.roi_windows(bb_h, bb_hr, bs, fh, fw, 0.25)
            win_o = ops.roi_windows(bb_or, None, bs, fh, fw, 0.25)
            halo_h, halo_o = ops.roi_windows(bb_h, bb_hr, bs, fh, fw, 0.25, dilate=1), ops.roi_windows(bb_or, None, bs, fh, fw, 0.25, dilate=1)
        grav = f32('gravity').view(bs, 3)
        # ---- trunk: shared stem / layer1, the two layer2 / layer3 stacks as groups, shared layer4 on the batch of 2N images (quirk Q6)
        x = ops.nchw_to_nhwc(rgb, 4)
        c1 = ops.maxpool_nhwc(ops.conv2d_nhwc(x, *self.stem, kh=7, kw=7, stride=2, pad=3, out_slope=0.01), 3, 2, 1)
        c2 = self._layer(c1, 'layer1_h')
        c3 = self._layer_g(c2, 'layer2', x_shared=True)                                       # (2N,32,32,512)
        c4 = self._layer_g(c3, 'layer3')                                                       # (2N,16,16,1024)
        c5 = self._layer(c4, 'layer4_h')                                                       # (2N,8,8,2048)
        # ---- top-down path: top layer and the two coarse laterals as groups; the stride-4 level per branch (its windows differ)
        p = ops.conv2d_nhwc(c5, *self.g_fpn['toplayer'], groups=2)
        if self.fpn_fuse:
            p = ops.conv2d_nhwc(c4, *self.g_fpn['latlayer1'], groups=2, res_up=p)
            p = ops.conv2d_nhwc(c3, *self.g_fpn['latlayer2'], groups=2, res_up=p)
        else:
            for lat, c in (('latlayer1', c4), ('latlayer2', c3)):
                q = ops.conv2d_nhwc(c, *self.g_fpn[lat], groups=2)
                p = ops.resize_bilinear_nhwc(p, q.shape[1], q.shape[2], out=q, accumulate=True)
        feats = {}
        for br, pb, halo, win in (('h', p[:bs], halo_h, win_h), ('o', p[bs:], halo_o, win_o)):
            if self.fpn_fuse:
                q = ops.conv2d_nhwc(c2, *self.fpn[f'latlayer3_{br}'], rows=halo, rows_scatter=halo is not None, res_up=pb)
            else:
                q = ops.conv2d_nhwc(c2, *self.fpn[f'latlayer3_{br}'], rows=halo, rows_scatter=halo is not None)
                q = ops.resize_bilinear_nhwc(pb, q.shape[1], q.shape[2], out=q, accumulate=True, rows=halo)
            feats[br] = ops.conv3x3(q, *self.fpn[f'smooth3_{br}'], winograd=self.winograd, rows=win)
        hand_feat, obj_feat = feats['h'], feats['o']
        # ---- RoI crops of both branches in one buffer each: the heads' inputs (2N,32,32,256), the encoders' inputs (2N,32,32,284)
        crop = torch.empty((2 * bs, R, R, 256), device=self.dev)
        enc_in = torch.zeros((2 * bs, R, R, self.g_enc['cin_pad']), device=self.dev)
        in_h, in_o = enc_in[:bs], enc_in[bs:]
        hf_hr = ops.roi_align_nhwc(hand_feat, bb_h, R, 0.25, win=win_h, out=crop[:bs])
        ops.roi_align_nhwc(hand_feat, bb_hr, R, 0.25, out=in_h, win=win_h)
        if win_o is not None:                                                                # one pooling pass, two destinations (VPHO.py:126-138)
            ops.roi_align_dual_nhwc(obj_feat, bb_or, R, 0.25, win_o, in_o, flip_w2=left_u8, out=crop[bs:])
        else:
            ops.roi_align_nhwc(obj_feat, bb_or, R, 0.25, out=crop[bs:])
            ops.roi_align_nhwc(obj_feat, bb_or, R, 0.25, flip_w=left_u8, out=in_o)                # VPHO.py:138
        hm_hand, hm_obj = self._hm_head_g(crop)                                              # (bs,64,64,21), (bs,64,64,27)
        ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_hand, bb_h, bb_hr), R, R, out=in_h, c_off=256)
        ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_obj, bb_o, bb_or, flip_w=left_u8), R, R, out=in_o, c_off=256)
        enc, st = self._encoder_g(enc_in)
        enc_h, enc_o = enc[:bs], enc[bs:]
        st_h, st_o = [t[:bs] for t in st], [t[bs:] for t in st]
        hand_heatmap, obj_heatmap = ops.nhwc_to_nchw(hm_hand), ops.nhwc_to_nchw(hm_obj)
        tok_o = self._cross(self.cross['obj'], st_h[1], st_o[1], grav, left_u8)
        hmn = self.head_mano
        a64 = self.head64                                                                    # the regression head with fp64 accumulation (ops.linear)
        h = ops.linear(ops.linear(enc_h, *hmn['l0'], out_slope=0.01, acc64=a64), *hmn['l2'], out_slope=0.01, acc64=a64)
        pose = ops.rot6d_to_axis_angle(ops.linear(h, *hmn['pose'], acc64=a64), 16)                      # (bs,48)
        shape = ops.linear(h, *hmn['shape'], acc64=a64)                                      # (bs,10)
        ctx = self.mano.shape(shape)
        ho3d = data['is_ho3d'].to(torch.uint8).contiguous() if 'is_ho3d' in data else None
        reg_vert, reg_joint = self.mano.fk(pose, ctx, 1, True, ho3d)
        tok_h = self._cross(self.cross['hand'], st_h[1], st_o[1], grav, left_u8)
        ph = self.phys
        scale = ops.linear(ops.linear(tok_h, *ph['s0'], out_slope=0.01), *ph['s2'])          # (bs*65,1)
        logits = ops.linear(ops.linear(tok_o, *ph['w0'], out_slope=0.01), *ph['w2'])         # (bs*65,8)
        force_local = ops.force_local(scale, logits, ph['anchor'], bs * 32, 32, 65, 0, 32).view(bs, 32, 3)
        return dict(hand_feat=hand_feat, obj_feat=obj_feat, roi_win_hand=win_h, roi_win_obj=win_o, hf_hr=hf_hr, enc_in_hand=in_h, enc_in_obj=in_o,
                    hm_hand_nhwc=hm_hand, hm_obj_nhwc=hm_obj, hand_heatmap=hand_heatmap, obj_heatmap=obj_heatmap,
                    encoding_hand=enc_h, encoding_obj=enc_o, stage_hand=st_h[1], stage_obj=st_o[1], mano_pose=pose, mano_shape=shape,
                    mano_ctx=ctx, reg_hand_vert=reg_vert, reg_hand_joint=reg_joint, tok_hand=tok_h, tok_obj=tok_o, force_local=force_local)

    def _features(self, data):
        rgb = data['rgb'].float().contiguous()
        bs = rgb.shape[0]
        f32 = lambda k: data[k].float().contiguous()
        is_right = data['is_right'].bool()
        left_u8 = (~is_right).to(torch.uint8).contiguous()
        R, HM = cfg.roi_size, cfg.heatmap_size
        bb_h, bb_o, bb_hr, bb_or = f32('bbox_hand'), f32('bbox_obj'), f32('bbox_hand_rect'), f32('bbox_obj_rect')
        win_h = win_o = None
        if self.roi_window:
            # the FPN outputs are read only through these RoIAligns (the reference's `of_or` on bbox_obj is never used, VPHO.py:127)
            fh, fw = rgb.shape[2] // 4, rgb.shape[3] // 4
            win_h = ops.roi_windows(bb_h, bb_hr, bs, fh, fw, 0.25)
            win_o = ops.roi_windows(bb_or, None, bs, fh, fw, 0.25)
            halo = {'h': ops.roi_windows(bb_h, bb_hr, bs, fh, fw, 0.25, dilate=1), 'o': ops.roi_windows(bb_or, None, bs, fh, fw, 0.25, dilate=1)}
        windows = None if win_h is None else {'h': (win_h, halo['h']), 'o': (win_o, halo['o'])}
        eh, eo = self.enc['hand'], self.enc['obj']
        in_h = torch.zeros((bs, R, R, eh['cin_pad']), device=self.dev)
        in_o = torch.zeros((bs, R, R, eo['cin_pad']), device=self.dev)
        grav = f32('gravity').view(bs, 3)
        tr = self._fpn_trunk(rgb)
        with self._side():                                                               # object branch
            obj_feat = self._fpn_branch('o', tr, windows)
            if win_o is not None:                                                        # one pooling pass, two destinations (VPHO.py:126-138)
                of_or_rect = ops.roi_align_dual_nhwc(obj_feat, bb_or, R, 0.25, win_o, in_o, flip_w2=left_u8)
            else:
                of_or_rect = ops.roi_align_nhwc(obj_feat, bb_or, R, 0.25)
                ops.roi_align_nhwc(obj_feat, bb_or, R, 0.25, flip_w=left_u8, out=in_o)           # VPHO.py:138
            hm_obj = self._hm_head(of_or_rect, self.hm['obj'])                           # (bs,64,64,27)
            ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_obj, bb_o, bb_or, flip_w=left_u8), R, R, out=in_o, c_off=256)
            enc_o, st_o = self._encoder(in_o, eo)
            obj_heatmap = ops.nhwc_to_nchw(hm_obj)
        hand_feat = self._fpn_branch('h', tr, windows)
        hf_hr = ops.roi_align_nhwc(hand_feat, bb_h, R, 0.25, win=win_h)
        ops.roi_align_nhwc(hand_feat, bb_hr, R, 0.25, out=in_h, win=win_h)
        hm_hand = self._hm_head(hf_hr, self.hm['hand'])                                  # (bs,64,64,21)
        ops.resize_bilinear_nhwc(ops.align_heatmap_nhwc(hm_hand, bb_h, bb_hr), R, R, out=in_h, c_off=256)
        enc_h, st_h = self._encoder(in_h, eh)
        hand_heatmap = ops.nhwc_to_nchw(hm_hand)
        self._join()
        with self._side():
            tok_o = self._cross(self.cross['obj'], st_h[1], st_o[1], grav, left_u8)
        hmn = self.head_mano
        a64 = self.head64                                                                    # the regression head with fp64 accumulation (ops.linear)
        h = ops.linear(ops.linear(enc_h, *hmn['l0'], out_slope=0.01, acc64=a64), *hmn['l2'], out_slope=0.01, acc64=a64)
        pose = ops.rot6d_to_axis_angle(ops.linear(h, *hmn['pose'], acc64=a64), 16)                  # (bs,48)
        shape = ops.linear(h, *hmn['shape'], acc64=a64)                                  # (bs,10)
        ctx = self.mano.shape(shape)
        ho3d = data['is_ho3d'].to(torch.uint8).contiguous() if 'is_ho3d' in data else None
        reg_vert, reg_joint = self.mano.fk(pose, ctx, 1, True, ho3d)
        tok_h = self._cross(self.cross['hand'], st_h[1], st_o[1], grav, left_u8)
        self._join()
        ph = self.phys
        scale = ops.linear(ops.linear(tok_h, *ph['s0'], out_slope=0.01), *ph['s2'])      # (bs*65,1)
        logits = ops.linear(ops.linear(tok_o, *ph['w0'], out_slope=0.01), *ph['w2'])     # (bs*65,8)
        force_local = ops.force_local(scale, logits, ph['anchor'], bs * 32, 32, 65, 0, 32).view(bs, 32, 3)
        return dict(hand_feat=hand_feat, obj_feat=obj_feat, roi_win_hand=win_h, roi_win_obj=win_o, hf_hr=hf_hr, enc_in_hand=in_h, enc_in_obj=in_o,
                    hm_hand_nhwc=hm_hand, hm_obj_nhwc=hm_obj, hand_heatmap=hand_heatmap, obj_heatmap=obj_heatmap,
                    encoding_hand=enc_h, encoding_obj=enc_o, stage_hand=st_h[1], stage_obj=st_o[1], mano_pose=pose, mano_shape=shape,
                    mano_ctx=ctx, reg_hand_vert=reg_vert, reg_hand_joint=reg_joint, tok_hand=tok_h, tok_obj=tok_o, force_local=force_local)

    # ------------------------------------------------------------------------------------------------ sampling
    def _prior(self, rows, dim):
        """sde.py:26-28: CPU default generator (the same stream as torch.randn(rows, dim)), drawn while the feature kernels
        are still running, filled straight into a persistent pinned buffer (two per shape, alternated, so the previous
        step's asynchronous upload is never overwritten).  The sigma(T0) factor is applied on the device after the upload:
        the same fp32 product, but no multi-threaded CPU region on the launch thread (OpenMP workers spin after one and
        eat the host's CPU quota)."""
        if self.device_prior:
            return torch.randn((rows, dim), device=self.dev)
        key = (rows, dim)
        slot = self._pin.setdefault(key, dict(bufs=[torch.empty((rows, dim), pin_memory=True) for _ in range(2)], i=0))
        slot['i'] ^= 1
        return slot['bufs'][slot['i']].normal_()

    # ------------------------------------------------------------------------------------------------ aggregation
    _AGG_F = ('mano_pose', 'mano_shape', 'hand_heatmap', 'obj_heatmap', 'force_local')
    _AGG_D = ('root_joint_flip', 'root_joint', 'cam_intr_crop_flip', 'bbox_hand', 'bbox_obj_rect', 'is_right', 'is_grasped')

    def _aggregate_from_tensors(self, t):
        """tensor-only signature of aggregate() for graph capture (sizes come from the shapes and the cfg values in the key)"""
        f = {k: t['f_' + k] for k in self._AGG_F}
        f['mano_ctx'] = (t['f_ctx_v'], t['f_ctx_j'])
        data = {k: t['d_' + k] for k in self._AGG_D}
        S = t['obj_pose'].shape[1]
        return self.aggregate(f, data, t['final58'], t['obj_pose'], S, self._agg_k[0], self._agg_k[1], oid=t['oid'])

    def aggregate_graphed(self, f, data, final58, obj_pose, S, k_hand, k_obj):
        t = {'f_' + k: f[k] for k in self._AGG_F}
        t['f_ctx_v'], t['f_ctx_j'] = f['mano_ctx']
        t.update({'d_' + k: data[k] for k in self._AGG_D})
        t.update(final58=final58, obj_pose=obj_pose, oid=self.agg.obj_ids(data['obj_name']))
        self._agg_k = (k_hand, k_obj)
        return self._aggregate_graph(t, (k_hand, k_obj, self.keep_states))

    def aggregate(self, f, data, final58, obj_pose, S, k_hand, k_obj, oid=None):
        """aggregation.py:1167-1353.  final58 (bs*S,58) f32, obj_pose (bs,S,9) f64."""
        A, M = self.agg, self.mano
        bs = obj_pose.shape[0]
        f32 = lambda k: data[k].float().contiguous()
        root_flip, root, Kmat = f32('root_joint_flip'), f32('root_joint'), f32('cam_intr_crop_flip').view(bs, 9)
        bb_h, bb_or = f32('bbox_hand'), f32('bbox_obj_rect')
        isr = data['is_right'].to(torch.uint8).contiguous()
        ungrasp = (~data['is_grasped'].bool()).to(torch.uint8).contiguous()
        oid = A.obj_ids(data['obj_name']) if oid is None else oid
        ctx = f['mano_ctx']
        dbg = dict(hand_topk=[], hand_val=[])
        if self.keep_states:
            dbg['cascade_state'], dbg['hand_score'] = [], []
        # 1. hand cascade
        pose = A.hand_candidates(final58, f['mano_pose'], bs, S)
        tp = None
        for level in range(4):
            observe = [j for l in range(level + 1, 5) for j in MANO_JOINT_LEVEL[l]]
            if self.keep_states:
                dbg['cascade_state'].append(pose.clone())
            _, joints = M.fk(pose.view(-1, 48), ctx, 2 * S, False)
            hv = A.hand_heat(joints.view(bs, 2 * S, 21, 3), root_flip, Kmat, bb_h, f['hand_heatmap'], observe)
            if self.keep_states:
                val, idx, tp, sc = A.hand_fuse_level(hv, pose, k_hand, level, want_topk_pose=(level == 3), want_scores=True)
                dbg['hand_score'].append(sc)
            else:
                val, idx, tp = A.hand_fuse_level(hv, pose, k_hand, level, want_topk_pose=(level == 3))
            dbg['hand_topk'].append(idx)
            dbg['hand_val'].append(val)
        fused_rows = pose.view(bs, 2 * S * 48)                                          # row b starts with candidate 0 = fused pose
        agg_vert, _ = M.fk(fused_rows, ctx, 1, True)
        fpnt, fglob = A.force_anchor(agg_vert, root_flip, f['force_local'], 1)
        # 2-4. object
        hm_o = f['obj_heatmap']
        sc = A.obj_heat_score(obj_pose, root, oid, isr, Kmat, bb_or, hm_o)
        tv, ti = A.topk(sc, k_obj)
        transl = A.obj_fuse(obj_pose, ti.view(bs, -1), A.topk_weights(tv).view(bs, -1))[:, 6:].contiguous()
        sc2 = A.obj_heat_score(obj_pose, root, oid, isr, Kmat, bb_or, hm_o, transl_override=transl)
        _, ri = A.topk(sc2, k_obj)
        cand = A.obj_cross(obj_pose, ti.view(bs, -1), ri.view(bs, -1))
        ps = A.obj_physics_score(cand, root, oid, isr, fpnt, fglob)
        _, pi = A.topk(ps, PHY_TOPK)
        hs = A.obj_heat_score(cand, root, oid, isr, Kmat, bb_or, hm_o)
        hval, hi = A.topk(hs, PHY_TOPK)
        obj_fused = A.obj_fuse(cand, pi.view(bs, -1), None, hi.view(bs, -1), A.topk_weights(hval).view(bs, -1), ungrasp)
        obj_vert = A.obj_verts(obj_fused, root, oid, isr)
        # 5. hand distal joints by pseudo-force
        cand58 = A.hand_phys_candidates(fused_rows, f['mano_shape'], tp)
        n_c = cand58.shape[1]
        cverts, _ = M.fk(cand58.view(-1, 58), ctx, n_c, True)
        fp2, fg2 = A.force_anchor(cverts, root_flip, f['force_local'], n_c)
        fs = A.hand_phys_score(fp2, fg2, obj_vert, bs, n_c)
        _, fidx = A.topk(fs, PHY_TOPK, 5)
        out58 = A.hand_phys_fuse(cand58, fidx)
        out_vert, out_joint = M.fk(out58, ctx, 1, True)
        dbg.update(transl_topk=ti, rot_topk=ri, phys_topk=pi, heat_topk=hi, phys_score=ps, transl_score=sc, rot_score=sc2, heat_score=hs, hand_phys_topk=fidx, hand_phys_score=fs, cand58=cand58, cand_vert=cverts, cand_force_point=fp2, cand_force_global=fg2,
                   cascade_pose=fused_rows[:, :48], force_point=fpnt, force_global=fglob, obj_vert=obj_vert, pose6d_candidate=cand, transl=transl)
        return dict(obj_agg_6d=obj_fused, hand_agg_mano=out58, hand_agg_vert=out_vert, hand_agg_joint=out_joint), dbg

    # ------------------------------------------------------------------------------------------------ whole path
    @torch.no_grad()
    def predict(self, data, noise_hand=None, noise_obj=None):
        S, T0, steps = cfg.sample_num, cfg.sample_T0, cfg.sampling_steps
        with torch.cuda.device(self.dev):
            if self.use_graphs:
                f = self._features_graph({k: v for k, v in data.items() if torch.is_tensor(v)}, (cfg.roi_size, cfg.heatmap_size, self.conv_terms, self.roi_window, self.feature_streams, self.winograd, self.fpn_fuse, self.down_fuse))
                keep = lambda t: t.clone()                 # graph-owned buffers are overwritten by the next replay
            else:
                f = self.features(data)
                keep = lambda t: t
            bs = f['mano_pose'].shape[0]
            sig = 0.01 * (50 / 0.01) ** T0
            init_h = self._prior(bs * S, 96) if noise_hand is None else noise_hand.float()
            init_o = self._prior(bs * S, 9) if noise_obj is None else noise_obj.float()
            init_h, init_o = init_h.to(self.dev, non_blocking=True) * sig, init_o.to(self.dev, non_blocking=True) * sig
            out = dict(reg_hand_vert=keep(f['reg_hand_vert']), reg_hand_joint=keep(f['reg_hand_joint']), hand_heatmap=keep(f['hand_heatmap']),
                       obj_heatmap=keep(f['obj_heatmap']), force_local=keep(f['force_local']))
            # The object sampler (9-d, 3 heads: ~150 workgroups per launch) cannot fill the chip on its own, so it runs
            # concurrently with the hand sampler on a second HIP stream, driven by its own host thread (each sampler
            # blocks on one 8-byte error norm per RK attempt; ctypes releases the GIL during the call).
            main = torch.cuda.current_stream()
            if self._obj_stream is None:
                self._obj_stream = torch.cuda.Stream(device=self.dev)
            obj_stream = self._obj_stream
            obj_stream.wait_stream(main)

            def run_obj():
                with torch.cuda.device(self.dev), torch.cuda.stream(obj_stream):
                    return self.score_obj.sample(f['encoding_obj'], init_o, S, T0, steps, xs_f64=True, x_f64=True)

            concurrent = os.environ.get('VPHO_SERIAL_SAMPLERS', '0') != '1' and not self.serial_samplers
            fut = self._obj_worker.submit(run_obj) if concurrent else None
            # hand hypotheses
            xs_h, x_h, st_h = self.score_hand.sample(f['encoding_hand'], init_h, S, T0, steps, xs_f64=False, x_f64=False)
            inproc = torch.empty((bs * S * steps, 58), device=self.dev)
            ops.rot6d_to_axis_angle(xs_h.view(bs * S * steps, 96), 16, out=inproc)
            ops.append_betas(f['mano_shape'], inproc, S * steps)
            final = torch.empty((bs * S, 58), device=self.dev)
            ops.rot6d_to_axis_angle(x_h, 16, out=final)
            ops.append_betas(f['mano_shape'], final, S)
            out['diff_inprocess_hand_mano'] = inproc.view(bs, S, steps, 58)
            out['diff_final_hand_mano'] = final.view(bs, S, 58)
            ctx = f['mano_ctx']
            viz = inproc.view(bs * S, steps, 58)[0, ::10].contiguous()                   # VPHO.py:250 (first sample, every 10th stamp)
            vv, vj = self.mano.fk(viz, ctx, viz.shape[0], True)
            out['diff_inprocess_hand_vert'], out['diff_inprocess_hand_joint'] = vv, vj
            fv, fj = self.mano.fk(final, ctx, S, True)
            out['diff_final_hand_vert'] = fv.view(bs, S, 778, 3)
            out['diff_final_hand_joint'] = fj.view(bs, S, 21, 3)
            # object hypotheses (stay fp64, quirk Q5)
            xs_o, x_o, st_o = fut.result() if fut is not None else run_obj()      # a worker exception re-raises here
            main.wait_stream(obj_stream)
            for t in (xs_o, x_o):
                t.record_stream(main)
            out['diff_inprocess_obj_6d'] = xs_o.view(bs, S, steps, 9)
            out['diff_final_obj_6d'] = x_o.view(bs, S, 9)
            for name, st in (('hand', st_h), ('obj', st_o)):
                if st['nan_count']:
                    print("\033[31mWarning: NaN detected in score evaluation. \033[0m")
            x_o9 = out['diff_final_obj_6d']
            if self.use_graphs:
                agg, dbg = self.aggregate_graphed(f, data, final, x_o9, S, cfg.topk_hand, cfg.topk_obj)
            else:
                agg, dbg = self.aggregate(f, data, final, x_o9, S, cfg.topk_hand, cfg.topk_obj)
            out['agg_obj_6d'] = keep(agg['obj_agg_6d'])
            out['agg_hand_mano'] = keep(agg['hand_agg_mano'])
            out['agg_hand_vert'] = keep(agg['hand_agg_vert'])
            out['agg_hand_joint'] = keep(agg['hand_agg_joint'])
            self.last_info = dict(features=f, hand_ode=st_h, obj_ode=st_o, agg=dbg, hand_x6d=x_h)      # x_h: the sampler's raw rot6d hypotheses
        return out
