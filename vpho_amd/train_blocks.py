"""Training-mode forward + backward of a ResNet bottleneck (SURVEY.md 8f row 4 building block): the composition of the
convolution, BatchNorm(train) and LeakyReLU kernels for ``Bottleneck.forward`` (lib/model/backbone_FPN_HFL.py:330-350) under
``model.train()`` and ``loss.backward()``.  Activations NHWC fp32, weights in the forward kernel's packed layout
(Cout, KH*KW*Cin); torch only allocates.  Composed into the end-to-end step by ``train_step.DiffusionTrainStep``."""
from . import ops
from . import conv_backward as CB

SLOPE = 0.01                                             # nn.LeakyReLU() default (backbone_FPN_HFL.py:326)
import os
FUSE_LRELU_BWD = os.environ.get('VPHO_TRAIN_FUSE_LRELU', '1') != '0'      # A/B aid: 0 = one stand-alone LeakyReLU backward per bottleneck (same bits)


class BottleneckTrain:
    """params: dict with packed conv weights 'conv1','conv2','conv3' (+'down') and BatchNorm ('bn1','bn2','bn3' (+'bnd')) each a dict
    gamma / beta / running_mean / running_var (device tensors, running stats updated in place)."""

    def __init__(self, params, stride=1):
        self.p, self.stride = params, stride

    def _bn(self, name, x, slope, res=None, partials=None):
        b = self.p[name]
        return ops.bn_train_forward(x, b['gamma'], b['beta'], b['running_mean'], b['running_var'], slope=slope, res=res, partials=partials)

    def _bn_back(self, name, c, saved):
        b = self.p[name]
        return ops.BnFuse(c, saved, b['gamma'], b['beta'])

    def forward(self, x):
        # every convolution leaves the column sums of its output for the BatchNorm behind it (ops.BnFuse)
        p, s = self.p, self.stride
        f1, f2, f3, fd = ops.BnFuse(), ops.BnFuse(), ops.BnFuse(), ops.BnFuse()
        c1 = ops.conv2d_nhwc(x, p['conv1'], bn=f1)
        a1, s1 = self._bn('bn1', c1, SLOPE, partials=f1)
        c2 = ops.conv3x3_train(a1, p['conv2'], bn=f2) if s == 1 else ops.conv2d_nhwc(a1, p['conv2'], kh=3, kw=3, stride=s, pad=1, bn=f2)
        a2, s2 = self._bn('bn2', c2, SLOPE, partials=f2)
        c3 = ops.conv2d_nhwc(a2, p['conv3'], bn=f3)
        if 'down' in p:
            cd = ops.conv2d_nhwc(x, p['down'], stride=s, bn=fd)
            res, sd = self._bn('bnd', cd, 1.0, partials=fd)
        else:
            cd, res, sd = None, x, None
        out, s3 = self._bn('bn3', c3, SLOPE, res=res, partials=f3)     # lrelu(bn3(c3) + shortcut): the add rides in the BatchNorm apply
        self.saved = dict(x=x, c1=c1, a1=a1, s1=s1, c2=c2, a2=a2, s2=s2, c3=c3, s3=s3, cd=cd, sd=sd, out=out)
        return out

    def bn3_fuse(self):
        """carrier for the sums of this block's bn3 backward, to be filled by the convolution that produces (and gates) the gradient at the
        block's output: the NEXT block's conv1 input-gradient convolution (backward(..., gate_input=True, bn_prev=...))"""
        S, b = self.saved, self.p['bn3']
        return ops.BnFuse(S['c3'], S['s3'], b['gamma'], b['beta'], stored_gate=True)

    def backward(self, dout, gated=False, gate_input=False, bn_prev=None, bn3=None):
        """-> dx, grads {name: tensor} for every conv weight (packed layout) and BatchNorm gamma/beta.
        gated: dout has already been taken through the backward of this block's closing LeakyReLU (by the block behind it, see
        gate_input).  gate_input (identity-shortcut blocks): this block's input IS the previous block's output, so the backward of THAT
        block's closing LeakyReLU rides in the epilogue of the convolution that produces dx (its gate = the input): the stand-alone
        element-wise pass of one block per pair goes.  bn_prev (with gate_input): the previous block's ``bn3_fuse()`` -- the same epilogue
        also leaves the two sums of that block's bn3 backward; bn3 (with gated): this block's own carrier, filled that way."""
        p, s, S = self.p, self.stride, self.saved
        H, W = S['x'].shape[1:3]
        g = {}
        dsum = dout if gated else ops.lrelu_bwd(dout, S['out'], SLOPE)
        dc3, g['bn3.gamma'], g['bn3.beta'] = ops.bn_train_backward(S['c3'], dsum, p['bn3']['gamma'], S['s3'], partials=bn3 if gated else None)
        g['conv3'] = CB.conv2d_wgrad(S['a2'], dc3, 1, 1)
        # the input-gradient convolutions take the activation's backward AND the two sums of the BatchNorm backward in their epilogue
        f2 = self._bn_back('bn2', S['c2'], S['s2'])
        da2 = CB.conv2d_dgrad(dc3, p['conv3'], S['a2'].shape[1:3], 1, 1, gate=(S['a2'], SLOPE), bn=f2)
        dc2, g['bn2.gamma'], g['bn2.beta'] = ops.bn_train_backward(S['c2'], da2, p['bn2']['gamma'], S['s2'], partials=f2)
        g['conv2'] = CB.conv2d_wgrad(S['a1'], dc2, 3, 3, s, 1)
        f1 = self._bn_back('bn1', S['c1'], S['s1'])
        da1 = CB.conv2d_dgrad(dc2, p['conv2'], (H, W), 3, 3, s, 1, gate=(S['a1'], SLOPE), bn=f1)
        dc1, g['bn1.gamma'], g['bn1.beta'] = ops.bn_train_backward(S['c1'], da1, p['bn1']['gamma'], S['s1'], partials=f1)
        g['conv1'] = CB.conv2d_wgrad(S['x'], dc1, 1, 1)
        if 'down' in p:
            dx = CB.conv2d_dgrad(dc1, p['conv1'], (H, W), 1, 1)
            dcd, g['bnd.gamma'], g['bnd.beta'] = ops.bn_train_backward(S['cd'], dsum, p['bnd']['gamma'], S['sd'])
            g['down'] = CB.conv2d_wgrad(S['x'], dcd, 1, 1, s, 0)
            dx = CB.conv2d_dgrad(dcd, p['down'], (H, W), 1, 1, 1, 0, res=dx) if s == 1 else ops.add_lrelu(dx, CB.conv2d_dgrad(dcd, p['down'], (H, W), 1, 1, s, 0))
        else:
            # identity shortcut: the sum rides in the dgrad epilogue (and, on request, the previous block's LeakyReLU backward behind it)
            dx = CB.conv2d_dgrad(dc1, p['conv1'], (H, W), 1, 1, res=dsum, gate=(S['x'], SLOPE) if gate_input else None, bn=bn_prev if gate_input else None)
        return dx, g


# ------------------------------------------------------------------------------------------------------------------------
def _bn_params(sd, key, dev):
    g = lambda s: sd[f'{key}.{s}'].detach().float().to(dev).contiguous().clone()
    return dict(gamma=g('weight'), beta=g('bias'), running_mean=g('running_mean'), running_var=g('running_var'))


def _unpack_grad(gp, cout, cin, kh, kw):
    """packed (Cout, kh*kw*cin_pad) gradient -> the reference's (Cout, Cin, kh, kw), as a VIEW (what callers and the module path read);
    the packed tensor rides along as ``packed_grad``: the flat gradient buffer and AdamW keep convolution weights in the packed layout
    (grad_buckets.GradBuckets.put)"""
    view = gp.view(cout, kh, kw, -1)[..., :cin].permute(0, 3, 1, 2)
    view.packed_grad = gp
    return view


class FPNTrain:
    """Training-mode forward + backward of the two-branch ResNet-50 / FPN backbone (``FPN.forward``,
    lib/model/backbone_FPN_HFL.py:70-109, under ``model.train()``): shared stem and layer1, hand / object layers 2-3, the SHARED
    layer4 applied to both branches in two separate calls (each with its own batch statistics, quirk Q6), top-down path with
    lateral 1x1 convolutions and bilinear up-sample-add, 3x3 smoothing.  ``backward`` returns the gradient of every parameter
    under the reference's state_dict names and layouts."""
    LAYERS = (('layer1_h', 3, 64, 1), ('layer2_h', 4, 128, 2), ('layer2_o', 4, 128, 2), ('layer3_h', 6, 256, 2), ('layer3_o', 6, 256, 2),
              ('layer4_h', 3, 512, 2))

    def __init__(self, sd, prefix, device):
        from .model.pack import pack_conv
        self.dev = dev = device
        self.pfx = prefix
        w = lambda k: sd[f'{prefix}.{k}'].detach().float()
        self.shapes = {}

        def conv(key, cin_pad=None):
            t = w(key + '.weight')
            self.shapes[key + '.weight'] = tuple(t.shape)
            return pack_conv(t, cin_pad).to(dev)

        self.stem = dict(conv=conv('layer0_h.0', 4), bn=_bn_params(sd, f'{prefix}.layer0_h.1', dev))
        self.blocks = {}
        for name, n, planes, stride in self.LAYERS:
            blks = []
            for i in range(n):
                k = f'{name}.0.{i}'
                p = dict(conv1=conv(k + '.conv1'), conv2=conv(k + '.conv2'), conv3=conv(k + '.conv3'),
                         bn1=_bn_params(sd, f'{prefix}.{k}.bn1', dev), bn2=_bn_params(sd, f'{prefix}.{k}.bn2', dev), bn3=_bn_params(sd, f'{prefix}.{k}.bn3', dev))
                if f'{prefix}.{k}.downsample.0.weight' in sd:
                    p['down'] = conv(k + '.downsample.0')
                    p['bnd'] = _bn_params(sd, f'{prefix}.{k}.downsample.1', dev)
                blks.append((k, p, stride if i == 0 else 1))
            self.blocks[name] = blks
        self.heads = {}
        for k in ('toplayer', 'latlayer1', 'latlayer2', 'latlayer3', 'smooth3'):
            for br in 'ho':
                self.heads[f'{k}_{br}'] = (conv(f'{k}_{br}'), w(f'{k}_{br}.bias').to(dev).contiguous())

    # -------------------------------------------------------------------------------------------------- forward
    def _run_layer(self, name, x, tag):
        seq = []
        for k, p, stride in self.blocks[name]:
            b = BottleneckTrain(p, stride)
            x = b.forward(x)
            seq.append((k, b))
        self.calls[(name, tag)] = seq
        return x

    def forward(self, rgb_nchw, windows=None):
        """rgb (N,3,H,W) -> p2_h, p2_o (N,H/4,W/4,256) NHWC.  windows = {'h': RoiWindows, 'o': RoiWindows} (optional): the only readers
        of the two outputs are RoIAligns over these windows, so the smoothing convolutions compute the window pixels only"""
        self.calls = {}
        x = ops.nchw_to_nhwc(rgb_nchw.float().contiguous(), 4)
        f0 = ops.BnFuse()
        c0 = ops.conv2d_nhwc(x, self.stem['conv'], kh=7, kw=7, stride=2, pad=3, bn=f0)
        b = self.stem['bn']
        a0, s0 = ops.bn_train_forward(c0, b['gamma'], b['beta'], b['running_mean'], b['running_var'], slope=SLOPE, partials=f0)
        c1 = ops.maxpool_nhwc(a0, 3, 2, 1)
        c2 = self._run_layer('layer1_h', c1, 'h')
        c3h, c3o = self._run_layer('layer2_h', c2, 'h'), self._run_layer('layer2_o', c2, 'o')
        c4h, c4o = self._run_layer('layer3_h', c3h, 'h'), self._run_layer('layer3_o', c3o, 'o')
        c5h = self._run_layer('layer4_h', c4h, 'h')
        c5o = self._run_layer('layer4_h', c4o, 'o')
        feats = dict(h=(c5h, c4h, c3h, c2), o=(c5o, c4o, c3o, c2))
        out, self.td = {}, {}
        for br in 'ho':
            c5, c4, c3, c2_ = feats[br]
            p = ops.conv2d_nhwc(c5, *self.heads[f'toplayer_{br}'])
            for lat, c in ((f'latlayer1_{br}', c4), (f'latlayer2_{br}', c3), (f'latlayer3_{br}', c2_)):
                q = ops.conv2d_nhwc(c, *self.heads[lat])
                p = ops.resize_bilinear_nhwc(p, q.shape[1], q.shape[2], out=q, accumulate=True)
            self.td[br] = p                                               # p2 before smoothing
            out[br] = ops.conv3x3_train(p, *self.heads[f'smooth3_{br}'], rows=None if windows is None else windows[br])
        self.saved = dict(x=x, c0=c0, a0=a0, s0=s0, c1=c1, feats=feats)
        return out['h'], out['o']

    # -------------------------------------------------------------------------------------------------- backward
    def _back_layer(self, name, tag, dy, grads):
        seq = self.calls[(name, tag)]
        gated, f3 = False, None
        for i in range(len(seq) - 1, -1, -1):
            k, b = seq[i]
            fuse = FUSE_LRELU_BWD and i > 0 and 'down' not in b.p          # block i's input = block i-1's output (same layer)
            f_prev = seq[i - 1][1].bn3_fuse() if fuse else None            # ... and the sums of block i-1's bn3 backward come with it
            dy, g = b.backward(dy, gated=gated, gate_input=fuse, bn_prev=f_prev, bn3=f3)
            gated, f3 = fuse, f_prev
            for gk, gv in g.items():
                if gk in ('conv1', 'conv2', 'conv3', 'down'):
                    key = f'{k}.{"downsample.0" if gk == "down" else gk}.weight'
                    if key in grads:                                      # layer4's second call: summed in the packed layout
                        prev = grads.pop(key).packed_grad                 # (on the stream that produced both terms)
                        gv = CB.on_wgrad_stream(lambda a=prev, b=gv: a + b, [prev, gv])
                    val = _unpack_grad(gv, *self.shapes[key])
                else:
                    bn, which = gk.split('.')
                    key = f'{k}.{"downsample.1" if bn == "bnd" else bn}.{"weight" if which == "gamma" else "bias"}'
                    val = gv
                grads[key] = grads[key] + val if key in grads else val    # layer4 is called twice: its gradients add up
        return dy

    def backward(self, dp2_h, dp2_o, on_ready=None, groups=None, halo=None):
        """groups = {'h': ops.window_groups(...), 'o': ...} (optional): where dp2_h / dp2_o can be non-zero (they come back through
        RoIAligns) -- the smoothing convolutions' weight gradients then skip the rest of the map; halo = {'h': RoiWindows dilated by one
        pixel, 'o': ...}: where their INPUT gradients can be non-zero (computed there only).
        on_ready(part, milestone): called with the gradients that have become final at 'fpn_top' (smoothing / lateral / top layers),
        'fpn_mid' (layers 4-2 of both branches) and 'fpn_end' (layer1 + stem) -- grad_buckets.py"""
        S, grads = self.saved, {}
        reported = set()

        def report(milestone):
            if on_ready is not None:
                on_ready({k: v for k, v in grads.items() if k not in reported}, milestone)
                reported.update(grads)
        add = lambda a, b_: b_ if a is None else ops.add_lrelu(a, b_)
        dfe = dict(h=[None] * 3, o=[None] * 3)                            # gradients reaching c5, c4, c3 of each branch
        dc2 = None
        for br, dp in (('h', dp2_h), ('o', dp2_o)):
            c5, c4, c3, c2 = S['feats'][br]
            wS = self.heads[f'smooth3_{br}'][0]
            grads[f'smooth3_{br}.weight'] = _unpack_grad(CB.conv2d_wgrad(self.td[br], dp, 3, 3, 1, 1, groups=None if groups is None else groups[br]),
                                                         *self.shapes[f'smooth3_{br}.weight'])
            grads[f'smooth3_{br}.bias'] = CB.conv2d_bias_grad(dp)
            d = CB.conv2d_dgrad(dp, wS, self.td[br].shape[1:3], 3, 3, 1, 1, rows=None if halo is None else halo[br])     # d p2 (pre-smoothing)
            for lat, c, slot in ((f'latlayer3_{br}', c2, None), (f'latlayer2_{br}', c3, 2), (f'latlayer1_{br}', c4, 1)):
                wl = self.heads[lat][0]
                grads[lat + '.weight'] = _unpack_grad(CB.conv2d_wgrad(c, d, 1, 1), *self.shapes[lat + '.weight'])
                grads[lat + '.bias'] = CB.conv2d_bias_grad(d)
                dc = CB.conv2d_dgrad(d, wl, c.shape[1:3], 1, 1)
                if slot is None:
                    dc2 = add(dc2, dc)
                else:
                    dfe[br][slot] = add(dfe[br][slot], dc)
                nxt = {f'latlayer3_{br}': c3, f'latlayer2_{br}': c4, f'latlayer1_{br}': c5}[lat]
                d = ops.resize_bilinear_bwd(d, nxt.shape[1], nxt.shape[2])                  # gradient of the up-sampled coarser map
            wt = self.heads[f'toplayer_{br}'][0]
            grads[f'toplayer_{br}.weight'] = _unpack_grad(CB.conv2d_wgrad(c5, d, 1, 1), *self.shapes[f'toplayer_{br}.weight'])
            grads[f'toplayer_{br}.bias'] = CB.conv2d_bias_grad(d)
            dfe[br][0] = add(dfe[br][0], CB.conv2d_dgrad(d, wt, c5.shape[1:3], 1, 1))
        report('fpn_top')
        # bottom-up path in reverse
        for br in 'ho':
            d4 = add(dfe[br][1], self._back_layer('layer4_h', br, dfe[br][0], grads))
            d3 = add(dfe[br][2], self._back_layer(f'layer3_{br}', br, d4, grads))
            dc2 = add(dc2, self._back_layer(f'layer2_{br}', br, d3, grads))
        report('fpn_mid')                                   # layer4_h's gradients are the sums over both branches: final only now
        dc1 = self._back_layer('layer1_h', 'h', dc2, grads)
        da0 = ops.lrelu_bwd(ops.maxpool_bwd(S['a0'], dc1, 3, 2, 1), S['a0'], SLOPE)
        dc0, grads['layer0_h.1.weight'], grads['layer0_h.1.bias'] = ops.bn_train_backward(S['c0'], da0, self.stem['bn']['gamma'], S['s0'])
        grads['layer0_h.0.weight'] = _unpack_grad(CB.conv2d_wgrad(S['x'], dc0, 7, 7, 2, 3), *self.shapes['layer0_h.0.weight'])
        report('fpn_end')
        return grads


# ------------------------------------------------------------------------------------------------------------------------
class EncoderTrain:
    """Training-mode forward + backward of ``Encoder`` (lib/model/encoding.py:38-73): 1x1 ``project``, 8 pre-activation
    ``Residual`` bottlenecks (:21-36: BN -> LeakyReLU -> 1x1 -> BN -> LeakyReLU -> 3x3 -> BN -> LeakyReLU -> 1x1, identity
    shortcut since numIn == numOut), a 2x2 max-pool after every second one.  Gradients under the reference's names."""

    def __init__(self, sd, prefix, device, cin_pad=None):
        from .model.pack import pack_conv
        self.dev = dev = device
        w = lambda k: sd[f'{prefix}.{k}'].detach().float()
        self.shapes = {}

        def conv(key, pad=None):
            t = w(key + '.weight')
            self.shapes[key + '.weight'] = tuple(t.shape)
            return pack_conv(t, pad).to(dev), w(key + '.bias').to(dev).contiguous()

        cin = w('project.weight').shape[1]
        self.cin, self.cin_pad = cin, (cin + 3) // 4 * 4 if cin_pad is None else cin_pad
        self.project = conv('project', self.cin_pad)
        self.blocks = []
        i = 0
        while f'{prefix}.reg.{i}.conv1.weight' in sd:
            k = f'reg.{i}'
            self.blocks.append((k, dict(bn=_bn_params(sd, f'{prefix}.{k}.bn', dev), conv1=conv(k + '.conv1'), bn1=_bn_params(sd, f'{prefix}.{k}.bn1', dev),
                                        conv2=conv(k + '.conv2'), bn2=_bn_params(sd, f'{prefix}.{k}.bn2', dev), conv3=conv(k + '.conv3'))))
            i += 1

    def forward(self, x):
        """x (N,32,32,cin_pad) NHWC (channels >= cin zero) -> encoding (N, C*2*2) in the reference's NCHW flatten order, stage maps"""
        bn = lambda p, t, f: ops.bn_train_forward(t, p['gamma'], p['beta'], p['running_mean'], p['running_var'], slope=SLOPE, partials=f)
        self.saved = dict(x=x, blocks=[], pools=[])
        fh = ops.BnFuse()                                                 # every convolution leaves the column sums of its output for the BatchNorm behind it
        h = ops.conv2d_nhwc(x, *self.project, bn=fh)
        stages = []
        for i, (k, p) in enumerate(self.blocks):
            f1, f2, fn = ops.BnFuse(), ops.BnFuse(), ops.BnFuse()
            a0, s0 = bn(p['bn'], h, fh)
            c1 = ops.conv2d_nhwc(a0, *p['conv1'], bn=f1)
            a1, s1 = bn(p['bn1'], c1, f1)
            c2 = ops.conv3x3_train(a1, *p['conv2'], bn=f2)
            a2, s2 = bn(p['bn2'], c2, f2)
            out = ops.conv2d_nhwc(a2, *p['conv3'], res=h, bn=fn)
            self.saved['blocks'].append(dict(h=h, a0=a0, s0=s0, c1=c1, a1=a1, s1=s1, c2=c2, a2=a2, s2=s2))
            h, fh = out, fn
            if i % 2 == 1:
                self.saved['pools'].append(h)
                h, fh = ops.maxpool_nhwc(h, 2, 2, 0), None               # the next BatchNorm reads the pooled map: its own pass
                stages.append(h)
        N = h.shape[0]
        return ops.nhwc_to_nchw(h).view(N, -1), stages

    def backward(self, d_encoding, d_stage1=None):
        """d_encoding (N, C*2*2) in the flatten order of forward's output; d_stage1: optional gradient reaching the second
        stage map (enc_*_ls[1], the cross modules' input).  -> d input (N,32,32,cin_pad), grads"""
        G = {}
        N = d_encoding.shape[0]
        C = self.saved['pools'][-1].shape[-1]
        sp = self.saved['pools'][-1].shape[1] // 2
        dh = d_encoding.view(N, C, sp, sp).permute(0, 2, 3, 1).contiguous()
        n_stage = len(self.saved['pools'])
        dh_colsum = None                                                  # column sums of dh when the kernel that wrote dh took them
        for i in reversed(range(len(self.blocks))):
            k, p = self.blocks[i]
            S = self.saved['blocks'][i]
            if i % 2 == 1:
                stage = i // 2
                if d_stage1 is not None and stage == 1:
                    dh = ops.add_lrelu(dh, d_stage1)
                dh, dh_colsum = ops.maxpool_bwd(self.saved['pools'][stage], dh, 2, 2, 0), None
            H, W = S['h'].shape[1:3]
            # out = conv3(a2) + h
            G[f'{k}.conv3.weight'] = _unpack_grad(CB.conv2d_wgrad(S['a2'], dh, 1, 1), *self.shapes[f'{k}.conv3.weight'])
            G[f'{k}.conv3.bias'] = dh_colsum if dh_colsum is not None else CB.conv2d_bias_grad(dh)
            fb = lambda name, c, saved: ops.BnFuse(c, saved, p[name]['gamma'], p[name]['beta'])
            f2 = fb('bn2', S['c2'], S['s2'])
            da2 = CB.conv2d_dgrad(dh, p['conv3'][0], (H, W), 1, 1, gate=(S['a2'], SLOPE), bn=f2)
            # (the bias gradient of a convolution in front of a BatchNorm = the column sums of that BatchNorm's input gradient: taken while it is written)
            dc2, G[f'{k}.bn2.weight'], G[f'{k}.bn2.bias'], G[f'{k}.conv2.bias'] = ops.bn_train_backward(S['c2'], da2, p['bn2']['gamma'], S['s2'], partials=f2, want_colsum=True)
            G[f'{k}.conv2.weight'] = _unpack_grad(CB.conv2d_wgrad(S['a1'], dc2, 3, 3, 1, 1), *self.shapes[f'{k}.conv2.weight'])
            f1 = fb('bn1', S['c1'], S['s1'])
            da1 = CB.conv2d_dgrad(dc2, p['conv2'][0], (H, W), 3, 3, 1, 1, gate=(S['a1'], SLOPE), bn=f1)
            dc1, G[f'{k}.bn1.weight'], G[f'{k}.bn1.bias'], G[f'{k}.conv1.bias'] = ops.bn_train_backward(S['c1'], da1, p['bn1']['gamma'], S['s1'], partials=f1, want_colsum=True)
            G[f'{k}.conv1.weight'] = _unpack_grad(CB.conv2d_wgrad(S['a0'], dc1, 1, 1), *self.shapes[f'{k}.conv1.weight'])
            f0 = fb('bn', S['h'], S['s0'])
            da0 = CB.conv2d_dgrad(dc1, p['conv1'][0], (H, W), 1, 1, gate=(S['a0'], SLOPE), bn=f0)
            # identity shortcut: d h = BatchNorm backward + d out, in one pass; its column sums are the bias gradient of whichever convolution
            # produced h (the previous block's conv3, or the projection)
            dh, G[f'{k}.bn.weight'], G[f'{k}.bn.bias'], dh_colsum = ops.bn_train_backward(S['h'], da0, p['bn']['gamma'], S['s0'], partials=f0, res=dh, want_colsum=True)
        x = self.saved['x']
        G['project.weight'] = _unpack_grad(CB.conv2d_wgrad(x, dh, 1, 1), *self.shapes['project.weight'])
        G['project.bias'] = dh_colsum if dh_colsum is not None else CB.conv2d_bias_grad(dh)
        dx = CB.conv2d_dgrad(dh, self.project[0], x.shape[1:3], 1, 1)
        return dx, G


# ------------------------------------------------------------------------------------------------------------------------
class HeatmapHeadTrain:
    """Training-mode forward + backward of ``HeadHeatmap2`` (lib/model/head_inplane.py:40-58,102-107): 3x3 conv, 3x3 conv +
    BatchNorm (+ LeakyReLU(True), the identity -- quirk Q1), ConvTranspose2d(4, 2, 1) + BatchNorm + ReLU, 1x1 conv.  The
    transposed convolution runs as four 2x2 phase convolutions (forward, as in inference) and their dgrad / wgrad (backward)."""
    TAP = {0: (3, 1), 1: (2, 0)}                           # kernel row used by output parity p at input offset d in {0,1}

    def __init__(self, sd, prefix, device):
        from .model.pack import pack_conv, pack_deconv4x4s2
        dev = self.dev = device
        w = lambda k: sd[f'{prefix}.{k}'].detach().float()
        self.shapes = {k: tuple(w(k).shape) for k in ('conv_layers.0.weight', 'conv_layers.1.weight', 'deconv_layers.0.weight', 'final_layer.weight')}
        self.c0 = (pack_conv(w('conv_layers.0.weight')).to(dev), w('conv_layers.0.bias').to(dev).contiguous())
        self.c1 = (pack_conv(w('conv_layers.1.weight')).to(dev), w('conv_layers.1.bias').to(dev).contiguous())
        self.bn1 = _bn_params(sd, f'{prefix}.conv_layers.2', dev)
        self.deconv = {k: (v[0].to(dev), v[1], v[2]) for k, v in pack_deconv4x4s2(w('deconv_layers.0.weight')).items()}
        self.bn2 = _bn_params(sd, f'{prefix}.deconv_layers.1', dev)
        self.final = (pack_conv(w('final_layer.weight')).to(dev), w('final_layer.bias').to(dev).contiguous())

    def forward(self, x):
        import torch
        bn = lambda p, t, slope, f=None: ops.bn_train_forward(t, p['gamma'], p['beta'], p['running_mean'], p['running_var'], slope=slope, partials=f)
        c0 = ops.conv3x3_train(x, *self.c0)
        f1 = ops.BnFuse()
        c1 = ops.conv3x3_train(c0, *self.c1, bn=f1)
        a1, s1 = bn(self.bn1, c1, 1.0, f1)
        N, H, W, _ = a1.shape
        co = self.shapes['deconv_layers.0.weight'][1]
        up = torch.empty((N, 2 * H, 2 * W, co), device=x.device)
        f2 = ops.BnFuse(parts=len(self.deconv))                       # the four phases append their column sums: no pass over the 2H x 2W map
        for (py, px), (wp, pady, padx) in self.deconv.items():
            ops.conv2d_nhwc(a1, wp, None, kh=2, kw=2, pad_y=pady, pad_x=padx, out_hw=(H, W),
                            out_view=(up, 4 * H * W * co, 4 * W * co, 2 * co, (py * 2 * W + px) * co), bn=f2)
        a2, s2 = bn(self.bn2, up, 0.0, f2)
        out = ops.conv2d_nhwc(a2, *self.final)
        self.saved = dict(x=x, c0=c0, c1=c1, a1=a1, s1=s1, up=up, a2=a2, s2=s2)
        return out

    def backward(self, dout):
        import torch
        S, G = self.saved, {}
        N, H, W, _ = S['a1'].shape
        G['final_layer.weight'] = _unpack_grad(CB.conv2d_wgrad(S['a2'], dout, 1, 1), *self.shapes['final_layer.weight'])
        G['final_layer.bias'] = CB.conv2d_bias_grad(dout)
        f2 = ops.BnFuse(S['up'], S['s2'], self.bn2['gamma'], self.bn2['beta'])
        da2 = CB.conv2d_dgrad(dout, self.final[0], S['a2'].shape[1:3], 1, 1, gate=(S['a2'], 0.0), bn=f2)
        dup, G['deconv_layers.1.weight'], G['deconv_layers.1.bias'] = ops.bn_train_backward(S['up'], da2, self.bn2['gamma'], S['s2'], partials=f2)
        cin, co = self.shapes['deconv_layers.0.weight'][:2]
        dwt = torch.zeros(self.shapes['deconv_layers.0.weight'], device=dout.device)
        da1 = None
        for (py, px), (wp, pady, padx) in self.deconv.items():
            dph = dup[:, py::2, px::2, :].contiguous()                               # this parity's outputs (N,H,W,co)
            g = CB.conv2d_wgrad(S['a1'], dph, 2, 2, 1, pad_y=pady, pad_x=padx).view(co, 2, 2, cin)
            CB.wgrad_join()                                                          # the taps are copied out on this stream
            for dy_ in (0, 1):
                for dx_ in (0, 1):
                    dwt[:, :, self.TAP[py][dy_], self.TAP[px][dx_]] = g[:, dy_, dx_, :].t()
            d = CB.conv2d_dgrad(dph, wp, (H, W), 2, 2, 1, pad_y=pady, pad_x=padx)
            da1 = d if da1 is None else ops.add_lrelu(da1, d)
        G['deconv_layers.0.weight'] = dwt
        dc1, G['conv_layers.2.weight'], G['conv_layers.2.bias'], G['conv_layers.1.bias'] = ops.bn_train_backward(S['c1'], da1, self.bn1['gamma'], S['s1'], want_colsum=True)
        G['conv_layers.1.weight'] = _unpack_grad(CB.conv2d_wgrad(S['c0'], dc1, 3, 3, 1, 1), *self.shapes['conv_layers.1.weight'])
        dc0 = CB.conv2d_dgrad(dc1, self.c1[0], S['c0'].shape[1:3], 3, 3, 1, 1)
        G['conv_layers.0.weight'] = _unpack_grad(CB.conv2d_wgrad(S['x'], dc0, 3, 3, 1, 1), *self.shapes['conv_layers.0.weight'])
        G['conv_layers.0.bias'] = CB.conv2d_bias_grad(dc0)
        dx = CB.conv2d_dgrad(dc0, self.c0[0], S['x'].shape[1:3], 3, 3, 1, 1)
        return dx, G


# ------------------------------------------------------------------------------------------------------------------------
class HeadManoTrain:
    """Training-mode forward + backward of ``HeadMano`` (lib/model/head_mano.py:30-87) with its four losses (:89-133): two
    Linear + LeakyReLU base layers, ``fc_pose`` (16 x rot6d) and ``fc_shape`` on the fp32-MFMA GEMM, then
    ``vpho_mano_train_f32`` (Gram-Schmidt -> MANO -> losses -> gradient at the 96 + 10 outputs) and the MLP backward.
    Parameters stay in the reference's (out, in) layout.  Gradients under the reference's names."""
    NAMES = ('base_layer.0', 'base_layer.2', 'fc_pose', 'fc_shape')

    def __init__(self, sd, prefix, mano, device):
        import torch
        self.dev, self.mano = device, mano
        g = lambda k: sd[f'{prefix}.{k}'].detach().float().to(device).contiguous().clone()
        self.p = {f'{n}.{s}': g(f'{n}.{s}') for n in self.NAMES for s in ('weight', 'bias')}
        self._pad = torch.zeros((2, self.p['fc_pose.weight'].shape[1]), device=device)
        self.last_outputs = None                         # (verts, joints) of the last forward (pd_dt of VPHO.py:221-222)

    def forward_backward(self, x, gt_vert, gt_joint, gt_rot6d, gt_shape, is_right, weights, is_ho3d=None):
        """x (bs,1024) encoding.  -> losses dict (weighted, 0-d fp64), d loss / d x (bs,1024), grads"""
        import torch
        from .train_score import _wgrad
        P = self.p
        h1 = ops.linear(x, P['base_layer.0.weight'], P['base_layer.0.bias'], out_slope=SLOPE)
        h2 = ops.linear(h1, P['base_layer.2.weight'], P['base_layer.2.bias'], out_slope=SLOPE)
        wcat = torch.cat([P['fc_pose.weight'], P['fc_shape.weight'], self._pad], 0).contiguous()             # (108, 512)
        bcat = torch.cat([P['fc_pose.bias'], P['fc_shape.bias'], self._pad[:, 0]], 0).contiguous()
        out = ops.linear(h2, wcat, bcat)
        rot6d, shape = out[:, :96].contiguous(), out[:, 96:106].contiguous()
        L, d6, ds, v_, j_ = self.mano.train(rot6d, shape, gt_vert, gt_joint, gt_rot6d, gt_shape, is_right, weights, want_outputs=True, is_ho3d=is_ho3d)
        self.last_outputs = (v_, j_)
        dout = torch.cat([d6, ds, torch.zeros_like(d6[:, :2])], 1).contiguous()                               # (bs, 108)
        G = {}
        dw = _wgrad(h2, dout)
        db = ops.colsum(dout)
        G['fc_pose.weight'], G['fc_shape.weight'] = dw[:96].contiguous(), dw[96:106].contiguous()
        G['fc_pose.bias'], G['fc_shape.bias'] = db[:96].contiguous(), db[96:106].contiguous()
        dh2 = ops.lrelu_bwd(ops.linear(dout, wcat.t().contiguous()), h2, SLOPE)
        G['base_layer.2.weight'], G['base_layer.2.bias'] = _wgrad(h1, dh2), ops.colsum(dh2)
        dh1 = ops.lrelu_bwd(ops.linear(dh2, P['base_layer.2.weight'].t().contiguous()), h1, SLOPE)
        G['base_layer.0.weight'], G['base_layer.0.bias'] = _wgrad(x, dh1), ops.colsum(dh1)
        dx = ops.linear(dh1, P['base_layer.0.weight'].t().contiguous())
        return L, dx, G


class CrossTrain:
    """Training-mode forward + backward of one ``CrossModule(8, 512)`` (lib/model/cross_module.py:91-137): 3x3 projections of the two
    8x8 stage maps -> (bs, 32, 512) token streams + the NeRF-embedded gravity token + sinusoidal positional code -> ONE post-norm
    ``nn.TransformerEncoderLayer`` (2 heads, FFN 2048, ReLU) whose sequence axis is the batch (quirk Q3).  The Linear / Conv2d layers
    run on the fp32-MFMA implicit-GEMM kernels (forward and dgrad) and the TN weight-gradient kernel; csrc/train_physics.hip holds
    the token / LayerNorm / attention backward.  ``p_drop``: the reference trains with its five Dropout sites at 0.1
    (PositionalEncoding, the attention probabilities, after self-attention, inside and after the FFN); the masks are Bernoulli
    draws from the device generator (same distribution as the reference's, not the same stream), applied with plain tensor
    multiplies except the attention mask, which the attention kernels apply themselves; the fixtures run every site at 0."""
    L = 'attn.layers.0'

    def __init__(self, sd, prefix, device, p_drop=0.0):
        import torch
        self.dev, self.p_drop = device, p_drop
        g = lambda k: sd[f'{prefix}.{k}'].detach().float().to(device).contiguous().clone()
        from .model.pack import pack_conv
        self.shapes = {k: tuple(sd[f'{prefix}.{k}.weight'].shape) for k in ('proj_hand', 'proj_obj')}
        self.conv = {k: [pack_conv(sd[f'{prefix}.{k}.weight'].detach().float(), None).to(device), g(f'{k}.bias')] for k in ('proj_hand', 'proj_obj')}
        self.p = {k: g(k) for k in ('gravity_proj.weight', 'gravity_proj.bias', f'{self.L}.self_attn.in_proj_weight', f'{self.L}.self_attn.in_proj_bias',
                                    f'{self.L}.self_attn.out_proj.weight', f'{self.L}.self_attn.out_proj.bias', f'{self.L}.linear1.weight',
                                    f'{self.L}.linear1.bias', f'{self.L}.linear2.weight', f'{self.L}.linear2.bias', f'{self.L}.norm1.weight',
                                    f'{self.L}.norm1.bias', f'{self.L}.norm2.weight', f'{self.L}.norm2.bias')}
        self.pe = g('pose_embedder.pe')[:, 0, :].contiguous()                 # (5000, 512) buffer
        self._gw64 = torch.nn.functional.pad(self.p['gravity_proj.weight'], (0, 1)).contiguous()       # 63 -> 64 input columns

    def _drop(self, x):
        import torch
        if self.p_drop <= 0.0:
            return x, None
        m = (torch.rand_like(x) >= self.p_drop).float() / (1.0 - self.p_drop)
        return x * m, m

    def forward(self, st_h, st_o, grav_emb):
        """st_h, st_o (bs,8,8,256) NHWC, grav_emb (bs,64) = ops.nerf_embed(gravity, flip) -> tokens (bs*65, 512)"""
        import torch
        P, L = self.p, self.L
        bs = st_h.shape[0]
        ph = ops.conv3x3_train(st_h, *self.conv['proj_hand'])
        po = ops.conv3x3_train(st_o, *self.conv['proj_obj'])
        self._gw64 = torch.nn.functional.pad(P['gravity_proj.weight'], (0, 1)).contiguous()
        ge = ops.linear(grav_emb, self._gw64, P['gravity_proj.bias'])
        x0, m0 = self._drop(ops.cross_tokens(ph, po, ge, self.pe[:bs].contiguous()).view(bs * 65, 512))
        qkv = ops.linear(x0, P[f'{L}.self_attn.in_proj_weight'], P[f'{L}.self_attn.in_proj_bias'])
        ma = None if self.p_drop <= 0.0 else ((torch.rand((65 * 2, bs, bs), device=self.dev) >= self.p_drop).float() / (1.0 - self.p_drop)).contiguous()
        att = ops.mha(qkv, bs, 65, 512, 2, drop=ma).view(bs * 65, 512)
        sa, m1 = self._drop(ops.linear(att, P[f'{L}.self_attn.out_proj.weight'], P[f'{L}.self_attn.out_proj.bias']))
        x1 = ops.add_layernorm(x0, sa, P[f'{L}.norm1.weight'], P[f'{L}.norm1.bias'])
        h, mh = self._drop(ops.linear(x1, P[f'{L}.linear1.weight'], P[f'{L}.linear1.bias'], out_slope=0.0))
        ff, m2 = self._drop(ops.linear(h, P[f'{L}.linear2.weight'], P[f'{L}.linear2.bias']))
        x2 = ops.add_layernorm(x1, ff, P[f'{L}.norm2.weight'], P[f'{L}.norm2.bias'])
        self.saved = dict(st_h=st_h, st_o=st_o, emb=grav_emb, x0=x0, qkv=qkv, att=att, sa=sa, x1=x1, h=h, ff=ff, bs=bs, masks=(m0, m1, mh, m2), attn_mask=ma)
        return x2

    def backward(self, d_x2, want_hand=True, want_obj=True):
        """d_x2 (bs*65,512) -> d st_h, d st_o (None for the stream the caller detached), grads under the reference's names"""
        from .train_score import _wgrad
        P, L, S = self.p, self.L, self.saved
        bs = S['bs']
        m0, m1, mh, m2 = S['masks']
        T = lambda w: w.t().contiguous()
        G = {}
        d_s2, gx2 = ops.layernorm_bwd(S['x1'], S['ff'], P[f'{L}.norm2.weight'], d_x2)
        G[f'{L}.norm2.weight'], G[f'{L}.norm2.bias'] = ops.colsum(gx2), ops.colsum(d_x2)
        d_ff = d_s2 if m2 is None else d_s2 * m2
        G[f'{L}.linear2.weight'], G[f'{L}.linear2.bias'] = _wgrad(S['h'], d_ff), ops.colsum(d_ff)
        d_h = ops.linear(d_ff, T(P[f'{L}.linear2.weight']))
        if mh is not None:
            d_h = d_h * mh
        d_h = ops.lrelu_bwd(d_h, S['h'], 0.0)
        G[f'{L}.linear1.weight'], G[f'{L}.linear1.bias'] = _wgrad(S['x1'], d_h), ops.colsum(d_h)
        d_x1 = ops.add_lrelu(d_s2, ops.linear(d_h, T(P[f'{L}.linear1.weight'])))
        d_s1, gx1 = ops.layernorm_bwd(S['x0'], S['sa'], P[f'{L}.norm1.weight'], d_x1)
        G[f'{L}.norm1.weight'], G[f'{L}.norm1.bias'] = ops.colsum(gx1), ops.colsum(d_x1)
        d_sa = d_s1 if m1 is None else d_s1 * m1
        G[f'{L}.self_attn.out_proj.weight'], G[f'{L}.self_attn.out_proj.bias'] = _wgrad(S['att'], d_sa), ops.colsum(d_sa)
        d_att = ops.linear(d_sa, T(P[f'{L}.self_attn.out_proj.weight']))
        dqkv = ops.mha_bwd(S['qkv'], d_att, bs, 65, 512, 2, drop=S['attn_mask'])
        G[f'{L}.self_attn.in_proj_weight'], G[f'{L}.self_attn.in_proj_bias'] = _wgrad(S['x0'], dqkv), ops.colsum(dqkv)
        d_x0 = ops.add_lrelu(d_s1, ops.linear(dqkv, T(P[f'{L}.self_attn.in_proj_weight'])))
        if m0 is not None:
            d_x0 = d_x0 * m0
        dph, dpo, dge = ops.cross_tokens_bwd(d_x0.view(bs, 65, 512))
        G['gravity_proj.weight'], G['gravity_proj.bias'] = _wgrad(S['emb'], dge)[:, :63].contiguous(), ops.colsum(dge)
        out = {}
        for key, st, dp, want in (('proj_hand', S['st_h'], dph, want_hand), ('proj_obj', S['st_o'], dpo, want_obj)):
            G[f'{key}.weight'] = _unpack_grad(CB.conv2d_wgrad(st, dp, 3, 3, 1, 1), *self.shapes[key])
            G[f'{key}.bias'] = CB.conv2d_bias_grad(dp)
            out[key] = CB.conv2d_dgrad(dp, self.conv[key][0], (8, 8), 3, 3, 1, 1) if want else None
        return out['proj_hand'], out['proj_obj'], G


class PhysicsTrain:
    """The physics branch of ``vpho_net.forward(mode='train')`` (lib/model/VPHO.py:164-172,205-212): gravity / CoM flips,
    ``cross_hand(hand maps, obj maps.detach())`` -> hand tokens, ``cross_obj(hand maps.detach(), obj maps)`` -> object tokens,
    ``HeadPhysics`` (fc_scale on the hand tokens, fc_weight / fc_CoM on the object tokens), ``from_local_to_global`` on the
    ground-truth vertices and the five losses of ``get_loss``; backward down to the two stage maps."""
    HEADS = ('fc_scale', 'fc_weight', 'fc_CoM')

    def __init__(self, sd, agg, device, p_drop=0.0):
        self.dev, self.agg = device, agg                     # agg: ops.Aggregation (ForceAnchor tables)
        self.cross = dict(hand=CrossTrain(sd, 'cross_hand', device, p_drop), obj=CrossTrain(sd, 'cross_obj', device, p_drop))
        g = lambda k: sd[f'head_physics.{k}'].detach().float().to(device).contiguous().clone()
        self.p = {f'{h}.{i}.{s}': g(f'{h}.{i}.{s}') for h in self.HEADS for i in (0, 2) for s in ('weight', 'bias')}
        self.anchor = g('anchor')

    def params(self):
        """{reference name: live tensor} of every trained tensor of the branch (conv weights: packed live + shape)"""
        out = {}
        for br, c in self.cross.items():
            for k, v in c.p.items():
                out[f'cross_{br}.{k}'] = v
            for k, (w, b) in c.conv.items():
                out[f'cross_{br}.{k}.bias'] = b
        for k, v in self.p.items():
            out[f'head_physics.{k}'] = v
        return out

    def forward_backward(self, st_h, st_o, gravity, obj_com, is_right, gt_vert, gt_force_local, is_grasped, weights):
        """st_h, st_o (bs,8,8,256) NHWC second-stage maps of the two encoders; gravity, obj_com (bs,1,3) as in the batch (un-flipped);
        gt_vert = gt_hand_vert_flip (bs,778,3); weights: (force, gravity, torque, supervised, CoM).
        -> losses dict (weighted 0-d fp64), d st_h, d st_o, grads {reference name: tensor}, force_local (bs,32,3)"""
        import torch
        from .train_score import _wgrad
        bs = st_h.shape[0]
        left = (~is_right.bool()).to(torch.uint8).contiguous()
        grav = gravity.float().reshape(bs, 3).contiguous()
        emb = ops.nerf_embed(grav, left)                                               # flip_point3d_by_mask_index + PosEmbedder (VPHO.py:167)
        sgn = torch.where(is_right.bool(), 1.0, -1.0).to(grav.dtype)[:, None]
        flip = lambda t: torch.cat([t[:, :1] * sgn, t[:, 1:]], 1).contiguous()
        grav_f, com_f = flip(grav), flip(obj_com.float().reshape(bs, 3))
        tok_h = self.cross['hand'].forward(st_h, st_o, emb).view(bs, 65, 512)
        tok_o = self.cross['obj'].forward(st_h, st_o, emb).view(bs, 65, 512)
        xh = tok_h[:, :32].reshape(bs * 32, 512).contiguous()                           # enc_phy_hand
        xo = tok_o[:, 32:64].reshape(bs * 32, 512).contiguous()                         # enc_phy_obj
        P = self.p
        hid = {h: ops.linear(x, P[f'{h}.0.weight'], P[f'{h}.0.bias'], out_slope=SLOPE) for h, x in (('fc_scale', xh), ('fc_weight', xo), ('fc_CoM', xo))}
        pad4 = lambda w, b: (torch.cat([w, w.new_zeros(4 - w.shape[0] % 4 if w.shape[0] % 4 else 0, w.shape[1])], 0).contiguous(),
                             torch.cat([b, b.new_zeros(4 - b.shape[0] % 4 if b.shape[0] % 4 else 0)], 0).contiguous())
        outs = {}
        for h in self.HEADS:
            w, b = pad4(P[f'{h}.2.weight'], P[f'{h}.2.bias'])
            outs[h] = ops.linear(hid[h], w, b)
        scale_raw = outs['fc_scale'][:, :1].contiguous()
        logits = outs['fc_weight'][:, :8].contiguous()
        com = outs['fc_CoM'][:, :3].contiguous()
        pts, frames = self.agg.anchor_frames(gt_vert.float().contiguous())
        fl, losses, d_sc, d_lg, d_cm = ops.physics_loss(scale_raw, logits, com, self.anchor, frames, pts, gt_force_local.float().contiguous(),
                                                        grav_f, com_f, is_grasped.to(torch.uint8).contiguous(), weights)
        Lz = dict(zip(('force_loss', 'gravity_loss', 'torque_loss', 'supervised_loss', 'CoM_loss'), losses.unbind(0)))
        # ---- backward: the three two-layer heads, then the two cross modules
        G = {}
        d_x = {}
        for h, x, d_out, n_out in (('fc_scale', xh, d_sc, 1), ('fc_weight', xo, d_lg, 8), ('fc_CoM', xo, d_cm, 3)):
            w, _ = pad4(P[f'{h}.2.weight'], P[f'{h}.2.bias'])
            dpad = torch.cat([d_out, d_out.new_zeros(d_out.shape[0], w.shape[0] - n_out)], 1).contiguous()
            G[f'head_physics.{h}.2.weight'] = _wgrad(hid[h], dpad)[:n_out].contiguous()
            G[f'head_physics.{h}.2.bias'] = ops.colsum(dpad)[:n_out].contiguous()
            dh = ops.lrelu_bwd(ops.linear(dpad, w.t().contiguous()), hid[h], SLOPE)
            G[f'head_physics.{h}.0.weight'], G[f'head_physics.{h}.0.bias'] = _wgrad(x, dh), ops.colsum(dh)
            d_x[h] = ops.linear(dh, P[f'{h}.0.weight'].t().contiguous())
        d_tok_h = torch.zeros((bs, 65, 512), device=self.dev)
        d_tok_h[:, :32] = d_x['fc_scale'].view(bs, 32, 512)
        d_tok_o = torch.zeros((bs, 65, 512), device=self.dev)
        d_tok_o[:, 32:64] = ops.add_lrelu(d_x['fc_weight'], d_x['fc_CoM']).view(bs, 32, 512)
        d_sth, _, gh = self.cross['hand'].backward(d_tok_h.view(bs * 65, 512), want_hand=True, want_obj=False)       # VPHO.py:170: obj maps detached
        _, d_sto, go = self.cross['obj'].backward(d_tok_o.view(bs * 65, 512), want_hand=False, want_obj=True)       # VPHO.py:171: hand maps detached
        G.update({f'cross_hand.{k}': v for k, v in gh.items()})
        G.update({f'cross_obj.{k}': v for k, v in go.items()})
        return Lz, d_sth, d_sto, G, fl.view(bs, 32, 3), dict(tok_hand=tok_h[:, :32], tok_obj=tok_o[:, 32:64], com=com.view(bs, 32, 3), scale=scale_raw.view(bs, 32))
