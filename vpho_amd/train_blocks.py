"""Training-mode forward + backward of a ResNet bottleneck (SURVEY.md 8f row 4 building block): the composition of the
convolution, BatchNorm(train) and LeakyReLU kernels for ``Bottleneck.forward`` (lib/model/backbone_FPN_HFL.py:330-350) under
``model.train()`` and ``loss.backward()``.  Activations NHWC fp32, weights in the forward kernel's packed layout
(Cout, KH*KW*Cin); torch only allocates.  Not yet driven by a full training step."""
from . import ops
from . import conv_backward as CB

SLOPE = 0.01                                             # nn.LeakyReLU() default (backbone_FPN_HFL.py:326)


class BottleneckTrain:
    """params: dict with packed conv weights 'conv1','conv2','conv3' (+'down') and BatchNorm ('bn1','bn2','bn3' (+'bnd')) each a dict
    gamma / beta / running_mean / running_var (device tensors, running stats updated in place)."""

    def __init__(self, params, stride=1):
        self.p, self.stride = params, stride

    def _bn(self, name, x, slope):
        b = self.p[name]
        return ops.bn_train_forward(x, b['gamma'], b['beta'], b['running_mean'], b['running_var'], slope=slope)

    def forward(self, x):
        p, s = self.p, self.stride
        c1 = ops.conv2d_nhwc(x, p['conv1'])
        a1, s1 = self._bn('bn1', c1, SLOPE)
        c2 = ops.conv2d_nhwc(a1, p['conv2'], kh=3, kw=3, stride=s, pad=1)
        a2, s2 = self._bn('bn2', c2, SLOPE)
        c3 = ops.conv2d_nhwc(a2, p['conv3'])
        b3, s3 = self._bn('bn3', c3, 1.0)
        if 'down' in p:
            cd = ops.conv2d_nhwc(x, p['down'], stride=s)
            res, sd = self._bn('bnd', cd, 1.0)
        else:
            cd, res, sd = None, x, None
        out = ops.add_lrelu(b3, res, SLOPE)
        self.saved = dict(x=x, c1=c1, a1=a1, s1=s1, c2=c2, a2=a2, s2=s2, c3=c3, s3=s3, cd=cd, sd=sd, out=out)
        return out

    def backward(self, dout):
        """-> dx, grads {name: tensor} for every conv weight (packed layout) and BatchNorm gamma/beta"""
        p, s, S = self.p, self.stride, self.saved
        H, W = S['x'].shape[1:3]
        g = {}
        dsum = ops.lrelu_bwd(dout, S['out'], SLOPE)
        dc3, g['bn3.gamma'], g['bn3.beta'] = ops.bn_train_backward(S['c3'], dsum, p['bn3']['gamma'], S['s3'])
        g['conv3'] = CB.conv2d_wgrad(S['a2'], dc3, 1, 1)
        da2 = ops.lrelu_bwd(CB.conv2d_dgrad(dc3, p['conv3'], S['a2'].shape[1:3], 1, 1), S['a2'], SLOPE)
        dc2, g['bn2.gamma'], g['bn2.beta'] = ops.bn_train_backward(S['c2'], da2, p['bn2']['gamma'], S['s2'])
        g['conv2'] = CB.conv2d_wgrad(S['a1'], dc2, 3, 3, s, 1)
        da1 = ops.lrelu_bwd(CB.conv2d_dgrad(dc2, p['conv2'], (H, W), 3, 3, s, 1), S['a1'], SLOPE)
        dc1, g['bn1.gamma'], g['bn1.beta'] = ops.bn_train_backward(S['c1'], da1, p['bn1']['gamma'], S['s1'])
        g['conv1'] = CB.conv2d_wgrad(S['x'], dc1, 1, 1)
        dx = CB.conv2d_dgrad(dc1, p['conv1'], (H, W), 1, 1)
        if 'down' in p:
            dcd, g['bnd.gamma'], g['bnd.beta'] = ops.bn_train_backward(S['cd'], dsum, p['bnd']['gamma'], S['sd'])
            g['down'] = CB.conv2d_wgrad(S['x'], dcd, 1, 1, s, 0)
            dx = ops.add_lrelu(dx, CB.conv2d_dgrad(dcd, p['down'], (H, W), 1, 1, s, 0))
        else:
            dx = ops.add_lrelu(dx, dsum)
        return dx, g
