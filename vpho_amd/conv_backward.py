"""Backward of the NHWC convolutions (driven by ``train_blocks`` / ``train_step``).

* ``conv2d_dgrad``: the gradient w.r.t. the input of a stride-1 convolution is a stride-1 convolution of dY with the spatially
  flipped, channel-transposed weights -- the forward fp32-MFMA implicit-GEMM kernel ``vpho_conv2d_nhwc_f32``, whose epilogue also
  carries the LeakyReLU backward (``gate``) and the other branch of a residual sum (``res``); for stride 2 every output-pixel
  parity (iy%2, ix%2) only sees the taps r with r = (iy + pad) mod 2, so dX is four small convolutions of dY written to
  interleaved positions (the same phase trick as the forward ConvTranspose);
* ``conv2d_wgrad``: dW[co][(r,s,ci)] = sum_p dY[p][co] * x_gathered[p][(r,s,ci)] with the pixel index as the reduction dimension:
  ``vpho_conv2d_wgrad_nhwc_f32`` (csrc/conv_wgrad.hip), an implicit TN GEMM that stages both NHWC operands k-major and gathers
  the taps itself -- no im2col, no transposes.  Channel counts that are not multiples of 4 (the 21 / 27-channel heat-map outputs)
  take the explicit path: a transpose, ``vpho_im2col_t_f32`` and split launches of the forward kernel.

Weights are in the forward kernel's packed layout (Cout, KH*KW*Cin), activations NHWC fp32.  torch only allocates and permutes.
Reference semantics: torch.nn.functional.conv2d's autograd (what ``loss.backward()`` does for every ``nn.Conv2d`` of
lib/model/backbone_FPN_HFL.py, encoding.py, head_inplane.py under lib/engine/train_diff_hand_obj.py:181-182).
"""
import torch

from . import ops


import os

USE_TN_WGRAD = os.environ.get('VPHO_WGRAD_IM2COL', '0') == '0'      # tuning aid: 1 = the explicit im2col + transpose path for every shape


_REV = {}


def _flip_transpose(w_packed, cout, cin, kh, kw):
    """(Cout, KH*KW*Cin) -> (Cin, KH*KW*Cout) with both spatial axes reversed"""
    if kh == 1 and kw == 1:
        return w_packed.t().contiguous()                        # nothing to flip: one transposing copy
    # reversing both spatial axes = reversing the flat tap index; one gather does the reversal and the transposition
    key = (kh * kw, w_packed.device)
    rev = _REV.get(key)
    if rev is None:
        rev = _REV[key] = torch.arange(kh * kw - 1, -1, -1, device=w_packed.device)
    return w_packed.view(cout, kh * kw, cin).permute(2, 1, 0).index_select(1, rev).reshape(cin, kh * kw * cout)


def _pad_rows4(w_packed):
    """the GEMM kernel wants a reduction length (= Cout of the forward convolution) that is a multiple of 4: zero rows"""
    extra = (4 - w_packed.shape[0] % 4) % 4
    return w_packed if extra == 0 else torch.cat([w_packed, w_packed.new_zeros(extra, w_packed.shape[1])], 0)


def _phase_taps(parity, k, pad):
    # input row iy = 2a + parity receives tap r from output row a + (parity + pad - r) / 2
    rs = [r for r in range(k) if (parity + pad - r) % 2 == 0]
    return sorted(((parity + pad - r) // 2, r) for r in rs)          # (offset of the output row, tap), ascending offset


def _phase_weights(w_packed, cin, kh, kw, ty, tx):
    """weights of one output-parity phase of a stride-2 input gradient: (cin, khp*kwp*cout) from the (padded) packed weights"""
    cout = w_packed.shape[0]
    w4 = w_packed.view(cout, kh, kw, cin)
    oy0, ox0 = ty[0][0], tx[0][0]
    khp, kwp = ty[-1][0] - oy0 + 1, tx[-1][0] - ox0 + 1
    sub = torch.zeros((cin, khp, kwp, cout), device=w_packed.device, dtype=w_packed.dtype)
    for offy, r in ty:
        for offx, s in tx:
            sub[:, offy - oy0, offx - ox0, :] = w4[:, r, s, :].t()
    return sub.reshape(cin, khp * kwp * cout)


class DgradWeightCache:
    """The input-gradient convolutions read the weights in layouts of their own (transposed, taps reversed, cut into stride-2 phases,
    zero rows up to a multiple of 4).  Built per call they cost a training step ~190 small launches (a transposing copy per 1x1
    convolution, zeros + tap copies per phase).  They only change when the weights do, i.e. once per optimiser step: this cache keeps
    every derived tensor in ONE flat buffer together with an index map into the concatenated packed weights (the layout recipe run once
    on an index tensor instead of on values; index 0 = a constant zero), and ``refresh()`` rebuilds all of them with one concatenation
    and one gather.  Owned by a DiffusionTrainStep and active only inside its backward (``with cache:``); everything else builds per call.
    Staleness is checked, not assumed: the version counter of every registered weight is recorded when its layouts are built, and a
    ``get`` that meets another version (an in-place write nobody announced: an optimiser other than the step's own, a finite-difference
    probe, ``load_state_dict``) rebuilds before it answers -- ``refresh()`` after the optimiser step is a batching optimisation, not a
    correctness requirement (the kernels' raw-pointer writes bump the counters themselves, ``ops.AdamWList.step``)."""

    def __init__(self):
        self.entries, self.weights, self.offset = {}, [], {}
        self.total, self.dirty, self.flat, self.map = 0, False, None, None
        self.versions = {}                                 # (data_ptr, shape) -> tensor._version the derived layouts were built from
        self.rebuilds = 0                                  # refreshes triggered by a version mismatch (tests)

    def __enter__(self):
        global _ACTIVE
        self._outer, _ACTIVE = _ACTIVE, self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._outer

    def get(self, w_packed, key, builder):
        wk = (w_packed.data_ptr(), tuple(w_packed.shape))
        k = wk + (key,)
        e = self.entries.get(k)
        if e is not None and self.versions.get(wk) != w_packed._version:
            self.rebuilds += 1
            self.refresh()                                 # somebody wrote the weights in place since the layouts were built
            e = self.entries[k]
        if e is None:
            assert w_packed.is_contiguous()
            if wk not in self.offset:                      # the list keeps the tensor alive: its address cannot be handed out again
                self.offset[wk] = self.total
                self.weights.append(w_packed)
                self.total += w_packed.numel()
                self.versions[wk] = w_packed._version
            elif self.versions.get(wk) != w_packed._version:      # a second layout of a weight that changed since its first one was built
                self.rebuilds += 1
                self.refresh()
            off = self.offset[wk]
            idx = torch.arange(off + 1, off + 1 + w_packed.numel(), device=w_packed.device, dtype=torch.int64).view(w_packed.shape)
            m = builder(idx)
            e = self.entries[k] = dict(tensor=builder(w_packed), map=m.reshape(-1).to(torch.int32), shape=tuple(m.shape))
            self.dirty = True
        return e['tensor']

    def refresh(self):
        """the weights have changed (optimiser step / load_params): rebuild every derived layout"""
        if not self.entries:
            return
        dev = self.weights[0].device
        if self.dirty:
            self.map = torch.cat([e['map'] for e in self.entries.values()])
            self.flat = torch.empty(self.map.numel(), device=dev, dtype=self.weights[0].dtype)
            off = 0
            for e in self.entries.values():
                n = e['map'].numel()
                e['tensor'] = self.flat[off:off + n].view(e['shape'])
                off += n
            self._zero = torch.zeros(1, device=dev, dtype=self.weights[0].dtype)
            self.dirty = False
        src = torch.cat([self._zero] + [w.reshape(-1) for w in self.weights])
        torch.index_select(src, 0, self.map, out=self.flat)
        for w in self.weights:
            self.versions[(w.data_ptr(), tuple(w.shape))] = w._version


_ACTIVE = None


def _derived(w_packed, key, builder):
    return builder(w_packed) if _ACTIVE is None else _ACTIVE.get(w_packed, key, builder)


def conv2d_dgrad(dy, w_packed, in_hw, kh, kw, stride=1, pad=0, pad_y=None, pad_x=None, gate=None, res=None, rows=None, bn=None):
    """dy (N,OH,OW,Cout), w_packed (Cout, kh*kw*Cin) -> dx (N,H,W,Cin) for y = conv2d(x, w, stride, pad) (pad_y / pad_x: the
    asymmetric top/left paddings of the transposed-convolution phases, stride 1 only).  ``gate`` = (y, slope): the result is
    additionally passed through the backward of the LeakyReLU that produced y = lrelu(x) (fused into the kernel's epilogue).
    ``res`` (stride 1): a gradient of the same shape added to the result (the other branch of a residual sum).
    ``bn`` (with gate; an ``ops.BnFuse`` holding the BatchNorm input / statistics / affine whose activation produced y): where the
    kernel of the shape allows (stride 1), the gate is recomputed from the BatchNorm input and the two sums of the BatchNorm backward are
    taken in the epilogue (``bn.live()`` afterwards); elsewhere ``gate`` is used as is."""
    N, OH, OW, cout = dy.shape
    H, W = in_hw
    cin = w_packed.shape[1] // (kh * kw)
    if kh == 3 and kw == 3 and stride == 1 and pad == 1 and pad_y is None and pad_x is None and res is None and (OH, OW) == (H, W) and dy.is_contiguous():
        # Winograd F(2x2,3x3) on the flipped / transposed weights where the shape allows; rows (RoiWindows dilated by the halo): dy is
        # zero outside the windows, so only their pixels are computed (the rest of dx is zero)
        dx = ops.conv3x3_dgrad_winograd(dy, w_packed, gate, rows=rows, bn=bn)
        if dx is not None:
            return dx
    if cout % 4:                                          # the GEMM kernel wants a reduction length that is a multiple of 4
        extra = 4 - cout % 4
        dy = torch.nn.functional.pad(dy, (0, extra)).contiguous()
        cout += extra
    py = pad if pad_y is None else pad_y
    px = pad if pad_x is None else pad_x
    if stride == 1:
        wt = _derived(w_packed, ('s1', kh, kw), lambda w, c=cout: _flip_transpose(_pad_rows4(w), c, cin, kh, kw))
        return ops.conv2d_nhwc(dy, wt, None, kh=kh, kw=kw, stride=1, pad_x=kw - 1 - px, pad_y=kh - 1 - py, out_hw=(H, W), gate=gate, res=res,
                               bn=bn if (gate is not None and bn is not None and bn.x is not None) else None)
    assert py == px == pad and res is None
    assert stride == 2 and H % 2 == 0 and W % 2 == 0, 'stride 1 or 2 (even input size)'
    phases = [(py, px) for py in (0, 1) for px in (0, 1) if _phase_taps(py, kh, pad) and _phase_taps(px, kw, pad)]
    # all four parities have taps (3x3): every pixel of dx is written by its phase; fewer (1x1 / stride 2: one): the others stay zero
    dx = (torch.empty if len(phases) == 4 else torch.zeros)((N, H, W, cin), device=dy.device, dtype=dy.dtype)
    fuse = gate is not None and bn is not None and bn.x is not None
    if fuse:
        bn.parts = len(phases)                              # every phase appends its partial sums (the pixels of the other parities stay zero: no terms)
    for py in (0, 1):
        ty = _phase_taps(py, kh, pad)
        for px in (0, 1):
            tx = _phase_taps(px, kw, pad)
            if not ty or not tx:
                continue
            oy0, ox0 = ty[0][0], tx[0][0]
            khp, kwp = ty[-1][0] - oy0 + 1, tx[-1][0] - ox0 + 1
            sub = _derived(w_packed, ('s2', kh, kw, pad, py, px), lambda w, ty=ty, tx=tx: _phase_weights(_pad_rows4(w), cin, kh, kw, ty, tx))
            # phase output (a, b) reads dY rows a + oy0 .. : a stride-1 convolution with padding -oy0 / -ox0
            ops.conv2d_nhwc(dy, sub, None, kh=khp, kw=kwp, stride=1, pad_y=-oy0, pad_x=-ox0,
                            out_hw=(H // 2, W // 2), out_view=(dx, H * W * cin, 2 * W * cin, 2 * cin, (py * W + px) * cin), gate=gate,
                            bn=bn if fuse else None)
    return dx


class WgradStream:
    """Weight gradients on a stream of their own.  A layer's weight gradient needs only dY and the saved input and nothing of the
    backward waits for it before the optimiser, while the chain dY -> BatchNorm backward -> input gradient -> next layer is strictly
    sequential: many of its launches (partial-sum reductions, finishing kernels, the short tails of small layers) leave most of the
    chip idle.  Inside ``with WgradStream(device):`` conv2d_wgrad queues its kernels on a second HIP stream behind everything the main
    stream has queued so far (so dY is complete), and ``join()`` makes the main stream wait for them -- called before gradients are
    handed on.  Every kernel is the same and deterministic; only the overlap changes.  Tensors crossing streams are registered with
    the caching allocator (record_stream) so that their memory is not reused while the other stream still works on it."""
    _streams = {}

    def __init__(self, device):
        self.dev = torch.device(device)
        key = (self.dev.type, self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        if key not in WgradStream._streams:
            WgradStream._streams[key] = torch.cuda.Stream(self.dev)
        self.side = WgradStream._streams[key]
        self.pending = False

    def __enter__(self):
        global _WGRAD_STREAM
        self._outer = _WGRAD_STREAM
        if os.environ.get('VPHO_WGRAD_STREAM', '1') != '0':        # 0: everything on the main stream (A/B aid)
            _WGRAD_STREAM = self
        return self

    def __exit__(self, *exc):
        global _WGRAD_STREAM
        self.join()
        _WGRAD_STREAM = self._outer

    def run(self, fn, inputs):
        main = torch.cuda.current_stream(self.dev)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            out = fn()
        for t in inputs:
            if t is not None:
                t.record_stream(self.side)
        out.record_stream(main)                             # consumed on the main stream after join()
        self.pending = True
        return out

    def join(self):
        if self.pending:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self.pending = False


_WGRAD_STREAM = None


def wgrad_join():
    if _WGRAD_STREAM is not None:
        _WGRAD_STREAM.join()


def on_wgrad_stream(fn, inputs):
    """fn() on the weight-gradient stream when one is active (see WgradStream), else right here"""
    return fn() if _WGRAD_STREAM is None else _WGRAD_STREAM.run(fn, inputs)


def conv2d_wgrad(x, dy, kh, kw, stride=1, pad=0, cin=None, pad_y=None, pad_x=None, groups=None):
    """x (N,H,W,ld), dy (N,OH,OW,Cout) -> dW (Cout, kh*kw*Cin) in the packed layout of the forward weights.
    ``groups`` (ops.window_groups of the output map): dy is zero outside those 32-pixel groups -- only they are reduced."""
    if _WGRAD_STREAM is not None and not getattr(_WGRAD_STREAM, '_inside', False):
        ws = _WGRAD_STREAM
        ws._inside = True
        try:
            extra = [] if groups is None else [t for t in (groups if isinstance(groups, (tuple, list)) else (groups,)) if torch.is_tensor(t)]
            return ws.run(lambda: conv2d_wgrad(x, dy, kh, kw, stride, pad, cin, pad_y, pad_x, groups), [x, dy] + extra)
        finally:
            ws._inside = False
    N, OH, OW, cout = dy.shape
    c_in = x.shape[-1] if cin is None else cin
    small = x.numel() * 4 < 3.9e9 and dy.numel() * 4 < 3.9e9          # the TN kernel addresses its operands by 32-bit byte offsets
    if USE_TN_WGRAD and small and cout % 4 and c_in % 4 == 0 and x.shape[-1] % 4 == 0 and groups is None:
        # 21 / 27 output channels (the heat-map heads): zero channels up to a multiple of 4 and the TN kernel, instead of the explicit
        # transpose + im2col path (0.22 -> 0.05 ms per head); the rows of the padding channels are dropped
        dyp = torch.nn.functional.pad(dy, (0, 4 - cout % 4))
        return conv2d_wgrad(x, dyp, kh, kw, stride, pad, cin, pad_y, pad_x)[:cout]
    if USE_TN_WGRAD and small and c_in % 4 == 0 and cout % 4 == 0 and x.shape[-1] % 4 == 0:
        return ops.conv2d_wgrad_nhwc(x, dy.contiguous(), kh, kw, stride, pad if pad_y is None else pad_y, pad if pad_x is None else pad_x, cin=cin,
                                     groups=groups)
    xg_t = ops.im2col_t(x, kh, kw, stride, pad if pad_y is None else pad_y, pad if pad_x is None else pad_x, OH, OW, cin=cin)   # (kh*kw*cin, P4)
    dy_t = ops.transpose(dy.reshape(N * OH * OW, cout))                          # (Cout, P4), zero-padded columns
    kc, P4 = xg_t.shape
    # The reduction runs over the pixels (P ~ 1e5) while the result is only Cout x kh*kw*Cin: split the pixel range over
    # blockIdx.y until the launch has a few tiles per CU, then add the partial results in a fixed order
    tiles = ((cout + 127) // 128) * ((kc + 127) // 128)
    splits = 1
    while tiles * splits < 512 and P4 // (splits * 2) >= 2048 and (P4 // (splits * 2)) % 4 == 0 and P4 % (splits * 2) == 0:
        splits *= 2
    if splits == 1:
        return ops.linear(dy_t, xg_t)
    ks = P4 // splits
    part = torch.empty((splits * cout, kc), device=x.device, dtype=x.dtype)
    ops.conv2d_nhwc(dy_t.view(cout, 1, 1, P4), xg_t, None, cin=ks, out=part.view(splits * cout, 1, 1, kc)[:cout],
                    split=(splits, P4, ks, ks, cout * kc))
    return ops.sum_repeats(part, 0, cout, splits, kc)


def conv2d_bias_grad(dy):
    N, OH, OW, cout = dy.shape
    return ops.colsum(dy.reshape(N * OH * OW, cout))
