"""ctypes binding of the C ABI in ``include/vpho_hip.h`` (``vpho_amd/libvpho_hip.so``).

Thin by design: tensors in, raw device pointers + the current HIP stream out.  There is NO fallback: importing this
module without the built extension raises, and every op checks that its tensors live on a GPU.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VPHO_HIP_LIB') or os.path.join(_HERE, 'libvpho_hip.so')   # override: A/B kernel builds

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} is missing: build the HIP extension first (python -m vpho_amd.build or '
                      f'__graft_entry__.build()); vpho_amd has no CPU fallback')
lib = C.CDLL(LIB_PATH)
lib.vpho_last_error.restype = C.c_char_p
lib.vpho_abi_version.restype = C.c_int
ABI_VERSION = 12                        # include/vpho_hip.h; a stale library must not be found out by a missing symbol halfway through a run
if lib.vpho_abi_version() != ABI_VERSION:
    raise ImportError(f'{LIB_PATH} implements ABI version {lib.vpho_abi_version()}, this binding expects {ABI_VERSION}: rebuild the '
                      f'extension (python -m vpho_amd.build --force)')
lib.vpho_obj_metrics_workspace_bytes.restype = C.c_longlong
lib.vpho_bn_workspace_bytes.restype = C.c_longlong
lib.vpho_conv2d_wgrad_workspace_bytes.restype = C.c_longlong
lib.vpho_mha_bwd_workspace_bytes.restype = C.c_longlong


class VphoError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise VphoError(lib.vpho_last_error().decode())


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise VphoError('vpho_amd ops take GPU tensors only (no CPU path)')
    if dtype is not None and t.dtype != dtype:
        raise VphoError(f'expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise VphoError('expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------- conv / linear
class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('bias', C.c_void_p), ('in_scale', C.c_void_p),
                ('in_shift', C.c_void_p), ('res', C.c_void_p), ('y', C.c_void_p),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('x_ld', C.c_int),
                ('Cout', C.c_int), ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad_y', C.c_int),
                ('pad_x', C.c_int), ('OH', C.c_int), ('OW', C.c_int),
                ('y_sn', C.c_longlong), ('y_sy', C.c_longlong), ('y_sx', C.c_longlong),
                ('r_sn', C.c_longlong), ('r_sy', C.c_longlong), ('r_sx', C.c_longlong),
                ('in_slope', C.c_float), ('out_slope', C.c_float),
                ('w_ld', C.c_int), ('splits', C.c_int), ('x_split', C.c_longlong), ('w_split', C.c_longlong), ('y_split', C.c_longlong),
                ('gate', C.c_void_p), ('gate_slope', C.c_float),
                ('row_map', C.c_void_p), ('row_count', C.c_void_p), ('rows_hint', C.c_int), ('rows_scatter', C.c_int),
                ('w_planes', C.c_void_p), ('plane_terms', C.c_int),
                ('res_up', C.c_void_p), ('ru_H', C.c_int), ('ru_W', C.c_int), ('ru_ld', C.c_int),
                ('x2', C.c_void_p), ('Cin2', C.c_int), ('x2_ld', C.c_int), ('stride2', C.c_int), ('H2', C.c_int), ('W2', C.c_int),
                ('groups', C.c_int), ('x_group', C.c_longlong), ('w_group', C.c_longlong), ('bias_group', C.c_longlong), ('y_group', C.c_longlong),
                ('res_group', C.c_longlong), ('x2_group', C.c_longlong), ('pre_group', C.c_longlong), ('ru_group', C.c_longlong),
                ('stats', C.c_void_p), ('stats_rows', C.c_void_p), ('stats_cap', C.c_int),
                ('bn_x', C.c_void_p), ('bn_mean', C.c_void_p), ('bn_invstd', C.c_void_p), ('bn_gamma', C.c_void_p), ('bn_beta', C.c_void_p)]


lib.vpho_conv2d_nhwc_f32.argtypes = [C.POINTER(ConvDesc), C.c_void_p]


def _addr(t):
    return None if t is None else t.data_ptr()


FUSE_BN = os.environ.get('VPHO_TRAIN_FUSE_BN', '1') != '0'     # A/B aid: 0 = every train-mode BatchNorm runs its own reduction pass


class BnFuse:
    """Carrier of the reductions of a train-mode BatchNorm taken in the epilogue of the convolution next to it (vpho_conv_desc.stats).
    Forward: ``f = BnFuse(); c = conv2d_nhwc(x, w, bn=f); a, saved = bn_train_forward(c, ..., partials=f)`` -- the convolution leaves the
    per-tile sums of its output in ``f.stats`` (``f.rows`` partial rows; 0 = the kernel of that shape has no such epilogue and the
    BatchNorm runs its own pass).  Backward: ``f = BnFuse(c, saved, gamma, beta); da = conv2d_dgrad(dy, w, ..., gate=(a, slope), bn=f)``
    -- the input-gradient convolution recomputes the activation's sign from the BatchNorm input ``c`` instead of reading the stored
    activation and leaves sum da | sum da * xhat -- then ``bn_train_backward(c, da, gamma, saved, partials=f)``."""
    __slots__ = ('stats', 'rows', 'x', 'mean', 'invstd', 'gamma', 'beta', 'stored_gate', 'parts', 'calls', 'failed')

    def __init__(self, x=None, saved=None, gamma=None, beta=None, stored_gate=False, parts=1):
        # stored_gate: keep reading the activation's sign from the stored activation (a residual block's closing activation, whose input is
        # BatchNorm + shortcut) and take only xhat from the BatchNorm input.
        # parts > 1: the map is written by several convolutions, each into its own strided part (the four phases of a transposed convolution
        # / of a stride-2 input gradient, ``out_view``): every launch appends its partial rows; the sums count only when all parts came
        self.stats, self.rows, self.x, self.gamma, self.beta, self.stored_gate = None, 0, x, gamma, beta, stored_gate
        self.mean, self.invstd = saved if saved is not None else (None, None)
        self.parts, self.calls, self.failed = parts, 0, False

    def live(self):
        return self.stats is not None and self.rows > 0 and not self.failed and self.calls == self.parts


# Opt-in split-bf16 convolutions (VPHO_CONV_MFMA=bf16x6|bf16x9, inference plan only; default: fp32 MFMA).  The three bf16 planes of
# a packed weight matrix are made on first use and kept on the tensor object.
import threading
_conv_split = threading.local()


class conv_split:
    """context: convolutions launched inside use split-bf16 products (terms = 6 or 9) where the kernel takes the shape"""
    def __init__(self, terms):
        self.terms = terms

    def __enter__(self):
        self.prev = getattr(_conv_split, 'terms', 0)
        _conv_split.terms = self.terms

    def __exit__(self, *exc):
        _conv_split.terms = self.prev


def bf16_planes(w):
    """(3, *w.shape) bf16 with planes[0] + planes[1] + planes[2] == w exactly (each piece the bf16 rounding of what is left)"""
    h = w.to(torch.bfloat16)
    m = (w - h.float()).to(torch.bfloat16)
    l = (w - h.float() - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, l], 0).contiguous()


def _planes_of(w):
    """the planes live ON the weight tensor object (an address-keyed cache would hand a new tensor at a recycled address the old
    tensor's planes); an in-place update of the weights (``_version``) re-splits them"""
    c = getattr(w, '_vpho_planes', None)
    if c is None or c[0] != w._version:
        c = (w._version, bf16_planes(w))
        w._vpho_planes = c
    return c[1]


def conv2d_nhwc(x, w, bias=None, *, kh=1, kw=1, stride=1, pad=0, pad_y=None, pad_x=None, out=None, out_hw=None,
                out_view=None, res=None, in_scale=None, in_shift=None, in_slope=1.0, out_slope=1.0, cin=None, split=None, gate=None,
                rows=None, rows_scatter=False, res_up=None, x2=None, stride2=1, groups=1, x_shared=False, x2_shared=False, bn=None):
    """x: (N,H,W,x_ld) fp32 NHWC, w: (Cout, kh*kw*Cin) packed.  Returns (N,OH,OW,Cout) (or writes ``out``).
    ``groups`` = G > 1: G convolutions of one shape in ONE launch (the twin hand / object branches, vpho_conv_desc.groups): w (G,Cout,K),
    bias (G,Cout), in_scale / in_shift (G,Cin); x, res, x2, res_up and the result hold the groups' images one after the other
    ((G*N,...): group g = images [g*N, (g+1)*N)); ``x_shared`` / ``x2_shared``: that input is ONE (N,...) tensor read by every group.
    ``res_up`` = a coarser (N,h,w,Cout) map added after bilinear up-sampling to the output grid (the FPN's top-down add fused into the
    lateral convolution, backbone_FPN_HFL.py:66-68; bit-identical to ``resize_bilinear_nhwc(..., accumulate=True)`` after the convolution).
    ``x2`` = a second (N,H2,W2,C2) input of a 1x1 convolution, read at stride ``stride2`` and concatenated behind x along the channels
    (w: (Cout, Cin + C2)): the projection shortcut of a bottleneck merged into conv3 (vpho_conv_desc.x2).

    ``out_view`` = (tensor, y_sn, y_sy, y_sx, element_offset) writes into a strided destination (concat buffers,
    transposed-convolution phases).  ``res`` is a contiguous (N,OH,OW,Cout) tensor.  ``gate`` = (tensor shaped like the
    destination, slope): y = gate > 0 ? y : slope * y (LeakyReLU backward fused into an input-gradient convolution).
    ``rows`` = a ``RoiWindows``: only the listed output pixels are computed and the result is the COMPACT (N*OH*OW, Cout)
    matrix whose first ``rows.count`` rows are live (``roi_align_nhwc(..., win=rows)`` reads it); with ``rows_scatter`` the listed
    pixels are written at their own positions of the ordinary (N,OH,OW,Cout) output and the other pixels are left untouched.
    ``bn`` = a ``BnFuse``: the reductions of the train-mode BatchNorm next to this convolution in its epilogue (see BnFuse; with
    ``bn.x`` the gate is recomputed from the BatchNorm input and ``gate`` -- which the caller still passes -- is used only where the
    fused epilogue does not apply)."""
    N, H, W, x_ld = x.shape
    G = groups
    if G > 1:
        assert w.dim() == 3 and w.shape[0] == G and w.is_contiguous() and (bias is None or (bias.shape == (G, w.shape[1]) and bias.is_contiguous()))
        assert rows is None and split is None and gate is None and x.is_contiguous()
        if not x_shared:
            assert N % G == 0, (N, G)
            N //= G
    cin = x_ld if cin is None else cin
    cout = w.shape[-2]
    assert split is not None or w.shape[-1] == kh * kw * cin + (0 if x2 is None else x2.shape[-1]), (w.shape, kh, kw, cin)
    py = pad if pad_y is None else pad_y
    px = pad if pad_x is None else pad_x
    if out_hw is None:
        OH, OW = (H + 2 * py - kh) // stride + 1, (W + 2 * px - kw) // stride + 1
    else:
        OH, OW = out_hw
    d = ConvDesc()
    d.x, d.w, d.bias = _addr(x), _addr(w), _addr(bias)
    d.in_scale, d.in_shift, d.res = _addr(in_scale), _addr(in_shift), _addr(res)
    for t in (x, w, bias, in_scale, in_shift, res):
        _ptr(t, torch.float32)
    if rows is not None:
        assert out_view is None and res is None and gate is None and rows.shape == (N, OH, OW) and (res_up is None or rows_scatter)
        if out is None:
            out = torch.empty((N, OH, OW, cout) if rows_scatter else (N * OH * OW, cout), device=x.device, dtype=torch.float32)
        _ptr(out, torch.float32)
        ld = out.shape[-1]
        d.y, d.y_sx, d.y_sy, d.y_sn = out.data_ptr(), ld, (ld * OW if rows_scatter else 0), (ld * OW * OH if rows_scatter else 0)
        d.rows_scatter = 1 if rows_scatter else 0
        d.row_map, d.row_count = rows.row_map.data_ptr(), rows.count.data_ptr()
        d.rows_hint = int(rows.count.item()) if _prof_on else 0      # the instrumented pass may synchronise; the product path never does
        ret = out
    elif out_view is not None:
        yt, d.y_sn, d.y_sy, d.y_sx, off = out_view
        _ptr(yt, torch.float32)
        d.y = yt.data_ptr() + 4 * off
        ret = yt
        if gate is not None:
            assert gate[0].shape == yt.shape and gate[0].is_contiguous() and yt.is_contiguous()
            d.gate, d.gate_slope = _ptr(gate[0], torch.float32).value + 4 * off, gate[1]
    else:
        if out is None:
            out = torch.empty((G * N, OH, OW, cout), device=x.device, dtype=torch.float32)
        _ptr(out, torch.float32)
        assert out.shape[0] == G * N
        ld = out.shape[-1]
        d.y, d.y_sx, d.y_sy, d.y_sn = out.data_ptr(), ld, ld * OW, ld * OW * OH
        ret = out
        if gate is not None:
            assert gate[0].shape == out.shape
            d.gate, d.gate_slope = _ptr(gate[0], torch.float32).value, gate[1]
    if res is not None:
        assert res.shape == (G * N, OH, OW, cout)
        d.r_sx, d.r_sy, d.r_sn = cout, cout * OW, cout * OW * OH
    if x2 is not None:
        assert kh == kw == 1 and rows is None and split is None and in_scale is None and x2.shape[0] == (N if (G == 1 or x2_shared) else G * N) and x2.is_contiguous()
        d.x2, d.Cin2, d.x2_ld, d.stride2, d.H2, d.W2 = _ptr(x2, torch.float32).value, x2.shape[3], x2.shape[3], stride2, x2.shape[1], x2.shape[2]
    if res_up is not None:
        assert res is None and res_up.shape[0] == G * N and res_up.shape[3] == cout and res_up.is_contiguous()
        d.res_up, d.ru_H, d.ru_W, d.ru_ld = _ptr(res_up, torch.float32).value, res_up.shape[1], res_up.shape[2], res_up.shape[3]
    if G > 1:
        assert in_scale is None or (in_scale.shape == (G, cin) and in_shift.shape == (G, cin) and in_scale.is_contiguous() and in_shift.is_contiguous())
        d.groups = G
        d.x_group = 0 if x_shared else N * H * W * x_ld
        # a strided destination (out_view: transposed-convolution phases) holds the groups' images one after the other as well
        d.w_group, d.bias_group, d.y_group = w.shape[1] * w.shape[2], cout, (ret.numel() // G if out_view is not None else N * OH * OW * ret.shape[-1])
        d.res_group = N * OH * OW * cout
        d.x2_group = 0 if (x2 is None or x2_shared) else N * x2.shape[1] * x2.shape[2] * x2.shape[3]
        d.pre_group = cin
        d.ru_group = 0 if res_up is None else N * res_up.shape[1] * res_up.shape[2] * res_up.shape[3]
    d.N, d.H, d.W, d.Cin, d.x_ld = N, H, W, cin, x_ld
    d.Cout, d.KH, d.KW, d.stride, d.pad_y, d.pad_x, d.OH, d.OW = cout, kh, kw, stride, py, px, OH, OW
    d.in_slope, d.out_slope = in_slope, out_slope
    if split is not None:                                   # (splits, w_ld, x_split, w_split, y_split): see vpho_conv_desc
        d.splits, d.w_ld, d.x_split, d.w_split, d.y_split = split
    terms = getattr(_conv_split, 'terms', 0)
    if terms and G == 1 and split is None and in_scale is None and gate is None and res_up is None and x2 is None and cin % 16 == 0 and w.is_contiguous() and w.shape[1] == kh * kw * cin:
        planes = _planes_of(w)                              # kept alive by the cache
        d.w_planes, d.plane_terms = planes.data_ptr(), terms
    rows_out = None
    if bn is not None and FUSE_BN and G == 1 and split is None and rows is None and in_scale is None and (out_view is None or bn.parts > 1) and not d.w_planes and cout % 4 == 0:
        cap = (N * OH * OW + 63) // 64                          # the smallest M-tile is 64 rows
        if bn.stats is None:
            bn.stats = torch.empty((cap * bn.parts, 2, cout), device=x.device, dtype=torch.float32)
        base = bn.rows                                          # parts > 1: this launch appends behind the rows of the earlier parts
        assert base + cap <= bn.stats.shape[0] and bn.stats.shape[2] == cout
        rows_out = C.c_int(0)
        d.stats, d.stats_cap, d.stats_rows = bn.stats.data_ptr() + base * 2 * cout * 4, cap, C.cast(C.pointer(rows_out), C.c_void_p)
        if bn.x is not None:
            assert gate is not None and bn.x.shape == ret.shape and bn.x.is_contiguous() and ret.is_contiguous()
            bx = _ptr(bn.x, torch.float32).value + (4 * out_view[4] if out_view is not None else 0)      # laid out like y, like the gate
            if bn.stored_gate:
                d.bn_x, d.bn_mean, d.bn_invstd = bx, _ptr(bn.mean, torch.float32).value, _ptr(bn.invstd, torch.float32).value
            else:
                d.gate, d.gate_slope = None, gate[1]
                d.bn_x = bx
                d.bn_mean, d.bn_invstd, d.bn_gamma, d.bn_beta = (_ptr(t, torch.float32).value for t in (bn.mean, bn.invstd, bn.gamma, bn.beta))
    elif bn is not None:
        bn.failed = True
    _check(lib.vpho_conv2d_nhwc_f32(C.byref(d), _stream()))
    if rows_out is not None:
        bn.calls += 1
        if rows_out.value == 0:
            bn.failed = True
        bn.rows += rows_out.value
    return ret


def conv3x3(x, w, bias=None, out_slope=1.0, winograd=True, rows=None, groups=1, x_shared=False):
    """3x3 / stride 1 / pad 1 convolution + bias + LeakyReLU of the inference plan: the Winograd F(2x2,3x3) kernel where its shape
    conditions hold (even H and W, Cin % 16 == 0, Cout % 64 == 0), else the direct implicit GEMM.  The transformed weights live on the
    weight tensor object and follow its version.  ``rows`` = a ``RoiWindows``: only the window pixels, as the compact (N*H*W, Cout)
    matrix (see conv2d_nhwc).  ``groups``: see conv2d_nhwc (w (G,Cout,9*Cin))."""
    N, H, W, x_ld = x.shape
    cout, k9 = w.shape[-2:]
    cin = k9 // 9
    if not winograd or H % 2 or W % 2 or cin % 16 or cout % 64 or x_ld != cin:
        return conv2d_nhwc(x, w, bias, kh=3, kw=3, pad=1, out_slope=out_slope, rows=rows, groups=groups, x_shared=x_shared)
    c = getattr(w, '_vpho_wino', None)
    if c is None or c[0] != w._version:
        from .model.pack import winograd_weights
        c = (w._version, winograd_weights(w) if groups == 1 else torch.stack([winograd_weights(w[g]) for g in range(groups)]).contiguous())
        w._vpho_wino = c
    return conv3x3_winograd(x, c[1], bias, out_slope, rows=rows, groups=groups, x_shared=x_shared)


def conv3x3_winograd(x, u, bias=None, out_slope=1.0, out=None, rows=None, groups=1, x_shared=False):
    """3x3 / stride 1 / pad 1 convolution in Winograd F(2x2,3x3) form; u = pack.winograd_weights(packed weights) (Cin/8, 16, Cout, 8);
    groups = G > 1: u (G, Cin/8, 16, Cout, 8), bias (G, Cout), x (G*N,H,W,Cin) or one shared (N,H,W,Cin) -> (G*N,H,W,Cout)"""
    N, H, W, x_ld = x.shape
    cout, cin = u.shape[-2], u.shape[-4] * 8
    assert u.shape[-3] == 16 and u.shape[-1] == 8
    if groups > 1:
        assert rows is None and u.dim() == 5 and u.shape[0] == groups and u.is_contiguous() and x.is_contiguous() and (bias is None or bias.shape == (groups, cout))
        if not x_shared:
            assert N % groups == 0
            N //= groups
        if out is None:
            out = torch.empty((groups * N, H, W, cout), device=x.device, dtype=torch.float32)
        _call('vpho_conv3x3_winograd_grouped_nhwc_f32', _f32(x), C.c_longlong(0 if x_shared else N * H * W * x_ld), _f32(u), _f32(bias), I(groups), I(N), I(H), I(W),
              I(cin), I(x_ld), I(cout), F(out_slope), _f32(out), I(out.shape[-1]))
        return out
    if rows is not None:
        assert rows.shape == (N, H, W)
        if out is None:
            out = torch.empty((N * H * W, cout), device=x.device, dtype=torch.float32)
        _call('vpho_conv3x3_winograd_rows_nhwc_f32', _f32(x), _f32(u), _f32(bias), I(N), I(H), I(W), I(cin), I(x_ld), I(cout), F(out_slope),
              _i32(rows.wins), _i32(rows.tiles()), I(int(rows.tiles()[N].item()) if _prof_on else 0), _f32(out), I(out.shape[-1]))
        return out
    if out is None:
        out = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    _call('vpho_conv3x3_winograd_nhwc_f32', _f32(x), _f32(u), _f32(bias), I(N), I(H), I(W), I(cin), I(x_ld), I(cout), F(out_slope), _f32(out),
          I(out.shape[-1]))
    return out


def winograd_weights_device(w, for_input_gradient=False):
    """u of pack.winograd_weights made on the DEVICE (training: the weights change every step, a host transform would synchronise),
    for the forward convolution or for its input-gradient convolution; kept on the weight tensor object, follows its version"""
    key = '_vpho_wino_dev_t' if for_input_gradient else '_vpho_wino_dev'
    c = getattr(w, key, None)
    if c is None or c[0] != w._version:
        cout, k9 = w.shape
        cin = k9 // 9
        o, i = (cin, cout) if for_input_gradient else (cout, cin)
        u = c[1] if c is not None else torch.empty((i // 8, 16, o, 8), device=w.device, dtype=torch.float32)
        _call('vpho_winograd_weights_f32', _f32(w), I(cout), I(cin), I(1 if for_input_gradient else 0), _f32(u))
        c = (w._version, u)
        setattr(w, key, c)
        if _WINO_BATCH is not None:
            _WINO_BATCH.add(w, key, 1 if for_input_gradient else 0, u)
    return c[1]


class WinogradWeightBatch:
    """Every transform requested inside ``with batch:`` is remembered; ``refresh()`` (after the optimiser step) redoes ALL of them with one
    launch (vpho_winograd_weights_multi_f32) and marks them current, so the step's convolutions find them ready -- instead of one small
    launch in front of each of the step's 81 Winograd convolutions."""

    def __init__(self):
        self.items, self.table, self.blocks = {}, None, 0

    def __enter__(self):
        global _WINO_BATCH
        self._outer, _WINO_BATCH = _WINO_BATCH, self
        return self

    def __exit__(self, *exc):
        global _WINO_BATCH
        _WINO_BATCH = self._outer

    def add(self, w, key, mode, u):
        k = (w.data_ptr(), key)
        if k not in self.items:
            self.items[k] = (w, key, mode, u)              # keeps w alive: its address stays its own
            self.table = None

    def refresh(self):
        if not self.items:
            return
        if self.table is None:
            import numpy as np
            rec = np.zeros(len(self.items), dtype=[('w', '<u8'), ('u', '<u8'), ('cout', '<i4'), ('cin', '<i4'), ('mode', '<i4'), ('blk0', '<i4')])
            blk = 0
            for n, (w, key, mode, u) in enumerate(self.items.values()):
                cout, cin = w.shape[0], w.shape[1] // 9
                _ptr(w, torch.float32), _ptr(u, torch.float32)
                rec[n] = (w.data_ptr(), u.data_ptr(), cout, cin, mode, blk)
                blk += (cout * cin + 255) // 256
            dev = next(iter(self.items.values()))[0].device
            self.table, self.blocks = torch.from_numpy(rec.view(np.uint8).copy()).to(dev), blk
        _call('vpho_winograd_weights_multi_f32', _ptr(self.table, torch.uint8), I(len(self.items)), LL(self.blocks))
        for w, key, mode, u in self.items.values():
            setattr(w, key, (w._version, u))


_WINO_BATCH = None


def winograd_ok(H, W, cin, cout, x_ld):
    return H % 2 == 0 and W % 2 == 0 and cin % 16 == 0 and cout % 64 == 0 and x_ld == cin and not getattr(_conv_split, 'terms', 0)


_TRAIN_WINOGRAD = os.environ.get('VPHO_TRAIN_WINOGRAD', '1') != '0'


def _wino8():
    return os.environ.get('VPHO_WINO8', '0') not in ('0', '')      # the 8-wave experiment kernel has no BatchNorm epilogue (read per call, like the library)


def _winograd_bn(x, u, bias, out_slope, cin, cout, bn, gate_slope=1.0):
    """full-map Winograd convolution with the BatchNorm reductions in its epilogue (vpho_conv3x3_winograd_stats_nhwc_f32)"""
    N, H, W, x_ld = x.shape
    out = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    cap = (N * H * W + 255) // 256
    bn.stats = torch.empty((cap, 2, cout), device=x.device, dtype=torch.float32)
    rows_out = C.c_int(0)
    bx = (bn.x, bn.mean, bn.invstd, bn.gamma, bn.beta) if bn.x is not None else (None,) * 5
    if bn.x is not None:
        assert bn.x.shape == out.shape and bn.x.is_contiguous()
    _call('vpho_conv3x3_winograd_stats_nhwc_f32', _f32(x), _f32(u), _f32(bias), I(N), I(H), I(W), I(cin), I(x_ld), I(cout), F(out_slope), _f32(out), I(cout),
          _f32(bn.stats), I(cap), C.byref(rows_out), *[_f32(t) for t in bx], F(gate_slope))
    bn.rows, bn.calls = rows_out.value, 1
    return out


def conv3x3_train(x, w, bias=None, out_slope=1.0, rows=None, bn=None):
    """3x3 / stride 1 / pad 1 convolution of the TRAINING path: Winograd F(2x2,3x3) with the weight transform on the device where the
    shape allows, else the direct kernel (VPHO_TRAIN_WINOGRAD=0: always the direct kernel).  ``rows`` (a RoiWindows): the output is
    only ever read inside these windows (RoIAlign), so only their pixels are computed -- in place in the ordinary (N,H,W,Cout) map,
    zeros elsewhere; same values as the full convolution on every window pixel."""
    N, H, W, x_ld = x.shape
    cout, cin = w.shape[0], w.shape[1] // 9
    if not _TRAIN_WINOGRAD or not winograd_ok(H, W, cin, cout, x_ld) or w.shape[1] != 9 * cin:
        return conv2d_nhwc(x, w, bias, kh=3, kw=3, pad=1, out_slope=out_slope, bn=bn)
    u = winograd_weights_device(w)
    if bn is not None and FUSE_BN and rows is None and not _wino8():
        return _winograd_bn(x, u, bias, out_slope, cin, cout, bn)
    if rows is not None:
        assert rows.shape == (N, H, W)
        out = torch.zeros((N, H, W, cout), device=x.device, dtype=torch.float32)
        _call('vpho_conv3x3_winograd_scatter_nhwc_f32', _f32(x), _f32(u), _f32(bias), I(N), I(H), I(W), I(cin), I(x_ld), I(cout), F(out_slope),
              _i32(rows.wins), _i32(rows.tiles()), _f32(out), I(cout))
        return out
    return conv3x3_winograd(x, u, bias, out_slope)


def conv3x3_dgrad_winograd(dy, w, gate=None, rows=None, bn=None):
    """input gradient of a 3x3 / stride 1 / pad 1 convolution with packed weights w (Cout, 9*Cin): dX = conv3x3(dY, flipped / transposed
    w), optionally through the backward of the LeakyReLU that produced the convolution's input (gate = (that input, slope)).
    ``rows`` (a RoiWindows, no gate): dY is zero outside the windows' interior, so dX is computed on the window pixels only and is
    zero elsewhere.  None when the shape is not one the Winograd kernel takes (the caller then uses the direct kernel)."""
    N, H, W, ld = dy.shape
    cout, cin = w.shape[0], w.shape[1] // 9
    if not _TRAIN_WINOGRAD or not winograd_ok(H, W, cout, cin, ld) or w.shape[1] != 9 * cin:
        return None
    u = winograd_weights_device(w, for_input_gradient=True)
    if rows is not None and gate is None:
        assert rows.shape == (N, H, W)
        out = torch.zeros((N, H, W, cin), device=dy.device, dtype=torch.float32)
        _call('vpho_conv3x3_winograd_scatter_nhwc_f32', _f32(dy), _f32(u), None, I(N), I(H), I(W), I(cout), I(ld), I(cin), F(1.0),
              _i32(rows.wins), _i32(rows.tiles()), _f32(out), I(cin))
        return out
    if gate is None:
        return conv3x3_winograd(dy, u, None, 1.0)
    g, slope = gate
    if bn is not None and bn.x is not None and FUSE_BN and not _wino8():
        return _winograd_bn(dy, u, None, 1.0, cout, cin, bn, gate_slope=slope)
    assert g.shape == (N, H, W, cin) and g.is_contiguous()
    out = torch.empty((N, H, W, cin), device=dy.device, dtype=torch.float32)
    _call('vpho_conv3x3_winograd_gate_nhwc_f32', _f32(dy), _f32(u), _f32(g), F(slope), I(N), I(H), I(W), I(cout), I(ld), I(cin), _f32(out), I(cin))
    return out


def linear(x, w, bias=None, out_slope=1.0, out=None, acc64=False):
    """x: (rows, cin) fp32, w: (cout, cin) -> (rows, cout).  Same kernel as conv2d_nhwc (1x1).  ``acc64``: products and sum in double,
    one rounding (vpho_linear_acc64_f32: the regression head, whose rounding noise the 6-D normalisation amplifies into the candidates)"""
    rows, cin = x.shape
    if acc64:
        assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == cin
        y = torch.empty((rows, w.shape[0]), device=x.device, dtype=torch.float32) if out is None else out
        _call('vpho_linear_acc64_f32', _f32(x), I(rows), I(cin), I(cin), _f32(w), _f32(bias), I(w.shape[0]), F(out_slope), _f32(y), I(y.shape[-1]))
        return y
    y = conv2d_nhwc(x.view(rows, 1, 1, cin), w, bias, out_slope=out_slope,
                    out=None if out is None else out.view(rows, 1, 1, w.shape[0]))
    return y.view(rows, w.shape[0])


# ----------------------------------------------------------------------------------------------- score net / sampler
class ScoreWeights(C.Structure):
    _fields_ = [('D', C.c_int), ('Dp', C.c_int), ('nheads', C.c_int)] + \
               [(k, C.c_void_p) for k in ('t_W', 't_w', 't_b', 'pe0_w', 'pe0_b', 'pe2_w', 'pe2_b', 'w1_t', 'w1_p',
                                          'w1_f', 'b1', 'w2', 'b2')] + [('w1_p_split', C.c_void_p), ('split_terms', C.c_int)]


class OdeStats(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('nfev', 'n_accepted', 'n_rejected', 'nan_count', 'status', 'n_log')]


lib.vpho_score_workspace_bytes.restype = C.c_longlong
lib.vpho_maxpool_bwd_workspace_bytes.restype = C.c_longlong
lib.vpho_maxpool_bwd_workspace_bytes.argtypes = [C.c_int] * 7
lib.vpho_score_workspace_bytes.argtypes = [C.POINTER(ScoreWeights), C.c_int, C.c_int]
lib.vpho_score_eval.argtypes = [C.POINTER(ScoreWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                C.c_void_p, C.c_longlong, C.c_void_p]
lib.vpho_ode_sample.argtypes = [C.POINTER(ScoreWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double,
                                C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong,
                                C.POINTER(OdeStats), C.c_void_p, C.c_int, C.c_void_p]


class ScoreNet:
    """Packed weights of one BaseDenoiser (keeps the device tensors alive) + scratch."""

    def __init__(self, sd, prefix, device):
        g = lambda k: sd[f'{prefix}.{k}'].detach().to(device=device, dtype=torch.float32)
        w1, b1 = g('head.head.0.weight'), g('head.head.0.bias')          # (n,1408,256), (n,256)
        w2, b2 = g('head.head.2.weight'), g('head.head.2.bias')          # (n,256,3), (n,3)
        n = w1.shape[0]
        self.nheads, self.D = n, 3 * n
        self.Dp = (self.D + 3) // 4 * 4
        NH = n * 256
        pe0 = g('pose_encoder.0.weight')                                 # (256, D)
        t = dict(
            t_W=g('t_encoder.0.W').contiguous(), t_w=g('t_encoder.1.weight').contiguous(), t_b=g('t_encoder.1.bias').contiguous(),
            pe0_w=torch.nn.functional.pad(pe0, (0, self.Dp - self.D)).contiguous(), pe0_b=g('pose_encoder.0.bias').contiguous(),
            pe2_w=g('pose_encoder.2.weight').contiguous(), pe2_b=g('pose_encoder.2.bias').contiguous(),
            w1_t=w1[:, :128, :].permute(1, 0, 2).reshape(128, NH).contiguous(),
            w1_p=w1[:, 128:384, :].permute(0, 2, 1).reshape(NH, 256).contiguous(),
            w1_f=w1[:, 384:, :].permute(0, 2, 1).reshape(NH, 1024).contiguous(),
            b1=b1.reshape(NH).contiguous(),
            w2=torch.nn.functional.pad(w2, (0, 1)).reshape(NH, 4).contiguous(), b2=b2.reshape(n * 3).contiguous())
        self.tensors = t
        self.c = ScoreWeights()
        self.c.D, self.c.Dp, self.c.nheads = self.D, self.Dp, n
        for k, v in t.items():
            setattr(self.c, k, v.data_ptr())
        self.device = device
        self._ws = None
        self.set_split(os.environ.get('VPHO_SCORE_MFMA', 'f32'))

    def set_split(self, mode):
        """'f32' (default: fp32 MFMA) | 'bf16x6' | 'bf16x9': the score head's 256 x 256 layer as split-bf16 products (opt-in)"""
        terms = {'f32': 0, 'bf16x6': 6, 'bf16x9': 9}[mode]
        if terms and 'w1_p_split' not in self.tensors:
            w = self.tensors['w1_p'].view(self.nheads, 256, 256)
            h = w.to(torch.bfloat16)
            m = (w - h.float()).to(torch.bfloat16)
            l = (w - h.float() - m.float()).to(torch.bfloat16)
            assert torch.equal(h.float() + m.float() + l.float(), w)       # three 8-bit pieces hold an fp32 exactly
            self.tensors['w1_p_split'] = torch.stack([h, m, l], 1).contiguous()      # (n, 3, 256, 256)
        self.c.w1_p_split = self.tensors['w1_p_split'].data_ptr() if terms else None
        self.c.split_terms = terms
        self.split_mode = mode

    def workspace(self, bs, S):
        need = lib.vpho_score_workspace_bytes(C.byref(self.c), bs, S)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def score(self, feat_img, x, t, S):
        """feat_img (bs,1024), x (bs*S, D), scalar t -> score (bs*S, D)   [denoiser.py:68-82]"""
        bs = feat_img.shape[0]
        ws = self.workspace(bs, S)
        out = torch.empty((bs * S, self.D), device=self.device, dtype=torch.float32)
        _check(lib.vpho_score_eval(C.byref(self.c), _ptr(feat_img, torch.float32), bs, S, _ptr(x, torch.float32), float(t),
                                   _ptr(out), _ptr(ws), ws.numel(), _stream()))
        return out

    def sample(self, feat_img, init_x, S, T0, num_steps, xs_f64, x_f64=True, eps=1e-5, rtol=3e-3, atol=3e-4, log_cap=4096):
        """cond_ode_sampler (score_based_model.py:45-105).  Returns xs (R,steps,D), x (R,D), stats dict."""
        bs = feat_img.shape[0]
        R = bs * S
        assert init_x.shape == (R, self.D)
        ws = self.workspace(bs, S)
        xs = torch.empty((R, num_steps, self.D), device=self.device, dtype=torch.float64 if xs_f64 else torch.float32)
        x = torch.empty((R, self.D), device=self.device, dtype=torch.float64 if x_f64 else torch.float32)
        st = OdeStats()
        log = (C.c_double * (4 * log_cap))()
        _check(lib.vpho_ode_sample(C.byref(self.c), _ptr(feat_img, torch.float32), bs, S, _ptr(init_x, torch.float32),
                                   float(T0), float(eps), int(num_steps), float(rtol), float(atol), _ptr(xs),
                                   1 if xs_f64 else 0, _ptr(x), 1 if x_f64 else 0, _ptr(ws), ws.numel(), C.byref(st), log, log_cap,
                                   _stream()))
        n = min(st.n_log, log_cap)
        steps = [(log[4 * i], log[4 * i + 1], log[4 * i + 2], bool(log[4 * i + 3])) for i in range(n)]
        return xs, x, dict(nfev=st.nfev, n_accepted=st.n_accepted, n_rejected=st.n_rejected, nan_count=st.nan_count,
                           steps=steps)


# ----------------------------------------------------------------------------------------------- glue kernels
def _i32(t):
    return _ptr(t, torch.int32)


def _u8(t):
    return None if t is None else _ptr(t, torch.uint8)


def _f32(t):
    return None if t is None else _ptr(t, torch.float32)


def _f64(t):
    return None if t is None else _ptr(t, torch.float64)


def _call(name, *args):
    _check(getattr(lib, name)(*args, _stream()))


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


LL, I, F = C.c_longlong, C.c_int, C.c_float


def nchw_to_nhwc(x, ld=None):
    N, Cc, H, W = x.shape
    ld = Cc if ld is None else ld
    y = _new((N, H, W, ld), x)
    _call('vpho_nchw_to_nhwc_f32', _f32(x), I(N), I(Cc), I(H), I(W), _f32(y), I(ld))
    return y


def nhwc_to_nchw(x, channels=None):
    N, H, W, ld = x.shape
    Cc = ld if channels is None else channels
    y = _new((N, Cc, H, W), x)
    _call('vpho_nhwc_to_nchw_f32', _f32(x), I(N), I(H), I(W), I(Cc), I(ld), _f32(y))
    return y


def maxpool_nhwc(x, k, stride, pad):
    N, H, W, Cc = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = _new((N, OH, OW, Cc), x)
    _call('vpho_maxpool_nhwc_f32', _f32(x), I(N), I(H), I(W), I(Cc), I(k), I(stride), I(pad), _f32(y))
    return y


def resize_bilinear_nhwc(x, OH, OW, out=None, c_off=0, accumulate=False, channels=None, rows=None):
    N, H, W, ldx = x.shape
    Cc = ldx if channels is None else channels
    if out is None:
        out = _new((N, OH, OW, Cc), x)
    assert out.shape[:3] == (N, OH, OW)
    if rows is not None:                                     # only the listed output pixels (RoiWindows), in place
        assert rows.shape == (N, OH, OW)
        _call('vpho_resize_bilinear_rows_nhwc_f32', _f32(x), I(N), I(H), I(W), I(Cc), I(ldx), I(OH), I(OW), _f32(out), I(out.shape[-1]),
              I(c_off), I(1 if accumulate else 0), _ptr(rows.row_map, torch.int32), _ptr(rows.count, torch.int32),
              I(int(rows.count.item()) if _prof_on else 0))
        return out
    _call('vpho_resize_bilinear_nhwc_f32', _f32(x), I(N), I(H), I(W), I(Cc), I(ldx), I(OH), I(OW), _f32(out), I(out.shape[-1]),
          I(c_off), I(1 if accumulate else 0))
    return out


class RoiWindows:
    """Per-image pixel windows of a feature map (device data): ``wins`` (N,5) int32 = (first row, y0, x0, w, h), ``row_map``
    the linear pixel index of every window pixel, ``count`` (1,) their number.  See vpho_roi_windows_i32."""
    def __init__(self, wins, row_map, count, shape):
        self.wins, self.row_map, self.count, self.shape = wins, row_map, count, shape
        self._tiles = None

    def tiles(self):
        """(N+1,) int32: first 2 x 2 Winograd tile of every image's window, [N] = their number (vpho_winograd_window_tiles_i32)"""
        if self._tiles is None:
            self._tiles = torch.empty((self.shape[0] + 1,), device=self.wins.device, dtype=torch.int32)
            _call('vpho_winograd_window_tiles_i32', _i32(self.wins), I(self.shape[0]), _i32(self._tiles))
        return self._tiles

    def to_map(self, rows):
        """Inspection helper (synchronises): the compact matrix scattered back to (N,H,W,C), zeros outside the windows, and the
        (N,H,W) mask of the window pixels."""
        n = int(self.count.item())
        N, H, W = self.shape
        idx = self.row_map[:n].long()
        full = torch.zeros((N * H * W, rows.shape[-1]), device=rows.device, dtype=rows.dtype)
        full[idx] = rows[:n]
        mask = torch.zeros((N * H * W,), device=rows.device, dtype=torch.bool)
        mask[idx] = True
        return full.view(N, H, W, -1), mask.view(N, H, W)


def roi_windows(boxes_a, boxes_b, N, H, W, spatial_scale, dilate=0):
    dev = boxes_a.device
    wins = torch.empty((N, 5), device=dev, dtype=torch.int32)
    row_map = torch.empty((N * H * W,), device=dev, dtype=torch.int32)
    count = torch.empty((1,), device=dev, dtype=torch.int32)
    _call('vpho_roi_windows_i32', _f32(boxes_a), _f32(boxes_b), I(N), I(H), I(W), F(spatial_scale), I(dilate), _ptr(wins, torch.int32),
          _ptr(row_map, torch.int32), _ptr(count, torch.int32))
    return RoiWindows(wins, row_map, count, (N, H, W))


def roi_align_nhwc(feat, boxes, out_size, spatial_scale, flip_w=None, out=None, c_off=0, win=None):
    if win is not None:                                      # feat = compact (rows, C) matrix of the windows' pixels
        N, H, W = win.shape
        Cc = feat.shape[-1]
        if out is None:
            out = _new((N, out_size, out_size, Cc), feat)
        _call('vpho_roi_align_window_nhwc_f32', _f32(feat), _ptr(win.wins, torch.int32), I(N), I(H), I(W), I(Cc), _f32(boxes),
              F(spatial_scale), I(out_size), _u8(flip_w), _f32(out), I(out.shape[-1]), I(c_off), I(int(win.count.item()) if _prof_on else 0))
        return out
    N, H, W, Cc = feat.shape
    if out is None:
        out = _new((N, out_size, out_size, Cc), feat)
    _call('vpho_roi_align_nhwc_f32', _f32(feat), I(N), I(H), I(W), I(Cc), _f32(boxes), F(spatial_scale), I(out_size), _u8(flip_w),
          _f32(out), I(out.shape[-1]), I(c_off))
    return out


def roi_align_dual_nhwc(feat, boxes, out_size, spatial_scale, win, out2, flip_w2=None, c_off2=0, out=None):
    """one pooling pass over the compact window matrix, two destinations: returns the plain crop (new tensor, or ``out``) and writes the
    same values into ``out2`` at channel offset ``c_off2`` under ``flip_w2`` (vpho_roi_align_window_dual_nhwc_f32)"""
    N, H, W = win.shape
    Cc = feat.shape[-1]
    if out is None:
        out = _new((N, out_size, out_size, Cc), feat)
    assert out.shape == (N, out_size, out_size, Cc) and out.is_contiguous()
    _call('vpho_roi_align_window_dual_nhwc_f32', _f32(feat), _ptr(win.wins, torch.int32), I(N), I(H), I(W), I(Cc), _f32(boxes),
          F(spatial_scale), I(out_size), _u8(None), _f32(out), I(Cc), I(0), _u8(flip_w2), _f32(out2), I(out2.shape[-1]), I(c_off2),
          I(int(win.count.item()) if _prof_on else 0))
    return out


def align_heatmap_nhwc(hm, bbox, bbox_rect, flip_w=None):
    N, S, _, Cc = hm.shape
    out = torch.empty_like(hm)
    _call('vpho_align_heatmap_nhwc_f32', _f32(hm), I(N), I(S), I(Cc), _f32(bbox), _f32(bbox_rect), _u8(flip_w), _f32(out))
    return out


def nerf_embed(g, flip_x=None):
    N = g.shape[0]
    out = _new((N, 64), g)
    _call('vpho_nerf_embed_f32', _f32(g), I(N), _u8(flip_x), _f32(out))
    return out


def cross_tokens(proj_hand, proj_obj, grav_emb, pe):
    bs = proj_hand.shape[0]
    out = _new((bs, 65, 512), proj_hand)
    _call('vpho_cross_tokens_f32', _f32(proj_hand), _f32(proj_obj), _f32(grav_emb), _f32(pe), I(bs), _f32(out))
    return out


def mha(qkv, S, B, E, nhead, drop=None):
    """drop: optional (B*nhead, S, S) keep-mask / (1 - p) on the attention probabilities (training)"""
    out = _new((S, B, E), qkv)
    _call('vpho_mha_dropout_f32', _f32(qkv), I(S), I(B), I(E), I(nhead), _f32(drop), _f32(out))
    return out


def add_layernorm(x, r, gamma, beta, eps=1e-5):
    E = x.shape[-1]
    out = torch.empty_like(x)
    _call('vpho_add_layernorm_f32', _f32(x), _f32(r), _f32(gamma), _f32(beta), LL(x.numel() // E), I(E), F(eps), _f32(out))
    return out


def force_local(scale, logits, anchor, rows, group=1, group_stride=1, off_scale=0, off_logits=0, friction=0.8):
    out = _new((rows, 3), scale)
    _call('vpho_force_local_f32', _f32(scale), I(scale.shape[1]), _f32(logits), I(logits.shape[1]), _f32(anchor), F(friction), LL(rows),
          I(group), I(group_stride), I(off_scale), I(off_logits), _f32(out))
    return out


def rot6d_to_axis_angle(x, rot_per_row, out=None, ldo=None):
    rows, ldx = x.shape
    if out is None:
        out = _new((rows, 3 * rot_per_row), x)
    _call('vpho_rot6d_to_axis_angle_f32', _f32(x), LL(rows), I(rot_per_row), I(ldx), _f32(out), I(out.shape[-1] if ldo is None else ldo))
    return out


def append_betas(betas, out, rows_per_image):
    rows = out.numel() // out.shape[-1]
    _call('vpho_append_betas_f32', _f32(betas), LL(rows), LL(rows_per_image), _f32(out), I(out.shape[-1]))
    return out


# ----------------------------------------------------------------------------------------------- MANO
class ManoTables(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('v_template', 'shapedirs', 'posedirs_t', 'J_regressor', 'weights', 'tip_posedirs_t', 'posedirs_mfma')]


class Mano:
    def __init__(self, mano, device):
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(device).contiguous()
        self.tensors = dict(v_template=t(mano['v_template']), shapedirs=t(mano['shapedirs']),
                            posedirs_t=t(mano['posedirs']).reshape(778 * 3, 135).t().contiguous(),
                            J_regressor=t(mano['J_regressor']), weights=t(mano['weights']))
        tips = [745, 317, 444, 556, 673, 728, 353, 442, 576, 694]                   # manopth tips, then the HO3D ones (csrc/mano.hip)
        cols = torch.tensor([3 * v + c for v in tips for c in range(3)], device=device)
        self.tensors['tip_posedirs_t'] = self.tensors['posedirs_t'][:, cols].contiguous()     # (135, 30)
        # the table in the operand order of the matrix-core kernel (include/vpho_hip.h): k = 2 (4 g + sl) + lh, vertex = 32 t + li
        pad = torch.zeros((136, 800, 3), device=device, dtype=torch.float32)
        pad[:135, :778] = self.tensors['posedirs_t'].view(135, 778, 3)
        frag = pad.view(17, 4, 2, 25, 32, 3).permute(3, 0, 2, 4, 1, 5).reshape(25, 17, 2, 32, 3, 4)        # [t][g][lh][li][j][e]
        self.tensors['posedirs_mfma'] = frag.permute(0, 1, 4, 2, 3, 5).contiguous()                            # [t][g][j][lh][li][e]
        self.c = ManoTables()
        for k, v in self.tensors.items():
            setattr(self.c, k, v.data_ptr())

    def shape(self, betas):
        n = betas.shape[0]
        vs, J = _new((n, 778, 3), betas), _new((n, 16, 3), betas)
        _call('vpho_mano_shape_f32', C.byref(self.c), _f32(betas), I(n), _f32(vs), _f32(J))
        return vs, J

    def train(self, rot6d, shape, gt_vert, gt_joint, gt_rot6d, gt_shape, is_right, weights, want_outputs=False, is_ho3d=None):
        """HeadMano tail in training mode (vpho_mano_train_f32): -> losses dict (0-d fp64 tensors, weighted), d_rot6d (bs,96),
        d_shape (bs,10)[, verts, joints]; weights = (vert, joint, mano_pose, mano_shape)"""
        bs = rot6d.shape[0]
        d6, ds = torch.empty_like(rot6d), torch.empty_like(shape)
        parts = _new((bs, 4), rot6d, torch.float64)
        verts = _new((bs, 778, 3), rot6d) if want_outputs else None
        joints = _new((bs, 21, 3), rot6d) if want_outputs else None
        _call('vpho_mano_train_f32', C.byref(self.c), _f32(rot6d), _f32(shape), _f32(gt_vert), _f32(gt_joint), _f32(gt_rot6d), _f32(gt_shape),
              _u8(is_right), _u8(is_ho3d), I(bs), F(weights[0]), F(weights[1]), F(weights[2]), F(weights[3]), _f32(d6), _f32(ds), _f64(parts),
              _f32(verts), _f32(joints))
        tot = parts.sum(0)
        L = dict(vert_loss=tot[0] * (weights[0] / (bs * 2334.0)), joint_loss=tot[1] * (weights[1] / (bs * 63.0)),
                 mano_pose_loss=tot[2] * (weights[2] / (bs * 96.0)), mano_shape_loss=tot[3] * (weights[3] / (bs * 10.0)))
        return (L, d6, ds, verts, joints) if want_outputs else (L, d6, ds)

    def fk(self, pose, shape_ctx, hands_per_image, want_verts=True, ho3d=None):
        """pose (n, >=48) rows; shape_ctx = self.shape(betas).  -> verts (n,778,3)|None, joints (n,21,3)"""
        n, ld = pose.shape
        vs, J = shape_ctx
        verts = _new((n, 778, 3), pose) if want_verts else None
        joints = _new((n, 21, 3), pose)
        _call('vpho_mano_fk_f32', C.byref(self.c), _f32(pose), I(ld), LL(n), I(hands_per_image), _f32(vs), _f32(J), _u8(ho3d),
              _f32(verts), _f32(joints))
        return verts, joints


# ----------------------------------------------------------------------------------------------- aggregation
class ObjTables(C.Structure):
    _fields_ = [('kpt', C.c_void_p), ('vert', C.c_void_p), ('com', C.c_void_p), ('n_kpt', I), ('n_vert', I), ('n_obj', I)]


class AnchorTables(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('face_idx', 'anchor_weight', 'vert2joint', 'skeleton')]


def _ids_to_device(ids, device):
    host = torch.tensor(ids, dtype=torch.int32).pin_memory()        # the caching host allocator keeps the block until the copy's event
    return host.to(device, non_blocking=True)


class Aggregation:
    """Device tables + thin wrappers of the aggregation kernels."""

    def __init__(self, assets, anchor_skeleton, device):
        ycb = assets['ycb']
        self.names = list(ycb.keys())
        self.name_to_id = {n: i for i, n in enumerate(self.names)}
        f = lambda a: torch.as_tensor(a, dtype=torch.float32).to(device).contiguous()
        stack = lambda key: torch.stack([torch.as_tensor(ycb[n][key], dtype=torch.float32).reshape(-1, 3) for n in self.names]).to(device).contiguous()
        self.kpt, self.vert = stack('kpt3d'), stack('verts_sampled')
        self.com = stack('CoM').reshape(len(self.names), 3).contiguous()
        self.obj = ObjTables(self.kpt.data_ptr(), self.vert.data_ptr(), self.com.data_ptr(), self.kpt.shape[1], self.vert.shape[1], len(self.names))
        a = assets['anchor']
        self.face = torch.as_tensor(a['face_vert_idx'], dtype=torch.int32).to(device).contiguous()
        self.aw = f(a['anchor_weight'])
        self.v2j = f(a['vert2joint'])
        self.skel = torch.as_tensor(anchor_skeleton, dtype=torch.int32).to(device).contiguous()
        self.anchor = AnchorTables(self.face.data_ptr(), self.aw.data_ptr(), self.v2j.data_ptr(), self.skel.data_ptr())
        self.device = device

    def obj_ids(self, names):
        """class indices of a batch as a device tensor: filled into pinned host memory and copied asynchronously on the CALLER's current
        stream, every call (a pageable-memory copy would park the calling thread behind everything queued on its stream; a cache shared
        by the evaluator's slot threads would hand a tensor allocated on one stream to kernels of another -- ADVICE r3)"""
        return _ids_to_device([self.name_to_id[n] for n in names], self.device)

    def hand_candidates(self, diff_pose, reg_pose, bs, S):
        pose = _new((bs, 2 * S, 48), diff_pose)
        _call('vpho_hand_candidates_f32', _f32(diff_pose), I(diff_pose.shape[-1]), _f32(reg_pose), I(bs), I(S), _f32(pose))
        return pose

    def hand_heat(self, joints, root, K, bbox, heatmap, observe):
        bs, Cn = joints.shape[:2]
        _, J, H, W = heatmap.shape
        out = _new((bs, Cn, len(observe)), joints)
        obs = (C.c_int * len(observe))(*observe)
        _call('vpho_hand_heat_f32', _f32(joints), _f32(root), _f32(K), _f32(bbox), _f32(heatmap), I(bs), I(Cn), I(J), I(H), I(W), obs,
              I(len(observe)), _f32(out))
        return out

    def hand_fuse_level(self, hv, pose, k, level, want_topk_pose=False, want_scores=False):
        bs, Cn, n_obs = hv.shape
        Fn = 1 if level == 0 else 5
        val, idx = _new((bs, Fn, k), hv), _new((bs, Fn, k), hv, torch.int32)
        tp = _new((bs, k, Fn, 3), hv) if want_topk_pose else None
        sc = _new((bs, Cn, Fn), hv) if want_scores else None
        _call('vpho_hand_fuse_level_f32', _f32(hv), I(n_obs), _f32(pose), I(bs), I(Cn), I(k), I(level), _f32(val), _i32(idx), _f32(tp), _f32(sc))
        return (val, idx, tp, sc) if want_scores else (val, idx, tp)

    def topk(self, scores, k, F_=1):
        """scores (rows, n) [F_=1] or (rows, n, F_) -> val, idx (rows, F_, k)"""
        rows, n = scores.shape[:2]
        val, idx = _new((rows, F_, k), scores), _new((rows, F_, k), scores, torch.int32)
        _call('vpho_topk_f32', _f32(scores), I(rows), I(n), I(F_), I(k), _f32(val), _i32(idx))
        return val, idx

    def topk_weights(self, val):
        w = torch.empty_like(val)
        _call('vpho_topk_weights_f32', _f32(val), I(val.numel() // val.shape[-1]), I(val.shape[-1]), _f32(w))
        return w

    def obj_heat_score(self, pose, root, obj_id, is_right, K, bbox, heatmap, transl_override=None):
        bs, n = pose.shape[:2]
        _, J, H, W = heatmap.shape
        assert J == self.obj.n_kpt
        score = _new((bs, n), root)
        _call('vpho_obj_heat_score', _f64(pose), I(n), _f64(transl_override), _f32(root), C.byref(self.obj), _i32(obj_id), _u8(is_right),
              _f32(K), _f32(bbox), _f32(heatmap), I(bs), I(H), I(W), _f32(score))
        return score

    def obj_cross(self, pose, transl_idx, rot_idx):
        bs, n = pose.shape[:2]
        ko = transl_idx.shape[-1]
        cand = _new((bs, ko * ko, 9), pose, torch.float64)
        _call('vpho_obj_cross_candidates', _f64(pose), I(n), _i32(transl_idx), _i32(rot_idx), I(bs), I(ko), _f64(cand))
        return cand

    def obj_physics_score(self, cand, root, obj_id, is_right, force_point, force_global):
        bs, n = cand.shape[:2]
        score = _new((bs, n), root)
        _call('vpho_obj_physics_score', _f64(cand), I(n), _f32(root), C.byref(self.obj), _i32(obj_id), _u8(is_right), _f32(force_point),
              _f32(force_global), I(bs), _f32(score))
        return score

    def obj_fuse(self, pose, idx_a, w_a, idx_b=None, w_b=None, pick_b=None):
        bs, n = pose.shape[:2]
        k = idx_a.shape[-1]
        fused = _new((bs, 9), pose, torch.float64)
        _call('vpho_obj_fuse_f64', _f64(pose), I(n), _i32(idx_a), _f32(w_a), None if idx_b is None else _i32(idx_b), _f32(w_b), _u8(pick_b),
              I(bs), I(k), _f64(fused))
        return fused

    def obj_verts(self, pose, root, obj_id, is_right):
        bs = pose.shape[0]
        out = _new((bs, self.obj.n_vert, 3), root)
        _call('vpho_obj_verts_f32', _f64(pose), _f32(root), C.byref(self.obj), _i32(obj_id), _u8(is_right), I(bs), _f32(out))
        return out

    def force_anchor(self, verts, root, force_local_, hands_per_image):
        n = verts.shape[0]
        fp, fg = _new((n, 32, 3), verts), _new((n, 32, 3), verts)
        _call('vpho_force_anchor_f32', C.byref(self.anchor), _f32(verts), _f32(root), _f32(force_local_), LL(n), I(hands_per_image), _f32(fp), _f32(fg))
        return fp, fg

    def contact_detect(self, hand_verts, hand_normals, obj_verts, obj_normals, normal_thresh=(-0.015, 0.01), vertical_thresh=0.01,
                       decay=(-0.005, 0.005)):
        """detect_hand_and_object_contact (physics_fn.py:47-117) for n samples: (n,Nh,3), (n,Nh,3), (n,No,3), (n,No,3) ->
        hand_contact (n,Nh), obj_contact (n,No), obj_contact_to_hand_vert (n,No) int32"""
        n, nh, _ = hand_verts.shape
        no = obj_verts.shape[1]
        hc, oc = _new((n, nh), hand_verts), _new((n, no), hand_verts)
        o2h = _new((n, no), hand_verts, torch.int32)
        args = (F(normal_thresh[0]), F(normal_thresh[1]), F(vertical_thresh), F(decay[0]), F(decay[1]))
        _call('vpho_contact_detect_f32', _f32(hand_verts), _f32(hand_normals), _f32(obj_verts), I(n), I(nh), I(no), *args, _f32(hc), None)
        _call('vpho_contact_detect_f32', _f32(obj_verts), _f32(obj_normals), _f32(hand_verts), I(n), I(no), I(nh), *args, _f32(oc), _i32(o2h))
        return hc, oc, o2h

    def force_contact(self, hand_contact, thresh=0.0):
        """ForceAnchor.get_force_contact + check_is_grasped (physics_fn.py:201-221): (n, >=778) -> (n,32), (n,) uint8"""
        n, ld = hand_contact.shape
        fc, gr = _new((n, 32), hand_contact), _new((n,), hand_contact, torch.uint8)
        _call('vpho_force_contact_f32', C.byref(self.anchor), _f32(hand_contact), I(ld), I(n), F(thresh), _f32(fc), _u8(gr))
        return fc, gr

    def anchor_frames(self, verts):
        """ForceAnchor.__call__ (physics_fn.py:224-257): verts (n,778,3) -> points (n,32,3), frames (n,32,3,3)"""
        n = verts.shape[0]
        pts, frames = _new((n, 32, 3), verts), _new((n, 32, 3, 3), verts)
        _call('vpho_anchor_frames_f32', C.byref(self.anchor), _f32(verts), LL(n), _f32(pts), _f32(frames))
        return pts, frames

    def force_optimize(self, verts, gravity, com, force_contact, is_grasped, batch_size, iters=3000, phase1=300, lr=1e-3):
        """ForceOptimizer.optimize_batch inner loop (force_optimization.py:110-207) for n = n_batches*batch_size samples."""
        n = verts.shape[0]
        assert n % batch_size == 0, 'pad the last batch: the batch-mean force loss couples the samples of a batch'
        pts, frames = self.anchor_frames(verts)
        fl, fg = _new((n, 32, 3), verts), _new((n, 32, 3), verts)
        scale, weight, losses = _new((n, 32), verts), _new((n, 32, 8), verts), _new((n // batch_size, 4), verts)
        _call('vpho_force_optimize_f32', _f32(pts), _f32(frames), _f32(gravity), _f32(com), _f32(force_contact), _u8(is_grasped),
              I(n // batch_size), I(batch_size), I(iters), I(phase1), F(lr), _f32(fl), _f32(fg), _f32(scale), _f32(weight), _f32(losses))
        return dict(force_local=fl, force_global=fg, scale=scale, weight=weight, losses=losses, force_point=pts, frames=frames)

    def hand_phys_candidates(self, agg_pose, betas, topk_pose):
        bs, k = topk_pose.shape[:2]
        out = _new((bs, k + 1, 58), agg_pose)
        _call('vpho_hand_phys_candidates_f32', _f32(agg_pose), I(agg_pose.shape[-1]), _f32(betas), _f32(topk_pose), I(bs), I(k), _f32(out))
        return out

    def hand_phys_score(self, force_point, force_global, obj_vert, bs, n_cand):
        out = _new((bs, n_cand, 5), force_point)
        _call('vpho_hand_phys_score_f32', _f32(force_point), _f32(force_global), _f32(obj_vert), I(obj_vert.shape[1]), I(bs), I(n_cand), _f32(out))
        return out

    def hand_phys_fuse(self, cand, idx):
        bs, n_cand = cand.shape[:2]
        out = _new((bs, 58), cand)
        _call('vpho_hand_phys_fuse_f32', _f32(cand), I(n_cand), _i32(idx), I(bs), I(idx.shape[-1]), _f32(out))
        return out


# ----------------------------------------------------------------------------------------------- metrics
def hand_metrics(pd, gt, per_point=False):
    """pd, gt (n,P,3) fp32 metres -> mean error (n,), Procrustes-aligned mean error (n,), [per-point errors (n,P)]
    (test.py:657-680)."""
    n, P, _ = pd.shape
    assert gt.shape == pd.shape
    me, pa = _new((n,), pd), _new((n,), pd)
    pp = _new((n, P), pd) if per_point else None
    _call('vpho_hand_metrics_f32', _f32(pd), _f32(gt), I(n), I(P), _f32(me), _f32(pa), _f32(pp))
    return (me, pa, pp) if per_point else (me, pa)


def obj_9d_to_rt(pose9, root_joint):
    """(n,9) fp64 [rot6d | t], (n,3) fp32 root -> (n,3,4) fp64 [R | t + root]  (transform_fn.py:85-90, train_diff_hand_obj.py:594-597)"""
    n = pose9.shape[0]
    rt = _new((n, 3, 4), pose9, torch.float64)
    _call('vpho_obj_9d_to_rt_f64', _f64(pose9), _f32(root_joint), I(n), _f64(rt))
    return rt


class ObjMetricTables(C.Structure):
    _fields_ = [('bbox3d', C.c_void_p), ('verts_sampled', C.c_void_p), ('verts', C.c_void_p), ('vert_offset', C.c_void_p),
                ('diameter', C.c_void_p), ('n_obj', I), ('n_sampled', I)]


from .ops_names import OBJ_METRIC_NAMES  # noqa: E402,F401


class ObjectMetrics:
    """TesterObject (lib/engine/test.py:240-503) on the device: fp64 model tables + thin wrapper of vpho_obj_metrics_f64."""

    def __init__(self, ycb, device):
        import numpy as np
        self.names = list(ycb.keys())
        self.name_to_id = {n: i for i, n in enumerate(self.names)}
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(device)
        self.bbox3d = d(np.stack([np.asarray(ycb[n]['bbox3d']).reshape(8, 3) for n in self.names]))
        self.verts_sampled = d(np.stack([np.asarray(ycb[n]['verts_sampled']).reshape(-1, 3) for n in self.names]))
        counts = [int(np.asarray(ycb[n]['verts']).reshape(-1, 3).shape[0]) for n in self.names]
        self.max_verts = max(counts)
        self.verts = d(np.concatenate([np.asarray(ycb[n]['verts']).reshape(-1, 3) for n in self.names]))
        self.vert_offset = torch.as_tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(device)
        self.diameter = d(np.array([float(ycb[n]['diameter']) for n in self.names]))
        self.c = ObjMetricTables(self.bbox3d.data_ptr(), self.verts_sampled.data_ptr(), self.verts.data_ptr(),
                                 self.vert_offset.data_ptr(), self.diameter.data_ptr(), len(self.names), self.verts_sampled.shape[1])
        self.device = device

    def obj_ids(self, names):
        """class indices of a batch as a device tensor: filled into pinned host memory and copied asynchronously on the CALLER's current
        stream, every call (a pageable-memory copy would park the calling thread behind everything queued on its stream; a cache shared
        by the evaluator's slot threads would hand a tensor allocated on one stream to kernels of another -- ADVICE r3)"""
        return _ids_to_device([self.name_to_id[n] for n in names], self.device)

    def __call__(self, pd_rt, gt_rt, cam_intr, obj_id):
        """pd_rt, gt_rt (n,3,4), cam_intr (n,3,3) fp64, obj_id (n,) int32 -> (n,16) fp64 in the order of OBJ_METRIC_NAMES."""
        n = pd_rt.shape[0]
        assert gt_rt.shape == (n, 3, 4) and pd_rt.shape == (n, 3, 4) and cam_intr.shape == (n, 3, 3) and obj_id.shape == (n,)
        need = lib.vpho_obj_metrics_workspace_bytes(C.byref(self.c), I(n), I(self.max_verts))
        if need < 0:
            raise VphoError('vpho_obj_metrics_workspace_bytes: bad argument')
        # a fresh block from torch's caching allocator per call (stream-ordered reuse): one object serves the evaluator's slots, whose
        # streams run concurrently -- a workspace kept on the object would be shared by kernels of different streams
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = _new((n, 16), pd_rt, torch.float64)
        _call('vpho_obj_metrics_f64', C.byref(self.c), _f64(pd_rt), _f64(gt_rt), _f64(cam_intr), _i32(obj_id), I(n), I(self.max_verts),
              _f64(out), _ptr(ws), LL(ws.numel()))
        return out


# ----------------------------------------------------------------------------------------------- score-network training
def _at(t, off=0):
    """device address of element `off` of a contiguous fp32 tensor"""
    _ptr(t, torch.float32)
    return C.c_void_p(t.data_ptr() + 4 * off)


def linear_into(x, w, bias, out, c_off, out_slope=1.0):
    """out[:, c_off : c_off + cout] = act(x @ w.T + bias) for a wider row-major `out` (concatenation buffers)."""
    rows, cin = x.shape
    ld = out.shape[1]
    conv2d_nhwc(x.view(rows, 1, 1, cin), w, bias, out_slope=out_slope, out_view=(out, ld, ld, ld, c_off))
    return out


def dsm_prepare(gt_pose, t, z, fourier_W, Dp):
    """t (reps,bs), z (reps,bs,D) -> x_t (rows,Dp), emb (rows,128), std (rows,)   rows = reps*bs, row = rep*bs + b"""
    reps, bs = t.shape
    D = gt_pose.shape[1]
    xt, emb, sd = _new((reps * bs, Dp), gt_pose), _new((reps * bs, 128), gt_pose), _new((reps * bs,), gt_pose)
    _call('vpho_dsm_prepare_f32', _f32(gt_pose), _f32(t), _f32(z), _f32(fourier_W), I(bs), I(reps), I(D), I(Dp), _f32(xt), _f32(emb), _f32(sd))
    return xt, emb, sd


def plinear2_fwd(h, w2, b2, std_rows, nheads):
    rows = h.shape[0]
    score = _new((rows, 3 * nheads), h)
    _call('vpho_plinear2_fwd_f32', _f32(h), _f32(w2), _f32(b2), _f32(std_rows), LL(rows), I(nheads), _f32(score))
    return score


def dsm_loss(score, z, std_rows, batch_times_reps):
    """-> loss (0-d fp64 device tensor), dout (rows,D)"""
    rows, D = score.shape
    dout = torch.empty_like(score)
    loss = _new((), score, torch.float64)
    ws = _new((1024,), score, torch.float64)
    _call('vpho_dsm_loss_f32', _f32(score), _f32(z), _f32(std_rows), LL(rows), I(D), I(batch_times_reps), _f32(dout), _f64(loss), _f64(ws), I(1024))
    return loss, dout


def mse_loss(pd, gt, weight=1.0):
    """-> weight * mean((pd-gt)^2) (0-d fp64 device tensor), its gradient w.r.t. pd"""
    assert pd.shape == gt.shape
    grad = torch.empty_like(pd)
    loss = _new((), pd, torch.float64)
    ws = _new((1024,), pd, torch.float64)
    _call('vpho_mse_loss_f32', _f32(pd), _f32(gt), LL(pd.numel()), F(weight), _f32(grad), _f64(loss), _f64(ws), I(1024))
    return loss, grad


def plinear2_bwd(h, dout, w2, nheads):
    rows = h.shape[0]
    dpre, dw2, db2 = torch.empty_like(h), torch.empty_like(w2), _new((nheads, 3), h)
    _call('vpho_plinear2_bwd_f32', _f32(h), _f32(dout), _f32(w2), LL(rows), I(nheads), _f32(dpre), _f32(dw2), _f32(db2))
    return dpre, dw2, db2


def relu_bwd(dy, dy_off, ld_dy, y, y_off, ld_y, rows, cols):
    """contiguous (rows, cols) = where(y_slice > 0, dy_slice, 0); slices are given by element offset + leading dimension"""
    dx = _new((rows, cols), dy)
    _call('vpho_relu_bwd_f32', _at(dy, dy_off), I(ld_dy), _at(y, y_off), I(ld_y), LL(rows), I(cols), _f32(dx), I(cols))
    return dx


def colsum(x):
    rows, cols = x.shape
    out = _new((cols,), x)
    ws = _bn_workspace(cols, x.device)
    _call('vpho_colsum_f32', _f32(x), I(cols), LL(rows), I(cols), _f32(out), _ptr(ws))
    return out


def sum_repeats(x, c_off, bs, reps, cols):
    out = _new((bs, cols), x)
    _call('vpho_sum_repeats_f32', _f32(x), I(x.shape[1]), I(c_off), I(bs), I(reps), I(cols), _f32(out))
    return out


def transpose(x, pad_to=4):
    """(rows, cols) -> (cols, rows rounded up to a multiple of pad_to), the padding columns zero (GEMM operands want a
    reduction length that is a multiple of 4)"""
    rows, cols = x.shape
    rp = (rows + pad_to - 1) // pad_to * pad_to
    y = _new((cols, rp), x) if rp == rows else torch.zeros((cols, rp), device=x.device, dtype=x.dtype)
    _call('vpho_transpose_f32', _f32(x), I(rows), I(cols), I(cols), _f32(y), I(rp))
    return y


def im2col_t(x, kh, kw, stride, pad_y, pad_x, OH, OW, cin=None):
    """x (N,H,W,ld) NHWC -> (kh*kw*cin, P rounded up to a multiple of 4) with P = N*OH*OW; see vpho_im2col_t_f32"""
    N, H, W, ld = x.shape
    cin = ld if cin is None else cin
    P = N * OH * OW
    ldo = (P + 3) // 4 * 4
    out = _new((kh * kw * cin, ldo), x)
    _call('vpho_im2col_t_f32', _f32(x), I(N), I(H), I(W), I(cin), I(ld), I(kh), I(kw), I(stride), I(pad_y), I(pad_x), I(OH), I(OW), _f32(out), LL(ldo))
    return out


def window_groups(win):
    """live 32-pixel groups of the (N,H,W) map of a ``RoiWindows`` (ascending list + count, device tensors): where a gradient that
    came back through RoIAlign can be non-zero (vpho_window_groups_i32)"""
    N, H, W = win.shape
    lst = torch.empty((N * H * W // 32,), device=win.wins.device, dtype=torch.int32)
    cnt = torch.empty((1,), device=win.wins.device, dtype=torch.int32)
    _call('vpho_window_groups_i32', _i32(win.wins), I(N), I(H), I(W), _i32(lst), _i32(cnt))
    return lst, cnt


def conv2d_wgrad_nhwc(x, dy, kh, kw, stride, pad_y, pad_x, cin=None, groups=None):
    """x (N,H,W,ld), dy (N,OH,OW,Cout) contiguous -> dW (Cout, kh*kw*cin) packed; implicit TN GEMM (csrc/conv_wgrad.hip).
    ``groups`` = window_groups(...) of the OUTPUT map: dy is zero outside them, the reduction skips everything else."""
    N, H, W, ld = x.shape
    _, OH, OW, cout = dy.shape
    cin = ld if cin is None else cin
    dw = _new((cout, kh * kw * cin), x)
    nbytes = lib.vpho_conv2d_wgrad_workspace_bytes(I(N), I(OH), I(OW), I(cin), I(cout), I(kh), I(kw))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes > 0 else None
    if groups is not None and OW % 32 == 0:
        _call('vpho_conv2d_wgrad_groups_nhwc_f32', _f32(x), I(N), I(H), I(W), I(cin), I(ld), _f32(dy), I(OH), I(OW), I(cout), I(cout),
              I(kh), I(kw), I(stride), I(pad_y), I(pad_x), _i32(groups[0]), _i32(groups[1]), _f32(dw), _ptr(ws))
        return dw
    _call('vpho_conv2d_wgrad_nhwc_f32', _f32(x), I(N), I(H), I(W), I(cin), I(ld), _f32(dy), I(OH), I(OW), I(cout), I(cout),
          I(kh), I(kw), I(stride), I(pad_y), I(pad_x), _f32(dw), _ptr(ws))
    return dw


_BN_WS = {}


def _bn_workspace(Cc, device):
    """scratch of the BatchNorm reductions (256 chunks x 2 x C doubles), one per (device, stream): every user is stream-ordered and done with it
    when its call's last kernel has run; allocating it per call cost the training step ~270 allocator round trips"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    need = lib.vpho_bn_workspace_bytes(I(Cc))
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _BN_WS[key] = torch.empty(max(need, 256 * 2 * 2048 * 8), dtype=torch.uint8, device=device)
    return ws


def bn_train_forward(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, slope=1.0, res=None, partials=None):
    """x (..., C) NHWC -> y, (save_mean, save_invstd); running stats updated in place (nn.BatchNorm2d.train()).
    partials: the BnFuse the convolution that produced x filled -- the reduction pass over x is then skipped"""
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y, mean, invstd = torch.empty_like(x), _new((Cc,), x), _new((Cc,), x)
    ws = _bn_workspace(Cc, x.device)
    if partials is not None and partials.live():
        assert partials.stats.shape[-1] == Cc
        _call('vpho_bn_train_forward_stats_f32', _f32(x), LL(rows), I(Cc), I(Cc), _f32(partials.stats), I(partials.rows), _f32(gamma), _f32(beta), F(eps), F(momentum),
              F(slope), _f32(running_mean), _f32(running_var), _f32(mean), _f32(invstd), _f32(res), _f32(y), _ptr(ws))
        return y, (mean, invstd)
    _call('vpho_bn_train_forward_f32', _f32(x), LL(rows), I(Cc), I(Cc), _f32(gamma), _f32(beta), F(eps), F(momentum), F(slope),
          _f32(running_mean), _f32(running_var), _f32(mean), _f32(invstd), _f32(res), _f32(y), _ptr(ws))
    return y, (mean, invstd)


def bn_train_backward(x, dy, gamma, saved, partials=None, res=None, want_colsum=False):
    """-> dx, dgamma, dbeta (, column sums of dx).  partials: the BnFuse the convolution that produced dy filled (sum dy | sum dy * xhat).
    res: a gradient of x's shape added to dx (the identity shortcut of a pre-activation residual block); want_colsum: also return the
    column sums of the stored dx (the bias gradient of the convolution that produced x), taken while dx is written"""
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx, dg, db = torch.empty_like(x), _new((Cc,), x), _new((Cc,), x)
    ws = _bn_workspace(Cc, x.device)
    live = partials is not None and partials.x is not None and partials.live()
    if live or res is not None or want_colsum:
        cs = _new((Cc,), x) if want_colsum else None
        if live:
            assert partials.stats.shape[-1] == Cc
        assert res is None or (res.shape == x.shape and res.is_contiguous())
        _call('vpho_bn_train_backward_stats_f32', _f32(x), _f32(dy), LL(rows), I(Cc), I(Cc), _f32(gamma), _f32(saved[0]), _f32(saved[1]),
              _f32(partials.stats) if live else None, I(partials.rows if live else 0), _f32(res), _f32(dx), _f32(dg), _f32(db), _f32(cs), _ptr(ws))
        return (dx, dg, db, cs) if want_colsum else (dx, dg, db)
    _call('vpho_bn_train_backward_f32', _f32(x), _f32(dy), LL(rows), I(Cc), I(Cc), _f32(gamma), _f32(saved[0]), _f32(saved[1]), _f32(dx), _f32(dg), _f32(db), _ptr(ws))
    return dx, dg, db


def lrelu_bwd(dy, y, slope):
    dx = torch.empty_like(dy)
    _call('vpho_lrelu_bwd_f32', _f32(dy), _f32(y), LL(dy.numel()), F(slope), _f32(dx))
    return dx


def maxpool_bwd(x, dy, k, stride, pad):
    N, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    need = lib.vpho_maxpool_bwd_workspace_bytes(N, H, W, Cc, k, stride, pad)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device) if need > 0 else None
    _call('vpho_maxpool_bwd_ws_nhwc_f32', _f32(x), _f32(dy), I(N), I(H), I(W), I(Cc), I(k), I(stride), I(pad), _f32(dx), _u8(ws))
    return dx


def resize_bilinear_bwd(dy, H, W):
    N, OH, OW, Cc = dy.shape
    dx = _new((N, H, W, Cc), dy)
    _call('vpho_resize_bilinear_bwd_nhwc_f32', _f32(dy), I(N), I(OH), I(OW), I(Cc), I(H), I(W), _f32(dx))
    return dx


def roi_align_bwd(dy, boxes, feat_hw, C, scale, flip_w=None, c_off=0, into=None):
    """dy (N,P,P,ld) -> feature gradient (N,H,W,C); `into` accumulates into an existing gradient tensor"""
    N, P, _, ld = dy.shape
    H, W = feat_hw
    df = torch.zeros((N, H, W, C), device=dy.device, dtype=dy.dtype) if into is None else into
    _call('vpho_roi_align_bwd_nhwc_f32', _f32(dy), I(ld), I(c_off), I(N), I(H), I(W), I(C), _f32(boxes), F(scale), I(P), _u8(flip_w), _f32(df))
    return df


def align_heatmap_bwd(dout, bbox, bbox_rect, flip_w=None):
    N, S, _, Cc = dout.shape
    dhm = torch.zeros_like(dout)
    _call('vpho_align_heatmap_bwd_nhwc_f32', _f32(dout), I(N), I(S), I(Cc), _f32(bbox), _f32(bbox_rect), _u8(flip_w), _f32(dhm))
    return dhm


def add_lrelu(a, b, slope=1.0):
    y = torch.empty_like(a)
    _call('vpho_add_lrelu_f32', _f32(a), _f32(b), LL(a.numel()), F(slope), _f32(y))
    return y


def adamw_(param, grad, m, v, step, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
    """in-place torch.optim.AdamW step on one tensor"""
    _call('vpho_adamw_f32', _f32(param), _f32(grad), _f32(m), _f32(v), LL(param.numel()), F(lr), F(beta1), F(beta2), F(eps), F(weight_decay),
          I(step), F(grad_scale))


class AdamWList:
    """torch.optim.AdamW on a fixed list of (param, grad, exp_avg, exp_avg_sq) tensors as ONE launch (vpho_adamw_multi_f32)"""
    def __init__(self, quads):
        self.keep = quads
        rows, blk = [], 0
        for p, g, m, v in quads:
            for t in (p, g, m, v):
                _ptr(t, torch.float32)
            assert p.numel() == g.numel() == m.numel() == v.numel()
            rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), blk])
            blk += (p.numel() + 1023) // 1024
        self.blocks, self.n = blk, len(rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(quads[0][0].device)

    def step(self, step, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
        _call('vpho_adamw_multi_f32', _ptr(self.table, torch.int64), I(self.n), LL(self.blocks), F(lr), F(beta1), F(beta2), F(eps), F(weight_decay),
              I(step), F(grad_scale))
        # the kernel wrote through raw pointers: tell torch, so that everything cached per weight VERSION (the Winograd transforms of
        # winograd_weights_device, the bf16 planes) is rebuilt -- no launch, a counter per tensor
        torch.autograd.graph.increment_version([q[0] for q in self.keep])


# ----------------------------------------------------------------------------------------------- physics branch (training)
def cross_tokens_bwd(dtok, want_hand=True, want_obj=True):
    bs = dtok.shape[0]
    dph = torch.zeros((bs, 8, 8, 256), device=dtok.device) if want_hand else None
    dpo = torch.zeros((bs, 8, 8, 256), device=dtok.device) if want_obj else None
    dge = _new((bs, 512), dtok)
    _call('vpho_cross_tokens_bwd_f32', _f32(dtok), I(bs), _f32(dph), _f32(dpo), _f32(dge))
    return dph, dpo, dge


def layernorm_bwd(x, r, gamma, dy, eps=1e-5):
    E = x.shape[-1]
    dx, gx = torch.empty_like(x), torch.empty_like(x)
    _call('vpho_layernorm_bwd_f32', _f32(x), _f32(r), _f32(gamma), _f32(dy), LL(x.numel() // E), I(E), F(eps), _f32(dx), _f32(gx))
    return dx, gx


def mha_bwd(qkv, d_out, S, B, E, nhead, drop=None):
    dqkv = torch.empty_like(qkv)
    need = lib.vpho_mha_bwd_workspace_bytes(I(S), I(B), I(nhead))
    if need < 0:
        raise VphoError(f'vpho_mha_bwd_workspace_bytes: bad argument (S={S}, B={B}, nhead={nhead}; at most 1024 positions)')
    ws = torch.empty(need, dtype=torch.uint8, device=qkv.device) if need > 0 else None
    _call('vpho_mha_bwd_ws_f32', _f32(qkv), _f32(d_out), I(S), I(B), I(E), I(nhead), _f32(drop), _f32(dqkv), _ptr(ws))
    return dqkv


def physics_loss(scale_raw, logits, com, anchor, frame, point, gt_force_local, gravity, gt_com, is_grasped, weights, friction=0.8):
    """-> force_local (bs*32,3), losses (5,) fp64 device [force, gravity, torque, supervised, CoM] (weighted), d scale_raw / d logits / d com"""
    bs = gravity.shape[0]
    fl, dsc, dlg, dcm = _new((bs * 32, 3), com), _new((bs * 32, 1), com), _new((bs * 32, 8), com), _new((bs * 32, 3), com)
    losses, ws = _new((5,), com, torch.float64), _new((bs * 5,), com, torch.float64)
    w5 = (C.c_float * 5)(*[float(w) for w in weights])
    _call('vpho_physics_loss_f32', _f32(scale_raw), _f32(logits), _f32(com), _f32(anchor), F(friction), _f32(frame), _f32(point),
          _f32(gt_force_local), _f32(gravity), _f32(gt_com), _u8(is_grasped), w5, I(bs), _f32(fl), _f32(dsc), _f32(dlg), _f32(dcm),
          _f64(losses), _f64(ws))
    return fl, losses, dsc, dlg, dcm


# ----------------------------------------------------------------------------------------------- profiling hooks
PROF_CLASSES = {'conv_igemm_128x128': 0, 'conv_igemm_64x64': 1, 'score_head': 2, 'conv_igemm_128x64': 3,
                'mano_fk': 4, 'obj_physics': 5, 'hand_fuse': 6, 'roi_align': 7, 'resize_bilinear': 8, 'conv_winograd': 9,
                'conv_wgrad_64x64': 10, 'conv_wgrad_128x128': 11, 'force_optim': 12, 'pose_encoder': 13}


_prof_on = False


def prof_enable(name, on=True):
    global _prof_on
    _prof_on = bool(on)
    _check(lib.vpho_prof_enable(I(PROF_CLASSES[name]), I(1 if on else 0)))


def prof_collect(name):
    ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
    _check(lib.vpho_prof_collect(I(PROF_CLASSES[name]), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
    return dict(total_ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
