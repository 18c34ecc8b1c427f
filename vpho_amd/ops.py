"""ctypes binding of the C ABI in ``include/vpho_hip.h`` (``vpho_amd/libvpho_hip.so``).

Thin by design: tensors in, raw device pointers + the current HIP stream out.  There is NO fallback: importing this
module without the built extension raises, and every op checks that its tensors live on a GPU.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvpho_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} is missing: build the HIP extension first (python -m vpho_amd.build or '
                      f'__graft_entry__.build()); vpho_amd has no CPU fallback')
lib = C.CDLL(LIB_PATH)
lib.vpho_last_error.restype = C.c_char_p
lib.vpho_abi_version.restype = C.c_int


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
                ('in_slope', C.c_float), ('out_slope', C.c_float)]


lib.vpho_conv2d_nhwc_f32.argtypes = [C.POINTER(ConvDesc), C.c_void_p]


def _addr(t):
    return None if t is None else t.data_ptr()


def conv2d_nhwc(x, w, bias=None, *, kh=1, kw=1, stride=1, pad=0, pad_y=None, pad_x=None, out=None, out_hw=None,
                out_view=None, res=None, in_scale=None, in_shift=None, in_slope=1.0, out_slope=1.0, cin=None):
    """x: (N,H,W,x_ld) fp32 NHWC, w: (Cout, kh*kw*Cin) packed.  Returns (N,OH,OW,Cout) (or writes ``out``).

    ``out_view`` = (tensor, y_sn, y_sy, y_sx, element_offset) writes into a strided destination (concat buffers,
    transposed-convolution phases).  ``res`` is a contiguous (N,OH,OW,Cout) tensor."""
    N, H, W, x_ld = x.shape
    cin = x_ld if cin is None else cin
    cout = w.shape[0]
    assert w.shape[1] == kh * kw * cin, (w.shape, kh, kw, cin)
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
    if out_view is not None:
        yt, d.y_sn, d.y_sy, d.y_sx, off = out_view
        _ptr(yt, torch.float32)
        d.y = yt.data_ptr() + 4 * off
        ret = yt
    else:
        if out is None:
            out = torch.empty((N, OH, OW, cout), device=x.device, dtype=torch.float32)
        _ptr(out, torch.float32)
        ld = out.shape[-1]
        d.y, d.y_sx, d.y_sy, d.y_sn = out.data_ptr(), ld, ld * OW, ld * OW * OH
        ret = out
    if res is not None:
        assert res.shape == (N, OH, OW, cout)
        d.r_sx, d.r_sy, d.r_sn = cout, cout * OW, cout * OW * OH
    d.N, d.H, d.W, d.Cin, d.x_ld = N, H, W, cin, x_ld
    d.Cout, d.KH, d.KW, d.stride, d.pad_y, d.pad_x, d.OH, d.OW = cout, kh, kw, stride, py, px, OH, OW
    d.in_slope, d.out_slope = in_slope, out_slope
    _check(lib.vpho_conv2d_nhwc_f32(C.byref(d), _stream()))
    return ret


def linear(x, w, bias=None, out_slope=1.0, out=None):
    """x: (rows, cin) fp32, w: (cout, cin) -> (rows, cout).  Same kernel as conv2d_nhwc (1x1)."""
    rows, cin = x.shape
    y = conv2d_nhwc(x.view(rows, 1, 1, cin), w, bias, out_slope=out_slope,
                    out=None if out is None else out.view(rows, 1, 1, w.shape[0]))
    return y.view(rows, w.shape[0])


# ----------------------------------------------------------------------------------------------- score net / sampler
class ScoreWeights(C.Structure):
    _fields_ = [('D', C.c_int), ('Dp', C.c_int), ('nheads', C.c_int)] + \
               [(k, C.c_void_p) for k in ('t_W', 't_w', 't_b', 'pe0_w', 'pe0_b', 'pe2_w', 'pe2_b', 'w1_t', 'w1_p',
                                          'w1_f', 'b1', 'w2', 'b2')]


class OdeStats(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('nfev', 'n_accepted', 'n_rejected', 'nan_count', 'status', 'n_log')]


lib.vpho_score_workspace_bytes.restype = C.c_longlong
lib.vpho_score_workspace_bytes.argtypes = [C.POINTER(ScoreWeights), C.c_int, C.c_int]
lib.vpho_score_eval.argtypes = [C.POINTER(ScoreWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                C.c_void_p, C.c_longlong, C.c_void_p]
lib.vpho_ode_sample.argtypes = [C.POINTER(ScoreWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double,
                                C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong,
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

    def sample(self, feat_img, init_x, S, T0, num_steps, xs_f64, eps=1e-5, rtol=3e-3, atol=3e-4, log_cap=4096):
        """cond_ode_sampler (score_based_model.py:45-105).  Returns xs (R,steps,D), x (R,D) f64, stats dict."""
        bs = feat_img.shape[0]
        R = bs * S
        assert init_x.shape == (R, self.D)
        ws = self.workspace(bs, S)
        xs = torch.empty((R, num_steps, self.D), device=self.device, dtype=torch.float64 if xs_f64 else torch.float32)
        x = torch.empty((R, self.D), device=self.device, dtype=torch.float64)
        st = OdeStats()
        log = (C.c_double * (4 * log_cap))()
        _check(lib.vpho_ode_sample(C.byref(self.c), _ptr(feat_img, torch.float32), bs, S, _ptr(init_x, torch.float32),
                                   float(T0), float(eps), int(num_steps), float(rtol), float(atol), _ptr(xs),
                                   1 if xs_f64 else 0, _ptr(x), _ptr(ws), ws.numel(), C.byref(st), log, log_cap, _stream()))
        n = min(st.n_log, log_cap)
        steps = [(log[4 * i], log[4 * i + 1], log[4 * i + 2], bool(log[4 * i + 3])) for i in range(n)]
        return xs, x, dict(nfev=st.nfev, n_accepted=st.n_accepted, n_rejected=st.n_rejected, nan_count=st.nan_count,
                           steps=steps)
