"""RoIAlign -- restatement of torchvision 0.17.0 ops.roi_align CPU kernel semantics
(roi_align_kernel.cpp: legacy aligned=False, sampling_ratio=-1 adaptive grid, bilinear
with the y<-1||y>H => 0 rule, clamp to [0,H-1]); call sites VPHO.py:125-128
(output_size=32, spatial_scale=1/4).

All RoI geometry in float32, like the kernel instantiated for float tensors (T = float): box edges, width / height, bin sizes,
the adaptive grid counts ceil(roi_width / pooled_width) and the sample coordinates.  This is not a nicety: the square hull of a
hand box that spans the whole 256-pixel crop is exactly 64 map pixels wide in float32 (grid count 2) and 64.000002 in double (grid
count 3) -- a different set of sample points for the whole RoI.
"""
import math

import numpy as np
import torch

_f = np.float32


def _geometry(roi, spatial_scale, ph, pw, offset=0.0, aligned=False):
    """float32 RoI geometry of roi_align_kernel.cpp: (x1, y1, bin_h, bin_w, grid_h, grid_w) with x1.. as np.float32"""
    sc, off = _f(spatial_scale), _f(offset)
    x1, y1, x2, y2 = (_f(_f(v) * sc - off) for v in roi)
    rw, rh = _f(x2 - x1), _f(y2 - y1)
    if not aligned:
        rw, rh = max(rw, _f(1.0)), max(rh, _f(1.0))
    bh, bw = _f(rh / _f(ph)), _f(rw / _f(pw))
    return x1, y1, bh, bw, int(math.ceil(_f(rh / _f(ph)))), int(math.ceil(_f(rw / _f(pw))))


def _bilinear(feat, y, x):
    # feat (C,H,W); y,x python floats
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return feat.new_zeros(C)
    y = max(y, 0.0)
    x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return (hy * hx) * feat[:, y_low, x_low] + (hy * lx) * feat[:, y_low, x_high] + \
           (ly * hx) * feat[:, y_high, x_low] + (ly * lx) * feat[:, y_high, x_high]


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """feat (N,C,H,W) f32; rois (K,5) [batch_idx,x1,y1,x2,y2] -> (K,C,ph,pw).  Scalar loops over
    bins/samples (vectorised over channels) -- small cases only."""
    ph, pw = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
    K = rois.shape[0]
    N, C, H, W = feat.shape
    out = feat.new_zeros(K, C, ph, pw)
    offset = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, bh, bw, gh, gw = _geometry([float(v) for v in rois[k, 1:5]], spatial_scale, ph, pw, offset, aligned)
        if sampling_ratio > 0:
            gh = gw = sampling_ratio
        count = max(gh * gw, 1)
        for i in range(ph):
            for j in range(pw):
                acc = feat.new_zeros(C)
                for iy in range(gh):
                    y = float(_f(_f(y1 + _f(_f(i) * bh)) + _f(_f(_f(iy + 0.5) * bh) / _f(gh))))
                    for ix in range(gw):
                        x = float(_f(_f(x1 + _f(_f(j) * bw)) + _f(_f(_f(ix + 0.5) * bw) / _f(gw))))
                        acc = acc + _bilinear(feat[b], y, x)
                out[k, :, i, j] = acc / count
    return out


def roi_align_fast(feat, rois, output_size, spatial_scale=1.0):
    """Vectorised (over bins) version of the same algorithm for aligned=False, sampling_ratio=-1;
    used for full-size oracle runs.  Equivalent to ``roi_align`` up to fp32 summation order."""
    ph, pw = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
    K = rois.shape[0]
    N, C, H, W = feat.shape
    out = feat.new_zeros(K, C, ph, pw)
    rois = rois.float()
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, bh, bw, gh, gw = _geometry([float(v) for v in rois[k, 1:5]], spatial_scale, ph, pw)
        t32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        ii = torch.arange(ph, dtype=torch.float32)
        jj = torch.arange(pw, dtype=torch.float32)
        acc = feat.new_zeros(C, ph, pw)
        for iy in range(gh):
            y = (t32(y1) + ii * t32(bh)) + (t32(iy + 0.5) * t32(bh)) / t32(gh)          # float32 tensor arithmetic, the kernel's order
            for ix in range(gw):
                x = (t32(x1) + jj * t32(bw)) + (t32(ix + 0.5) * t32(bw)) / t32(gw)
                acc = acc + _bilinear_grid(feat[b], y, x)
        out[k] = acc / max(gh * gw, 1)
    return out


def _bilinear_grid(feat, y, x):
    C, H, W = feat.shape
    vy = ~((y < -1.0) | (y > H))
    vx = ~((x < -1.0) | (x > W))
    y = y.clamp(min=0.0)
    x = x.clamp(min=0.0)
    yl, xl = y.long(), x.long()
    ycap, xcap = yl >= H - 1, xl >= W - 1
    yl = torch.where(ycap, torch.full_like(yl, H - 1), yl)
    xl = torch.where(xcap, torch.full_like(xl, W - 1), xl)
    yh = torch.where(ycap, yl, yl + 1)
    xh = torch.where(xcap, xl, xl + 1)
    y = torch.where(ycap, yl.float(), y)
    x = torch.where(xcap, xl.float(), x)
    ly, lx = y - yl.float(), x - xl.float()
    hy, hx = 1 - ly, 1 - lx
    g = lambda a, b: feat[:, a][:, :, b]
    v = (hy[:, None] * hx[None]) * g(yl, xl) + (hy[:, None] * lx[None]) * g(yl, xh) + \
        (ly[:, None] * hx[None]) * g(yh, xl) + (ly[:, None] * lx[None]) * g(yh, xh)
    return v * (vy[:, None] & vx[None]).float()
