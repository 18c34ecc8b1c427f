"""RoIAlign -- restatement of torchvision 0.17.0 ops.roi_align CPU kernel semantics
(roi_align_kernel.cpp: legacy aligned=False, sampling_ratio=-1 adaptive grid, bilinear
with the y<-1||y>H => 0 rule, clamp to [0,H-1]); call sites VPHO.py:125-128
(output_size=32, spatial_scale=1/4).
"""
import math
import torch


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
        # float32 arithmetic like the C++ kernel (T = float)
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        x1 = f32(f32(rois[k, 1]) * spatial_scale - offset)
        y1 = f32(f32(rois[k, 2]) * spatial_scale - offset)
        x2 = f32(f32(rois[k, 3]) * spatial_scale - offset)
        y2 = f32(f32(rois[k, 4]) * spatial_scale - offset)
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = max(gh * gw, 1)
        for i in range(ph):
            for j in range(pw):
                acc = feat.new_zeros(C)
                for iy in range(gh):
                    y = y1 + i * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = x1 + j * bw + (ix + 0.5) * bw / gw
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
        x1, y1, x2, y2 = [(rois[k, i] * spatial_scale).item() for i in range(1, 5)]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bh, bw = rh / ph, rw / pw
        gh, gw = int(math.ceil(rh / ph)), int(math.ceil(rw / pw))
        ii = torch.arange(ph, dtype=torch.float32)
        jj = torch.arange(pw, dtype=torch.float32)
        acc = feat.new_zeros(C, ph, pw)
        for iy in range(gh):
            y = y1 + ii * bh + (iy + 0.5) * bh / gh
            for ix in range(gw):
                x = x1 + jj * bw + (ix + 0.5) * bw / gw
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
