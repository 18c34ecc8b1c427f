"""Weight packing for the HIP kernels: NHWC implicit-GEMM layout, eval-mode BatchNorm folding."""
import torch


def bn_affine(sd, p, eps=1e-5):
    """Eval-mode BatchNorm as per-channel (scale, shift)."""
    scale = sd[p + '.weight'] / torch.sqrt(sd[p + '.running_var'] + eps)
    return scale, sd[p + '.bias'] - sd[p + '.running_mean'] * scale


def pack_conv(w, cin_pad=None):
    """(Cout, Cin, kh, kw) -> (Cout, kh*kw*cin_pad) with k = (r*kw + s)*cin_pad + c."""
    cout, cin, kh, kw = w.shape
    cin_pad = cin if cin_pad is None else cin_pad
    wp = w.permute(0, 2, 3, 1)
    if cin_pad != cin:
        wp = torch.nn.functional.pad(wp, (0, cin_pad - cin))
    return wp.reshape(cout, kh * kw * cin_pad).contiguous()


def fold_conv_bn(sd, conv, bn=None, cin_pad=None, device=None):
    """conv (+ following BatchNorm) -> (packed weight, bias) on ``device``."""
    w = sd[conv + '.weight'].float()
    b = sd.get(conv + '.bias')
    b = torch.zeros(w.shape[0]) if b is None else b.float()
    if bn is not None:
        s, t = bn_affine(sd, bn)
        w = w * s[:, None, None, None]
        b = b * s + t
    return pack_conv(w, cin_pad).to(device), b.contiguous().to(device)


def pack_deconv4x4s2(w):
    """ConvTranspose2d(k=4, s=2, p=1) weight (Cin, Cout, 4, 4) -> four 2x2 sub-convolutions, one per output parity
    (py, px): out[2a+py, 2b+px] = sum_{dy,dx in {0,1}} in[a+dy-(1-py), b+dx-(1-px)] * w[:, :, ky(py,dy), kx(px,dx)]
    with ky(0,.) = (3,1), ky(1,.) = (2,0).  Returns {(py,px): (Cout, 4*Cin) packed, pad_y, pad_x}."""
    tap = {0: (3, 1), 1: (2, 0)}
    out = {}
    for py in (0, 1):
        for px in (0, 1):
            sub = torch.stack([torch.stack([w[:, :, tap[py][dy], tap[px][dx]] for dx in (0, 1)], -1) for dy in (0, 1)], -2)
            # sub: (Cin, Cout, 2(dy), 2(dx)) -> conv weight (Cout, Cin, 2, 2)
            out[(py, px)] = (pack_conv(sub.permute(1, 0, 2, 3).contiguous()), 1 - py, 1 - px)
    return out


def winograd_weights(w_packed, cin=None):
    """packed 3x3 weights (Cout, 9*Cin) -> U = G g G^T of Winograd F(2x2, 3x3), computed in fp64 (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]),
    frequency f = 4*fy + fx, stored STAGE-TILED as (Cin/8, 16, Cout, 8): the 8 input channels x 16 frequencies x Cout rows the kernel
    fills into LDS for one k stage are one contiguous block (a fill instruction reads 1 KB of consecutive bytes instead of 32-byte
    pieces of 32 different rows).  Which 8 channels a stage holds follows the kernel's 16-byte input loads: stage 2 ss + e takes one
    channel pair of every 4-channel load of the 16-channel group ss: channels 16 ss + 4 (j >> 1) + 2 e + (j & 1) in slot j."""
    cout = w_packed.shape[0]
    cin = w_packed.shape[1] // 9 if cin is None else cin
    # on the HOST: a one-time weight transform, and a torch contraction on the device would pull rocBLAS kernels into the process
    g = w_packed.detach().double().cpu().view(cout, 3, 3, cin)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum('ar,orsc,bs->aboc', G, g, G)                     # (4, 4, Cout, Cin)
    assert cin % 16 == 0
    j = torch.arange(8)
    chan = (16 * torch.arange(cin // 16)[:, None, None] + 2 * torch.arange(2)[None, :, None] + (4 * (j >> 1) + (j & 1))[None, None, :]).reshape(-1)
    U = U.reshape(16, cout, cin)[:, :, chan].reshape(16, cout, cin // 8, 8).permute(2, 0, 1, 3)          # (Cin/8, 16, Cout, 8)
    return U.float().contiguous().to(w_packed.device)
