"""fp32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct convolution (CPU emulation of the arithmetic: transforms and the
per-frequency channel sums in float32, weights transformed in float64 like pack.winograd_weights), relative to the output range, against a
float64 convolution.  Layer shapes of the feature path's >= 32 x 32 stride-1 3x3 convolutions.  No GPU; ~1 minute.
    python scripts/wino_f4_error_study.py"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
f32, f64 = torch.float32, torch.float64
# F(2x2,3x3)
B2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=f64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=f64)
A2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=f64)
# F(4x4,3x3), Lavin & Gray's points {0, +-1, +-2}
B4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=f64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=f64)
A4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=f64)
# F(4x4,3x3) with the points {0, +-1/2, +-1} scaled (a better-conditioned choice): built from the Vandermonde construction
def toom_cook(points, m, r):
    """Winograd matrices (A^T, G, B^T) of F(m, r) for the given finite points + infinity (Toom-Cook), float64"""
    n = m + r - 1
    pts = list(points)
    assert len(pts) == n - 1
    V = lambda rows, cols: np.array([[p ** j for j in range(cols)] for p in pts] + [[0] * (cols - 1) + [1]], dtype=np.float64)
    At = V(n, m).T                                     # (m, n)
    Gm = V(n, r)                                       # (n, r)
    # scale rows of G by 1 / prod_{k != i} (p_i - p_k); B^T from the inverse transposed Vandermonde of the full polynomial basis
    N = np.array([np.prod([pts[i] - pts[k] for k in range(n - 1) if k != i]) for i in range(n - 1)] + [1.0])
    Gm = Gm / N[:, None]
    Vn = np.array([[p ** j for j in range(n)] for p in pts] + [[0] * (n - 1) + [1]], dtype=np.float64)
    Bt = (np.linalg.inv(Vn).T * N[:, None])            # rows scaled back
    # M(x) = prod (x - p_k) contributes to the "infinity" row: standard correction
    coef = np.poly(pts)[::-1]                          # ascending coefficients of M(x), degree n-1
    Bt[-1] = coef
    return torch.tensor(At), torch.tensor(Gm), torch.tensor(Bt)


def check(At, G, Bt, m):
    """exactness of a matrix triple in float64 on a random 1-D problem"""
    g, d = torch.randn(3, dtype=f64), torch.randn(m + 2, dtype=f64)
    y = At @ ((G @ g) * (Bt @ d))
    ref = torch.stack([(d[i:i + 3] * g).sum() for i in range(m)])
    return float((y - ref).abs().max())


def wino(x, w, At, G, Bt, m):
    """x (N,C,H,W) f32, w (K,C,3,3) f32 -> (N,K,H,W): F(m x m, 3x3), pad 1; transforms and channel sums in float32"""
    N, C, H, W = x.shape
    K = w.shape[0]
    n = m + 2
    U = torch.einsum('ar,kcrs,bs->abkc', G, w.double(), G).float()                     # weights in fp64, stored fp32 (like the product)
    xp = F.pad(x, (1, 1, 1, 1))
    th, tw = H // m, W // m
    # patches (N, C, th, tw, n, n)
    p = xp.unfold(2, n, m).unfold(3, n, m)
    Btf, Atf = Bt.float(), At.float()
    V = torch.einsum('ai,nctuij,bj->abnctu', Btf, p, Btf)                              # fp32 input transform
    M = torch.einsum('abkc,abnctu->abnktu', U, V)                                      # fp32 sums over channels (einsum accumulates in fp32)
    Y = torch.einsum('ia,abnktu,jb->nktiuj', Atf, M, Atf)                              # fp32 output transform
    return Y.reshape(N, K, th * m, tw * m)


for name, (At, G, Bt, m) in {'F(2x2,3x3)': (A2, G2, B2, 2), 'F(4x4,3x3) points 0,+-1,+-2': (A4, G4, B4, 4)}.items():
    print(f'{name}: exactness in float64 {check(At, G, Bt, m):.1e}')
alt = toom_cook([0.0, 0.5, -0.5, 1.0, -1.0], 4, 3)
print(f'F(4x4,3x3) points 0,+-1/2,+-1: exactness in float64 {check(*alt, 4):.1e}')
print('layer                      direct fp32     F(2x2)          F(4x4) 0,+-1,+-2   F(4x4) 0,+-1/2,+-1     (max |err| / max |y|;  rms err / rms y)')
for (N, H, C, K) in [(2, 64, 64, 64), (2, 32, 128, 128), (2, 64, 256, 256), (2, 32, 256, 128)]:
    x = torch.randn(N, C, H, H)
    x = torch.where(x > 0, x, 0.01 * x)                                                # post-LeakyReLU statistics
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    sc, rms = float(ref.abs().max()), float(ref.pow(2).mean().sqrt())
    out = {'direct': F.conv2d(x, w, padding=1), 'f2': wino(x, w, A2, G2, B2, 2), 'f4': wino(x, w, A4, G4, B4, 4), 'f4b': wino(x, w, *alt, 4)}
    e = {k: ((v.double() - ref).abs().max().item() / sc, (v.double() - ref).pow(2).mean().sqrt().item() / rms) for k, v in out.items()}
    print(f'{C:4d} -> {K:4d} at {H:2d} x {H:2d}   ' + '   '.join(f'{e[k][0]:.1e} / {e[k][1]:.1e}' for k in ('direct', 'f2', 'f4', 'f4b')))
