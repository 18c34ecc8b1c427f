"""SECOND, independent formulations of the three third-party leaves: none of them shares code or
an algorithm with ``oracle/rotations.py``, ``oracle/roi_align.py``, ``oracle/mano.py`` or the HIP kernels, so a slip in a
restatement cannot cancel against itself (round 3's float32-geometry bug sat in a leaf that had ONE restatement).

* rotations: ``scipy.spatial.transform.Rotation`` (installed; scalar-LAST quaternions, rotation vectors with angle in [0, pi]);
* RoIAlign (torchvision ``roi_align``, aligned=False, sampling_ratio=-1; call sites VPHO.py:125-128): the float32 sample
  coordinates built explicitly as one (K, ph*gh, pw*gw) grid, the interpolation done by ``F.grid_sample(align_corners=True,
  padding_mode='border')`` in float64, the kernel's "outside [-1, H] contributes zero" rule as a mask, bins = mean pooling;
* MANO (manopth ``ManoLayer``; call site head_mano.py:78-87): the equations of the MANO / SMPL papers in float64 numpy --
  Rodrigues' formula through scipy, blend shapes as tensor contractions, world transforms by walking a PARENT table,
  skinning as a weighted sum of 4x4 matrices; no quaternions, no level lists.
Test infrastructure only.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.spatial.transform import Rotation


# ------------------------------------------------------------------------------------------------------------ rotations
def scipy_quat_wxyz(rot, canonical=True):
    q = rot.as_quat(canonical=canonical)                                   # x, y, z, w
    return np.concatenate([q[..., 3:], q[..., :3]], -1)


def rot_from_wxyz(q):
    q = np.asarray(q, np.float64)
    return Rotation.from_quat(np.concatenate([q[..., 1:], q[..., :1]], -1))


def rot6d_to_matrix_by_cross_products(d6):
    """not Gram-Schmidt: the third axis first (normalised a1 x a2), the second as b3 x b1"""
    d6 = np.asarray(d6, np.float64)
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / np.linalg.norm(a1, axis=-1, keepdims=True)
    b3 = np.cross(b1, a2)
    b3 = b3 / np.linalg.norm(b3, axis=-1, keepdims=True)
    b2 = np.cross(b3, b1)
    return np.stack([b1, b2, b3], -2)                                      # rows (pytorch3d's convention)


def rotation_angle_between(m1, m2):
    """geodesic distance (rad) between rotation matrices: a comparison that has no branch cuts"""
    r = np.einsum('...ij,...kj->...ik', np.asarray(m1, np.float64), np.asarray(m2, np.float64))
    return np.linalg.norm(Rotation.from_matrix(r.reshape(-1, 3, 3)).as_rotvec(), axis=-1)


def random_rotations(n, seed, near_pi=0, tiny=0):
    """n uniformly random rotations + ``near_pi`` with angle pi - U(0, 1e-3) + ``tiny`` with angle < 1e-6, as rotation vectors"""
    rng = np.random.default_rng(seed)
    rv = [Rotation.random(n, random_state=seed).as_rotvec()]
    for cnt, ang in ((near_pi, lambda k: np.pi - rng.uniform(0, 1e-3, k)), (tiny, lambda k: rng.uniform(0, 1e-6, k))):
        if cnt:
            ax = rng.normal(size=(cnt, 3))
            rv.append(ax / np.linalg.norm(ax, axis=-1, keepdims=True) * ang(cnt)[:, None])
    return np.concatenate(rv, 0)


# ------------------------------------------------------------------------------------------------------------- RoIAlign
def roi_align_by_grid_sample(feat, rois, out_size, spatial_scale):
    """feat (N, C, H, W); rois (K, 5) [image, x1, y1, x2, y2] -> (K, C, out, out) float64"""
    N, C, H, W = feat.shape
    f32 = torch.float32
    r = rois.to(f32)
    sc = torch.tensor(spatial_scale, dtype=f32)
    x1, y1, x2, y2 = (r[:, i] * sc for i in (1, 2, 3, 4))                   # T = float in the kernel: every step below stays float32
    one = torch.ones((), dtype=f32)
    rw, rh = torch.maximum(x2 - x1, one), torch.maximum(y2 - y1, one)
    P = torch.tensor(float(out_size), dtype=f32)
    bw, bh = rw / P, rh / P
    gw, gh = torch.ceil(rw / P).long(), torch.ceil(rh / P).long()
    out = torch.zeros((r.shape[0], C, out_size, out_size), dtype=torch.float64)
    featd = feat.double()
    for k in range(r.shape[0]):
        def coords(start, bin_, g):
            p = torch.arange(out_size, dtype=f32)[:, None]
            i = torch.arange(int(g), dtype=f32)[None, :]
            return ((start + p * bin_) + ((i + 0.5) * bin_) / torch.tensor(float(g), dtype=f32)).reshape(-1)     # (out * g,)
        ys, xs = coords(y1[k], bh[k], gh[k]), coords(x1[k], bw[k], gw[k])
        ok = ((ys >= -1.0) & (ys <= H))[:, None] & ((xs >= -1.0) & (xs <= W))[None, :]
        yn = ys.double() * 2 / (H - 1) - 1                                 # align_corners=True: -1 <-> pixel 0, +1 <-> pixel H-1
        xn = xs.double() * 2 / (W - 1) - 1
        grid = torch.stack(torch.broadcast_tensors(xn[None, :], yn[:, None]), -1)[None]
        img = featd[int(r[k, 0]):int(r[k, 0]) + 1]
        smp = F.grid_sample(img, grid, mode='bilinear', padding_mode='border', align_corners=True)[0] * ok.double()[None]
        g_h, g_w = int(gh[k]), int(gw[k])
        out[k] = smp.view(C, out_size, g_h, out_size, g_w).mean(dim=(2, 4))
    return out


def random_boxes(n, seed, size=256.0):
    """xyxy boxes in crop pixels: generic ones, boxes leaving the crop on every side, sub-pixel ones, and boxes whose width times
    the scale 1/4 is an EXACT multiple of the 32 bins in float32 (the case that bit round 3's oracle)"""
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g) * size
    half = torch.rand(n, 2, generator=g) * size * 0.6 + 0.5
    b = torch.cat([c - half, c + half], 1)
    k = n // 8
    b[:k] = torch.cat([c[:k] - 0.3 * torch.rand(k, 2, generator=g), c[:k] + 0.3 * torch.rand(k, 2, generator=g)], 1)      # sub-pixel at 1/4 scale
    w = (torch.randint(1, 3, (k, 1), generator=g) * 128).float()                                                         # 32 or 64 map pixels wide
    x0 = torch.rand(k, 2, generator=g) * 60 - 20
    b[k:2 * k] = torch.cat([x0, x0 + w], 1)
    b[2 * k] = torch.tensor([1.9469828605651855, 0.0, 257.9469909667969, 256.0])
    return b


# ----------------------------------------------------------------------------------------------------------------- MANO
MANO_PARENT = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]        # kintree of MANO_RIGHT (index, middle, little, ring, thumb)
MANO_TIPS = [745, 317, 444, 556, 673]                                       # manopth: thumb, index, middle, ring, little
# manopth's 21-joint order from [16 skeleton joints | 5 tips]: wrist, then thumb / index / middle / ring / little with their tips
MANO_21 = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def mano_lbs_fp64(mano, pose, betas):
    """pose (B, 48) axis-angle, betas (B, 10) -> verts (B, 778, 3), joints (B, 21, 3) in METRES, centred on the wrist
    (ManoLayer(center_idx=0, flat_hand_mean=True, use_pca=False) / 1000, head_mano.py:48-55,86-87).  Rodrigues' formula without
    manopth's 1e-8 added to the vector before the norm (a 1e-8 rad difference)."""
    a = {k: np.asarray(v, np.float64) for k, v in mano.items()}
    pose, betas = np.asarray(pose, np.float64), np.asarray(betas, np.float64)
    B = pose.shape[0]
    Rm = Rotation.from_rotvec(pose.reshape(-1, 3)).as_matrix().reshape(B, 16, 3, 3)
    v_shaped = a['v_template'][None] + np.einsum('vck,bk->bvc', a['shapedirs'], betas)
    J = np.einsum('jv,bvc->bjc', a['J_regressor'], v_shaped)
    feat = (Rm[:, 1:] - np.eye(3)).reshape(B, 135)
    v_posed = v_shaped + np.einsum('vcp,bp->bvc', a['posedirs'], feat)
    G = np.zeros((B, 16, 4, 4))
    for k in range(16):
        loc = np.zeros((B, 4, 4))
        loc[:, :3, :3] = Rm[:, k]
        loc[:, 3, 3] = 1
        p = MANO_PARENT[k]
        loc[:, :3, 3] = J[:, k] - (J[:, p] if p >= 0 else 0)
        G[:, k] = loc if p < 0 else G[:, p] @ loc
    joints16 = G[:, :, :3, 3].copy()
    Grel = G.copy()                                                        # remove the rest pose: G'_k = G_k [I | -J_k]
    Grel[:, :, :3, 3] -= np.einsum('bkij,bkj->bki', G[:, :, :3, :3], J)
    T = np.einsum('vk,bkij->bvij', a['weights'], Grel)
    vh = np.concatenate([v_posed, np.ones((B, 778, 1))], -1)
    verts = np.einsum('bvij,bvj->bvi', T, vh)[..., :3]
    j21 = np.concatenate([joints16, verts[:, MANO_TIPS]], 1)[:, MANO_21]
    root = j21[:, :1]
    return verts - root, j21 - root
