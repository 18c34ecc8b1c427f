"""Hand metrics -- numpy restatement of the reference's TesterHand.criterion_MJE_PAMJE (lib/engine/test.py:657-680) and
rigid_transform_3D_AtoB / rigid_align_AtoB (lib/utils/transform_fn.py:43-66).  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import numpy as np


def rigid_align_AtoB(A, B):
    n = A.shape[0]
    cA, cB = A.mean(0), B.mean(0)
    H = (A - cA).T @ (B - cB) / n
    U, s, Vh = np.linalg.svd(H)
    R = Vh.T @ U.T
    if np.linalg.det(R) < 0:
        s[-1] = -s[-1]
        Vh[2] = -Vh[2]
        R = Vh.T @ U.T
    c = 1.0 / np.var(A, axis=0).sum() * np.sum(s)
    t = -(c * R) @ cA + cB
    return (c * R @ A.T).T + t


def mje_pamje(gt, pd):
    """gt, pd (n,3) -> mean error, Procrustes-aligned mean error, per-point errors (metres)."""
    je = np.linalg.norm(gt - pd, axis=-1)
    pa = np.linalg.norm(gt - rigid_align_AtoB(pd, gt), axis=-1).mean(-1)
    return je.mean(-1), pa, je


# ------------------------------------------------------------------ object metrics (lib/engine/test.py:155-193, 240-503)
FSCORE_TH = (0.002, 0.005, 0.010, 0.020, 0.050, 0.100)
OBJ_METRIC_NAMES = ('MCE', 'OCE', 'MCE2', 'ADD', 'ADDS', 'ADD01d', 'ADDS01d', 'REP', 'REP5', 'CD',
                    'FSCORE@2mm', 'FSCORE@5mm', 'FSCORE@10mm', 'FSCORE@2cm', 'FSCORE@5cm', 'FSCORE@10cm')


def obj_9d_to_rt(pose9, root_joint):
    """transform_fn.py:85-90 + train_diff_hand_obj.py:594-597: (n,9) rot6d+t, (n,3) -> (n,3,4) with the root added."""
    from . import rotations as R
    import torch
    p = torch.as_tensor(pose9)
    rt = torch.cat([R.rotation_6d_to_matrix(p[..., :6]), p[..., 6:9, None]], dim=-1)
    rt[..., 3] = rt[..., 3] + torch.as_tensor(root_joint).to(rt.dtype)
    return rt.numpy()


def _transform(pts, rt):
    """np.einsum("ni,ij->nj", pts, rt[:, :3].T) + rt[:, 3] in fp64 (the tables are fp64 in the reference, base.py:222-238)."""
    return np.asarray(pts, np.float64) @ np.asarray(rt[:, :3], np.float64).T + np.asarray(rt[:, 3], np.float64)


def _nn_min(a, b, chunk=512):
    """min_j ||a_i - b_j|| for fp32 point sets by direct differences.  The reference calls torch.cdist, whose matmul
    expansion (|a|^2 + |b|^2 - 2ab, used above 25 points) carries an absolute error of ~1e-7 in d^2 at camera-space
    magnitudes; the restatement is the exact quantity and is compared with the fixture at that tolerance."""
    a, b = a.astype(np.float32), b.astype(np.float32)
    out = np.empty(a.shape[0], np.float32)
    for i in range(0, a.shape[0], chunk):
        d = a[i:i + chunk, None, :] - b[None, :, :]
        out[i:i + chunk] = np.sqrt((d * d).sum(-1)).min(-1)
    return out


def object_metrics(mesh, pd_rt, gt_rt, cam_intr):
    """One sample, single hypothesis: mesh = {'bbox3d','verts_sampled','verts','diameter'}; pd_rt, gt_rt (3,4); cam_intr (3,3).
    Returns the 16 values of OBJ_METRIC_NAMES (TesterObject.__call__, test.py:240-352)."""
    # criterion_MCE_OCE (test.py:354-374), fp64
    pb, gb = _transform(mesh['bbox3d'], pd_rt), _transform(mesh['bbox3d'], gt_rt)
    mce = np.linalg.norm(pb - gb, axis=-1).mean(-1)
    oce = np.linalg.norm(pb.mean(-2) - gb.mean(-2), axis=-1)
    # sampled vertices: fp64 transform, then fp32 (test.py:417-421,441-445)
    pv64, gv64 = _transform(mesh['verts_sampled'], pd_rt), _transform(mesh['verts_sampled'], gt_rt)
    pv, gv = pv64.astype(np.float32), gv64.astype(np.float32)
    # criterion_MCE2 -> compute_obj_metrics_dexycb (test.py:155-193): corners of the two axis-aligned boxes
    sel = np.array([[0, 1, 0, 0, 1, 0, 1, 1], [0, 0, 1, 0, 1, 1, 0, 1], [0, 0, 0, 1, 0, 1, 1, 1]])
    def aabb(v):
        mm = np.stack([v.min(0), v.max(0)], 1)                      # (3,2)
        return np.stack([mm[0, sel[0]], mm[1, sel[1]], mm[2, sel[2]]], 1)
    mce2 = np.linalg.norm(aabb(pv) - aabb(gv), axis=-1).astype(np.float32).mean()
    # criterion_ADD_REP (test.py:425-458)
    add = np.linalg.norm(pv - gv, axis=-1).mean()
    adds = _nn_min(pv, gv).mean()
    K = np.asarray(cam_intr, np.float64)
    pp = (pv64 @ K.T) / (pv64[:, 2:3] + 1e-7)
    gp = (gv64 @ K.T) / (gv64[:, 2:3] + 1e-7)
    rep = np.linalg.norm(pp[:, :2] - gp[:, :2], axis=-1).mean()
    add01d, adds01d = float(add <= mesh['diameter'] * 0.1), float(adds <= mesh['diameter'] * 0.1)       # test.py:505-516
    rep5 = float(rep < 5)                                                                               # test.py:518-519
    # criterion_FSCORE (test.py:460-503): full vertex set, both directions
    pf, gf = _transform(mesh['verts'], pd_rt).astype(np.float32), _transform(mesh['verts'], gt_rt).astype(np.float32)
    d_p2g, d_g2p = _nn_min(pf, gf), _nn_min(gf, pf)
    cd = 0.5 * (d_p2g.mean() + d_g2p.mean())
    fs = []
    for th in FSCORE_TH:
        prec, rec = np.float32((d_p2g < th).mean()), np.float32((d_g2p < th).mean())
        fs.append(np.float32(2) * prec * rec / (prec + rec + np.float32(1e-6)))
    return np.array([mce, oce, mce2, add, adds, add01d, adds01d, rep, rep5, cd] + fs, np.float64)
