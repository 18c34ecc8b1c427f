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
