"""Hand-object contact detection -- numpy restatement of lib/utils/physics_fn.py:47-117 (detect_hand_and_object_contact:
nearest neighbour both ways, signed normal / tangential distance gates, double-sigmoid weight), :201-208
(ForceAnchor.get_force_contact) and :210-221 (check_is_grasped).  Nearest neighbours by brute force (the reference uses an
sklearn ball tree; identical up to exact distance ties).  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import numpy as np

FINGER_LABEL = dict(palm=[5, 12, 19, 18, 26, 25], thumb=[6, 0, 1, 2, 3, 4], index=[7, 8, 9, 11, 10], middle=[13, 14, 15, 17, 16],
                    ring=[20, 21, 22, 24, 23], pinky=[27, 28, 29, 31, 30])           # physics_fn.py:125-177


def _nn(q, t):
    d2 = ((q[:, None, :] - t[None, :, :]) ** 2).sum(-1)
    return d2.argmin(1)


def contact_weight(x, normal_thresh, decay):
    mid1, mid2 = (decay[0] + normal_thresh[0]) / 2, (decay[1] + normal_thresh[1]) / 2

    def fn(v):
        with np.errstate(over='ignore'):
            m1 = 1 + np.exp(-1600 * (v - mid1))
            m2 = 1 + np.exp(1600 * (v - mid2))
            m3 = 1 / (m1 * m2 + 1e-10)
        m3[~np.isfinite(m1)] = 0
        m3[~np.isfinite(m2)] = 0
        return m3
    return fn(np.asarray(x, dtype=np.float64)) / fn(np.array([0.0]))


def detect(hand_verts, hand_normals, obj_verts, obj_normals, normal_thresh=(-0.015, 0.01), vertical_thresh=0.01, decay=(-0.005, 0.005)):
    def one_way(q, qn, t):
        ind = _nn(q, t)
        vec = q - t[ind]
        nd = (vec * qn).sum(-1)
        vd = np.linalg.norm(vec - nd[..., None] * qn, axis=-1)
        mask = (nd > normal_thresh[0]) & (nd < normal_thresh[1]) & (vd < vertical_thresh)
        w = contact_weight(nd, normal_thresh, decay)
        w[~mask] = 0
        return w, mask, ind
    hw, _, _ = one_way(hand_verts, hand_normals, obj_verts)
    ow, om, oi = one_way(obj_verts, obj_normals, hand_verts)
    o2h = np.full(ow.shape, -1, dtype=np.int32)
    o2h[om] = oi[om]
    return hw, ow, o2h


def force_contact(anchor, hand_contact):
    face = np.asarray(anchor['face_vert_idx']).reshape(-1)
    aw = np.concatenate([np.ones((32, 1)), np.asarray(anchor['anchor_weight'], dtype=np.float64)], 1)
    fc = hand_contact[..., face].reshape(hand_contact.shape[:-1] + (32, 3))
    return (fc * (aw / aw.sum(1, keepdims=True))).sum(-1)


def is_grasped(fc, thresh=0.0):
    return sum(int(fc[..., FINGER_LABEL[k]].sum(-1) > thresh) for k in FINGER_LABEL) >= 2
