"""Candidate aggregation -- torch-CPU restatement of the reachable subset of the reference's
lib/model/aggregation.py (HOI_Aggregator.__call__ :1167-1353 and what it calls) together with
transform_fn.average_quaternion (:101-125), physics_fn.ForceAnchor.__call__ (:224-257),
hand_fn.Vert2Joint (:434-448), physics.from_local_to_global (:362-371), head_object.HeadObject (:36-67).
TEST INFRASTRUCTURE -- see oracle/__init__.py.

Top-k ties: the reference's ``torch.topk`` leaves the order of equal values unspecified; the oracle (and the HIP
kernel) define it as "larger value first, then smaller index" (stable descending sort).
"""
import torch
import torch.nn.functional as F

from . import rotations as R
from .mano import get_hand_verts

MANO_PARAMS_LEVEL = {0: [0, 1, 2],
                     1: [39, 40, 41, 3, 4, 5, 12, 13, 14, 30, 31, 32, 21, 22, 23],
                     2: [42, 43, 44, 6, 7, 8, 15, 16, 17, 33, 34, 35, 24, 25, 26],
                     3: [45, 46, 47, 9, 10, 11, 18, 19, 20, 36, 37, 38, 27, 28, 29]}      # hand_fn.py:240-247
MANO_JOINT_LEVEL = {0: [0], 1: [1, 5, 9, 13, 17], 2: [2, 6, 10, 14, 18], 3: [3, 7, 11, 15, 19],
                    4: [4, 8, 12, 16, 20]}                                                  # hand_fn.py:250-263
FINGER_FORCE_LEVEL = [[1, 2, 3, 4], [8, 9, 10, 11], [14, 15, 16, 17], [21, 22, 23, 24], [28, 29, 30, 31]]  # :584-590


def topk_stable(x, k, dim=1):
    if k > x.shape[dim]:
        raise RuntimeError('selected index k out of range')
    v, i = torch.sort(x, dim=dim, descending=True, stable=True)
    sl = [slice(None)] * x.dim()
    sl[dim] = slice(0, k)
    return v[tuple(sl)], i[tuple(sl)]


def project(pt3d, K):
    """aggregation.py:24-32"""
    p = torch.einsum('b...ij,blj->b...il', pt3d, K)
    return p[..., :2] / p[..., 2:]


def average_quaternion(Q, W=None):
    """transform_fn.py:101-125"""
    if W is None:
        W = torch.ones_like(Q[..., 0])
    wsum = W.sum(dim=-1, keepdim=True)
    oq = ((Q[..., 0:1] > 0).float() - 0.5) * 2 * Q
    A = torch.einsum('...ni,...nj->...nij', oq, oq)
    A = torch.sum(A * W[..., None, None], -3)
    A = A / wsum.reshape(*Q.shape[:-2], 1, 1)
    q = torch.linalg.eigh(A)[1][..., -1]
    return ((q[..., 0:1] > 0).float() - 0.5) * 2 * q


def average_rot6d(rot6d, weights=None):
    """aggregation.py:50-56"""
    if weights is None:
        weights = torch.ones_like(rot6d[..., 0]) / rot6d.shape[-2]
    q = average_quaternion(R.matrix_to_quaternion(R.rotation_6d_to_matrix(rot6d)), weights)
    return R.matrix_to_rotation_6d(R.quaternion_to_matrix(q))


def vert2anchor(anchor, anchor_skeleton, verts):
    """physics_fn.py:224-257 -> anchors (...,32,3), frames (...,32,3,3) [columns x,y,z]."""
    face = torch.as_tensor(anchor['face_vert_idx']).reshape(-1).long()
    aw = torch.as_tensor(anchor['anchor_weight']).to(verts.dtype)
    v2j = torch.as_tensor(anchor['vert2joint']).to(verts.dtype)
    sk = torch.as_tensor(anchor_skeleton).long()
    iv = verts[..., face, :].reshape(verts.shape[:-2] + (-1, 3, 3))
    b1 = iv[..., 1, :] - iv[..., 0, :]
    b2 = iv[..., 2, :] - iv[..., 0, :]
    joints = torch.einsum('...ij,ki->...kj', verts, v2j)
    dy = joints[..., sk[:, 1], :] - joints[..., sk[:, 0], :]
    dz = torch.cross(b1, b2, dim=-1)
    dz = dz / (dz.norm(dim=-1, keepdim=True) + 1e-8)
    dy = dy / (dy.norm(dim=-1, keepdim=True) + 1e-8)
    dx = torch.cross(dy, dz, dim=-1)
    dy = torch.cross(dz, dx, dim=-1)
    dy = dy / (dy.norm(dim=-1, keepdim=True) + 1e-8)
    frame = torch.stack([dx, dy, dz], dim=-1)
    pts = aw[:, 0:1] * b1 + aw[:, 1:2] * b2 + iv[..., 0, :]       # anchor_weight columns after the prepended 1
    return pts, frame


def local_to_global(anchor, anchor_skeleton, force_local, verts):
    """physics.py:362-371"""
    pts, frame = vert2anchor(anchor, anchor_skeleton, verts)
    return pts, torch.einsum('...bi,...bji->...bj', force_local, frame)


def object_points(ycb, pose, names, which):
    """head_object.py:36-61: R(rot6d) @ p + t for the per-name table ``which`` in {kpt3d, verts_sampled, CoM}."""
    pts = torch.stack([torch.as_tensor(ycb[n][which]).to(pose.dtype).reshape(-1, 3) for n in names], 0)
    rot = R.rotation_6d_to_matrix(pose[..., :6])
    return torch.einsum('bvi,b...ji->b...vj', pts, rot) + pose[..., 6:].unsqueeze(-2)


def flip_x(pt, is_right):
    """head_object.py:63-67 (sign flip of x where ~is_right), out of place."""
    sgn = torch.where(is_right, 1.0, -1.0).to(pt.dtype).reshape((-1,) + (1,) * (pt.dim() - 2))
    out = pt.clone()
    out[..., 0] = out[..., 0] * sgn
    return out


def _bicubic_lookup(heatmap, pt2d, ids):
    """aggregation.py:206-213 / 765-771: channel i sampled at point i, bicubic, zeros padding, align_corners=False."""
    vals = []
    for i in ids:
        v = F.grid_sample(heatmap[:, [i]], pt2d[:, :, [i]], align_corners=False, mode='bicubic')
        vals.append(v.squeeze(1))
    return torch.cat(vals, dim=-1)


def _norm_to_bbox(pt2d, bbox):
    b = bbox[:, None, None, :]
    return 2 * (pt2d - b[..., :2]) / (b[..., 2:] - b[..., :2]) - 1


# ------------------------------------------------------------------ hand cascade  (aggregation.py:115-284)
def hand_level_scores(mano, pose, betas, root_flip, K, heatmap, bbox, level):
    """aggregation.py:196-218,242-248: FK of every candidate, projection, bicubic look-up of the joints of the deeper levels,
    level sum (level 0) / per-finger mean (levels 1-3).  pose (bs,C,48), betas (bs,10) -> (bs,C) | (bs,C,5).  The arithmetic runs
    in pose.dtype (fp32 = the reference's; fp64 = the referee of oracle/referee.py)."""
    bs, C = pose.shape[:2]
    observe = [j for l in range(level + 1, 5) for j in MANO_JOINT_LEVEL[l]]
    shape = betas[:, None].expand(bs, C, 10).reshape(-1, 10)
    _, joint = get_hand_verts(mano, pose.reshape(-1, 48), shape)
    joint = joint.reshape(bs, C, 21, 3) + root_flip[:, None, None]
    pt2d = _norm_to_bbox(project(joint, K), bbox)
    hv = _bicubic_lookup(heatmap, pt2d, observe)                                            # (bs,C,m)
    if level == 0:
        return hv.sum(-1)
    return hv.reshape(bs, C, len(observe) // 5, 5).mean(dim=-2)                             # (bs,C,5)


def hand_cascade(mano, pose_diff, pose_reg, betas, root_flip, K, heatmap, bbox, k):
    """pose_diff (bs,S,48) f32, pose_reg (bs,48), betas (bs,10).  Returns dict with fused pose (bs,48), per-level
    top-k indices, level-3 top-k distal poses (bs,k,5,3), verts/joints of the fused pose."""
    bs, S = pose_diff.shape[:2]
    pose = torch.cat([pose_diff, pose_reg[:, None].expand(bs, S, 48)], 1).clone()          # (bs,2S,48)
    out = dict(topk=[], val=[], weight=[], score=[], state=[])
    for level in range(4):
        fuse = MANO_PARAMS_LEVEL[level]
        if level == 0:
            pose[:, S:, fuse] = pose[:, :S, fuse]                                           # quirk Q7
        out['state'].append(pose.clone())                                                   # the candidates this level scores
        hv = hand_level_scores(mano, pose, betas, root_flip, K, heatmap, bbox, level)       # (bs,2S) | (bs,2S,5)
        if level == 0:
            val, idx = topk_stable(hv, k, dim=1)                                            # (bs,k)
            w = (val + 1e-8) / (val.sum(dim=1, keepdim=True) + 1e-8)
            sel = torch.gather(pose[:, :, fuse], 1, idx[:, :, None].expand(bs, k, 3))       # (bs,k,3)
            q = R.axis_angle_to_quaternion(sel.reshape(bs, k, 1, 3)).permute(0, 2, 1, 3)    # (bs,1,k,4)
            aa = R.quaternion_to_axis_angle(average_quaternion(q, w[:, None])).reshape(bs, 3)
            pose[:, :, fuse] = pose[:, :, fuse] * 0 + aa[:, None]
        else:
            val, idx = topk_stable(hv, k, dim=1)                                            # (bs,k,5)
            w = ((val + 1e-8) / (val.sum(dim=1, keepdim=True) + 1e-8)).permute(0, 2, 1)     # (bs,5,k)
            jid = torch.tensor(fuse).reshape(5, 3)[:, 0] // 3
            p16 = pose.reshape(bs, 2 * S, 16, 3)
            bi = torch.arange(bs)[:, None, None].expand(bs, k, 5)
            sel = p16[bi, idx, jid[None, None].expand(bs, k, 5)]                            # (bs,k,5,3)
            q = R.axis_angle_to_quaternion(sel).permute(0, 2, 1, 3)                         # (bs,5,k,4)
            aa = R.quaternion_to_axis_angle(average_quaternion(q, w)).reshape(bs, 15)
            pose[:, :, fuse] = pose[:, :, fuse] * 0 + aa[:, None]
            out['topk_pose_l%d' % level] = sel
        out['topk'].append(idx)
        out['val'].append(val)
        out['weight'].append(w)
        out['score'].append(hv)                                                             # all candidates: (bs,2S) / (bs,2S,5)
    fused = pose[:, 0].clone()
    v, j = get_hand_verts(mano, fused, betas)
    out.update(fused_pose=fused, agg_vert=v, agg_joint=j)
    return out


# ------------------------------------------------------------------ object selection (aggregation.py:742-780,947-997)
def obj_heat_scores(ycb, pose6d, root, names, is_right, K, heatmap, bbox, dtype=torch.float32):
    """aggregation.py:753-777 (the reference casts the fp64 poses with .float(), :753; dtype=float64: the referee)"""
    p = pose6d.clone().to(dtype)
    p[..., 6:] = p[..., 6:] + root.unsqueeze(1)
    pt = flip_x(object_points(ycb, p, names, 'kpt3d'), is_right)
    pt2d = _norm_to_bbox(project(pt, K), bbox)
    return _bicubic_lookup(heatmap, pt2d, list(range(heatmap.shape[1]))).sum(-1)


def obj_heat_topk(ycb, pose6d, root, names, is_right, K, heatmap, bbox, k, dtype=torch.float32):
    hv = obj_heat_scores(ycb, pose6d, root, names, is_right, K, heatmap, bbox, dtype=dtype)
    val, idx = topk_stable(hv, k, dim=1)
    return idx, (val + 1e-8) / (val.sum(dim=1, keepdim=True) + 1e-8), hv


def nearest(x, y):
    """min / argmin over y of ||x - y|| by direct differences (the reference uses torch.cdist, aggregation.py:1115-1158)."""
    d = (x[..., :, None, :] - y[..., None, :, :]).norm(dim=-1)
    return d.min(dim=-1)


def obj_physics_scores(ycb, pose6d, root, names, is_right, force_point, force_global, dtype=torch.float32):
    """aggregation.py:958-987"""
    p = pose6d.clone().to(dtype)
    p[..., 6:] = p[..., 6:] + root.unsqueeze(1)
    verts = flip_x(object_points(ycb, p, names, 'verts_sampled'), is_right)                # (bs,n,2048,3)
    com = flip_x(object_points(ycb, p, names, 'CoM'), is_right)                            # (bs,n,1,3)
    fn = force_global.norm(dim=-1)
    fw = fn / fn.sum(dim=-1, keepdim=True)
    dmin, amin = nearest(force_point[:, None], verts)                                      # (bs,n,32)
    score = (dmin * fw[:, None]).sum(-1)
    fg = force_global / fn[:, :, None]
    nn_v = torch.gather(verts, 2, amin[..., None].expand(*amin.shape, 3))
    r = force_point[:, None] - nn_v - com
    L = torch.cross(fg[:, None].expand_as(r), r, dim=-1).sum(-2).norm(dim=-1)
    return -(score * L)


def obj_physics_topk(ycb, pose6d, root, names, is_right, force_point, force_global, k, dtype=torch.float32):
    score = obj_physics_scores(ycb, pose6d, root, names, is_right, force_point, force_global, dtype=dtype)
    val, idx = topk_stable(score, k, dim=1)
    return idx, torch.ones_like(val) / k, score


def fuse_topk(pose6d, idx, weight):
    """aggregation.py:729-740 (dtype follows pose6d: f64 for the sampler's object poses, quirk Q5)."""
    bs = pose6d.shape[0]
    sel = torch.gather(pose6d, 1, idx[:, :, None].expand(bs, idx.shape[1], 9))
    trans = (sel[:, :, 6:] * weight[:, :, None]).sum(dim=1)
    return torch.cat([average_rot6d(sel[..., :6], weights=weight), trans], dim=-1)


# ------------------------------------------------------------------ hand physics (aggregation.py:537-626)
def hand_physics_scores(mano, anchor, anchor_skeleton, pose58, root_flip, force_local, obj_vert):
    """aggregation.py:553-592: per-finger pseudo-force score of every candidate -> (bs,5,n)"""
    bs, n = pose58.shape[:2]
    p = pose58.reshape(-1, 58)
    vert, _ = get_hand_verts(mano, p[:, :48], p[:, 48:])
    vert = vert.reshape(bs, n, 778, 3) + root_flip[:, None, None]
    fl = force_local[:, None].expand(bs, n, 32, 3)
    fp, fg = local_to_global(anchor, anchor_skeleton, fl, vert)                            # (bs,n,32,3)
    fn = fg.norm(dim=-1)
    fw = fn / fn.sum(dim=-1, keepdim=True)
    dmin, _ = nearest(fp, obj_vert[:, None])
    I = (fg / fn[..., None]).sum(-2).norm(dim=-1)
    score = -(fw * dmin * I[:, :, None])
    return torch.stack([score[:, :, FINGER_FORCE_LEVEL[f]].sum(dim=-1) for f in range(5)], 1)


def hand_physics(mano, anchor, anchor_skeleton, pose58, root_flip, force_local, obj_vert, k):
    bs, n = pose58.shape[:2]
    score = hand_physics_scores(mano, anchor, anchor_skeleton, pose58, root_flip, force_local, obj_vert)
    fuse = pose58[:, 0].clone()
    topks, scores = [], []
    for f in range(5):
        fs = score[:, f]
        _, idx = topk_stable(fs, k, dim=1)
        fidx = MANO_PARAMS_LEVEL[2][3 * f:3 * f + 3] + MANO_PARAMS_LEVEL[3][3 * f:3 * f + 3]
        sel = torch.gather(pose58[:, :, fidx], 1, idx[:, :, None].expand(bs, k, 6)).reshape(bs, k, 2, 3)
        q = average_quaternion(R.axis_angle_to_quaternion(sel).permute(0, 2, 1, 3))
        fuse[:, fidx] = R.quaternion_to_axis_angle(q).reshape(bs, 6)
        topks.append(idx)
        scores.append(fs)
    v, j = get_hand_verts(mano, fuse[:, :48], fuse[:, 48:])
    return dict(agg_pose=fuse, agg_vert=v, agg_joint=j, topk=torch.stack(topks, 1), score=torch.stack(scores, 1), cand=pose58)


# ------------------------------------------------------------------ HOI_Aggregator.__call__ (aggregation.py:1167-1353)
def hoi_aggregate(assets, anchor_skeleton, *, cam_intrinsic, root_joint_flip, root_joint, is_right, force_local,
                  is_grasped, hand_pose_diff, hand_pose_regression, hand_shape, hand_heatmap, hand_bbox, hand_topk,
                  obj_pose6d, obj_heatmap, obj_bbox, obj_topk, obj_name, phy_topk=5, dtype=torch.float32):
    """dtype: the arithmetic of the object branch's scores (float32 = the reference, which casts the sampler's fp64 object poses with
    .float(), aggregation.py:753; float64: the end-to-end judge of oracle/judge_fp64.py, every other input given in double too)."""
    mano, ycb, anchor = assets['mano'], assets['ycb'], assets['anchor']
    bs = root_joint.shape[0]
    S = hand_pose_diff.shape[0] // bs
    betas = hand_shape.reshape(bs, S, 10)[:, 0]
    h = hand_cascade(mano, hand_pose_diff.reshape(bs, S, 48), hand_pose_regression, betas, root_joint_flip,
                     cam_intrinsic, hand_heatmap, hand_bbox, hand_topk)
    agg_mano = torch.cat([h['fused_pose'], betas], -1)
    fpnt, fglob = local_to_global(anchor, anchor_skeleton, force_local, h['agg_vert'] + root_joint_flip[:, None])

    common = dict(root=root_joint, names=obj_name, is_right=is_right, dtype=dtype)
    t_idx, t_w, t_hv = obj_heat_topk(ycb, obj_pose6d, K=cam_intrinsic, heatmap=obj_heatmap, bbox=obj_bbox, k=obj_topk, **common)
    transl = fuse_topk(obj_pose6d, t_idx, t_w)[:, 6:]
    upd = obj_pose6d.clone()
    upd[..., 6:] = transl[:, None]
    r_idx, _, r_hv = obj_heat_topk(ycb, upd, K=cam_intrinsic, heatmap=obj_heatmap, bbox=obj_bbox, k=obj_topk, **common)
    g = lambda idx: torch.gather(obj_pose6d, 1, idx[:, :, None].expand(bs, obj_topk, 9))
    ct = g(t_idx)[:, :, None, 6:].expand(bs, obj_topk, obj_topk, 3)
    cr = g(r_idx)[:, None, :, :6].expand(bs, obj_topk, obj_topk, 6)
    cand = torch.cat([cr, ct], -1).reshape(bs, -1, 9)
    p_idx, p_w, p_score = obj_physics_topk(ycb, cand, force_point=fpnt, force_global=fglob, k=phy_topk, **common)
    m_idx, m_w, m_hv = obj_heat_topk(ycb, cand, K=cam_intrinsic, heatmap=obj_heatmap, bbox=obj_bbox, k=phy_topk, **common)
    new_idx = torch.where(is_grasped[:, None], p_idx, m_idx)
    new_w = torch.where(is_grasped[:, None], p_w, m_w)
    obj_fused = fuse_topk(cand, new_idx, new_w)
    p = obj_fused.clone().to(dtype)
    p[..., 6:] = p[..., 6:] + root_joint
    obj_vert = flip_x(object_points(ycb, p, obj_name, 'verts_sampled'), is_right)

    lvl3 = agg_mano[:, MANO_PARAMS_LEVEL[2]].reshape(bs, 1, 5, 3)
    lvl4 = torch.cat([h['topk_pose_l3'][:, :hand_topk], agg_mano[:, MANO_PARAMS_LEVEL[3]].reshape(bs, 1, 5, 3)], 1)
    n = hand_topk + 1
    cpose = agg_mano[:, None, :48].repeat(1, n, 1)
    cpose[:, :, MANO_PARAMS_LEVEL[2]] = lvl3.expand(bs, n, 5, 3).reshape(bs, n, 15)
    cpose[:, :, MANO_PARAMS_LEVEL[3]] = lvl4.reshape(bs, n, 15)
    cpose = torch.cat([cpose, agg_mano[:, None, 48:].expand(bs, n, 10)], -1)
    hp = hand_physics(mano, anchor, anchor_skeleton, cpose, root_joint_flip, force_local, obj_vert, phy_topk)
    return dict(obj_agg_6d=obj_fused, pose6d_candidate=cand, agg_obj_vert=obj_vert,
                hand_agg_mano=hp['agg_pose'], hand_agg_vert=hp['agg_vert'], hand_agg_joint=hp['agg_joint'],
                dbg=dict(hand=h, transl_topk=t_idx, rot_topk=r_idx, phys_topk=p_idx, heat_topk=m_idx,
                         phys_score=p_score, transl_score=t_hv, rot_score=r_hv, heat_score=m_hv, hand_phys=hp, cascade_mano=agg_mano, force_point=fpnt, force_global=fglob,
                         transl=transl, obj_vert=obj_vert, pose6d_candidate=cand))
