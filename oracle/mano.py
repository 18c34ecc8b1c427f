"""MANO linear-blend-skinning forward kinematics -- restatement of
manopth.manolayer.ManoLayer.forward (hassony2/manopth, un-pinned; configured by the
reference at head_mano.py:48-55 as ncomps=45, center_idx=0, flat_hand_mean=True,
side='right', use_pca=False) plus rodrigues_layer.batch_rodrigues / quat2mat and
tensutils.th_posemap_axisang / subtract_flat_id.  Output in millimetres like manopth;
``get_hand_verts`` applies the reference's /1000 (head_mano.py:78-87).

``assets`` keys: v_template (778,3), shapedirs (778,3,10), posedirs (778,3,135),
J_regressor (16,778), weights (778,16).
"""
import torch

LEV1 = [1, 4, 7, 10, 13]
LEV2 = [2, 5, 8, 11, 14]
LEV3 = [3, 6, 9, 12, 15]
REORDER = [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15]
TIPS_RIGHT = [745, 317, 444, 556, 673]
JOINT_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def quat2mat(quat):
    nq = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack(
        [w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
         2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
         2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(B, 3, 3)


def batch_rodrigues(axisang):
    axisang_norm = torch.norm(axisang + 1e-8, p=2, dim=1)
    angle = torch.unsqueeze(axisang_norm, -1)
    axisang_normalized = torch.div(axisang, angle)
    angle = angle * 0.5
    quat = torch.cat([torch.cos(angle), torch.sin(angle) * axisang_normalized], dim=1)
    return quat2mat(quat).view(-1, 9)


def _with_zeros(t):  # (B,3,4) -> (B,4,4)
    pad = t.new_zeros(t.shape[0], 1, 4)
    pad[:, 0, 3] = 1.0
    return torch.cat([t, pad], 1)


def mano_forward(assets, pose, betas):
    """pose (B,48) axis-angle, betas (B,10) -> verts (B,778,3) mm, joints (B,21,3) mm."""
    t = lambda k: torch.as_tensor(assets[k], dtype=pose.dtype)
    v_template = t('v_template')[None]
    shapedirs, posedirs = t('shapedirs'), t('posedirs')
    J_reg, weights = t('J_regressor'), t('weights')
    B = pose.shape[0]
    rot_mats = batch_rodrigues(pose.contiguous().view(-1, 3)).view(B, 16 * 9)
    eye = torch.eye(3, dtype=pose.dtype).view(1, 9).repeat(1, 16)
    pose_map = (rot_mats - eye)[:, 9:]
    root_rot = rot_mats[:, :9].view(B, 3, 3)
    rot_map = rot_mats[:, 9:]

    v_shaped = torch.matmul(shapedirs, betas.transpose(1, 0)).permute(2, 0, 1) + v_template
    th_j = torch.matmul(J_reg, v_shaped)
    v_posed = v_shaped + torch.matmul(posedirs, pose_map.transpose(0, 1)).permute(2, 0, 1)

    root_j = th_j[:, 0, :].contiguous().view(B, 3, 1)
    root_trans = _with_zeros(torch.cat([root_rot, root_j], 2))
    all_rots = rot_map.view(B, 15, 3, 3)
    l1r, l2r, l3r = (all_rots[:, [i - 1 for i in L]] for L in (LEV1, LEV2, LEV3))
    l1j, l2j, l3j = th_j[:, LEV1], th_j[:, LEV2], th_j[:, LEV3]

    all_tf = [root_trans.unsqueeze(1)]
    l1rel = _with_zeros(torch.cat([l1r, (l1j - root_j.transpose(1, 2)).unsqueeze(3)], 3).view(-1, 3, 4))
    root_flt = root_trans.unsqueeze(1).repeat(1, 5, 1, 1).view(B * 5, 4, 4)
    l1 = torch.matmul(root_flt, l1rel)
    all_tf.append(l1.view(B, 5, 4, 4))
    l2rel = _with_zeros(torch.cat([l2r, (l2j - l1j).unsqueeze(3)], 3).view(-1, 3, 4))
    l2 = torch.matmul(l1, l2rel)
    all_tf.append(l2.view(B, 5, 4, 4))
    l3rel = _with_zeros(torch.cat([l3r, (l3j - l2j).unsqueeze(3)], 3).view(-1, 3, 4))
    l3 = torch.matmul(l2, l3rel)
    all_tf.append(l3.view(B, 5, 4, 4))
    results = torch.cat(all_tf, 1)[:, REORDER]

    joint_js = torch.cat([th_j, th_j.new_zeros(B, 16, 1)], 2)
    tmp2 = torch.matmul(results, joint_js.unsqueeze(3))
    results2 = (results - torch.cat([tmp2.new_zeros(B, 16, 4, 3), tmp2], 3)).permute(0, 2, 3, 1)
    T = torch.matmul(results2, weights.transpose(0, 1))
    rest_h = torch.cat([v_posed.transpose(2, 1), torch.ones((B, 1, 778), dtype=T.dtype)], 1)
    verts = (T * rest_h.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]
    jtr = results[:, :, :3, 3]
    jtr = torch.cat([jtr, verts[:, TIPS_RIGHT]], 1)[:, JOINT_ORDER]
    center = jtr[:, 0].unsqueeze(1)
    jtr = jtr - center
    verts = verts - center
    return verts * 1000, jtr * 1000


def get_hand_verts(assets, pose, shape):
    """head_mano.py:78-87 -- metres."""
    v, j = mano_forward(assets, pose, shape)
    return v / 1000, j / 1000
