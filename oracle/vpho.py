"""``vpho_net.forward(data, mode='predict')`` -- torch-CPU restatement of the reference's lib/model/VPHO.py:90-304
(+ postprocess_diffusion_hand :306-331, align_hm_to_bbox_rectangle :333-346, flip helpers :349-364).
TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import torch
import torch.nn.functional as F

from . import nets as N
from . import rotations as R
from .mano import get_hand_verts
from .roi_align import roi_align_fast
from .aggregation import hoi_aggregate

MANOPTH_TO_MANOLAYER = [0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 4, 8, 12, 16, 20]   # argsort of hand_fn.py:8


def align_hm_to_bbox_rectangle(hm, bbox, bbox_rect, size=64):
    """VPHO.py:333-346 (meshgrid 'ij' stacked as (xx,yy): transposing resample, quirk Q2)."""
    xx, yy = torch.meshgrid(torch.arange(size), torch.arange(size), indexing='ij')
    xx = xx / (size - 1) * 2 - 1
    yy = yy / (size - 1) * 2 - 1
    rel = (bbox_rect[:, 2:] - bbox_rect[:, :2]) / (bbox[:, 2:] - bbox[:, :2])
    grid = torch.stack((xx * rel[:, 0][:, None, None], yy * rel[:, 1][:, None, None]), dim=-1)
    return F.grid_sample(hm, grid, mode='bilinear', align_corners=False)


def flip_w(t, is_flip):
    return torch.where(is_flip.reshape(-1, *[1] * (t.dim() - 1)), t.flip(-1), t)


def flip_point_x(p, is_flip):
    out = p.clone()
    out[is_flip, ..., 0] *= -1
    return out


def joints_ho3d(vert, joint):
    """hand_fn.py:454-461"""
    j = joint[..., MANOPTH_TO_MANOLAYER, :].clone()
    j[..., [16, 17, 18, 19, 20], :] = vert[..., [728, 353, 442, 576, 694], :]
    return j


def features(sd, assets, data, roi_size=32, heatmap_size=64):
    """VPHO.py:112-172 -> dict of intermediate tensors."""
    bs = data['rgb'].shape[0]
    hf, of = N.fpn(sd, 'feature_extractor', data['rgb'])
    idx = torch.arange(bs).float()[:, None]
    roi = lambda feat, key: roi_align_fast(feat, torch.cat((idx, data[key].float()), 1), (roi_size, roi_size), 0.25)
    hf_hr, hf_hr_rect, of_or_rect = roi(hf, 'bbox_hand'), roi(hf, 'bbox_hand_rect'), roi(of, 'bbox_obj_rect')
    hm_hand = N.head_heatmap2(sd, 'head_hm_hand', hf_hr)
    hm_obj = N.head_heatmap2(sd, 'head_hm_obj', of_or_rect)
    hm_hand_rect = align_hm_to_bbox_rectangle(hm_hand, data['bbox_hand'], data['bbox_hand_rect'], heatmap_size)
    hm_obj_rect = align_hm_to_bbox_rectangle(hm_obj, data['bbox_obj'], data['bbox_obj_rect'], heatmap_size)
    is_left = ~data['is_right']
    of_or_rect = flip_w(of_or_rect, is_left)
    hm_obj_rect_ori = flip_w(hm_obj_rect, is_left)
    hm_hand_rs = F.interpolate(hm_hand_rect, size=(roi_size, roi_size), mode='bilinear', align_corners=False)
    hm_obj_rs = F.interpolate(hm_obj_rect_ori, size=(roi_size, roi_size), mode='bilinear', align_corners=False)
    enc_h, st_h = N.encoder(sd, 'encoder_hand', torch.cat((hf_hr_rect, hm_hand_rs), 1))
    enc_o, st_o = N.encoder(sd, 'encoder_obj', torch.cat((of_or_rect, hm_obj_rs), 1))
    pose, shape = N.head_mano(sd, 'head_mano', enc_h)
    vert, joint = get_hand_verts(assets['mano'], pose, shape)
    m = data['is_ho3d']
    if m.any():
        joint = joint.clone()
        joint[m] = joints_ho3d(vert[m], joint[m])
    grav = flip_point_x(data['gravity'], is_left)
    ph, _, _ = N.cross_module(sd, 'cross_hand', st_h[1], st_o[1], grav)
    _, po, _ = N.cross_module(sd, 'cross_obj', st_h[1], st_o[1], grav)
    force_local = N.head_physics(sd, 'head_physics', ph, po)
    return dict(hand_feat=hf, obj_feat=of, hf_hr=hf_hr, hf_hr_rect=hf_hr_rect, of_or_rect=of_or_rect,
                hand_heatmap=hm_hand, obj_heatmap=hm_obj, encoding_hand=enc_h, encoding_obj=enc_o,
                stage_hand=st_h[1], stage_obj=st_o[1], mano_pose=pose, mano_shape=shape, reg_hand_vert=vert,
                reg_hand_joint=joint, tok_hand=ph, tok_obj=po, force_local=force_local)


def postprocess_diffusion_hand(x6d, betas):
    """VPHO.py:306-331 'mano_pose' branch: (..., 96) rot6d -> (..., 48) axis-angle, append betas."""
    aa = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(x6d.reshape(*x6d.shape[:-1], 16, 6)))
    aa = aa.reshape(*x6d.shape[:-1], 48)
    b = betas.reshape(betas.shape[0], *[1] * (aa.dim() - 2), 10).expand(*aa.shape[:-1], 10)
    return torch.cat((aa, b), -1)


def predict(sd, assets, anchor_skeleton, data, *, sample_num, sample_T0, sampling_steps, topk_hand, topk_obj,
            noise_hand=None, noise_obj=None):
    """Returns (output dict as VPHO.py:229-304, info).  ``noise_*``: standard normals (bs*S, D) drawn like
    sde.py:26-28 (CPU default generator, hand first) when not supplied."""
    bs = data['rgb'].shape[0]
    S = sample_num
    f = features(sd, assets, data)
    out = dict(reg_hand_vert=f['reg_hand_vert'], reg_hand_joint=f['reg_hand_joint'], hand_heatmap=f['hand_heatmap'],
               obj_heatmap=f['obj_heatmap'], force_local=f['force_local'])
    sig = N.ve_prior_sigma(sample_T0)
    if noise_hand is None:
        noise_hand = torch.randn(bs * S, 96)
    feat_h = f['encoding_hand'][:, None].repeat(1, S, 1).reshape(-1, 1024)
    xs_h, x_h, info_h = N.ode_sample(sd, 'denoiser_hand', feat_h, noise_hand * sig, sample_T0, sampling_steps)
    xs_h, x_h = xs_h.float(), x_h.float()
    inproc = postprocess_diffusion_hand(xs_h.reshape(bs, S, sampling_steps, 96), f['mano_shape'])
    final = postprocess_diffusion_hand(x_h.reshape(bs, S, 96), f['mano_shape'])
    out['diff_inprocess_hand_mano'] = inproc
    out['diff_final_hand_mano'] = final
    ip = inproc.reshape(-1, sampling_steps, 58)[0, ::10]
    v, j = get_hand_verts(assets['mano'], ip[:, :48], ip[:, 48:])
    out['diff_inprocess_hand_vert'], out['diff_inprocess_hand_joint'] = v, j
    fl = final.reshape(-1, 58)
    v, j = get_hand_verts(assets['mano'], fl[:, :48], fl[:, 48:])
    out['diff_final_hand_vert'] = v.reshape(bs, S, 778, 3)
    out['diff_final_hand_joint'] = j.reshape(bs, S, 21, 3)
    if noise_obj is None:
        noise_obj = torch.randn(bs * S, 9)
    feat_o = f['encoding_obj'][:, None].repeat(1, S, 1).reshape(-1, 1024)
    xs_o, x_o, info_o = N.ode_sample(sd, 'denoiser_obj', feat_o, noise_obj * sig, sample_T0, sampling_steps)
    out['diff_inprocess_obj_6d'] = xs_o.reshape(bs, S, -1, 9)
    out['diff_final_obj_6d'] = x_o.reshape(bs, S, 9)
    agg = hoi_aggregate(assets, anchor_skeleton,
                        cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                        root_joint=data['root_joint'], is_right=data['is_right'], force_local=f['force_local'],
                        is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(),
                        hand_pose_regression=f['mano_pose'], hand_shape=fl[:, 48:], hand_heatmap=f['hand_heatmap'],
                        hand_bbox=data['bbox_hand'], hand_topk=topk_hand, obj_pose6d=out['diff_final_obj_6d'],
                        obj_heatmap=f['obj_heatmap'], obj_bbox=data['bbox_obj_rect'], obj_topk=topk_obj,
                        obj_name=data['obj_name'])
    out['agg_obj_6d'] = agg['obj_agg_6d']
    out['agg_hand_mano'] = agg['hand_agg_mano']
    out['agg_hand_vert'] = agg['hand_agg_vert']
    out['agg_hand_joint'] = agg['hand_agg_joint']
    return out, dict(features=f, hand_ode=info_h, obj_ode=info_o, agg=agg['dbg'], hand_x6d=x_h)
