"""Pseudo-force label optimisation -- torch-CPU restatement of the inner loop of the reference's
ForceOptimizer.optimize_batch (lib/engine/force_optimization.py:110-207) with HeadForce.get_local_force
(lib/model/physics.py:277-288) and from_local_to_global (:362-371).  Pinned by tests/golden/golden_force_optim.npz: the reference's
own optimize_batch run unchanged for its 3000 iterations (tests/golden/make_golden_force_optim.py stubs only the two dataset
modules the file imports but the tree lacks, force_optimization.py:12-13, and the accelerate object).
TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import torch
import torch.nn.functional as F

from .aggregation import vert2anchor


def anchor_cone():
    a = torch.arange(0, 2 * torch.pi, 2 * torch.pi / 8)[:8]
    return torch.stack([torch.cos(a), torch.sin(a), torch.ones_like(a)], dim=-1) / 8        # physics.py:183-188


def get_local_force(scale, weight, friction=0.8):
    """physics.py:277-288 (single softmax here, unlike HeadPhysics)."""
    scale = torch.abs(scale)
    weight = torch.softmax(weight, dim=-1)
    anchor = anchor_cone()
    anchor[:, :2] *= friction
    d = torch.einsum('...ij,jk->...ik', weight, anchor)
    d = d / (d.norm(dim=-1, keepdim=True) + 1e-8)
    return d * scale[..., None]


def optimize(anchor, anchor_skeleton, vert3d, gravity, com, force_contact, is_grasped, iters=3000, phase1=300, lr=1e-3):
    """vert3d (B,778,3), gravity/com (B,1,3) already in the flipped frame, force_contact (B,32).
    Returns force_local, force_global (B,32,3) (zeroed where not grasped), last losses, final scale/weight."""
    B = vert3d.shape[0]
    scale_p = torch.nn.Parameter(torch.ones(B, 32) * 0.05)
    weight_p = torch.nn.Parameter(torch.zeros(B, 32, 8))
    opt1 = torch.optim.AdamW([weight_p], betas=(0.9, 0.999), eps=1e-8, lr=lr)
    opt2 = torch.optim.AdamW([scale_p, weight_p], betas=(0.9, 0.999), eps=1e-8, lr=lr)
    mask = force_contact > 0.1
    pts, frame = vert2anchor(anchor, anchor_skeleton, vert3d)
    losses = None
    for i in range(iters):
        scale = scale_p.clone() * mask
        weight = weight_p.clone()
        fl = get_local_force(scale, weight)
        fg = torch.einsum('...bi,...bji->...bj', fl, frame)
        res = (fg.sum(1, keepdim=True) + gravity).squeeze(1)
        force_loss = torch.norm(res, dim=-1).mean()
        sw = force_loss.detach()
        cos = torch.einsum('...i,...i->...', fg.sum(1, keepdim=True), -1 * gravity)
        gravity_loss = F.mse_loss(cos, torch.ones_like(cos))
        moment = torch.cross(pts - com, fg, dim=-1).sum(1)
        moment_loss = torch.norm(moment, dim=-1).mean() * 30 / (100 * sw ** 2 + 1e-8)
        scale_norm = scale / (scale.norm(dim=-1, keepdim=True).detach() + 1e-8).detach()
        fc_norm = force_contact / (force_contact.norm(dim=-1, keepdim=True).detach() + 1e-8)
        dist = torch.log(torch.abs(fc_norm / (scale_norm + 1e-8)) + 1e-8) * mask
        dist_loss = (dist ** 2).mean() * 0.1 / (1000 * sw ** 2 + 1e-8)
        if i < phase1:
            loss, opt = gravity_loss, opt1
        else:
            loss, opt = force_loss + moment_loss + dist_loss, opt2
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses = (float(force_loss), float(gravity_loss), float(moment_loss), float(dist_loss))
    fl, fg = fl.detach().clone(), fg.detach().clone()
    fl[~is_grasped] = 0
    fg[~is_grasped] = 0
    return dict(force_local=fl, force_global=fg, losses=losses, scale=scale_p.detach().clone(), weight=weight_p.detach().clone(),
                force_point=pts)
