"""The HIP kernels of the three third-party leaves against the SECOND formulations of tests/_leaf_independent.py (scipy Rotation,
grid_sample RoIAlign, direct float64 MANO LBS) -- not against the oracle: tests/test_leaf_crosschecks.py holds the oracle to the
same formulations on the CPU, so oracle and kernel are each pinned independently.  Reference call sites:
VPHO.py:125-128 (roi_align), VPHO.py:316-323 (rot6d -> matrix -> axis-angle), head_mano.py:78-87 (ManoLayer)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from tests import _leaf_independent as L

pytestmark = pytest.mark.gpu


def test_rot6d_to_axis_angle_kernel_agrees_with_scipy():
    """rot6d_to_aa_kernel = Gram-Schmidt -> matrix -> quaternion (best-conditioned candidate) -> axis-angle, 16 rotations per row
    like postprocess_diffusion_hand; 1e5 random rotations + 2 000 near pi + 2 000 below the 1e-6 Taylor branch, and scaled /
    sheared 6-vectors (the sampler's raw output is not orthonormal)"""
    from vpho_amd import ops
    rv = L.random_rotations(100_000, seed=21, near_pi=2000, tiny=2000)
    rv = rv[:len(rv) // 16 * 16]
    m = Rotation.from_rotvec(rv).as_matrix()
    rng = np.random.default_rng(1)
    d6 = m[:, :2, :].reshape(-1, 6).copy()
    k = len(d6) // 2                                                        # second half: a1 scaled, a2 = scaled row 1 + a multiple of row 0
    s1, s2, sh = rng.uniform(0.2, 4, (k, 1)), rng.uniform(0.5, 4, (k, 1)), rng.uniform(-1, 1, (k, 1))
    d6[k:, 3:] = d6[k:, 3:] * s2 + sh * d6[k:, :3]
    d6[k:, :3] *= s1
    x = torch.as_tensor(d6, dtype=torch.float32).view(-1, 96).cuda()
    aa = ops.rot6d_to_axis_angle(x, 16).cpu().double().numpy().reshape(-1, 3)
    assert np.isfinite(aa).all()
    want_m = L.rot6d_to_matrix_by_cross_products(x.cpu().double().numpy().reshape(-1, 6))      # of the float32 inputs the kernel saw
    err = L.rotation_angle_between(Rotation.from_rotvec(aa).as_matrix(), want_m)
    assert float(err.max()) < 3e-6, float(err.max())
    ang = np.linalg.norm(rv, axis=-1)
    mid = (ang < 2.5) & (np.arange(len(rv)) < k)
    assert float(np.abs(aa - rv)[mid].max()) < 5e-6                        # the rotation vector itself away from pi (conditioning 1 / sin)
    assert float(np.linalg.norm(aa, axis=-1).max()) <= np.pi + 1e-5        # standardised quaternion: angle in [0, pi]


@pytest.mark.parametrize('C', [4, 8])
def test_roi_align_kernel_agrees_with_the_grid_sample_formulation(C):
    """roi_align_nhwc_kernel<4> on 1 000 random boxes (one per image, like VPHO.py:117-128) incl. exactly-integer bin sizes,
    boxes that leave the map, sub-pixel boxes; also through the W-flip (VPHO.py:138)"""
    from vpho_amd import ops
    n = 1000
    g = torch.Generator().manual_seed(7 + C)
    maps = torch.randn(4, C, 64, 64, generator=g)
    img = torch.randint(0, 4, (n,), generator=g)
    boxes = L.random_boxes(n, seed=8)
    want = L.roi_align_by_grid_sample(maps, torch.cat([img[:, None].float(), boxes], 1), 32, 0.25)        # (n, C, 32, 32)
    feat = maps.permute(0, 2, 3, 1)[img].contiguous().cuda()                                              # (n, 64, 64, C) NHWC
    got = ops.roi_align_nhwc(feat, boxes.cuda(), 32, 0.25).cpu().permute(0, 3, 1, 2).double()
    assert float((got - want).abs().max()) < 2e-5
    flip = (torch.arange(n) % 2).to(torch.uint8)
    gotf = ops.roi_align_nhwc(feat, boxes.cuda(), 32, 0.25, flip_w=flip.cuda()).cpu().permute(0, 3, 1, 2).double()
    wantf = torch.where(flip.bool()[:, None, None, None], want.flip(-1), want)
    assert float((gotf - wantf).abs().max()) < 2e-5


@pytest.mark.parametrize('n_img,per_img', [(3, 7), (40, 31), (64, 100)])
def test_mano_fk_kernels_agree_with_a_direct_float64_lbs(assets, n_img, per_img):
    """(3, 7): one hand per workgroup; (40, 31): the physics candidates' shape, 16-hand blocks of the packed-FMA kernel;
    (64, 100) = 6 400 hands with vertices: the matrix-core kernel.  Rodrigues inside the kernels is manopth's quaternion route;
    the reference here is scipy's rotation matrices in a parent-table LBS"""
    from vpho_amd import ops
    rng = np.random.default_rng(n_img * 100 + per_img)
    n = n_img * per_img
    pose = rng.normal(size=(n, 48)) * 0.5
    pose[0] = 0
    pose[1, 3:] = 0
    pose[2] *= 3.0
    betas = rng.normal(size=(n_img, 10)) * 0.8
    M = ops.Mano(assets['mano'], 'cuda')
    p32, b32 = torch.as_tensor(pose, dtype=torch.float32), torch.as_tensor(betas, dtype=torch.float32)
    ctx = M.shape(b32.cuda())
    verts, joints = M.fk(p32.cuda(), ctx, per_img, True)
    _, joints_only = M.fk(p32.cuda(), ctx, per_img, False)
    pick = np.unique(np.linspace(0, n - 1, 96).astype(int))
    wv, wj = L.mano_lbs_fp64(assets['mano'], p32.double().numpy()[pick], b32.double().numpy()[pick // per_img])
    assert float(np.abs(verts.cpu().double().numpy()[pick] - wv).max()) < 2e-6           # metres
    assert float(np.abs(joints.cpu().double().numpy()[pick] - wj).max()) < 2e-6
    assert float(np.abs(joints_only.cpu().double().numpy()[pick] - wj).max()) < 2e-6
