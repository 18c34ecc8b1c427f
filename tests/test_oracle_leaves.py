"""Known-answer tests of the oracle's third-party leaves: the arithmetic the reference takes from packages that are
neither in its tree nor installed here -- pytorch3d rotation conversions, manopth's MANO layer, torchvision's RoIAlign -- and
scipy's RK45, which IS installed and is compared step by step.  torchvision / pytorch3d / manopth cannot be imported in the
build container (no wheels), so these leaves are pinned by closed-form identities instead of fixtures of the packages.
CPU only.
"""
import math

import numpy as np
import pytest
import torch

from oracle import rotations as R
from oracle import mano as M
from oracle import roi_align as RA
from oracle import rk45 as RK
from vpho_amd.assets import synthetic_assets

torch.manual_seed(0)


def _rand_rot(n, dtype=torch.float64):
    q = torch.randn(n, 4, dtype=dtype)
    q = q / q.norm(dim=-1, keepdim=True)
    return R.quaternion_to_matrix(q)


# ------------------------------------------------------------------------------------------------- rotation conversions
def test_rot6d_matrix_round_trip_and_orthonormality():
    d6 = torch.randn(500, 6, dtype=torch.float64)
    m = R.rotation_6d_to_matrix(d6)
    eye = torch.eye(3, dtype=torch.float64)
    assert float((m @ m.transpose(-1, -2) - eye).abs().max()) < 1e-12
    assert float((torch.linalg.det(m) - 1).abs().max()) < 1e-12
    # rows 0/1 of the matrix ARE the 6-d representation (pytorch3d convention: first two ROWS), and it is a fixed point
    assert torch.equal(R.matrix_to_rotation_6d(m), m[:, :2, :].reshape(-1, 6))
    assert float((R.rotation_6d_to_matrix(R.matrix_to_rotation_6d(m)) - m).abs().max()) < 1e-12
    # Gram-Schmidt: first row is the normalised first triple, second row lies in span(a1, a2) with a positive a2 component
    a1, a2 = d6[:, :3], d6[:, 3:]
    assert float((m[:, 0] - a1 / a1.norm(dim=-1, keepdim=True)).abs().max()) < 1e-12
    assert bool(((m[:, 1] * a2).sum(-1) > 0).all())
    assert float((m[:, 2] * a1).sum(-1).abs().max()) < 1e-9 and float((m[:, 2] * a2).sum(-1).abs().max()) < 1e-9


def test_known_rotations():
    # 90 degrees about z: x -> y
    aa = torch.tensor([[0.0, 0.0, math.pi / 2]], dtype=torch.float64)
    m = R.axis_angle_to_matrix(aa)[0]
    want = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    assert float((m - want).abs().max()) < 1e-12
    q = R.axis_angle_to_quaternion(aa)[0]
    assert float((q - torch.tensor([math.cos(math.pi / 4), 0, 0, math.sin(math.pi / 4)], dtype=torch.float64)).abs().max()) < 1e-12
    # identity
    z = torch.zeros(1, 3, dtype=torch.float64)
    assert torch.equal(R.axis_angle_to_quaternion(z), torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64))
    assert float((R.axis_angle_to_matrix(z)[0] - torch.eye(3, dtype=torch.float64)).abs().max()) == 0.0
    # quaternion (real part first) of a 180-degree turn about x
    assert float((R.quaternion_to_matrix(torch.tensor([[0.0, 1.0, 0, 0]]))[0] - torch.diag(torch.tensor([1.0, -1.0, -1.0]))).abs().max()) == 0.0


def test_axis_angle_quaternion_round_trip_incl_small_angle_branch():
    # generic angles
    aa = torch.randn(300, 3, dtype=torch.float64)
    aa = aa / aa.norm(dim=-1, keepdim=True) * torch.rand(300, 1, dtype=torch.float64) * 3.0
    q = R.axis_angle_to_quaternion(aa)
    assert float((q.norm(dim=-1) - 1).abs().max()) < 1e-12
    assert float((R.quaternion_to_axis_angle(q) - aa).abs().max()) < 1e-10
    # the Taylor branch (|angle| < 1e-6): sin(x/2)/x ~ 1/2 - x^2/48 -- continuous with the exact branch at the switch
    for ang in (0.0, 1e-9, 9.9e-7, 1.01e-6, 1e-5):
        v = torch.tensor([[ang, 0.0, 0.0]], dtype=torch.float64)
        q = R.axis_angle_to_quaternion(v)[0]
        exact = math.sin(ang / 2) if ang > 0 else 0.0
        assert abs(float(q[1]) - exact) < 1e-18 + 1e-12 * abs(exact), ang
        back = R.quaternion_to_axis_angle(q[None])[0]
        assert abs(float(back[0]) - ang) < 1e-15 + 1e-9 * ang, ang
    # fp32 (the dtype of the path): same branch point, round trip to fp32 resolution
    aa32 = torch.tensor([[3e-7, -2e-7, 1e-7], [0.3, -0.2, 0.1], [0.0, 0.0, 0.0]])
    assert float((R.quaternion_to_axis_angle(R.axis_angle_to_quaternion(aa32)) - aa32).abs().max()) < 1e-7


def test_matrix_to_quaternion_takes_every_candidate_branch():
    # rotations by ~pi about x, y, z make the i, j, k candidate the best conditioned one; small rotations the r candidate
    cases = {0: torch.tensor([0.1, 0.05, -0.02]), 1: torch.tensor([3.1, 0.01, 0.02]), 2: torch.tensor([0.02, 3.1, 0.01]),
             3: torch.tensor([0.01, -0.02, 3.1])}
    for branch, aa in cases.items():
        aa = aa.double()[None]
        m = R.axis_angle_to_matrix(aa)
        t = torch.stack([1 + m[0, 0, 0] + m[0, 1, 1] + m[0, 2, 2], 1 + m[0, 0, 0] - m[0, 1, 1] - m[0, 2, 2],
                         1 - m[0, 0, 0] + m[0, 1, 1] - m[0, 2, 2], 1 - m[0, 0, 0] - m[0, 1, 1] + m[0, 2, 2]])
        assert int(t.argmax()) == branch
        q = R.matrix_to_quaternion(m)
        assert float(q[0, 0]) >= 0                       # pytorch3d >= 0.7.6: standardised sign (real part non-negative)
        assert float((R.quaternion_to_matrix(q) - m).abs().max()) < 1e-12
        assert float((R.matrix_to_axis_angle(m) - aa).abs().max()) < 1e-9
    # random rotations: matrix -> quaternion -> matrix is the identity, and q / -q map to the same matrix
    m = _rand_rot(1000)
    q = R.matrix_to_quaternion(m)
    assert float((R.quaternion_to_matrix(q) - m).abs().max()) < 1e-12
    assert float((R.quaternion_to_matrix(-q) - m).abs().max()) < 1e-12
    assert bool((q[:, 0] >= 0).all())


# ------------------------------------------------------------------------------------------------------------- MANO layer
@pytest.fixture(scope='module')
def mano():
    return synthetic_assets(0)['mano']


def test_mano_zero_pose_is_template_plus_shape_blend(mano):
    t = lambda k: torch.as_tensor(mano[k]).double()
    betas = torch.randn(3, 10, dtype=torch.float64) * 0.7
    v, j = M.mano_forward({k: np.asarray(a, np.float64) for k, a in mano.items()}, torch.zeros(3, 48, dtype=torch.float64), betas)
    v_shaped = t('v_template')[None] + torch.einsum('vck,bk->bvc', t('shapedirs'), betas)
    J = torch.einsum('jv,bvc->bjc', t('J_regressor'), v_shaped)
    # manopth: centred on joint 0 (center_idx=0), millimetres (the fp32 skinning weights of a vertex sum to 1 +- 1e-7: 5e-8 m)
    assert float((v / 1000 - (v_shaped - J[:, :1])).abs().max()) < 5e-8
    # joints: the 16 regressed joints in manopth's output order + 5 finger-tip vertices
    order16 = [M.JOINT_ORDER.index(i) for i in range(16)]
    assert float((j[:, order16] / 1000 - (J - J[:, :1])).abs().max()) < 1e-9
    tips = [M.JOINT_ORDER.index(16 + f) for f in range(5)]
    assert float((j[:, tips] - v[:, M.TIPS_RIGHT]).abs().max()) < 1e-9
    # zero betas: joints = J_regressor @ v_template
    v0, j0 = M.mano_forward({k: np.asarray(a, np.float64) for k, a in mano.items()}, torch.zeros(1, 48, dtype=torch.float64), torch.zeros(1, 10, dtype=torch.float64))
    J0 = t('J_regressor') @ t('v_template')
    assert float((j0[0, order16] / 1000 - (J0 - J0[:1])).abs().max()) < 1e-9


def test_mano_global_rotation_is_rigid_and_bones_keep_their_length(mano):
    a64 = {k: np.asarray(a, np.float64) for k, a in mano.items()}
    betas = torch.randn(4, 10, dtype=torch.float64) * 0.5
    pose = torch.randn(4, 48, dtype=torch.float64) * 0.4
    v, j = M.mano_forward(a64, pose, betas)
    # same articulation, wrist rotation removed: the whole hand differs by exactly that rotation about joint 0
    p0 = pose.clone()
    p0[:, :3] = 0
    v0, j0 = M.mano_forward(a64, p0, betas)
    Rg = R.axis_angle_to_matrix(pose[:, :3])
    assert float((torch.einsum('bij,bvj->bvi', Rg, v0) - v).abs().max()) < 1e-6      # manopth's +1e-8 inside the norm: ~1e-8 rad
    assert float((torch.einsum('bij,bvj->bvi', Rg, j0) - j).abs().max()) < 1e-6
    # kinematic chain: parent-child joint distances do not depend on the pose (tips are skinned vertices: excluded)
    _, jrest = M.mano_forward(a64, torch.zeros(4, 48, dtype=torch.float64), betas)
    for f in range(5):
        chain = [0, 1 + 4 * f, 2 + 4 * f, 3 + 4 * f]
        for a, b in zip(chain[:-1], chain[1:]):
            assert float(((j[:, a] - j[:, b]).norm(dim=-1) - (jrest[:, a] - jrest[:, b]).norm(dim=-1)).abs().max()) < 1e-6
    # get_hand_verts = /1000 (head_mano.py:86-87)
    vm, jm = M.get_hand_verts(a64, pose, betas)
    assert torch.equal(vm, v / 1000) and torch.equal(jm, j / 1000)


def test_batch_rodrigues_matches_the_quaternion_route():
    aa = torch.randn(200, 3, dtype=torch.float64)
    m = M.batch_rodrigues(aa).view(-1, 3, 3)
    assert float((m - R.axis_angle_to_matrix(aa)).abs().max()) < 1e-7                 # manopth adds 1e-8 before the norm
    assert float((M.batch_rodrigues(torch.zeros(1, 3, dtype=torch.float64)).view(3, 3) - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-12


# --------------------------------------------------------------------------------------------------------------- RoIAlign
def test_roi_align_constant_and_ramp_maps():
    H = W = 16
    const = torch.full((1, 3, H, W), 2.5)
    rois = torch.tensor([[0, 8.0, 12.0, 40.0, 52.0], [0, 0.0, 0.0, 63.0, 63.0], [0, 20.0, 20.0, 21.0, 21.5]])
    for fn in (RA.roi_align, RA.roi_align_fast):
        out = fn(const, rois, (4, 4), 0.25)
        assert float((out - 2.5).abs().max()) < 1e-6, fn.__name__
    # linear ramp f(y,x) = 2x + 3y + 1: bilinear interpolation reproduces it, so every bin = f(bin centre) (legacy pixel model:
    # sample coordinate c reads the ramp at c, no half-pixel shift) as long as all samples stay inside [0, H-1]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    ramp = (2 * xx + 3 * yy + 1)[None, None]
    roi = torch.tensor([[0, 8.0, 12.0, 40.0, 52.0]])                                   # x 2..10, y 3..13 at scale 1/4
    x1, y1, x2, y2 = 2.0, 3.0, 10.0, 13.0
    ph = pw = 4
    cy = y1 + (torch.arange(ph) + 0.5) * (y2 - y1) / ph
    cx = x1 + (torch.arange(pw) + 0.5) * (x2 - x1) / pw
    want = 2 * cx[None, :] + 3 * cy[:, None] + 1
    for fn in (RA.roi_align, RA.roi_align_fast):
        out = fn(ramp, roi, (ph, pw), 0.25)[0, 0]
        assert float((out - want).abs().max()) < 1e-4, fn.__name__
    # adaptive sampling grid: ceil(roi / bins) samples per bin and axis; an RoI smaller than one pixel is widened to 1 (legacy)
    tiny = RA.roi_align(ramp, torch.tensor([[0, 20.0, 20.0, 20.4, 20.4]]), (2, 2), 0.25)[0, 0]
    cy = 5.0 + (torch.arange(2) + 0.5) * 0.5
    assert float((tiny - (2 * cy[None, :] + 3 * cy[:, None] + 1)).abs().max()) < 1e-4


def test_roi_align_scalar_equals_vectorised_incl_borders_and_outside():
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(2, 5, 16, 16, generator=g)
    rois = torch.tensor([[0, -20.0, -12.0, 30.0, 44.0],      # reaches outside on the low side: samples with y < -1 contribute 0
                         [1, 30.0, 28.0, 90.0, 80.0],        # reaches outside on the high side: clamp to H-1, zero beyond H
                         [0, 3.3, 7.7, 58.1, 49.9],
                         [1, 10.0, 10.0, 11.0, 10.5]])       # sub-pixel RoI
    a = RA.roi_align(feat, rois, (6, 6), 0.25)
    b = RA.roi_align_fast(feat, rois, (6, 6), 0.25)
    assert float((a - b).abs().max()) < 2e-6
    # every output bin lies in the convex hull of the feature values it can reach (weights are non-negative and sum to <= 1)
    assert float(a.abs().max()) <= float(feat.abs().max()) + 1e-6
    # the batch index column selects the image
    c = RA.roi_align_fast(feat, torch.tensor([[1, 3.3, 7.7, 58.1, 49.9]]), (6, 6), 0.25)
    d = RA.roi_align_fast(feat[1:], torch.tensor([[0, 3.3, 7.7, 58.1, 49.9]]), (6, 6), 0.25)
    assert torch.equal(c, d)


# ------------------------------------------------------------------------------------------------------------------ RK45
def _scipy_run(fun, t0, tf, y0, rtol, atol, max_step, t_eval):
    from scipy.integrate import solve_ivp
    calls = []

    def f(t, y):
        calls.append(t)
        return fun(t, y)

    res = solve_ivp(f, (t0, tf), y0, method='RK45', rtol=rtol, atol=atol, max_step=max_step, t_eval=t_eval)
    return res, calls


@pytest.mark.parametrize('case', ['decay_backward', 'oscillator', 'stiffish'])
def test_rk45_is_bit_identical_to_scipy(case):
    rng = np.random.default_rng(5)
    if case == 'decay_backward':            # integrates from T0 down to eps like score_based_model.py:86-91
        A = rng.normal(size=(40, 40)) * 0.3
        fun = lambda t, y: A @ y * (0.5 + t) + np.sin(3 * t)
        t0, tf, y0 = 0.65, 1e-5, rng.normal(size=40) * 2.5
    elif case == 'oscillator':
        fun = lambda t, y: np.concatenate([y[50:], -4.0 * y[:50]])
        t0, tf, y0 = 0.0, 3.0, rng.normal(size=100)
    else:
        fun = lambda t, y: -50.0 * (y - np.cos(t))
        t0, tf, y0 = 0.0, 1.0, rng.normal(size=7)
    t_eval = np.linspace(t0, tf, 50)
    kw = dict(rtol=3e-3, atol=3e-4, max_step=10)
    res, calls = _scipy_run(fun, t0, tf, y0, t_eval=t_eval, **kw)
    mine_calls = []

    def f(t, y):
        mine_calls.append(t)
        return fun(t, y)

    out = RK.solve_rk45(f, t0, tf, y0, kw['rtol'], kw['atol'], kw['max_step'], t_eval)
    assert out['nfev'] == res.nfev
    assert mine_calls == calls                             # the same RHS evaluation times, in the same order: same steps, same rejections
    assert np.array_equal(out['y'], res.y)                  # dense output at every stamp, bit for bit
    assert len(out['steps']) == (res.nfev - 2) // 6
    assert any(not s[3] for s in out['steps']) or case != 'stiffish'   # the stiff case exercises step rejection


def test_roi_align_geometry_is_float32_like_the_kernel():
    """roi_align_kernel.cpp runs its RoI geometry in T = float for float tensors.  The square hull of a hand box that spans the whole
    256-pixel crop, [1.9469828605651855, 0, 257.9469909667969, 256] x 1/4: 64.48674774 - 0.48674572 is exactly 64 in float32 ->
    ceil(64 / 32) = 2 samples per bin; in double it is 64.000002 -> 3 samples, other sample points for the whole RoI.  (Found by the
    64-image reference fixture: the HIP kernel computed 2, the double-precision restatement 3.)"""
    import numpy as np
    from oracle import roi_align as RA
    box = [1.9469828605651855, 0.0, 257.9469909667969, 256.0]
    x1, y1, bh, bw, gh, gw = RA._geometry(box, 0.25, 32, 32)
    assert (gh, gw) == (2, 2) and bw == np.float32(2.0) and isinstance(x1, np.float32)
    assert (257.9469909667969 * 0.25 - 1.9469828605651855 * 0.25) / 32 > 2.0          # what double arithmetic would have rounded up
    # scalar and vectorised forms agree on that box, and 2 x 2 samples per bin it is: a map that is linear in x is reproduced at the
    # mean sample position of every bin
    feat = torch.arange(64, dtype=torch.float32)[None, None, None, :].expand(1, 1, 64, 64).contiguous()
    rois = torch.tensor([[0.0] + box])
    a, b = RA.roi_align(feat, rois, 32, 0.25), RA.roi_align_fast(feat, rois, 32, 0.25)
    assert torch.allclose(a, b, atol=1e-5)
    want = float(x1) + 2.0 * torch.arange(32, dtype=torch.float32) + 1.0                # bin centre = mean of the samples at +0.5, +1.5
    assert torch.allclose(b[0, 0, 5, :31], want[:31], atol=1e-5)                      # the last bin reaches past the map edge
