"""Independent cross-checks of the oracle's three third-party leaves (CPU): every function of ``oracle/rotations.py`` against
``scipy.spatial.transform.Rotation``, ``oracle/roi_align.py`` against a grid_sample formulation on 1 000 random boxes, and
``oracle/mano.py`` against a direct float64 LBS from the MANO paper's equations (tests/_leaf_independent.py).  The packages the
reference takes these from (pytorch3d, torchvision, manopth; call sites VPHO.py:125-128,316-323, head_mano.py:10-26,78-87) are not
installable here; a second formulation is what can be had.  The -m gpu twins (tests/test_gpu_leaf_crosschecks.py) hold the HIP
kernels to the same second formulations.
"""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import mano as OM
from oracle import roi_align as RA
from oracle import rotations as R
from tests import _leaf_independent as L

N_ROT = 100_000


@pytest.fixture(scope='module')
def rotvecs():
    return L.random_rotations(N_ROT, seed=11, near_pi=2000, tiny=2000)


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def test_every_rotation_conversion_agrees_with_scipy(rotvecs):
    rot = Rotation.from_rotvec(rotvecs)
    m, q = rot.as_matrix(), L.scipy_quat_wxyz(rot)                         # q: real part first and >= 0, like pytorch3d >= 0.7.6
    ang = np.linalg.norm(rotvecs, axis=-1)
    # axis-angle -> quaternion / matrix (incl. the Taylor branch below 1e-6 and angles up to pi)
    assert float((R.axis_angle_to_quaternion(_t(rotvecs)) - _t(q)).abs().max()) < 1e-12
    assert float((R.axis_angle_to_matrix(_t(rotvecs)) - _t(m)).abs().max()) < 1e-12
    # quaternion -> matrix, also for non-unit quaternions and for -q
    s = np.random.default_rng(0).uniform(0.3, 3.0, (len(q), 1))
    assert float((R.quaternion_to_matrix(_t(q * s)) - _t(m)).abs().max()) < 1e-12
    assert float((R.quaternion_to_matrix(_t(-q)) - _t(m)).abs().max()) < 1e-12
    # matrix -> quaternion: all four candidate branches occur in the sample; equal to scipy's canonical quaternion up to the sign
    # of the whole quaternion where the real part is ~0 (angle ~ pi: either sign is "standardised")
    got = R.matrix_to_quaternion(_t(m)).numpy()
    branch = np.argmax(np.stack([1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2], 1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2],
                                 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2]], -1), -1)
    assert set(branch.tolist()) == {0, 1, 2, 3}
    sign = np.where(np.abs(q[:, :1]) < 1e-6, np.sign((got * q).sum(-1, keepdims=True)), 1.0)
    assert float(np.abs(got - sign * q).max()) < 1e-9
    assert bool((got[:, 0] >= 0).all())
    # quaternion -> axis-angle and matrix -> axis-angle: the rotation vector itself (angle in [0, pi]); within 1e-6 of pi the axis
    # sign is arbitrary, so those rows are compared as rotations
    back_q, back_m = R.quaternion_to_axis_angle(_t(q)).numpy(), R.matrix_to_axis_angle(_t(m)).numpy()
    safe = ang < np.pi - 1e-6
    assert float(np.abs(back_q - rotvecs)[safe].max()) < 1e-9 and float(np.abs(back_m - rotvecs)[safe].max()) < 1e-7
    assert float(L.rotation_angle_between(Rotation.from_rotvec(back_m).as_matrix(), m).max()) < 1e-7
    # pytorch3d's quaternion_to_axis_angle does NOT reduce q with a negative real part: the angle comes out in (pi, 2 pi]; same rotation
    neg = R.quaternion_to_axis_angle(_t(-q[safe & (ang > 1e-3)])).numpy()
    assert float(np.linalg.norm(neg, axis=-1).min()) > np.pi - 1e-9
    assert float(L.rotation_angle_between(Rotation.from_rotvec(neg).as_matrix(), m[safe & (ang > 1e-3)]).max()) < 1e-9


def test_rot6d_conversions_agree_with_a_cross_product_construction_and_scipy(rotvecs):
    m = Rotation.from_rotvec(rotvecs[:20000]).as_matrix()
    # a rotation's first two rows ARE its 6-d code and convert back to it
    d6 = R.matrix_to_rotation_6d(_t(m))
    assert float((R.rotation_6d_to_matrix(d6) - _t(m)).abs().max()) < 1e-12
    # arbitrary (non-orthonormal) 6-vectors: Gram-Schmidt == the cross-product construction; a proper rotation for scipy
    raw = np.random.default_rng(3).normal(size=(20000, 6)) * np.random.default_rng(4).uniform(0.1, 5, (20000, 1))
    got = R.rotation_6d_to_matrix(_t(raw)).numpy()
    assert float(np.abs(got - L.rot6d_to_matrix_by_cross_products(raw)).max()) < 1e-9
    assert float(np.abs(Rotation.from_matrix(got).as_matrix() - got).max()) < 1e-9       # scipy would re-orthonormalise an improper input
    assert float(np.abs(np.linalg.det(got) - 1).max()) < 1e-9


def test_float32_conversions_stay_within_float32_of_scipy(rotvecs):
    """the dtype of the path; matrix_to_axis_angle is ill-conditioned near pi (d angle / d matrix ~ 1 / sin), so compare rotations"""
    rv = rotvecs[:20000]
    m = Rotation.from_rotvec(rv).as_matrix()
    aa32 = R.matrix_to_axis_angle(_t(m, torch.float32)).double().numpy()
    assert float(L.rotation_angle_between(Rotation.from_rotvec(aa32).as_matrix(), m).max()) < 2e-6
    m32 = R.axis_angle_to_matrix(_t(rv, torch.float32)).double().numpy()
    assert float(np.abs(m32 - m).max()) < 1e-6
    mid = np.linalg.norm(rv, axis=-1) < 2.5
    assert float(np.abs(aa32 - rv)[mid].max()) < 5e-6


def test_manopth_rodrigues_agrees_with_scipy(rotvecs):
    rv = rotvecs[:20000]
    m = OM.batch_rodrigues(_t(rv)).view(-1, 3, 3).numpy()
    assert float(np.abs(m - Rotation.from_rotvec(rv).as_matrix()).max()) < 5e-8        # manopth adds 1e-8 to the vector before the norm


# ------------------------------------------------------------------------------------------------------------- RoIAlign
@pytest.mark.parametrize('out_size', [32, 7])
def test_roi_align_agrees_with_the_grid_sample_formulation_on_random_boxes(out_size):
    """1 000 boxes at the path's scale 1/4 on 64 x 64 maps (32 bins: the path's geometry incl. exactly-integer bin sizes) and
    300 at 7 bins (ragged sample counts); boxes that leave the map, sub-pixel boxes, the box of round 3's bug"""
    n = 1000 if out_size == 32 else 300
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(4, 3, 64, 64, generator=g)
    boxes = L.random_boxes(n, seed=6)
    rois = torch.cat([torch.randint(0, 4, (n, 1), generator=g).float(), boxes], 1)
    want = L.roi_align_by_grid_sample(feat, rois, out_size, 0.25)
    got = RA.roi_align_fast(feat, rois, out_size, 0.25)
    assert float((got.double() - want).abs().max()) < 2e-5                   # fp32 interpolation weights against float64 ones
    # the scalar transcription on a subset (slow loops)
    pick = torch.arange(0, n, n // 8)
    got_s = RA.roi_align(feat, rois[pick], out_size, 0.25)
    assert float((got_s.double() - want[pick]).abs().max()) < 2e-5
    # sample counts: the two formulations must have made the same adaptive grid, or constant maps would still agree but these not
    assert float(want.abs().max()) > 1.0


# ----------------------------------------------------------------------------------------------------------------- MANO
def test_mano_agrees_with_a_direct_float64_lbs(assets):
    mano = assets['mano']
    rng = np.random.default_rng(9)
    pose = rng.normal(size=(64, 48)) * 0.5
    pose[:4] = 0
    pose[4:8, 3:] = 0                                                       # wrist rotation only
    pose[8:12] *= 3.0                                                       # large angles
    betas = rng.normal(size=(64, 10)) * 0.8
    betas[:2] = 0
    wv, wj = L.mano_lbs_fp64(mano, pose, betas)
    a64 = {k: np.asarray(a, np.float64) for k, a in mano.items()}
    gv, gj = OM.get_hand_verts(a64, _t(pose), _t(betas))
    assert float(np.abs(gv.numpy() - wv).max()) < 1e-8 and float(np.abs(gj.numpy() - wj).max()) < 1e-8      # metres; manopth's +1e-8 rad
    # float32, the dtype of the path
    gv32, gj32 = OM.get_hand_verts(mano, _t(pose, torch.float32), _t(betas, torch.float32))
    assert float(np.abs(gv32.double().numpy() - wv).max()) < 2e-6 and float(np.abs(gj32.double().numpy() - wj).max()) < 2e-6
