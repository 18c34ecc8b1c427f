"""Score network + on-device RK45 sampler (rows a10/a11 of SURVEY.md 8) against the oracle and the reference fixtures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_blocks.npz'))


def seeded(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(size=shape) * scale).astype(np.float32))


@pytest.fixture(scope='module')
def nets(sd):
    from vpho_amd import ops
    return {k: ops.ScoreNet(sd, f'denoiser_{k}', 'cuda') for k in ('hand', 'obj')}


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_score_matches_oracle(sd, nets, name, D):
    from oracle import nets as N
    bs, S = 3, 50                      # 150 rows: one full 128-row tile + a ragged one; rows share per-image features
    feat, x = seeded((bs, 1024), 30, 0.3), seeded((bs * S, D), 31, 1.5)
    for t in (0.65, 0.3, 1e-5):
        ref = N.denoiser(sd, f'denoiser_{name}', feat[:, None].repeat(1, S, 1).reshape(-1, 1024), x, torch.full((bs * S, 1), t))
        got = nets[name].score(feat.cuda(), x.cuda(), t, S).cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-5 * scale, (t, (got - ref).abs().max().item(), scale)


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_register_ring_pose_encoder_is_bit_identical_to_the_lds_ring_kernel(nets, name, D):
    """round 4: pose_encoder_reg_kernel (weight fragments in a register ring, 35 KB of LDS, 3 barriers) against round 3's
    pose_encoder_kernel (143 KB LDS ring, VPHO_PE_RING=1): the same k order of the same fp32 MFMAs, so the whole score is bit-identical;
    ragged last block (6 387 = 199 x 32 + 19 rows), and a whole ODE solve (the stage-state prologue: y + h sum c_j K_j in fp64)"""
    import os
    bs, S = 3, 2129
    feat, x = seeded((bs, 1024), 60, 0.3).cuda(), seeded((bs * S, D), 61, 1.5).cuda()
    init = seeded((8 * 50, D), 62, 20.0).cuda()
    feat8 = seeded((8, 1024), 63, 0.3).cuda()
    res = {}
    for ring in ('1', '0'):
        os.environ['VPHO_PE_RING'] = ring
        try:
            sc = nets[name].score(feat, x, 0.3, S).clone()
            xs, xf, st = nets[name].sample(feat8, init, 50, 0.65, 12, xs_f64=True)
            res[ring] = (sc, xs.clone(), xf.clone(), st['nfev'])
        finally:
            os.environ.pop('VPHO_PE_RING', None)
    assert torch.isfinite(res['0'][0]).all() and float(res['0'][0].abs().max()) > 0
    assert torch.equal(res['0'][0], res['1'][0])
    assert res['0'][3] == res['1'][3] and torch.equal(res['0'][1], res['1'][1]) and torch.equal(res['0'][2], res['1'][2])


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_64_row_pose_encoder_is_bit_identical_to_the_32_row_kernel(nets, name, D):
    """round 6: pose_encoder_reg64_kernel (64 hypotheses per workgroup: every weight fragment feeds two matrix instructions; the default above
    8 192 rows, VPHO_PE_ROWS=64 / 32 forces either): same k order per output, so scores and a whole ODE solve are bit-identical; ragged last
    block (6 387 = 99 x 64 + 51 rows), a launch at the switch-over size, and the stage-state prologue in controller mode"""
    import os
    bs, S = 3, 2129
    feat, x = seeded((bs, 1024), 60, 0.3).cuda(), seeded((bs * S, D), 61, 1.5).cuda()
    init = seeded((8 * 50, D), 62, 20.0).cuda()
    feat8 = seeded((8, 1024), 63, 0.3).cuda()
    featb, xb = seeded((64, 1024), 64, 0.3).cuda(), seeded((64 * 256, D), 65, 1.5).cuda()      # 16 384 rows: the 64-row kernel by default
    res = {}
    for rows in ('32', '64'):
        os.environ['VPHO_PE_ROWS'] = rows
        try:
            sc = nets[name].score(feat, x, 0.3, S).clone()
            scb = nets[name].score(featb, xb, 0.3, 256).clone()
            xs, xf, st = nets[name].sample(feat8, init, 50, 0.65, 12, xs_f64=True)
            res[rows] = (sc, xs.clone(), xf.clone(), st['nfev'], scb)
        finally:
            os.environ.pop('VPHO_PE_ROWS', None)
    assert torch.isfinite(res['64'][0]).all() and float(res['64'][0].abs().max()) > 0
    assert torch.equal(res['32'][0], res['64'][0]) and torch.equal(res['32'][4], res['64'][4])
    assert res['32'][3] == res['64'][3] and torch.equal(res['32'][1], res['64'][1]) and torch.equal(res['32'][2], res['64'][2])
    assert torch.equal(nets[name].score(featb, xb, 0.3, 256), res['32'][4])                       # the default choice at 16 384 rows


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
@pytest.mark.parametrize('bs,S', [(64, 100), (5, 64), (3, 2129), (2, 600), (7, 40)])
def test_persistent_score_head_is_bit_identical_to_the_one_tile_kernels(nets, name, D, bs, S):
    """round 5: (a) the epilogue's per-image terms (cimg) come from an LDS copy of the <= 3 images a 128-row tile spans (sample_num >= 64)
    instead of 64 global loads per lane (VPHO_HEAD_CB=0 keeps the global loads); (b) score_head_pers_kernel: a workgroup walks several
    tiles, requests the next tile's first stages and tables behind the current tile's last barrier, one output per thread
    (opt-in, VPHO_HEAD_PERS=1; default: one workgroup per tile).  Same values, same order of additions: every score bit-identical -- README batch (tiles
    straddling two and three images, tail tiles, 3.1 tiles per workgroup), sample_num 64 (the smallest that takes the LDS path), ragged
    last tiles, sample_num 40 (global loads, one-tile kernel whatever the switches); and a whole ODE solve (controller mode)."""
    import os
    feat, x = seeded((bs, 1024), 70, 0.3).cuda(), seeded((bs * S, D), 71, 1.5).cuda()
    res = {}
    for key, env in (('base', dict(VPHO_HEAD_CB='0', VPHO_HEAD_PERS='0')), ('cb', dict(VPHO_HEAD_PERS='0')), ('pers', dict(VPHO_HEAD_PERS='1'))):
        os.environ.update(env)
        try:
            res[key] = nets[name].score(feat, x, 0.3, S).clone()
            torch.cuda.synchronize()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert torch.isfinite(res['pers']).all() and float(res['pers'].abs().max()) > 0
    assert torch.equal(res['base'], res['cb']) and torch.equal(res['base'], res['pers'])


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_persistent_score_head_inside_an_ode_solve(nets, name, D):
    import os
    init, feat8 = seeded((8 * 100, D), 72, 20.0).cuda(), seeded((8, 1024), 73, 0.3).cuda()
    res = {}
    for pers in ('0', '1'):
        os.environ['VPHO_HEAD_PERS'] = pers
        try:
            xs, xf, st = nets[name].sample(feat8, init, 100, 0.65, 12, xs_f64=True)
            res[pers] = (xs.clone(), xf.clone(), st['nfev'])
        finally:
            os.environ.pop('VPHO_HEAD_PERS', None)
    assert res['0'][2] == res['1'][2] and torch.equal(res['0'][0], res['1'][0]) and torch.equal(res['0'][1], res['1'][1])


def test_score_tail_tiles_match_oracle(sd, nets):
    """6 387 rows x 32 heads = 1 600 tiles on 512 workgroup slots: the launch runs 48 ordinary tiles per head and the remaining 243
    rows as 32-row tail tiles (the last one ragged).  Rows of every kind of tile against the oracle, and BIT-IDENTICAL to a launch of
    the same rows that is small enough to take ordinary tiles only: a hypothesis' score does not depend on the batch it is evaluated in
    (both tile kinds sum a row's 256 hidden units as the same eight partial sums of 32 in the same order)."""
    from oracle import nets as N
    bs, S = 3, 2129
    R = bs * S
    feat, x = seeded((bs, 1024), 50, 0.3), seeded((R, 96), 51, 1.5)
    got = nets['hand'].score(feat.cuda(), x.cuda(), 0.3, S).cpu()
    assert torch.isfinite(got).all()
    pick = torch.cat([torch.arange(0, 64), torch.arange(6100, 6150), torch.arange(6144 - 8, R)])      # first tiles, last full tile, all tail rows
    ref = N.denoiser(sd, 'denoiser_hand', feat[pick // S], x[pick], torch.full((len(pick), 1), 0.3))
    scale = ref.abs().max().item()
    assert (got[pick] - ref).abs().max().item() <= 2e-5 * scale
    # the tail rows again as the head of a 2-image launch of 250 rows each (500 rows: ordinary tiles only)
    rows = torch.arange(R - 250, R)
    assert int(rows[0]) // S == int(rows[-1]) // S == 2
    small = nets['hand'].score(feat[[2, 2]].cuda(), torch.cat([x[rows], x[rows]]).cuda(), 0.3, 250).cpu()[:250]
    assert torch.equal(small, got[rows])


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_ode_sampler_matches_oracle_and_reference_fixture(sd, nets, name, D):
    """Same run as the reference fixture (tests/golden/make_golden.py: 8 rows, 5 stamps, T0=0.65, seed 5)."""
    from oracle import nets as N
    torch.manual_seed(5)
    init = torch.randn(8, D) * N.ve_prior_sigma(0.65)
    feat = seeded((8, 1024), 22, 0.3)
    xs_o, x_o, info = N.ode_sample(sd, f'denoiser_{name}', feat, init, 0.65, 5)
    xs, x, st = nets[name].sample(feat.cuda(), init.cuda(), 1, 0.65, 5, xs_f64=True)
    assert st['nfev'] == info['nfev'] == int(G[f'ode_{name}_nfev'])
    assert st['nan_count'] == 0
    # accepted/rejected sequence and step sizes follow the oracle's controller
    assert [s[3] for s in st['steps']] == [s[3] for s in info['steps']]
    np.testing.assert_allclose([s[1] for s in st['steps']], [s[1] for s in info['steps']], rtol=2e-4)
    # tolerance: 1e-3 of north_star on the sampled pose, in practice ~1e-5
    assert (xs.cpu() - xs_o).abs().max().item() < 1e-3
    assert (x.cpu() - x_o).abs().max().item() < 1e-3
    assert np.abs(xs.cpu().numpy() - G[f'ode_{name}_xs']).max() < 1e-3
    assert np.abs(x.cpu().numpy() - G[f'ode_{name}_x']).max() < 1e-3


def test_ode_sampler_ragged_rows_and_f32_trajectory(sd, nets):
    from oracle import nets as N
    bs, S, steps = 3, 7, 4             # 21 rows
    feat = seeded((bs, 1024), 40, 0.3)
    init = seeded((bs * S, 96), 41, N.ve_prior_sigma(0.4))
    xs_o, x_o, info = N.ode_sample(sd, 'denoiser_hand', feat[:, None].repeat(1, S, 1).reshape(-1, 1024), init, 0.4, steps)
    xs, x, st = nets['hand'].sample(feat.cuda(), init.cuda(), S, 0.4, steps, xs_f64=False)
    assert xs.dtype == torch.float32 and st['nfev'] == info['nfev']
    assert (xs.cpu().double() - xs_o).abs().max().item() < 1e-3
    assert (x.cpu() - x_o).abs().max().item() < 1e-3


@pytest.mark.parametrize('name,D', [('hand', 96), ('obj', 9)])
def test_full_batch_rk45_follows_the_reference_step_by_step(sd_contrast, name, D):
    """README batch: R = 64 images x 100 hypotheses = 6400 rows under ONE RK45 controller (quirk Q5, score_based_model.py:91).
    Fixture = the reference's own cond_ode_sampler + scipy on the same encodings and prior draw (make_golden_ode_fullbatch.py):
    the device-side controller takes the same number of RHS evaluations, accepts / rejects the same attempts with the same step
    sizes, and the strided samples / dense-output stamps agree to 1e-3 (north star; observed ~1e-5)."""
    from oracle import nets as N
    from tests import _ode_fixture as OF
    from vpho_amd import ops
    net = ops.ScoreNet(sd_contrast, f'denoiser_{name}', 'cuda')
    enc, init = OF.inputs(name, D, N.ve_prior_sigma(OF.T0))
    xs, x, st = net.sample(enc.cuda(), init.cuda(), OF.S, OF.T0, OF.STEPS, xs_f64=True)
    torch.cuda.synchronize()
    assert st['nan_count'] == 0
    ex, exs = OF.check(name, xs.cpu(), x.cpu(), st['steps'], st['nfev'])
    print(name, 'max abs x', ex, 'xs', exs)


@pytest.mark.parametrize('name,D', [('obj', 9), ('hand', 96)])
def test_nan_guard_matches_reference(sd_contrast, name, D):
    """The guard of score_based_model.py:65-72 against the reference's own sampler run on a score network that returns NaN in one head and
    +-inf in another at every evaluation (make_golden_nan_guard.py): NaN and +-inf are zeroed inside the solve (same RHS-evaluation
    count as scipy, the planted dimensions keep their start value, the NaNs are counted), and the final denoise evaluation is NOT
    guarded, so the returned sample is NaN / -inf / +inf exactly where the reference's is."""
    from oracle import nets as N
    from tests import _nan_fixture as NF
    from vpho_amd import ops
    net = ops.ScoreNet(NF.planted_state_dict(sd_contrast, name), f'denoiser_{name}', 'cuda')
    enc, init = NF.inputs(name, D, N.ve_prior_sigma(NF.T0))
    xs, x, st = net.sample(enc.cuda(), init.cuda(), NF.S, NF.T0, NF.STEPS, xs_f64=True)
    torch.cuda.synchronize()
    assert st['nan_count'] == 3 * NF.BS * NF.S * (st['nfev'] - 1)            # three NaN entries per row and guarded evaluation
    NF.check(name, xs, x, init, st['nfev'], tol=1e-4)
    # the bare score (denoiser.py:68-82 without the sampler's wrapper) passes NaN / inf through
    sc = net.score(enc.cuda(), init.cuda(), 0.3, NF.S).cpu()
    nh, ih = (int(v) for v in NF.F[f'{name}_plant'])
    assert bool(torch.isnan(sc[:, 3 * nh:3 * nh + 3]).all()) and bool(torch.isinf(sc[:, 3 * ih:3 * ih + 3]).all())
