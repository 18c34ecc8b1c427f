"""The reference's cond_ode_sampler with NaN and +-inf planted in the score (tests/golden/golden_nan_guard.npz, written by
make_golden_nan_guard.py from the reference's own sampler + the installed scipy): the guard of score_based_model.py:65-72 zeroes NaN and
+-inf inside the solve, the final denoise evaluation (:95-102) is not guarded."""
import os

import numpy as np
import torch

F = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_nan_guard.npz'))
BS, S, STEPS = (int(v) for v in F['cfg'])
T0 = float(F['T0'])


def planted_state_dict(sd, name):
    nh, ih = (int(v) for v in F[f'{name}_plant'])
    sd = dict(sd)
    b = sd[f'denoiser_{name}.head.head.2.bias'].clone()
    b[nh] = float('nan')
    b[ih] = torch.tensor([float('inf'), float('-inf'), float('inf')])
    sd[f'denoiser_{name}.head.head.2.bias'] = b
    return sd


def inputs(name, D, sigma):
    enc = torch.from_numpy((np.random.default_rng(int(F[f'{name}_feat_seed'])).normal(size=(BS, 1024)) * 0.3).astype(np.float32))
    state = torch.get_rng_state()
    torch.manual_seed(int(F[f'{name}_draw_seed']))
    init = torch.randn(BS * S, D) * sigma
    torch.set_rng_state(state)
    return enc, init


def check(name, xs, x, init, nfev, tol=1e-4):
    """xs (R,STEPS,D), x (R,D) of the side under test; the same RHS-evaluation count as scipy made; the planted dimensions keep
    their start value through the trajectory and are NaN / +-inf in the returned sample exactly where the reference's are."""
    nh, ih = (int(v) for v in F[f'{name}_plant'])
    rx, rxs = torch.as_tensor(F[f'{name}_x']).double(), torch.as_tensor(F[f'{name}_xs']).double()
    assert nfev == len(F[f'{name}_tcalls']), (nfev, len(F[f'{name}_tcalls']))
    x, xs = x.double().cpu(), xs.double().cpu()
    assert bool(torch.isfinite(xs).all()) and float((xs - rxs).abs().max()) < tol
    planted = [3 * nh + i for i in range(3)] + [3 * ih + i for i in range(3)]
    for d in planted:                                   # zero right-hand side: the dimension never moves
        assert float((xs[:, :, d] - init.double()[:, None, d]).abs().max()) < 1e-6, d
    assert bool(torch.isnan(x[:, 3 * nh:3 * nh + 3]).all()) and torch.equal(torch.isnan(x), torch.isnan(rx))
    assert torch.equal(x[:, 3 * ih:3 * ih + 3], rx[:, 3 * ih:3 * ih + 3])          # -inf / +inf / -inf: the unguarded denoise step
    fin = torch.isfinite(rx)
    assert float((x[fin] - rx[fin]).abs().max()) < tol
