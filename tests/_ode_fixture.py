"""The reference's full-batch ODE solves (tests/golden/golden_ode_fullbatch.npz, written by make_golden_ode_fullbatch.py from the
reference's own cond_ode_sampler + the installed scipy): R = 64 x 100 rows under ONE RK45 controller (quirk Q5)."""
import os

import numpy as np
import torch

F = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'golden_ode_fullbatch.npz'))
BS, S, STEPS = (int(v) for v in F['cfg'])
T0 = float(F['T0'])


def seeded(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(size=shape) * scale).astype(np.float32))


def inputs(name, D, sigma):
    """(per-image encodings (BS,1024), prior draw x sigma (BS*S, D)) exactly as the fixture's run made them (sde.py:26-28)."""
    enc = seeded((BS, 1024), int(F[f'{name}_feat_seed']), 0.3)
    state = torch.get_rng_state()
    torch.manual_seed(int(F[f'{name}_draw_seed']))
    init = torch.randn(BS * S, D) * sigma
    torch.set_rng_state(state)
    return enc, init


def reference_steps(name):
    """[(t, |h|, accepted)] of every RK45 attempt scipy made, recovered from the times at which the reference's denoiser was called:
    2 start-up calls (f(T0), the probe of select_initial_step), then per attempt t + h*(1/5, 3/10, 4/5, 8/9, 1) and the
    first-same-as-last evaluation at t + h, then the predictor's call at eps (score_based_model.py:95-104)."""
    c = np.asarray(F[f'{name}_tcalls'], dtype=np.float64)
    assert (len(c) - 3) % 6 == 0 and abs(c[-1] - 1e-5) < 1e-9
    n = (len(c) - 3) // 6
    steps = []
    for i in range(n):
        a = c[2 + 6 * i: 8 + 6 * i]
        h = (a[4] - a[0]) * 5.0 / 4.0                   # (t + h) - (t + h/5)
        t = a[4] - h
        assert abs(a[5] - a[4]) < 1e-6
        steps.append([t, abs(h), None])
    for i in range(n):
        t_next = steps[i + 1][0] if i + 1 < n else None
        t, h = steps[i][0], steps[i][1]
        steps[i][2] = True if t_next is None else bool(abs(t_next - (t - h)) < abs(t_next - t))
    return [tuple(s) for s in steps]


def check(name, xs, x, steps, nfev, x_tol=1e-3):
    """xs (R,STEPS,D), x (R,D) CPU tensors of the side under test; steps [(t, h, err, accepted)]; nfev incl. the predictor call."""
    ref = reference_steps(name)
    assert nfev == len(F[f'{name}_tcalls']), (nfev, len(F[f'{name}_tcalls']))
    assert [bool(s[3]) for s in steps] == [r[2] for r in ref], 'accept / reject sequence'
    np.testing.assert_allclose([abs(s[1]) for s in steps], [r[1] for r in ref], rtol=5e-4)       # tcalls are float32 values
    np.testing.assert_allclose([s[0] for s in steps], [r[0] for r in ref], rtol=5e-4, atol=1e-6)
    ex = float((x.double()[::16] - torch.as_tensor(F[f'{name}_x'])).abs().max())
    exs = float((xs.double()[::128] - torch.as_tensor(F[f'{name}_xs']).double()).abs().max())
    assert ex < x_tol and exs < x_tol, (ex, exs)
    return ex, exs
