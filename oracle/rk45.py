"""Dormand-Prince RK45 with the scipy ``solve_ivp`` controller -- restatement of
scipy.integrate._ivp.rk (RK45 tableau, rk_step, _step_impl, RkDenseOutput),
_ivp.common (select_initial_step, RMS norm) and the t_eval loop of _ivp.ivp.solve_ivp,
as called by the reference at score_based_model.py:91
(method='RK45', rtol=3e-3, atol=3e-4, max_step=10, t_eval=linspace(T, eps, num_steps)).
Pinned against the installed scipy by tests/test_oracle_leaves.py.
"""
import numpy as np

C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1])
A = np.array([
    [0, 0, 0, 0, 0],
    [1 / 5, 0, 0, 0, 0],
    [3 / 40, 9 / 40, 0, 0, 0],
    [44 / 45, -56 / 15, 32 / 9, 0, 0],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729, 0],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
])
B = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])
E = np.array([-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40])
P = np.array([
    [1, -8048581381 / 2820520608, 8663915743 / 2820520608, -12715105075 / 11282082432],
    [0, 0, 0, 0],
    [0, 131558114200 / 32700410799, -68118460800 / 10900136933, 87487479700 / 32700410799],
    [0, -1754552775 / 470086768, 14199869525 / 1410260304, -10690763975 / 1880347072],
    [0, 127303824393 / 49829197408, -318862633887 / 49829197408, 701980252875 / 199316789632],
    [0, -282668133 / 205662961, 2019193451 / 616988883, -1453857185 / 822651844],
    [0, 40617522 / 29380423, -110615467 / 29380423, 69997945 / 29380423],
])
SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0
ERR_EXP = -1.0 / 5.0


def _norm(x):
    return np.linalg.norm(x) / x.size ** 0.5


def select_initial_step(fun, t0, y0, t_bound, max_step, f0, direction, order, rtol, atol):
    interval_length = abs(t_bound - t0)
    scale = atol + np.abs(y0) * rtol
    d0 = _norm(y0 / scale)
    d1 = _norm(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval_length)
    y1 = y0 + h0 * direction * f0
    f1 = fun(t0 + h0 * direction, y1)
    d2 = _norm((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1 / (order + 1))
    return min(100 * h0, h1, interval_length, max_step)


def solve_rk45(fun, t0, tf, y0, rtol, atol, max_step, t_eval):
    """Returns dict(y=(n, len(t_eval)), nfev, steps=[(t_old, h, error_norm, accepted)], t, y_final)."""
    y = np.asarray(y0, dtype=np.float64).copy()
    n = y.size
    nfev = [0]

    def f(t, yy):
        nfev[0] += 1
        return np.asarray(fun(t, yy), dtype=np.float64)

    direction = np.sign(tf - t0) if tf != t0 else 1.0
    t = float(t0)
    fcur = f(t, y)
    h_abs = select_initial_step(f, t, y, tf, max_step, fcur, direction, 4, rtol, atol)
    K = np.empty((7, n))
    te = np.asarray(t_eval, dtype=np.float64)
    if tf < t0:
        te = te[::-1]
        te_i = te.shape[0]
    else:
        te_i = 0
    ys, log = [], []
    while t != tf:
        min_step = 10 * np.abs(np.nextafter(t, direction * np.inf) - t)
        h_abs = max_step if h_abs > max_step else (min_step if h_abs < min_step else h_abs)
        accepted = rejected = False
        while not accepted:
            if h_abs < min_step:
                raise RuntimeError('RK45: step size too small')
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - tf) > 0:
                t_new = tf
            h = t_new - t
            h_abs = np.abs(h)
            K[0] = fcur
            for s in range(1, 6):
                dy = np.dot(K[:s].T, A[s, :s]) * h
                K[s] = f(t + C[s] * h, y + dy)
            y_new = y + h * np.dot(K[:-1].T, B)
            f_new = f(t + h, y_new)
            K[-1] = f_new
            scale = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
            err = _norm(np.dot(K.T, E) * h / scale)
            if err < 1:
                factor = MAX_FACTOR if err == 0 else min(MAX_FACTOR, SAFETY * err ** ERR_EXP)
                if rejected:
                    factor = min(1, factor)
                log.append((t, h, err, True))
                h_abs *= factor
                accepted = True
            else:
                log.append((t, h, err, False))
                h_abs *= max(MIN_FACTOR, SAFETY * err ** ERR_EXP)
                rejected = True
        t_old, y_old = t, y
        t, y, fcur = t_new, y_new, f_new
        # dense output on the t_eval stamps inside (t_old, t]
        if direction > 0:
            te_new = np.searchsorted(te, t, side='right')
            step = te[te_i:te_new]
        else:
            te_new = np.searchsorted(te, t, side='left')
            step = te[te_new:te_i][::-1]
        if step.size > 0:
            Q = K.T.dot(P)
            x = (step - t_old) / h
            p = np.cumprod(np.tile(x, (4, 1)), axis=0)
            yy = h * np.dot(Q, p) + y_old[:, None]
            ys.append(yy)
            te_i = te_new
    return dict(y=np.hstack(ys) if ys else np.zeros((n, 0)), nfev=nfev[0], steps=log, y_final=y)
