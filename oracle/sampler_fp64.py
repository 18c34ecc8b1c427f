"""fp64 referee of the SAMPLER (test infrastructure, like the rest of oracle/): the probability-flow ODE of
``cond_ode_sampler`` (lib/model/score_based_model.py:45-105) solved in float64 -- score network (lib/model/denoiser.py:68-82), sigma(t),
RK45 stage algebra and the final denoise step all in double -- ON THE ACCEPTED STEP SEQUENCE of a given fp32 solve.

Why: the fp64 referee of oracle/referee.py judges the SELECTION chain given the candidates; nothing judged the candidates themselves.
HIP and oracle agree to ~2.5e-5 on the final hypotheses, and that is what flips near-ties in the top-k chain -- but whose rounding is it?
On a fixed step sequence the rows of a solve are independent (the only coupling of a batch is the step-size controller, quirk Q5), so
the fp64 trajectory of any subset of rows is the exact solution of the same discrete scheme: ``max |x_side - x_fp64|`` is the arithmetic
error of that side (fp32 network, fp32 stage values -- the reference's numpy stores f32 stages, oracle/nets.py:185-188 --, fp32 products)
and the two sides' errors can be compared:  ratio = err(HIP) / err(oracle);  <= 1 means the kernels are at least as close to the exact
scheme as the reference's own arithmetic.
"""
import numpy as np
import torch

from . import nets as N
from .rk45 import A, B, C


def _double_sd(sd, p):
    return {k: v.double() for k, v in sd.items() if k.startswith(p + '.')}


def _rhs64(sd64, p, feat, y, t):
    """-(1/2) g(t)^2 score(y, t | feat) in float64"""
    R = y.shape[0]
    ts = torch.full((R, 1), float(t), dtype=torch.float64)
    s = N.denoiser(sd64, p, feat, y, ts)
    sigma = N.SIGMA_MIN * (N.SIGMA_MAX / N.SIGMA_MIN) ** float(t)
    g2 = sigma * sigma * 2.0 * (np.log(N.SIGMA_MAX) - np.log(N.SIGMA_MIN))
    return -0.5 * g2 * s


def solve_on_steps(sd, p, feat, init_x, steps, num_steps, rows=None, dense=False, T0=None):
    """feat (R,1024), init_x (R,D) (the prior draw already scaled), steps = the fp32 solve's log [(t_old, h, err, accepted)]
    (oracle.nets.ode_sample info['steps']).  rows: index tensor of the rows to solve (default all).  Returns x (len(rows), D) float64 =
    the sampler's final output (after the denoise step) in exact arithmetic on that step sequence; with ``dense`` also xs
    (len(rows), num_steps, D): the dense output at t_eval = linspace(T0, eps, num_steps) (RkDenseOutput, scipy _ivp/rk.py), float64."""
    from .rk45 import P
    sd64 = _double_sd(sd, p)
    rows = torch.arange(init_x.shape[0]) if rows is None else rows
    f = feat[rows].double()
    y = init_x[rows].double()
    te = np.linspace(T0, N.EPS_T, num_steps) if dense else None
    xs = torch.zeros((y.shape[0], num_steps, y.shape[1]), dtype=torch.float64) if dense else None
    Pm = torch.tensor(P, dtype=torch.float64)
    with torch.no_grad():
        k_first = None
        for (t, h, _err, accepted) in steps:
            if not accepted:
                continue
            K = [_rhs64(sd64, p, f, y, t) if k_first is None else k_first]
            for s in range(1, 6):
                dy = sum(A[s, j] * K[j] for j in range(s)) * h
                K.append(_rhs64(sd64, p, f, y + dy, t + C[s] * h))
            y_new = y + h * sum(B[j] * K[j] for j in range(6))
            K.append(_rhs64(sd64, p, f, y_new, t + h))              # first-same-as-last: the next step's first stage
            k_first = K[6]
            if dense:
                lo, hi = min(t, t + h), max(t, t + h)
                sel = [i for i, tv in enumerate(te) if (lo < tv <= hi if h > 0 else lo <= tv < hi) or (i == 0 and tv == t)]
                Kt = torch.stack(K, -1)                             # (rows, D, 7)
                Q = Kt @ Pm                                          # (rows, D, 4)
                for i in sel:
                    xx = (te[i] - t) / h
                    pw = torch.tensor([xx, xx ** 2, xx ** 3, xx ** 4], dtype=torch.float64)
                    xs[:, i] = y + h * (Q @ pw)
            y = y_new
        ts = torch.full((y.shape[0], 1), N.EPS_T, dtype=torch.float64)
        sigma = N.SIGMA_MIN * (N.SIGMA_MAX / N.SIGMA_MIN) ** N.EPS_T
        g2 = sigma * sigma * 2.0 * (np.log(N.SIGMA_MAX) - np.log(N.SIGMA_MIN))
        grad = N.denoiser(sd64, p, f, y, ts)
        y = y + (0 - g2 * grad) * ((1 - N.EPS_T) / num_steps)
    return (y, xs) if dense else y


def compare(sd, p, feat_oracle, init_x, steps, num_steps, x_hip, x_oracle, feat_hip=None, steps_hip=None, stride=4):
    """max / rms |x - x_fp64| of both sides over every ``stride``-th row.  Each side is held against the fp64 solve of ITS OWN encoding
    (feat_hip) on ITS OWN accepted step sequence (steps_hip: the HIP solve's log; steps: the oracle's).  The step-size controller
    amplifies rounding: the error estimate h * sum_j E_j K_j is a difference ~1e-3 of the stages, so stage values that differ by 1e-7
    give error norms that differ by ~1e-4 and next step sizes that differ by ~2e-5 -- two equally valid discrete schemes whose
    solutions differ by more than either side's arithmetic error.  What is compared is the sampler's ARITHMETIC, not the feature path's
    (``upstream_max_abs``) and not the controller's choice of h; ``err_hip_on_oracle_scheme_max`` is the HIP result against the fp64 solve
    of the ORACLE's encoding and step sequence, i.e. including both."""
    rows = torch.arange(0, init_x.shape[0], stride)
    x64o = solve_on_steps(sd, p, feat_oracle, init_x, steps, num_steps, rows)
    own = feat_hip is not None or steps_hip is not None
    x64h = solve_on_steps(sd, p, (feat_oracle if feat_hip is None else feat_hip.cpu()), init_x, steps if steps_hip is None else steps_hip, num_steps, rows) if own else x64o
    dh, do = x_hip[rows].double().cpu() - x64h, x_oracle[rows].double() - x64o
    eh, eo = float(dh.abs().max()), float(do.abs().max())
    rh, ro = float(dh.pow(2).mean().sqrt()), float(do.pow(2).mean().sqrt())
    acc = lambda st: [(float(a[0]), float(a[1])) for a in st if a[3]]
    sh, so = acc(steps if steps_hip is None else steps_hip), acc(steps)
    return {'err_hip_max': eh, 'err_oracle_max': eo, 'ratio_max': eh / max(eo, 1e-300), 'err_hip_rms': rh, 'err_oracle_rms': ro,
            'ratio_rms': rh / max(ro, 1e-300), 'err_hip_on_oracle_scheme_max': float((x_hip[rows].double().cpu() - x64o).abs().max()),
            'hip_vs_oracle_max': float((x_hip[rows].double().cpu() - x_oracle[rows].double()).abs().max()),
            'step_size_rel_diff_max': (max(abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(sh, so)) if len(sh) == len(so) and so else None),
            'rows': int(rows.numel()), 'x_scale': float(x64o.abs().max()), 'accepted_steps': len(so)}
