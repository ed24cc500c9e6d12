"""Probability-flow ODE sampler — drop-in for get_ode_sampler (sgmse/sampling/__init__.py:71-141).

The reference integrates dx/dt = theta (y - x) - 1/2 g(t)^2 score(x, t, y) from T to eps with
scipy.integrate.solve_ivp(RK45, rtol = atol = 1e-5) on a flattened host copy of the state (one
device <-> host round trip per right-hand side).  Here the solver state lives on the device:
an explicit Dormand-Prince 5(4) integrator with scipy's step-size controller (same error norm,
safety factor 0.9, growth limits [0.2, 10], same initial-step heuristic), so the accepted steps and
the number of function evaluations follow scipy's.  The solver algebra runs in fused HIP kernels (one pass per
Runge-Kutta stage, one for the scaled error norm); every right-hand side is one NCSN++ evaluation on the HIP engine
plus one fused drift pass; the host reads one scalar per attempted step.
"""
import math

import numpy as np
import torch

from ..sdes import OUVESDE
from .predictors import ReverseDiffusionPredictor

# Dormand-Prince coefficients (scipy.integrate._ivp.rk.RK45)
_C = [0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1]
_A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
      [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
      [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
_B = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
_E = [-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40]
_SAFETY, _MIN_FACTOR, _MAX_FACTOR = 0.9, 0.2, 10.0
_ERR_EXP = -1.0 / 5.0


def get_ode_sampler(sde, score_fn, y, inverse_scaler=None, denoise=True, rtol=1e-5, atol=1e-5, method="RK45",
                    eps=3e-2, device=None, noise_fn=None, seed=None, conditioning=None, **kwargs):
    """Probability-flow ODE sampler: Dormand-Prince RK45 with scipy's step controller (solve_ivp's `RK45`, which
    sampling/__init__.py:71-141 runs on the host over the flattened COMPLEX state: its norms are
    ||v|| / sqrt(n) over the n complex elements).  Everything per element runs in HIP kernels: one fused pass per
    stage (storm_rk_combine) and one for the scaled error norm (storm_rk_scaled_sumsq); the host reads ONE scalar per
    attempted step (the error norm that decides acceptance) - no per-stage synchronisation."""
    if method != "RK45":
        raise NotImplementedError("only RK45 (Dormand-Prince) is implemented on the device")
    from .. import ops
    from .noise import NoiseSource
    noise = NoiseSource(seed=seed, noise_fn=noise_fn)
    predictor = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False, noise=noise)
    rsde = sde.reverse(score_fn, probability_flow=True)

    def ode_sampler(z=None, **unused):
        with torch.no_grad():
            yy = y.contiguous()
            B = yy.shape[0]
            n = yy.numel()
            nfev = 0

            def f(t, x):
                nonlocal nfev
                nfev += 1
                vec_t = torch.full((B,), float(t), device=yy.device, dtype=torch.float32)
                if isinstance(sde, OUVESDE):               # fused: theta (y - x) - 1/2 g^2 score in one pass
                    score = rsde._score(x, vec_t, (yy,), dict(conditioning=conditioning))
                    return ops.ouve_pf_drift(sde, x, yy, score.contiguous(), vec_t)
                return rsde.sde(x, vec_t, yy, conditioning=conditioning)[0].contiguous()

            def norm(sumsq):                                   # scipy: ||v|| / sqrt(size); ONE host read
                return math.sqrt(float(sumsq) / n)

            zz, sd, off = noise.next(yy)
            x = (sde.prior_sampling(yy.shape, yy, z=zz, seed=sd, offset=off) if z is None else z).contiguous()
            t, t_end = float(sde.T), float(eps)
            direction = -1.0
            f0 = f(t, x)
            # scipy's select_initial_step (three scalars, once)
            d0 = norm(ops.rk_scaled_sumsq(x, None, [x], None, 1.0, atol, rtol, mode=-1))
            d1 = norm(ops.rk_scaled_sumsq(x, None, [f0], None, 1.0, atol, rtol, mode=-1))
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            x1 = ops.rk_combine(x, [f0], [1.0], h0 * direction)
            f1 = f(t + h0 * direction, x1)
            d2 = norm(ops.rk_scaled_sumsq(x, None, [f1, f0], None, 1.0, atol, rtol, mode=-2)) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1 / 5)
            h_abs = min(100 * h0, h1)
            fk = f0
            while (t - t_end) * direction < 0:
                min_step = 10 * abs(np.nextafter(t, direction * np.inf) - t)
                h_abs = max(h_abs, min_step)
                accepted, rejected = False, False
                while not accepted:
                    if h_abs < min_step:
                        raise RuntimeError("ODE step size underflow")
                    h = h_abs * direction
                    t_new = t + h
                    if direction * (t_new - t_end) > 0:
                        t_new = t_end
                    h = t_new - t
                    h_abs = abs(h)
                    K = [fk]
                    for s in range(1, 6):
                        K.append(f(t + _C[s] * h, ops.rk_combine(x, K, _A[s][:s], h)))
                    x_new = ops.rk_combine(x, K, _B, h)
                    f_new = f(t_new, x_new)
                    K.append(f_new)
                    err_norm = norm(ops.rk_scaled_sumsq(x, x_new, K, _E, h, atol, rtol))
                    if err_norm < 1:
                        factor = _MAX_FACTOR if err_norm == 0 else min(_MAX_FACTOR, _SAFETY * err_norm ** _ERR_EXP)
                        if rejected:
                            factor = min(1.0, factor)
                        h_abs *= factor
                        accepted = True
                    else:
                        h_abs *= max(_MIN_FACTOR, _SAFETY * err_norm ** _ERR_EXP)
                        rejected = True
                t, x, fk = t_new, x_new, f_new
            nfe = nfev
            if denoise:
                vec_eps = torch.ones(B, device=yy.device) * eps
                _, x = predictor.denoise_fn(x, vec_eps, yy, conditioning=conditioning)
            if inverse_scaler is not None:
                x = inverse_scaler(x)
            return x, nfe

    return ode_sampler
