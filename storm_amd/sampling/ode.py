"""Probability-flow ODE sampler — drop-in for get_ode_sampler (sgmse/sampling/__init__.py:71-141).

The reference integrates dx/dt = theta (y - x) - 1/2 g(t)^2 score(x, t, y) from T to eps with
scipy.integrate.solve_ivp(RK45, rtol = atol = 1e-5) on a flattened host copy of the state (one
device <-> host round trip per right-hand side).  Here the solver state lives on the device:
an explicit Dormand-Prince 5(4) integrator with scipy's step-size controller (same error norm,
safety factor 0.9, growth limits [0.2, 10], same initial-step heuristic), so the accepted steps and
the number of function evaluations follow scipy's.  The solver algebra runs in fused HIP kernels (one pass per
Runge-Kutta stage, one for the scaled error norm); every right-hand side is one NCSN++ evaluation on the HIP engine
plus one fused drift pass; the host reads the per-row error sums once per attempted step.

Batches.  solve_ivp sees ONE flattened state, so `get_ode_sampler(sde, score_fn, y)` over a batch couples its rows through
the error norm exactly like the reference function does (per_row=False, the default: reference semantics for whatever
batch it is handed).  The reference MODEL never hands it more than one utterance (model.py:224-244, minibatch = 1): every
utterance has its own accepted / rejected step sequence.  per_row=True keeps that inside a batch: every row has its own
t, h, error norm and accept / reject decision, so row b of a batched run equals the batch-1 run of utterance b - this is
what ScoreModel.enhance_batch uses.  A row that has reached eps LEAVES the batch (compact=True, the default with per_row):
its end state is set aside and the state tensors shrink to the rows still integrating, so the network is never evaluated
on a finished row (compact=False: finished rows idle with h = 0 until the slowest row is done - the round-5 behaviour).
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
                    eps=3e-2, device=None, noise_fn=None, seed=None, conditioning=None, per_row=False, compact=True, **kwargs):
    """Probability-flow ODE sampler: Dormand-Prince RK45 with scipy's step controller (solve_ivp's `RK45`, which
    sampling/__init__.py:71-141 runs on the host over the flattened COMPLEX state: its norms are
    ||v|| / sqrt(n) over the n complex elements).  Everything per element runs in HIP kernels: one fused pass per
    stage (storm_rk_combine_rows) and one for the scaled error sums (storm_rk_scaled_sumsq_rows); the host reads the B row
    sums once per attempted step (they decide acceptance) - no per-stage synchronisation.
    per_row: one step controller per row instead of one for the whole batch (module docstring).
    compact (per_row only): rows that reached eps leave the batch instead of idling in every further evaluation.
    Returns fn() -> (x, nfe); nfe = score evaluations executed; fn.nfev_rows = evaluations each row needed on its own;
    fn.rows_evaluated = rows summed over the executed evaluations (what the network really computed)."""
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
            n_row = yy.numel() // B
            groups = [[b] for b in range(B)] if per_row else [list(range(B))]
            G = len(groups)
            shrink = bool(compact) and per_row and B > 1
            executed = rows_evaluated = 0
            # what the right-hand side reads besides x: the rows still integrating (all of them until a row leaves)
            y_c, cond_c = yy, conditioning

            def rows(vals):                                   # per group -> per row
                return [vals[g] for g, ids in enumerate(groups) for _ in ids]

            def f(t_g, x):
                nonlocal executed, rows_evaluated
                executed += 1
                rows_evaluated += x.shape[0]
                t_host = torch.tensor(rows(t_g), dtype=torch.float32)
                vec_t = t_host.to(yy.device)
                if isinstance(sde, OUVESDE):               # fused: theta (y - x) - 1/2 g^2 score in one pass; g(t) for the B
                    score = rsde._score(x, vec_t, (y_c,), dict(conditioning=cond_c))           # rows in the reference's own ops
                    return ops.ouve_pf_drift_g(sde, x, y_c, score.contiguous(), sde.diffusion(t_host))
                if hasattr(sde, "drift_rows"):             # coefficient-table form (OUVPSDE): a(t_b), g(t_b) per row
                    score = rsde._score(x, vec_t, (y_c,), dict(conditioning=cond_c))
                    return ops.sde_pf_drift_rows(x, y_c, score.contiguous(), sde.drift_rows(t_host), sde.diffusion(t_host))
                return rsde.sde(x, vec_t, y_c, conditioning=cond_c)[0].contiguous()

            def norms(sumsq_rows):                            # scipy: ||v|| / sqrt(size) per solver state; ONE host read
                s = sumsq_rows.cpu().tolist()
                return [math.sqrt(sum(s[b] for b in ids) / (n_row * len(ids))) for ids in groups]

            zz, sd, off = noise.next(yy)
            # (a caller's start state is cloned: accepted rows are copied into x32 in place, and the caller may reuse z)
            x32 = sde.prior_sampling(yy.shape, yy, z=zz, seed=sd, offset=off).contiguous() if z is None else z.contiguous().clone()
            x = x32.to(torch.complex128)                      # the solver state is complex128, as in scipy (module docstring)
            t_end, direction = float(eps), -1.0
            t = [float(sde.T)] * G
            f0 = f(t, x32)
            # scipy's select_initial_step, per solver state (three scalars each, once)
            d0 = norms(ops.rk_scaled_sumsq_rows(x, None, [], None, None, atol, rtol, mode=-3))
            d1 = norms(ops.rk_scaled_sumsq_rows(x, None, [f0], None, None, atol, rtol, mode=-1))
            h0 = [1e-6 if (d0[g] < 1e-5 or d1[g] < 1e-5) else 0.01 * d0[g] / d1[g] for g in range(G)]
            x1 = ops.rk_combine_rows(x, [f0], [1.0], rows([h * direction for h in h0]))
            f1 = f([t[g] + h0[g] * direction for g in range(G)], x1)
            d2 = norms(ops.rk_scaled_sumsq_rows(x, None, [f1, f0], None, None, atol, rtol, mode=-2))
            h_abs = []
            for g in range(G):
                dd2 = d2[g] / h0[g]
                h1 = max(1e-6, h0[g] * 1e-3) if (d1[g] <= 1e-15 and dd2 <= 1e-15) else (0.01 / max(d1[g], dd2)) ** (1 / 5)
                h_abs.append(min(100 * h0[g], h1))
            nfev = [2] * G
            fk = f0
            new_step, rejected = [True] * G, [False] * G
            min_step = [0.0] * G
            active = [(t[g] - t_end) * direction < 0 for g in range(G)]
            # compaction (per_row): `orig[g]` = the batch row that solver state g integrates; finished rows are parked in x_done
            orig = list(range(G))
            x_done, nfev_done = None, [0] * G
            while any(active):
                h, t_new = [0.0] * G, list(t)
                for g in range(G):
                    if not active[g]:
                        continue
                    if new_step[g]:                           # RK45._step_impl's preamble
                        min_step[g] = 10 * abs(np.nextafter(t[g], direction * np.inf) - t[g])
                        h_abs[g] = max(h_abs[g], min_step[g])
                        new_step[g], rejected[g] = False, False
                    if h_abs[g] < min_step[g]:
                        raise RuntimeError("ODE step size underflow")
                    tn = t[g] + h_abs[g] * direction
                    if direction * (tn - t_end) > 0:
                        tn = t_end
                    t_new[g], h[g] = tn, tn - t[g]
                    h_abs[g] = abs(h[g])
                    nfev[g] += 6
                hr = rows(h)                                  # (idle rows: h = 0, the stages leave them where they are)
                K = [fk]
                for s in range(1, 6):
                    K.append(f([t[g] + _C[s] * h[g] for g in range(G)], ops.rk_combine_rows(x, K, _A[s][:s], hr)))
                x_new, x_new32 = ops.rk_combine_rows(x, K, _B, hr, want64=True)
                f_new = f(t_new, x_new32)
                K.append(f_new)
                err = norms(ops.rk_scaled_sumsq_rows(x, x_new, K, _E, hr, atol, rtol))
                accept = [False] * G
                for g in range(G):
                    if not active[g]:
                        continue
                    e = err[g]
                    if e < 1:
                        factor = _MAX_FACTOR if e == 0 else min(_MAX_FACTOR, _SAFETY * e ** _ERR_EXP)
                        if rejected[g]:
                            factor = min(1.0, factor)
                        h_abs[g] *= factor
                        accept[g], new_step[g] = True, True
                        t[g] = t_new[g]
                        active[g] = (t[g] - t_end) * direction < 0
                    else:
                        h_abs[g] *= max(_MIN_FACTOR, _SAFETY * e ** _ERR_EXP)
                        rejected[g] = True
                if all(accept):
                    x, x32, fk = x_new, x_new32, f_new
                elif any(accept):                             # the accepted rows move on; the others retry from where they are
                    acc_rows = rows(accept)
                    ops.copy_rows(x, x_new, acc_rows)
                    ops.copy_rows(x32, x_new32, acc_rows)
                    fk = ops.copy_rows(fk.clone() if fk is f0 else fk, f_new, acc_rows)
                if shrink and not all(active) and any(active):
                    # rows that reached eps leave: park their end state, shrink every per-row tensor to the rows still integrating
                    if x_done is None:
                        x_done = torch.empty_like(yy, dtype=x32.dtype)
                    gone = [g for g in range(G) if not active[g]]
                    keep = [g for g in range(G) if active[g]]
                    x_done.index_copy_(0, torch.tensor([orig[g] for g in gone], device=yy.device),
                                       x32.index_select(0, torch.tensor(gone, device=yy.device)))
                    for g in gone:
                        nfev_done[orig[g]] = nfev[g]
                    ki = torch.tensor(keep, device=yy.device)
                    x, x32, fk, y_c = x.index_select(0, ki), x32.index_select(0, ki), fk.index_select(0, ki), y_c.index_select(0, ki)
                    if isinstance(cond_c, (list, tuple)):
                        cond_c = [c.index_select(0, ki) for c in cond_c]
                    elif torch.is_tensor(cond_c):
                        cond_c = cond_c.index_select(0, ki)
                    pick = lambda vals: [vals[g] for g in keep]
                    t, h_abs, nfev, new_step, rejected, min_step, active, orig = (pick(t), pick(h_abs), pick(nfev), pick(new_step),
                                                                                    pick(rejected), pick(min_step), pick(active), pick(orig))
                    G = len(keep)
                    groups = [[g] for g in range(G)]
            if x_done is not None:                            # the rows that were still in the batch at the end join the parked ones
                x_done.index_copy_(0, torch.tensor(orig, device=yy.device), x32)
                for g in range(G):
                    nfev_done[orig[g]] = nfev[g]
                x, nfev_rows = x_done, nfev_done
            else:
                x, nfev_rows = x32, rows(nfev)
            ode_sampler.nfev_rows = nfev_rows
            ode_sampler.rows_evaluated = rows_evaluated
            nfe = executed
            if denoise:
                vec_eps = torch.ones(B, device=yy.device) * eps
                _, x = predictor.denoise_fn(x, vec_eps, yy, conditioning=conditioning)
            if inverse_scaler is not None:
                x = inverse_scaler(x)
            return x, nfe

    ode_sampler.nfev_rows = None
    ode_sampler.rows_evaluated = None
    return ode_sampler
