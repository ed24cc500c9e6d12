"""SDE definitions for the reverse sampler — drop-in for sgmse/sdes.py (OUVESDE, OUVPSDE, SDERegistry).

The state-sized arithmetic of the hot path (prior sampling, predictor / corrector updates) runs in
the fused HIP kernels of storm_amd/csrc/sde.hip through ``storm_amd.sampling``; the tensor-level
methods here (``sde``, ``marginal_prob``, ``discretize``, ``reverse``) keep the reference's API
for other callers (e.g. the ODE sampler's drift) and are plain tensor expressions on whatever
device the inputs live.
"""
import abc
import warnings

import numpy as np
import torch

from . import ops
from .util.registry import Registry

SDERegistry = Registry("SDE")

def _bc(v, x):
    return v.view(*v.size(), *((1,) * (x.ndim - v.ndim))) if v.ndim < x.ndim else v


class SDE(abc.ABC):
    """Abstract SDE (sdes.py:20-163)."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        pass

    @abc.abstractmethod
    def sde(self, x, t, *args):
        pass

    @abc.abstractmethod
    def marginal_prob(self, x, t, *args):
        pass

    @abc.abstractmethod
    def prior_sampling(self, shape, *args):
        pass

    @abc.abstractmethod
    def copy(self):
        pass

    def discretize(self, x, t, *args):
        """Euler-Maruyama discretisation x_{i+1} = x_i + f_i + G_i z_i with dt = 1/N (sdes.py:73-90)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t, *args)
        f = drift * dt
        G = diffusion * torch.sqrt(torch.tensor(dt, device=t.device))
        return f, G

    def reverse(oself, score_model, probability_flow=False, diffusion_power_gradient=None):
        """Reverse-time SDE/ODE (sdes.py:92-159).  diffusion_power_gradient(x, t): subtracted from the total drift when the
        diffusion depends on the state (sdes.py:98-99, 137-138); None for both registered SDEs."""
        N, T, sde_fn, discretize_fn = oself.N, oself.T, oself.sde, oself.discretize

        class RSDE(oself.__class__):
            def __init__(self):
                self.N = N
                self.probability_flow = probability_flow
                self.diffusion_power_gradient = diffusion_power_gradient

            @property
            def T(self):
                return T

            def _score(self, x, t, args, kwargs):
                if kwargs.get("conditioning") is not None:
                    return score_model(x, t, score_conditioning=kwargs["conditioning"], sde_input=args[0])
                return score_model(x, t, *args)

            def sde(self, x, t, *args, **kwargs):
                parts = self.rsde_parts(x, t, *args, **kwargs)
                return parts["total_drift"], parts["diffusion"]

            def rsde_parts(self, x, t, *args, **kwargs):
                sde_drift, sde_diffusion = sde_fn(x, t, *args)
                score = self._score(x, t, args, kwargs)
                sde_diffusion = _bc(sde_diffusion, x)
                score_drift = -sde_diffusion ** 2 * score * (0.5 if self.probability_flow else 1.)
                diffusion = torch.zeros_like(sde_diffusion) if self.probability_flow else sde_diffusion
                total_drift = sde_drift + score_drift
                if diffusion_power_gradient is not None:
                    total_drift = total_drift - diffusion_power_gradient(x, t)
                return {"total_drift": total_drift, "diffusion": diffusion, "sde_drift": sde_drift,
                        "sde_diffusion": sde_diffusion, "score_drift": score_drift, "score": score}

            def discretize(self, x, t, *args, **kwargs):
                f, G = discretize_fn(x, t, *args)
                G = _bc(G, x)
                rev_f = f - G ** 2 * self._score(x, t, args, kwargs) * (0.5 if self.probability_flow else 1.)
                rev_G = torch.zeros_like(G) if self.probability_flow else G
                return rev_f, rev_G

        return RSDE()


@SDERegistry.register("ouve")
class OUVESDE(SDE):
    """Ornstein-Uhlenbeck variance-exploding SDE  dx = theta (y - x) dt + sigma(t) dw  (sdes.py:166-252)."""

    def __init__(self, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=1000, **ignored_kwargs):
        super().__init__(N)
        self.theta, self.sigma_min, self.sigma_max = theta, sigma_min, sigma_max
        self.logsig = np.log(self.sigma_max / self.sigma_min)
        self.N = N

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N)

    @property
    def T(self):
        return 1

    def diffusion(self, t):
        """g(t) = sigma_min (sigma_max / sigma_min)^t sqrt(2 logsig) in the reference's torch operations (sdes.py:203-207)"""
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return sigma * np.sqrt(2 * self.logsig)

    def sde(self, x, t, y):
        return self.theta * (y - x), self.diffusion(t)

    def _mean(self, x0, t, y):
        e = torch.exp(-self.theta * t)[:, None, None, None]
        return e * x0 + (1 - e) * y

    def _std(self, t, **kwargs):
        sigma_min, theta, logsig = self.sigma_min, self.theta, self.logsig
        return torch.sqrt((sigma_min ** 2 * torch.exp(-2 * theta * t) * (torch.exp(2 * (theta + logsig) * t) - 1) * logsig)
                          / (theta + logsig))

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)

    def prior_sampling(self, shape, y, z=None, seed=0, offset=0):
        """y + z * std(1)  (sdes.py:233-237) — fused HIP kernel; z=None draws in-kernel (Philox)."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        return ops.ouve_prior(self, y.contiguous(), z=z, seed=seed, offset=offset)

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for OU SDE not yet implemented!")

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--sde-n", type=int, default=1000)
        parser.add_argument("--theta", type=float, default=1.5)
        parser.add_argument("--sigma-min", type=float, default=0.05)
        parser.add_argument("--sigma-max", type=float, default=0.5)
        return parser


@SDERegistry.register("ouvp")
class OUVPSDE(SDE):
    """Ornstein-Uhlenbeck variance-preserving SDE  dx = 1/2 beta(t) stiffness (y - x) dt + sqrt(beta(t)) dw,
    beta(t) = beta_min + t (beta_max - beta_min)  (sdes.py:255-326).

    The samplers reach it through `drift_rows(t)` / `diffusion(t)`: the per-row coefficients a(t_b), g(t_b) in the reference's
    own fp32 expressions, handed to the coefficient-table kernels (storm_sde_*_rows).  As upstream, the `ald` corrector
    rejects it (correctors.py:69); `langevin` / `none` and both predictors and the ODE sampler take it."""

    def __init__(self, beta_min, beta_max, stiffness=1, N=1000, **ignored_kwargs):
        super().__init__(N)
        self.beta_min, self.beta_max, self.stiffness = beta_min, beta_max, stiffness
        self.N = N

    def copy(self):
        return OUVPSDE(self.beta_min, self.beta_max, self.stiffness, N=self.N)

    @property
    def T(self):
        return 1

    def _beta(self, t):
        return self.beta_min + t * (self.beta_max - self.beta_min)

    def drift_rows(self, t):
        """a(t): the drift is a(t) (y - x)  (sdes.py:294)"""
        return 0.5 * self.stiffness * self._beta(t)

    def diffusion(self, t):
        return torch.sqrt(self._beta(t))

    def sde(self, x, t, y):
        return _bc(self.drift_rows(t), y) * (y - x), self.diffusion(t)

    def _mean(self, x0, t, y):
        b0, b1, s = self.beta_min, self.beta_max, self.stiffness
        x0y_fac = torch.exp(-0.25 * s * t * (t * (b1 - b0) + 2 * b0))[:, None, None, None]
        return y + x0y_fac * (x0 - y)

    def _std(self, t, **kwargs):
        b0, b1, s = self.beta_min, self.beta_max, self.stiffness
        return (1 - torch.exp(-0.5 * s * t * (t * (b1 - b0) + 2 * b0))) / s

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)

    def prior_sampling(self, shape, y, z=None, seed=0, offset=0):
        """y + z * std(1)  (sdes.py:306-310) — fused HIP kernel; z=None draws in-kernel (Philox)."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        std = self._std(torch.ones((y.shape[0],), device=y.device))
        return ops.sde_prior_rows(y.contiguous(), std, z=z, seed=seed, offset=offset)

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for OU SDE not yet implemented!")

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--sde-n", type=int, default=1000)
        parser.add_argument("--beta-min", type=float, required=True)
        parser.add_argument("--beta-max", type=float, required=True)
        parser.add_argument("--stiffness", type=float, default=1)
        return parser
