"""Oracle: OUVE / OUVP SDE terms and the predictor-corrector sampler, PyTorch CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  * OUVESDE.sde/_std/prior_sampling and SDE.discretize  (sgmse/sdes.py:200-237, 73-90)
  * OUVPSDE.sde/_mean/_std/prior_sampling                (sgmse/sdes.py:255-310)
  * RSDE.discretize                                      (sgmse/sdes.py:147-157)
  * ReverseDiffusionPredictor / EulerMaruyamaPredictor   (sgmse/sampling/predictors.py:41-69)
  * AnnealedLangevinDynamics / LangevinCorrector         (sgmse/sampling/correctors.py:37-93)
  * the pc_sampler loop                                  (sgmse/sampling/__init__.py:54-66)
with the op order of the reference so that, given the same injected noise,
the sampler algebra is bit-exact against it.

Noise is injected: ``noise`` is an iterator/callable yielding complex tensors in
the reference's draw order (prior, then per step ``n_steps`` corrector draws
followed by one predictor draw).
"""
import math

import numpy as np
import torch


class OUVE:
    def __init__(self, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30):
        self.theta, self.sigma_min, self.sigma_max, self.N = theta, sigma_min, sigma_max, N
        self.logsig = np.log(self.sigma_max / self.sigma_min)        # sdes.py:190
        self.T = 1

    def sde(self, x, t, y):                                          # sdes.py:200-208
        drift = self.theta * (y - x)
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        diffusion = sigma * np.sqrt(2 * self.logsig)
        return drift, diffusion

    def std(self, t):                                                # sdes.py:215-228
        sigma_min, theta, logsig = self.sigma_min, self.theta, self.logsig
        return torch.sqrt(
            (sigma_min ** 2 * torch.exp(-2 * theta * t)
             * (torch.exp(2 * (theta + logsig) * t) - 1) * logsig)
            / (theta + logsig))

    def discretize(self, x, t, y):                                   # sdes.py:86-90
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t, y)
        f = drift * dt
        G = diffusion * torch.sqrt(torch.tensor(dt))
        return f, G

    def prior(self, y, z):                                           # sdes.py:233-237
        std = self.std(torch.ones((y.shape[0],)))
        return y + z * std[:, None, None, None]


class OUVP:
    """OUVPSDE (sdes.py:255-326): dx = 1/2 beta(t) stiffness (y - x) dt + sqrt(beta(t)) dw."""

    def __init__(self, beta_min, beta_max, stiffness=1, N=30):
        self.beta_min, self.beta_max, self.stiffness, self.N = beta_min, beta_max, stiffness, N
        self.T = 1

    def beta(self, t):                                               # sdes.py:290-291
        return self.beta_min + t * (self.beta_max - self.beta_min)

    def sde(self, x, t, y):                                          # sdes.py:293-296
        b = self.beta(t)
        drift = 0.5 * self.stiffness * b.view(-1, *((1,) * (y.ndim - 1))) * (y - x)
        return drift, torch.sqrt(b)

    def mean(self, x0, t, y):                                        # sdes.py:298-301
        b0, b1, s = self.beta_min, self.beta_max, self.stiffness
        fac = torch.exp(-0.25 * s * t * (t * (b1 - b0) + 2 * b0))[:, None, None, None]
        return y + fac * (x0 - y)

    def std(self, t):                                                # sdes.py:303-305
        b0, b1, s = self.beta_min, self.beta_max, self.stiffness
        return (1 - torch.exp(-0.5 * s * t * (t * (b1 - b0) + 2 * b0))) / s

    discretize = OUVE.discretize                                     # SDE.discretize is shared (sdes.py:73-90)

    def prior(self, y, z):                                           # sdes.py:306-310
        std = self.std(torch.ones((y.shape[0],)))
        return y + z * std[:, None, None, None]


def pf_drift(sde, score_fn, x, t, y):
    """RSDE.sde with probability_flow=True (sdes.py:117-145): sde_drift + (-(g^2) score 1/2)."""
    drift, diffusion = sde.sde(x, t, y)
    diffusion = _bc(diffusion, x)
    return drift + (-diffusion ** 2 * score_fn(x, t, y) * 0.5)


def _bc(v, x):
    return v.view(*v.size(), *((1,) * (x.ndim - v.ndim)))


def revdiff_step(sde, score_fn, x, t, y, z):
    """ReverseDiffusionPredictor.update_fn (predictors.py:62-69) + RSDE.discretize (sdes.py:147-157)."""
    f, G = sde.discretize(x, t, y)
    G = _bc(G, x)
    rev_f = f - G ** 2 * score_fn(x, t, y)
    x_mean = x - rev_f
    return x_mean + G * z, x_mean


def euler_maruyama_step(sde, score_fn, x, t, y, z):
    """EulerMaruyamaPredictor.update_fn (predictors.py:46-54) + RSDE.sde/rsde_parts (sdes.py:117-145)."""
    dt = -1.0 / sde.N
    drift, diffusion = sde.sde(x, t, y)
    diffusion = _bc(diffusion, x)
    total_drift = drift + (-diffusion ** 2 * score_fn(x, t, y) * 1.0)
    x_mean = x + total_drift * dt
    return x_mean + diffusion * np.sqrt(-dt) * z, x_mean


def ald_step(sde, score_fn, x, t, y, z, snr):
    """One inner iteration of AnnealedLangevinDynamics.update_fn (correctors.py:76-93)."""
    std = sde.std(t)
    grad = score_fn(x, t, y)
    step_size = _bc((snr * std) ** 2 * 2, x)
    x_mean = x + step_size * grad
    return x_mean + z * torch.sqrt(step_size * 2), x_mean


def langevin_step(sde, score_fn, x, t, y, z, snr):
    """One inner iteration of LangevinCorrector.update_fn (correctors.py:45-61); norms are batch means."""
    grad = score_fn(x, t, y)
    grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
    noise_norm = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
    step_size = ((snr * noise_norm / grad_norm) ** 2 * 2).unsqueeze(0)
    step_size = _bc(step_size, x)
    x_mean = x + step_size * grad
    return x_mean + z * torch.sqrt(step_size * 2), x_mean


def pc_sample(sde, score_fn, y, noise, predictor="reverse_diffusion", corrector="ald",
              corrector_steps=1, snr=0.5, eps=3e-2, denoise=True, trace=None):
    """pc_sampler (sampling/__init__.py:54-66).  ``noise()`` returns the next complex draw
    shaped like y.  Returns (x_result, nfe)."""
    B = y.shape[0]
    xt = sde.prior(y, noise())
    xt_mean = xt
    timesteps = torch.linspace(sde.T, eps, sde.N)
    n_corr = 0 if corrector == "none" else corrector_steps
    for i in range(sde.N):
        vec_t = torch.ones(B) * timesteps[i]
        for _ in range(n_corr):
            if corrector == "ald":
                xt, xt_mean = ald_step(sde, score_fn, xt, vec_t, y, noise(), snr)
            elif corrector == "langevin":
                xt, xt_mean = langevin_step(sde, score_fn, xt, vec_t, y, noise(), snr)
            else:
                raise ValueError(corrector)
        if predictor == "reverse_diffusion":
            xt, xt_mean = revdiff_step(sde, score_fn, xt, vec_t, y, noise())
        elif predictor == "euler_maruyama":
            xt, xt_mean = euler_maruyama_step(sde, score_fn, xt, vec_t, y, noise())
        elif predictor == "none":
            xt_mean = xt
        else:
            raise ValueError(predictor)
        if trace is not None:
            trace.append(xt.clone())
    x_result = xt_mean if (denoise and sde.N) else xt
    return x_result, sde.N * (n_corr + 1)


def complex_randn(shape, gen):
    """torch.randn_like on a complex tensor: each component ~ N(0, 1/2)."""
    return torch.randn(*shape, dtype=torch.complex64, generator=gen)
