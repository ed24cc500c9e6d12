"""Predictor-corrector (and ODE) samplers — drop-in for sgmse/sampling/__init__.py.

``get_pc_sampler(...)`` keeps the reference signature and returns ``fn() -> (x, nfe)``.  Extra
keyword-only knobs: ``noise_fn`` (inject the draws, for parity runs), ``seed`` (in-kernel Philox
stream for production runs), ``langevin_per_row`` / ``langevin_group`` (how the Langevin corrector's batch-coupled
step size is formed, see LangevinCorrector).  All update functions work IN PLACE on the state they are handed."""
import torch

from .correctors import Corrector, CorrectorRegistry
from .noise import NoiseSource
from .predictors import Predictor, PredictorRegistry, ReverseDiffusionPredictor

__all__ = ["PredictorRegistry", "CorrectorRegistry", "Predictor", "Corrector", "get_pc_sampler", "get_ode_sampler",
           "NoiseSource"]


def get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, denoise=True, eps=3e-2, snr=0.1,
                   corrector_steps=1, probability_flow: bool = False, conditioning=None, intermediate=False,
                   noise_fn=None, seed=None, langevin_per_row=False, langevin_group=None, **kwargs):
    """PC sampler (sampling/__init__.py:27-68): prior draw, then N x (corrector, predictor) on the
    time grid linspace(T, eps, N); returns the last predictor mean and nfe = N (corrector_steps + 1)."""
    predictor_cls = PredictorRegistry.get_by_name(predictor_name)
    corrector_cls = CorrectorRegistry.get_by_name(corrector_name)
    noise = NoiseSource(seed=seed, noise_fn=noise_fn)
    if predictor_name == "none":
        predictor = predictor_cls()
    else:
        predictor = predictor_cls(sde, score_fn, probability_flow=probability_flow, noise=noise)
    if corrector_name == "none":
        corrector = corrector_cls()
    else:
        extra = dict(per_row=langevin_per_row, group=langevin_group) if corrector_name == "langevin" else {}
        corrector = corrector_cls(sde, score_fn, snr=snr, n_steps=corrector_steps, noise=noise, **extra)

    def pc_sampler():
        with torch.no_grad():
            yy = y.contiguous()
            z, sd, off = noise.next(yy)
            xt = sde.prior_sampling(yy.shape, yy, z=z, seed=sd, offset=off)
            xt_mean = xt
            timesteps = torch.linspace(sde.T, eps, sde.N, device=yy.device)
            vec_ts = (torch.ones(sde.N, yy.shape[0], device=yy.device) * timesteps[:, None]).contiguous()
            for i in range(sde.N):
                vec_t = vec_ts[i]
                xt, xt_mean = corrector.update_fn(xt, vec_t, yy, conditioning=conditioning)
                xt, xt_mean = predictor.update_fn(xt, vec_t, yy, conditioning=conditioning)
            x_result = xt_mean if (denoise and sde.N) else xt
            ns = sde.N * (corrector.n_steps + 1)
            return x_result, ns

    return pc_sampler


def get_ode_sampler(*args, **kwargs):
    from .ode import get_ode_sampler as impl
    return impl(*args, **kwargs)
