"""Predictors of the PC sampler — drop-in for sgmse/sampling/predictors.py.  With an OUVESDE (or an SDE that
gives per-row coefficients: OUVPSDE) the update is one fused HIP kernel (csrc/sde.hip); the score comes from ``score_fn`` (NCSN++ engine)."""
import abc

from .. import ops
from ..sdes import OUVESDE
from ..util.registry import Registry
from .noise import NoiseSource

PredictorRegistry = Registry("Predictor")


def _score(score_fn, x, t, args, kwargs):
    if kwargs.get("conditioning") is not None:
        return score_fn(x, t, score_conditioning=kwargs["conditioning"], sde_input=args[0])
    return score_fn(x, t, *args)


class Predictor(abc.ABC):
    def __init__(self, sde, score_fn, probability_flow=False, noise=None):
        super().__init__()
        if not isinstance(sde, OUVESDE) and not hasattr(sde, "drift_rows"):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not supported by the HIP engine: an SDE either is an OUVESDE or "
                                      "gives its per-row coefficients (drift_rows(t), diffusion(t)) as OUVPSDE does.")
        self.sde = sde
        self.rsde = sde.reverse(score_fn)           # the reference ignores probability_flow here (predictors.py:18)
        self.score_fn = score_fn
        self.probability_flow = probability_flow
        self.noise = noise if noise is not None else NoiseSource()

    @abc.abstractmethod
    def update_fn(self, x, t, *args, **kwargs):
        pass

    def _step(self, kind, x, t, args, kwargs, noise_free=False):
        y = args[0]
        score = _score(self.score_fn, x, t, args, kwargs)
        z, seed, off = (None, 0, 0) if noise_free else self.noise.next(x)
        if not isinstance(self.sde, OUVESDE):      # coefficient-table form: a(t_b), g(t_b) from the SDE's own fp32 expressions
            return ops.sde_predictor_step_rows(self.sde, x.contiguous(), score.contiguous(), y.contiguous(), t.contiguous(),
                                               kind=kind, z=z, noise_free=noise_free, seed=seed, offset=off)
        return ops.ouve_predictor_step(self.sde, x.contiguous(), score.contiguous(), y.contiguous(), t.contiguous(),
                                       kind=kind, z=z, noise_free=noise_free, seed=seed, offset=off)


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, t, *args, **kwargs):
        return self._step(1, x, t, args, kwargs)


@PredictorRegistry.register("reverse_diffusion")
class ReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, t, *args, **kwargs):
        return self._step(0, x, t, args, kwargs)

    def denoise_fn(self, x, t, *args, **kwargs):
        """noise-free predictor step (the ODE sampler's final denoise, sampling/__init__.py:97-100)"""
        return self._step(0, x, t, args, kwargs, noise_free=True)


@PredictorRegistry.register("none")
class NonePredictor(Predictor):
    def __init__(self, *args, **kwargs):
        pass

    def update_fn(self, x, t, *args, **kwargs):
        return x, x
