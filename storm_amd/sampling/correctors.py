"""Correctors of the PC sampler — drop-in for sgmse/sampling/correctors.py (fused HIP updates)."""
import abc

from .. import ops
from ..sdes import OUVESDE
from ..util.registry import Registry
from .noise import NoiseSource
from .predictors import _score

CorrectorRegistry = Registry("Corrector")


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps, noise=None):
        super().__init__()
        self.sde = sde
        self.score_fn = score_fn
        self.snr = snr
        self.n_steps = n_steps
        self.noise = noise if noise is not None else NoiseSource()

    @abc.abstractmethod
    def update_fn(self, x, t, *args, **kwargs):
        pass


@CorrectorRegistry.register(name="langevin")
class LangevinCorrector(Corrector):
    """step = 2 (snr * mean_b ||z_b|| / mean_b ||score_b||)^2 — batch-coupled (correctors.py:45-61).

    The coupling is over whatever batch the sampler was given.  `per_row=True` gives every row its own step size
    (= B independent batch-1 runs, which is what the reference CLI computes per file: used by enhance_batch);
    `group` = a torch.distributed process group: the means are taken over the batches of ALL its ranks (a 2-float
    all-reduce per step), so a sharded run reproduces the unsharded batch-mean exactly.  Like every update_fn here it
    updates x IN PLACE and returns (x, x_mean)."""

    def __init__(self, sde, score_fn, snr, n_steps, noise=None, per_row=False, group=None):
        super().__init__(sde, score_fn, snr, n_steps, noise)
        self.per_row, self.group = per_row, group

    def update_fn(self, x, t, *args, **kwargs):
        x_mean = x
        for _ in range(self.n_steps):
            grad = _score(self.score_fn, x, t, args, kwargs)
            z, seed, off = self.noise.next(x)
            if z is None:
                z = ops.complex_randn(x.shape, x.device, seed, off)
            x, x_mean = ops.langevin_step(x.contiguous(), grad.contiguous(), z.contiguous(), self.snr, per_row=self.per_row,
                                          group=self.group)
        return x, x_mean


@CorrectorRegistry.register(name="ald")
class AnnealedLangevinDynamics(Corrector):
    """step = 2 (snr * std(t))^2 (correctors.py:64-93)."""

    def __init__(self, sde, score_fn, snr, n_steps, noise=None):
        super().__init__(sde, score_fn, snr, n_steps, noise)
        if not isinstance(sde, OUVESDE):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t, *args, **kwargs):
        x_mean = x
        for _ in range(self.n_steps):
            grad = _score(self.score_fn, x, t, args, kwargs)
            z, seed, off = self.noise.next(x)
            x, x_mean = ops.ouve_ald_step(self.sde, x.contiguous(), grad.contiguous(), t.contiguous(), self.snr, z=z,
                                          seed=seed, offset=off)
        return x, x_mean


@CorrectorRegistry.register(name="none")
class NoneCorrector(Corrector):
    def __init__(self, *args, **kwargs):
        self.snr = 0
        self.n_steps = 0

    def update_fn(self, x, t, *args, **kwargs):
        return x, x
