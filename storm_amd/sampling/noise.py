"""Noise source shared by the prior / predictor / corrector updates of one sampler run.

Default: counter-based Philox noise generated inside the update kernels — ``next`` only hands out
(seed, offset) pairs, so no noise tensor is ever materialised.  Parity runs inject the reference's
draws through ``noise_fn`` (draw order: prior, then per step the corrector draws followed by the
predictor draw — sampling/__init__.py:57-63)."""
import torch


class NoiseSource:
    def __init__(self, seed=None, noise_fn=None):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.seed, self.offset, self.noise_fn = int(seed), 0, noise_fn

    def next(self, like):
        """returns (z or None, seed, offset) for one complex draw shaped like `like`"""
        if self.noise_fn is not None:
            z = self.noise_fn()
            if z.shape != like.shape:
                raise ValueError(f"injected noise has shape {tuple(z.shape)}, expected {tuple(like.shape)}")
            return z.to(device=like.device, dtype=torch.complex64).contiguous(), 0, 0
        self.offset += 1
        return None, self.seed, self.offset
