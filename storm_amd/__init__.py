"""storm_amd — MI355X-native (gfx950) engine for the StoRM / SGMSE reverse-SDE sampling hot path.

Host code is Python on PyTorch-ROCm and keeps the reference's surface (ScoreModel.enhance,
get_pc_sampler, registries); all compute is hand-written HIP behind the C ABI in
include/storm_hip.h (storm_amd/csrc -> libstorm_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"


def set_batch_invariant(on: bool = True) -> None:
    """Serving with dynamic batching: make an utterance's result independent - bit for bit - of what it is batched with.

    By default a few launch decisions look at the whole call (which conv tile fills the chip in the fewest rounds, whether a one-utterance
    call splits K or the attention's key loop); the kernels they choose between sum in different orders, so a row agrees with itself across
    batch sizes to the rounding of its 16-bit activations, not bit for bit.  With this switch (library switch STORM_BATCH_INVARIANT, also
    read from the environment at load) every such decision is taken per image.  Costs the batch-aware selections."""
    from . import _lib
    _lib.check(_lib.lib().storm_set_switch(b"STORM_BATCH_INVARIANT", int(bool(on))), "storm_set_switch")
