"""storm_amd — MI355X-native (gfx950) engine for the StoRM / SGMSE reverse-SDE sampling hot path.

Host code is Python on PyTorch-ROCm and keeps the reference's surface (ScoreModel.enhance,
get_pc_sampler, registries); all compute is hand-written HIP behind the C ABI in
include/storm_hip.h (storm_amd/csrc -> libstorm_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
