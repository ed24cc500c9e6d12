#!/usr/bin/env python
"""Debug probe (GPU): what bounds the GroupNorm + FIR kernels - time them with and without the SiLU (the `silu` flag of storm_gn_apply is a
compile-time variant of the kernels), at the network's shapes, N(0, 1) data, alternating."""
import sys
import torch
sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
g = torch.Generator().manual_seed(0)
for resample, name in ((2, "down"), (1, "up"), (0, "plain")):
    for (B, H, W, C) in ((16, 256, 512, 128), (16, 128, 256, 256)) if resample != 1 else ((16, 128, 256, 256), (16, 64, 128, 256)):
        x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(dev)
        st = ops.gn_stats(x)
        gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        times = {True: [], False: []}
        for silu in (True, False):
            for _ in range(3):
                ops.gn_apply(x, st, gam, bet, silu=silu, resample=resample)
        for _ in range(20):
            for silu in (True, False):
                e0, e1 = ev(), ev()
                e0.record()
                ops.gn_apply(x, st, gam, bet, silu=silu, resample=resample)
                e1.record()
                times[silu].append((e0, e1))
        torch.cuda.synchronize()
        med = {c: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] for c, v in times.items()}
        print(f"{name:5s} {B} x {H} x {W} x {C}: with SiLU {1e3 * med[True]:7.1f} us | without {1e3 * med[False]:7.1f} us", flush=True)
