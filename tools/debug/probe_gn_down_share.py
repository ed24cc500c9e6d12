#!/usr/bin/env python
"""Debug probe (GPU): gn_apply_down barrier-free (STORM_GN_DOWN_SHARE=1) against the LDS-shared activation (=2), alternating, N(0, 1) data,
at the shapes of NCSN++'s down path; checks that both give the same bits."""
import sys
import torch
sys.path.insert(0, ".")
from storm_amd import ops, _lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
g = torch.Generator().manual_seed(0)
dts = [torch.bfloat16] + ([torch.float16] if "--f16" in sys.argv else [])
for dt in dts:
    for B in (16, 8, 4, 2, 1):
        for (H, W, C) in ((256, 512, 128), (128, 256, 256), (64, 128, 256)):
            x = torch.randn(B, H, W, C, generator=g).to(dt).to(dev)
            st = ops.gn_stats(x)
            gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
            outs, times = {}, {1: [], 2: []}
            for sh in (1, 2):
                L.check(lib.storm_set_switch(b"STORM_GN_DOWN_SHARE", sh), "switch")
                for _ in range(3):
                    outs[sh] = ops.gn_apply(x, st, gam, bet, resample=2)
            for _ in range(24):
                for sh in (1, 2):
                    L.check(lib.storm_set_switch(b"STORM_GN_DOWN_SHARE", sh), "switch")
                    e0, e1 = ev(), ev()
                    e0.record()
                    ops.gn_apply(x, st, gam, bet, resample=2)
                    e1.record()
                    times[sh].append((e0, e1))
            torch.cuda.synchronize()
            L.check(lib.storm_set_switch(b"STORM_GN_DOWN_SHARE", 0), "switch")
            same = all(torch.equal(outs[1][k], outs[2][k]) for k in range(2))
            med = {c: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] for c, v in times.items()}
            nbytes = B * H * W * C * 2 * 1.5
            print(f"{str(dt)[6:]:8s} B={B:2d} {H:3d}x{W:<4d} C={C}: barrier-free {1e3 * med[1]:7.1f} us ({nbytes / med[1] / 1e9:4.2f} TB/s) | shared {1e3 * med[2]:7.1f} us ({nbytes / med[2] / 1e9:4.2f} TB/s) | same bits: {same}", flush=True)
