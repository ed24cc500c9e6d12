#!/usr/bin/env python
"""Debug probe (GPU): K slices (STORM_SPLITK = 1 / 2 / 4 / 8) for the one-utterance launches of the 64 x 128 and 32 x 64 levels, alternating,
fused GroupNorm operand + statistics as in the network, cold-ish (a 256 MB fill between timed launches evicts the weights from the L2)."""
import sys
import torch
sys.path.insert(0, ".")
from storm_amd import ops, _lib as L  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
lib = L.lib()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for (B, H, W, cin, sc) in ((1, 64, 128, 256, 0), (1, 64, 128, 512, 0), (1, 64, 128, 256, 256), (1, 32, 64, 256, 0), (1, 32, 64, 512, 0), (2, 64, 128, 256, 0), (4, 32, 64, 256, 0)):
    x = torch.randn(B, H, W, cin, generator=g).to(dt).to(dev)
    w = ops.pack_conv_weight((torch.randn(256, cin, 3, 3, generator=g) * 0.03).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * torch.randn(B, cin, generator=g), 0.1 * torch.randn(B, cin, generator=g)).to(dev)
    segs = [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)]
    if sc:
        xs = torch.randn(B, H, W, sc, generator=g).to(dt).to(dev)
        segs.append(ops.Seg(xs, ops.pack_conv_weight((torch.randn(256, sc, 1, 1, generator=g) * 0.03).to(dev), dt), 1))
    kw = dict(bias=torch.randn(256, generator=g).to(dev), gn_partials=True, scale=0.7)
    res = {}
    for S in (1, 2, 4, 8):
        L.check(lib.storm_set_switch(b"STORM_SPLITK", S), "switch")
        for _ in range(3):
            ops.conv(segs, 256, **kw)
    times = {S: [] for S in (1, 2, 4, 8)}
    for _ in range(15):
        for S in (1, 2, 4, 8):
            L.check(lib.storm_set_switch(b"STORM_SPLITK", S), "switch")
            junk.fill_(1)
            e0, e1 = ev(), ev()
            e0.record()
            ops.conv(segs, 256, **kw)
            e1.record()
            times[S].append((e0, e1))
    torch.cuda.synchronize()
    L.check(lib.storm_set_switch(b"STORM_SPLITK", 0), "switch")
    med = {S: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] for S, v in times.items()}
    print(f"B={B} {H}x{W} cin={cin} shortcut={sc}: " + " | ".join(f"S={S}: {1e3 * m:6.1f} us" for S, m in med.items()), flush=True)
