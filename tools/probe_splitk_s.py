#!/usr/bin/env python
"""Few-tile 3x3 layers: the unsplit 128-cout tile against the split of K into 2 / 4 / 8 slices (STORM_SPLITK through the switch hook)."""
import sys

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
from storm_amd import _lib as L  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
for (B, H, W, cin) in [(8, 16, 64, 256), (8, 16, 64, 512), (8, 8, 32, 256), (8, 8, 32, 512), (8, 4, 16, 256), (8, 4, 16, 512), (1, 32, 64, 256), (1, 16, 32, 512)]:
    cout = 256
    x = rnd(B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    segs = [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)]
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=True, scale=0.7, skip=rnd(B, H, W, cout).to(dt).to(dev))
    line = f"{B}x{H}x{W} {cin}->{cout}"
    for S in (1, 2, 4, 8, 0):
        L.check(L.lib().storm_set_switch(b"STORM_SPLITK", S), "set")
        for _ in range(5):
            y, part = ops.conv(segs, cout, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(50):
            y, part = ops.conv(segs, cout, **kw)
        e1.record()
        torch.cuda.synchronize()
        line += f" | S={S if S else 'auto'} {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us"
    L.lib().storm_set_switch(b"STORM_SPLITK", 0)
    print(line, flush=True)
