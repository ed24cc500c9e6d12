#!/usr/bin/env python
"""3x3 layers with very few pixel tiles (the deep levels of ncsnpplarge at configs[3]: 8 x (64 x 256 ... 4 x 16) pixels, 256 -> 256 and
512 -> 256): which conv_igemm / conv_pipe variant serves them best (STORM_CONV_VARIANT through the switch hook)."""
import sys

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
from storm_amd import _lib as L  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
for (B, H, W, cin) in [(8, 64, 256, 256), (8, 32, 128, 256), (8, 32, 128, 512), (8, 16, 64, 256), (8, 16, 64, 512), (8, 8, 32, 256), (8, 8, 32, 512), (8, 4, 16, 256),
                       (16, 32, 64, 256), (16, 32, 64, 512)]:
    cout = 256
    x = rnd(B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    segs = [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)]
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=True, scale=0.7)
    fl = 2 * B * H * W * cout * cin * 9
    line = f"{B}x{H}x{W} {cin}->{cout}"
    for variant in (-1, 7, 9, 10):
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", variant), "set")
        try:
            for _ in range(3):
                y, part = ops.conv(segs, cout, **kw)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(20):
                y, part = ops.conv(segs, cout, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f" | v{variant} {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF"
        except Exception as e:  # noqa: BLE001
            line += f" | v{variant} n/a"
    L.lib().storm_set_switch(b"STORM_CONV_VARIANT", -1)
    print(line, flush=True)
