#!/usr/bin/env python
"""Does the data decide the speed?  The same conv_pipe launch (256 -> 256 @128x256x16, plain operand) on random, small-magnitude and
all-zero operands: same instructions, different switching activity - a power-limited part clocks higher on the quieter data."""
import sys
import torch
sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
B, H, W, cin, cout = 16, 128, 256, 256, 256
fl = 2 * B * H * W * cout * cin * 9
cases = {"random N(0,1) x, N(0,0.05) w": (torch.randn(B, H, W, cin, generator=g), torch.randn(cout, cin, 3, 3, generator=g) * 0.05),
         "x = 1.0 everywhere": (torch.ones(B, H, W, cin), torch.randn(cout, cin, 3, 3, generator=g) * 0.05),
         "x = 0": (torch.zeros(B, H, W, cin), torch.randn(cout, cin, 3, 3, generator=g) * 0.05),
         "x = 0, w = 0": (torch.zeros(B, H, W, cin), torch.zeros(cout, cin, 3, 3))}
for name, (x, w) in cases.items():
    xd, wd = x.to(dt).to(dev), ops.pack_conv_weight(w.to(dev), dt)
    for _ in range(3):
        y = ops.conv([ops.Seg(xd, wd, 9)], cout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv([ops.Seg(xd, wd, 9)], cout)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:32s} {ms:.3f} ms  {fl / ms / 1e9:6.0f} TF/s")
