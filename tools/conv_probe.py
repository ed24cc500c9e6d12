#!/usr/bin/env python
"""Micro-probe: run a few representative launches of the hot kernels (for rocprofv3 --pmc passes)."""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--reps", type=int, default=3)
p.add_argument("--B", type=int, default=16)
args = p.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(*s, generator=g)


cases = [("c256_128x256", 256, 256, 128, 256), ("c128_256x512", 128, 128, 256, 512), ("c512_64x128", 512, 256, 64, 128)]
import os  # noqa: E402
if os.environ.get("PROBE_SMALL"):       # the low-resolution layers: fewer tiles than CUs x 2
    cases = [("c256_32x64", 256, 256, 32, 64), ("c512_32x64", 512, 256, 32, 64), ("c256_64x128", 256, 256, 64, 128)]
if os.environ.get("PROBE_NARROW"):      # the output-pyramid convolutions (4 output channels): HBM-bound
    cases = [("c128to4_256x512", 128, 4, 256, 512), ("c256to4_128x256", 256, 4, 128, 256), ("c8to128_256x512", 8, 128, 256, 512)]
if os.environ.get("PROBE_SHORTCUT"):    # Conv_1 of a BigGAN block: 3x3 over h + fused 1x1 shortcut over x (two-phase K-chunks)
    cases = []
if os.environ.get("PROBE_KSWEEP"):      # fixed output tile work, growing K: separates per-tile from per-phase cost
    cases = [(f"k{c}_256_128x256", c, 256, 128, 256) for c in (64, 128, 256, 512)] + [(f"k{c}_128_256x512", c, 128, 256, 512) for c in (32, 64, 128, 256)]
if os.environ.get("PROBE_SHORTCUT"):
    for name, cin, cout, H, W in [("sc256_128x256", 256, 256, 128, 256), ("sc256_256x512", 256, 256, 256, 512)]:
        x = rnd(args.B, H, W, cin).to(dt).to(dev); xs = rnd(args.B, H, W, cin).to(dt).to(dev)
        w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
        ws = ops.pack_conv_weight((rnd(cout, cin, 1, 1) * 0.05).to(dev), dt)
        b = rnd(cout).to(dev)
        _, part = ops.conv([ops.Seg(x, ops.pack_conv_weight(torch.eye(cin).reshape(cin, cin, 1, 1).to(dev), dt), 1)], cin, gn_partials=True)
        _, ss = ops.gn_finalize(part, gamma=(1 + 0.1 * rnd(cin)).to(dev), beta=(0.1 * rnd(cin)).to(dev), count=H * W)
        for tag, segs, kw in [("plain", [ops.Seg(x, w, 9), ops.Seg(xs, ws, 1)], {}),
                              ("gn", [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True), ops.Seg(xs, ws, 1)], {}),
                              ("gn+stats", [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True), ops.Seg(xs, ws, 1)], dict(gn_partials=True, scale=0.7)),
                              ("gn+stats, no shortcut", [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], dict(gn_partials=True, scale=0.7))]:
            for _ in range(args.reps):
                y = ops.conv(segs, cout, bias=b, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                y = ops.conv(segs, cout, bias=b, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            fl = 2 * args.B * H * W * cout * cin * (9 + (len(segs) - 1))
            print(f"{name} [{tag}]: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF")
for name, cin, cout, H, W in cases:
    x = rnd(args.B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    b = rnd(cout).to(dev)
    for _ in range(args.reps):
        y = ops.conv([ops.Seg(x, w, 9)], cout, bias=b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        y = ops.conv([ops.Seg(x, w, 9)], cout, bias=b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.reps * 1e3
    fl = 2 * args.B * H * W * cout * cin * 9
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF")
    # GN on the conv output
    gam, bet = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    for _ in range(args.reps):
        st = ops.gn_stats(y)
        a = ops.gn_apply(y, st, gam, bet)
    torch.cuda.synchronize()
