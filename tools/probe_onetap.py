#!/usr/bin/env python
"""What do the one-tap chunks (fused 1x1 shortcut) of conv_pipe cost, and which part of a one-tap phase is it?

The same 256 -> 256 3x3 layer at a bench shape with a fused GroupNorm operand + statistics as in the network, (a) plain, (b) with a
one-tap shortcut of `--sc` channels read from a tensor of that many channels, (c) the shortcut read from a SMALL tensor (one image
broadcast: bstride 0 is not expressible, so a batch of ONE image repeated is emulated by B = 1 ... instead: the shortcut source is
L2-resident when --small), each under the work-skipping instantiations of the profiling library (STORM_LIB = libstorm_hip_prof.so):
ABL 0 full, 128 no patch DMA / transform, 8 no weight DMA, 16 no fragment reads, 32 no MFMAs.

    STORM_LIB=storm_amd/csrc/libstorm_hip_prof.so python tools/probe_onetap.py [--H 256 --W 512 --sc 256 --reps 20]
"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
from storm_amd import _lib as L  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--reps", type=int, default=20)
p.add_argument("--B", type=int, default=16)
p.add_argument("--H", type=int, default=256)
p.add_argument("--W", type=int, default=512)
p.add_argument("--cin", type=int, default=256)
p.add_argument("--sc", type=int, default=256)
p.add_argument("--abl", default="0,128,8,16,32")
args = p.parse_args()
dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
B, H, W, cin, cout, sc = args.B, args.H, args.W, args.cin, 256, args.sc
x = rnd(B, H, W, cin).to(dt).to(dev)
w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
xs = rnd(B, H, W, sc).to(dt).to(dev)
ws = ops.pack_conv_weight((rnd(cout, sc, 1, 1) * 0.05).to(dev), dt)
kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=True, scale=0.7)
lib = L.lib()
L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", 3), "variant")
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
print(f"256 -> 256 @ {H} x {W} x {B}, fused GroupNorm operand + statistics; shortcut = 1x1 over {sc} channels; lib {L.LIB_PATH.split('/')[-1]}")
for abl in [int(v) for v in args.abl.split(",")]:
    L.check(lib.storm_set_switch(b"STORM_CONV_ABLATE", abl), "ablate")
    res = {}
    for name, segs in (("plain", [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)]),
                       ("shortcut", [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True), ops.Seg(xs, ws, 1)])):
        for _ in range(3):
            ops.conv(segs, cout, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(args.reps):
            ops.conv(segs, cout, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / args.reps
    k9, k1 = cin * 9, sc
    ideal = res["plain"] * (k9 + k1) / k9
    tiles_per_wg = B * (H // 8) * (W // 32) / 256
    print(f"ABL {abl:4d}: plain {res['plain']:.3f} ms | with shortcut {res['shortcut']:.3f} ms | the shortcut's {2 * sc // 64} phases cost "
          f"{res['shortcut'] - res['plain']:.3f} ms = {(res['shortcut'] - res['plain']) / (ideal - res['plain']):.2f} x their share of K "
          f"({1e3 * (res['shortcut'] - res['plain']) / tiles_per_wg / (2 * sc // 64):.2f} us per one-tap phase and tile; a nine-tap phase: "
          f"{1e3 * res['plain'] / tiles_per_wg / (2 * 9 * cin // 64):.2f} us)", flush=True)
L.check(lib.storm_set_switch(b"STORM_CONV_ABLATE", 0), "ablate")
L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", -1), "variant")
