#!/usr/bin/env python
"""Per-tile wave timeline of conv_pipe128.hip (profiling instantiation, STORM_CONV_ABLATE=64): six s_memtime stamps per tile and wave -
tile start, main loop begins, main loop done, hand-over done, epilogue stores issued, statistics written - over a persistent
workgroup's whole tile walk.

  python tools/pipe128_trace.py [--cin 128 --H 256 --W 512] [--gn]
"""
import argparse
import os
import sys

import numpy as np
import torch

os.environ["STORM_CONV_ABLATE"] = "64"
os.environ["STORM_CONV_VARIANT"] = "4"
os.environ.setdefault("STORM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "storm_amd", "csrc", "libstorm_hip_prof.so"))
sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--B", type=int, default=16)
p.add_argument("--cin", type=int, default=128)
p.add_argument("--H", type=int, default=256)
p.add_argument("--W", type=int, default=512)
p.add_argument("--gn", action="store_true", help="fused GroupNorm + SiLU operand")
args = p.parse_args()
dev = torch.device("cuda:0")
SLOTS, NW, cout = 512, 8, 128
trace = torch.zeros(256 * NW * SLOTS, dtype=torch.int64, device=dev)
os.environ["STORM_CONV_TRACE_PTR"] = hex(trace.data_ptr())
g = torch.Generator().manual_seed(0)
x = torch.randn(args.B, args.H, args.W, args.cin, generator=g).to(torch.bfloat16).to(dev)
w = ops.pack_conv_weight((torch.randn(cout, args.cin, 3, 3, generator=g) * 0.05).to(dev), torch.bfloat16)
b = torch.randn(cout, generator=g).to(dev)
ss = ops.pack_gn_ss(1 + 0.1 * torch.randn(args.B, args.cin, generator=g), 0.1 * torch.randn(args.B, args.cin, generator=g)).to(dev) if args.gn else None
segs = [ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)]
print(ops.conv_kernel_name(segs, cout, bias=b))
for _ in range(3):
    y, part = ops.conv(segs, cout, bias=b, gn_partials=True)
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
y, part = ops.conv(segs, cout, bias=b, gn_partials=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy().reshape(256, NW, SLOTS).astype(np.int64)
st = t[:, :, 8:]
ntile = int((st[0, 0] > 0).sum()) // 6
st = st[:, :, :6 * ntile].reshape(256, NW, ntile, 6)
span = np.median(st[:, :, -1, 5].max(axis=1) - st[:, :, 0, 0].min(axis=1))
tick_us = ms * 1e3 / span
print(f"{ms:.3f} ms, {ntile} tiles per workgroup, {tick_us * 1e3:.3f} ns per tick (calibrated on the launch)")
names = ["part B: chunk 1 issue, chunk 0 wait + transform", "main loop", "hand-over: drain, barrier, next tile's first loads", "epilogue: transpose + stores",
         "statistics", "-> next tile start (barrier)"]
for gname, sel in (("leading waves 0-3", slice(0, 4)), ("lagging waves 4-7", slice(4, 8))):
    print(gname)
    d = [st[:, sel, :, i + 1] - st[:, sel, :, i] for i in range(5)]
    d.append(st[:, sel, 1:, 0] - st[:, sel, :-1, 5])
    tot = 0.0
    for nm, v in zip(names, d):
        m = float(v.mean()) * tick_us
        tot += m
        print(f"   {nm:52s} {m:7.2f} us   (p10 {np.percentile(v, 10) * tick_us:6.2f}, p90 {np.percentile(v, 90) * tick_us:6.2f})")
    print(f"   {'tile period':52s} {tot:7.2f} us")
