#!/usr/bin/env python
"""Per-tile wave timeline of conv_duo.hip (profiling instantiation, STORM_CONV_ABLATE=64 [+1 no patch DMA, +2 no weight DMA, +4 no transform]): six s_memtime stamps per tile and wave -
tile start, main loop begins, main loop done, hand-over done, epilogue stores issued, statistics written - over a persistent
workgroup's whole tile walk, and three stamps per phase (stream done, counted wait done, barrier passed) of its first two tiles.

  python tools/duo_trace.py [--cin 128 --H 256 --W 512] [--gn] [--abl 64]
"""
import argparse
import os
import sys

import numpy as np
import torch

os.environ.setdefault("STORM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "storm_amd", "csrc", "libstorm_hip_prof.so"))
p = argparse.ArgumentParser()
p.add_argument("--B", type=int, default=16)
p.add_argument("--cin", type=int, default=128)
p.add_argument("--H", type=int, default=256)
p.add_argument("--W", type=int, default=512)
p.add_argument("--gn", action="store_true", help="fused GroupNorm + SiLU operand")
p.add_argument("--abl", type=int, default=64)
p.add_argument("--cout", type=int, default=128)
args = p.parse_args()
os.environ["STORM_CONV_ABLATE"] = str(args.abl)
os.environ["STORM_CONV_VARIANT"] = "5"
sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
SLOTS, NW, cout = 512, 4, args.cout
NWG = min(512, args.B * ((args.H + 7) // 8) * ((args.W + 31) // 32) * ((cout + 127) // 128))
trace = torch.zeros(NWG * NW * SLOTS, dtype=torch.int64, device=dev)
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
t = trace.cpu().numpy().reshape(NWG, NW, SLOTS).astype(np.int64)
st = t[:, :, 8:208]
ntile = int((st[0, 0] > 0).sum()) // 6
st = st[:, :, :6 * ntile].reshape(NWG, NW, ntile, 6)
span = np.median(st[:, :, -1, 5].max(axis=1) - st[:, :, 0, 0].min(axis=1))
tick_us = ms * 1e3 / span
print(f"{ms:.3f} ms, {ntile} tiles per workgroup, {tick_us * 1e3:.3f} ns per tick (calibrated on the launch)")
names = ["part B: chunk 0 wait + transform, third tap's weights", "main loop", "hand-over: drain, barrier, next tile's first loads", "epilogue: transpose + stores",
         "statistics", "-> next tile start (barrier)"]
for gname, sel in (("waves 0-3", slice(0, 4)),):
    print(gname)
    d = [st[:, sel, :, i + 1] - st[:, sel, :, i] for i in range(5)]
    d.append(st[:, sel, 1:, 0] - st[:, sel, :-1, 5])
    tot = 0.0
    for nm, v in zip(names, d):
        m = float(v.mean()) * tick_us
        tot += m
        print(f"   {nm:52s} {m:7.2f} us   (p10 {np.percentile(v, 10) * tick_us:6.2f}, p90 {np.percentile(v, 90) * tick_us:6.2f})")
    print(f"   {'tile period':52s} {tot:7.2f} us")
# per-phase stamps of the first two tiles: stream (barrier -> last MFMA issued), counted wait, barrier
ps = t[:, :, 208:]
nph = int((ps[0, 0] > 0).sum()) // 3
if nph:
    ps = ps[:, :, :3 * nph].reshape(NWG, NW, nph, 3)
    stream = (ps[:, :, 1:, 0] - ps[:, :, :-1, 2]) * tick_us
    wait = (ps[:, :, :, 1] - ps[:, :, :, 0]) * tick_us
    bar = (ps[:, :, :, 2] - ps[:, :, :, 1]) * tick_us
    per = nph // 2
    print(f"per phase ({nph} phases stamped, {per} per tile): mean stream {stream.mean():.3f} us, counted wait {wait.mean():.3f} us, barrier {bar.mean():.3f} us")
    print("   phase:  stream   wait   barrier   (second stamped tile, mean over workgroups and waves, us)")
    for i in range(per, min(nph, per + 18)):
        sv = stream[:, :, i - 1].mean() if i > 0 else float("nan")
        print(f"   {i - per:4d}   {sv:7.3f} {wait[:, :, i].mean():7.3f} {bar[:, :, i].mean():7.3f}")
