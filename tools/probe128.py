#!/usr/bin/env python
"""A/B probe of the <= 128-cout 3x3 layers of the bench shape: conv_pipe128.hip vs conv_igemm.hip (forced through the
library's STORM_CONV_VARIANT switch), with the fusions the network uses (GroupNorm-apply operand, statistics epilogue, temb bias,
fused 1x1 shortcut); d = rel-L2 of the output vs the generic kernel's, p = max difference of the statistics partials."""
import argparse
import os
import sys

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
from storm_amd import _lib as L  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--reps", type=int, default=5)
p.add_argument("--B", type=int, default=16)
p.add_argument("--nogn", action="store_true", help="plain operand: no fused GroupNorm-apply + SiLU on the load")
p.add_argument("--nosilu", action="store_true", help="fused GroupNorm affine without the SiLU (what the transcendentals cost)")
p.add_argument("--nostats", action="store_true", help="no fused GroupNorm statistics in the epilogue (what they cost)")
p.add_argument("--only", type=int, default=-1)
p.add_argument("--modes", default="", help="comma list of kernels to time (default: all)")
args = p.parse_args()
dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731

# (name, H, W, cin of the 3x3 operand, channels of the fused 1x1 shortcut or 0): idx 6/8, 105, 109, 107/111, 11, 13 of the op list
CASES = [("128->128 @256x512", 256, 512, 128, 0), ("384->128 @256x512", 256, 512, 384, 0), ("256->128 @256x512", 256, 512, 256, 0),
         ("128->128 +1x1(256) @256x512", 256, 512, 128, 256), ("128->128 @128x256", 128, 256, 128, 0),
         ("128->128 +1x1(128) @128x256", 128, 256, 128, 128), ("128->128 @256x1024 (8 s)", 256, 1024, 128, 0),
         # 256-cout layers (conv_pipe.hip on both sides of the A/B): what the fused GroupNorm operand costs there (--nogn)
         ("256->256 @128x256", 128, 256, 256, 0, 256), ("512->256 @128x256", 128, 256, 512, 0, 256), ("256->256 @256x512", 256, 512, 256, 0, 256)]
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
for ci_, case in enumerate(CASES):
    name, H, W, cin, sc = case[:5]
    if args.only >= 0 and ci_ != args.only:
        continue
    B = args.B if W <= 512 else args.B // 2
    cout = case[5] if len(case) > 5 else 128
    x = rnd(B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    segs = [ops.Seg(x, w, 9, gn_ss=None if args.nogn else ss, gn_silu=not args.nosilu)]
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=not args.nostats, scale=0.7)
    if sc:
        segs.append(ops.Seg(rnd(B, H, W, sc).to(dt).to(dev), ops.pack_conv_weight((rnd(cout, sc, 1, 1) * 0.05).to(dev), dt), 1))
    fl = 2 * B * H * W * cout * (cin * 9 + sc)
    out = {}
    lib = L.lib()
    MODES = {"duo": 5, "p128": 4, "half": 9, "igemm": 0} if cout <= 128 else {"pipe": 3, "half": 9, "igemm": 2}
    if args.modes:
        MODES = {k: v for k, v in MODES.items() if k in args.modes.split(",")}
        if not MODES:
            continue
    for sw, variant in MODES.items():
        L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", variant), "storm_set_switch")
        kn = ops.conv_kernel_name(segs, cout, bias=kw["bias"], tbias=kw["tbias"], scale=0.7)
        unpack = (lambda r: r) if not args.nostats else (lambda r: (r, torch.zeros(1)))
        for _ in range(2):
            y, part = unpack(ops.conv(segs, cout, **kw))
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(args.reps):
            y, part = unpack(ops.conv(segs, cout, **kw))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out[sw] = (ms, y.float(), part, kn)
    lib.storm_set_switch(b"STORM_CONV_VARIANT", -1)
    ref = out["igemm"] if "igemm" in out else out[list(MODES)[0]]
    line = f"{name:30s}"
    for sw in MODES:
        ms, y, part, kn = out[sw]
        d = float((y - ref[1]).norm() / ref[1].norm())
        pd = float((part - ref[2]).abs().max() / ref[2].abs().max())
        line += f" | {sw} {ms:.3f} ms {fl / ms / 1e9:5.0f} TF d={d:.1e} p={pd:.1e}"
    print(line + " | " + out[list(MODES)[0]][3].split("<")[0][7:], flush=True)
