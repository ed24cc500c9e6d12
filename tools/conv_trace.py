#!/usr/bin/env python
"""Wave-level timeline of the conv kernel (profiling instantiation STORM_CONV_ABLATE=64): every wave stamps
s_memtime at the phase boundaries of its tile; this tool launches one conv, reads the stamps back and prints
where a wave's time goes (barrier waits, MFMA phases, LDS commits, prologue, epilogue, store drain).

  STORM_CONV_VARIANT=0 python tools/conv_trace.py [--cin 256 --cout 256 --H 128 --W 256]
"""
import argparse
import os
import sys

import numpy as np
import torch

os.environ["STORM_CONV_ABLATE"] = str(64 + int(os.environ.get("STORM_TRACE_EXTRA", "0")))
# stamps exist only in the profiling build (python -m storm_amd.build --profiling)
os.environ.setdefault("STORM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "storm_amd", "csrc", "libstorm_hip_prof.so"))
sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--B", type=int, default=16)
p.add_argument("--cin", type=int, default=256)
p.add_argument("--cout", type=int, default=256)
p.add_argument("--H", type=int, default=128)
p.add_argument("--W", type=int, default=256)
p.add_argument("--out", default="")
args = p.parse_args()

dev = torch.device("cuda:0")
SLOTS = 512
variant = int(os.environ.get("STORM_CONV_VARIANT", "0"))
nwaves = 8 if variant in (1, 2, 3) else 4   # (variant 3 = conv_pipe.hip)
bn = 256 if variant in (2, 3) else 128
tiles = args.B * ((args.H + 7) // 8) * ((args.W + 31) // 32)
n_ct = (args.cout + bn - 1) // bn
vblocks = 8 * ((tiles + 7) // 8) * n_ct
trace = torch.zeros(vblocks * nwaves * SLOTS, dtype=torch.int64, device=dev)
os.environ["STORM_CONV_TRACE_PTR"] = hex(trace.data_ptr())

g = torch.Generator().manual_seed(0)
x = torch.randn(args.B, args.H, args.W, args.cin, generator=g).to(torch.bfloat16).to(dev)
w = ops.pack_conv_weight((torch.randn(args.cout, args.cin, 3, 3, generator=g) * 0.05).to(dev), torch.bfloat16)
b = torch.randn(args.cout, generator=g).to(dev)
for _ in range(3):
    y = ops.conv([ops.Seg(x, w, 9)], args.cout, bias=b)
torch.cuda.synchronize()
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
y = ops.conv([ops.Seg(x, w, 9)], args.cout, bias=b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = trace.cpu().numpy().reshape(vblocks, nwaves, SLOTS).astype(np.int64)
if args.out:
    np.save(args.out, t)

valid = t[:, 0, 1] > 0
t = t[valid]
nb = t.shape[0]
# (s_memtime counters are not synchronised across XCDs / SEs: only differences on one CU are meaningful)
hw_ = t[:, 0, 0]
cu_ = ((hw_ >> 8) & 0xF) | (((hw_ >> 13) & 0x7) << 4) | (((hw_ >> 32) & 0xF) << 8)
cspans = []
for c in np.unique(cu_):                     # first start -> last end on one CU ~ the kernel duration
    sel = cu_ == c
    cspans.append(int(t[sel][:, :, 503].max() - t[sel][:, :, 1].min()))
span = float(np.median(cspans))
tick_ns = ms * 1e6 / span           # s_memtime ticks -> ns (tick = shader clock): calibrated on the launch duration
print(f"effective shader clock ~ {1.0 / tick_ns:.2f} GHz")
print(f"variant {variant}: {ms:.3f} ms, {nb} workgroups x {nwaves} waves, span {span} ticks, {tick_ns:.3f} ns/tick")

if variant == 3:
    raw = t[:, :, 4:4 + 16 * 30].reshape(nb, nwaves, 30, 16)
    steps = int((raw[0, 0, :, 0] > 0).sum())
    # stamps per tap-step (slot 4 + 16 * step + idx), in time order
    idx = [0, 1, 2, 4, 6, 7, 12, 8, 9, 10, 11]
    st = raw[:, :, :steps][..., idx]
    us3 = lambda d: d * tick_ns / 1e3  # noqa: E731
    names = ["S(h0): fragment reads + weight DMA issue", "S(h0): [patch piece] vmcnt [transform]", "S(h0): lgkmcnt + barrier",
             "C(h0): reads + 16 MFMA", "C(h0): lgkmcnt + barrier",
             "S(h1): fragment reads + weight DMA issue", "S(h1): [patch piece] vmcnt [transform]", "S(h1): lgkmcnt + barrier",
             "C(h1): reads + 16 MFMA", "C(h1): lgkmcnt + barrier"]
    print(f"per wave, first {steps} tap-steps (us; tick->us calibrated on the launch; every stamp costs ~0.09 us itself):")
    for gname, sel in (("g0 (waves 0-3)", slice(0, 4)), ("g1 (waves 4-7)", slice(4, 8))):
        print(f" {gname}: tile total {us3((t[:, sel, 503] - t[:, sel, 1]).mean()):.2f}, prologue {us3((t[:, sel, 2] - t[:, sel, 1]).mean()):.2f}")
        for i, nm in enumerate(names):
            d = us3((st[:, sel, :, i + 1] - st[:, sel, :, i]).astype(np.float64))
            print(f"   {nm:52s} mean/step {d.mean():7.3f}   p10 {np.percentile(d, 10):7.3f}   p90 {np.percentile(d, 90):7.3f}")
        adv = us3((st[:, sel, 1:, 0] - st[:, sel, :-1, 10]).astype(np.float64))
        print(f"   {'step end -> next step (chunk change every 9th)':52s} mean/step {adv.mean():7.3f}   p10 {np.percentile(adv, 10):7.3f}   p90 {np.percentile(adv, 90):7.3f}")
        whole = us3((st[:, sel, 1:, 0] - st[:, sel, :-1, 0]).astype(np.float64))
        print(f"   {'whole step':52s} mean      {whole.mean():7.3f}")
        # per-step profile of S(h0) wait (vmcnt) and the chunk position
        w0 = us3((st[:, sel, :, 2] - st[:, sel, :, 1]).astype(np.float64)).mean(axis=(0, 1))
        w1 = us3((st[:, sel, :, 7] - st[:, sel, :, 6]).astype(np.float64)).mean(axis=(0, 1))
        print("   vmcnt interval by tap-step, h0:", " ".join(f"{v:.2f}" for v in w0[:18]))
        print("   vmcnt interval by tap-step, h1:", " ".join(f"{v:.2f}" for v in w1[:18]))
    print(f" epilogue: drain+barrier {us3((t[:, :, 501] - t[:, :, 500]).mean()):.2f}, transpose+stores {us3((t[:, :, 502] - t[:, :, 501]).mean()):.2f}, store drain {us3((t[:, :, 503] - t[:, :, 502]).mean()):.2f}")
    sys.exit(0)
steps = int(((t[0, 0, 4:400].reshape(-1, 4)[:, 0]) > 0).sum())
nchunks = int((t[0, 0, 400:500].reshape(-1, 4)[:, 0] > 0).sum())
us = lambda d: d * tick_ns / 1e3  # noqa: E731


def stat(name, d):
    d = us(np.asarray(d, dtype=np.float64))
    print(f"  {name:34s} mean {d.mean():8.3f} us   p10 {np.percentile(d, 10):8.3f}   p90 {np.percentile(d, 90):8.3f}")
    return d.mean()


print(f"per wave and tile ({steps} steps, {nchunks} non-prefetched patch stages):")
total = stat("tile total (start -> stores drained)", t[:, :, 503] - t[:, :, 1])
stat("prologue (start -> main loop)", t[:, :, 2] - t[:, :, 1])
st = t[:, :, 4:4 + 4 * steps].reshape(nb, nwaves, steps, 4)
bar = (st[..., 1] - st[..., 0]).sum(-1)
comp = (st[..., 2] - st[..., 1]).sum(-1)
commit = (st[..., 3] - st[..., 2]).sum(-1)
nxt = np.concatenate([st[:, :, 1:, 0], t[:, :, 500][..., None]], -1)
between = (nxt - st[..., 3]).sum(-1)
m_bar = stat("sum of tap barriers", bar)
m_comp = stat("sum of MFMA phases", comp)
m_commit = stat("sum of weight/patch LDS commits", commit)
m_between = stat("sum of between-step (loads issue...)", between)
if nchunks:
    pc = t[:, :, 400:400 + 4 * nchunks].reshape(nb, nwaves, nchunks, 4)
    stat("  of which: patch barrier", (pc[..., 1] - pc[..., 0]).sum(-1))
    stat("  of which: patch half 1 load+commit", (pc[..., 2] - pc[..., 1]).sum(-1))
    stat("  of which: patch half 2 load+commit", (pc[..., 3] - pc[..., 2]).sum(-1))
stat("epilogue barrier", t[:, :, 501] - t[:, :, 500])
stat("epilogue (transpose + stores issue)", t[:, :, 502] - t[:, :, 501])
stat("store drain (vmcnt 0)", t[:, :, 503] - t[:, :, 502])
mfma_ideal = 2.0 * 64 * 128 * args.cin * 9 / (1024 * 2.4e9) * 1e6 if variant != 1 else 0
print(f"  ideal MFMA time of one wave's tile at 2.4 GHz: {mfma_ideal:.2f} us;  per-step MFMA phase mean {us(comp.mean() / steps):.3f} us")

# per-step detail of the median workgroup, wave 0
order = np.argsort(t[:, 0, 503] - t[:, 0, 1])
mid = order[len(order) // 2]
print(f"median workgroup (index {mid}), wave 0, first 12 steps: [barrier, mfma, commit] us")
for s in range(min(steps, 12)):
    a_ = st[mid, 0, s]
    print(f"    step {s:2d}: {us(a_[1] - a_[0]):7.3f} {us(a_[2] - a_[1]):7.3f} {us(a_[3] - a_[2]):7.3f}")

# co-residency: workgroups per CU over time
hw = t[:, 0, 0]
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 32) & 0xF) << 8)
ids, counts = np.unique(cu, return_counts=True)
print(f"distinct (xcc, se, cu) ids: {len(ids)}; workgroups per id: min {counts.min()} max {counts.max()}")
one = np.where(cu == ids[0])[0]
one = one[np.argsort(t[one, 0, 1])]
t_begin = t[one[0], 0, 1]
print("timeline on one CU (us from launch start): start, main loop, epilogue start, end")
for i in one[:10]:
    print(f"    wg {i:5d}: {us(t[i, 0, 1] - t_begin):8.2f} {us(t[i, 0, 2] - t_begin):8.2f} {us(t[i, 0, 500] - t_begin):8.2f} {us(t[i, 0, 503] - t_begin):8.2f}")
