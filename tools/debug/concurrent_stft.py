#!/usr/bin/env python
"""Debug: STFT front end on several streams at once (GPU)."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from storm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
mode = sys.argv[1]
lens_all = [[26864, 25777, 24967], [20800, 20188, 20035], [17108, 16511], [12582, 11319, 8781]]
ins = []
for bl in lens_all:
    yb = torch.zeros(len(bl), max(bl))
    for k, n_ in enumerate(bl):
        yb[k, :n_] = 0.1 * torch.randn(n_, generator=g)
    ins.append((yb.to(dev), bl))


def front(yb, bl, rl=None):
    if mode == "nolen":
        bl = None
    lens = bl if rl is None else rl
    peak = ops.peak_abs(yb, lens)
    Y = ops.stft(yb, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64, lengths=lens)
    return peak, Y


solo = [front(yb, bl) for yb, bl in ins]
torch.cuda.synchronize()
pre = [torch.tensor(bl, dtype=torch.int32, device=dev) for _, bl in ins]
torch.cuda.synchronize()
bad = {}
nl = 3
streams = [torch.cuda.Stream() for _ in range(nl)]
start = threading.Barrier(nl)


def lane(j):
    with torch.cuda.stream(streams[j]):
        start.wait()
        for rep in range(200):
            k = (rep + j) % len(ins)
            yb, bl = ins[k]
            peak, Y = front(yb, bl, pre[k] if mode == "prelen" else None)
            streams[j].synchronize()
            if not torch.equal(peak, solo[k][0]):
                bad[("peak", k)] = bad.get(("peak", k), 0) + 1
            if not torch.equal(Y, solo[k][1]):
                bad[("Y", k)] = bad.get(("Y", k), 0) + 1


th = [threading.Thread(target=lane, args=(j,)) for j in range(nl)]
[t.start() for t in th]
[t.join() for t in th]
print(mode, "mismatches:", bad or "none")
