#!/usr/bin/env python
"""Experiment: one score evaluation of batch 16 on one stream against two evaluations of batch 8 on two streams (separate handles and
workspaces), so that one stream's kernel tails / ramps can be filled by the other's work.  Prints ms per 16 utterances."""
import copy
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from storm_amd.backbones.ncsnpp import NCSNpp  # noqa: E402

dev = torch.device("cuda:0")
net = NCSNpp().to(dev)
bench.randomize(net, 0)
net.set_compute_dtype(torch.bfloat16)
net2 = copy.deepcopy(net)
F_, T_ = 256, 512
g = torch.Generator().manual_seed(1)
def cplx(B):
    return torch.complex(torch.randn(B, F_, T_, generator=g), torch.randn(B, F_, T_, generator=g)).to(dev)
x16, y16, t16 = cplx(16), cplx(16), torch.rand(16, generator=g).to(dev) * 0.9 + 0.05
xa, ya, ta = x16[:8].contiguous(), y16[:8].contiguous(), t16[:8].contiguous()
xb, yb, tb = x16[8:].contiguous(), y16[8:].contiguous(), t16[8:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        net.forward_parts([x16, y16], t16)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def two(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1):
            net.forward_parts([xa, ya], ta)
        with torch.cuda.stream(s2):
            net2.forward_parts([xb, yb], tb)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def seq8(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        net.forward_parts([xa, ya], ta)
        net.forward_parts([xb, yb], tb)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for f in (one, two, seq8):
    f(3)
for rep in range(2):
    print(f"batch 16, one stream: {one(20):.3f} ms | 2 x batch 8, two streams: {two(20):.3f} ms | 2 x batch 8, one stream: {seq8(20):.3f} ms", flush=True)
