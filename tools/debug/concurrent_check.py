#!/usr/bin/env python
"""Debug: where do concurrent micro-batches differ from the sequential stream?  (GPU)"""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncsnpp_ref as NR
from storm_amd import distributed as D
from storm_amd.model import ScoreModel

dev = torch.device("cuda:0")
COMMON = dict(sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5, nf=16)
m = ScoreModel(backbone="ncsnpp", **COMMON)
m.dnn.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(nf=16, input_channels=4), seed=5))
m.eval(no_ema=True)
m = m.to(dev)
m.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp16")
g = torch.Generator().manual_seed(77)
lens = [int(v) for v in torch.randint(6000, 30001, (11,), generator=g)]
batches = []
for ids in D.bucket_by_frames(lens, 3):
    bl = [lens[i] for i in ids]
    yb = torch.zeros(len(ids), max(bl))
    for k, n_ in enumerate(bl):
        yb[k, :n_] = 0.1 * torch.randn(n_, generator=g)
    batches.append((yb.to(dev), None if len(set(bl)) == 1 else bl))
print("batches:", [(tuple(y.shape), bl) for y, bl in batches])
kw = dict(sampler_type="pc", N=3, corrector="ald", snr=0.5)


def one(kb):
    k, (yb, bl) = kb
    return m.enhance_batch(yb, seed=100 + k, lengths=bl, **kw)


def run(n, graph):
    m.dnn.set_graph(graph)
    return [o.cpu() for o in D.run_concurrent(one, list(enumerate(batches)), n)]


def cmp(tag, a, b):
    bad = []
    for k, (x, y) in enumerate(zip(a, b)):
        for r in range(x.shape[0]):
            if not torch.equal(x[r], y[r]):
                bad.append((k, r, float((x[r] - y[r]).abs().max()), float(x[r].abs().max())))
    print(tag, "OK" if not bad else f"MISMATCH {bad}")


ref = run(1, 0)
cmp("seq vs seq (eager)", ref, run(1, 0))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    side = run(1, 0)
s.synchronize()
cmp("seq on a side stream vs main (eager)", ref, side)
with torch.cuda.stream(s):
    side = run(1, 0)
s.synchronize()
cmp("seq on a side stream again", ref, side)
for n in (2, 3):
    for rep in range(2):
        cmp(f"{n} lanes, eager, rep {rep}", ref, run(n, 0))
cmp("seq, graph on, pass 1", ref, run(1, 1))
cmp("seq, graph on, pass 2", ref, run(1, 1))
cmp("seq, graph on, pass 3", ref, run(1, 1))
for rep in range(3):
    cmp(f"3 lanes, graph on, rep {rep}", ref, run(3, 1))
