#!/usr/bin/env python
"""Debug: the score network's forward on two streams at once against its solo result (GPU)."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncsnpp_ref as NR
from storm_amd.backbones.ncsnpp import NCSNpp
from storm_amd import _lib as L

dev = torch.device("cuda:0")
nf, prec = int(sys.argv[1]), sys.argv[2]
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[3].split(",")]      # BxT per lane
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    L.check(L.lib().storm_set_switch(k.encode(), int(v)), k)
kw = dict(nf=nf, input_channels=4)
net = NCSNpp(**kw)
net.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(**kw), seed=5))
net = net.to(dev)
net.set_compute_dtype({"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[prec])
g = torch.Generator().manual_seed(3)
ins = []
for (B, T) in shapes:
    x = (torch.randn(B, 2, 256, T, dtype=torch.complex64, generator=g) * 0.5).to(dev)
    t = (torch.rand(B, generator=g) * 0.9 + 0.05).to(dev)
    ins.append((x, t))
solo = [net(x, t).clone() for x, t in ins]
torch.cuda.synchronize()
assert all(torch.equal(net(x, t), s) for (x, t), s in zip(ins, solo)), "solo run not reproducible"
streams = [torch.cuda.Stream() for _ in ins]
bad = [0] * len(ins)
worst = [0.0] * len(ins)
rows = [set() for _ in ins]
start = threading.Barrier(len(ins))


def lane(k):
    x, t = ins[k]
    with torch.cuda.stream(streams[k]), torch.no_grad():
        start.wait()
        for r in range(reps):
            y = net(x, t)
            streams[k].synchronize()
            if not torch.equal(y, solo[k]):
                bad[k] += 1
                d = (y - solo[k]).abs().amax(dim=(1, 2, 3))
                worst[k] = max(worst[k], float(d.max()))
                rows[k] |= {int(i) for i in torch.nonzero(d > 0).flatten()}


th = [threading.Thread(target=lane, args=(k,)) for k in range(len(ins))]
[t.start() for t in th]
[t.join() for t in th]
print(f"nf={nf} {prec} shapes={shapes} switches={sys.argv[5:]}: mismatching evaluations per lane {bad} of {reps}, worst abs diff {worst}, rows {rows}, |y|max {[float(s.abs().max()) for s in solo]}")
