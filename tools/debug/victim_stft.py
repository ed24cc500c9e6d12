#!/usr/bin/env python
"""Debug: a victim kernel (STFT, static LDS) looping on one stream while the score network runs on another (GPU)."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncsnpp_ref as NR
from storm_amd import ops, _lib as L
from storm_amd.backbones.ncsnpp import NCSNpp

dev = torch.device("cuda:0")
nf, prec, B, T = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
fusion = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else None
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    L.check(L.lib().storm_set_switch(k.encode(), int(v)), k)
kw = dict(nf=nf, input_channels=4)
net = NCSNpp(**kw)
net.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(**kw), seed=5))
net = net.to(dev)
net.set_compute_dtype({"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[prec])
g = torch.Generator().manual_seed(3)
x = (torch.randn(B, 2, 256, T, dtype=torch.complex64, generator=g) * 0.5).to(dev)
t = (torch.rand(B, generator=g) * 0.9 + 0.05).to(dev)
y0 = net(x, t).clone()
if fusion is not None:
    h = net._get_handle(L.dt(net.compute_dtype), dev)
    L.check(L.lib().storm_ncsnpp_set_fusion(h, *fusion), "fusion")
    y0 = net(x, t).clone()
wav = (0.1 * torch.randn(3, 12582, generator=g)).to(dev)
peak = ops.peak_abs(wav)
Y0 = ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64).clone()
torch.cuda.synchronize()
stop = False
bad = {"stft": 0, "net": 0, "n_stft": 0, "n_net": 0}
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()


def victim():
    with torch.cuda.stream(s0):
        while not stop:
            Y = ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64)
            s0.synchronize()
            bad["n_stft"] += 1
            if not torch.equal(Y, Y0):
                bad["stft"] += 1


def aggressor():
    global stop
    with torch.cuda.stream(s1), torch.no_grad():
        for _ in range(40):
            y = net(x, t)
            s1.synchronize()
            bad["n_net"] += 1
            if not torch.equal(y, y0):
                bad["net"] += 1
    stop = True


th = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
[q.start() for q in th]
[q.join() for q in th]
print(f"nf={nf} {prec} B={B} T={T} fusion={fusion} switches={sys.argv[6:]}: {bad}")
