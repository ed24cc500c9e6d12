#!/usr/bin/env python
"""Debug: victim x aggressor matrix - is the corruption tied to MY victim (stft), MY aggressor (conv_igemm WM = 1), or neither? (GPU)"""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from storm_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
dt = torch.bfloat16
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
wav = (0.1 * torch.randn(3, 12582, generator=g)).to(dev)
peak = ops.peak_abs(wav)
xa = nhwc(torch.randn(8, 256, 32, 64, generator=g)).to(dt).to(dev)
wa = ops.pack_conv_weight((torch.randn(256, 256, 3, 3, generator=g) * 0.05).to(dev), dt)
A = torch.randn(2048, 2048, generator=g).to(dt).to(dev)
Bm = torch.randn(2048, 2048, generator=g).to(dt).to(dev)
big = torch.randn(64, 1 << 18, generator=g, dtype=torch.float64).to(dev)
sp = torch.randn(3, 256, 128, dtype=torch.complex64, generator=g).to(dev)


def conv_v(variant):
    def run():
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", variant), "switch")
        y = ops.conv([ops.Seg(xa, wa, 9)], 256)
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", -1), "switch")
        return y
    return run


victims = {
    "stft (mine: LDS + fp64)": lambda: ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64),
    "istft (mine)": lambda: ops.istft(sp, 128 * 127, peak, spec_factor=0.15, spec_abs_exponent=0.5),
    "torch fp64 sum": lambda: big.sum(dim=1),
    "torch rfft": lambda: torch.fft.rfft(wav, n=8192),
}
aggressors = {"conv v7 (mine, 177 VGPR)": conv_v(7), "conv v0 (mine, 256 VGPR)": conv_v(0), "torch bf16 matmul": lambda: A @ Bm, "torch fp64 exp": lambda: big.exp()}
for vn, vf in victims.items():
    v0 = vf().clone()
    torch.cuda.synchronize()
    for an, af in aggressors.items():
        s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
        stop = [False]
        cnt = {"bad": 0, "n": 0, "agg": 0}

        def victim():
            with torch.cuda.stream(s0):
                while not stop[0]:
                    y = vf()
                    s0.synchronize()
                    cnt["n"] += 1
                    cnt["bad"] += int(not torch.equal(y, v0))

        def aggressor():
            with torch.cuda.stream(s1):
                t0 = time.time()
                while time.time() - t0 < 1.5:
                    af()
                    cnt["agg"] += 1
                    if cnt["agg"] % 16 == 0:
                        s1.synchronize()
                s1.synchronize()
            stop[0] = True

        th = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
        [q.start() for q in th]
        [q.join() for q in th]
        print(f"victim {vn:26s} aggressor {an:26s} corrupted {cnt['bad']:5d} of {cnt['n']:6d}   (aggressor launches {cnt['agg']})")
