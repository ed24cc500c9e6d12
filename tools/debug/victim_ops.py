#!/usr/bin/env python
"""Debug: which single kernel corrupts a co-resident victim (STFT, 12 KiB of static LDS) on another stream? (GPU)"""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from storm_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
wav = (0.1 * torch.randn(3, 12582, generator=g)).to(dev)
peak = ops.peak_abs(wav)
Y0 = ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64).clone()
dt = torch.bfloat16
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()


def conv_case(cin, cout, B, H, W, variant, gn=False, taps=9):
    x = nhwc(torch.randn(B, cin, H, W, generator=g)).to(dt).to(dev)
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) * 0.05).to(dev), dt)
    ss = None
    if gn:
        _, part = ops.conv([ops.Seg(x, w, taps)], cout, gn_partials=True) if cin == cout else (None, None)
    seg = [ops.Seg(x, w, taps)]

    def run():
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", variant), "switch")
        y = ops.conv(seg, cout)
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", -1), "switch")
        return y
    name = ops.conv_kernel_name(seg, cout) if variant < 0 else f"variant {variant}"
    return run, name


cases = {
    "igemm128_v0": conv_case(128, 128, 4, 64, 128, 0),
    "igemm_v2": conv_case(256, 256, 4, 64, 128, 2),
    "pipe_v3": conv_case(256, 256, 4, 64, 128, 3),
    "pipe128_v4": conv_case(128, 128, 4, 64, 128, 4),
    "igemm64_v7": conv_case(256, 256, 4, 32, 64, 7),
    "pipe_half_v9": conv_case(256, 256, 4, 32, 64, 9),
    "small_out32": conv_case(64, 32, 4, 64, 128, -1),
    "igemm_1x1": conv_case(256, 256, 4, 64, 128, -1, taps=1),
}
which = sys.argv[1:] or list(cases)
for nm in which:
    run, kname = cases[nm]
    y0 = run().clone()
    torch.cuda.synchronize()
    stop = [False]
    bad = {"stft": 0, "n_stft": 0, "agg": 0, "n_agg": 0}
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def victim():
        with torch.cuda.stream(s0):
            while not stop[0]:
                Y = ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64)
                s0.synchronize()
                bad["n_stft"] += 1
                if not torch.equal(Y, Y0):
                    bad["stft"] += 1
                    if bad["stft"] <= 3:
                        d = torch.nonzero(torch.view_as_real(Y) != torch.view_as_real(Y0))
                        rows, fs, ts = sorted(set(d[:, 0].tolist())), sorted(set(d[:, 1].tolist())), sorted(set(d[:, 2].tolist()))
                        i0 = d[0].tolist()
                        print(f"   mismatch: {d.shape[0]} floats; rows {rows}, f {fs[:6]}..{fs[-3:]} ({len(fs)}), frames {ts[:8]} ({len(ts)}); first at {i0}: got {torch.view_as_real(Y)[i0[0], i0[1], i0[2]].tolist()} want {torch.view_as_real(Y0)[i0[0], i0[1], i0[2]].tolist()}")
                        Y1 = ops.stft(wav, peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64)
                        s0.synchronize()
                        print("   recomputed equal:", torch.equal(Y1, Y0), " bad copy unchanged:", int((torch.view_as_real(Y) != torch.view_as_real(Y0)).sum()))

    def aggressor():
        with torch.cuda.stream(s1):
            import time
            t0 = time.time()
            while time.time() - t0 < 2.5:
                y = run()
                bad["n_agg"] += 1
                if bad["n_agg"] % 16 == 0:
                    s1.synchronize()
            s1.synchronize()
            bad["agg"] += int(not torch.equal(y, y0))
        stop[0] = True

    th = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
    [q.start() for q in th]
    [q.join() for q in th]
    print(f"{nm:14s} {kname[:70]:70s} {bad}")
