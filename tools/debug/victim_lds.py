#!/usr/bin/env python
"""Debug: does a conv kernel on another stream write into the LDS of a co-resident workgroup of ANOTHER kernel? (GPU)"""
import ctypes as C, os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from storm_amd import ops, _lib as L

V = C.CDLL(os.path.join(ROOT, "tools", "debug", "libvictim.so"))
V.victim_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
dt = torch.bfloat16
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()


def conv_case(cin, cout, B, H, W, variant, taps=9):
    x = nhwc(torch.randn(B, cin, H, W, generator=g)).to(dt).to(dev)
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=g) * 0.05).to(dev), dt)
    seg = [ops.Seg(x, w, taps)]

    def run():
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", variant), "switch")
        y = ops.conv(seg, cout)
        L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", -1), "switch")
        return y
    return run


cases = {"igemm128_v0": conv_case(128, 128, 8, 64, 128, 0), "igemm_v2": conv_case(256, 256, 8, 64, 128, 2), "pipe_v3": conv_case(256, 256, 8, 64, 128, 3),
         "pipe128_v4": conv_case(128, 128, 8, 64, 128, 4), "igemm64_v7": conv_case(256, 256, 8, 32, 64, 7), "pipe_half_v9": conv_case(256, 256, 8, 32, 64, 9),
         "small_out32": conv_case(64, 32, 8, 64, 128, -1), "igemm_1x1": conv_case(256, 256, 8, 64, 128, -1, taps=1), "none": None}
lds_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = sys.argv[3:]
for nm, run in cases.items():
    if only and nm not in only:
        continue
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    first = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    stop = [False]
    n = [0, 0]

    def victim():
        with torch.cuda.stream(s0):
            while not stop[0]:
                V.victim_launch(bad.data_ptr(), first.data_ptr(), 1024, lds_bytes, 3000 if mode < 2 else 20000, mode, s0.cuda_stream)   # (s_memtime ticks at 100 MHz: 30 us)
                n[0] += 1
                if n[0] % 8 == 0:
                    s0.synchronize()
            s0.synchronize()

    def aggressor():
        with torch.cuda.stream(s1):
            import time
            t0 = time.time()
            while time.time() - t0 < 2.5:
                if run is not None:
                    run()
                n[1] += 1
                if n[1] % 16 == 0:
                    s1.synchronize()
            s1.synchronize()
        stop[0] = True

    th = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
    [q.start() for q in th]
    [q.join() for q in th]
    torch.cuda.synchronize()
    f = first.tolist()
    print(f"{nm:14s} victim launches {n[0]:5d}: corrupted LDS words {int(bad[0])}" + (f"  first: word/lane {f[0]} = {f[1] & 0xffffffff:#010x} in block {f[2]} (thread / ref {f[3] & 0xffffffff:#x})" if int(bad[0]) else ""))
