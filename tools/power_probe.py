#!/usr/bin/env python
"""What the chip does under a sustained convolution: socket power, shader clock and temperature (rocm-smi, sampled from a side
thread every 100 ms) while ONE kernel variant of tools/probe128.py's first cases runs back to back for a few seconds - the evidence
behind "the <= 128-cout layers are power limited, not schedule limited".

  python tools/power_probe.py [--seconds 4] [--modes igemm,duo,p128,idle,mfma]
"""
import argparse
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402
from storm_amd import _lib as L  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--seconds", type=float, default=4.0)
p.add_argument("--modes", default="idle,igemm,duo,p128,pipe,zeros")
p.add_argument("--cin", type=int, default=128)
p.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
p.add_argument("--abl", default="", help="comma list of STORM_CONV_ABLATE values (profiling library, STORM_LIB=...): every mode is run once per value")
p.add_argument("--nogn", action="store_true", help="plain operand")
args = p.parse_args()
dev, dt = torch.device("cuda:0"), (torch.bfloat16 if args.dtype == "bf16" else torch.float16)
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
samples, stop = [], False


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
    except Exception as e:  # noqa: BLE001
        return {"err": str(e)}
    r = {}
    m = re.search(r"Power \(W\):\s*([\d.]+)", out) or re.search(r"Socket Power.*?:\s*([\d.]+)", out)
    if m:
        r["W"] = float(m.group(1))
    m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    if m:
        r["sclk"] = int(m.group(1))
    m = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", out)
    if m:
        r["T"] = float(m.group(1))
    if not r:
        r["raw"] = out[-400:]
    return r


def poll():
    while not stop:
        samples.append((time.time(), smi()))
        time.sleep(0.05)


def run(mode):
    global stop, samples
    B, H, W, cin = 16, 256, 512, args.cin
    cout = 256 if mode == "pipe" else 128
    if mode == "pipe":
        cin = 256
    x = (torch.zeros(B, H, W, cin) if mode == "zeros" else rnd(B, H, W, cin)).to(dt).to(dev)
    w = ops.pack_conv_weight(((torch.zeros if mode == "zeros" else rnd)(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    segs = [ops.Seg(x, w, 9, gn_ss=None if args.nogn else ss, gn_silu=True)]
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=True, scale=0.7)
    variant = {"igemm": 0, "duo": 5, "p128": 4, "pipe": -1, "zeros": 0}.get(mode, -1)
    L.check(L.lib().storm_set_switch(b"STORM_CONV_VARIANT", variant), "storm_set_switch")
    samples, stop = [], False
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n, ms_tot = 0, 0.0
    if mode == "idle":
        time.sleep(1.5)
    else:
        kn = ops.conv_kernel_name(segs, cout, bias=kw["bias"], tbias=kw["tbias"], scale=0.7)
        while time.time() - t0 < args.seconds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.conv(segs, cout, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms_tot += e0.elapsed_time(e1)
            n += 50
            last = e0.elapsed_time(e1) / 50
    stop = True
    th.join()
    late = [s for t, s in samples if t - t0 > 0.5 * (time.time() - t0)]           # second half: the settled state
    def med(k):
        v = sorted(s[k] for s in late if k in s)
        return v[len(v) // 2] if v else float("nan")
    fl = 2 * B * H * W * cout * cin * 9
    line = f"{args.dtype} {mode:6s} power {med('W'):7.1f} W  sclk {med('sclk'):6.0f} MHz  T {med('T'):5.1f} C  ({len(late)} samples)"
    if n:
        line += f"  | {kn.split('<')[0][7:]:20s} first-to-last mean {ms_tot / n:.3f} ms, settled {last:.3f} ms = {fl / last / 1e9:5.0f} TF/s"
    print(line, flush=True)
    if late and "raw" in late[-1]:
        print(late[-1]["raw"])


for m in args.modes.split(","):
    if args.abl:
        for a in args.abl.split(","):
            L.check(L.lib().storm_set_switch(b"STORM_CONV_ABLATE", int(a)), "storm_set_switch")
            print(f"STORM_CONV_ABLATE={a}: ", end="")
            run(m)
        L.lib().storm_set_switch(b"STORM_CONV_ABLATE", 0)
    else:
        run(m)
