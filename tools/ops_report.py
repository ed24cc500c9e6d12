#!/usr/bin/env python
"""Per-op report of a bench.py --ops-json dump: achieved TFLOP/s per conv launch, TB/s per GN launch."""
import json
import sys

sys.path.insert(0, ".")
from storm_amd import _lib as L
from storm_amd.backbones.plan import NCSNppConfig, ParamLayout, Program

rows = json.load(open(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = NCSNppConfig()
prog = Program(cfg, ParamLayout(cfg, L.BF16), B, 256, 512)
tot = 0
agg = {}
for r, op in zip(rows, prog.ops):
    c, ms = r["code"], r["ms"]
    tot += ms
    if c == 4:
        key = f"conv {r['H']:3d}x{r['W']:<4d} cin{str(r['cin']):6s} cout{r['Cout']:4d} taps{r['taps']}"
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += r["flops"]
    elif c == 5:
        Ca, Cb, Bq, HW, G = [int(op.i[j]) for j in range(5)]
        key = f"gn_stats C{Ca + Cb:3d} HW{HW:6d}"
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += Bq * HW * (Ca + Cb) * 2
    elif c == 6:
        Ca, Cb, Bq, H, W, G, silu, rs = [int(op.i[j]) for j in range(8)]
        o = {0: 1, 1: 4, 2: 0.25}[rs]
        key = f"gn_apply C{Ca + Cb:3d} {H}x{W} rs{rs}"
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += Bq * H * W * (Ca + Cb) * 2 * (1 + o * (2 if rs else 1))
for k, (n, ms, q) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    unit = "TF" if k.startswith("conv") else "TB/s"
    rate = q / ms / 1e9
    print(f"{ms:7.3f} ms  x{n:2d}  {k:50s} {rate:8.2f} {unit}")
print("total", tot)
