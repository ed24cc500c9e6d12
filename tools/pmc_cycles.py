#!/usr/bin/env python
"""Per-dispatch cycle counters of conv_probe runs (rocprofv3 --pmc output) -> one line per (kernel, grid) with
cycles per launch: clock-independent attribution of what a work-skipping instantiation saves (the chip clocks to its power budget,
so TIME comparisons between instantiations that move different data are confounded)."""
import collections
import csv
import glob
import sys

root, tag = sys.argv[1], sys.argv[2]
pats = sys.argv[3].split(",") if len(sys.argv) > 3 else ["conv_"]          # kernel-name substrings to report
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(p_ in k for p_ in pats):
            continue
        acc[(k.split("(")[0].replace("storm::", "").replace("void ", ""), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), c in sorted(acc.items()):
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    gui = med.get("GRBM_GUI_ACTIVE", 0) / 8.0                    # summed over the 8 XCDs
    busy = med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    n_disp = len(next(iter(c.values())))
    line = f"{tag:8s} {k[:64]:64s} grid {int(grid):8d} threads x{n_disp:4d}: cycles/launch {gui:10.0f}"
    if busy and gui:
        line += f"  mfma-busy {busy / (gui * 1024):5.3f}"
    for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if n in med and med.get("SQ_WAVE_CYCLES"):
            line += f"  {n[3:]} {med[n] / med['SQ_WAVE_CYCLES']:5.3f}" if n != "SQ_WAVE_CYCLES" else ""
    for n in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if n in med and gui:
            line += f"  {n[3:]}/cyc {med[n] / (gui * 256):6.3f}"          # per CU and cycle
    print(line)
