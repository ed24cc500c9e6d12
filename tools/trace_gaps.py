#!/usr/bin/env python
"""Busy time against wall time per score evaluation from a rocprofv3 --kernel-trace CSV (the small-batch regime: what do the gaps
between ~120 short launches cost?).  An evaluation = the kernels between two consecutive output_head_kernel launches.

    python tools/trace_gaps.py <rocprofv3 output dir> [--skip N]   ->  one line of JSON
"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 60
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print(json.dumps({"error": "no kernel_trace.csv under " + d}))
        return
    rows = []
    with open(files[0]) as f:
        rd = csv.DictReader(f)
        col = lambda part: next(c for c in rd.fieldnames if part in c.lower())
        cs, ce, cn = col("start"), col("end"), col("kernel_name")
        for r in rd:
            rows.append((int(r[cs]), int(r[ce]), r[cn]))
    rows.sort()
    evals, cur = [], []
    for s, e, n in rows:
        cur.append((s, e, n))
        if "output_head_kernel" in n:
            evals.append(cur)
            cur = []
    evals = evals[skip:]                                   # (warm-up steps: first launches load code objects)
    if len(evals) < 2:
        print(json.dumps({"error": "fewer than two evaluations after the skip"}))
        return
    if "--timeline" in sys.argv:                           # one evaluation as a table: start offset, duration, gap before (us), kernel
        ev = evals[len(evals) // 2]
        t0, prev = ev[0][0], ev[0][0]
        out = sys.argv[sys.argv.index("--timeline") + 1]
        with open(out, "w") as f:
            for s, e, n in ev:
                f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {max(0, s - prev) / 1e3:6.1f}  {n.replace('storm::', '').replace('void ', '')[:90]}\n")
                prev = e
    busy = [sum(e - s for s, e, _ in ev) for ev in evals]
    span = [evals[k][-1][1] - evals[k - 1][-1][1] for k in range(1, len(evals))]     # head end -> next head end (sampler update kernels included)
    inner = [ev[-1][1] - ev[0][0] for ev in evals]                                     # first kernel start -> head end
    gaps = {}
    for ev in evals:
        for a, b in zip(ev, ev[1:]):
            g = max(0, b[0] - a[1])
            k = b[2].split("<")[0].replace("storm::", "").replace("void ", "")
            gaps.setdefault(k, [0, 0])
            gaps[k][0] += g
            gaps[k][1] += 1
    med = lambda v: sorted(v)[len(v) // 2]
    top = sorted(gaps.items(), key=lambda kv: -kv[1][0])[:8]
    print(json.dumps({
        "evaluations": len(evals), "kernels_per_evaluation": med([len(ev) for ev in evals]),
        "busy_us_median": med(busy) / 1e3, "first_kernel_to_head_end_us_median": med(inner) / 1e3,
        "head_end_to_head_end_us_median": med(span) / 1e3,
        "gap_us_before_kernel_per_evaluation": {k: round(v[0] / len(evals) / 1e3, 2) for k, v in top},
        "mean_gap_us_before_kernel": {k: round(v[0] / max(v[1], 1) / 1e3, 2) for k, v in top},
    }))


if __name__ == "__main__":
    main()
