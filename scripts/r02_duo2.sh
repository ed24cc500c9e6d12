#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 1; do STORM_CONV_DUO=$v python bench.py --no-cpu-baseline --steps 1 --warmup 1 --ops-json gpurun_out/ops_duo$v.json > gpurun_out/bench_duo$v.json 2>/dev/null; done
python - <<PY
import json
r0 = json.load(open("gpurun_out/ops_duo0.json")); r1 = json.load(open("gpurun_out/ops_duo1.json"))
for a, b in zip(r0, r1):
    if a["code"] == 4 and a.get("kernel") != b.get("kernel"):
        print(a["idx"], a["H"], a["W"], a["cin"], a["taps"], "igemm %.3f ms %4.0f TF | duo %.3f ms %4.0f TF" % (a["ms"], a["flops"] / a["ms"] / 1e9, b["ms"], b["flops"] / b["ms"] / 1e9))
PY
