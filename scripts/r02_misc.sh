#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py tests/test_net.py -m gpu -q --tb=short -k "groupnorm or resblock or fir or tiny_net or fp16" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --ops-json gpurun_out/ops_misc.json > gpurun_out/bench_misc.json 2> gpurun_out/bench_misc.err
python - <<PY
import json
b = json.load(open("gpurun_out/bench_misc.json")); rows = json.load(open("gpurun_out/ops_misc.json"))
print(round(b["value"], 3), "utt/s | ", b["roofline"]["ms_by_op_kind"], [(r["idx"], round(r["ms"], 3)) for r in rows if r["code"] == 6])
print(b["roofline"]["conv3x3_by_kernel"])
PY
