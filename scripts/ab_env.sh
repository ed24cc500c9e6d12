#!/bin/bash
# A/B a kernel variant selected by an environment variable:  bash scripts/ab_env.sh VAR v1 v2 ...
VAR=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $VAR=$v python bench.py --steps 1 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_${VAR}_$v.json > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
  python - <<PY
import json
r = json.load(open("gpurun_out/ab_${VAR}_$v.json"))
print("$VAR=$v", "utt/s %.3f" % r["value"], "ms/nfe %.2f" % r["ms_per_nfe_batch"], "conv TF %.0f" % r["roofline"]["achieved"], r["roofline"]["ms_by_op_kind"])
PY
done
