#!/bin/bash
# round 6, visit a: the new -m gpu tests (upfirdn2d ABI, Lightning-format checkpoints) and the default bench line with `other_configs`
TAG=r06a
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 900 python -m pytest tests -m gpu -q --tb=short -k "upfirdn or lightning or checkpoint" > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; head -c 300 gpurun_out/bench_$TAG.json; echo
echo "bench wall ${SECONDS}s"
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r06a.json"))
print("value", r["value"], "roofline", r["roofline"]["frac"], "hbm", r["roofline_hbm"]["family_ms_per_nfe"])
for k, v in r.get("other_configs", {}).items():
    print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "workload"})
PY
