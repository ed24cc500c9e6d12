#!/bin/bash
# round 6, visit i: the enhance_stream GPU tests, then the driver's own bench command (with other_configs and the in-run traffic measurement), timed
TAG=${1:-r06i}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "enhance_stream or forward_group or conv_group" > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -4 gpurun_out/pytest_gpu_$TAG.log
SECONDS=0; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench wall ${SECONDS}s"
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r06i.json"))
rf = r["roofline"]
print("value", round(r["value"], 3), "roofline frac", round(rf["frac"], 4), "traffic", rf["traffic"], "|", rf["traffic_source"][:120])
print("hbm", r["roofline_hbm"]["family_ms_per_nfe"], r["roofline_hbm"]["traffic"], "power", rf.get("power"))
for k, v in r.get("other_configs", {}).items():
    print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a not in ("workload", "grouped")})
PY
tail -3 gpurun_out/bench_$TAG.err
