#!/bin/bash
# GPU visit for conv_pipe128.hip: parity of the new kernel, A/B probe on the bench-shape layers, net-level tests, bench line.
TAG=${1:-r02c}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py -m gpu -q --tb=short -k "pipelined_128cout" > gpurun_out/p128_tests_$TAG.log 2>&1; tail -3 gpurun_out/p128_tests_$TAG.log
timeout 300 python tools/probe128.py > gpurun_out/p128_probe_$TAG.log 2>&1; cat gpurun_out/p128_probe_$TAG.log | tail -12
timeout 900 python -m pytest tests/test_net.py -m gpu -q --tb=short -s > gpurun_out/p128_net_$TAG.log 2>&1; grep -E "rel-L2|passed|failed|Error" gpurun_out/p128_net_$TAG.log | tail -15
timeout 600 python bench.py --no-cpu-baseline --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; head -c 400 gpurun_out/bench_$TAG.json; echo
python - <<PY
import json
b = json.load(open("gpurun_out/bench_$TAG.json"))
print(b["value"], b["ms_per_step"], json.dumps(b["roofline"]["conv3x3_by_kernel"]), b["roofline"]["ms_by_op_kind"])
PY
