#!/bin/bash
TAG=${1:-r02e}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py -m gpu -q --tb=short -k "pipelined_128cout" 2>&1 | tail -2
echo "== fused GN"; timeout 300 python tools/probe128.py 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/pmc_$TAG; mkdir -p gpurun_out/pmc_$TAG
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$TAG/sq -o p -- python tools/probe128.py --only 2 --reps 2 > gpurun_out/pmc_$TAG/sq.log 2>&1
python tools/pmc_cycles.py gpurun_out/pmc_$TAG/sq "fused"
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
timeout 600 python bench.py --no-cpu-baseline --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python - <<PY
import json
b = json.load(open("gpurun_out/bench_$TAG.json"))
print(b["value"], b["ms_per_step"], json.dumps(b["roofline"]["conv3x3_by_kernel"]), b["roofline"]["ms_by_op_kind"])
PY
