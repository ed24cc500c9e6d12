#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py -m gpu -q --tb=short -k "conv or pipelined" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --ops-json gpurun_out/ops_misc.json > gpurun_out/bench_misc.json 2> gpurun_out/bench_misc.err
python - <<PY
import json
b = json.load(open("gpurun_out/bench_misc.json")); rows = json.load(open("gpurun_out/ops_misc.json"))
print(round(b["value"], 3), "utt/s | ", b["roofline"]["ms_by_op_kind"])
print(b["roofline"]["conv3x3_by_kernel"])
print([(r["idx"], round(r["ms"], 3), round(r["flops"] / r["ms"] / 1e9)) for r in rows if r["code"] == 4 and "pipe" in r.get("kernel", "")])
PY
rm -rf gpurun_out/pmc_misc; mkdir -p gpurun_out/pmc_misc
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_misc/sq -o p -- python tools/probe128.py --only 7 --reps 2 > gpurun_out/pmc_misc/sq.log 2>&1
python tools/pmc_cycles.py gpurun_out/pmc_misc/sq "256->256@128x256 fused"
find gpurun_out/pmc_misc -name "*kernel_trace.csv" -delete
