#!/bin/bash
# One GPU-box visit: parity tests, default bench line, rocprofv3 kernel-trace of the same bench command.
# Usage (from the repo root, through gpurun):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1
tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
cat gpurun_out/prof_bench_$TAG.json
find gpurun_out/prof_$TAG -name "*stats*" | head
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
# keep the big per-dispatch trace out of the pull-back budget
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
