#!/bin/bash
# Round 5: rocprofv3 kernel trace of the one-utterance bench line with the final library (per-kernel stats + tools/trace_gaps.py)
TAG=${1:-r05f}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_b1 -o trace -- python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/prof_bench_${TAG}_b1.json 2> gpurun_out/prof_${TAG}_b1.err
f=$(find gpurun_out/prof_${TAG}_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
python tools/trace_gaps.py gpurun_out/prof_${TAG}_b1 | tee gpurun_out/gaps_${TAG}_b1.json
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
