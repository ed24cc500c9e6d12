#!/bin/bash
# Round 5: kernel timeline of ONE one-utterance evaluation with the final library (rocprofv3 kernel trace -> tools/trace_gaps.py --timeline)
TAG=${1:-r05p}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}_b1 -o trace -- python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/prof_bench_${TAG}_b1.json 2> gpurun_out/prof_${TAG}_b1.err
python tools/trace_gaps.py gpurun_out/prof_${TAG}_b1 --timeline gpurun_out/timeline_${TAG}_b1.txt | tee gpurun_out/gaps_${TAG}_b1.json
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
