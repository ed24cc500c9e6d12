#!/bin/bash
# configs[3] (ncsnpplarge, 8 x 8 s, 100 evaluations) per kernel and per op with the final library
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/b.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -o trace -- python bench.py --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --ops-json gpurun_out/r04e_cfg3_ops.json > gpurun_out/r04e_cfg3_prof_bench.json 2> gpurun_out/prof_cfg3.err
f=$(find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04e_cfg3_rocprofv3_kernel_stats.csv && head -12 "$f" | cut -c1-200
find gpurun_out/prof_cfg3 -name "*kernel_trace.csv" -delete
