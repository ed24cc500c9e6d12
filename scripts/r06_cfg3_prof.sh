#!/bin/bash
# round 6: configs[3] (ncsnpplarge, 8 x 8 s, 100 evaluations) per kernel (rocprofv3 --kernel-trace --stats) and per op with the round's last library
TAG=${1:-r06u}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_$TAG.log 2>&1
rm -rf gpurun_out/prof_${TAG}_cfg3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_cfg3 -o trace -- python bench.py --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-traffic --ops-json gpurun_out/${TAG}_cfg3_ops.json > gpurun_out/${TAG}_cfg3_prof_bench.json 2> gpurun_out/${TAG}_cfg3_prof.err
f=$(find gpurun_out/prof_${TAG}_cfg3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_cfg3_rocprofv3_kernel_stats.csv && head -14 "$f" | cut -c1-160
find gpurun_out/prof_${TAG}_cfg3 -name "*kernel_trace.csv" -delete
