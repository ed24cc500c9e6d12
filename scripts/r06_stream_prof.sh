#!/bin/bash
# round 6: per-kernel time of the grouped configs[4]-style stream (PC sampler, 4 reverse steps = 8 evaluations of the 32 utterances)
TAG=${1:-r06g}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_$TAG.log 2>&1
for mode in grouped seq; do
  extra=""; [ $mode = seq ] && extra="--no-group"
  rm -rf gpurun_out/prof_${TAG}_$mode
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$mode -o trace -- python bench.py --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 1 --N 4 --no-cpu-baseline $extra > gpurun_out/prof_bench_${TAG}_$mode.json 2> gpurun_out/prof_${TAG}_$mode.err
  f=$(find gpurun_out/prof_${TAG}_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_stream_${mode}_kernel_stats.csv && head -25 "$f" | cut -c1-200
  find gpurun_out/prof_${TAG}_$mode -name "*kernel_trace.csv" -delete
done
