#!/bin/bash
# round 6, visit h: the whole -m gpu suite with the grouped evaluation in the library, then the configs[4]-style stream (grouped convolutions + grouped
# GroupNorm finalizes) against one micro-batch after the other
TAG=${1:-r06h}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -6 gpurun_out/pytest_gpu_$TAG.log
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'])" || tail -5 gpurun_out/bench_${TAG}_$tag.err; }
run pc_grouped --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
run pc_seq --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-group
run ode_grouped --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
