#!/bin/bash
# Round 5, visit j: the rounds-aware dispatch ladder + regenerated table: full -m gpu suite, the bench line, small-call / configs[3] / configs[4] lines
TAG=${1:-r05j}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'ms/nfe', r['ms_per_nfe_batch'] and round(r['ms_per_nfe_batch'],3), 'nfe', r['config']['nfe_per_utterance'])" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
run b16 --ops-json gpurun_out/ops_$TAG.json
STORM_CONV_TABLE=0 run b16_notable --no-cpu-baseline --no-roofline --no-h2d
run b1 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b1.json
run b2 --batch 2 --steps 4 --warmup 2 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b2.json
run b4 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b4.json
run b8 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-h2d
run fp16 --precision fp16 --steps 1 --warmup 1 --no-cpu-baseline
run cfg3 --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline
run cfg4pc --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run cfg4 --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
