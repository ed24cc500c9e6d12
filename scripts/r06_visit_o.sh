#!/bin/bash
# round 6, visit o: the ODE sampler's row compaction (rows that reached eps leave their micro-batch) - ODE GPU tests, then the configs[4] stream
# with the rows leaving against the rows idling (same box, same library)
TAG=${1:-r06o}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 1200 python -m pytest tests -m gpu -q --tb=short -k "ode or stream" > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -4 gpurun_out/pytest_gpu_$TAG.log
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'], 'calls/rows', r.get('grouped_calls_rows'))" || tail -5 gpurun_out/bench_${TAG}_$tag.err; }
run ode_leave --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run ode_idle --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --ode-idle
run ode_leave2 --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run pc_grouped --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
