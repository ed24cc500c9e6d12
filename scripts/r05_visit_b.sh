#!/bin/bash
# Round 5, visit b: the small-call rules (split K for launches of <= 64 workgroups, short strips in the resampling kernels) A/B'd at one /
# two / four utterances per call, micro-batches of the ragged stream on 1 - 4 HIP streams (PC and ODE sampler), the full -m gpu suite
# with the measured parity errors as an artefact, and the default bench line.
TAG=${1:-r05b}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -5 gpurun_out/pytest_gpu_$TAG.log
line() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', 'utt/s', round(r['value'],3), 'ms/step', round(r['ms_per_step'],1), 'ms/nfe', round(r['ms_per_nfe_batch'] or 0,3), 'rtf', r['rtf'] and round(r['rtf'],4), 'nfe', r['config']['nfe_per_utterance'])" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
for rep in 1 2; do
STORM_SPLITK_SMALL=0 STORM_GN_ROWS=16 line b1_old_$rep --batch 1 --steps 5 --warmup 2 --no-roofline --no-h2d
line b1_new_$rep --batch 1 --steps 5 --warmup 2 --no-roofline --no-h2d
done
STORM_GN_ROWS=16 line b1_split_only --batch 1 --steps 5 --warmup 2 --no-roofline --no-h2d
line b1 --batch 1 --steps 5 --warmup 2 --ops-json gpurun_out/ops_${TAG}_b1.json
STORM_SPLITK_SMALL=0 STORM_GN_ROWS=16 line b2_old --batch 2 --steps 4 --warmup 2 --no-roofline --no-h2d
line b2 --batch 2 --steps 4 --warmup 2 --ops-json gpurun_out/ops_${TAG}_b2.json
STORM_SPLITK_SMALL=0 STORM_GN_ROWS=16 line b4_old --batch 4 --steps 3 --warmup 1 --no-roofline --no-h2d
line b4 --batch 4 --steps 3 --warmup 1 --ops-json gpurun_out/ops_${TAG}_b4.json
line b16 --ops-json gpurun_out/ops_${TAG}.json
STORM_SPLITK_SMALL=0 STORM_GN_ROWS=16 line cfg4pc_old --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0
for k in 1 2 3 4; do line cfg4pc_s$k --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --streams $k; done
line cfg4pc_s3_warm --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 1 --streams 3
for k in 1 3; do line cfg4ode_s$k --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --streams $k; done
