#!/bin/bash
# Round 5, visit m: the batch-invariant mode (STORM_BATCH_INVARIANT) - its GPU tests, and what it costs at 16 / 4 / 1 utterances per call
# (alternating with the default rules on the same box); the configs[4] ODE rows test with its counts in the parity artefact.
TAG=${1:-r05m}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 1200 python -m pytest tests -q -m gpu --tb=short -k "batch_invariant or configs4_ode or enhancement_cli or batch_independence or kernel_selection_is_transparent or split_k_layers" > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'ms/nfe', r['ms_per_nfe_batch'] and round(r['ms_per_nfe_batch'],3))" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
Q="--no-cpu-baseline --no-roofline --no-h2d"
run b16 --steps 2 --warmup 1 $Q
STORM_BATCH_INVARIANT=1 run b16_inv --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --ops-json gpurun_out/ops_${TAG}_b16_inv.json
run b16_again --steps 2 --warmup 1 $Q
STORM_BATCH_INVARIANT=1 run b16_inv_again --steps 2 --warmup 1 $Q
run b4 --batch 4 --steps 3 --warmup 1 $Q
STORM_BATCH_INVARIANT=1 run b4_inv --batch 4 --steps 3 --warmup 1 $Q
run b1 --batch 1 --steps 5 --warmup 2 $Q
STORM_BATCH_INVARIANT=1 run b1_inv --batch 1 --steps 5 --warmup 2 $Q
STORM_BATCH_INVARIANT=1 run cfg4pc_inv --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 $Q
