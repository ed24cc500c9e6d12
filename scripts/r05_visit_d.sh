#!/bin/bash
# Round 5, visit d: A/B of the attention key split and of the re-derived dispatch ladder at one / two / four utterances per call and on the
# ragged stream, then the standard visit (scripts/gpu_round.sh: full -m gpu suite with the parity artefact, bench line with power, rocprofv3, PMC).
TAG=${1:-r05d}
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-roofline --no-h2d "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', 'utt/s', round(r['value'],3), 'ms/step', round(r['ms_per_step'],1), 'ms/nfe', round(r['ms_per_nfe_batch'] or 0,3), 'nfe', r['config']['nfe_per_utterance'])" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
for b in 1 2 4; do
STORM_ATTN_SPLIT=1 line b${b}_noattnsplit --batch $b --steps 4 --warmup 2
line b${b}_new --batch $b --steps 4 --warmup 2
done
STORM_ATTN_SPLIT=1 line cfg4pc_noattnsplit --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0
line cfg4pc_new --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0
bash scripts/gpu_round.sh $TAG
