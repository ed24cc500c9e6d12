#!/bin/bash
# bench lines beyond the default: fp16, fp32, configs[3] (ncsnpplarge 8 s, 50 PC + 1 corrector), configs[4]-style (ragged 2-10 s, ODE, fp16)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_net.py tests/test_ops.py -m gpu -q -x -s -k "fp16 or f16" 2>&1 | grep -v "^$" | tail -5
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_r02_$tag.json 2> gpurun_out/bench_r02_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_r02_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'], (r.get('roofline') or {}).get('frac'))" || tail -3 gpurun_out/bench_r02_$tag.err; }
run fp16 --precision fp16 --steps 1 --warmup 1 --no-cpu-baseline
run fp32 --precision fp32 --steps 1 --warmup 0 --no-cpu-baseline
run cfg3 --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline
run cfg4 --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run cfg4pc --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
