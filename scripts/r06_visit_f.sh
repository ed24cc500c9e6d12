#!/bin/bash
# round 6, visit f: the grouped launch (conv_pipe GROUP instantiation, storm_ncsnpp_forward_group, ScoreModel.enhance_stream) - parity on the GPU, then the
# configs[4]-style stream with and without grouping, then the default bench line (the headline must not move: the product kernel's source is shared)
TAG=r06f
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 1500 python -m pytest tests -m gpu -q --tb=short -k "conv_group or forward_group or configs4_real_shape or conv_pipelined_kernels or half_tile" > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -4 gpurun_out/pytest_gpu_$TAG.log
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'])" || tail -5 gpurun_out/bench_${TAG}_$tag.err; }
run pc_grouped --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
run pc_seq --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-group
run pc_grouped_bf16 --stream 32 --precision bf16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
run ode_grouped --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run ode_seq --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-group
run default --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs
