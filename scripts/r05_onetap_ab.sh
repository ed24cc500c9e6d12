#!/bin/bash
# Round 5: the one-tap chunks of conv_pipe as a ring of 32-channel half images (new) against the whole-chunk fetch of rounds 2 - 4 (old library:
# the same tree with conv_pipe.hip of the previous commit, built into libstorm_hip_onetap_old.so): parity on the GPU, then the bench
# alternating between the two libraries, per-op tables kept.
TAG=${1:-r05c}
mkdir -p gpurun_out
export TMPDIR=/tmp
STORM_PARITY_JSON=gpurun_out/parity_${TAG}.json timeout 1500 python -m pytest tests -m gpu -q --tb=short -k "conv_pipelined_kernels or conv_split_k or half_tile or full_size_kernel or bench_shape or batch_independence or split_k_layers or small_call or (full_width_60 and f14 and bf16)" > gpurun_out/pytest_${TAG}.log 2>&1; tail -4 gpurun_out/pytest_${TAG}.log
OLD=$PWD/storm_amd/csrc/libstorm_hip_onetap_old.so
line() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-h2d --steps 2 --warmup 1 "$@" --ops-json gpurun_out/ops_${TAG}_$tag.json > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); ops=json.load(open('gpurun_out/ops_${TAG}_$tag.json'))
one=[o for o in ops if o['code']==4 and o.get('taps')==[9,1]]
print('$tag', 'utt/s', round(r['value'],3), 'ms/nfe', round(r['ms_per_nfe_batch'],3), '[9,1] ops ms', round(sum(o['ms'] for o in one),3), {o['idx']: round(o['ms'],3) for o in one if o['idx'] in (19,87,91,95,103)})" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
for rep in 1 2 3; do
STORM_LIB=$OLD line old_$rep
line new_$rep
done
