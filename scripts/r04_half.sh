#!/bin/bash
# round 4: conv_pipe_kernel<T, 128, 8> (choose_variant 9) - GPU parity of the forced variant, the few-tile probe, a short bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/half_build.log 2>&1
timeout 900 python -m pytest tests/test_ops.py -q -x -m gpu -k "half_tile or (test_conv_pipelined_kernels and hip)" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/probe_small.py 2>&1 | tee gpurun_out/r04_probe_small_half.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_half.json > gpurun_out/bench_half.json 2> gpurun_out/bench_half.err; head -c 300 gpurun_out/bench_half.json; echo
timeout 600 python bench.py --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_half_cfg3.json 2> gpurun_out/bench_half_cfg3.err; head -c 300 gpurun_out/bench_half_cfg3.json; echo
