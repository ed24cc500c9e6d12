#!/bin/bash
# round 4: split-K for the few-tile layers - GPU parity (ops + the full-width / ncsnpplarge goldens through the planner), the few-tile
# probe (v-1 = the dispatcher: split-K below 65 workgroups; v9 = the unsplit 128-cout tile), configs[3] and configs[1] lines
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/splitk_build.log 2>&1
timeout 900 python -m pytest tests/test_ops.py tests/test_net.py -q -x -m gpu -k "split_k or few_pixel or full_width or large_net or bench_shape or transparent or pipe128_bench" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/probe_small.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_probe_small_splitk.txt
timeout 600 python bench.py --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_splitk_cfg3.json > gpurun_out/bench_splitk_cfg3.json 2> gpurun_out/bench_splitk_cfg3.err; head -c 300 gpurun_out/bench_splitk_cfg3.json; echo
STORM_CONV_VARIANT=-1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_splitk.json > gpurun_out/bench_splitk.json 2> gpurun_out/bench_splitk.err; head -c 300 gpurun_out/bench_splitk.json; echo
