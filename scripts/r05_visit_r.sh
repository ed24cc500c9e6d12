#!/bin/bash
# Round 5, visit r: the round's FINAL library - smoke, the full -m gpu suite, the determinism / batch-invariance screens (the shared-activation
# down-sampling kernel exchanges through double-buffered LDS: a race would show as a bit difference between repetitions), the default bench line
# with the per-op table (kernel names from the launchers)
TAG=${1:-r05r}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 600 python tools/stress_determinism.py 30 > gpurun_out/stress_b16_$TAG.txt 2>&1; tail -1 gpurun_out/stress_b16_$TAG.txt
timeout 600 python tools/stress_determinism.py 2 invariant > gpurun_out/stress_invariant_$TAG.txt 2>&1; tail -1 gpurun_out/stress_invariant_$TAG.txt
timeout 900 python bench.py --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cut -c1-400 gpurun_out/bench_$TAG.json
