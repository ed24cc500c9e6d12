#!/bin/bash
# One GPU-box visit at the end of a work block: smoke, the -m gpu suite, the default bench line with cpu_baseline, rocprofv3
# kernel stats of the same bench command, HBM traffic PMC passes (-> conv_traffic.json), SQ counters of the dominant kernel.
TAG=${1:-r02b}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -2 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json | head -c 600; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
bash scripts/pmc_bench.sh pmcb_$TAG 2>&1 | tail -6
rm -rf gpurun_out/pmc_$TAG; mkdir -p gpurun_out/pmc_$TAG
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_$TAG/sq -o p -- python bench.py --steps 1 --warmup 0 --N 2 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$TAG/sq.log 2>&1
python tools/pmc_cycles.py gpurun_out/pmc_$TAG/sq "bench"
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
