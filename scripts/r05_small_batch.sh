#!/bin/bash
# Round 5, visit a: the small-batch regime (the reference's own operating point is ONE utterance per call, enhancement.py:66-72).
# bench lines at batch 1 / 2 / 4 with eager launches and with HIP-graph replay (alternating), the per-op table and the rocprofv3
# kernel trace of the batch-1 line (sum of kernel durations against the wall time per evaluation = what the gaps cost), the new GPU tests.
TAG=${1:-r05a}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_${TAG}_new.json timeout 900 python -m pytest tests -m gpu -q --tb=short -k "graph_replay or enhancement_cli or rccl_group_of_one or pipelined_128cout" > gpurun_out/pytest_new_$TAG.log 2>&1; tail -5 gpurun_out/pytest_new_$TAG.log
line() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', 'utt/s', round(r['value'],3), 'ms/step', round(r['ms_per_step'],1), 'ms/nfe', round(r['ms_per_nfe_batch'] or 0,3), 'rtf', r['rtf'] and round(r['rtf'],4), 'graph', r['graph'], 'nfe_ms_profiled', (r.get('roofline') or {}).get('nfe_ms_profiled'))" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
for rep in 1 2; do
line b1_eager_$rep --batch 1 --graph off --steps 5 --warmup 2 --no-roofline --no-h2d
line b1_graph_$rep --batch 1 --graph on --steps 5 --warmup 2 --no-roofline --no-h2d
done
line b1 --batch 1 --graph auto --steps 5 --warmup 2 --ops-json gpurun_out/ops_${TAG}_b1.json
line b2_eager --batch 2 --graph off --steps 4 --warmup 2 --no-roofline --no-h2d
line b2_graph --batch 2 --graph on --steps 4 --warmup 2 --ops-json gpurun_out/ops_${TAG}_b2.json
line b4_eager --batch 4 --graph off --steps 3 --warmup 1 --no-roofline --no-h2d
line b4_graph --batch 4 --graph on --steps 3 --warmup 1 --ops-json gpurun_out/ops_${TAG}_b4.json
line b16_graph --batch 16 --graph on --steps 1 --warmup 1 --no-roofline --no-h2d
line b16_eager --batch 16 --graph off --steps 1 --warmup 1 --no-roofline --no-h2d
for g in off on; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_b1_$g -o trace -- python bench.py --batch 1 --graph $g --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/prof_bench_${TAG}_b1_$g.json 2> gpurun_out/prof_${TAG}_b1_$g.err
f=$(find gpurun_out/prof_${TAG}_b1_$g -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
python tools/trace_gaps.py gpurun_out/prof_${TAG}_b1_$g | tee gpurun_out/gaps_${TAG}_b1_$g.json
done
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
line cfg4pc_eager --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --graph off
line cfg4pc_graph --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --graph on
