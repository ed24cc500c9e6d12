#!/bin/bash
# One GPU-box visit (round 5 form): smoke, the -m gpu suite, the default bench line (roofline + roofline_hbm + cpu_baseline), rocprofv3
# kernel stats of the same bench command (socket power / clock sampled beside the bench line), HBM-traffic PMC passes (-> conv_traffic.json), SQ counters; the other configs' lines.
TAG=${1:-r04a}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
grep -h "rel-L2\|nfev\|vs reference" gpurun_out/pytest_gpu_$TAG.log | head -20
python tools/power_trace.py gpurun_out/power_$TAG.csv & PT=$!
timeout 900 python bench.py --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; head -c 400 gpurun_out/bench_$TAG.json; echo
kill $PT; sleep 0.3; python tools/power_trace.py --summary gpurun_out/power_$TAG.csv | tee gpurun_out/power_summary_$TAG.txt
if [ "$2" != "short" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-traffic > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
bash scripts/pmc_bench.sh pmcb_$TAG 2>&1 | tail -4
rm -rf gpurun_out/pmc_$TAG; mkdir -p gpurun_out/pmc_$TAG
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_$TAG/sq -o p -- python bench.py --steps 1 --warmup 0 --N 2 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$TAG/sq.log 2>&1
python tools/pmc_cycles.py gpurun_out/pmc_$TAG/sq "$TAG" "conv_,gn_apply,attention,gn_finalize,fir_,pyramid" | tee gpurun_out/pmc_summary_$TAG.txt | head -30
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
fi
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'], (r.get('roofline') or {}).get('frac'))" || tail -3 gpurun_out/bench_${TAG}_$tag.err; }
run fp16 --precision fp16 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-traffic
run cfg3 --backbone ncsnpplarge --seconds 8 --N 50 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline
run cfg4 --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run cfg4pc --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
# the small-call regime (the reference's own operating point is ONE utterance per call): latency / RTF lines with their per-op tables
run b1 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b1.json
run b2 --batch 2 --steps 4 --warmup 2 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b2.json
run b4 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --ops-json gpurun_out/ops_${TAG}_b4.json
