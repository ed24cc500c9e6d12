#!/bin/bash
# round 6, visit q: the pyramids at 1024 threads + the grouped GroupNorm-apply / FIR launches: the -m gpu suite, the bench line with its per-op table,
# rocprofv3 kernel stats of the bench command and of the grouped stream, the configs[4] lines
TAG=${1:-r06q}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --ops-json gpurun_out/ops_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; head -c 300 gpurun_out/bench_$TAG.json; echo
python - <<PY
import json
r = json.load(open("gpurun_out/bench_$TAG.json"))
print("value", round(r["value"], 3), "frac", round(r["roofline"]["frac"], 4), "kinds", r["roofline"]["ms_by_op_kind"])
print("hbm family", r["roofline_hbm"]["family_ms_per_nfe"], {k: (v["ms_per_nfe"], v["tb_per_s"]) for k, v in r["roofline_hbm"]["by_kernel"].items() if "pyramid" in k})
for k, v in r.get("other_configs", {}).items():
    print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a not in ("workload", "grouped")})
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-traffic > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_rocprofv3_kernel_stats.csv && grep -i "pyramid\|fir_\|pack_input\|output_head" "$f" | cut -c1-160
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/prof_${TAG}_stream
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stream -o trace -- python bench.py --stream 32 --precision fp16 --batch 16 --steps 1 --warmup 1 --N 4 --no-cpu-baseline > gpurun_out/prof_bench_${TAG}_stream.json 2> gpurun_out/prof_${TAG}_stream.err
f=$(find gpurun_out/prof_${TAG}_stream -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_stream_grouped_kernel_stats.csv && head -30 "$f" | cut -c1-150
find gpurun_out/prof_${TAG}_stream -name "*kernel_trace.csv" -delete
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG}_$tag.json 2> gpurun_out/bench_${TAG}_$tag.err; python -c "
import json,sys; r=json.load(open('gpurun_out/bench_${TAG}_$tag.json')); print('$tag', round(r['value'],3), r['unit'], 'ms/step', round(r['ms_per_step'],1), 'nfe', r['config']['nfe_per_utterance'], 'calls/rows', r.get('grouped_calls_rows'))" || tail -5 gpurun_out/bench_${TAG}_$tag.err; }
run cfg4pc --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
run cfg4pc_seq --stream 32 --precision fp16 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-group
run cfg4 --stream 32 --sampler ode --precision fp16 --batch 16 --steps 1 --warmup 0 --no-cpu-baseline
run b1 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs
