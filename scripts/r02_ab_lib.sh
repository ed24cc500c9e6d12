#!/bin/bash
# same-box A/B of two library builds over the bench's op-by-op profile (STORM_LIB selects the library)
export TMPDIR=/tmp; mkdir -p gpurun_out
for tag in new old new old; do
  lib=$PWD/storm_amd/csrc/libstorm_hip.so; [ $tag = old ] && lib=$PWD/storm_amd/csrc/libstorm_hip_old.so
  STORM_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --ops-json gpurun_out/ops_ab_$tag.json > gpurun_out/bench_ab_$tag.json 2> gpurun_out/bench_ab_$tag.err
  python - <<PY
import json
b = json.load(open("gpurun_out/bench_ab_$tag.json"))
k = b["roofline"]["conv3x3_by_kernel"]
print("$tag", round(b["value"], 3), "utt/s", {n.split("<")[0][7:]: (v["ms_per_nfe"], v["tflops"]) for n, v in k.items()})
PY
done

