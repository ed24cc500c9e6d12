#!/bin/bash
# round 6: the first chunk's fused GroupNorm transform at a tile's start shared out over all eight waves (conv_pipe / conv_pipe128) against the
# four lagging waves (libstorm_hip_old.so = the previous commit), alternating processes on one box: the layer probe, then the whole bench
mkdir -p gpurun_out
export TMPDIR=/tmp
OLD=$PWD/storm_amd/csrc/libstorm_hip_old.so; NEW=$PWD/storm_amd/csrc/libstorm_hip.so
OUT=gpurun_out/r06_partb_ab.txt; : > $OUT
for rep in 1 2 3; do
  for which in old new; do
    lib=$OLD; [ $which = new ] && lib=$NEW
    echo "== probe $which rep $rep" | tee -a $OUT
    STORM_LIB=$lib timeout 600 python tools/probe128.py --reps 30 --modes p128,pipe 2>&1 | grep -v amdgpu.ids | cut -c1-100 | tee -a $OUT
  done
done
for rep in 1 2 3; do
  for which in old new; do
    lib=$OLD; [ $which = new ] && lib=$NEW
    STORM_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-traffic > gpurun_out/ab_partb_${which}_$rep.json 2> gpurun_out/ab_partb_${which}_$rep.err
    python - <<PY | tee -a $OUT
import json
r = json.load(open("gpurun_out/ab_partb_${which}_$rep.json"))
k = r["roofline"]
print("bench $which rep $rep", "utt/s %.3f" % r["value"], "ms/nfe %.3f" % r["ms_per_nfe_batch"], "conv %.3f ms" % k["ms_by_op_kind"]["conv"], {n.split("::")[1][:28]: v["ms_per_nfe"] for n, v in k["conv3x3_by_kernel"].items()})
PY
  done
done
