#!/bin/bash
# conv_pipe128 A/B: who transforms (XF) x setprio, on one fused layer (256->128 @256x512x16).  HISTORICAL: the STORM_P128_MODE switch existed
# only in the experiment build that produced profiles/r02_pipe128_ab.txt (all variants within 2 %); the product kernel is XF = 0.
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in 0 1 2 3 4 5; do
  echo "== STORM_P128_MODE=$m (XF=$((m/2)) NOPRIO=$((m%2)))"
  STORM_P128_MODE=$m timeout 120 python tools/probe128.py --only 2 --reps 5 2>&1 | grep -v amdgpu.ids
  STORM_P128_MODE=$m timeout 120 python tools/probe128.py --only 2 --reps 5 --nogn 2>&1 | grep -v amdgpu.ids
done
