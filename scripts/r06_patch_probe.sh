#!/bin/bash
# round 6: where does conv_pipe128's patch path go - the memory system or the LDS-DMA / transform mechanics?  (STORM_CONV_ABLATE=256: hot-region DMA)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m storm_amd.build --profiling > gpurun_out/build_prof.log 2>&1 || { tail -5 gpurun_out/build_prof.log; exit 1; }
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
OUT=gpurun_out/r06_patch_probe.txt
: > $OUT
for cin in 384 128; do
  echo "== conv_pipe128, ${cin} -> 128 @ 16 x 256 x 512: fused operand, then plain operand" | tee -a $OUT
  timeout 300 python tools/power_probe.py --seconds 3 --modes p128 --cin $cin --abl 0,256,128 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  timeout 300 python tools/power_probe.py --seconds 3 --modes p128 --cin $cin --nogn --abl 0,256,128 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
