#!/bin/bash
# round 6 ("first the missing evidence"): work-skipping power ablations of the two kernels that run the <= 128-cout 3x3
# layers, each instantiation sustained at the socket cap (tools/power_probe.py: time at fixed power ~ energy).  Profiling library only.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m storm_amd.build --profiling > gpurun_out/build_prof.log 2>&1 || { tail -5 gpurun_out/build_prof.log; exit 1; }
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
OUT=gpurun_out/r06_power_ablations_128cout.txt
: > $OUT
for cin in 128 384 256; do
  echo "== conv_pipe128, ${cin} -> 128 @ 16 x 256 x 512, fused GroupNorm + SiLU operand" | tee -a $OUT
  timeout 300 python tools/power_probe.py --seconds 3 --modes p128 --cin $cin --abl 0,8,16,128,136,1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  echo "== conv_igemm (generic 128-cout tile, LDS-DMA weights), ${cin} -> 128 @ 16 x 256 x 512, fused operand" | tee -a $OUT
  timeout 300 python tools/power_probe.py --seconds 3 --modes igemm --cin $cin --abl 0,16,2,4,20,8,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
echo "== plain operand (no fused GroupNorm), 128 -> 128" | tee -a $OUT
timeout 200 python tools/power_probe.py --seconds 3 --modes p128 --cin 128 --nogn --abl 0,16,1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT
timeout 200 python tools/power_probe.py --seconds 3 --modes igemm --cin 128 --nogn --abl 0,2,8 2>&1 | grep -v amdgpu.ids | tee -a $OUT
