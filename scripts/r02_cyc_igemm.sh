#!/bin/bash
# cycle-based ablation of conv_igemm's 128-cout 3x3 instantiation (profiling build) on tools/conv_probe.py (c128_256x512 is the row to read)
export TMPDIR=/tmp
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
mkdir -p gpurun_out/cyci
for abl in ${ABLS:-0 8 1 2 3}; do
  rm -rf gpurun_out/cyci/a$abl
  STORM_CONV_ABLATE=$abl STORM_CONV_VARIANT=0 STORM_CONV_DMA=0 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/cyci/a$abl -o p -- python tools/conv_probe.py --reps 2 > gpurun_out/cyci/a$abl.log 2>&1
  python tools/pmc_cycles.py gpurun_out/cyci/a$abl "abl $abl"
done
find gpurun_out/cyci -name "*kernel_trace.csv" -delete
