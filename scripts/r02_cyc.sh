#!/bin/bash
# cycle-based ablation of conv_pipe (profiling build): GRBM_GUI_ACTIVE per launch for every work-skipping instantiation
export TMPDIR=/tmp
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
mkdir -p gpurun_out/cyc
for abl in ${ABLS:-0 8 128 136 16 32 152 184 256}; do
  rm -rf gpurun_out/cyc/a$abl
  STORM_CONV_ABLATE=$abl STORM_CONV_VARIANT=3 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/cyc/a$abl -o p -- python tools/conv_probe.py --reps 2 > gpurun_out/cyc/a$abl.log 2>&1
  python tools/pmc_cycles.py gpurun_out/cyc/a$abl "abl $abl"
done
find gpurun_out/cyc -name "*kernel_trace.csv" -delete
