#!/bin/bash
# conv_pipe128: is the fused GroupNorm transform the bound?  (plain vs fused operand; SQ counters of one layer)
TAG=${1:-r02d}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fused GN"; timeout 300 python tools/probe128.py 2>&1 | grep -v amdgpu.ids
echo "== plain";    timeout 300 python tools/probe128.py --nogn 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/pmc_$TAG; mkdir -p gpurun_out/pmc_$TAG
for mode in "" "--nogn"; do
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_$TAG/sq$mode -o p -- python tools/probe128.py --only 2 --reps 2 $mode > gpurun_out/pmc_$TAG/sq$mode.log 2>&1
  python tools/pmc_cycles.py gpurun_out/pmc_$TAG/sq$mode "fused$mode"
done
find gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
