#!/bin/bash
# PMC passes (each counter group in its own rocprofv3 run, --kernel-trace only) on the kernel probe.
TAG=${1:-pmc}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
python tools/conv_probe.py --reps 2 > gpurun_out/$TAG/probe.log 2>&1; cat gpurun_out/$TAG/probe.log
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$TAG/$name -o p -- python tools/conv_probe.py --reps 1 > gpurun_out/$TAG/$name.log 2>&1; find gpurun_out/$TAG/$name -name "*counter_collection.csv" | head -2; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
du -sh gpurun_out/$TAG
