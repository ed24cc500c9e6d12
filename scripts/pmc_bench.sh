#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes over a short bench run.
TAG=${1:-pmcb}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$TAG/$c -o p -- python bench.py --steps 1 --warmup 0 --N 2 --no-cpu-baseline --no-roofline > gpurun_out/$TAG/$c.log 2>&1
done
f=$(find gpurun_out/$TAG/FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find gpurun_out/$TAG/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $f $w gpurun_out/$TAG/conv_traffic.json
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
