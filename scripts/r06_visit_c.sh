#!/bin/bash
# round 6, visit c: fixture F16 on the GPU (three precisions) + the two-stream reproducer with the LDS-overlap characterisation
TAG=r06c
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_$TAG.log 2>&1
STORM_PARITY_JSON=gpurun_out/parity_${TAG}_f16.json timeout 1200 python -m pytest tests/test_model.py -m gpu -q --tb=short -s -k "configs4_real_shape" > gpurun_out/pytest_gpu_${TAG}_f16.log 2>&1; grep -h "F16\|passed\|failed\|Error\|assert" gpurun_out/pytest_gpu_${TAG}_f16.log | tail -12
hipcc --offload-arch=gfx950 -O3 tools/debug/concurrent_repro.hip -o gpurun_out/concurrent_repro && { timeout 600 gpurun_out/concurrent_repro 4 2>&1 | tee gpurun_out/r06_concurrent_repro.txt; }
rm -f gpurun_out/concurrent_repro
