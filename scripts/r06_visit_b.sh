#!/bin/bash
# round 6, visit b: fixture F16 (configs[4] at its real shape) on the GPU, the whole -m gpu suite, the two-stream reproducer with the MFMA + LDS aggressor
TAG=r06b
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
STORM_PARITY_JSON=gpurun_out/parity_${TAG}_f16.json timeout 900 python -m pytest tests/test_model.py -m gpu -q --tb=short -s -k "configs4_real_shape" > gpurun_out/pytest_gpu_${TAG}_f16.log 2>&1; grep -h "F16\|passed\|failed" gpurun_out/pytest_gpu_${TAG}_f16.log | tail -8
STORM_PARITY_JSON=gpurun_out/parity_$TAG.json timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -5 gpurun_out/pytest_gpu_$TAG.log
hipcc --offload-arch=gfx950 -O3 tools/debug/concurrent_repro.hip -o gpurun_out/concurrent_repro && { for i in 1 2; do timeout 120 gpurun_out/concurrent_repro 6; done 2>&1 | tee gpurun_out/r06_concurrent_repro.txt; }
rm -f gpurun_out/concurrent_repro
