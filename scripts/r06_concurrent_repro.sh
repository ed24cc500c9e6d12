#!/bin/bash
# round 6: the standalone two-stream reproducer (tools/debug/concurrent_repro.hip) - no Python, no torch
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/debug/concurrent_repro.hip -o gpurun_out/concurrent_repro || exit 1
for i in 1 2; do timeout 120 gpurun_out/concurrent_repro 5; done 2>&1 | tee gpurun_out/r06_concurrent_repro.txt
rm -f gpurun_out/concurrent_repro
