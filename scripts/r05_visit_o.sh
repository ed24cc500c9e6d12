#!/bin/bash
# Round 5, visit o: the batch-invariant mode at the bench shape (every row alone / in 2s / 4s / 8s == the row in the batch of 16, bf16 + fp16),
# the determinism screens on the final library, and the tuner's view of the final ladder + table (no --write: exceptions left are listed).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/stress_determinism.py 3 invariant > gpurun_out/stress_invariant.txt 2>&1; tail -2 gpurun_out/stress_invariant.txt
timeout 600 python tools/stress_determinism.py 20 > gpurun_out/stress_b16.txt 2>&1; tail -2 gpurun_out/stress_b16.txt
timeout 600 python tools/stress_determinism.py 20 small > gpurun_out/stress_small.txt 2>&1; tail -2 gpurun_out/stress_small.txt
timeout 900 python tools/tune_dispatch.py --reps 24 > gpurun_out/tune_dispatch_final.log 2>&1; grep -A40 "exceptions to the ladder" gpurun_out/tune_dispatch_final.log | head -40; grep -c "below --keep" gpurun_out/tune_dispatch_final.log
