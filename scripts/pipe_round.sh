#!/bin/bash
# GPU round for the pipelined conv kernel: parity tests, race screen, probe timings per variant, wave timeline
STORM_CONV_VARIANT=3 python -m pytest tests/test_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -2
timeout 600 python tools/conv_check.py 2>&1 | tail -3
for v in 0 2 3; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
STORM_CONV_VARIANT=3 python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids | tail -60
