#!/bin/bash
# GPU round for the pipelined conv kernel: parity tests, race screen, probe timings per variant
STORM_CONV_VARIANT=3 python -m pytest tests/test_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -2
timeout 600 python tools/conv_check.py 2>&1 | tail -3
for v in 0 3; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
for abl in 1 2 4; do
  echo "pipe ablate $abl: $(STORM_CONV_VARIANT=3 STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
