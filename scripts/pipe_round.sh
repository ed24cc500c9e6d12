#!/bin/bash
# GPU round for the pipelined conv kernels: parity tests, race screen, probe timings per variant
python -m pytest tests/test_ops.py -m gpu -x -q -k "pipelined" 2>&1 | tail -1
for v in 5; do STORM_CONV_VARIANT=$v python -m pytest tests/test_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -1; done
timeout 600 python tools/conv_check.py 2>&1 | grep -v "^ok" | tail -8
for v in 0 3 5; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
for abl in 8 16 32 56; do
  echo "variant 5 ablate $abl: $(STORM_CONV_VARIANT=5 STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
