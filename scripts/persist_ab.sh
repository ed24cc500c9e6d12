#!/bin/bash
# A/B of the persistent tile loop: STORM_CONV_PERSIST=0 (one workgroup per tile) / 1 (resident grid) / 2 (2x resident)
python -m pytest tests/test_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -1
for p in 0 1 2; do for v in 0 2; do
  echo "persist $p variant $v: $(STORM_CONV_PERSIST=$p STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
bash scripts/ab_env.sh STORM_CONV_PERSIST 0 1 2 1 0
