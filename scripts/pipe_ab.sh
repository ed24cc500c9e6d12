#!/bin/bash
CHECK_VARIANTS=3 timeout 600 python tools/conv_check.py 2>&1 | grep -v "^ok" | tail -2
for i in 1 2 3; do for v in 0 3; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
