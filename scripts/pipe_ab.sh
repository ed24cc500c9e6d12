#!/bin/bash
for abl in 0 4 8 16 32 24 48 56; do
  echo "pipe ablate $abl: $(STORM_CONV_VARIANT=3 STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
