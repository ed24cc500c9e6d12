#!/bin/bash
# finer ablation: 16 = no weight loads, 32 = no patch loads (after the first chunk), 8 = no epilogue, 40 = no patch + no epilogue
for v in 0 2; do for abl in 0 16 32 8 40; do
  echo "variant $v ablate $abl: $(STORM_CONV_VARIANT=$v STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 3 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
