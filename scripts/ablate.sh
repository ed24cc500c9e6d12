#!/bin/bash
# conv kernel ablation study on the probe shapes: STORM_CONV_ABLATE bits 1=no MFMA 2=no LDS fragment reads
# 4=no global loads after the first chunk 8=no epilogue
for v in 0 2; do for abl in 0 1 2 4 8 6 5; do
  echo "variant $v ablate $abl: $(STORM_CONV_VARIANT=$v STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 3 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
