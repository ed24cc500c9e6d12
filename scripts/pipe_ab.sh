#!/bin/bash
for abl in 0 1 2; do
  echo "pipe ablate $abl: $(STORM_CONV_VARIANT=3 STORM_CONV_ABLATE=$abl python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
for ex in 0 2; do STORM_TRACE_EXTRA=$ex STORM_CONV_VARIANT=3 python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids | sed -n 1,24p; done
