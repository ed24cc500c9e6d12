#!/bin/bash
for i in 1 2; do
  echo "variant 0          : $(STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
  echo "variant 0 frag pipe: $(STORM_FRAG_PIPE=1 STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
