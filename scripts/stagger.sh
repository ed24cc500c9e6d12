#!/bin/bash
# STORM_CONV_STAGGER = (bit << 16) | sleep_iterations : workgroups with bit `bit` of blockIdx set start late
for bit in 0 3 4 8; do for n in 3 6 12; do
  val=$(( (bit << 16) | n ))
  echo "bit $bit n $n: $(STORM_CONV_VARIANT=0 STORM_CONV_STAGGER=$val python tools/conv_probe.py --reps 3 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
echo "none: $(STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 3 2>&1 | grep -E '^c' | tr '\n' ' ')"
