#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do for sk in 0 1 8 16; do
  echo "skew_min_tiles $sk: $(STORM_CONV_SKEW=$sk STORM_CONV_VARIANT=3 python tools/conv_probe.py --reps 10 2>&1 | grep -E '^c256|^c512' | tr '\n' ' ') | $(STORM_CONV_SKEW=$sk PROBE_SHORTCUT=1 STORM_CONV_VARIANT=3 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^sc256_256x512 .(plain|gn.stats).:' | tr '\n' ' ')"
done; done
