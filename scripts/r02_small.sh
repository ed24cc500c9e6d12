#!/bin/bash
# the 32x64 (and 64x128) layers: which kernel family fills the chip best?
export TMPDIR=/tmp
for v in 0 1 2 3; do echo "== STORM_CONV_VARIANT=$v"; PROBE_SMALL=1 STORM_CONV_VARIANT=$v timeout 120 python tools/conv_probe.py --reps 10 2>&1 | grep -E "^c[0-9]" ; done
