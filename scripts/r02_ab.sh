#!/bin/bash
export TMPDIR=/tmp
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
for i in 1 2 3; do for abl in 0 4096; do
  echo "abl $abl: $(STORM_CONV_ABLATE=$abl STORM_CONV_VARIANT=3 timeout 300 python tools/conv_probe.py --reps 10 2>&1 | grep -E '^c' | tr '\n' ' ')"
done; done
