#!/bin/bash
# round-2 diagnostic: wave timeline of the chunk-unrolled conv_pipe kernel + ablation timings (profiling build)
export TMPDIR=/tmp
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
for sh in "--cin 256 --cout 256 --H 128 --W 256" "--cin 256 --cout 256 --H 256 --W 512"; do
  echo "=== trace variant 3 $sh"
  STORM_CONV_VARIANT=3 timeout 300 python tools/conv_trace.py $sh 2>&1 | tail -36
done
for abl in 0 8 16 32 128 184; do
  echo "abl $abl: $(STORM_CONV_ABLATE=$abl STORM_CONV_VARIANT=3 timeout 300 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
