#!/bin/bash
# round 2: chunk-unrolled conv_pipe - parity on the GPU, race screen, probe timings, ablations (profiling build)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_ops.py -m gpu -q -x -k "conv" 2>&1 | tail -2
timeout 600 python tools/conv_check.py 2>&1 | tail -2
for i in 1 2; do for v in 0 3; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ') | $(PROBE_SHORTCUT=1 STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^sc' | tr '\n' ' ')"
done; done
export STORM_LIB=$PWD/storm_amd/csrc/libstorm_hip_prof.so
for abl in 0 8 16 32 128 184; do
  echo "abl $abl: $(STORM_CONV_ABLATE=$abl STORM_CONV_VARIANT=3 timeout 300 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
