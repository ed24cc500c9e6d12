#!/bin/bash
# round 4, final-visit extras: few-tile probes with the final dispatch (split-K / 128-cout tile), HBM yardsticks
export TMPDIR=/tmp
python tools/probe_small.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_probe_small.txt
python tools/probe_splitk_s.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_probe_splitk_s.txt
python tools/hbm_write_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_hbm_probe.txt
