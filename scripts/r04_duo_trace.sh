#!/bin/bash
# wave timelines of conv_duo.hip (profiling library): full kernel and the work-skipping instantiations, plain and fused operand
for abl in 64 65 66 67; do echo "== abl $abl (plain operand, 128 -> 128)"; timeout 120 python tools/duo_trace.py --abl $abl 2>&1 | grep -v amdgpu.ids; done
echo "== abl 64, fused operand"; timeout 120 python tools/duo_trace.py --abl 64 --gn 2>&1 | grep -v amdgpu.ids
echo "== abl 68, fused operand, no transform"; timeout 120 python tools/duo_trace.py --abl 68 --gn 2>&1 | grep -v amdgpu.ids | head -12
echo "== abl 64, 128 x 256"; timeout 120 python tools/duo_trace.py --abl 64 --H 128 --W 256 2>&1 | grep -v amdgpu.ids | head -12
