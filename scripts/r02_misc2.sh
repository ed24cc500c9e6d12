#!/bin/bash
export TMPDIR=/tmp
for i in 7 8 9; do
  timeout 200 python tools/probe128.py --only $i --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-110
  timeout 200 python tools/probe128.py --only $i --reps 5 --nogn 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
