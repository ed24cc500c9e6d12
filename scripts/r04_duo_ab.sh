#!/bin/bash
# fair A/B of conv_duo vs conv_igemm vs conv_pipe128: each mode in its own process, alternating, 20 repetitions after 2 warm-ups
CASES=${CASES:-"0 1"}
for rep in 1 2; do
for m in igemm duo p128; do
  for c in $CASES; do timeout 120 python tools/probe128.py --reps 20 --modes $m --only $c $1 2>&1 | grep -v amdgpu.ids; done
done
done
