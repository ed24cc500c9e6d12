#!/bin/bash
# conv_duo.hip on the GPU: forced-variant parity tests, then the A/B probe of the <= 128-cout layers (tools/probe128.py)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -q -x -m gpu -k "test_conv_pipelined_128cout_kernel and hip and 5" -p no:cacheprovider 2>&1 | tail -5
for fl in "" "--nogn"; do
  echo "== flags: $fl"
  timeout 300 python tools/probe128.py --reps 5 --modes duo,p128,igemm $fl 2>&1 | grep -v "^$" | head -8
done
