#!/bin/bash
# round 4: conv_pipe_kernel<T, 128, 8> (variant 9) on the <= 128-cout layers of the bench shape against conv_igemm / conv_pipe128 -
# every (layer, kernel) pair in its own process, alternating, 20 repetitions (sustained load: the part sits at its power cap)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/half_build.log 2>&1
out=gpurun_out/r04_half_fair_ab.txt; : > $out
for idx in 0 1 2 3 4; do
  for rep in 1 2; do
    for mode in igemm half p128; do
      timeout 120 python tools/probe128.py --reps 20 --only $idx --modes $mode 2>&1 | tail -1 | tee -a $out
    done
  done
done
