#!/bin/bash
# round 4: time embedding + Dense_0 on a side stream (program.hip) - parity, then the alternating whole-bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/side_build.log 2>&1
timeout 900 python -m pytest tests/test_net.py tests/test_ops.py -q -x -m gpu -k "side_stream or few_pixel_tiles or transparent or bench_shape" -p no:cacheprovider 2>&1 | tail -5
bash scripts/ab_env.sh STORM_SIDE_STREAM 1 0 2>&1 | tee gpurun_out/r04_side_ab.txt
for i in 1 2; do for v in 1 0; do
  STORM_SIDE_STREAM=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('STORM_SIDE_STREAM=$v', round(r['value'],3), 'utt/s', round(r['ms_per_nfe_batch'],3), 'ms/nfe')"
done; done 2>&1 | tee -a gpurun_out/r04_side_ab.txt
