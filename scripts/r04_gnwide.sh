#!/bin/bash
# round 4: GroupNorm + FIR resampling kernels with 32 slots (512 contiguous bytes) of a pixel per workgroup instead of 8 - parity, A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/b.log 2>&1
timeout 900 python -m pytest tests/test_ops.py -q -x -m gpu -k "groupnorm or fir or resblock" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do for v in 1 0; do
  STORM_GN_WIDE=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); h=r['roofline_hbm']
print('STORM_GN_WIDE=$v', round(r['value'],3), 'utt/s', round(r['ms_per_nfe_batch'],3), 'ms/nfe', r['roofline']['ms_by_op_kind']['gn_apply'], {k.split('::')[-1][:40]:(round(v['ms_per_nfe'],3), v.get('tb_per_s')) for k,v in h['by_kernel'].items() if 'gn_apply' in k})"
done; done 2>&1 | tee gpurun_out/r04_gnwide_ab.txt
