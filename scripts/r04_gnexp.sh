#!/bin/bash
# round 4: non-temporal output stores in the HBM-bound family (STORM_GN_NT bits: 1 gn_apply_up, 2 gn_apply_down, 4 conv_thin), rows per strip
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/b.log 2>&1
run() { env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); h=r['roofline_hbm']
print('$*', round(r['value'],3), 'gn_apply', r['roofline']['ms_by_op_kind']['gn_apply'], {k.split('::')[-1][:30]:(v['ms_per_nfe'], v['tb_per_s']) for k,v in h['by_kernel'].items() if 'gn_apply' in k or 'narrow conv: 3x3 8' in k or '1x1 8' in k})"; }
run STORM_GN_NT=0
run STORM_GN_NT=1
run STORM_GN_NT=3
run STORM_GN_NT=5
run STORM_GN_NT=7
run STORM_GN_NT=1 STORM_GN_ROWS=8
run STORM_GN_NT=0
run STORM_GN_NT=1
