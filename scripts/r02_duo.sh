#!/bin/bash
# conv_duo.hip: GPU parity, A/B probe against conv_igemm / conv_pipe128 on the <= 128-cout layers, whole-bench A/B (STORM_CONV_DUO)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -m gpu -q --tb=short -k "pipelined_128cout" 2>&1 | tail -2
echo "== fused GN"; timeout 300 python tools/probe128.py --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-250
echo "== plain"; timeout 300 python tools/probe128.py --reps 5 --nogn 2>&1 | grep -v amdgpu.ids | cut -c1-250
for v in 0 1 0 1; do STORM_CONV_DUO=$v python bench.py --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('DUO=$v', round(b['value'],3), b['roofline']['ms_by_op_kind']['conv'], {n.split('<')[0][7:]: (v['ms_per_nfe'], v['tflops']) for n, v in b['roofline']['conv3x3_by_kernel'].items()})"; done
