#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_ops.py tests/test_net.py tests/test_model.py tests/test_sampler.py -m gpu -q -x -s -k "attention or tiny or full_width or bench_shape or selection or ragged or surfaces or silent or si_sdr or langevin or ode or time_argument" 2>&1 | grep -v "^$" | tail -15
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_nfe_batch'], r['roofline']['frac'], r['roofline']['ms_by_op_kind'])"
