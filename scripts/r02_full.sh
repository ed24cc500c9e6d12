#!/bin/bash
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_nfe_batch'], r['roofline']['frac'], r['roofline']['ms_by_op_kind'])"
