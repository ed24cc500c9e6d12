#!/bin/bash
python -m pytest tests/test_ops.py tests/test_net.py -m gpu -q -x 2>&1 | tail -1
timeout 600 python tools/conv_check.py 2>&1 | grep -v "^ok" | tail -3
for i in 1 2; do for v in 0 3; do
  echo "variant $v: $(STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ') | $(PROBE_SHORTCUT=1 STORM_CONV_VARIANT=$v python tools/conv_probe.py --reps 5 2>&1 | grep -E '^sc' | tr '\n' ' ')"
done; done
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_ep.json 2>gpurun_out/bench_ep.err; python -c "
import json; r=json.load(open('gpurun_out/bench_ep.json')); print(r['value'], r['ms_per_nfe_batch'], r['roofline']['achieved'], r['roofline']['ms_by_op_kind'])"
