#!/bin/bash
# is the chip at its power cap under the bench?  sample rocm-smi while bench.py runs
export TMPDIR=/tmp; mkdir -p gpurun_out
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -8
python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > gpurun_out/bench_power.json 2>/dev/null &
BP=$!
sleep 45
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
python -c "import json; b=json.load(open('gpurun_out/bench_power.json')); print(b['value'])"
