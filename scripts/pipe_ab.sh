#!/bin/bash
python -m pytest tests/test_ops.py tests/test_net.py -m gpu -q -x 2>&1 | tail -1
timeout 600 python tools/conv_check.py 2>&1 | grep -v "^ok" | tail -4
for i in 1 2 3; do
  echo "variant 0 reg: $(STORM_CONV_DMA=0 STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
  echo "variant 0 dma: $(STORM_CONV_DMA=1 STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 5 2>&1 | grep -E '^c' | tr '\n' ' ')"
done
echo "ksweep v0 dma: $(PROBE_KSWEEP=1 STORM_CONV_VARIANT=0 python tools/conv_probe.py --reps 5 2>&1 | grep -E "^k" | sed "s/ ms.*TF//" | tr "\n" " ")"
