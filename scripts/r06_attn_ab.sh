#!/bin/bash
# round 6: the bottleneck attention at the bench batch (128 query-block workgroups = half the chip) with / without a two-way key-range split
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for rep in 1 2 3; do
for v in 0 2 4; do
  STORM_ATTN_SPLIT=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-traffic --profile-nfe 4 > gpurun_out/ab_attn_${v}_$rep.json 2> gpurun_out/ab_attn_${v}_$rep.err
  python - <<PY
import json
r = json.load(open("gpurun_out/ab_attn_${v}_$rep.json"))
k = r["roofline"]["ms_by_op_kind"]
print("STORM_ATTN_SPLIT=$v rep $rep", "utt/s %.3f" % r["value"], "ms/nfe %.3f" % r["ms_per_nfe_batch"], "attention %.3f ms" % k["attention"], "conv %.2f" % k["conv"])
PY
done
done
