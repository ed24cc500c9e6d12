#!/bin/bash
# SQ counters of the conv probe per kernel variant (each counter group in its own rocprofv3 run, --kernel-trace only)
TAG=${1:-pmc_pipe}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
run() { v=$1; name=$2; shift; shift; STORM_CONV_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$TAG/v${v}_$name -o p -- python tools/conv_probe.py --reps 1 > gpurun_out/$TAG/v${v}_$name.log 2>&1; }
for v in 0 3; do
  run $v sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run $v sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM
  run $v sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_WAVES
done
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, collections
for v in (0, 3):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"gpurun_out/%s/v{v}_*/**/*counter_collection.csv" % "TAGX".replace("TAGX", __import__("os").environ.get("TAG", "pmc_pipe")), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv" not in k or "pack" in k: continue
            key = k[:60]
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])].add(r["Dispatch_Id"])
    for k, c in acc.items():
        print("variant", v, k)
        for name, val in sorted(c.items()):
            print(f"   {name:28s} {val / max(1, len(n[(k, name)])):16.0f} per launch ({len(n[(k, name)])} launches)")
PY
