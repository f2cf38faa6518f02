#!/bin/bash
# (SQ counters only: a pass with TA_* / TCP_* counters aborted inside rocprofv3 and hung until the outer timeout on this pool)
# usage: tools/prof_cfg2.sh [noise|smooth]  -- PMC passes of the LUT-only fused kernel (BASELINE config 2, 24 MP), kernel-trace only
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
FRAME=${1:-noise}
OUT=gpurun_out/pmc_cfg2_$FRAME
rm -rf $OUT; mkdir -p $OUT
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -- python3 bench.py --config cfg2_24mp --frame $FRAME --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "r2f::" not in row["Kernel_Name"]: continue
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {v / n[k][c]:.4g}  (per launch, {n[k][c]} launches)")
PY
