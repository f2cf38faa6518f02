#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <stage> [W H variant]   -- separate PMC passes, kernel-trace only
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass$i -- python3 tools/prof_stage.py "$@" > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:.4g}")
PY
