#!/bin/bash
# usage: tools/prof_traffic.sh <tag> <stage>   -- L2-miss (HBM-side) read/write traffic of one stage for xcd_remap = 0, 1, 2
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; STAGE=${2:-halation}
OUT=gpurun_out/traffic_$TAG
rm -rf $OUT; mkdir -p $OUT
for X in 0 1 2; do
  export XCD=$X
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/x${X}_fetch -- python3 tools/prof_stage.py $STAGE > $OUT/x${X}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/x${X}_write -- python3 tools/prof_stage.py $STAGE > $OUT/x${X}_write.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for x in (0, 1, 2):
    d = collections.defaultdict(list)
    for f in glob.glob("$OUT/x%d_*/**/*counter_collection.csv" % x, recursive=True):
        for row in csv.DictReader(open(f)):
            if "stencil_kernel" in row["Kernel_Name"]:
                d[row["Counter_Name"]].append(float(row["Counter_Value"]))
    fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    print("xcd_remap=%d  $STAGE: reads %.3f GB (FETCH_SIZE x2)  writes %.3f GB  total %.3f GB  (%d launches)" % (x, 2 * fe * 1024 / 1e9, wr * 1024 / 1e9, (2 * fe + wr) * 1024 / 1e9, len(d["FETCH_SIZE"])))
PY
