#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/cal_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- ./tools/ubench/copy_cal > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- ./tools/ubench/copy_cal > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        print(row["Kernel_Name"][:40], row["Counter_Name"], row["Counter_Value"])
PY
