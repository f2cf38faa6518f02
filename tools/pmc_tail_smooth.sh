cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/pmc_tail_smooth
FRAME=smooth rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tail_smooth/p1 -- python3 tools/prof_stage.py tail > gpurun_out/pmc_tail_smooth/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_tail_smooth/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tail_kernel" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:50]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:.4g}")
PY
