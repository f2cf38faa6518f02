#!/bin/bash
# usage: tools/profile_round.sh r01
# (1) rocprofv3 --kernel-trace --stats of the default bench.py command, (2) HBM traffic counters of the
# dominant kernel in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), with the
# FETCH_SIZE x2 correction for gfx950 (MI355X_MICROARCH.md, HBM; re-checked by tools/cal_pmc.sh).
# Summaries land in gpurun_out/profile_<tag>/summary/ ; copy them to profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT/summary
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alone > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alone > $OUT/pmc_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
out, tag = "$OUT", "$TAG"
rows = []
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/summary/{tag}_kernel_stats.csv", "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alone (9 renders of the 100 MP frame: 2 warm-up + 5 timed + 2 for the\n# per-pass breakdown)\n")
    fh.write("# torch's frame-generation kernels are folded into one line\n")
    fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    other = [0, 0.0]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if "r2f::" in r["Name"]:
            fh.write('"%s",%s,%s,%s,%.2f,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                     100 * float(r["TotalDurationNs"]) / total, r["MinNs"], r["MaxNs"]))
        else:
            other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"])
    fh.write('"(torch: synthetic frame generation, copies)",%d,%.0f,,%.2f,,\n' % (other[0], other[1], 100 * other[1] / total))
print(open(f"{out}/summary/{tag}_kernel_stats.csv").read())
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "r2f::" in row["Kernel_Name"]:
            tot[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, d in tot.items():
    fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    summary[k] = {"FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write, "launches": len(d["FETCH_SIZE"]),
                  "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
                  "note": "FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B (checked with tools/cal_pmc.sh: 1 GiB copy reads report 512 MiB for 4-B and 16-B per-lane loads; WRITE_SIZE exact)"}
json.dump(summary, open(f"{out}/summary/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
tail -1 $OUT/bench_under_rocprof.log | cut -c1-400
