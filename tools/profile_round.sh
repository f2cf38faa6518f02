#!/bin/bash
# usage: tools/profile_round.sh <tag> [config] [label] [extra bench.py arguments ...]
#   e.g.  tools/profile_round.sh r06 cfg4_100mp
#         tools/profile_round.sh r06 cfg4_100mp complex128 --opt stencil_fft_scratch96_auto=0     (-> r06_cfg4_100mp_complex128_*)
# (1) rocprofv3 --kernel-trace --stats of the bench.py command for that configuration, (2) L2 <-> fabric traffic of every r2f
# kernel in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; kernel-trace only beside them), with the
# FETCH_SIZE x 2 correction for gfx950 (MI355X_MICROARCH.md, HBM; re-checked by tools/cal_pmc.sh).
# Every run passes --no-breakdown (round 6): the profiled process holds nothing but the product's own renders -- warm-up frames and
# timed steps of the same captured graph -- so calls = renders x launches per render for every kernel, and bytes_per_step is the
# replayed frame's own (round 5's captures averaged graph renders with eager stage-by-stage ones that used another scratch element).
# Summaries land in gpurun_out/profile_<tag>_<config>[_<label>]/summary/ ; copy them to profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r06}
CFG=${2:-cfg4_100mp}
LABEL=${3:-}
shift; shift; shift
EXTRA="$*"
NAME=$CFG${LABEL:+_$LABEL}
OUT=gpurun_out/profile_${TAG}_$NAME
STEPS=5; WARM=2; PSTEPS=2; PWARM=1
rm -rf $OUT; mkdir -p $OUT/summary
ARGS="--config $CFG --no-cpu-baseline --no-alone --no-pcie --no-breakdown $EXTRA"
[ "$CFG" = cfg5_batch ] && ARGS="$ARGS --frames 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps $STEPS --warmup $WARM $ARGS > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps $PSTEPS --warmup $PWARM $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps $PSTEPS --warmup $PWARM $ARGS > $OUT/pmc_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections, datetime, sys
sys.path.insert(0, ".")
from bench import source_hash
out, tag, cfg, name = "$OUT", "$TAG", "$CFG", "$NAME"
steps, warm, psteps, pwarm = $STEPS, $WARM, $PSTEPS, $PWARM
frames = 8 if cfg == "cfg5_batch" else 1
rows = []
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
# renders in the traced run: the kernels launched once per render (front / tail / the fused LUT pass) tell
once = [int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("tail_kernel", "front_fast_kernel", "front_kernel", "lut3d_kernel"))]
renders = min(once) if once else (steps + warm + 2) * frames
with open(f"{out}/summary/{tag}_{name}_kernel_stats.csv", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup {warm} $ARGS\n")
    fh.write(f"# ({renders} renders of the frame, nothing else: warm-up frames + {steps} timed steps, all of them the product's own r2f_render (the first kernel by kernel, the rest one captured HIP graph); torch's frame-generation kernels are folded into one line\n")
    fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    other = [0, 0.0]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if "r2f::" in r["Name"] and "stream_copy_kernel" not in r["Name"]:
            fh.write('"%s",%s,%s,%s,%.2f,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                     100 * float(r["TotalDurationNs"]) / total, r["MinNs"], r["MaxNs"]))
        else:
            other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"])
    fh.write('"(torch: synthetic frame generation, copies)",%d,%.0f,,%.2f,,\n' % (other[0], other[1], 100 * other[1] / total))
print(open(f"{out}/summary/{tag}_{name}_kernel_stats.csv").read())
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "r2f::" in row["Kernel_Name"] and "stream_copy_kernel" not in row["Kernel_Name"]:  # (bench.py's copy-ceiling measurement is not part of a step)
            tot[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
once = [len(d["FETCH_SIZE"]) for k, d in tot.items() if any(n in k for n in ("tail_kernel", "front_fast_kernel", "front_kernel", "lut3d_kernel"))]
prenders = min(once) if once else (psteps + pwarm + 2) * frames   # renders in the counter runs (once-per-render kernels)
per_step = 0.0
for k, d in tot.items():
    fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    n = len(d["FETCH_SIZE"])
    b = (2 * fetch + write) * 1024
    summary[k] = {"FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write, "launches": n, "launches_per_render": n / prenders,
                  "hbm_bytes_per_launch": b, "bytes_per_render": b * n / prenders}
    per_step += b * n / prenders * frames
summary["_renders_in_counter_run"] = prenders
H, W = {"cfg4_100mp": (8192, 12288), "cfg3_45mp": (5504, 8256), "cfg2_24mp": (4000, 6000), "cfg5_batch": (4000, 6000)}[cfg]
summary["_meta"] = {
    "config": cfg, "frame": "noise", "label": "$LABEL", "extra_args": "$EXTRA", "source_hash": source_hash(), "date": datetime.datetime.now().isoformat(timespec="seconds"),
    "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps {psteps} --warmup {pwarm} $ARGS (two passes)",
    "bytes_per_step": per_step, "algorithmic_bytes_per_step": 24.0 * H * W * frames, "ratio": per_step / (24.0 * H * W * frames),
    "renders_in_counter_run": prenders, "eager_stage_renders_in_counter_run": 0,
    "note": "every render of the counter runs is the product's own r2f_render (--no-breakdown): bytes_per_step = sum of bytes_per_render. "
            "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch, averaged over the launches of each kernel: FETCH_SIZE doubled "
            "because gfx950 tallies 128-B read requests at 64 B (tools/cal_pmc.sh: a 1 GiB copy reads 512 MiB); WRITE_SIZE exact. "
            "L2 <-> fabric requests, Infinity Cache hits included."}
json.dump(summary, open(f"{out}/summary/{tag}_{name}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(summary["_meta"], indent=1))
PY
grep "^{" $OUT/bench_under_rocprof.log | tail -1 | cut -c1-600
