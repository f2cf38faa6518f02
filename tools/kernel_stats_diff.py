"""Per-kernel average durations of two rocprofv3 --kernel-trace --stats runs side by side (development aid):
    python tools/kernel_stats_diff.py dirA dirB
"""
import csv
import glob
import sys


def load(d):
    rows = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "r2f::" in r["Name"]:
                rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"{'kernel':90s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} | {'calls':>6s} {'avg us':>9s} {'total ms':>9s}")
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[2] + b.get(k, (0, 0, 0))[2])):
    x, y = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
    print(f"{k[:90]:90s} {x[0]:6d} {x[1]:9.2f} {x[2]:9.3f} | {y[0]:6d} {y[1]:9.2f} {y[2]:9.3f}")
print(f"{'sum of r2f kernels':90s} {'':6s} {'':9s} {sum(v[2] for v in a.values()):9.3f} | {'':6s} {'':9s} {sum(v[2] for v in b.values()):9.3f}")
