#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py r2f_front.hip [-DNAME=VALUE ...] [--filter substring] [--spills]

Objects go to /tmp; nothing is written under the repository.
"""

from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raw2film_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def resources(src: str, defines=()):
    path = src if os.path.isabs(src) else os.path.join(CSRC, src)
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", "-o", f"/tmp/_res_{os.path.basename(src)}.o", path,
           "-Rpass-analysis=kernel-resource-usage", *defines]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit(res.stderr)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(r2f::\w+(?: const)?(?:, int)?\)$", "", n)
        r["short"] = n.replace("r2f::", "")
    return rows


def main():
    args = sys.argv[1:]
    flt, spills_only = None, False
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    if "--spills" in args:
        spills_only = True
        args.remove("--spills")
    src = args[0]
    rows = resources(src, [a for a in args[1:] if a.startswith("-D")])
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
    for r in rows:
        if flt and flt not in r["short"]:
            continue
        scratch = int(r.get("ScratchSize [bytes/lane]", "0"))
        if spills_only and scratch == 0:
            continue
        print(f"{r['short'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} {scratch:8d} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4s} {r.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()
