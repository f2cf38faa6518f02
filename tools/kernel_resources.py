"""Register / scratch / occupancy table of every kernel of the library, from hipcc's own remarks (no GPU needed):

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt

Compiles the kernel sources with the product's flags plus -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import build as B  # noqa: E402

rows = []
with tempfile.TemporaryDirectory() as d:
    for src in B.SOURCES:
        if not src.endswith(".hip"):
            continue
        cmd, rc, log = B._compile_one(B._hipcc(), src, os.path.join(d, src + ".o"), ["-Rpass-analysis=kernel-resource-usage"])
        if rc:
            raise SystemExit(log)
        for blk in re.split(r"remark: Function Name: ", log)[1:]:
            name = blk.split()[0]

            def g(key):
                m = re.search(re.escape(key) + r": (\d+)", blk)
                return int(m.group(1)) if m else -1

            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\((anonymous namespace)\)::", "", dem)
            dem = re.sub(r"^void ", "", dem)
            dem = re.sub(r"\(r2f::.*$", "", dem).replace("r2f::", "")
            rows.append((src, dem, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"),
                         g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
print("# hipcc --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py); LDS = static only (the FFT and")
print("# tile kernels take theirs dynamically); waves/SIMD = the register-limited occupancy the compiler reports")
print(f"{'file':<16}{'kernel':<58}{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'spilled':>8}{'scratch B':>10}{'waves/SIMD':>11}{'LDS':>7}")
for r in rows:
    print(f"{r[0]:<16}{r[1][:57]:<58}{r[2]:>5}{r[3]:>5}{r[4]:>5}{r[5]:>8}{r[6]:>10}{r[7]:>11}{r[8]:>7}")
spilled = [r for r in rows if r[5] > 0 or r[6] > 0]
print(f"# {len(rows)} kernels; with spilled registers or scratch: {len(spilled)}" + "".join(f"\n#   {r[1]}: {r[5]} VGPRs spilled, {r[6]} B/lane" for r in spilled))
