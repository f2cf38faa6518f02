"""Build libr2f_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m raw2film_amd.build            # rebuild if sources are newer than the .so
    python -m raw2film_amd.build --force
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libr2f_hip.so")
SOURCES = ["r2f_kernels.hip", "r2f_fft.hip", "r2f_api.hip"]
HEADERS = ["r2f_device.h", "r2f_launch.h", os.path.join("..", "..", "include", "r2f.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the r2f HIP library cannot be built on this machine")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into raw2film_amd/libr2f_hip.so; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [
        _hipcc(),
        "-O3",
        "-std=c++17",
        f"--offload-arch={ARCH}",
        "-fPIC",
        "-shared",
        "-Wall",
        "-Wno-unused-function",
        "-o",
        LIB_PATH + ".tmp",
    ] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
