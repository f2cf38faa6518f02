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
SOURCES = ["r2f_kernels.hip", "r2f_fft.hip", "r2f_front.hip", "r2f_post.hip", "r2f_api.hip", "r2f_plan.cpp"]
HEADERS = ["r2f_device.h", "r2f_launch.h", "r2f_fft_math.h", "r2f_plan.h", os.path.join("..", "..", "include", "r2f.h")]
ARCH = "gfx950"
EXPORTS_MAP = os.path.join(CSRC, "r2f_exports.map")  # only the r2f_* entry points of include/r2f.h stay dynamic


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the r2f HIP library cannot be built on this machine")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [EXPORTS_MAP]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(hipcc: str, src: str, obj: str, defines: list[str]) -> tuple[str, int, str]:
    # -fvisibility=hidden: the library's dynamic symbols are the R2F_API entry points of include/r2f.h and nothing else (no r2f::
    # launchers, planners or kernel stubs for a second copy of the library in the same process to interpose)
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function",
           "-c", "-o", obj, os.path.join(CSRC, src)] + defines
    if src.endswith(".hip"):  # (r2f_plan.cpp is host-only C++: no device pass)
        cmd.insert(3, f"--offload-arch={ARCH}")
    res = subprocess.run(cmd, capture_output=True, text=True)
    return " ".join(cmd), res.returncode, res.stdout + res.stderr


def build(force: bool = False, verbose: bool = False, defines: list[str] | None = None, out: str | None = None) -> str:
    """Compile every HIP source for gfx950 (one hipcc per source, in parallel) and link them into
    raw2film_amd/libr2f_hip.so; returns its path.  `defines` (-D...) and `out` build development variants elsewhere."""
    from concurrent.futures import ThreadPoolExecutor

    target = out or LIB_PATH
    if not force and not defines and not out and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    # objects live in a scratch directory outside the tree (development variants used to leave ~80 MB of them under csrc/,
    # which then rode along with every snapshot pushed to the GPU box)
    import tempfile

    objdir = tempfile.mkdtemp(prefix="r2f_obj_")
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    with ThreadPoolExecutor(len(SOURCES)) as pool:
        results = list(pool.map(lambda so: _compile_one(hipcc, so[0], so[1], list(defines or [])), zip(SOURCES, objs)))
    for cmd, rc, log in results:
        if verbose:
            print(cmd, flush=True)
        if rc != 0:
            raise RuntimeError(f"hipcc failed ({rc}): {cmd}\n{log}")
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS_MAP}", "-o", target + ".tmp"] + objs
    if verbose:
        print(" ".join(link), flush=True)
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    os.replace(target + ".tmp", target)
    shutil.rmtree(objdir, ignore_errors=True)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
