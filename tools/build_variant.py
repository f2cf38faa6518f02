"""Build development variants of libr2f_hip.so into tools/_var/ (git-ignored; the .so files travel to the GPU box with gpurun).

    python tools/build_variant.py name[:src1.hip,src2.hip]=-DA=1,-DB=2 [more variants ...]

Only the sources named after the colon are recompiled with the defines (default: r2f_fft.hip); the other objects come from one
shared base build (kept under /tmp for the session), so a variant costs one hipcc run instead of six.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import build as B  # noqa: E402

BASE = "/tmp/r2f_variant_base"
OUT = os.path.join(ROOT, "tools", "_var")


def obj_of(d, s):
    return os.path.join(d, os.path.splitext(s)[0] + ".o")


def compile_all(d, sources, defines):
    os.makedirs(d, exist_ok=True)
    hipcc = B._hipcc()
    with ThreadPoolExecutor(max(1, min(6, len(sources)))) as pool:
        res = list(pool.map(lambda s: B._compile_one(hipcc, s, obj_of(d, s), defines), sources))
    for cmd, rc, log in res:
        if rc:
            raise SystemExit(f"hipcc failed: {cmd}\n{log}")


def base_objects():
    deps = [os.path.join(B.CSRC, s) for s in B.SOURCES + B.HEADERS]
    stale = [s for s in B.SOURCES
             if not os.path.exists(obj_of(BASE, s)) or any(os.path.getmtime(d) > os.path.getmtime(obj_of(BASE, s)) for d in deps)]
    if stale:
        compile_all(BASE, stale, [])


def main():
    os.makedirs(OUT, exist_ok=True)
    base_objects()
    for spec in sys.argv[1:]:
        head, _, defs = spec.partition("=")
        name, _, srcs = head.partition(":")
        sources = srcs.split(",") if srcs else ["r2f_fft.hip"]
        defines = [d for d in defs.split(",") if d]
        d = f"/tmp/r2f_variant_{name}"
        compile_all(d, sources, defines)
        objs = [obj_of(d if s in sources else BASE, s) for s in B.SOURCES]
        out = os.path.join(OUT, f"lib_{name}.so")
        subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={B.EXPORTS_MAP}", "-o", out] + objs, check=True)
        print(out, flush=True)


if __name__ == "__main__":
    main()
