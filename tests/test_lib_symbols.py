"""The C-ABI library builds, loads without a GPU, and exports every symbol include/r2f.h declares."""

import ctypes
import os
import re

from raw2film_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "r2f.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(r2f_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m raw2film_amd.build` (the driver's build() does)"
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert getattr(raw, name) is not None
    assert b"gfx950" in lib.r2f_version()


def test_dynamic_symbol_table_is_exactly_the_c_abi():
    """Built with -fvisibility=hidden and csrc/r2f_exports.map: `nm -D --defined-only` lists the entry points of include/r2f.h
    and nothing else -- no r2f:: launchers, planners or kernel stubs for a second copy of the library to interpose."""
    import shutil
    import subprocess

    nm = shutil.which("nm")
    if nm is None:
        import pytest

        pytest.skip("no nm")
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_functions()


def test_struct_layouts_match_the_header():
    # r2f_params: 2 x u32, 2 x f32, 2 x i32, 2 x f32 ; r2f_planes: ptr, i64, 2 x i32
    assert ctypes.sizeof(_lib.Params) == 32
    assert ctypes.sizeof(_lib.Planes) == 24
    assert _lib.Planes.plane_stride.offset == 8 and _lib.Planes.gy0.offset == 16


def test_workspace_bytes_needs_no_gpu():
    lib = _lib.load()
    p = _lib.Params(_lib.F_HALATION | _lib.F_MTF | _lib.F_GRAIN, 1, 1e-6, 0.25, 0, 0, 0.0, 0.0)
    assert lib.r2f_workspace_bytes(ctypes.byref(p), 100, 200) == 2 * 3 * 100 * 200 * 4
    p.flags = _lib.F_GRAIN
    assert lib.r2f_workspace_bytes(ctypes.byref(p), 100, 200) == 3 * 100 * 200 * 4
    p.flags = 0
    assert lib.r2f_workspace_bytes(ctypes.byref(p), 100, 200) == 0
    p.flags, p.burn_cell = _lib.F_GRAIN | _lib.F_BURN, 10  # grain -> planes, burn map 10 x 20 (x4 floats of scratch)
    assert lib.r2f_workspace_bytes(ctypes.byref(p), 100, 200) == 2 * 3 * 100 * 200 * 4 + 4 * 10 * 20 * 4


def test_header_is_plain_c(tmp_path):
    """include/r2f.h is the boundary a cgo / JNI / ctypes binding reads: it has to compile as C99 without any HIP or C++ header."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest

        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "r2f.h"\nint main(void) { r2f_ctx* c = 0; r2f_params p; (void)c; (void)p; return 0; }\n')
    res = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
