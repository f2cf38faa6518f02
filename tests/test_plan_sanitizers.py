"""The host-side planners of libr2f_hip.so (raw2film_amd/csrc/r2f_plan.cpp: FFT window choice and batch sizing, the direct
stencils' entry-list builder, tile orders, LANCZOS4 / Gaussian / curve tables, workspace sizes) under AddressSanitizer and
UndefinedBehaviorSanitizer.  GPU-side sanitizers are not available on this pool; this is the part of the
library that can run under one: the same translation unit the library links is compiled by g++ with
`-fsanitize=address,undefined -fno-sanitize-recover=all` together with tests/plan_fuzz.cpp, which drives it over frame shapes
1 x 1 ... 16384^2, tap boxes 1 ... 400 (odd, even, non-square, sparse, disc-like, mirrored), shard row ranges and every value
of the FFT options, and checks each plan against its own contract (windows cover the rows and columns, batches cover the pairs,
an entry list reproduces the taps it was built from -- every tap once per output row -- inside the LDS rows of its phase, a
tile order is a permutation ...).  Round 4: 229 000 cases over 13 seeds, nothing found (the round-2 division by zero in the
window choice is the kind of bug this is for)."""

import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("plan_fuzz") / "plan_fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Werror",
           os.path.join(ROOT, "tests", "plan_fuzz.cpp"), os.path.join(ROOT, "raw2film_amd", "csrc", "r2f_plan.cpp"), "-o", out]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 20261003])
def test_planners_are_clean_under_asan_and_ubsan(fuzz_binary, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([fuzz_binary, str(seed), os.environ.get("R2F_PLAN_FUZZ_CASES", "2500")], capture_output=True, text=True, env=env,
                         timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-4000:]
    assert "cases ok" in res.stdout


# ---- the same planners through the product library's plan-only exports (no GPU, no context)
def _lib():
    from raw2film_amd import _lib

    return _lib, _lib.load()


def test_plan_fft_matches_the_shapes_the_bench_configurations_run():
    _l, lib = _lib()
    p = _l.FftPlan()
    # cfg 4: 85 x 85 non-zero halation taps on 12288 x 8192, complex128 scratch -> 256 x 512 windows, 172 x 428 valid
    assert lib.r2f_plan_fft(85, 85, 12288, 8192, 2, 16, 0, 0, 512, 192, 2, ctypes.byref(p)) == 0
    assert (p.ny, p.nx, p.vy, p.vx) == (256, 512, 172, 428)
    assert p.gx == 29 and p.ntiles == 29 * 48 and p.pairs == 2 * 696 and p.streams == 2
    assert p.batch * p.launches >= p.pairs and p.scratch_bytes == p.batch * 2 * 256 * 512 * 16
    # ... and its 35 x 35 MTF on complex64 scratch
    assert lib.r2f_plan_fft(35, 35, 12288, 8192, 3, 8, 0, 0, 512, 192, 2, ctypes.byref(p)) == 0
    assert (p.ny, p.nx, p.vy, p.vx) == (256, 512, 222, 476)
    # a 1/8 row shard's halation call (1 024 own rows + 2 x 17): the window rows are chosen for the rows of the call
    assert lib.r2f_plan_fft(85, 85, 12288, 1058, 2, 16, 0, 0, 512, 192, 2, ctypes.byref(p)) == 0
    whole = _l.FftPlan()
    assert lib.r2f_plan_fft(85, 85, 12288, 1058, 2, 16, 0, 256, 512, 192, 2, ctypes.byref(whole)) == 0
    cost = lambda q: q.ntiles * q.ny * q.nx  # noqa: E731 -- window elements transformed
    assert cost(p) <= cost(whole)
    # a forced axis that cannot hold the box is refused, not divided by
    assert lib.r2f_plan_fft(300, 20, 1000, 1000, 1, 16, 0, 256, 512, 192, 2, ctypes.byref(p)) == 0 and p.ny == 512  # (tall boxes ignore it)
    assert lib.r2f_plan_fft(401, 20, 1000, 1000, 1, 16, 0, 0, 512, 192, 2, ctypes.byref(p)) == _l.EINVAL


def test_plan_stencil_checks_its_entry_list_and_reports_the_geometry():
    _l, lib = _lib()
    from raw2film_amd import stencils

    k = np.ascontiguousarray(stencils.halation_stencil(12288 / 36.0, 1.0, halation_green_factor=0.3))  # 87 x 87 x 3, blue = identity
    out = (ctypes.c_int * 8)()
    rc = lib.r2f_plan_stencil(k.ctypes.data, k.shape[0], k.shape[1], k.shape[2], 0, 4, 128, 64, 80 * 1024, 1, out)
    assert rc == 0
    entries, rowsteps, phases, sym, kh, kw, rs, rows = list(out)
    assert sym == 1 and kh == 85 and kw >= 85 and rs == 128 + (kw + 3) // 4 * 4 and phases >= 2 and (rows * rs + 16) * 4 <= 80 * 1024
    assert lib.r2f_plan_stencil(k.ctypes.data, k.shape[0], k.shape[1], k.shape[2], 2, 4, 128, 64, 80 * 1024, 1, out) == 0
    assert list(out)[:2] == [4, 4] and out[4] == 1 and out[5] == 1  # the identity plane: one tap, used by each of a lane's 4 rows
    assert lib.r2f_plan_stencil(k.ctypes.data, k.shape[0], k.shape[1], k.shape[2], 0, 4, 128, 64, 8 * 1024, 1, out) == _l.ETOOLARGE


def test_plan_tile_order_is_a_permutation():
    _l, lib = _lib()
    for gx, gy, band in ((96, 128, 0), (1, 1, 0), (7, 3, 2), (200, 1, 0), (3, 500, 5)):
        order = np.full(gx * gy, -1, dtype=np.int32)
        assert lib.r2f_plan_tile_order(gx, gy, band, order.ctypes.data) == 0
        assert sorted(order.tolist()) == list(range(gx * gy))
