"""Timed CPU baseline for bench.py: the NumPy oracle ("port" of the reference's CPU path) run on
a bounded sample of the benchmark workload on the box's own host cores.

TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is a reported baseline, never the
thing shipped: the product path has no CPU implementation.

Method: the sample is a set of independent 1.5 MP tiles rendered with exactly the benchmark's LUTs and
stencils (same px/mm, hence the same 87x87 / 35x35 / 9x9 taps as the full frame; the mirror padding of a
1024 x 1536 tile is 1.14x its area, against 1.3x for the 512 x 768 tiles of round 1, so the baseline is no
longer understated by the tiling); one worker process per host core (scipy.fft pinned to one thread per
worker), tiles dealt round-robin.
Stencils use FFT correlation with mirror padding -- what OpenCV's filter2D does for kernels
larger than 11x11 on the reference's CPU path -- so the baseline is not artificially slow.
"""

from __future__ import annotations

import multiprocessing as mp
import os
import time

import numpy as np

from . import stages as st

_STATE = {}


def _init(p, tile_hw, seed):
    st.FFT_WORKERS = 1
    _STATE["p"] = p
    _STATE["hw"] = tile_hw
    _STATE["seed"] = seed


def _tile(i):
    H, W = _STATE["hw"]
    rng = np.random.default_rng(_STATE["seed"] + i)
    img = (0.18 * 2.0 ** rng.normal(0.0, 1.5, (H, W, 1)) * rng.uniform(0.6, 1.4, (H, W, 3))).astype(np.float32)
    out = st.render(img, _STATE["p"])
    return float(out[0, 0, 0])


def usable_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota if the container has one
    (an oversubscribed pool of 256 workers on a 32-CPU quota measures the scheduler, not the code)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def time_cpu_baseline(p: st.RenderInputs, tile_hw=(1024, 1536), target_seconds: float = 15.0, cores: int | None = None,
                      seed: int = 1234) -> dict:
    """Render tiles for about `target_seconds` of wall time; returns the cpu_baseline object of bench.py."""
    if cores is None:
        cores = usable_cores()
    H, W = tile_hw
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_init, initargs=(p, tile_hw, seed)) as pool:
        t0 = time.perf_counter()
        pool.map(_tile, range(cores))  # calibration round: one tile per core
        t_round = time.perf_counter() - t0
        rounds = int(max(1, min(64, round(target_seconds / max(t_round, 1e-3)) - 1)))
        t0 = time.perf_counter()
        pool.map(_tile, range(cores, cores * (rounds + 1)), chunksize=1)
        dt = time.perf_counter() - t0
    n_tiles = cores * rounds
    mp_done = n_tiles * H * W / 1e6
    # the same tile on ONE process with the machine otherwise idle (NumPy's element-wise work is single-threaded; SciPy's
    # FFT pinned to one thread): SURVEY 8d's "(i) single process" figure next to the all-cores one
    _init(p, tile_hw, seed)
    t0 = time.perf_counter()
    _tile(0)
    t_single = time.perf_counter() - t0
    return {
        "single_process": {"value": H * W / 1e6 / t_single, "unit": "MP/s", "cores": 1,
                           "sample": f"one {W}x{H} tile in {t_single:.2f} s"},
        "value": mp_done / dt,
        "unit": "MP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_tiles} tiles of {W}x{H} px ({mp_done:.1f} MP) of the same workload (same LUTs and stencil sizes), "
                  f"{dt:.1f} s wall on {cores} worker processes",
    }
