"""Timed CPU baseline for bench.py: the NumPy oracle ("port" of the reference's CPU path) run on
a bounded sample of the benchmark workload on the box's own host cores.

TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is a reported baseline, never the
thing shipped: the product path has no CPU implementation.

Method: the sample is a set of independent tiles rendered with exactly the benchmark's LUTs and
stencils (same px/mm, hence the same 87x87 / 35x35 / 9x9 taps as the full frame); one worker
process per host core (scipy.fft pinned to one thread per worker), tiles dealt round-robin.
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


def time_cpu_baseline(p: st.RenderInputs, tile_hw=(512, 768), target_seconds: float = 15.0, cores: int | None = None,
                      seed: int = 1234) -> dict:
    """Render tiles for about `target_seconds` of wall time; returns the cpu_baseline object of bench.py."""
    if cores is None:
        cores = len(os.sched_getaffinity(0))
    H, W = tile_hw
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_init, initargs=(p, tile_hw, seed)) as pool:
        t0 = time.perf_counter()
        pool.map(_tile, range(cores))  # calibration round: one tile per core
        t_round = time.perf_counter() - t0
        rounds = int(max(1, min(8, round(target_seconds / max(t_round, 1e-3)) - 1)))
        t0 = time.perf_counter()
        pool.map(_tile, range(cores, cores * (rounds + 1)), chunksize=1)
        dt = time.perf_counter() - t0
    n_tiles = cores * rounds
    mp_done = n_tiles * H * W / 1e6
    return {
        "value": mp_done / dt,
        "unit": "MP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_tiles} tiles of {W}x{H} px ({mp_done:.1f} MP) of the same workload (same LUTs and stencil sizes), "
                  f"{dt:.1f} s wall on {cores} worker processes",
    }
