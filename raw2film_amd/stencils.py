"""Host-side construction of the per-render stencils (tiny NumPy work, runs once per
parameter change; the reference's GPU path builds them on the host too,
gpu_processor.py:792-854).  Results are bit-identical to the reference's own functions
(tests/test_host_stencils.py checks them against tests/golden/*.npz).

  halation_stencil  <->  effects.compute_halation_kernel   effects.py:239-263 (+ :200-217)
  mtf_stencil       <->  effects.mtf_kernel                effects.py:165-185 (+ :114-162)
"""

from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
from scipy.ndimage import gaussian_filter

F32 = np.float32


def halation_psf(size_px: float) -> np.ndarray:
    """Radial 1/d^2 point spread clipped linearly to zero at radius size/2 (effects.py:200-217).
    Side length 2*floor(ceil(size)/2)+1, centre tap weight 1 before normalisation; float64."""
    side = 2 * math.floor(math.ceil(size_px) / 2) + 1
    c = math.ceil(side / 2) - 1
    ii, jj = np.indices((side, side))
    d2 = ((ii - c) ** 2 + (jj - c) ** 2).astype(np.float64)
    radius = size_px / 2
    psf = np.ones_like(d2)
    nz = d2 > 0
    psf[nz] = (1 / d2[nz]) * np.maximum((radius - np.sqrt(d2[nz])) / radius, 0)
    return psf / np.sum(psf)


def halation_stencil(
    scale: float,
    halation_size: float = 1.0,
    halation_red_factor: float = 1.0,
    halation_green_factor: float = 0.4,
    halation_blue_factor: float = 0.0,
    halation_intensity: float = 1.0,
    bw: bool = False,
) -> np.ndarray:
    """(k, k, 3) float32 with K_c = (psf*f_c + delta)/(1 + f_c): filtering with it gives
    (x + f_c * (psf (*) x)) / (1 + f_c).  `scale` in px/mm; psf diameter = scale/4*size px."""
    if bw:  # effects.py:248-250
        halation_red_factor = halation_blue_factor = halation_green_factor
    factors = halation_intensity * np.asarray(
        [halation_red_factor, halation_green_factor, halation_blue_factor], dtype=F32
    )
    psf = halation_psf(scale / 4 * halation_size).astype(F32)
    stencil = psf[..., None] * np.ones(3, dtype=F32)
    stencil *= factors
    mid = stencil.shape[0] // 2
    stencil[mid, mid] += 1.0
    stencil /= factors + 1.0
    return stencil


def _mtf_layer(logf: np.ndarray, vals: np.ndarray, scale: float) -> np.ndarray:
    """|ifft2| of the radial MTF sampled on a 0.1 mm window at `scale` px/mm (effects.py:123-162)."""
    taps = round(0.1 / (1 / scale))
    if taps % 2 == 0:
        taps += 1
    f1 = np.fft.fftfreq(taps, d=1 / scale)
    fx, fy = np.meshgrid(f1, f1)
    response = np.interp(np.log1p(np.sqrt(fx**2 + fy**2)), logf, vals, left=1, right=0)
    psf = np.fft.fftshift(np.abs(np.fft.ifft2(response)))
    return psf / np.sum(psf)


def mtf_stencil_from_table(mtf, scale: float, sharpening_strength: float = 0.0, sharpening_sigma: float = 1.0) -> np.ndarray:
    """(k, k, 3) float32 from `stock.mtf` = [(log1p f grid, response)] * 3."""
    stencil = np.stack(
        [_mtf_layer(np.asarray(lf), np.asarray(v), scale) for lf, v in mtf], axis=-1, dtype=F32
    )
    if sharpening_strength:
        # the reference blurs over all three axes, channel axis included (effects.py:181)
        blurred = gaussian_filter(stencil, sigma=sharpening_sigma * scale / 50)
        stencil += sharpening_strength * (stencil - blurred)
    return stencil


@lru_cache(maxsize=50)  # same cache shape as effects.py:165
def mtf_stencil(stock, scale: float, sharpening_strength: float = 0.0, sharpening_sigma: float = 1.0) -> np.ndarray:
    return mtf_stencil_from_table(stock.mtf, scale, sharpening_strength, sharpening_sigma)


def vertical_reach(stencil: np.ndarray) -> tuple[int, int]:
    """(rows above, rows below) the anchor that hold a non-zero tap in any channel: the halo a
    row shard needs for this stencil (anchor = kh // 2, as cv.filter2D / convolution.wgsl:31)."""
    k = np.asarray(stencil)
    if k.ndim == 2:
        k = k[..., None]
    rows = np.nonzero(np.any(k != 0, axis=(1, 2)))[0]
    if rows.size == 0:
        return 0, 0
    anchor = k.shape[0] // 2
    return max(anchor - int(rows[0]), 0), max(int(rows[-1]) - anchor, 0)


def vertical_reach_per_channel(stencil: np.ndarray) -> list[tuple[int, int]]:
    """`vertical_reach` of each of the three channels on its own: a channel whose stencil is a single tap at the anchor (the blue
    layer of a colour stock's halation, effects.py:248-263) reaches no neighbouring row, so a row shard need not exchange halo
    rows of that plane for this stencil."""
    k = np.asarray(stencil)
    if k.ndim == 2:
        k = k[..., None]
    return [vertical_reach(k[..., c if k.shape[2] > 1 else 0]) for c in range(3)]
