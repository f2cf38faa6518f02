"""Oracle restatement of the reference's host-side stencil-kernel builders.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned bit-for-bit against
`tests/golden/halation_kernels.npz` and `tests/golden/mtf_kernels.npz`, which
`tools/make_golden.py` produced by running the reference's own functions.

The restatements are written array-at-a-time (the reference loops per tap); the
IEEE operation order per element is kept, which is what makes them bit-exact.
"""

from __future__ import annotations

import math

import numpy as np
from scipy import ndimage

DTYPE = np.float32  # spectral_film_lut.config.DEFAULT_DTYPE (effects.py:15); fp32 per every upload in gpu_processor.py


def exponential_blur_kernel(size: float) -> np.ndarray:
    """Halation point-spread function -- restates effects.py:200-217.

    n = 2*floor(ceil(size)/2) + 1 taps per side, R = size/2.  With d2 the squared
    tap distance from the centre tap: weight 1 at d2 == 0, otherwise
    (1/d2) * max((R - sqrt(d2)) / R, 0); finally divided by the sum.  float64.
    """
    radius = size / 2
    n = 2 * math.floor(math.ceil(size) / 2) + 1
    centre = math.ceil(n / 2)  # 1-based centre index, effects.py:205
    off = np.arange(1, n + 1) - centre
    d2 = (off[:, None] ** 2 + off[None, :] ** 2).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        falloff = np.maximum((radius - np.sqrt(d2)) / radius, 0)
        k = (1 / d2) * falloff
    k[d2 == 0] = 1
    k /= np.sum(k)
    return k


def compute_halation_kernel(
    scale: float,
    halation_size: float = 1.0,
    halation_red_factor: float = 1.0,
    halation_green_factor: float = 0.4,
    halation_blue_factor: float = 0.0,
    halation_intensity: float = 1.0,
    bw: bool = False,
) -> np.ndarray:
    """Per-channel halation stencil -- restates effects.py:239-263.

    f = intensity * [red, green, blue] (all = green for a b/w stock, :248-250);
    K_c = (psf * f_c + delta_centre) / (f_c + 1), float32 (n, n, 3), psf at scale/4*size px.
    """
    if bw:
        halation_red_factor = halation_blue_factor = halation_green_factor
    psf = exponential_blur_kernel(scale / 4 * halation_size).astype(DTYPE)
    f = halation_intensity * np.array([halation_red_factor, halation_green_factor, halation_blue_factor], dtype=DTYPE)
    k = np.repeat(psf[:, :, None], 3, axis=2)
    k *= f
    c = k.shape[0] // 2
    k[c, c, :] += 1.0
    k /= f + 1.0
    return k


def mtf_curve(logf, vals):
    """MTF response as a function of cycles/mm -- restates effects.py:114-120:
    linear interpolation on a log1p(frequency) axis, 1 below the table, 0 above."""
    logf = np.asarray(logf)
    vals = np.asarray(vals)
    return lambda freq: np.interp(np.log1p(freq), logf, vals, left=1, right=0)


def compute_kernel_from_function(func, kernel_size_mm: float, pixel_size_mm: float) -> np.ndarray:
    """Spatial kernel of a radial transfer function -- restates effects.py:123-143.

    n = round(size_mm / pixel_mm), made odd by +1; H = func(|f|) on the fftfreq grid;
    K = fftshift(|ifft2(H)|) / sum.  float64 (n, n).
    """
    n = round(kernel_size_mm / pixel_size_mm)
    n += 1 - (n % 2)
    freq = np.fft.fftfreq(n, d=pixel_size_mm)
    fxx, fyy = np.meshgrid(freq, freq)
    radial = np.sqrt(fxx**2 + fyy**2)
    k = np.fft.fftshift(np.abs(np.fft.ifft2(func(radial))))
    k /= np.sum(k)
    return k


def mtf_kernel_layer(logf, vals, scale: float) -> np.ndarray:
    """One colour layer's MTF stencil -- restates effects.py:159-162 (0.1 mm support, 1/scale mm pixels)."""
    return compute_kernel_from_function(mtf_curve(logf, vals), 0.1, 1 / scale)


def mtf_kernel(mtf, scale: float, sharpening_strength: float = 0.0, sharpening_sigma: float = 1.0) -> np.ndarray:
    """(n, n, 3) float32 MTF stencil -- restates effects.py:165-185; `mtf` is `stock.mtf`
    (the reference's lru_cache on the stock object is a caller concern).

    The optional unsharp term runs scipy's gaussian_filter over ALL THREE axes of
    the stacked kernel, the channel axis included (effects.py:181) -- kept as is.
    """
    k = np.stack([mtf_kernel_layer(lf, v, scale) for lf, v in mtf], axis=-1, dtype=DTYPE)
    if sharpening_strength:
        blurred = ndimage.gaussian_filter(k, sigma=sharpening_sigma * scale / 50)
        k += sharpening_strength * (k - blurred)
    return k
