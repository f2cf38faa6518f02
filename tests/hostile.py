"""Hostile table generators for the parity battery (tests/test_gpu_hostile.py, tests/test_gpu_fuzz.py, tools/parity_budget.py).

Every parity test used to draw its LUT contents from the three smooth analytic stand-in stocks ("What's missing" 3).
These generators produce what a real `spectral_film_lut` export may look like at its worst -- and worse: other sizes than the
defaults, non-uniform abscissae, noise from texel to texel, steps, exact 0 / 1 plateaus and blacks below the contract's 1e-3
floor.  Layouts: gpu_processor.py:307-409, 565-611; lut_1d.wgsl:43-51; grain.wgsl:78-89; utils.py:247-380.

`rough` in [0, 1] scales the texel-to-texel noise.  The end-to-end contract (1e-5 relative on the display values) is a statement
about the pipeline's arithmetic, not about tables that amplify a one-ulp difference of an intermediate by orders of magnitude, so
the whole-path tests use rough <= 0.25 and bounded curve slopes; the per-stage tests, which feed both sides identical inputs, use
rough = 1.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def lut2d(rng, n: int, rough: float = 1.0, base=None, black_texel: bool = True) -> np.ndarray:
    """(n, n, 3) input LUT: positive layer exposures per unit (X+Y+Z).  rough = 1: independent uniform texels in [0.05, 2];
    smaller values blend towards `base` (resampled to n) or a smooth ramp, keeping +-rough relative noise per texel.
    black_texel: one texel at 1e-5 among neighbours of order 1 -- exposures next to it dive towards the log clip with a relative
    slope of 1e5 per texel; for the per-stage test (identical inputs on both sides), not for the whole path."""
    g = np.linspace(0.0, 1.0, n)
    x, y = np.meshgrid(g, g, indexing="ij")
    if base is not None and base.shape[0] == n:
        smooth = np.asarray(base, dtype=np.float64)
    elif base is not None:  # bilinear resample of the base table to n x n
        b = np.asarray(base, dtype=np.float64)
        m = b.shape[0]
        t = g * (m - 1)
        i0 = np.clip(np.floor(t).astype(int), 0, m - 2)
        f = t - i0
        rows = b[i0] * (1 - f)[:, None, None] + b[i0 + 1] * f[:, None, None]
        smooth = rows[:, i0] * (1 - f)[None, :, None] + rows[:, i0 + 1] * f[None, :, None]
    else:
        smooth = np.stack([0.3 + 1.2 * x, 0.3 + 1.2 * y, 0.4 + np.abs(1.0 - x - y)], axis=-1)
    noise = rng.uniform(-1.0, 1.0, (n, n, 3))
    if rough >= 1.0:
        out = rng.uniform(0.05, 2.0, (n, n, 3))
    else:
        out = smooth * (1.0 + rough * noise)
    if black_texel:
        out[rng.integers(0, n), rng.integers(0, n)] = 1e-5
    return np.maximum(out, 1e-6).astype(F32)


def curve(rng, m: int, x_lo: float = -4.0, x_hi: float = 1.5, v_lo: float = 0.05, v_hi: float = 3.5, max_slope: float = 3.0,
          uniform: bool = False, monotone: bool = False) -> np.ndarray:
    """(4, m) table, row 0 = abscissae (NON-uniform unless asked: random steps between 0.05x and 4x the mean, so the device's
    uniform-axis guess is off by many cells), rows 1..3 = a bounded random walk whose slope stays under max_slope."""
    if uniform:
        xp = np.linspace(x_lo, x_hi, m)
    else:
        steps = rng.uniform(0.05, 4.0, m - 1) ** 2
        xp = x_lo + (x_hi - x_lo) * np.concatenate([[0.0], np.cumsum(steps)]) / steps.sum()
    xp = xp.astype(F32).astype(np.float64)
    xp = np.maximum.accumulate(xp)
    rows = [xp]
    dx = np.diff(xp)
    for c in range(3):
        s = rng.uniform(0.0 if monotone else -1.0, 1.0, m - 1) * max_slope
        f = np.concatenate([[0.0], np.cumsum(s * dx)])
        f = v_lo + (f - f.min()) / max(f.max() - f.min(), 1e-9) * (v_hi - v_lo) * rng.uniform(0.6, 1.0)
        rows.append(f)
    return np.stack(rows).astype(F32)


def grain_lut(rng, m: int, steps: int = 6, amp: float = 0.06) -> np.ndarray:
    """(4, m) grain LUT on a uniform density axis [0, 4]: piecewise constant amplitudes with `steps` jumps per channel (each jump
    happens across ONE cell, grain.wgsl:78-89 interpolates linearly inside it)."""
    xp = np.linspace(0.0, 4.0, m)
    rows = [xp]
    for c in range(3):
        levels = rng.uniform(0.15, 1.0, steps + 1) * amp
        edges = np.sort(rng.choice(np.arange(1, m - 1), steps, replace=False))
        rows.append(levels[np.searchsorted(edges, np.arange(m), side="right")])
    return np.stack(rows).astype(F32)


def lut3d(rng, n: int, rough: float = 1.0, base=None) -> np.ndarray:
    """(n, n, n, 3) output LUT with exact 0 and exact 1 plateaus and blacks far below 1e-3.
    rough = 1: independent uniform texels in [0, 1] with 5 % exact zeros and 5 % exact ones.
    Otherwise: a print-like exponential fall-off along the density axes -- values from 1 down to ~1e-6, the way a print stock's
    transmittance behaves, so relative error stays meaningful below the floor -- with +-rough relative texel noise, an exact-1
    plateau at low density and an exact-0 plateau at the far corner."""
    if rough >= 1.0:
        out = rng.uniform(0.0, 1.0, (n, n, n, 3))
        sel = rng.uniform(0.0, 1.0, (n, n, n, 3))
        out[sel < 0.05] = 0.0
        out[sel > 0.95] = 1.0
        return out.astype(F32)
    if base is not None and base.shape[0] == n:
        v = np.asarray(base, dtype=np.float64) ** 2  # squares a [5e-3, 1] stand-in down to 2.5e-5
    else:
        d = np.linspace(0.0, 4.0, n)
        dr, dg, db = np.meshgrid(d, d, d, indexing="ij")
        mix = np.stack([0.8 * dr + 0.1 * dg + 0.1 * db, 0.1 * dr + 0.8 * dg + 0.1 * db, 0.1 * dr + 0.1 * dg + 0.8 * db], axis=-1)
        v = 10.0 ** (-2.8 * (mix - 0.9))  # 1 at D <= 0.9, ~3e-5 at D = 2.5: every frame with shadows and highlights reaches both plateaus
    v = v * (1.0 + rough * rng.uniform(-1.0, 1.0, v.shape))
    v = 1.02 * v - 3e-5  # exact zeros where the fall-off has reached ~3e-5, exact ones on top
    return np.clip(v, 0.0, 1.0).astype(F32)


def roughen(rng, p, n2: int, m1: int, n3: int, rough2: float = 0.2, rough3: float = 0.1):
    """Replace the tables of oracle RenderInputs `p` by hostile ones the END-TO-END contract can bear: a 2-D LUT of side n2 with
    +-rough2 relative texel noise around the stock's own, a monotone density curve of m1 points on a non-uniform axis (slopes
    <= 1.5), a print-like 3-D LUT of side n3 that reaches exact 1 and exact 0, a stepped grain LUT."""
    # (texel noise scaled with the texel pitch, so the table's slope per unit chromaticity -- what turns one ulp of the index into
    # a relative error of the exposure -- does not grow with n2; and n3 >= 17: across a coarser cell a print-like fall-off drops by
    # a factor of 25, and linear interpolation inside it has 7 times the relative slope of the curve it samples.  Measured with
    # R2F_FUZZ_CASES=150: 11 of 50 hostile cases over 1e-5, up to 3.9e-5, all with n3 = 9 or n2 >= 100 at full noise)
    p.lut_2d = lut2d(rng, n2, rough2 * min(1.0, 32.0 / (n2 - 1)), base=p.lut_2d, black_texel=False)
    p.lut_1d = curve(rng, m1, v_lo=0.08, v_hi=3.6, max_slope=1.5, monotone=True)
    p.lut_3d = lut3d(rng, n3, rough3)
    if p.grain_lut is not None:
        p.grain_lut = grain_lut(rng, 256, amp=0.03)
    return p


# ---------------------------------------------------------------------------------------------- 12-byte scratch element
# Frames for the search behind the guard of the halation's 12-byte FFT scratch element (r2f_api.hip dyn_rule).
# The element rounds every scratch value to 2^-37 of itself, so what it costs a pixel scales with the energy of the WINDOW the
# pixel shares, while the guard sees max |x| and min x of the frame: the worst frames put as much bright, spectrally rich signal as
# possible into a window and a shadow further than the stencil's reach (43 px at the 100 MP pitch) from all of it.
SCRATCH96_KINDS = ("holes", "blocks", "stripes", "checker", "gradient", "half", "speculars")
SCRATCH96_FILLS = ("u50", "u0", "binary", "jitter", "lognormal")


def scratch96_frame(rng, H: int, W: int, kind: str, fill: str, lo: float, hi: float) -> np.ndarray:
    """(H, W, 3) float32 exposure-like frame: a shadow field at `lo` (x U(1, 3), or flat), bright regions at up to `hi`."""

    def bright(shape):
        if fill == "u50":
            v = rng.uniform(0.5, 1.0, shape)
        elif fill == "u0":
            v = rng.uniform(0.0, 1.0, shape)
        elif fill == "binary":
            v = rng.integers(0, 2, shape).astype(np.float64)
        elif fill == "jitter":  # nearly flat, every mantissa bit in use
            v = 1.0 - rng.uniform(0.0, 2.0 ** -int(rng.integers(4, 20)), shape)
        else:  # lognormal body clipped at 1: a few speculars over a bright body
            v = np.minimum(np.exp(rng.normal(-2.0, 1.0, shape)), 1.0)
        return hi * v

    dark = lo * (rng.uniform(1.0, 3.0, (H, W, 3)) if rng.integers(0, 2) else np.ones((H, W, 3)))
    img = dark.copy()
    reach = 48
    if kind == "holes":  # bright everywhere but in a few dark squares whose centres see no bright sample
        img = bright((H, W, 3))
        for _ in range(int(rng.integers(2, 9))):
            s = int(rng.integers(2 * reach + 4, 2 * reach + 80))
            y, x = int(rng.integers(0, max(H - s, 1))), int(rng.integers(0, max(W - s, 1)))
            img[y:y + s, x:x + s] = dark[y:y + s, x:x + s]
    elif kind == "blocks":
        for _ in range(int(rng.integers(1, 7))):
            h, w = int(rng.integers(8, H // 2)), int(rng.integers(8, W // 2))
            y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
            img[y:y + h, x:x + w] = bright((h, w, 3))
    elif kind == "stripes":  # bright stripes over part of the frame, a dark field beside them
        p = int(rng.integers(2, 9))
        split = int(rng.integers(W // 3, 2 * W // 3))
        b = bright((H, W, 3))
        if rng.integers(0, 2):
            img[::p, :split] = b[::p, :split]
        else:
            img[:, :split:p] = b[:, :split:p]
    elif kind == "checker":
        c = int(rng.integers(2 * reach + 4, 3 * reach))
        yy, xx = np.indices((H, W))
        m = ((yy // c + xx // c) % 2).astype(bool)
        img[m] = bright((H, W, 3))[m]
    elif kind == "gradient":  # a smooth ramp from hi down to the shadows, bright noise on top of its upper half
        ramp = np.geomspace(hi, lo, W)[None, :, None] * np.ones((H, 1, 3))
        img = np.maximum(ramp * rng.uniform(0.5, 1.0, (H, W, 3)), dark)
    elif kind == "half":
        split = int(rng.integers(W // 4, 3 * W // 4))
        img[:, :split] = bright((H, split, 3))
    else:  # isolated speculars and a patch (tools/scratch96_probe.py)
        img[::int(rng.integers(40, 120)), ::int(rng.integers(40, 160))] = hi
        img[H // 2:H // 2 + 40, W // 2:W // 2 + 60] = bright((40, 60, 3))
    # the guard's hi and lo are the frame's own extremes: pin them so that the ratio is the one asked for
    img = np.clip(img, lo, hi)
    img[0, 0] = lo
    img[H - 1, W - 1] = hi
    return img.astype(F32)
