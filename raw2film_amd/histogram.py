"""Caller-side RGB histogram of the rendered bitmap (the widget next to the preview, gui.py:2225-2226).

The reference's CPU path calls ``utils.generate_histogram(image, height=80)`` (utils.py:145-223) on the uint8 frame the
processor returned; its GPU path runs histogram.wgsl's three passes on the output texture.  Here the counting pass over the
frame (3 B/px, the only part that scales with the frame) runs on the device straight from the render's uint8 output
(``r2f_histogram_u8``), and the 768-value remainder -- log1p, 3-bin smoothing, scaling, the 256-column bar image -- is the
same NumPy float32 arithmetic as the reference's, on the host.

Numba note.  Upstream runs under ``@njit``; numba promotes ``float32_array * height`` (an int64) to float64 where NumPy 2
keeps float32, so the final bar heights are computed in double there.  ``numba_semantics=True`` (default) follows the
compiled reference; ``False`` follows the same source run by plain CPython/NumPy (what tests/golden/histogram.npz holds).
The two differ by at most one pixel of bar height on a rare bin.
"""

from __future__ import annotations

import numpy as np

HUES_DEG = (29.23, 142.50, 264.05)  # utils.py:97


def _oklch_to_srgb(L: float, C: float, h_deg: float) -> np.ndarray:
    """Oklch -> non-linear sRGB in [0, 1] (unclipped).  Stand-in for ``colour.convert(..., "Oklch", "sRGB")`` (utils.py:99):
    colour-science is not installable offline, so this is Ottosson's published Oklab matrices + the sRGB OETF."""
    a, b = C * np.cos(np.radians(h_deg)), C * np.sin(np.radians(h_deg))
    l_ = L + 0.3963377774 * a + 0.2158037573 * b
    m_ = L - 0.1055613458 * a - 0.0638541728 * b
    s_ = L - 0.0894841775 * a - 1.2914855480 * b
    l, m, s = l_**3, m_**3, s_**3
    lin = np.array([
        4.0767416621 * l - 3.3077115913 * m + 0.2309699292 * s,
        -1.2684380046 * l + 2.6097574011 * m - 0.3413193965 * s,
        -0.0041960863 * l - 0.7034186147 * m + 1.7076147010 * s,
    ])
    mag = np.abs(lin)
    enc = np.where(mag <= 0.0031308, 12.92 * mag, 1.055 * mag ** (1 / 2.4) - 0.055)
    return np.sign(lin) * enc


def precompute_mix_table(red=None, green=None, blue=None) -> np.ndarray:
    """(2, 2, 2, 4) uint8 table indexed [red_active, green_active, blue_active] -> RGBA (utils.py:91-138)."""
    if red is None or green is None or blue is None:
        red, green, blue = [np.clip(_oklch_to_srgb(0.6, 0.2, h), 0, 1) * 255 for h in HUES_DEG]
    lin = [(np.asarray(c).astype(np.float32) / 255.0) ** 2.2 for c in (red, green, blue)]
    table = np.zeros((2, 2, 2, 4), dtype=np.uint8)
    for r in (0, 1):
        for g in (0, 1):
            for b in (0, 1):
                if not (r or g or b):
                    continue  # background: transparent black
                mix = np.clip(r * lin[0] + g * lin[1] + b * lin[2], 0.0, 1.0)
                table[r, g, b, 0:3] = np.round(mix ** (1.0 / 2.2) * 255.0).astype(np.uint8)
                table[r, g, b, 3] = 255
    peak = (table[1, 1, 1, :3] / 255.0) ** 2.2
    table[1, 1, 1, :3] = peak.mean() ** (1.0 / 2.2) * 255.0
    return table


MIX_TABLE = precompute_mix_table()


def bar_heights(counts, height: int = 100, numba_semantics: bool = True) -> np.ndarray:
    """(3, 256) bin counts -> (3, 256) int32 bar heights (utils.py:167-201)."""
    f = np.asarray(counts).astype(np.int32).astype(np.float32)
    max_val = f.max()
    if max_val == 0:
        max_val = np.float32(1)
    f = np.log1p(f / max_val)
    left = np.concatenate([f[:, :1], f[:, :-1]], axis=1)
    right = np.concatenate([f[:, 1:], f[:, -1:]], axis=1)
    # float32 adds, then the division by 3: correctly rounded once in either semantics
    smoothed = ((left + f + right) / np.float32(3)).astype(np.float32)
    max_val = smoothed.max()
    if max_val == 0:
        max_val = np.float32(1)
    if numba_semantics:
        return ((smoothed.astype(np.float64) * height) / np.float64(max_val)).astype(np.int32)
    return ((smoothed * np.float32(height)) / max_val).astype(np.int32)


def render_bars(heights, mix_table=MIX_TABLE, height: int = 100) -> np.ndarray:
    """(3, 256) bar heights -> (height, 256, 4) uint8 image (utils.py:203-221)."""
    y = np.arange(height)[:, None]
    on = [(y >= height - np.asarray(heights[c])[None, :]).astype(np.intp) for c in range(3)]
    return np.asarray(mix_table)[on[0], on[1], on[2]]


def histogram_from_counts(counts, mix_table=MIX_TABLE, height: int = 100, numba_semantics: bool = True) -> np.ndarray:
    return render_bars(bar_heights(counts, height, numba_semantics), mix_table, height)


def scale_to_canvas(hist, target_h: int, target_w: int) -> np.ndarray:
    """The GPU path's blit of the bar image onto the widget's canvas (shaders/scale_texture.wgsl:5-29, dispatched at
    gpu_processor.py:1885): nearest source texel at floor(uv * src_size) with uv = target pixel / target size, float32
    arithmetic like the shader.  256 x 80 texels -- host work."""
    hist = np.asarray(hist)
    sh, sw = hist.shape[:2]
    ys = ((np.arange(target_h, dtype=np.float32) / np.float32(target_h)) * np.float32(sh)).astype(np.int32)
    xs = ((np.arange(target_w, dtype=np.float32) / np.float32(target_w)) * np.float32(sw)).astype(np.int32)
    return hist[np.minimum(ys, sh - 1)[:, None], np.minimum(xs, sw - 1)[None, :]]


def generate_histogram(image, mix_table=MIX_TABLE, height: int = 100, *, ctx, numba_semantics: bool = True) -> np.ndarray:
    """``utils.generate_histogram`` for a uint8 (H, W, 3) frame: a device tensor (no copy) or a NumPy array (uploaded).
    ``ctx`` is the HipContext that counts on the device; there is no host counting path."""
    import torch

    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise ValueError("generate_histogram expects a uint8 image")
        image = torch.from_numpy(np.ascontiguousarray(image)).to(ctx.device)
    counts = ctx.histogram_counts(image.contiguous()).cpu().numpy()
    return histogram_from_counts(counts, mix_table, height, numba_semantics)
