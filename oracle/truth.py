"""The path's stages evaluated in float64 throughout: what `oracle.stages` would give with exact arithmetic.

TEST INFRASTRUCTURE ONLY (same import rules as the rest of `oracle/`).  Not a second oracle: the parity target stays
`oracle.stages` (float32 where the reference is float32).  This module answers one question for the tests -- how much of a
difference between the HIP path and the oracle is the float32 ORACLE'S OWN distance from the exact value of the same formulas on
the same float32 inputs and tables?  tests/test_gpu_fuzz.py asserts

    |hip - truth|  <=  1e-5 * max(|truth|, 1e-3)  +  |oracle - truth|  +  conditioning

for hostile tables and with chroma NR: the contract, the float32 oracle's own distance from the exact value at
that sample, and what the case's tables make of ONE float32 ulp of each plane between two stages (`conditioning` below: measured
per sample -- the oracle's stencils are float64 FFTs rounded once, more exact than any float32 sum of products can be).
A rough table (a 3-D LUT that jumps by 1 across a cell) turns one ulp of its float32 argument into 1e-5 of its value for the
oracle and the device alike; against the truth that shows as |oracle - truth| of the same size.

Same formulas, same branch rules, same reference lines as `oracle.stages` (each function names its twin); inputs and tables
are the float32 values converted exactly; every intermediate and the result are float64.  The one quantisation that is part of
the DEFINITION of a stage is kept: the grain's uniform variates are float32(hash) * 2^-32 (noise.wgsl:27).
Piecewise-linear stages are continuous across their cell / triangle / tetrahedron boundaries, so a cell chosen differently in
float32 and float64 at a boundary changes the value by rounding only.
"""

from __future__ import annotations

import numpy as np
from scipy import fft as sfft
from scipy import ndimage

from . import stages as st

F64 = np.float64


def apply_matrix3x3(image, m):
    """stages.apply_matrix3x3 (S0) in float64."""
    image = np.asarray(image, dtype=F64)
    m = np.asarray(m, dtype=np.float32).astype(F64)
    return image @ m.T


def apply_2d_lut(image, lut):
    """stages.apply_2d_lut (S1, lut_2d.wgsl:18-108) in float64."""
    image = np.asarray(image, dtype=F64)
    lut = np.asarray(lut, dtype=np.float32).astype(F64)
    n = lut.shape[0]
    X, Y, Z = image[..., 0], image[..., 1], image[..., 2]
    S = (X + Y) + Z
    dark = S < float(np.float32(1e-12))
    inv_sum = (n - 1) / np.where(dark, 1.0, S)
    r, g = X * inv_sum, Y * inv_sum
    fr, fg = np.floor(r), np.floor(g)
    ri = np.clip(fr, 0, n - 2).astype(np.int64)
    gi = np.clip(fg, 0, n - 2).astype(np.int64)
    rf, gf = r - fr, g - fg
    fsum = rf + gf
    lower = fsum <= 1.0
    wr = np.where(lower, rf, 1.0 - gf)[..., None]
    wg = np.where(lower, gf, 1.0 - rf)[..., None]
    ws = np.where(lower, 1.0 - fsum, fsum - 1.0)[..., None]
    s_val = np.where(lower[..., None], lut[ri, gi], lut[ri + 1, gi + 1])
    out = ((lut[ri + 1, gi] * wr + lut[ri, gi + 1] * wg) + s_val * ws) * S[..., None]
    out[dark] = 0
    return out


def convolve_2d(image, kernel):
    """stages.convolve_2d / correlate_reflect101 (S2, S5: cv.filter2D semantics) without the rounding of the result."""
    image = np.asarray(image, dtype=F64)
    kernel = np.asarray(kernel, dtype=np.float32).astype(F64)
    out = np.empty_like(image)
    H, W = image.shape[:2]
    for c in range(image.shape[-1]):
        k = kernel if kernel.ndim == 2 else kernel[..., c if kernel.shape[-1] > 1 else 0]
        kh, kw = k.shape
        ay, ax = kh // 2, kw // 2
        padded = np.pad(image[..., c], ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), mode="reflect")
        fh = sfft.next_fast_len(padded.shape[0], real=True)
        fw = sfft.next_fast_len(padded.shape[1], real=True)
        spec = sfft.rfft2(padded, (fh, fw), workers=st.FFT_WORKERS) * sfft.rfft2(k[::-1, ::-1], (fh, fw), workers=st.FFT_WORKERS)
        out[..., c] = sfft.irfft2(spec, (fh, fw), workers=st.FFT_WORKERS)[kh - 1:kh - 1 + H, kw - 1:kw - 1 + W]
    return out


def log_clip(image):
    """stages.log_clip (S3) in float64 (the floor is the float32 value of 1e-6, lut_1d.wgsl:24)."""
    return np.log10(np.maximum(np.asarray(image, dtype=F64), float(np.float32(st.LOG_EPS))))


def multi_channel_interp(image, lut_1d):
    """stages.multi_channel_interp (S4 and the grain LUT): np.interp on float64 arguments."""
    image = np.asarray(image, dtype=F64)
    lut_1d = np.asarray(lut_1d, dtype=np.float32).astype(F64)
    out = np.empty_like(image)
    for c in range(3):
        out[..., c] = np.interp(image[..., c], lut_1d[0], lut_1d[1 + c])
    return out


def gaussian_noise(xs, ys, seed, mono=False):
    """stages.gaussian_noise (noise.wgsl:23-62) with the uniform variates as defined there (float32(hash) * 2^-32) and the
    Box-Muller arithmetic in float64."""
    vx, vy, vz = st.pcg3d(xs, ys, seed)
    inv = float(np.float32(1.0) / np.float32(0xFFFFFFFF))
    two_pi = float(np.float32(2.0 * 3.14159265359))
    ux = vx.astype(np.float32).astype(F64) * inv
    uy = vy.astype(np.float32).astype(F64) * inv
    u1 = np.maximum(ux, float(np.float32(1e-7)))
    r1 = np.sqrt(-2.0 * np.log(u1))
    n_r = r1 * np.cos(two_pi * uy)
    if mono:
        return np.stack([n_r, n_r, n_r], axis=-1)
    n_g = r1 * np.sin(two_pi * uy)
    u3 = np.maximum(vz.astype(np.float32).astype(F64) * inv, float(np.float32(1e-7)))
    s12 = u1 + uy
    n_b = np.sqrt(-2.0 * np.log(u3)) * np.cos(two_pi * (s12 - np.floor(s12)))
    return np.stack([n_r, n_g, n_b], axis=-1)


def apply_grain(density, grain_lut, grain_kernel, seed, mono=False):
    """stages.apply_grain (S6 + the clip, grain.wgsl:48-89) in float64."""
    density = np.asarray(density, dtype=F64)
    H, W = density.shape[:2]
    k = np.asarray(grain_kernel, dtype=np.float32).astype(F64)
    if k.ndim == 2:
        k = k[..., None]
    kh, kw = k.shape[:2]
    ay, ax = kh // 2, kw // 2
    ys = np.clip(np.arange(-ay, H + kh - 1 - ay), 0, H - 1)
    xs = np.clip(np.arange(-ax, W + kw - 1 - ax), 0, W - 1)
    noise = gaussian_noise(xs[None, :], ys[:, None], seed, mono)
    G = np.zeros((H, W, 3))
    for c in range(3):
        kc = k[..., c if k.shape[-1] > 1 else 0]
        for i in range(kh):
            for j in range(kw):
                if kc[i, j] != 0.0:
                    G[..., c] += kc[i, j] * noise[i:i + H, j:j + W, c]
    return np.maximum(density + G * multi_channel_interp(density, grain_lut), 0.0)


def burn(image, d_ref, highlight_burn, burn_scale=50.0):
    """stages.burn (S7, effects.py:360-418) in float64: area shrink, clip, Gaussian, zoom(order=1) + edge pad, subtract, clip."""
    image = np.asarray(image, dtype=F64)
    H, W = image.shape[:2]
    cell, h_lo, w_lo = st.burn_geometry(H, W, burn_scale)
    down = st.area_table(H, h_lo) @ image[..., 1] @ st.area_table(W, w_lo).T
    down = np.clip(down - float(np.float32(d_ref)), 0, None)
    blurred = ndimage.gaussian_filter(down, sigma=3, truncate=2)
    up = ndimage.zoom(blurred, cell, order=1)
    up = np.pad(up, [(0, max(H - up.shape[0], 0)), (0, max(W - up.shape[1], 0))], mode="edge")[:H, :W]
    return np.clip(image - float(np.float32(highlight_burn)) * up[..., None], 0, None)


def apply_lut_tetrahedral(image, lut, scale=1.0):
    """stages.apply_lut_tetrahedral (S8, utils.py:247-380) in float64: same truncation, edge rule and tetrahedron choice."""
    image = np.asarray(image, dtype=F64)
    lut = np.asarray(lut, dtype=np.float32).astype(F64)
    n = lut.shape[0]
    t = image * (float(scale) * (n - 1))
    i0 = np.trunc(t).astype(np.int64)
    edge = i0 >= n - 1
    d = np.where(edge, 1.0, t - i0)
    i0 = np.where(edge, n - 2, i0)
    i1 = i0 + 1
    i0 = np.where(i0 < 0, i0 + n, i0)
    i1 = np.where(i1 < 0, i1 + n, i1)
    order = np.argsort(-d, axis=-1, kind="stable")  # largest fraction first; ties change nothing (the interpolant is continuous)
    idx = i0.copy()
    out = lut[idx[..., 0], idx[..., 1], idx[..., 2]]
    prev = out
    for step in range(3):
        axis = order[..., step]
        sel = np.arange(3) == axis[..., None]
        idx = np.where(sel, i1, idx)
        cur = lut[idx[..., 0], idx[..., 1], idx[..., 2]]
        out = out + np.take_along_axis(d, axis[..., None], axis=-1) * (cur - prev)
        prev = cur
    return out


def apply_lut_trilinear(image, lut, scale=st.LUT3D_SCALE):
    """stages.apply_lut_trilinear without the final rounding."""
    image = np.asarray(image, dtype=F64)
    lut = np.asarray(lut, dtype=np.float32).astype(F64)
    n = lut.shape[0]
    t = np.clip(image * scale, 0.0, 1.0) * (n - 1)
    i0 = np.minimum(np.floor(t).astype(np.int64), n - 2)
    f = t - i0
    out = np.zeros(image.shape)
    for dr in (0, 1):
        for dg in (0, 1):
            for db in (0, 1):
                w = (np.where(dr, f[..., 0], 1 - f[..., 0]) * np.where(dg, f[..., 1], 1 - f[..., 1]) * np.where(db, f[..., 2], 1 - f[..., 2]))
                out += w[..., None] * lut[i0[..., 0] + dr, i0[..., 1] + dg, i0[..., 2] + db]
    return out


def chroma_nr_filter(image, size):
    """stages.chroma_nr_filter (effects.py:547-561) in float64; the taps are the reference's float32 taps."""
    image = np.asarray(image, dtype=F64)
    X, Y, Z = image[..., 0], image[..., 1], image[..., 2]
    denom = (X + Y) + Z
    ok = denom > float(np.float32(1e-8))
    safe = np.where(ok, denom, 1.0)
    planes = [np.where(ok, X / safe, 0.0), np.where(ok, Y / safe, 0.0)]
    k = st.chroma_kernel_1d(size).astype(F64)
    r = len(k) // 2
    for ch in (0, 1):
        pad = np.pad(planes[ch], ((0, 0), (r, r)), mode="edge")
        h = sum(k[i] * pad[:, i:i + X.shape[1]] for i in range(len(k)))
        pad = np.pad(h, ((r, r), (0, 0)), mode="edge")
        planes[ch] = sum(k[i] * pad[i:i + X.shape[0], :] for i in range(len(k)))
    cx, cy = planes
    ok = cy > float(np.float32(1e-8))
    inv = Y / np.where(ok, cy, 1.0)
    out = np.stack([cx * inv, Y, ((1.0 - cx) - cy) * inv], axis=-1)
    out[~ok] = 0
    return out


PLANES = ("exposure", "density", "mtf", "grain")  # the float32 planes between the stages (what render's `nudge` can move)


def render(image, p: st.RenderInputs, chroma_nr: int = 0, nudge: str | None = None, rel: float = 0.0):
    """stages.render (cpu_processor.py:363-405) in float64: [chroma NR ->] S0 -> S1 -> [S2] -> S3 -> S4 -> [S5] -> [S6] -> [S7] -> S8.

    nudge / rel: multiply ONE of the planes that every float32 implementation stores between two stages (PLANES: the exposure
    after S1, the density after S4, the MTF's output, the grained density) by (1 + rel) -- `conditioning` below measures with it
    how far the tables carry an ulp of such a plane into the output."""
    def moved(x, name):
        return x * (1.0 + rel) if nudge == name else x

    x = np.asarray(image, dtype=np.float32).astype(F64)
    if chroma_nr:
        x = chroma_nr_filter(x, chroma_nr)
    if p.matrix is not None:
        x = apply_matrix3x3(x, p.matrix)
    x = moved(apply_2d_lut(x, p.lut_2d), "exposure")
    if p.halation_kernel is not None:
        x = convolve_2d(x, p.halation_kernel)
    x = moved(multi_channel_interp(log_clip(x), p.lut_1d), "density")
    if p.mtf_kernel is not None:
        x = moved(convolve_2d(x, p.mtf_kernel), "mtf")
    if p.grain_lut is not None:
        gk = p.grain_kernel if p.grain_kernel is not None else np.ones((1, 1), dtype=np.float32)
        x = moved(apply_grain(x, p.grain_lut, gk, p.seed, p.grain_mono), "grain")
    if p.highlight_burn:
        x = burn(x, p.d_ref, p.highlight_burn, p.burn_scale)
    if p.lut3d_mode == "tetrahedral":
        return apply_lut_tetrahedral(x, p.lut_3d, st.LUT3D_SCALE)
    return apply_lut_trilinear(x, p.lut_3d, st.LUT3D_SCALE)


def conditioning(image, p: st.RenderInputs, chroma_nr: int = 0, ulps: float = 1.0, exact=None):
    """Per output sample: how far the output moves when each float32 plane between two stages moves by `ulps` float32 ulps
    (relative 2^-23 each), summed over the planes.  The device and the reference's own GPU path keep those planes in float32 and
    fill them with float32 sums of products; an ulp of such a plane is nobody's error.  With smooth tables it is worth ~5e-7 of a
    mid-tone and up to ~3e-6 of a dark output (an ulp of a density of 3 against an output of 0.01); stepped or noisy tables multiply
    it -- by a factor this function measures instead of assuming: the test allows exactly what the tables themselves turn one ulp
    per plane into (profiles/r04_parity_budget.txt)."""
    if exact is None:
        exact = render(image, p, chroma_nr)
    rel = ulps * 2.0 ** -23
    total = np.zeros_like(exact)
    for name in PLANES:
        if name == "mtf" and p.mtf_kernel is None:
            continue
        if name == "grain" and p.grain_lut is None:
            continue
        total += np.abs(render(image, p, chroma_nr, nudge=name, rel=rel) - exact)
    return total
