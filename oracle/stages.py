"""Oracle restatement of the per-pixel stages S0..S8 and their orchestration.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py for the import rules and for
which functions are pinned against the reference and which are "parity unpinned").

Arrays are float32 (H, W, 3) interleaved, like the reference's (`cpu_processor.py:136`).
Stage numbering follows SURVEY.md section 8a.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import math

import numpy as np
from scipy import fft as sfft
from scipy import ndimage

F32 = np.float32

# data.py:128-135 (Rec.709 linear RGB -> CIE XYZ), the S0 matrix for synthetic linear-RGB frames.
REC709_TO_XYZ = np.array(
    [[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]],
    dtype=F32,
)

FFT_WORKERS = -1  # threads scipy.fft may use inside one correlation (-1 = all); the baseline pool sets 1

LOG_EPS = 1e-6  # lut_1d.wgsl:24
LUT3D_SCALE = 0.25  # cpu_processor.py:405, lut_3d.wgsl:1


# ------------------------------------------------------------------ pre-path: chroma NR
def chroma_kernel_1d(size: int) -> np.ndarray:
    """effects.py:554-556 + gaussian_kernel_1d :421-435: 2*size+1 taps, sigma = 0.3*((taps-1)/2 - 1) + 0.8,
    exp in double, stored and normalised in float32.  PINNED bit-exact (tests/golden/chroma_nr.npz)."""
    taps = int(size) * 2 + 1
    sigma = 0.3 * ((taps - 1) * 0.5 - 1) + 0.8
    x = np.arange(taps) - taps // 2
    k = np.exp(-(x * x) / (2.0 * sigma * sigma)).astype(F32)
    k /= k.sum()
    return k


def xyz_to_xyY(image: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """effects.py:496-518: x = X/(X+Y+Z), y = Y/(X+Y+Z) (0 when the sum <= eps), third channel Y.  float32."""
    image = np.asarray(image, dtype=F32)
    X, Y, Z = image[..., 0], image[..., 1], image[..., 2]
    denom = (X + Y) + Z
    ok = denom > F32(eps)
    safe = np.where(ok, denom, F32(1))
    return np.stack([np.where(ok, X / safe, F32(0)), np.where(ok, Y / safe, F32(0)), Y], axis=-1).astype(F32)


def xyY_to_xyz(image: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """effects.py:521-544: inv = Y/y; X = x*inv, Z = (1 - x - y)*inv; all 0 when y <= eps.  float32."""
    image = np.asarray(image, dtype=F32)
    cx, cy, Y = image[..., 0], image[..., 1], image[..., 2]
    ok = cy > F32(eps)
    inv = Y / np.where(ok, cy, F32(1))
    out = np.stack([cx * inv, Y, ((F32(1.0) - cx) - cy) * inv], axis=-1)
    out[~ok] = 0
    return out.astype(F32)


def chroma_nr_filter(image: np.ndarray, size: int) -> np.ndarray:
    """effects.chroma_nr_filter, effects.py:547-561: XYZ -> xyY, separable Gaussian on the x and y planes only
    (horizontal then vertical, coordinates clamped to the frame, float64 accumulator under numba, float32 stored
    between the passes), back to XYZ.  Pinned to 1e-6 against the reference's own code run by CPython (which
    accumulates in float32, see tools/make_golden.py)."""
    xyY = xyz_to_xyY(image)
    k = chroma_kernel_1d(size).astype(np.float64)
    r = len(k) // 2
    out = xyY.copy()
    for ch in (0, 1):
        plane = xyY[..., ch].astype(np.float64)
        pad = np.pad(plane, ((0, 0), (r, r)), mode="edge")
        h = sum(k[i] * pad[:, i:i + plane.shape[1]] for i in range(len(k))).astype(F32).astype(np.float64)
        pad = np.pad(h, ((r, r), (0, 0)), mode="edge")
        out[..., ch] = sum(k[i] * pad[i:i + plane.shape[0], :] for i in range(len(k))).astype(F32)
    return xyY_to_xyz(out)


# --------------------------------------------------------------------------- S0
def apply_matrix3x3(image: np.ndarray, m: np.ndarray) -> np.ndarray:
    """S0 camera->scene 3x3: out = M . in per pixel, fp32.

    Upstream this happens inside LibRaw (`raw_conversion.py:38-48`,
    `output_color=ColorSpace(5)` = XYZ); for linear-Rec.709 inputs M = data.py:128-135.
    Evaluated as ((m0*r + m1*g) + m2*b) in float32.
    """
    image = np.asarray(image, dtype=F32)
    m = np.asarray(m, dtype=F32)
    r, g, b = image[..., 0], image[..., 1], image[..., 2]
    out = np.empty_like(image)
    for i in range(3):
        out[..., i] = (m[i, 0] * r + m[i, 1] * g) + m[i, 2] * b
    return out


# --------------------------------------------------------------------------- S1
def apply_2d_lut(image: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """S1 chromaticity-indexed 2-D input LUT.  PARITY UNPINNED (sfl.xy_lut.apply_2d_lut,
    called at cpu_processor.py:364); restates lut_2d.wgsl:18-108 in float32.

    S = X+Y+Z; S < 1e-12 -> 0.  r = X*(n-1)/S, g = Y*(n-1)/S; cell = clamp(floor, 0, n-2);
    fractions are `fract` of the UNclamped coordinate; barycentric on the lower triangle
    if rf+gf <= 1 else on the upper one; result * S.  LUT texel for (x-index, y-index) is
    lut[x, y] (lut_2d.wgsl:10-16 loads texel (y, x); rows are the first numpy axis,
    gpu_processor.py:364-376).
    """
    image = np.asarray(image, dtype=F32)
    lut = np.asarray(lut, dtype=F32)
    n = lut.shape[0]
    X, Y, Z = image[..., 0], image[..., 1], image[..., 2]
    S = (X + Y) + Z
    dark = S < F32(1e-12)
    Ssafe = np.where(dark, F32(1.0), S)
    inv_sum = F32(n - 1) / Ssafe
    r = X * inv_sum
    g = Y * inv_sum
    fr = np.floor(r)
    fg = np.floor(g)
    ri = np.clip(fr, 0, n - 2).astype(np.int64)
    gi = np.clip(fg, 0, n - 2).astype(np.int64)
    rf = (r - fr).astype(F32)
    gf = (g - fg).astype(F32)
    fsum = rf + gf
    lower = fsum <= F32(1.0)
    r_val = lut[ri + 1, gi]
    g_val = lut[ri, gi + 1]
    s_lo = lut[ri, gi]
    s_hi = lut[ri + 1, gi + 1]
    one = F32(1.0)
    wr = np.where(lower, rf, one - gf)[..., None]
    wg = np.where(lower, gf, one - rf)[..., None]
    ws = np.where(lower, one - fsum, fsum - one)[..., None]
    s_val = np.where(lower[..., None], s_lo, s_hi)
    out = ((r_val * wr + g_val * wg) + s_val * ws) * S[..., None]
    out[dark] = 0
    return out.astype(F32)


# ------------------------------------------------------------------ S2 / S5 core
def _reflect101_pad(plane: np.ndarray, top: int, bottom: int, left: int, right: int) -> np.ndarray:
    # np.pad(mode="reflect") IS reflect-101 (edge sample not repeated) and folds repeatedly
    # when the pad exceeds the extent, like cv::borderInterpolate(BORDER_REFLECT_101).
    return np.pad(plane, ((top, bottom), (left, right)), mode="reflect")


def correlate_reflect101(plane: np.ndarray, kernel: np.ndarray, method: str = "fft") -> np.ndarray:
    """`cv.filter2D(plane, -1, kernel)` semantics (effects.py:148-153): correlation (kernel
    not flipped), anchor at the kernel centre (kh//2, kw//2), BORDER_REFLECT_101, float32 in,
    float32 out.  OpenCV is absent here, so this restates its documented behaviour; the
    arithmetic runs in float64 (OpenCV's own DFT path rounds differently at the 1e-7 level).

    method="fft": zero-padded real FFT in float64 (exact to ~1e-15 of the plane's peak).
    method="direct": scipy.ndimage.correlate(mode="mirror") -- same semantics, used to cross-check.
    """
    plane = np.asarray(plane, dtype=F32)
    k = np.asarray(kernel, dtype=np.float64)
    kh, kw = k.shape
    if method == "direct":
        assert kh % 2 == 1 and kw % 2 == 1
        return ndimage.correlate(plane.astype(np.float64), k, mode="mirror").astype(F32)
    ay, ax = kh // 2, kw // 2
    H, W = plane.shape
    padded = _reflect101_pad(plane.astype(np.float64), ay, kh - 1 - ay, ax, kw - 1 - ax)
    fh = sfft.next_fast_len(padded.shape[0], real=True)
    fw = sfft.next_fast_len(padded.shape[1], real=True)
    # correlation == convolution with the flipped kernel; "valid" part starts at (kh-1, kw-1)
    spec = sfft.rfft2(padded, (fh, fw), workers=FFT_WORKERS) * sfft.rfft2(k[::-1, ::-1], (fh, fw), workers=FFT_WORKERS)
    full = sfft.irfft2(spec, (fh, fw), workers=FFT_WORKERS)
    return full[kh - 1 : kh - 1 + H, kw - 1 : kw - 1 + W].astype(F32)


def convolve_2d(image: np.ndarray, kernel: np.ndarray, method: str = "fft") -> np.ndarray:
    """effects.py:146-156: 2-D kernel -> same taps for every channel; 3-D kernel -> channel c
    filtered with kernel[..., c].  Returns a new float32 array (the reference works in place)."""
    image = np.asarray(image, dtype=F32)
    kernel = np.asarray(kernel)
    out = np.empty_like(image)
    for c in range(image.shape[-1]):
        kc = kernel if kernel.ndim == 2 else kernel[..., c if kernel.shape[-1] > 1 else 0]
        out[..., c] = correlate_reflect101(image[..., c], kc, method)
    return out


def halation(image: np.ndarray, halation_kernel: np.ndarray, method: str = "fft") -> np.ndarray:
    """S2, effects.py:266-287 with the kernel from `oracle.kernels.compute_halation_kernel`.
    Acts on linear exposure, before the log (cpu_processor.py:368-378)."""
    return convolve_2d(image, halation_kernel, method)


def film_sharpness(image: np.ndarray, mtf_kernel: np.ndarray, method: str = "fft") -> np.ndarray:
    """S5, effects.py:188-197 with the kernel from `oracle.kernels.mtf_kernel`.  Acts on density."""
    return convolve_2d(image, mtf_kernel, method)


# ----------------------------------------------------------------------- S3 / S4
def log_clip(image: np.ndarray) -> np.ndarray:
    """S3.  PARITY UNPINNED (sfl.utils.log_clip, cpu_processor.py:378); restates
    lut_1d.wgsl:23-26: log10(max(x, 1e-6)), float32."""
    return np.log10(np.maximum(np.asarray(image, dtype=F32), F32(LOG_EPS))).astype(F32)


def multi_channel_interp(image: np.ndarray, lut_1d: np.ndarray) -> np.ndarray:
    """S4.  PARITY UNPINNED (sfl.utils.multi_channel_interp, cpu_processor.py:380).
    LUT layout (gpu_processor.py:307-328): row 0 = xp, rows 1..3 = per-channel fp.
    out_c = np.interp(x_c, xp, fp_c): linear, clamped at both ends (lut_1d.wgsl:43-51
    without the sampler's half-texel shift).  float64 inside np.interp, float32 out."""
    image = np.asarray(image, dtype=F32)
    lut_1d = np.asarray(lut_1d)
    xp = lut_1d[0].astype(np.float64)
    out = np.empty_like(image)
    for c in range(3):
        out[..., c] = np.interp(image[..., c].astype(np.float64), xp, lut_1d[1 + c].astype(np.float64))
    return out


# --------------------------------------------------------------------------- S6
def pcg3d(x: np.ndarray, y: np.ndarray, seed: int):
    """noise.wgsl:14-20 -- PCG3D hash on uint32 with wrap-around.  Returns (vx, vy, vz)."""
    with np.errstate(over="ignore"):
        mul, inc = np.uint32(1664525), np.uint32(1013904223)
        vx = x.astype(np.uint32) * mul + inc
        vy = y.astype(np.uint32) * mul + inc
        vz = np.full(np.broadcast(x, y).shape, np.uint32(seed & 0xFFFFFFFF), dtype=np.uint32) * mul + inc
        vx, vy = np.broadcast_arrays(vx, vy)
        vx = vx.copy()
        vy = vy.copy()

        def mix():
            nonlocal vx, vy, vz
            vx = vx + vy * vz
            vy = vy + vz * vx
            vz = vz + vx * vy

        mix()
        s = np.uint32(16)
        vx = vx ^ (vx >> s)
        vy = vy ^ (vy >> s)
        vz = vz ^ (vz >> s)
        mix()
    return vx, vy, vz


_TWO_PI = F32(2.0 * 3.14159265359)  # noise.wgsl:39 (abstract-float constant folded, then f32)
_INV_U32 = F32(1.0) / F32(0xFFFFFFFF)  # noise.wgsl:27: 1/f32(0xffffffff) == 2**-32


def gaussian_noise(xs: np.ndarray, ys: np.ndarray, seed: int, mono: bool = False) -> np.ndarray:
    """S6a white Gaussian field N(x, y, c) for integer pixel coordinates (broadcastable
    `xs`, `ys`) -- restates noise.wgsl:23-62 (RGB) and noise_bw.wgsl:23-56 (mono), float32.

    u = f32(hash) * 2**-32; n_r = sqrt(-2 ln max(u.x,1e-7)) cos(2pi u.y), n_g = ... sin(...),
    n_b = sqrt(-2 ln max(u.z,1e-7)) cos(2pi fract(max(u.x,1e-7) + u.y)); mono: n_r in all three.
    """
    vx, vy, vz = pcg3d(xs, ys, seed)
    ux = vx.astype(F32) * _INV_U32
    uy = vy.astype(F32) * _INV_U32
    u1 = np.maximum(ux, F32(1e-7))
    r1 = np.sqrt(F32(-2.0) * np.log(u1))
    th1 = _TWO_PI * uy
    n_r = r1 * np.cos(th1)
    if mono:
        return np.stack([n_r, n_r, n_r], axis=-1).astype(F32)
    n_g = r1 * np.sin(th1)
    u3 = np.maximum(vz.astype(F32) * _INV_U32, F32(1e-7))
    s12 = u1 + uy
    th2 = _TWO_PI * (s12 - np.floor(s12))
    n_b = np.sqrt(F32(-2.0) * np.log(u3)) * np.cos(th2)
    return np.stack([n_r, n_g, n_b], axis=-1).astype(F32)


def grain_field(
    H: int, W: int, seed: int, grain_kernel: np.ndarray, mono: bool = False, row0: int = 0, H_global: int | None = None,
    col0: int = 0, W_global: int | None = None,
) -> np.ndarray:
    """S6b G = K_g (*) N with the noise texture clamped at the image border
    (grain.wgsl:48-75: start = p - k//2, taps in row-major order, `clamp(coord, 0, dims-1)`).
    Rows [row0, row0+H) x columns [col0, col0+W) of a frame of H_global x W_global pixels (row shards and
    test windows evaluate the hash at true global coordinates, so the field is identical for any sharding)."""
    if H_global is None:
        H_global = row0 + H
    if W_global is None:
        W_global = col0 + W
    k = np.asarray(grain_kernel, dtype=np.float64)
    if k.ndim == 2:
        k = k[..., None]
    kh, kw = k.shape[:2]
    ay, ax = kh // 2, kw // 2
    ys = np.clip(np.arange(row0 - ay, row0 + H + kh - 1 - ay), 0, H_global - 1)
    xs = np.clip(np.arange(col0 - ax, col0 + W + kw - 1 - ax), 0, W_global - 1)
    noise = gaussian_noise(xs[None, :], ys[:, None], seed, mono).astype(np.float64)
    out = np.zeros((H, W, 3), dtype=np.float64)
    for c in range(3):
        kc = k[..., c if k.shape[-1] > 1 else 0]
        for i in range(kh):
            for j in range(kw):
                if kc[i, j] != 0.0:
                    out[..., c] += kc[i, j] * noise[i : i + H, j : j + W, c]
    return out.astype(F32)


def apply_grain(
    density: np.ndarray,
    grain_lut: np.ndarray,
    grain_kernel: np.ndarray,
    seed: int,
    mono: bool = False,
    row0: int = 0,
    H_global: int | None = None,
    col0: int = 0,
    W_global: int | None = None,
) -> np.ndarray:
    """S6 = effects.py:220-236 + the clip at cpu_processor.py:397.  PARITY UNPINNED
    (sfl.generate_grain / FilmSpectral.grain_transform); restates grain.wgsl:78-89:
    out_c = D_c + G_c * interp(D_c; xp_g, fp_g[c]) with the (4, m) LUT indexed by density,
    then max(out, 0)."""
    density = np.asarray(density, dtype=F32)
    H, W = density.shape[:2]
    G = grain_field(H, W, seed, grain_kernel, mono, row0, H_global, col0, W_global)
    factor = multi_channel_interp(density, grain_lut)
    return np.maximum(density + G * factor, F32(0.0)).astype(F32)


# --------------------------------------------------------------------------- S7
def resize_area(plane: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`cv.resize(plane, (out_w, out_h), interpolation=cv.INTER_AREA)` for a float32 plane that is being
    shrunk -- the down-sample of effects.down_up_blur (effects.py:370-375).  PARITY UNPINNED: OpenCV is not
    installed; this restates its documented area resampling (imgproc/resize.cpp, computeResizeAreaTab): with
    scale = src/dst, destination sample d averages the source interval [d*scale, (d+1)*scale): whole source
    pixels weigh 1/cell, the two partially covered ones by their covered fraction (fractions below 1e-3 are
    dropped, as OpenCV does), cell = min(scale, src - d*scale).  Separable; float64 here, float32 out."""
    plane = np.asarray(plane, dtype=F32)
    wy, wx = area_table(plane.shape[0], out_h), area_table(plane.shape[1], out_w)
    return (wy @ plane.astype(np.float64) @ wx.T).astype(F32)


def area_table(ssize: int, dsize: int) -> np.ndarray:
    """(dsize, ssize) weights of OpenCV's computeResizeAreaTab for one axis (see resize_area)."""
    scale = ssize / dsize
    w = np.zeros((dsize, ssize))
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1 = int(np.ceil(f1))
        s2 = min(int(np.floor(f2)), ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            w[d, s1 - 1] += (s1 - f1) / cell
        w[d, s1:s2] += 1.0 / cell
        if f2 - s2 > 1e-3:
            w[d, s2] += min(min(f2 - s2, 1.0), cell) / cell
    return w


def burn_geometry(H: int, W: int, burn_scale: float):
    """(cell, h_lo, w_lo) of effects.down_up_blur: cell = ceil(min(H, W)/scale), low-res size (H//cell, W//cell)."""
    cell = int(np.ceil(min(H, W) / burn_scale))
    return cell, H // cell, W // cell


def burn_map(green: np.ndarray, d_ref: float, burn_scale: float = 50.0) -> np.ndarray:
    """The blurred low-res highlight map of effects.burn / down_up_blur (effects.py:360-418), BEFORE the
    up-sample: INTER_AREA shrink of the green density, clip(x - d_ref, 0), gaussian_filter(sigma=3, truncate=2)."""
    green = np.asarray(green, dtype=F32)
    cell, h_lo, w_lo = burn_geometry(green.shape[0], green.shape[1], burn_scale)
    down = resize_area(green, h_lo, w_lo)
    down = np.clip(down - F32(d_ref), 0, None)
    return ndimage.gaussian_filter(down, sigma=3, truncate=2)


def burn(image: np.ndarray, d_ref: float, highlight_burn: float, burn_scale: float = 50.0) -> np.ndarray:
    """S7 highlight burn, effects.py:396-418 for a 3-channel image: the green density's blurred highlight
    map, up-sampled with ndimage.zoom(order=1) and edge-padded to the frame (effects.py:381-388), is
    subtracted from all three channels; then clip at 0.  d_ref = stock.d_ref[1] (or [0])."""
    image = np.asarray(image, dtype=F32)
    H, W = image.shape[:2]
    cell, _, _ = burn_geometry(H, W, burn_scale)
    blurred = burn_map(image[..., 1], d_ref, burn_scale)
    return burn_apply(image, blurred, cell, highlight_burn)


def burn_upsample(blurred: np.ndarray, cell: int, H: int, W: int) -> np.ndarray:
    """effects.py:381-388: ndimage.zoom(order=1) by the shrink factor, edge-padded / cropped to (H, W)."""
    up = ndimage.zoom(blurred, cell, order=1)
    return np.pad(up, [(0, max(H - up.shape[0], 0)), (0, max(W - up.shape[1], 0))], mode="edge")[:H, :W]


def burn_apply(image: np.ndarray, blurred: np.ndarray, cell: int, highlight_burn: float, row0: int = 0,
               H_global: int | None = None) -> np.ndarray:
    """Subtract the up-sampled map from rows [row0, row0 + image rows) of an H_global-row frame; clip at 0."""
    image = np.asarray(image, dtype=F32)
    H = image.shape[0] + row0 if H_global is None else H_global
    up = burn_upsample(blurred, cell, H, image.shape[1])[row0:row0 + image.shape[0]]
    return np.clip(image - F32(highlight_burn) * up[..., None], 0, None).astype(F32)


# --------------------------------------------------------------------------- S8
def apply_lut_tetrahedral(image: np.ndarray, lut: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """S8 tetrahedral 3-D LUT -- restates utils.py:247-380 with numba's promotion rules
    (float32 pixel * float64 scale -> float64; float32 LUT differences; float64 blend; one
    rounding on the float32 store).  PINNED bit-for-bit against tests/golden/tetrahedral.npz.

    t = x*scale*(n-1); i0 = int(t) (truncation); i0 >= n-1 -> (n-2, d=1) else d = t - i0.
    Six-way tetrahedron select with the reference's `>=` tie rules (utils.py:298-376).
    Negative base indices wrap like Python/numba indexing does (only reachable for x <= -1/(scale*(n-1))).
    """
    image = np.asarray(image, dtype=F32)
    lut = np.asarray(lut, dtype=F32)
    n = lut.shape[0]
    s = float(scale) * (n - 1)
    t = image.astype(np.float64) * s
    i0 = np.trunc(t).astype(np.int64)
    edge = i0 >= n - 1
    d = np.where(edge, 1.0, t - i0)
    i0 = np.where(edge, n - 2, i0)
    i1 = i0 + 1
    i0 = np.where(i0 < 0, i0 + n, i0)
    i1 = np.where(i1 < 0, i1 + n, i1)
    r0, g0, b0 = i0[..., 0], i0[..., 1], i0[..., 2]
    r1, g1, b1 = i1[..., 0], i1[..., 1], i1[..., 2]
    dr, dg, db = d[..., 0, None], d[..., 1, None], d[..., 2, None]
    c000 = lut[r0, g0, b0]
    c111 = lut[r1, g1, b1]
    drv, dgv, dbv = dr[..., 0], dg[..., 0], db[..., 0]
    # tetrahedron id per pixel, same branch order as utils.py:298-376
    t1 = (drv >= dgv) & (dgv >= dbv)  # dr >= dg >= db : c100, c110
    t2 = (drv >= dgv) & ~(dgv >= dbv) & (drv >= dbv)  # dr >= db > dg : c100, c101
    t3 = (drv >= dgv) & ~(dgv >= dbv) & ~(drv >= dbv)  # db > dr >= dg : c001, c101
    t4 = ~(drv >= dgv) & (dbv >= dgv)  # db >= dg > dr : c001, c011
    t5 = ~(drv >= dgv) & ~(dbv >= dgv) & (dbv >= drv)  # dg > db >= dr : c010, c011
    # t6: dg > dr > db : c010, c110
    # first step axis / second step axis / third step axis per tetrahedron
    first = np.where(t1 | t2, 0, np.where(t3 | t4, 2, 1))
    third = np.where(t1, 2, np.where(t2 | t3, 1, np.where(t4 | t5, 0, 2)))
    second = 3 - first - third
    idx0 = np.stack([r0, g0, b0], axis=-1)
    idx1 = np.stack([r1, g1, b1], axis=-1)

    def corner(step_mask):
        sel = np.where(step_mask, idx1, idx0)
        return lut[sel[..., 0], sel[..., 1], sel[..., 2]]

    ax = np.arange(3)
    m_a = ax == first[..., None]
    m_b = m_a | (ax == second[..., None])
    ca = corner(m_a)
    cb = corner(m_b)
    d_first = np.take_along_axis(d, first[..., None], axis=-1)
    d_second = np.take_along_axis(d, second[..., None], axis=-1)
    d_third = np.take_along_axis(d, third[..., None], axis=-1)
    # ((c000 + d1*(ca-c000)) + d2*(cb-ca)) + d3*(c111-cb): float32 differences, float64 blend
    out = c000 + d_first * (ca - c000)
    out = out + d_second * (cb - ca)
    out = out + d_third * (c111 - cb)
    return out.astype(F32)


def apply_lut_trilinear(image: np.ndarray, lut: np.ndarray, scale: float = LUT3D_SCALE) -> np.ndarray:
    """GPU-variant S8 (lut_3d.wgsl:27-40): clamp(x*scale, 0, 1), trilinear between texel
    centres, fp32 LUT (the upstream texture is rgba16float; not modelled).  Offered by the
    build as `lut3d_mode=1`; the parity target of the path is the tetrahedral form."""
    image = np.asarray(image, dtype=F32)
    lut = np.asarray(lut, dtype=np.float64)
    n = lut.shape[0]
    t = np.clip(image.astype(np.float64) * scale, 0.0, 1.0) * (n - 1)
    i0 = np.minimum(np.floor(t).astype(np.int64), n - 2)
    f = t - i0
    out = np.zeros(image.shape, dtype=np.float64)
    for dr in (0, 1):
        for dg in (0, 1):
            for db in (0, 1):
                w = (
                    np.where(dr, f[..., 0], 1 - f[..., 0])
                    * np.where(dg, f[..., 1], 1 - f[..., 1])
                    * np.where(db, f[..., 2], 1 - f[..., 2])
                )
                out += w[..., None] * lut[i0[..., 0] + dr, i0[..., 1] + dg, i0[..., 2] + db]
    return out.astype(F32)


def to_uint8(image: np.ndarray) -> np.ndarray:
    """S9 cpu_processor.py:407: (image * 255).astype(uint8) -- truncation, fp32 product.
    (Values are in [0, 1] after the 3-D LUT; the clip only guards the cast.)"""
    return np.clip(np.asarray(image, dtype=F32) * F32(255.0), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ orchestration T
@dataclass
class RenderInputs:
    """Everything `CpuProcessor.process` has in hand when the hot loop starts
    (cpu_processor.py:363): LUT arrays and per-render stencils.  None disables a stage
    exactly where the reference gates it (cpu_processor.py:368,382,387)."""

    lut_2d: np.ndarray  # (n, n, 3)  negative_film.get_input_lut(...)       cpu_processor.py:160
    lut_1d: np.ndarray  # (4, m)     negative_film.get_density_curve(...)   cpu_processor.py:182
    lut_3d: np.ndarray  # (n,n,n,3)  create_lut(..., linear_scaling=4)      cpu_processor.py:232
    matrix: np.ndarray | None = None  # S0; None = input already XYZ
    halation_kernel: np.ndarray | None = None  # (k, k, 3)
    mtf_kernel: np.ndarray | None = None  # (k, k, 3)
    grain_lut: np.ndarray | None = None  # (4, m)
    grain_kernel: np.ndarray | None = None  # (kh, kw[, 3])
    grain_mono: bool = False
    seed: int = 0
    lut3d_mode: str = "tetrahedral"
    highlight_burn: float = 0.0  # S7 off when 0 (cpu_processor.py:399); the caller applies the stock gate
    burn_scale: float = 50.0
    d_ref: float = 0.0
    stages: dict = field(default_factory=dict)


def render(image: np.ndarray, p: RenderInputs, method: str = "fft", keep_stages: bool = False) -> np.ndarray:
    """The hot loop of `CpuProcessor.process`, cpu_processor.py:363-405, as float32 (H, W, 3)
    in -> float32 (H, W, 3) display-referred out (before the `*255` cast, :407):
    S0 -> S1 -> [S2] -> S3 -> S4 -> [S5] -> [S6 + clip] -> [S7] -> S8."""
    x = np.asarray(image, dtype=F32)
    if p.matrix is not None:
        x = apply_matrix3x3(x, p.matrix)
    x = apply_2d_lut(x, p.lut_2d)
    if keep_stages:
        p.stages["exposure"] = x
    if p.halation_kernel is not None:
        x = halation(x, p.halation_kernel, method)
        if keep_stages:
            p.stages["halation"] = x
    x = multi_channel_interp(log_clip(x), p.lut_1d)
    if keep_stages:
        p.stages["density"] = x
    if p.mtf_kernel is not None:
        x = film_sharpness(x, p.mtf_kernel, method)
        if keep_stages:
            p.stages["mtf"] = x
    if p.grain_lut is not None:
        gk = p.grain_kernel if p.grain_kernel is not None else np.ones((1, 1), dtype=F32)  # gpu_processor.py:931-932
        x = apply_grain(x, p.grain_lut, gk, p.seed, p.grain_mono)
        if keep_stages:
            p.stages["grain"] = x
    if p.highlight_burn:
        x = burn(x, p.d_ref, p.highlight_burn, p.burn_scale)
        if keep_stages:
            p.stages["burn"] = x
    if p.lut3d_mode == "tetrahedral":
        x = apply_lut_tetrahedral(x, p.lut_3d, LUT3D_SCALE)
    else:
        x = apply_lut_trilinear(x, p.lut_3d, LUT3D_SCALE)
    return x


# ---------------------------------------------------------------------------------------------- free rotation (pre-path)
def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """cv.getRotationMatrix2D: [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]], a = s cos, b = s sin (documented form)."""
    ang = angle_deg * np.pi / 180.0
    a, b = np.cos(ang) * scale, np.sin(ang) * scale
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def invert_affine(m):
    """cv.invertAffineTransform (what warpAffine applies when WARP_INVERSE_MAP is not set)."""
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]], [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]])


def warp_affine_linear(img, m_dst_to_src, out_shape=None, offset=(0, 0)):
    """cv.warpAffine(img, M, dsize, flags=INTER_LINEAR), BORDER_CONSTANT 0, restated from the documented definition
    dst(x, y) = src(M^-1 (x, y, 1)) with bilinear interpolation: float32 source coordinates and a two-step float32 lerp
    (the shape of OpenCV >= 4.11's linear warp kernels).  PARITY UNPINNED: OpenCV is not installed here and its SIMD/scalar
    paths differ in FMA use, so only tolerance-level agreement is meaningful anyway."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape[:2]
    oh, ow = (H, W) if out_shape is None else out_shape
    m = np.asarray(m_dst_to_src, dtype=np.float64).astype(np.float32)
    xf = (np.arange(ow) + offset[1]).astype(np.float32)[None, :]
    yf = (np.arange(oh) + offset[0]).astype(np.float32)[:, None]
    sx = (xf * m[0, 0] + yf * m[0, 1]) + m[0, 2]
    sy = (xf * m[1, 0] + yf * m[1, 1]) + m[1, 2]
    x0f, y0f = np.floor(sx), np.floor(sy)
    ax, ay = (sx - x0f)[..., None], (sy - y0f)[..., None]
    x0 = np.clip(x0f, -2, W + 1).astype(np.int64)
    y0 = np.clip(y0f, -2, H + 1).astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1), :3]
        return np.where(ok[..., None], v, np.float32(0))

    t00, t01, t10, t11 = tap(y0, x0), tap(y0, x0 + 1), tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
    top = t00 + ax * (t01 - t00)
    bot = t10 + ax * (t11 - t10)
    return (top + ay * (bot - top)).astype(np.float32)


def rotate(img, degrees):
    """effects.rotate, effects.py:46-75: warp about the frame centre by -degrees, then the centred crop."""
    if not degrees:
        return img
    H, W = img.shape[:2]
    rot = rotation_matrix_2d((W / 2, H / 2), -degrees, 1.0)  # :50-51
    out = warp_affine_linear(img, invert_affine(rot))  # :52
    aspect = H / W
    angle = math.fabs(degrees) * math.pi / 180  # :54
    if aspect < 1:  # :56-62
        total_height, aspect, switch = H, 1 / aspect, True
    else:
        switch, total_height = False, W
    w = total_height / (aspect * math.sin(angle) + math.cos(angle))  # :64
    h = w * aspect
    if switch:
        w, h = h, w
    ch, cw = int((H - h) // 2), int((W - w) // 2)  # :68-69
    return out[ch:H - ch, cw:W - cw]


# ---------------------------------------------------------------------------------------------- LANCZOS4 up-scale (post-path)
# utils.resolution_scaling (utils.py:237-242) -> cv.resize(uint8 image, dsize, interpolation=cv.INTER_LANCZOS4): the way back
# from the `max_scale` pipeline resolution to the requested one (cpu_processor.py:128-134, 411-412).  PARITY UNPINNED: OpenCV
# is not installed here; this restates the generic C++ path of opencv/modules/imgproc/src/resize.cpp for CV_8U (8-tap separable
# filter, coefficients rounded to 11-bit fixed point, exact integer accumulation, one rounding at the end, replicated border).
def lanczos4_coeffs(x):
    """interpolateLanczos4(float x, float* coeffs): the C++ mixes float and double exactly like this."""
    x = np.float32(x)
    s45 = 0.70710678118654752440084436210485
    cs = ((1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45))
    y0 = -float(np.float32(x + np.float32(3))) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    coeffs = np.zeros(8, dtype=np.float32)
    total = np.float32(0)
    for i in range(8):
        y0_ = np.float32(np.float32(x + np.float32(3)) - np.float32(i))
        if abs(y0_) >= np.float32(1e-6):
            y = -float(y0_) * math.pi * 0.25
            coeffs[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        else:
            coeffs[i] = np.float32(1e30)
        total = np.float32(total + coeffs[i])
    inv = np.float32(np.float32(1) / total)
    return (coeffs * inv).astype(np.float32)


def lanczos4_table(ssize: int, dsize: int):
    """Per destination index: source index of tap 3 (floor of the source coordinate) and the 8 fixed-point weights."""
    scale = 1.0 / (float(dsize) / float(ssize))  # scale_x = 1. / inv_scale_x
    ofs = np.zeros(dsize, dtype=np.int32)
    coef = np.zeros((dsize, 8), dtype=np.int16)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        ofs[d] = sx
        c = lanczos4_coeffs(fx) * np.float32(2048)  # INTER_RESIZE_COEF_SCALE
        coef[d] = np.clip(np.rint(c), -32768, 32767).astype(np.int16)  # saturate_cast<short>: round half to even
    return ofs, coef


def resize_lanczos4_u8(img, out_h: int, out_w: int):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W = img.shape[:2]
    xo, xa = lanczos4_table(W, out_w)
    yo, ya = lanczos4_table(H, out_h)
    src = img.astype(np.int64)
    cols = np.clip(xo[:, None] - 3 + np.arange(8)[None, :], 0, W - 1)  # (out_w, 8), replicated border
    hor = np.einsum("hwkc,wk->hwc", src[:, cols, :], xa.astype(np.int64))  # (H, out_w, C) exact integers
    rows = np.clip(yo[:, None] - 3 + np.arange(8)[None, :], 0, H - 1)
    ver = np.einsum("hkwc,hk->hwc", hor[rows], ya.astype(np.int64))
    return np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)  # FixedPtCast<int, uchar, 22>


def resolution_scaling_u8_up(img, resolution):
    """utils.resolution_scaling's up-scaling branch (utils.py:226-244) on the rendered uint8 frame."""
    h, w = img.shape[:2]
    f = min(resolution[0] / h, resolution[1] / w)
    if f > 1:
        return resize_lanczos4_u8(img, round(h * f), round(w * f))
    return img
