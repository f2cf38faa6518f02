"""Oracle restatements of the caller-side steps after the path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Status: `blit_transform` inputs/outputs are PINNED by the reference's own
`_bind_copy_to_dst` (tests/golden/blit_transform.npz, tools/make_golden_blit.py); the three image operations below are
PARITY UNPINNED -- OpenCV and wgpu are not installed here, they restate cv::resize's INTER_AREA code path for CV_8U
(imgproc/src/resize.cpp) and the in-tree WGSL shaders copy_to_int.wgsl / histogram.wgsl / scale_texture.wgsl.
"""

from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _area_tab(ssize: int, dsize: int):
    """cv::computeResizeAreaTab: per destination index the list of (source index, float32 weight)."""
    scale = ssize / dsize
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = math.ceil(f1), math.floor(f2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        ent = []
        if s1 - f1 > 1e-3:
            ent.append((s1 - 1, F32((s1 - f1) / cell)))
        for s in range(s1, s2):
            ent.append((s, F32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            ent.append((s2, F32(min(min(f2 - s2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def resize_area_u8(image: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv.resize(uint8 (H, W, 3), (out_w, out_h), interpolation=cv.INTER_AREA), shrinking: utils.resolution_scaling
    (utils.py:226-236) as cpu_processor.py:411-412 applies it to the finished frame.  Integer factors: resizeAreaFast_ (integer
    block sums; 2 x 2 -> (s + 2) >> 2, otherwise saturate_cast<uchar>(sum * (1.f / area))); other factors: resizeArea_ (float32
    weights, x first into a row buffer, rows combined with beta, multiplications and additions rounded separately,
    saturate_cast<uchar> = round half to even)."""
    image = np.asarray(image, dtype=np.uint8)
    H, W = image.shape[:2]
    sx, sy = W / out_w, H / out_h
    if float(int(sx)) == sx and float(int(sy)) == sy:
        isx, isy = int(sx), int(sy)
        s = image[: out_h * isy, : out_w * isx].astype(np.int64).reshape(out_h, isy, out_w, isx, 3).sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            return ((s + 2) >> 2).astype(np.uint8)
        v = s.astype(F32) * (F32(1.0) / F32(isx * isy))
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    xtab, ytab = _area_tab(W, out_w), _area_tab(H, out_h)
    src = image.astype(F32)
    out = np.empty((out_h, out_w, 3), np.uint8)
    # row buffers: buf[sy][dx] = sum_k S[sy][sx_k] * alpha_k, k ascending from 0
    buf = np.zeros((H, out_w, 3), F32)
    for dx, ent in enumerate(xtab):
        acc = np.zeros((H, 3), F32)
        for s, a in ent:
            acc = (acc + (src[:, s, :] * a).astype(F32)).astype(F32)
        buf[:, dx, :] = acc
    for dy, ent in enumerate(ytab):
        total = None
        for s, b in ent:
            term = (b * buf[s]).astype(F32)
            total = term if total is None else (total + term).astype(F32)
        out[dy] = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    return out


def blit_rgba8(image_f32: np.ndarray, dst_h: int, dst_w: int, t: dict) -> np.ndarray:
    """shaders/copy_to_int.wgsl:19-51 with a linear clamp-to-edge sampler (gpu_processor.py:241-246): float32 throughout."""
    img = np.asarray(image_f32, dtype=F32)
    H, W = img.shape[:2]
    ys, xs = np.meshgrid(np.arange(dst_h, dtype=F32) + F32(0.5), np.arange(dst_w, dtype=F32) + F32(0.5), indexing="ij")
    u = ((xs - F32(t["offset_x"])) * F32(t["scale_x"])).astype(F32)
    v = ((ys - F32(t["offset_y"])) * F32(t["scale_y"])).astype(F32)
    inside = (u >= 0) & (u <= 1) & (v >= 0) & (v <= 1)
    fx = (u * F32(W) - F32(0.5)).astype(F32)
    fy = (v * F32(H) - F32(0.5)).astype(F32)
    x0f, y0f = np.floor(fx), np.floor(fy)
    tx, ty = (fx - x0f).astype(F32)[..., None], (fy - y0f).astype(F32)[..., None]
    x0 = np.clip(x0f.astype(np.int64), 0, W - 1)
    x1 = np.clip(x0f.astype(np.int64) + 1, 0, W - 1)
    y0 = np.clip(y0f.astype(np.int64), 0, H - 1)
    y1 = np.clip(y0f.astype(np.int64) + 1, 0, H - 1)
    with np.errstate(invalid="ignore"):
        top = img[y0, x0] + tx * (img[y0, x1] - img[y0, x0])
        bot = img[y1, x0] + tx * (img[y1, x1] - img[y1, x0])
        rgb = (top + ty * (bot - top)).astype(F32)
    out = np.zeros((dst_h, dst_w, 4), np.uint8)
    q = np.rint(np.clip(rgb, 0, 1) * F32(255)).astype(np.uint8)
    out[inside, :3] = q[inside]
    out[inside, 3] = 255
    canvas = ~inside & (xs >= F32(t["canvas_min_x"])) & (xs <= F32(t["canvas_max_x"])) & (ys >= F32(t["canvas_min_y"])) & (
        ys <= F32(t["canvas_max_y"]))
    cc = np.rint(np.clip(np.asarray(t["canvas_color"], F32), 0, 1) * F32(255)).astype(np.uint8)
    out[canvas, :3] = cc
    out[canvas, 3] = 255
    return out


def histogram_render(counts: np.ndarray, mix_table: np.ndarray, height: int, target_hw=None):
    """histogram.wgsl pass2_process (:62-128) + pass3_render (:130-160) and scale_texture.wgsl, float32 like the shaders.
    counts: (3, 256) integers.  Returns (bar image (height, 256, 4) uint8, target image or None, heights (3, 256))."""
    c = np.asarray(counts).astype(F32)
    m = F32(c.max())
    if not m > 0:
        m = F32(1.0)
    lg = np.log(F32(1.0) + c / m).astype(F32)
    left = np.concatenate([lg[:, :1], lg[:, :-1]], axis=1)
    right = np.concatenate([lg[:, 1:], lg[:, -1:]], axis=1)
    sm = (((left + lg).astype(F32) + right).astype(F32) / F32(3.0)).astype(F32)
    fm = F32(sm.max())
    if fm == 0:
        fm = F32(1.0)
    heights = ((sm * F32(height)).astype(F32) / fm).astype(F32).astype(np.uint32)
    y = np.arange(height, dtype=np.int64)[:, None]
    lim = height - heights.astype(np.int64)  # (3, 256)
    is_r, is_g, is_b = (y >= lim[0][None, :]), (y >= lim[1][None, :]), (y >= lim[2][None, :])
    mix = np.asarray(mix_table, dtype=np.uint8).reshape(2, 2, 2, 4)
    image = mix[is_r.astype(int), is_g.astype(int), is_b.astype(int)]
    target = None
    if target_hw is not None:
        th, tw = target_hw
        sx = ((np.arange(tw, dtype=F32) / F32(tw)) * F32(256.0)).astype(np.int64)
        sy = ((np.arange(th, dtype=F32) / F32(th)) * F32(height)).astype(np.int64)
        target = image[np.minimum(sy, height - 1)[:, None], np.minimum(sx, 255)[None, :]]
    return image, target, heights


def lanczos4_table_f32(ssize: int, dsize: int):
    """cv::resize's INTER_LANCZOS4 tables for CV_32F: source index of tap 3 and the eight float32 weights of
    interpolateLanczos4 (oracle.stages.lanczos4_coeffs), no fixed-point conversion."""
    from . import stages as st

    scale = 1.0 / (float(dsize) / float(ssize))
    ofs = np.zeros(dsize, dtype=np.int32)
    coef = np.zeros((dsize, 8), dtype=F32)
    for d in range(dsize):
        fx = F32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        ofs[d] = sx
        coef[d] = st.lanczos4_coeffs(F32(fx - F32(sx)))
    return ofs, coef


def resize_lanczos4_f32(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv.resize(float32 (H, W, 3), (out_w, out_h), interpolation=cv.INTER_LANCZOS4): utils.resolution_scaling's up-scaling
    branch applied to the float frame before the path (utils.py:237-242, cpu_processor.py:134).  PARITY UNPINNED (no OpenCV
    here): the generic C++ path -- HResizeLanczos4 sums the 8 products of a row left to right, VResizeLanczos4 the 8 rows top
    to bottom, float32 with separate roundings, replicated border."""
    img = np.asarray(img, dtype=F32)
    H, W = img.shape[:2]
    xo, xa = lanczos4_table_f32(W, out_w)
    yo, ya = lanczos4_table_f32(H, out_h)
    cols = np.clip(xo[:, None] - 3 + np.arange(8)[None, :], 0, W - 1)  # (out_w, 8)
    hor = None
    for j in range(8):
        term = (img[:, cols[:, j], :] * xa[None, :, j, None]).astype(F32)
        hor = term if hor is None else (hor + term).astype(F32)
    rows = np.clip(yo[:, None] - 3 + np.arange(8)[None, :], 0, H - 1)  # (out_h, 8)
    out = None
    for k in range(8):
        term = (hor[rows[:, k]] * ya[:, k, None, None]).astype(F32)
        out = term if out is None else (out + term).astype(F32)
    return out


def decode_u16(frame_u16: np.ndarray, exp_comp: float) -> np.ndarray:
    """raw_to_linear's last two lines (raw_conversion.py:50-52) on LibRaw's 16-bit output, then the upload clamp of the GPU
    path (gpu_processor.py:275): float32(u) / 65535.0, times 2 ** exp_comp as NumPy applies a Python float to a float32 array."""
    rgb = frame_u16[..., :3].astype(np.float32) / 65535.0
    rgb *= 2**exp_comp
    return np.clip(rgb, 0, 65504)


def calc_exposure(rgb: np.ndarray, ref_exposure: float = 0.18, metadata: dict | None = None) -> float:
    """color_processing.calc_exposure (color_processing.py:71-99): exposure compensation in stops from the power mean of every
    second green sample; the exponent comes from the EXIF exposure triangle when there is one (f/4 when EXIF has no aperture)."""
    lum = rgb[::2, ::2, 1]
    factor = 3
    if metadata is not None:
        n = metadata.get("EXIF:FNumber")
        if n and n != "undef":
            factor = n**2 / metadata["EXIF:ISO"] / metadata["EXIF:ExposureTime"]
        else:
            factor = 4**2 / metadata["EXIF:ISO"] / metadata["EXIF:ExposureTime"]
        factor = math.sqrt(factor) + 1
    average = (lum ** (1 / factor)).mean() ** factor
    return math.log2(ref_exposure / average)
