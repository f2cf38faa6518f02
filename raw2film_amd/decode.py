"""The hand-off from RAW decoding to the render path.

Upstream's `raw_to_linear` (raw_conversion.py:33-53) ends with two array operations on LibRaw's 16-bit output,

    rgb = rgb.astype(float32) / 65535.0
    rgb *= 2 ** calc_exposure(rgb, metadata=...)

and everything after them is this package's path.  LibRaw itself stays on raw2film's side of the boundary; the two lines can
move to the device (`HipContext.decode_u16`, `r2f_decode_u16`), so that a decoded frame crosses PCIe as uint16 -- 6 bytes per
pixel instead of the 16 of upstream's float RGBA payload -- which is what bounds batch export.  This module holds the host
part: the auto-exposure statistic (`calc_exposure`, color_processing.py:71-99), computed from a quarter of the green samples,
and the float32 factor the frame is multiplied by.
"""

from __future__ import annotations

import math

import numpy as np

U16_DIVISOR = np.float32(65535.0)


def exposure_root(metadata: dict | None) -> float:
    """The exponent `factor` of calc_exposure (color_processing.py:77-91): 3 without metadata, otherwise
    sqrt(N^2 / ISO / t) + 1 with the aperture N defaulting to f/4 when EXIF has none (missing, falsy or "undef")."""
    if metadata is None:
        return 3
    n = metadata.get("EXIF:FNumber")
    aperture_sq = n**2 if (n and n != "undef") else 4**2
    return math.sqrt(aperture_sq / metadata["EXIF:ISO"] / metadata["EXIF:ExposureTime"]) + 1


def auto_exposure(frame: np.ndarray, ref_exposure: float = 0.18, metadata: dict | None = None) -> float:
    """calc_exposure (color_processing.py:71-99): stops of exposure compensation that bring the frame's power mean of the green
    channel (every second sample in both directions) to `ref_exposure`.  `frame`: the decoded (H, W, 3+) frame, float32 in
    [0, 1] or LibRaw's uint16 (divided by 65535 here, like raw_conversion.py:50).  float32 arithmetic throughout, as NumPy
    evaluates upstream's expression."""
    green = frame[::2, ::2, 1]
    if green.dtype == np.uint16:
        green = green.astype(np.float32) / U16_DIVISOR
    green = np.asarray(green, dtype=np.float32)
    root = exposure_root(metadata)
    # NumPy's own operator semantics, on purpose: an array power by a Python float runs the float32 ufunc, a float32 SCALAR
    # raised to a Python float takes NumPy's scalar path (which does not round like the ufunc), and a Python float divided by
    # a float32 scalar is a float32 division -- written with the operators, this evaluates exactly as upstream's expression
    # does under whatever NumPy is installed
    mean_of_roots = (green ** (1 / root)).mean()
    average = mean_of_roots**root
    return math.log2(ref_exposure / average)


def exposure_factor(exp_comp: float) -> np.float32:
    """The float32 scalar `rgb *= 2 ** exp_comp` multiplies a float32 frame by (raw_conversion.py:52)."""
    return np.float32(2**exp_comp)


def decode_u16_host(frame_u16: np.ndarray, exp_comp: float) -> np.ndarray:
    """The two lines on the host, for callers without a device buffer at hand (and the reference point of the device kernel)."""
    rgb = frame_u16[..., :3].astype(np.float32) / U16_DIVISOR
    rgb *= exposure_factor(exp_comp)
    return np.minimum(rgb, np.float32(65504.0))  # the GPU path's upload clamp, gpu_processor.py:275
