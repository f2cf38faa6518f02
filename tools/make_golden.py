#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the reference's OWN in-tree functions.

Run in the dev container only (the reference tree does not exist on the GPU box):

    python3 -B tools/make_golden.py            # -B: never write __pycache__ into /root/reference

What is executed (imported from /root/reference/src, nothing is copied):
  raw2film.utils.apply_lut_tetrahedral            utils.py:247-380
  raw2film.effects.exponential_blur_kernel        effects.py:200-217
  raw2film.effects.compute_halation_kernel        effects.py:239-263
  raw2film.effects.compute_kernel_from_function   effects.py:123-143
  raw2film.effects.mtf_kernel_layer / mtf_kernel  effects.py:159-185
  raw2film.effects.crop_image / get_canvas_data   effects.py:77-111, 290-333
  raw2film.effects.chroma_nr_filter (+ helpers)   effects.py:421-561

Third-party modules that are absent from this image are replaced by inert stubs
(SURVEY.md appendix A).  The only stub with a body is `numba.njit` (identity
decorator) / `prange` (= range); every number in the fixtures is therefore
produced by the reference's Python source run by CPython + NumPy + SciPy.

Numba-vs-CPython note for apply_lut_tetrahedral: under numba the expression
`image[y, x, 0] * scale` is float32*float64 -> float64, and `dr * (c100 - c000)`
is float64 * (float32 array) -> float64, i.e. the interpolation runs in double
and is rounded to float32 once on the store into `out`.  Under plain NumPy 2
(NEP 50) a float32 scalar times a Python float stays float32.  Passing the image
as a float64 array (holding exactly the float32 values) while keeping the LUT
float32 reproduces numba's promotion rules step by step, so that variant
("numba_semantic") is the one the oracle is pinned to bit-for-bit; the
float32-image variant ("nep50_semantic") is stored too and must agree to 1e-6.
One residue remains: for a channel at/above the LUT's upper edge the reference assigns
the Python literal `dr = 1.0` (utils.py:275), which numba types as float64 but NEP 50
treats as a weak scalar, so on those pixels this fixture carries one extra float32
rounding (<= 1 ulp); tests/test_oracle_golden.py compares them to 1 ulp, all others bit-exact.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_stubs():
    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    numba = types.ModuleType("numba")
    numba.njit = njit
    numba.prange = range
    sys.modules["numba"] = numba

    cv2 = types.ModuleType("cv2")
    for name in ("INTER_AREA", "INTER_LANCZOS4", "INTER_LINEAR"):
        setattr(cv2, name, 0)
    for name in ("filter2D", "resize", "getRotationMatrix2D", "warpAffine"):
        setattr(cv2, name, None)
    sys.modules["cv2"] = cv2

    lensfunpy = types.ModuleType("lensfunpy")
    lensfunpy_util = types.ModuleType("lensfunpy.util")
    lensfunpy.util = lensfunpy_util
    sys.modules["lensfunpy"] = lensfunpy
    sys.modules["lensfunpy.util"] = lensfunpy_util
    sys.modules["exiftool"] = types.ModuleType("exiftool")
    sys.modules["rawpy"] = types.ModuleType("rawpy")

    colour = types.ModuleType("colour")
    colour.convert = lambda v, src, dst: np.array([0.5, 0.5, 0.5])
    sys.modules["colour"] = colour

    sfl = types.ModuleType("spectral_film_lut")
    cfg = types.ModuleType("spectral_film_lut.config")
    cfg.DEFAULT_DTYPE = np.float32
    fs = types.ModuleType("spectral_film_lut.film_spectral")
    fs.FilmSpectral = type("FilmSpectral", (), {})
    gg = types.ModuleType("spectral_film_lut.grain_generation")
    gg.generate_grain = None
    gg.grain_kernel = None
    ut = types.ModuleType("spectral_film_lut.utils")
    ut.create_lut = ut.log_clip = ut.multi_channel_interp = None
    sys.modules.update(
        {
            "spectral_film_lut": sfl,
            "spectral_film_lut.config": cfg,
            "spectral_film_lut.film_spectral": fs,
            "spectral_film_lut.grain_generation": gg,
            "spectral_film_lut.utils": ut,
        }
    )


def synthetic_mtf():
    """MTF table in the shape FilmSpectral.mtf has at its call site effects.py:174:
    an iterable of (log1p(cycles/mm) grid, response) per colour layer."""
    f = np.array([0.0, 1, 2, 5, 10, 20, 30, 40, 50, 60, 80, 100, 150, 200, 400])
    logf = np.log1p(f)
    layers = []
    for f50, bump in ((55.0, 0.06), (75.0, 0.08), (45.0, 0.03)):
        vals = (1.0 + bump * np.exp(-(((f - 12.0) / 10.0) ** 2))) / (1.0 + (f / f50) ** 2) ** 0.75
        layers.append((logf.copy(), vals))
    return layers


def tetra_inputs(rng, n, hw=40):
    """Densities in [0, 4.4): random, exact grid nodes, exact ties, >= upper edge."""
    img = rng.uniform(0.0, 4.4, (hw, hw, 3)).astype(np.float32)
    step = np.float32(4.0 / (n - 1))
    # exact grid nodes on a row
    img[0, :, :] = (rng.integers(0, n, (hw, 3)).astype(np.float32)) * step
    # ties dr == dg, dg == db, all equal
    img[1, :, 1] = img[1, :, 0]
    img[2, :, 2] = img[2, :, 1]
    img[3, :, 0] = img[3, :, 2]
    img[4, :, :] = img[4, :, :1]
    # at / above the upper edge (density 4.0 maps to index n-1)
    img[5, :, 0] = 4.0
    img[6, :, :] = 4.0
    img[7, :, 1] = 4.3999
    img[8, :, :] = 0.0
    return img


def main():
    _install_stubs()
    sys.path.insert(0, REF_SRC)
    import raw2film.effects as ref_effects  # noqa: E402
    import raw2film.utils as ref_utils  # noqa: E402

    os.makedirs(OUT_DIR, exist_ok=True)
    rng = np.random.default_rng(20260630)

    # ---- 1. tetrahedral 3-D LUT (utils.py:247) --------------------------------
    tet = {}
    for n in (2, 5, 17, 33):
        lut = rng.uniform(0.0, 1.0, (n, n, n, 3)).astype(np.float32)
        img = tetra_inputs(rng, n)
        out_numba = ref_utils.apply_lut_tetrahedral(img.astype(np.float64), lut, 0.25)
        out_nep50 = ref_utils.apply_lut_tetrahedral(img, lut, 0.25)
        assert out_numba.dtype == np.float32 and out_nep50.dtype == np.float32
        tet[f"lut_{n}"] = lut
        tet[f"img_{n}"] = img
        tet[f"out_numba_semantic_{n}"] = out_numba
        tet[f"out_nep50_semantic_{n}"] = out_nep50
    np.savez_compressed(os.path.join(OUT_DIR, "tetrahedral.npz"), **tet)

    # ---- 2. halation kernels (effects.py:200, :239) ---------------------------
    hal = {}
    sizes = [1.0, 2.0, 3.0, 3.56, 4.0, 5.5, 10.5, 41.67, 57.33, 85.33]
    hal["sizes"] = np.array(sizes)
    for i, s in enumerate(sizes):
        k = ref_effects.exponential_blur_kernel(s)
        assert k.dtype == np.float64
        hal[f"blur_{i}"] = k
    variants = []
    vi = 0
    for scale in (14.22, 42.0, 166.67):
        for bw in (False, True):
            for green in (0.0, 0.3, 1.0):
                for intensity, size in ((1.0, 1.0), (0.5, 1.7)):
                    k = ref_effects.compute_halation_kernel(
                        scale,
                        halation_size=size,
                        halation_green_factor=green,
                        halation_intensity=intensity,
                        bw=bw,
                    )
                    assert k.dtype == np.float32
                    variants.append((scale, size, green, intensity, float(bw)))
                    hal[f"halk_{vi}"] = k
                    vi += 1
    for scale in (229.33, 341.33):  # cfg 3 / cfg 4 at GUI defaults (gui.py:499-504)
        k = ref_effects.compute_halation_kernel(scale, halation_size=1.0, halation_green_factor=0.3, halation_intensity=1.0)
        variants.append((scale, 1.0, 0.3, 1.0, 0.0))
        hal[f"halk_{vi}"] = k
        vi += 1
    hal["variants"] = np.array(variants)  # columns: scale,size,green,intensity,bw
    np.savez_compressed(os.path.join(OUT_DIR, "halation_kernels.npz"), **hal)

    # ---- 3. MTF kernels (effects.py:114-185) ----------------------------------
    mtf = {}
    layers = synthetic_mtf()
    mtf["logf"] = layers[0][0]
    mtf["vals"] = np.stack([v for _, v in layers])
    stock = type("Stock", (), {})()
    stock.mtf = layers
    scales = [14.22, 42.0, 166.67, 229.33, 341.33]
    mtf["scales"] = np.array(scales)
    for i, sc in enumerate(scales):
        lay = ref_effects.mtf_kernel_layer(layers[1][0], layers[1][1], sc)
        assert lay.dtype == np.float64
        mtf[f"layer_g_{i}"] = lay
        k0 = ref_effects.mtf_kernel(stock, sc, 0.0, 1.0)
        k1 = ref_effects.mtf_kernel(stock, sc, 0.5, 1.0)
        k2 = ref_effects.mtf_kernel(stock, sc, 1.25, 0.6)
        assert k0.dtype == np.float32
        mtf[f"kernel_s0_{i}"] = k0
        mtf[f"kernel_s05_{i}"] = k1
        mtf[f"kernel_s125_sig06_{i}"] = k2
    # compute_kernel_from_function with an analytic transfer function, even/odd rounding cases
    ckf = []
    for j, (size_mm, px_mm) in enumerate(((0.1, 1 / 166.67), (0.1, 1 / 60.0), (0.2, 1 / 37.4), (0.05, 1 / 341.33))):
        k = ref_effects.compute_kernel_from_function(lambda f: np.exp(-((f / 40.0) ** 2)), size_mm, px_mm)
        mtf[f"ckf_{j}"] = k
        ckf.append((size_mm, px_mm))
    mtf["ckf_args"] = np.array(ckf)
    np.savez_compressed(os.path.join(OUT_DIR, "mtf_kernels.npz"), **mtf)

    # ---- 4. host-side geometry of the rows next to the path: crop_image, get_canvas_data ----
    geo = {}
    crop_cases, crop_boxes = [], []
    for (h, w) in ((400, 600), (600, 400), (333, 517), (512, 512), (1000, 300)):
        for aspect in (1.5, 1.0, 65 / 24, 24 / 36):
            for zoom in (1, 1.3, 2.5):
                for flip in (False, True):
                    idx = np.arange(h * w, dtype=np.int64).reshape(h, w, 1)
                    out = ref_effects.crop_image(idx, zoom=zoom, aspect=aspect, flip=flip)
                    y0, x0 = divmod(int(out[0, 0, 0]), w)
                    crop_cases.append((h, w, aspect, zoom, float(flip)))
                    crop_boxes.append((y0, x0, out.shape[0], out.shape[1]))
    geo["crop_cases"] = np.array(crop_cases)
    geo["crop_boxes"] = np.array(crop_boxes)
    canvas_cases, canvas_out = [], []
    modes = ["Proportional white", "Proportional black", "Uniform white", "Uniform black", "Fixed white", "Fixed black"]
    for (h, w) in ((400, 600), (600, 400), (4000, 6000), (512, 512)):
        for mi, mode in enumerate(modes):
            for scale, ratio in ((1.0, 1.0), (1.1, 0.8), (1.25, 1.5)):
                res, color, off = ref_effects.get_canvas_data((h, w, 3), mode, scale, ratio)
                canvas_cases.append((h, w, mi, scale, ratio))
                canvas_out.append((res[0], res[1], color[0], color[1], color[2], off[0], off[1]))
    geo["canvas_modes"] = np.array(modes)
    geo["canvas_cases"] = np.array(canvas_cases)
    geo["canvas_out"] = np.array(canvas_out)
    np.savez_compressed(os.path.join(OUT_DIR, "geometry.npz"), **geo)

    # ---- 5. chroma noise reduction (effects.py:421-561): the separable masked Gaussian in xyY ----
    # CPython runs the numba loops with a float32 accumulator (`acc = 0.0` is a weak Python float, NEP 50) where numba
    # keeps float64; the fixture therefore pins the oracle to ~1e-6, not bit for bit.
    nr = {}
    sizes = [1, 2, 3, 5, 10]
    nr["sizes"] = np.array(sizes)
    for i, size in enumerate(sizes):
        taps = int(size) * 2 + 1
        sigma = 0.3 * ((taps - 1) * 0.5 - 1) + 0.8
        nr[f"kernel_{i}"] = ref_effects.gaussian_kernel_1d(taps, sigma)
    xyz = (0.18 * 2.0 ** rng.normal(0, 1.5, (28, 36, 1)) * rng.uniform(0.6, 1.4, (28, 36, 3))).astype(np.float32)
    xyz[3, 4] = 0.0  # denom <= eps branch
    xyz[5, 6] = (1e-9, 0.0, 2e-9)
    nr["xyz"] = xyz
    nr["xyY"] = ref_effects.XYZ_to_xyY(xyz)
    for i, size in enumerate(sizes[:4]):
        nr[f"out_{i}"] = ref_effects.chroma_nr_filter(xyz, size)
    np.savez_compressed(os.path.join(OUT_DIR, "chroma_nr.npz"), **nr)

    for name in ("tetrahedral", "halation_kernels", "mtf_kernels", "geometry", "chroma_nr"):
        p = os.path.join(OUT_DIR, name + ".npz")
        print(f"{p}: {os.path.getsize(p) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
