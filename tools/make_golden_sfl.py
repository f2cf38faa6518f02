#!/usr/bin/env python3
"""Pin the sfl half of the oracle: run spectral_film_lut's OWN functions on this repo's fixture inputs and write
tests/golden/sfl.npz (arrays only) plus one BundleStock bundle per stock.

CANNOT RUN IN THE BUILD CONTAINER: `spectral_film_lut` (raw2film's dependency, pyproject.toml:28, `>=0.8.0`, unpinned) is neither
installed nor vendored there and there is no network.  On any machine that has raw2film's environment:

    pip install spectral-film-lut            # or: the environment raw2film itself runs in
    python3 tools/make_golden_sfl.py [--negative "Kodak Portra 400"] [--print "Kodak 2383"] [--loader module:callable]
    python3 -m pytest tests/test_oracle_golden.py -k sfl      # the tests that were skipping now pin oracle/stages.py S1/S3/S4/S6

What is executed -- exactly the calls raw2film makes on the hot path ("sfl" row):
    negative.get_input_lut(exp_kelvin, tint, exp_comp)                cpu_processor.py:160
    spectral_film_lut.xy_lut.apply_2d_lut(image, lut)                 cpu_processor.py:364       -> S1
    spectral_film_lut.utils.log_clip(image)            (in place)     cpu_processor.py:378       -> S3
    negative.get_density_curve(push_pull=, color_masking=)            cpu_processor.py:182
    spectral_film_lut.utils.multi_channel_interp(image, lut_1d)       cpu_processor.py:380       -> S4
    negative.get_grain_curve(scale, adx=False, bw_grain=)             gpu_processor.py:913
    spectral_film_lut.grain_generation.grain_kernel(1 / scale, grain_size_mm=, grain_sigma=)  gpu_processor.py:927
    negative.grain_transform(density, scale, adx=False, bw_grain=)    effects.py:233             -> S6c (the factor the grain is multiplied by)
    spectral_film_lut.utils.create_lut(negative, print, mode="print", ..., linear_scaling=4.0)   cpu_processor.py:232-253
(generate_grain, effects.py:231, draws from NumPy's global RNG and has no counterpart here: the build's grain field is the GPU
processor's hash noise, noise.wgsl + grain.wgsl, pinned bit-exactly by tests/test_gpu_parity.py::test_pcg3d_hash_bit_exact.)

The one sfl API raw2film never calls itself is the construction of the `filmstocks` dict (its GUI receives it from
spectral_film_lut.film_loader.load_ui, __main__.py:27-31).  This script looks for it in the places sfl is known to keep it and
otherwise takes `--loader module:callable` -- any callable returning {name: FilmSpectral}.

Nothing of sfl is copied: the .npz files hold inputs, tables and outputs as arrays (plus short str labels).
"""

from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT_DIR = os.path.join(ROOT, "tests", "golden")

# GUI defaults the fixtures are made for (gui.py:486-531): exp_kelvin 6000, color_masking 1.0, grain 6 um / sigma 0.4
EXP_KELVIN, TINT, EXP_COMP, PUSH_PULL, COLOR_MASKING = 6000, 0.0, 0.0, 0.0, 1.0
GRAIN_SIZE_MM, GRAIN_SIGMA = 0.006, 0.4
SCALES = (166.67, 341.33)  # px/mm of BASELINE configs 2/5 and 4


def find_stocks(loader: str | None) -> dict:
    if loader:
        mod, _, fn = loader.partition(":")
        return dict(getattr(importlib.import_module(mod), fn)())
    tried = []
    for mod, names in (("spectral_film_lut.film_loader", ("load_filmstocks", "load_stocks", "get_filmstocks", "FILMSTOCKS")),
                       ("spectral_film_lut", ("FILMSTOCKS", "filmstocks", "load_filmstocks")),
                       ("spectral_film_lut.filmstocks", ("FILMSTOCKS", "filmstocks", "load_filmstocks"))):
        try:
            m = importlib.import_module(mod)
        except ImportError as e:
            tried.append(f"{mod}: {e}")
            continue
        for n in names:
            obj = getattr(m, n, None)
            if obj is None:
                tried.append(f"{mod}.{n}: absent")
                continue
            stocks = obj() if callable(obj) else obj
            if isinstance(stocks, dict) and stocks:
                return stocks
            tried.append(f"{mod}.{n}: not a non-empty dict")
    raise SystemExit("could not find sfl's filmstocks dict; pass --loader module:callable returning {name: FilmSpectral}.  Tried:\n  "
                     + "\n  ".join(tried))


def pick(stocks: dict, want: str):
    if want in stocks:
        return want, stocks[want]
    hits = [k for k in stocks if want.lower() in str(k).lower()]
    if len(hits) == 1:
        return hits[0], stocks[hits[0]]
    raise SystemExit(f"stock {want!r}: {'ambiguous: ' + ', '.join(map(str, hits)) if hits else 'not found'}; available: "
                     + ", ".join(sorted(map(str, stocks)))[:2000])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--negative", default="Kodak Portra 400")  # gui.py:487
    ap.add_argument("--print", dest="print_film", default="Kodak 2383")
    ap.add_argument("--loader", default=None, help="module:callable returning sfl's {name: FilmSpectral} dict")
    args = ap.parse_args()

    try:
        from spectral_film_lut.grain_generation import grain_kernel
        from spectral_film_lut.utils import create_lut, log_clip, multi_channel_interp
        from spectral_film_lut.xy_lut import apply_2d_lut
    except ImportError as e:
        raise SystemExit(f"spectral_film_lut is not importable here ({e}): run this where raw2film itself runs")
    import spectral_film_lut

    from raw2film_amd import filmstock
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.synthetic import synthetic_frame

    stocks = find_stocks(args.loader)
    neg_name, neg = pick(stocks, args.negative)
    prt_name, prt = pick(stocks, args.print_film)

    # the repo's own fixture frame (seeded; tests/helpers.py), as XYZ like raw_to_linear hands it over, plus the S1 corner cases the
    # parity tests plant: an all-zero pixel (S < 1e-12) and a near-black one
    rgb = synthetic_frame(96, 128, seed=5)
    xyz = np.einsum("ij,hwj->hwi", np.asarray(REC709_TO_XYZ, np.float32), rgb).astype(np.float32)
    xyz[0, :4] = 0.0
    xyz[1, :4] = [1e-9, 0.0, 0.0]

    out = {"in_xyz": xyz, "negative_name": np.str_(neg_name), "print_name": np.str_(prt_name),
           "sfl_version": np.str_(getattr(spectral_film_lut, "__version__", "unknown")),
           "settings": np.asarray([EXP_KELVIN, TINT, EXP_COMP, PUSH_PULL, COLOR_MASKING, GRAIN_SIZE_MM, GRAIN_SIGMA], np.float64),
           "scales": np.asarray(SCALES, np.float64)}
    lut_2d = np.asarray(neg.get_input_lut(EXP_KELVIN, TINT, EXP_COMP))
    out["lut_2d"] = lut_2d
    s1 = np.asarray(apply_2d_lut(xyz.copy(), lut_2d))
    out["s1_exposure"] = s1
    s3 = s1.copy()
    log_clip(s3)  # in place; the return value is ignored upstream too (cpu_processor.py:378)
    out["s3_log"] = s3
    lut_1d = np.asarray(neg.get_density_curve(push_pull=PUSH_PULL, color_masking=COLOR_MASKING))
    out["lut_1d"] = lut_1d
    s4 = np.asarray(multi_channel_interp(s3.copy(), lut_1d))
    out["s4_density"] = s4
    for i, scale in enumerate(SCALES):
        for bw in (False, True):
            tag = f"{i}_{'bw' if bw else 'rgb'}"
            if getattr(neg, "rms_density", None) is not None:
                out[f"grain_lut_{tag}"] = np.asarray(neg.get_grain_curve(scale, adx=False, bw_grain=bw))
                out[f"grain_factor_{tag}"] = np.asarray(neg.grain_transform(s4.copy(), scale, adx=False, bw_grain=bw))
        k = grain_kernel(1 / scale, grain_size_mm=GRAIN_SIZE_MM, grain_sigma=GRAIN_SIGMA)
        out[f"grain_kernel_{i}"] = np.ones((1, 1), np.float32) if k is None else np.asarray(k)  # gpu_processor.py:931-932
    lut_3d = np.asarray(create_lut(
        neg, prt, mode="print", input_colorspace=None, adx_coding=False, cube=False, red_light=0.0, green_light=0.0, blue_light=0.0,
        projector_kelvin=6500, shadow_comp=0.0, sat_adjust=1.0, gamma_func="sRGB", inversion_gamma=4.0, idealized_curve=False,
        inversion=False, white_balance=False, white_clip=False, linear_scaling=4.0, color_masking=COLOR_MASKING))
    out["lut_3d"] = lut_3d
    if getattr(neg, "mtf", None) is not None:
        layers = list(neg.mtf)
        out["mtf_logf"] = np.stack([np.asarray(a, np.float64) for a, _ in layers])
        out["mtf_vals"] = np.stack([np.asarray(b, np.float64) for _, b in layers])
    out["density_measure"] = np.str_(str(neg.density_measure))
    out["d_ref"] = np.asarray(getattr(neg, "d_ref", (1.0, 1.0, 1.0)), np.float64)
    if getattr(neg, "rms_density", None) is not None:
        out["rms_density"] = np.asarray(neg.rms_density, np.float64)

    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, "sfl.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in out.items()))

    # a BundleStock the product can render with (HipProcessor takes it wherever it takes a stock): the real tables of this pair
    bundle = {k: out[k] for k in ("lut_2d", "lut_1d", "lut_3d", "density_measure", "d_ref") if k in out}
    if "grain_lut_1_rgb" in out:
        bundle["grain_lut"] = out["grain_lut_1_rgb"]
        bundle["grain_kernel"] = out["grain_kernel_1"]
        bundle["rms_density"] = out["rms_density"]
    if "mtf_logf" in out:
        bundle["mtf_logf"], bundle["mtf_vals"] = out["mtf_logf"], out["mtf_vals"]
    bpath = os.path.join(OUT_DIR, "sfl_bundle.npz")
    filmstock.save_bundle(bpath, **bundle)
    print(f"wrote {bpath} (load with raw2film_amd.filmstock.load_bundle)")


if __name__ == "__main__":
    main()
