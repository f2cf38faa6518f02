"""Shared test helpers: build oracle inputs and HIP-side state from the same stock/settings."""

from __future__ import annotations

import numpy as np

from oracle import kernels as ok
from oracle import stages as st
from raw2film_amd import filmstock
from raw2film_amd.synthetic import synthetic_frame  # noqa: F401

SEED = 20260630


def stocks():
    s = filmstock.builtin_stocks()
    return s["Kodak Portra 400"], s["Kodak 2383"], s["Kodak Tri-X 400"]


def oracle_inputs(neg, prt, scale, *, halation=True, mtf=True, grain=2, seed=SEED, matrix=True,
                  halation_green_factor=0.3, halation_size=1.0, halation_intensity=1.0,
                  sharpening_strength=0.0, sharpening_sigma=1.0, grain_size=6.0, grain_sigma=0.4,
                  exp_kelvin=6000, color_masking=1.0):
    """RenderInputs for the oracle, built with the ORACLE's kernel builders (independent of the
    product's host code) and the stock's LUT generators (LUT contents are inputs to the path)."""
    p = st.RenderInputs(
        lut_2d=neg.get_input_lut(exp_kelvin, 0.0, 0.0),
        lut_1d=neg.get_density_curve(push_pull=0.0, color_masking=color_masking),
        lut_3d=filmstock.create_lut(neg, prt, color_masking=color_masking),
        matrix=st.REC709_TO_XYZ if matrix else None,
        seed=seed,
    )
    if halation:
        p.halation_kernel = ok.compute_halation_kernel(
            scale, halation_size=halation_size, halation_green_factor=halation_green_factor,
            halation_intensity=halation_intensity, bw=neg.density_measure == "bw")
    if mtf and neg.mtf is not None:
        p.mtf_kernel = ok.mtf_kernel(neg.mtf, scale, sharpening_strength, sharpening_sigma)
    if grain and neg.rms_density is not None:
        p.grain_lut = neg.get_grain_curve(scale, adx=False, bw_grain=grain == 1)
        p.grain_kernel = filmstock.grain_kernel(1 / scale, grain_size / 1000, grain_sigma)
        p.grain_mono = grain == 1
    return p


def rel_err(a, b, floor):
    """max |a-b| / max(|b|, floor)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def assert_close(a, b, tol=1e-5, floor=1e-3, what=""):
    e = rel_err(a, b, floor)
    assert e <= tol, f"max rel err {e:.3e} > {tol:.1e} (floor {floor}): {what}"
    return e
