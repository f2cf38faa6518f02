"""Product-side host stencil builders vs vectors made by the reference's own functions."""

import os

import numpy as np
import pytest

from raw2film_amd import stencils


@pytest.fixture(scope="module")
def hal(golden_dir):
    return np.load(os.path.join(golden_dir, "halation_kernels.npz"))


@pytest.fixture(scope="module")
def mtf(golden_dir):
    return np.load(os.path.join(golden_dir, "mtf_kernels.npz"))


def test_halation_psf_bit_exact(hal):
    for i, s in enumerate(hal["sizes"]):
        np.testing.assert_array_equal(stencils.halation_psf(float(s)), hal[f"blur_{i}"])


def test_halation_stencil_bit_exact(hal):
    for i, (scale, size, green, intensity, bw) in enumerate(hal["variants"]):
        k = stencils.halation_stencil(
            float(scale), halation_size=float(size), halation_green_factor=float(green),
            halation_intensity=float(intensity), bw=bool(bw),
        )
        assert k.dtype == np.float32
        np.testing.assert_array_equal(k, hal[f"halk_{i}"])


def test_halation_blue_channel_is_identity(hal):
    k = stencils.halation_stencil(166.67, halation_green_factor=0.3)
    mid = k.shape[0] // 2
    blue = k[..., 2].copy()
    assert blue[mid, mid] == 1.0
    blue[mid, mid] = 0
    assert not blue.any()


def test_mtf_stencil_bit_exact(mtf):
    table = [(mtf["logf"], v) for v in mtf["vals"]]
    for i, sc in enumerate(mtf["scales"]):
        sc = float(sc)
        np.testing.assert_array_equal(stencils.mtf_stencil_from_table(table, sc), mtf[f"kernel_s0_{i}"])
        np.testing.assert_array_equal(stencils.mtf_stencil_from_table(table, sc, 0.5, 1.0), mtf[f"kernel_s05_{i}"])
        np.testing.assert_array_equal(stencils.mtf_stencil_from_table(table, sc, 1.25, 0.6), mtf[f"kernel_s125_sig06_{i}"])


def test_mtf_stencil_is_cached_per_stock():
    from raw2film_amd.filmstock import builtin_stocks

    stock = builtin_stocks()["Kodak Portra 400"]
    a = stencils.mtf_stencil(stock, 166.67, 0.0, 1.0)
    b = stencils.mtf_stencil(stock, 166.67, 0.0, 1.0)
    assert a is b and a.shape == (17, 17, 3)
