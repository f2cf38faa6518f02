"""Pin the oracle against vectors produced by the reference's own functions
(tools/make_golden.py ran them from /root/reference; only the data travels)."""

import os

import numpy as np
import pytest

from oracle import kernels as ok
from oracle import stages as st


@pytest.fixture(scope="module")
def tet(golden_dir):
    return np.load(os.path.join(golden_dir, "tetrahedral.npz"))


@pytest.fixture(scope="module")
def hal(golden_dir):
    return np.load(os.path.join(golden_dir, "halation_kernels.npz"))


@pytest.fixture(scope="module")
def mtf(golden_dir):
    return np.load(os.path.join(golden_dir, "mtf_kernels.npz"))


@pytest.mark.parametrize("n", [2, 5, 17, 33])
def test_tetrahedral_bit_exact_numba_semantic(tet, n):
    """Bit-exact wherever CPython+NumPy reproduces numba's float64 promotion.

    For a channel at/above the upper edge the reference assigns the Python literal
    `dr = 1.0` (utils.py:275): numba types it float64, but under NEP 50 the literal is a
    "weak" scalar and `1.0 * (c100 - c000)` stays float32, so the fixture (made without
    numba) carries one extra float32 rounding on those pixels.  They must agree to 1 ulp.
    """
    img = tet[f"img_{n}"]
    out = st.apply_lut_tetrahedral(img, tet[f"lut_{n}"], 0.25)
    ref = tet[f"out_numba_semantic_{n}"]
    assert out.dtype == np.float32
    edge = (np.trunc(img.astype(np.float64) * (0.25 * (n - 1))) >= n - 1).any(axis=-1)
    assert (~edge).sum() > 500
    np.testing.assert_array_equal(out[~edge], ref[~edge])
    np.testing.assert_allclose(out[edge], ref[edge], rtol=0, atol=6e-8)


@pytest.mark.parametrize("n", [2, 5, 17, 33])
def test_tetrahedral_close_to_nep50_semantic(tet, n):
    out = st.apply_lut_tetrahedral(tet[f"img_{n}"], tet[f"lut_{n}"], 0.25)
    np.testing.assert_allclose(out, tet[f"out_nep50_semantic_{n}"], rtol=0, atol=1e-6)


def test_exponential_blur_kernel_bit_exact(hal):
    for i, s in enumerate(hal["sizes"]):
        k = ok.exponential_blur_kernel(float(s))
        ref = hal[f"blur_{i}"]
        assert k.shape == ref.shape and k.dtype == np.float64
        np.testing.assert_array_equal(k, ref)
        assert abs(k.sum() - 1.0) < 1e-12


def test_halation_kernel_sizes_match_survey_table(hal):
    # scale -> halation kernel side
    for scale, side in ((14.22, 5), (166.67, 43), (229.33, 59), (341.33, 87)):
        assert ok.compute_halation_kernel(scale).shape == (side, side, 3)


def test_compute_halation_kernel_bit_exact(hal):
    for i, (scale, size, green, intensity, bw) in enumerate(hal["variants"]):
        k = ok.compute_halation_kernel(
            float(scale), halation_size=float(size), halation_green_factor=float(green),
            halation_intensity=float(intensity), bw=bool(bw),
        )
        ref = hal[f"halk_{i}"]
        assert k.dtype == np.float32 and k.shape == ref.shape
        np.testing.assert_array_equal(k, ref)


def test_mtf_kernels_bit_exact(mtf):
    layers = [(mtf["logf"], v) for v in mtf["vals"]]
    for i, sc in enumerate(mtf["scales"]):
        sc = float(sc)
        np.testing.assert_array_equal(ok.mtf_kernel_layer(layers[1][0], layers[1][1], sc), mtf[f"layer_g_{i}"])
        np.testing.assert_array_equal(ok.mtf_kernel(layers, sc, 0.0, 1.0), mtf[f"kernel_s0_{i}"])
        np.testing.assert_array_equal(ok.mtf_kernel(layers, sc, 0.5, 1.0), mtf[f"kernel_s05_{i}"])
        np.testing.assert_array_equal(ok.mtf_kernel(layers, sc, 1.25, 0.6), mtf[f"kernel_s125_sig06_{i}"])


def test_mtf_kernel_sizes_match_survey_table(mtf):
    layers = [(mtf["logf"], v) for v in mtf["vals"]]
    for scale, side in ((14.22, 1), (166.67, 17), (229.33, 23), (341.33, 35)):
        assert ok.mtf_kernel(layers, scale).shape == (side, side, 3)


def test_compute_kernel_from_function_bit_exact(mtf):
    for j, (size_mm, px_mm) in enumerate(mtf["ckf_args"]):
        k = ok.compute_kernel_from_function(lambda f: np.exp(-((f / 40.0) ** 2)), float(size_mm), float(px_mm))
        np.testing.assert_array_equal(k, mtf[f"ckf_{j}"])


@pytest.fixture(scope="module")
def nr(golden_dir):
    return np.load(os.path.join(golden_dir, "chroma_nr.npz"))


def test_chroma_nr_kernel_bit_exact(nr):
    for i, size in enumerate(nr["sizes"]):
        np.testing.assert_array_equal(st.chroma_kernel_1d(int(size)), nr[f"kernel_{i}"])


def test_chroma_nr_colour_space_round_trip_bit_exact(nr):
    np.testing.assert_array_equal(st.xyz_to_xyY(nr["xyz"]), nr["xyY"])


def test_chroma_nr_filter_matches_reference(nr):
    for i, size in enumerate(nr["sizes"][:4]):
        out = st.chroma_nr_filter(nr["xyz"], int(size))
        ref = nr[f"out_{i}"]
        assert out.dtype == np.float32
        np.testing.assert_allclose(out, ref, rtol=2e-6, atol=1e-9)
        np.testing.assert_array_equal(out[..., 1], nr["xyz"][..., 1] * (nr["xyY"][..., 1] > 1e-8))  # Y passes through
