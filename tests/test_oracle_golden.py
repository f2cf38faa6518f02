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


# ----------------------------------------------------------------------------------------------------- the sfl half (S1 / S3 / S4 / S6)
# spectral_film_lut is absent from the build container ("parity unpinned" for these stages: they restate the
# in-tree WGSL twins).  tools/make_golden_sfl.py, run once wherever raw2film's own environment exists, writes tests/golden/sfl.npz
# from sfl's own functions; from then on these tests pin oracle/stages.py to it.  Until then they skip.
@pytest.fixture(scope="module")
def sfl(golden_dir):
    path = os.environ.get("R2F_SFL_GOLDEN", os.path.join(golden_dir, "sfl.npz"))
    if not os.path.exists(path):
        pytest.skip("tests/golden/sfl.npz absent: run tools/make_golden_sfl.py on a machine that has spectral_film_lut")
    return np.load(path, allow_pickle=False)


def _rel(a, b, floor):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def test_sfl_apply_2d_lut_pins_s1(sfl):
    """cpu_processor.py:364 against oracle.stages.apply_2d_lut (restated from lut_2d.wgsl:18-108): same triangle, same texels, fp32
    arithmetic in possibly another order -- a few ulp of the exposure."""
    got = st.apply_2d_lut(sfl["in_xyz"], sfl["lut_2d"])
    assert got.shape == sfl["s1_exposure"].shape
    assert _rel(got, sfl["s1_exposure"], 1e-6) <= 2e-6


def test_sfl_log_clip_pins_s3(sfl):
    """cpu_processor.py:378: the clip bound of sfl's log_clip is what SURVEY could not see (the WGSL twin uses 1e-6)."""
    got = st.log_clip(sfl["s1_exposure"])
    assert _rel(got, sfl["s3_log"], 1e-3) <= 2e-6


def test_sfl_multi_channel_interp_pins_s4(sfl):
    """cpu_processor.py:380 against np.interp per channel on the (4, m) table."""
    got = st.multi_channel_interp(sfl["s3_log"], sfl["lut_1d"])
    assert _rel(got, sfl["s4_density"], 1e-3) <= 2e-6


@pytest.mark.parametrize("tag", ["0_rgb", "0_bw", "1_rgb", "1_bw"])
def test_sfl_grain_transform_pins_the_grain_factor_of_s6(sfl, tag):
    """effects.py:233 `stock.grain_transform(rgb, scale, adx=False, bw_grain=)` against the build's factor, interp(D; grain LUT) with
    the (4, m) table of `get_grain_curve` (grain.wgsl:78-89, gpu_processor.py:913)."""
    if f"grain_lut_{tag}" not in sfl.files:
        pytest.skip("the stock has no rms_density: no grain")
    got = st.multi_channel_interp(sfl["s4_density"], sfl[f"grain_lut_{tag}"])
    want = np.broadcast_to(sfl[f"grain_factor_{tag}"], got.shape)
    assert _rel(got, want, 1e-4) <= 1e-5


def test_sfl_bundle_renders_like_the_tables_it_was_made_from(sfl, golden_dir):
    """The BundleStock written next to sfl.npz hands the product exactly these tables; the oracle rendered with them is the
    reference's LUT-only frame (S1 + S3 + S4 + S8) to the contract."""
    from raw2film_amd import filmstock

    path = os.path.join(os.path.dirname(os.environ.get("R2F_SFL_GOLDEN", os.path.join(golden_dir, "sfl.npz"))), "sfl_bundle.npz")
    if not os.path.exists(path):
        pytest.skip("sfl_bundle.npz absent")
    b = filmstock.load_bundle(path)
    np.testing.assert_array_equal(b.get_input_lut(6000, 0.0, 0.0), sfl["lut_2d"].astype(np.float32))
    np.testing.assert_array_equal(b.get_density_curve(), sfl["lut_1d"].astype(np.float32))
    p = st.RenderInputs(lut_2d=b.get_input_lut(), lut_1d=b.get_density_curve(), lut_3d=b.output_lut(), matrix=None, seed=0)
    out = st.render(sfl["in_xyz"], p)
    ref = st.apply_lut_tetrahedral(sfl["s4_density"].astype(np.float32), sfl["lut_3d"].astype(np.float32), 0.25)
    assert _rel(out, ref, 1e-3) <= 1e-5
