"""Self-consistency of the oracle's restatements (the parts no reference vector can pin)."""

import numpy as np
import pytest

from oracle import kernels as ok
from oracle import stages as st

from helpers import oracle_inputs, stocks, synthetic_frame


def test_fft_correlation_equals_direct_mirror_correlation():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 3, (45, 61)).astype(np.float32)
    img[10, 12] = 500.0
    for k in (3, 9, 21):
        ker = rng.uniform(0, 1, (k, k))
        ker /= ker.sum()
        a = st.correlate_reflect101(img, ker, "fft")
        b = st.correlate_reflect101(img, ker, "direct")
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)


def test_correlation_is_not_convolution_and_anchor_is_centre():
    img = np.zeros((9, 9), np.float32)
    img[4, 4] = 1.0
    ker = np.arange(9, dtype=np.float64).reshape(3, 3)
    out = st.correlate_reflect101(img, ker)
    # correlation spreads the FLIPPED kernel around an impulse
    np.testing.assert_allclose(out[3:6, 3:6], ker[::-1, ::-1], atol=1e-6)


def test_reflect101_border_when_kernel_exceeds_frame():
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    ker = np.zeros((9, 9))
    ker[0, 4] = 1.0  # picks the pixel 4 rows above
    out = st.correlate_reflect101(img, ker)
    # rows ... 1 2 1 | 0 1 2 | 1 0 1 ...: the frame folds more than once under a 9-tap column
    ref = np.pad(img, ((4, 4), (0, 0)), mode="reflect")[0:3]
    np.testing.assert_allclose(out, ref, atol=1e-6)
    np.testing.assert_allclose(st.correlate_reflect101(img, ker, "direct"), ref, atol=1e-6)


def _pcg3d_scalar(x, y, z):
    m = 0xFFFFFFFF
    x, y, z = (x * 1664525 + 1013904223) & m, (y * 1664525 + 1013904223) & m, (z * 1664525 + 1013904223) & m
    x = (x + y * z) & m
    y = (y + z * x) & m
    z = (z + x * y) & m
    x ^= x >> 16
    y ^= y >> 16
    z ^= z >> 16
    x = (x + y * z) & m
    y = (y + z * x) & m
    z = (z + x * y) & m
    return x, y, z


def test_pcg3d_matches_scalar_integer_arithmetic():
    xs, ys = np.array([[0, 1, 12287, 4095]]), np.array([[0], [8191], [77]])
    vx, vy, vz = st.pcg3d(xs, ys, 20260630)
    for i, y in enumerate(ys[:, 0]):
        for j, x in enumerate(xs[0]):
            assert (int(vx[i, j]), int(vy[i, j]), int(vz[i, j])) == _pcg3d_scalar(int(x), int(y), 20260630)


def test_gaussian_field_statistics_and_mono():
    n = st.gaussian_noise(np.arange(512)[None, :], np.arange(512)[:, None], 42)
    assert n.dtype == np.float32 and n.shape == (512, 512, 3)
    assert np.all(np.abs(n.mean(axis=(0, 1))) < 0.01) and np.all(np.abs(n.std(axis=(0, 1)) - 1) < 0.01)
    m = st.gaussian_noise(np.arange(64)[None, :], np.arange(64)[:, None], 42, mono=True)
    np.testing.assert_array_equal(m[..., 0], m[..., 1])
    np.testing.assert_array_equal(m[..., 0], n[:64, :64, 0])


def test_grain_field_is_invariant_to_row_sharding():
    k = np.outer(np.hanning(7), np.hanning(7)).astype(np.float32)
    whole = st.grain_field(40, 33, 7, k)
    parts = [st.grain_field(b - a, 33, 7, k, row0=a, H_global=40) for a, b in ((0, 13), (13, 14), (14, 40))]
    np.testing.assert_array_equal(np.concatenate(parts), whole)


def test_2d_lut_with_constant_table_scales_by_sum():
    lut = np.ones((8, 8, 3), np.float32) * np.array([0.5, 1.0, 2.0], np.float32)
    img = np.random.default_rng(1).uniform(0, 2, (5, 6, 3)).astype(np.float32)
    out = st.apply_2d_lut(img, lut)
    S = img.sum(-1, keepdims=True)
    np.testing.assert_allclose(out, S * np.array([0.5, 1.0, 2.0]), rtol=1e-6)
    assert np.all(st.apply_2d_lut(np.zeros((2, 2, 3), np.float32), lut) == 0)


def test_2d_lut_is_homogeneous_in_exposure():
    neg, prt, _ = stocks()
    lut = neg.get_input_lut(6000, 0, 0)
    img = synthetic_frame(16, 16, seed=2)
    a = st.apply_2d_lut(img * np.float32(4.0), lut)
    b = st.apply_2d_lut(img, lut) * np.float32(4.0)
    np.testing.assert_allclose(a, b, rtol=2e-6)


def test_curve_interp_matches_numpy_and_clamps():
    neg, _, _ = stocks()
    curve = neg.get_density_curve(0.0, 1.0)
    x = np.linspace(-6, 3, 1001, dtype=np.float32).reshape(-1, 1, 1).repeat(3, axis=2)
    out = st.multi_channel_interp(x, curve)
    assert out[0, 0, 0] == curve[1, 0] and out[-1, 0, 2] == curve[3, -1]
    assert np.all(np.diff(out[:, 0, 1]) >= 0)


def test_tetrahedral_hits_lattice_points_and_is_continuous():
    rng = np.random.default_rng(3)
    lut = rng.uniform(0, 1, (9, 9, 9, 3)).astype(np.float32)
    idx = rng.integers(0, 9, (50, 1, 3))
    img = (idx * 0.5).astype(np.float32)  # 4 / (n-1) = 0.5 per lattice step
    out = st.apply_lut_tetrahedral(img, lut, 0.25)
    np.testing.assert_allclose(out[:, 0], lut[idx[:, 0, 0], idx[:, 0, 1], idx[:, 0, 2]], rtol=0, atol=2e-7)
    base = rng.uniform(0.3, 3.5, (200, 1, 3)).astype(np.float32)
    eps = np.float32(1e-4)
    d = np.abs(st.apply_lut_tetrahedral(base + eps, lut, 0.25) - st.apply_lut_tetrahedral(base, lut, 0.25))
    assert d.max() < 1e-3  # no jumps across tetrahedron / cell boundaries


def test_uint8_is_truncation():
    x = np.array([[[0.0, 0.999 / 255, 1.0 / 255]], [[0.5, 254.999 / 255, 1.0]]], np.float32)
    np.testing.assert_array_equal(st.to_uint8(x), np.array([[[0, 0, 1]], [[127, 254, 255]]], np.uint8))


def test_stage_gating_matches_reference_order():
    neg, prt, _ = stocks()
    img = synthetic_frame(24, 24, seed=4)
    p = oracle_inputs(neg, prt, 100.0)
    st.render(img, p, keep_stages=True)
    assert list(p.stages) == ["exposure", "halation", "density", "mtf", "grain"]
    p2 = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0)
    st.render(img, p2, keep_stages=True)
    assert list(p2.stages) == ["exposure", "density"]


@pytest.mark.parametrize("size", [(512, 512)])
def test_config1_plumbing_negative_only(size):
    """BASELINE.json config 0: 512x512 synthetic linear-RGB, negative-only LUT, CPU path."""
    neg, _, _ = stocks()
    H, W = size
    p = oracle_inputs(neg, None, max(H, W) / 36.0, halation=False, mtf=False, grain=0)
    out = st.render(synthetic_frame(H, W), p)
    assert out.shape == (H, W, 3) and out.dtype == np.float32
    assert 0.0 <= out.min() and out.max() <= 1.0 and 0.2 < out.mean() < 0.8
    assert ok.compute_halation_kernel(max(H, W) / 36.0).shape == (5, 5, 3)


def test_area_resize_is_a_block_mean_for_integer_factors_and_conserves_mass():
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 3, (120, 180)).astype(np.float32)
    np.testing.assert_allclose(st.resize_area(a, 40, 60), a.reshape(40, 3, 60, 3).mean(axis=(1, 3)), rtol=1e-6)
    for out in ((7, 11), (49, 74), (1, 1)):
        d = st.resize_area(a, *out)
        assert d.shape == out and abs(float(d.mean()) - float(a.mean())) < 1e-3 * float(a.mean()) + 1e-3
    t = st.area_table(100, 7)
    np.testing.assert_allclose(t.sum(axis=1), 1.0, atol=1e-12)  # every destination sample is a weighted mean


def test_burn_only_touches_highlights_and_uses_green():
    dens = np.full((60, 90, 3), 0.8, np.float32)
    out = st.burn(dens, d_ref=1.2, highlight_burn=0.9, burn_scale=10.0)
    np.testing.assert_array_equal(out, dens)  # nothing above d_ref -> map is zero
    dens[20:40, 30:60, 1] = 3.0  # only the green layer is dense
    out = st.burn(dens, 1.2, 0.9, 10.0)
    delta = dens - out
    assert delta.max() > 0.3 and np.allclose(delta[..., 0], delta[..., 2]) and delta.min() >= 0
    assert (out >= 0).all()
    cell, h_lo, w_lo = st.burn_geometry(60, 90, 10.0)
    assert (cell, h_lo, w_lo) == (6, 10, 15)


def test_float64_truth_evaluation_agrees_with_the_float32_oracle_on_smooth_tables():
    """oracle/truth.py (the fuzz criterion's third party: the same formulas in float64 throughout) against oracle.stages on the
    stand-in stocks: the float32 oracle is within a few float32 roundings of it, stage by stage and end to end -- which is what
    makes |oracle - truth| a measure of table roughness in tests/test_gpu_fuzz.py and nothing else."""
    from oracle import truth

    neg, prt, bw = stocks()
    H, W = 72, 104
    img = synthetic_frame(H, W, seed=3)
    for stock, kw in ((neg, {}), (neg, dict(grain=1)), (bw, {}), (neg, dict(halation=False, mtf=False, grain=0))):
        p = oracle_inputs(stock, prt, 200.0, **kw)
        ref, exact = st.render(img, p), truth.render(img, p)
        assert exact.dtype == np.float64
        assert np.max(np.abs(ref - exact) / np.maximum(np.abs(exact), 1e-3)) <= 3e-6
    p = oracle_inputs(neg, prt, 200.0)
    p.highlight_burn, p.burn_scale, p.d_ref = 0.6, 20.0, float(neg.d_ref[1])
    assert np.max(np.abs(st.render(img, p) - truth.render(img, p)) / np.maximum(np.abs(truth.render(img, p)), 1e-3)) <= 3e-6
    # chroma NR ahead of the path (on XYZ): the division by the blurred chromaticity costs the float32 oracle more
    p = oracle_inputs(neg, prt, 200.0, matrix=False)
    xyz = st.apply_matrix3x3(img, st.REC709_TO_XYZ)
    a, b = st.render(st.chroma_nr_filter(xyz, 2), p), truth.render(xyz, p, chroma_nr=2)
    assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= 3e-5
    # per stage, float32 inputs on both sides
    x = st.apply_2d_lut(xyz, p.lut_2d)
    assert np.max(np.abs(x - truth.apply_2d_lut(xyz, p.lut_2d)) / np.maximum(np.abs(x), 1e-4)) <= 2e-6
    d = st.multi_channel_interp(st.log_clip(x), p.lut_1d)
    assert np.max(np.abs(d - truth.multi_channel_interp(truth.log_clip(x), p.lut_1d))) <= 2e-6
    o = st.apply_lut_tetrahedral(d, p.lut_3d, 0.25)
    assert np.max(np.abs(o - truth.apply_lut_tetrahedral(d, p.lut_3d, 0.25))) <= 2e-7
    assert np.max(np.abs(st.apply_lut_trilinear(d, p.lut_3d) - truth.apply_lut_trilinear(d, p.lut_3d))) <= 2e-7
    g = st.apply_grain(d, p.grain_lut, p.grain_kernel, 77)
    assert np.max(np.abs(g - truth.apply_grain(d, p.grain_lut, p.grain_kernel, 77))) <= 2e-6


def test_conditioning_term_of_the_fuzz_criterion_is_what_the_tables_make_of_an_ulp_per_plane():
    """truth.conditioning: the change of the float64 output when each float32 plane between two stages moves by an ulp.  With the
    smooth stand-in tables that is ~1.2e-6 of the output per ulp (median; 3e-6 at dark outputs) -- an eighth of the contract's 1e-5 --;
    tables with steps multiply it, by a factor the function MEASURES per sample (the point of the term: tests/test_gpu_fuzz.py)."""
    import hostile
    from oracle import truth

    neg, prt, _ = stocks()
    H, W = 60, 88
    img = synthetic_frame(H, W, seed=5)
    p = oracle_inputs(neg, prt, 97.3, grain=1)
    exact = truth.render(img, p)
    cond = truth.conditioning(img, p, ulps=2.0, exact=exact)  # (the fuzz test uses ulps = 1)
    assert cond.shape == exact.shape and (cond >= 0).all()
    rel = cond / np.maximum(np.abs(exact), 1e-3)
    smooth = np.max(rel)
    # (dark outputs: an ulp of a density of 3 is 2.4e-7, the print LUT's slope there carries it into an output of 0.01 -- the
    # contract's 1e-5 of a dark output is a couple of ulps of the plane it was computed from, smooth tables or not)
    assert 0 < smooth <= 8e-6 and np.median(rel) <= 3e-6, (smooth, np.median(rel))  # i.e. ~1.2e-6 per ulp: the contract's 1e-5 is ~8 ulps
    # linear in the number of ulps, and no plane nudged -> nothing
    np.testing.assert_allclose(truth.conditioning(img, p, ulps=4.0, exact=exact), 2.0 * cond, rtol=0.2, atol=1e-9)
    assert np.array_equal(truth.render(img, p, nudge="mtf", rel=0.0), exact)
    q = oracle_inputs(neg, prt, 97.3, grain=1)
    hostile.roughen(np.random.default_rng(1750364522), q, 64, 4096, 24)  # the tables of one of the soak's four cases
    rough = truth.conditioning(img, q, ulps=2.0)
    assert np.max(rough / np.maximum(np.abs(truth.render(img, q)), 1e-3)) > 3 * smooth
