"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Tolerances (north_star: 1e-5 relative fp32 per channel; hash bit-identical):
  * stage outputs:   |hip - oracle| <= 1e-5 * max(|oracle|, floor) with the floor stated per test
    (densities and display values live in [0, 4] / [0, 1]; the floor keeps near-zero values from
    turning an absolute 1e-7 rounding difference into a meaningless relative one);
  * PCG3D hash:      bit-exact uint32;
  * uint8 output:    <= 1 LSB on <= 1e-4 of the samples (truncation at fp32 rounding boundaries).
The floor is the contract's 1e-3 for every density and display value; linear exposure (before the
log) uses 1e-4.  Measured worst case of the whole path at that floor: 2.6e-6 (tools/parity_budget.py,
profiles/r02_parity_budget.txt).
"""

import numpy as np
import pytest

from oracle import kernels as ok
from oracle import stages as st

from helpers import SEED, assert_close, oracle_inputs, rel_err, stocks, synthetic_frame

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from raw2film_amd.context import HipContext

    c = HipContext(0)
    yield c
    c.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_planes(a):
    return dev(np.ascontiguousarray(np.transpose(a, (2, 0, 1))))


def from_planes(t):
    return np.transpose(t.cpu().numpy(), (1, 2, 0))


def setup_ctx(ctx, p):
    ctx.set_matrix3x3(p.matrix)
    ctx.set_lut2d(p.lut_2d)
    ctx.set_curve1d(p.lut_1d)
    ctx.set_lut3d(p.lut_3d)
    if p.halation_kernel is not None:
        ctx.set_kernel(0, p.halation_kernel)
    if p.mtf_kernel is not None:
        ctx.set_kernel(1, p.mtf_kernel)
    if p.grain_lut is not None:
        ctx.set_grain_lut(p.grain_lut)
        ctx.set_kernel(2, p.grain_kernel if p.grain_kernel is not None else np.ones((1, 1), np.float32))
    return ctx.make_params(
        matrix=p.matrix is not None, halation=p.halation_kernel is not None, mtf=p.mtf_kernel is not None,
        grain=p.grain_lut is not None, grain_mono=p.grain_mono, seed=p.seed,
        lut3d_mode=0 if p.lut3d_mode == "tetrahedral" else 1)


# ------------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("layout", ["hwc3", "hwc4", "chw"])
@pytest.mark.parametrize("shape", [(64, 96), (37, 53)])
def test_front_stages_and_layouts(ctx, layout, shape):
    H, W = shape
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0)
    img = synthetic_frame(H, W, seed=3)
    img[0, :4] = 0.0  # S < 1e-12 branch of the 2-D LUT
    img[1, :4] = [1e-9, 0, 0]
    ref_out = st.render(img, p, keep_stages=True)
    params = setup_ctx(ctx, p)
    if layout == "hwc3":
        t = dev(img)
    elif layout == "hwc4":
        t = dev(np.concatenate([img, np.ones((H, W, 1), np.float32)], axis=-1))
    else:
        t = to_planes(img)
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_front(t, params, 0, dst=E)
    assert_close(from_planes(E), p.stages["exposure"], 1e-5, 1e-4, "exposure")
    D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_front(t, params, 1, dst=D)
    assert_close(from_planes(D), p.stages["density"], 1e-5, 1e-3, "density")
    out, u8 = ctx.render(t, params, want_f32=True, want_u8=True)
    assert_close(out.cpu().numpy(), ref_out, 1e-5, 1e-3, "output")
    ref_u8 = st.to_uint8(ref_out)
    diff = np.abs(u8.cpu().numpy().astype(int) - ref_u8.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4


def test_tetrahedral_against_reference_golden_vectors(ctx, golden_dir):
    """S8 on the GPU vs the vectors produced by the reference's own apply_lut_tetrahedral."""
    import os

    tet = np.load(os.path.join(golden_dir, "tetrahedral.npz"))
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0)
    for n in (2, 5, 17, 33):
        img, lut = tet[f"img_{n}"], tet[f"lut_{n}"]
        setup_ctx(ctx, p)
        ctx.set_lut3d(lut)
        params = ctx.make_params()
        H, W = img.shape[:2]
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        ctx.stage_tail(to_planes(img), params, out_f32=out, y0=0, y1=H, H_global=H)
        np.testing.assert_allclose(out.cpu().numpy(), tet[f"out_numba_semantic_{n}"], rtol=0, atol=4e-7)


def test_trilinear_mode(ctx):
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0)
    p.lut3d_mode = "trilinear"
    img = synthetic_frame(48, 64, seed=5)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, _ = ctx.render(dev(img), params)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "trilinear output")


# ------------------------------------------------------------------------------- stencils
@pytest.mark.parametrize("shape,ksize", [((96, 160), 5), ((150, 200), 43), ((70, 133), 87), ((20, 24), 59), ((1, 40), 5), ((33, 1), 7)])
def test_generic_stencil_reflect101(ctx, shape, ksize):
    """Plain per-channel correlation with reflect-101 borders, incl. frames smaller than the stencil
    (multiple reflections), single-row/column frames, and widths that are not multiples of 4."""
    H, W = shape
    rng = np.random.default_rng(ksize)
    k = rng.uniform(0, 1, (ksize, ksize, 3)).astype(np.float32)
    k /= k.sum(axis=(0, 1), keepdims=True)
    img = rng.uniform(0.0, 2.0, (H, W, 3)).astype(np.float32)
    ref = st.convolve_2d(img, k, method="direct")
    ctx.set_kernel(1, k)
    src, dst = to_planes(img), torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_stencil(1, src, dst, y0=0, y1=H, H_global=H)
    assert_close(from_planes(dst), ref, 1e-5, 1e-3, f"stencil {ksize}")


@pytest.mark.parametrize("ksize", [173, 301])
def test_very_wide_stencils_fall_back_to_narrow_tiles(ctx, ksize):
    """halation_size / sharpening can ask for stencils far wider than a 128-px tile row in LDS: the narrow tile
    variant and, beyond that, the one-workgroup-per-CU budget take over; results stay exact."""
    rng = np.random.default_rng(ksize)
    H, W = 70, 90
    img = rng.uniform(0.0, 2.0, (H, W, 3)).astype(np.float32)
    k = ok.exponential_blur_kernel(float(ksize - 1)).astype(np.float32)
    assert k.shape == (ksize, ksize)
    ref = st.convolve_2d(img, k)
    ctx.set_kernel(1, k)
    src, dst = to_planes(img), torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_stencil(1, src, dst, y0=0, y1=H, H_global=H)
    assert_close(from_planes(dst), ref, 1e-5, 1e-3, f"wide stencil {ksize}")


def test_stencil_non_square_and_even_sizes(ctx):
    rng = np.random.default_rng(11)
    img = rng.uniform(0.0, 1.0, (40, 72, 3)).astype(np.float32)
    k = rng.uniform(-0.5, 1, (6, 9, 1)).astype(np.float32)  # even height, anchor (3, 4), negative taps
    ctx.set_kernel(2, k)
    src, dst = to_planes(img), torch.empty((3, 40, 72), dtype=torch.float32, device="cuda")
    ctx.stage_stencil(2, src, dst, y0=0, y1=40, H_global=40)
    pad = np.pad(img, ((3, 2), (4, 4), (0, 0)), mode="reflect").astype(np.float64)
    ref = np.zeros((40, 72, 3))
    for i in range(6):
        for j in range(9):
            ref += k[i, j, 0] * pad[i:i + 40, j:j + 72]
    assert_close(from_planes(dst), ref, 1e-5, 1e-3, "non-square stencil")


@pytest.mark.parametrize("variant", [0, 1])
def test_stencil_tile_variants_agree_bitwise(ctx, variant):
    """Every tile variant accumulates taps in the same order -> identical bits."""
    rng = np.random.default_rng(7)
    img = rng.uniform(0.0, 2.0, (130, 270, 3)).astype(np.float32)
    k = ok.compute_halation_kernel(100.0, halation_green_factor=0.3)
    ctx.set_kernel(0, k)
    src = to_planes(img)
    outs = []
    for v in (0, variant):
        ctx.set_option("stencil_variant", v)
        dst = torch.empty((3, 130, 270), dtype=torch.float32, device="cuda")
        ctx.stage_stencil(0, src, dst, y0=0, y1=130, H_global=130)
        outs.append(dst.cpu().numpy())
    ctx.set_option("stencil_variant", -1)
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("lds_kb", [160, 80, 48])
def test_stencil_phasing_is_bitwise_neutral(ctx, lds_kb):
    """The LDS budget only changes how many row steps share a tile fill, never the tap order."""
    rng = np.random.default_rng(8)
    img = rng.uniform(0.0, 2.0, (140, 200, 3)).astype(np.float32)
    ctx.set_kernel(0, ok.compute_halation_kernel(229.33, halation_green_factor=0.3))
    src = to_planes(img)
    outs = []
    for kb in (80, lds_kb):
        ctx.set_option("stencil_lds_kb", kb)
        dst = torch.empty((3, 140, 200), dtype=torch.float32, device="cuda")
        ctx.stage_stencil(0, src, dst, y0=0, y1=140, H_global=140)
        outs.append(dst.cpu().numpy())
    ctx.set_option("stencil_lds_kb", 80)
    np.testing.assert_array_equal(outs[0], outs[1])


def test_mirror_symmetric_fast_path(ctx):
    """Bit-for-bit mirror-symmetric stencils take the paired-tap path; it must agree with the plain
    path to rounding, and a stencil that is symmetric only up to one ulp must not take it."""
    rng = np.random.default_rng(9)
    H, W = 100, 150
    img = rng.uniform(0.0, 2.0, (H, W, 3)).astype(np.float32)
    src = to_planes(img)
    ctx.set_option("stencil_fft", 0)  # this test is about the two entry-list forms of the direct kernel
    ctx.set_option("stencil_fixed", 0)

    def run(k, sym):
        ctx.set_option("stencil_sym", sym)
        ctx.set_kernel(1, k)
        dst = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_stencil(1, src, dst, y0=0, y1=H, H_global=H)
        ctx.set_option("stencil_sym", 1)
        return from_planes(dst)

    try:
        for k in (ok.compute_halation_kernel(341.33, halation_green_factor=0.3),   # r = 42 (even)
                  ok.compute_halation_kernel(229.33, halation_green_factor=0.3),   # r = 28
                  ok.compute_halation_kernel(60.0),                                 # r = 7 (odd -> padded)
                  ok.mtf_kernel(stocks()[0].mtf, 341.33), ok.mtf_kernel(stocks()[0].mtf, 229.33, 0.7, 1.0)):
            a, b = run(k, 1), run(k, 0)
            ref = st.convolve_2d(img, k)
            # The contract floor (1e-3) for non-negative taps.  The unsharp-masked MTF kernel has negative taps: an fp32
            # direct sum is accurate relative to sum |w x|, not to the cancelled result, hence the 1e-2 floor for the
            # DIRECT form of that kernel (forced here; the render path runs it as an fp64 FFT, tests/test_gpu_fft.py).
            floor = 1e-3 if (k >= 0).all() else 1e-2
            assert_close(a, ref, 1e-5, floor, "sym path")
            assert_close(b, ref, 1e-5, floor, "plain path")
            assert np.abs(a - b).max() <= 2e-6
            assert not np.array_equal(a, b), "the symmetric path was not taken"
        k = ok.compute_halation_kernel(229.33, halation_green_factor=0.3).copy()
        k[3, 20, 0] = np.nextafter(k[3, 20, 0], np.float32(1))  # break the symmetry of the red plane by one ulp
        np.testing.assert_array_equal(run(k, 1)[..., 0], run(k, 0)[..., 0])
    finally:
        ctx.set_option("stencil_fft", 1)
        ctx.set_option("stencil_fixed", 1)


@pytest.mark.parametrize("n", [3, 5, 9, 13, 15, 17, 21, 23, 25])
@pytest.mark.parametrize("epilogue", [0, 1])
def test_small_square_stencils_unrolled_direct_form(ctx, n, epilogue):
    """Square mirror-symmetric stencils run the direct kernel's fully unrolled form (stencil_fixed) up to the size where the FFT
    form overtakes it: 23 x 23 against complex128 scratch (the halation, epilogue = 1), 19 x 19 against the cheaper complex64
    scratch of the MTF passes (epilogue = 0); beyond that the FFT form.  Against the oracle, against the entry list,
    per-channel taps, with and without the halation epilogue, on row ranges with halo rows of any origin."""
    rng = np.random.default_rng(100 + n)
    H, W = 150, 203
    k = rng.uniform(0.0, 1.0, (n, n, 3)).astype(np.float32)  # (one sign: taps of both signs take the float64 FFT form whatever their size)
    k = (k + k[:, ::-1]) / 2
    k /= k.sum(axis=(0, 1), keepdims=True)
    img = rng.uniform(0.01, 2.0, (H, W, 3)).astype(np.float32)
    img[rng.integers(0, H, 20), rng.integers(0, W, 20)] = 16.0
    which = 0 if epilogue else 1
    if epilogue:
        ctx.set_curve1d(stocks()[0].get_density_curve(push_pull=0.0, color_masking=1.0))
    ref = st.convolve_2d(img, k)
    if epilogue:
        ref = st.multi_channel_interp(st.log_clip(ref), stocks()[0].get_density_curve(push_pull=0.0, color_masking=1.0))
    r = n // 2

    def run(fixed):
        ctx.set_option("stencil_fixed", fixed)
        ctx.set_kernel(which, k)
        params = ctx.make_params()
        out = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        for y0, y1 in ((0, 41), (41, 42), (42, H)):
            lo, hi = max(y0 - r, 0), min(y1 + r, H)
            src = to_planes(img[lo:hi])
            if epilogue:
                ctx.stage_halation(src, out, params, src_gy0=lo, dst_gy0=0, y0=y0, y1=y1, H_global=H)
            else:
                ctx.stage_mtf(src, out, params, src_gy0=lo, dst_gy0=0, y0=y0, y1=y1, H_global=H)
        whole = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        if epilogue:
            ctx.stage_halation(to_planes(img), whole, params, y0=0, y1=H, H_global=H)
        else:
            ctx.stage_mtf(to_planes(img), whole, params, y0=0, y1=H, H_global=H)
        fft = [c["fft"] for c in ctx.stencil_stats(which)]
        assert [c["unrolled"] for c in ctx.stencil_stats(which)] == [n // 2 if fixed and n <= last_unrolled else 0] * 3
        return from_planes(out), from_planes(whole), fft

    last_unrolled = 23 if epilogue else 19
    a, a_whole, fft = run(1)
    assert fft == ([1, 1, 1] if n > last_unrolled else [0, 0, 0])
    assert_close(a, ref, 1e-5, 1e-3, f"{n} x {n} unrolled")
    if n <= last_unrolled:
        np.testing.assert_array_equal(a, a_whole)  # direct forms: bit for bit whatever the row range
    b, _, fft_b = run(0)
    assert fft_b == ([1, 1, 1] if n >= 21 else [0, 0, 0])  # without the unrolled form the FFT threshold is 400 taps
    assert_close(b, ref, 1e-5, 1e-3, f"{n} x {n} entry list / FFT")
    ctx.set_option("stencil_fixed", 1)


@pytest.mark.parametrize("scale", [14.22, 166.67, 341.33])
def test_halation_stage(ctx, scale):
    neg, prt, _ = stocks()
    H, W = 96, 144
    p = oracle_inputs(neg, prt, scale, mtf=False, grain=0)
    img = synthetic_frame(H, W, seed=9)
    st.render(img, p, keep_stages=True)
    params = setup_ctx(ctx, p)
    E = to_planes(p.stages["exposure"])
    D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)
    assert_close(from_planes(D), p.stages["density"], 1e-5, 1e-3, "halation+curve")


def test_halation_bw_stock(ctx):
    _, prt, bw = stocks()
    H, W = 64, 96
    p = oracle_inputs(bw, prt, 120.0, mtf=False, grain=0)
    img = synthetic_frame(H, W, seed=10)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, _ = ctx.render(dev(img), params)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "bw halation output")


@pytest.mark.parametrize("strength", [0.0, 0.7])
def test_mtf_stage(ctx, strength):
    neg, prt, _ = stocks()
    H, W = 80, 120
    scale = 229.33
    p = oracle_inputs(neg, prt, scale, halation=False, grain=0, sharpening_strength=strength)
    img = synthetic_frame(H, W, seed=12)
    st.render(img, p, keep_stages=True)
    params = setup_ctx(ctx, p)
    D = to_planes(p.stages["density"])
    D2 = torch.empty_like(D)
    ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
    assert_close(from_planes(D2), p.stages["mtf"], 1e-5, 1e-3, "mtf")


# ------------------------------------------------------------------------------- grain
def test_pcg3d_hash_bit_exact(ctx):
    params = ctx.make_params(seed=SEED)
    H, W = 67, 301
    h, _ = ctx.stage_noise(params, 5, 5 + H, W, want_noise=False)
    ys, xs = np.arange(5, 5 + H)[:, None], np.arange(W)[None, :]
    vx, vy, vz = st.pcg3d(xs, ys, SEED)
    got = h.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got[0], vx)
    np.testing.assert_array_equal(got[1], vy)
    np.testing.assert_array_equal(got[2], vz)


@pytest.mark.parametrize("mono", [False, True])
def test_gaussian_field(ctx, mono):
    params = ctx.make_params(seed=12345, grain_mono=mono)
    H, W = 128, 256
    _, n = ctx.stage_noise(params, 0, H, W, want_hash=False)
    ref = st.gaussian_noise(np.arange(W)[None, :], np.arange(H)[:, None], 12345, mono)
    got = from_planes(n)
    assert np.max(np.abs(got - ref)) <= 1e-5  # |n| < 6; fp32 log/sin/cos differ by ~1e-6 between libms
    assert abs(got.mean()) < 0.02 and abs(got.std() - 1.0) < 0.02


@pytest.mark.parametrize("grain", [2, 1])
@pytest.mark.parametrize("grain_size", [6.0, 1.0, 14.0])
def test_tail_grain(ctx, grain, grain_size):
    neg, prt, _ = stocks()
    H, W = 100, 140
    p = oracle_inputs(neg, prt, 341.33, halation=False, mtf=False, grain=grain, grain_size=grain_size)
    img = synthetic_frame(H, W, seed=13)
    ref = st.render(img, p, keep_stages=True)
    params = setup_ctx(ctx, p)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ctx.stage_tail(to_planes(p.stages["density"]), params, out_f32=out, y0=0, y1=H, H_global=H)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "grain tail")


# ------------------------------------------------------------------------------- whole path
@pytest.mark.parametrize("shape,scale", [((160, 240), 166.67), ((131, 203), 341.33), ((256, 384), 229.33)])
def test_full_pipeline(ctx, shape, scale):
    neg, prt, _ = stocks()
    H, W = shape
    p = oracle_inputs(neg, prt, scale)
    img = synthetic_frame(H, W, seed=21)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, u8 = ctx.render(dev(img), params, want_f32=True, want_u8=True)
    e = assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "full pipeline")
    print(f"full pipeline {shape} scale {scale}: max err {e:.2e}")
    diff = np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(ref).astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4


def test_row_shards_match_whole_frame_bitwise(ctx):
    """Stage calls on row shards (with halos) reproduce the whole-frame render bit for bit --
    the property the multi-GPU row tiler relies on."""
    neg, prt, _ = stocks()
    H, W = 150, 200
    scale = 200.0
    p = oracle_inputs(neg, prt, scale)
    img = synthetic_frame(H, W, seed=22)
    params = setup_ctx(ctx, p)
    t = dev(img)
    # (like for like: the stage entry points keep complex128 scratch for the halation; a whole-frame render may choose the 12-byte
    # element from the frame's range -- tests/test_gpu_fft.py -- and then agrees to that element's rounding instead of bit for bit)
    ctx.set_option("stencil_fft_scratch96_auto", 0)
    whole, _ = ctx.render(t, params)
    ctx.set_option("stencil_fft_scratch96_auto", 1)
    rh = p.halation_kernel.shape[0] // 2
    rm = p.mtf_kernel.shape[0] // 2
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    bounds = [0, 40, 97, 150]
    for a, b in zip(bounds[:-1], bounds[1:]):
        # rows this shard needs at each level
        d_lo, d_hi = max(a - rm, 0), min(b + rm, H)
        e_lo, e_hi = max(d_lo - rh, 0), min(d_hi + rh, H)
        E = torch.empty((3, e_hi - e_lo, W), dtype=torch.float32, device="cuda")
        ctx.stage_front(t[e_lo:e_hi].contiguous(), params, 0, in_gy0=e_lo, dst=E, dst_gy0=e_lo, H_global=H)
        D = torch.empty((3, d_hi - d_lo, W), dtype=torch.float32, device="cuda")
        ctx.stage_halation(E, D, params, src_gy0=e_lo, dst_gy0=d_lo, y0=d_lo, y1=d_hi, H_global=H)
        D2 = torch.empty((3, b - a, W), dtype=torch.float32, device="cuda")
        ctx.stage_mtf(D, D2, params, src_gy0=d_lo, dst_gy0=a, y0=a, y1=b, H_global=H)
        ctx.stage_tail(D2, params, src_gy0=a, out_f32=out, out_gy0=0, y0=a, y1=b, H_global=H)
    np.testing.assert_array_equal(out.cpu().numpy(), whole.cpu().numpy())


# ------------------------------------------------------------------------------- errors
def test_errors_are_reported(ctx):
    from raw2film_amd.context import HipContext

    c = HipContext(0)
    img = dev(synthetic_frame(8, 8))
    with pytest.raises(ValueError, match="input LUT not set"):
        c.render(img, c.make_params())
    with pytest.raises(ValueError):
        c.set_lut2d(np.zeros((4, 5, 3), np.float32))
    with pytest.raises(ValueError):
        c.set_curve1d(np.zeros((3, 8), np.float32))
    c.close()


def test_stencil_stages_refuse_overlapping_planes(ctx):
    """The stencil stages are out of place (include/r2f.h): a destination that shares bytes with the source is an error, not a
    race; the pointwise grain stage may run exactly in place."""
    neg, prt, _ = stocks()
    H, W = 64, 96
    p = oracle_inputs(neg, prt, 200.0)
    params = setup_ctx(ctx, p)
    buf = torch.rand((4, H, W), dtype=torch.float32, device="cuda") + 0.1
    D = buf[:3]
    for call in (lambda: ctx.stage_mtf(D, D, params, y0=0, y1=H, H_global=H),
                 lambda: ctx.stage_halation(D, D, params, y0=0, y1=H, H_global=H),
                 lambda: ctx.stage_stencil(1, D, buf[1:4], y0=0, y1=H, H_global=H)):  # shifted by one plane: still overlaps
        with pytest.raises(ValueError, match="overlap"):
            call()
    before = D.clone()
    separate = torch.empty_like(before)
    ctx.stage_grain(before, separate, params, y0=0, y1=H, H_global=H)
    ctx.stage_grain(D, D, params, y0=0, y1=H, H_global=H)  # exactly in place: allowed, same result
    assert torch.equal(D, separate)


def test_two_contexts_in_one_process_leave_the_current_device_alone():
    """Every C-ABI entry binds its context's device for the call and restores the caller's: contexts can be
    interleaved from one thread.  On a one-GPU box both sit on device 0; with more GPUs the second one gets device 1 and the
    current device is switched between the calls."""
    from raw2film_amd.context import HipContext

    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 200.0)
    img = synthetic_frame(72, 100, seed=77)
    ref = st.render(img, p)
    n = torch.cuda.device_count()
    devs = (0, 1 if n > 1 else 0)
    ctxs = [HipContext(d) for d in devs]
    try:
        assert torch.cuda.current_device() == 0  # r2f_create did not move it
        prm = [setup_ctx(c, p) for c in ctxs]
        outs = []
        for k, (c, q, d) in enumerate(zip(ctxs, prm, devs)):
            torch.cuda.set_device(devs[1 - k])  # the OTHER context's device is current while this one renders
            with torch.cuda.device(d):
                t = torch.from_numpy(img).to(f"cuda:{d}")
            o, _ = c.render(t, q)
            assert torch.cuda.current_device() == devs[1 - k]
            outs.append(o.cpu().numpy())
        torch.cuda.set_device(0)
        for o in outs:
            assert_close(o, ref, 1e-5, 1e-3, "two contexts")
        np.testing.assert_array_equal(outs[0], outs[1])
    finally:
        torch.cuda.set_device(0)
        for c in ctxs:
            c.close()


# ------------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 7), (9, 129), (130, 5)])
def test_tiny_and_ragged_frames_full_pipeline(ctx, shape):
    """Frames smaller than any tile, than every stencil, and with widths that defeat the float4 paths."""
    neg, prt, _ = stocks()
    H, W = shape
    p = oracle_inputs(neg, prt, 120.0)  # 31-tap halation, 13-tap MTF, grain on
    img = synthetic_frame(H, W, seed=50 + H)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, u8 = ctx.render(dev(img), params, want_f32=True, want_u8=True)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, f"tiny frame {shape}")
    assert np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(ref).astype(int)).max() <= 1


def test_black_frame_and_speculars(ctx):
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 200.0)
    params = setup_ctx(ctx, p)
    black = np.zeros((40, 56, 3), np.float32)  # S < 1e-12 everywhere: exposure 0, log clipped at 1e-6
    out, _ = ctx.render(dev(black), params)
    assert_close(out.cpu().numpy(), st.render(black, p), 1e-5, 1e-3, "black frame")
    hot = synthetic_frame(64, 96, seed=61)
    hot[10, 20] = 65504.0  # the largest value the decode path lets through (gpu_processor.py:275)
    hot[40:43, 60:63] = 4000.0
    out, _ = ctx.render(dev(hot), params)
    assert_close(out.cpu().numpy(), st.render(hot, p), 1e-5, 1e-3, "speculars")


def test_empty_row_range_is_a_no_op_and_empty_frame_is_an_error(ctx):
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0)
    params = setup_ctx(ctx, p)
    E = torch.zeros((3, 8, 16), dtype=torch.float32, device="cuda")
    D = torch.full((3, 8, 16), -1.0, dtype=torch.float32, device="cuda")
    ctx.stage_halation(E, D, params, y0=3, y1=3, H_global=8)
    assert float(D.min()) == -1.0 and float(D.max()) == -1.0
    with pytest.raises(ValueError):
        ctx.stage_halation(E, D, params, y0=0, y1=9, H_global=8)  # rows outside the frame
    small = torch.zeros((3, 4, 16), dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError, match="not inside"):
        ctx.stage_mtf(small, D, params, src_gy0=0, dst_gy0=0, y0=0, y1=8, H_global=8)  # source rows missing


def test_uint8_only_output(ctx):
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 150.0)
    img = synthetic_frame(72, 100, seed=62)
    params = setup_ctx(ctx, p)
    f32, u8 = ctx.render(dev(img), params, want_f32=False, want_u8=True)
    assert f32 is None and u8.dtype == torch.uint8
    d = np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(st.render(img, p)).astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-4


# ------------------------------------------------------------------------------- S7 highlight burn
@pytest.mark.parametrize("shape,burn_scale", [((150, 230), 50.0), ((96, 96), 8.0), ((301, 177), 25.0)])
def test_burn_area_sums_and_map(ctx, shape, burn_scale):
    H, W = shape
    rng = np.random.default_rng(H)
    dens = rng.uniform(0.0, 3.0, (H, W, 3)).astype(np.float32)
    cell, h_lo, w_lo = st.burn_geometry(H, W, burn_scale)
    params = ctx.make_params(burn_strength=0.5, burn_cell=cell, burn_d_ref=1.2)
    D = to_planes(dens)
    sums = ctx.stage_burn_sums(D, params, y0=0, y1=H, H_global=H)
    assert tuple(sums.shape) == (h_lo, w_lo)
    np.testing.assert_allclose(sums.cpu().numpy(), st.resize_area(dens[..., 1], h_lo, w_lo), rtol=2e-6, atol=0)
    # row shards' partial sums add up to the whole
    a = ctx.stage_burn_sums(D, params, y0=0, y1=H // 3, H_global=H)
    b = ctx.stage_burn_sums(D, params, y0=H // 3, y1=H, H_global=H)
    np.testing.assert_allclose((a + b).cpu().numpy(), sums.cpu().numpy(), rtol=2e-6, atol=0)
    bmap = ctx.stage_burn_map(sums, params, W=W, H_global=H)
    np.testing.assert_allclose(bmap.cpu().numpy(), st.burn_map(dens[..., 1], 1.2, burn_scale), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(139, 229), (140, 228), (133, 224), (61, 97)])
def test_burn_upsample_edges_like_scipy_zoom(ctx, shape):
    """Frames that are not a multiple of the shrink factor get an edge-padded map, and SciPy's zoom can turn its last
    column / row into 0 (coordinate one ulp past the last sample) -- the device path must do exactly the same."""
    neg, prt, _ = stocks()
    H, W = shape
    p = oracle_inputs(neg, prt, 166.67, halation=False, mtf=False, grain=0)
    p.highlight_burn, p.burn_scale, p.d_ref = 0.5, 20.0, float(neg.d_ref[1])
    img = synthetic_frame(H, W, seed=77)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    params.flags |= 32
    params.burn_cell, params.burn_strength, params.burn_d_ref = st.burn_geometry(H, W, 20.0)[0], 0.5, float(neg.d_ref[1])
    out, _ = ctx.render(dev(img), params)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, f"burn edges {shape}")


@pytest.mark.parametrize("grain", [2, 0])
def test_full_pipeline_with_highlight_burn(ctx, grain):
    neg, prt, _ = stocks()
    H, W = 180, 260
    p = oracle_inputs(neg, prt, 150.0, grain=grain)
    p.highlight_burn, p.burn_scale, p.d_ref = 0.7, 50.0, float(neg.d_ref[1])
    img = synthetic_frame(H, W, seed=70)
    img[40:90, 60:140] *= 12.0  # a bright region so that the map is not flat
    ref = st.render(img, p, keep_stages=True)
    assert np.abs(p.stages["burn"] - p.stages["grain" if grain else "mtf"]).max() > 0.05
    params = setup_ctx(ctx, p)
    params.flags |= 32
    params.burn_cell, params.burn_strength, params.burn_d_ref = st.burn_geometry(H, W, 50.0)[0], 0.7, float(neg.d_ref[1])
    out, _ = ctx.render(dev(img), params)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "pipeline with burn")


# ------------------------------------------------------------------------------- pre-path chroma NR
@pytest.mark.parametrize("layout", ["hwc3", "hwc4", "chw"])
@pytest.mark.parametrize("shape,size", [((28, 36), 2), ((70, 1100), 5), ((33, 50), 10), ((1, 9), 3)])
def test_chroma_nr_against_oracle(ctx, layout, shape, size):
    H, W = shape
    xyz = st.apply_matrix3x3(synthetic_frame(H, W, seed=80 + size), st.REC709_TO_XYZ)
    xyz[0, :2] = 0.0
    ref = st.chroma_nr_filter(xyz, size)
    if layout == "hwc3":
        t = dev(xyz)
    elif layout == "hwc4":
        t = dev(np.concatenate([xyz, np.ones((H, W, 1), np.float32)], axis=-1))
    else:
        t = to_planes(xyz)
    out = from_planes(ctx.chroma_nr(t, size))
    assert_close(out, ref, 5e-6, 1e-4, f"chroma nr {shape} size {size}")


def test_chroma_nr_against_reference_golden(ctx, golden_dir):
    import os

    nr = np.load(os.path.join(golden_dir, "chroma_nr.npz"))
    for i, size in enumerate(nr["sizes"][:4]):
        out = from_planes(ctx.chroma_nr(dev(nr["xyz"]), int(size)))
        np.testing.assert_allclose(out, nr[f"out_{i}"], rtol=4e-6, atol=1e-9)


def test_chroma_nr_row_shards_bitwise(ctx):
    H, W, size = 60, 80, 4
    xyz = st.apply_matrix3x3(synthetic_frame(H, W, seed=90), st.REC709_TO_XYZ)
    t = dev(xyz)
    whole = ctx.chroma_nr(t, size)
    tmp = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_chroma_nr_h(t, tmp, size)
    out = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    for a, b in ((0, 17), (17, 41), (41, 60)):
        lo, hi = max(a - size, 0), min(b + size, H)
        ctx.stage_chroma_nr_v(tmp[:, lo:hi].contiguous(), out, size, src_gy0=lo, y0=a, y1=b, H_global=H)
    np.testing.assert_array_equal(out.cpu().numpy(), whole.cpu().numpy())


# ------------------------------------------------------------------------------- pre-path area down-scale
@pytest.mark.parametrize("layout", ["hwc3", "chw"])
@pytest.mark.parametrize("shape,out", [((120, 180), (40, 60)), ((133, 217), (37, 61)), ((64, 64), (64, 64)), ((50, 70), (1, 1)), ((90, 30), (7, 29))])
def test_resize_area_against_oracle(ctx, layout, shape, out):
    H, W = shape
    img = synthetic_frame(H, W, seed=95)
    ref = np.stack([st.resize_area(img[..., c], out[0], out[1]) for c in range(3)], axis=-1)
    t = dev(img) if layout == "hwc3" else to_planes(img)
    got = from_planes(ctx.resize_area(t, out[0], out[1]))
    assert_close(got, ref, 2e-6, 1e-6, f"area resize {shape}->{out}")


def test_grain_field_split_equals_the_fused_tail_bit_for_bit(ctx):
    """r2f_stage_grain_field + r2f_stage_tail_field (the field made ahead of time, e.g. on a side stream) == r2f_stage_tail."""
    neg, prt, _ = stocks()
    H, W = 150, 212
    for grain in (2, 1):
        p = oracle_inputs(neg, prt, 341.33, halation=False, mtf=False, grain=grain, seed=77)
        params = setup_ctx(ctx, p)
        rng = np.random.default_rng(grain)
        dens = rng.uniform(0.0, 3.5, (H, W, 3)).astype(np.float32)
        D = to_planes(dens)
        fused = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        ctx.stage_tail(D, params, out_f32=fused, y0=0, y1=H, H_global=H)
        F = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_grain_field(F, params, y0=0, y1=H, H_global=H)
        split = torch.empty_like(fused)
        ctx.stage_tail_field(D, F, params, out_f32=split, y0=0, y1=H, H_global=H)
        assert torch.equal(fused, split)
        # and in two row ranges with their own field buffers (what a row shard does)
        for y0, y1 in ((0, 70), (70, H)):
            Fp = torch.empty((3, y1 - y0, W), dtype=torch.float32, device="cuda")
            ctx.stage_grain_field(Fp, params, dst_gy0=y0, y0=y0, y1=y1, H_global=H)
            part = torch.empty((y1 - y0, W, 3), dtype=torch.float32, device="cuda")
            ctx.stage_tail_field(D, Fp, params, field_gy0=y0, out_f32=part, out_gy0=y0, y0=y0, y1=y1, H_global=H)
            assert torch.equal(part, fused[y0:y1])
        ref = st.apply_lut_tetrahedral(st.apply_grain(dens, p.grain_lut, p.grain_kernel, p.seed, grain == 1), p.lut_3d, 0.25)
        assert_close(split.cpu().numpy(), ref, 1e-5, 1e-3, "split tail vs oracle")


@pytest.mark.parametrize("n", [3, 5, 7, 9, 11, 13, 15, 19, 21])
@pytest.mark.parametrize("mono, per_channel", [(False, False), (True, False), (False, True)])
def test_small_square_grain_stencils_unrolled_form_against_the_oracle_and_the_entry_list(ctx, n, mono, per_channel):
    """Square mirror-symmetric grain stencils up to 19 x 19 take the fully unrolled form (stencil_fixed<R, 2>), 21 x 21 and
    anything irregular the generic entry list; both reproduce the oracle's field (K_g * N at global coordinates) and each
    other to rounding, with shared or per-channel taps, colour or monochrome noise, on row ranges of any origin."""
    rng = np.random.default_rng(n)
    H, W = 150, 203
    k = rng.uniform(0.1, 1.0, (n, n, 3 if per_channel else 1)).astype(np.float32)
    k = (k + k[:, ::-1]) / 2  # left-right mirror symmetric, like sfl's grain kernels; nothing special vertically
    k /= np.sqrt((k ** 2).sum(axis=(0, 1), keepdims=True))
    ctx.set_kernel(2, k if per_channel else k[..., 0])
    ctx.set_grain_lut(stocks()[0].get_grain_curve(341.33, adx=False, bw_grain=False))  # the field does not use it; the stage wants one
    params = ctx.make_params(seed=4321, grain=True, grain_mono=mono)
    noise = st.gaussian_noise(np.arange(W)[None, :], np.arange(H)[:, None], 4321, mono)
    # grain.wgsl:63-75: the field reads the noise with coordinates clamped to the frame
    r = n // 2
    padded = np.pad(noise, ((r, r), (r, r), (0, 0)), mode="edge").astype(np.float64)
    ref = np.zeros((H, W, 3))
    for c in range(3):
        kc = k[..., c if per_channel else 0].astype(np.float64)
        for i in range(n):
            for j in range(n):
                ref[..., c] += kc[i, j] * padded[i:i + H, j:j + W, c]
    fields = {}
    for fixed in (1, 0):
        ctx.set_option("grain_fixed", fixed)
        assert [c["unrolled"] for c in ctx.stencil_stats(2)] == [n // 2 if fixed and n <= 19 else 0] * 3
        F = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        for y0, y1 in ((0, 37), (37, 38), (38, H)):  # the tile grid follows y0: every pixel must not care
            ctx.stage_grain_field(F, params, dst_gy0=0, y0=y0, y1=y1, H_global=H)
        fields[fixed] = from_planes(F)
        assert np.abs(fields[fixed] - ref).max() <= 5e-6 * np.abs(ref).max(), (fixed, n)  # the noise itself is good to 1e-5 of |n| < 6
        whole = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_grain_field(whole, params, dst_gy0=0, y0=0, y1=H, H_global=H)
        np.testing.assert_array_equal(from_planes(whole), fields[fixed])  # bit for bit whatever the row range
    ctx.set_option("grain_fixed", 1)
    assert np.abs(fields[1] - fields[0]).max() <= 1e-6 * np.abs(ref).max()  # two summation orders of the same 2-D sum


@pytest.mark.parametrize("n", [3, 5, 9, 13, 19])
@pytest.mark.parametrize("mono, per_channel", [(False, True), (False, False), (True, False), (True, True)])
def test_separable_grain_stencils_run_as_two_1d_passes(ctx, n, mono, per_channel):
    """A grain stencil that is u v^T to fp32 rounding (any Gaussian-like kernel: the stand-in of filmstock.grain_kernel, per
    channel or shared) runs as two 1-D passes; the field agrees with the 2-D sum of the same fp32 taps to rounding, is bit for
    bit independent of the row range, and a stencil that is NOT rank one (one tap off by 1e-5) keeps the 2-D form."""
    H, W = 150, 203
    r = n // 2
    ax = np.arange(-r, r + 1)
    sig = np.array([0.35 * r + 0.3, 0.3 * r + 0.4, 0.4 * r + 0.35])
    k = np.exp(-(ax[:, None, None] ** 2 + ax[None, :, None] ** 2) / (2.0 * sig[None, None, :] ** 2))
    k = (k / np.sqrt((k ** 2).sum(axis=(0, 1), keepdims=True))).astype(np.float32)
    if not per_channel:
        k = np.repeat(k[..., :1], 3, axis=2)
    # monochrome noise with per-channel taps keeps the 2-D form (one noise plane cannot be filtered in place three ways)
    two_pass = not (mono and per_channel)
    ctx.set_grain_lut(stocks()[0].get_grain_curve(341.33, adx=False, bw_grain=False))
    params = ctx.make_params(seed=99, grain=True, grain_mono=mono)
    noise = st.gaussian_noise(np.arange(W)[None, :], np.arange(H)[:, None], 99, mono)
    padded = np.pad(noise, ((r, r), (r, r), (0, 0)), mode="edge").astype(np.float64)

    def field(kk, y_ranges=((0, H),)):
        ctx.set_kernel(2, kk)
        F = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        for y0, y1 in y_ranges:
            ctx.stage_grain_field(F, params, dst_gy0=0, y0=y0, y1=y1, H_global=H)
        return from_planes(F)

    ref = np.zeros((H, W, 3))
    for c in range(3):
        for i in range(n):
            for j in range(n):
                ref[..., c] += float(k[i, j, c]) * padded[i:i + H, j:j + W, c]
    sep = field(k)
    assert [c["separable"] for c in ctx.stencil_stats(2)] == [1, 1, 1]
    assert np.abs(sep - ref).max() <= 5e-6 * np.abs(ref).max()
    np.testing.assert_array_equal(field(k, ((0, 37), (37, 38), (38, H))), sep)
    ctx.set_option("grain_separable", 0)
    try:
        full2d = field(k)
        assert [c["separable"] for c in ctx.stencil_stats(2)] == [0, 0, 0]
    finally:
        ctx.set_option("grain_separable", 1)
    assert np.abs(sep - full2d).max() <= 2e-6 * np.abs(ref).max() and np.array_equal(sep, full2d) != two_pass
    if n >= 5:
        bad = k.copy()
        bad[0, 1, :] *= np.float32(1.0 + 1e-3)
        bad[0, n - 2, :] = bad[0, 1, :]  # still mirror symmetric, no longer rank one
        got = field(bad)
        assert [c["separable"] for c in ctx.stencil_stats(2)] == [0, 0, 0]
        ref_bad = ref.copy()
        for c in range(3):
            for j in (1, n - 2):
                ref_bad[..., c] += (float(bad[0, j, c]) - float(k[0, j, c])) * padded[0:H, j:j + W, c]
        assert np.abs(got - ref_bad).max() <= 5e-6 * np.abs(ref).max()


@pytest.mark.parametrize("layout", ["hwc3", "hwc4", "chw"])
@pytest.mark.parametrize("matrix", [True, False])
def test_fused_pointwise_fast_path_matches_generic(ctx, layout, matrix):
    """BASELINE config 2's single fused kernel has a specialised form (r2f_front.hip: 32-bit table offsets, packed channel
    pairs, non-negative cell arithmetic for the 3-D LUT); it must give the generic kernel's bits -- float and uint8 -- for every
    input layout, with and without S0, including pixels in the S < 1e-12 branch, densities past the LUT's last cell, exact
    ties between the tetrahedron's fractions, and a density curve that dips below zero (the wave then takes the general cell
    arithmetic, negative indices wrapping like Python's)."""
    neg, prt, _ = stocks()
    H, W = 96, 256
    p = oracle_inputs(neg, prt, 166.67, halation=False, mtf=False, grain=0, matrix=matrix)
    img = synthetic_frame(H, W, seed=31)
    img[0, :8] = 0.0
    img[1, :8] = [1e-9, 0, 0]
    img[2, :16] = 0.18  # grey: equal fractions on the three axes after a neutral curve
    img[3, :4] = 60000.0
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    t = {"hwc3": lambda: dev(img), "hwc4": lambda: dev(np.concatenate([img, np.ones((H, W, 1), np.float32)], axis=-1)),
         "chw": lambda: to_planes(img)}[layout]()

    def both(curve=None):
        if curve is not None:
            ctx.set_curve1d(curve)
        outs = []
        for fast in (1, 0):
            ctx.set_option("front_fast", fast)
            outs.append(ctx.render(t, params, want_f32=True, want_u8=True, layout=layout))
        ctx.set_option("front_fast", 1)
        return outs

    (f_fast, u_fast), (f_gen, u_gen) = both()
    assert torch.equal(f_fast, f_gen) and torch.equal(u_fast, u_gen)
    # (without S0 the Rec.709 numbers are read as XYZ: saturated "colours" whose red exposure is a difference of LUT terms a
    # thousand times larger -- fp32 rounding noise of either implementation is then 1e-5 of the result; the comparison with the
    # oracle is made on the realistic input, the bit-equality of the two kernels on both)
    assert_close(f_fast.cpu().numpy(), ref, 1e-5 if matrix else 5e-5, 1e-3, "fast fused pass")
    # a curve shifted below zero: negative densities reach the 3-D LUT
    low = p.lut_1d.copy()
    low[1:] -= 0.6
    (f_fast, u_fast), (f_gen, u_gen) = both(low)
    assert torch.equal(f_fast, f_gen) and torch.equal(u_fast, u_gen)
    ctx.set_curve1d(p.lut_1d)


@pytest.mark.parametrize("bw", [False, True])
def test_identity_halation_channels_are_finished_by_the_front_kernel(ctx, bw):
    """Whole-frame renders hand the halation's single-tap channels (blue on a colour stock: f_b = 0) to the front kernel: tap
    weight, log and curve there, density written directly, no exposure plane and no pointwise pass for them.  Same bits as the
    path through the exposure plane; a black-and-white stock (halation on all three layers) has nothing to hand over."""
    neg, prt, bwstock = stocks()
    H, W = 200, 320
    p = oracle_inputs(bwstock if bw else neg, prt, 250.0)
    img = synthetic_frame(H, W, seed=77)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    t = dev(img)
    E = torch.full((3, H, W), -1.0, dtype=torch.float32, device="cuda")
    D = torch.full((3, H, W), -1.0, dtype=torch.float32, device="cuda")
    mask = ctx.stage_front_split(t, params, E, D)
    assert mask == (0 if bw else 4)
    E_plain = torch.empty_like(E)
    ctx.stage_front(t, params, 0, dst=E_plain)
    if bw:
        assert torch.equal(E, E_plain) and float(D.max()) == -1.0
    else:
        assert torch.equal(E[:2], E_plain[:2]) and float(E[2].max()) == -1.0  # the blue exposure plane is never written
        D_ref = torch.empty_like(D)
        ctx.stage_halation(E_plain, D_ref, params, y0=0, y1=H, H_global=H)
        ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H, identity_done=mask)
        assert torch.equal(D, D_ref)
    outs = []
    ctx.set_option("stencil_fft_scratch96_auto", 0)  # (only the fast front kernel records the exposure range the 12-byte element is chosen from)
    for fast in (1, 0):  # r2f_render takes the split front when the fast kernel applies
        ctx.set_option("front_fast", fast)
        outs.append(ctx.render(t, params)[0])
    ctx.set_option("front_fast", 1)
    ctx.set_option("stencil_fft_scratch96_auto", 1)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0].cpu().numpy(), ref, 1e-5, 1e-3, "render with the split front")


@pytest.mark.parametrize("scale", [229.33, 341.33])
def test_full_pipeline_on_a_photograph_like_frame(ctx, scale):
    """Smooth gradients, flat patches (exact ties in the tetrahedron choice, densities sitting on LUT nodes and curve breakpoints),
    a hard edge and a specular -- the kind of content the white-noise frames never produce -- against the oracle at the contract."""
    neg, prt, _ = stocks()
    H, W = 320, 448
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    lum = 0.18 * 2.0 ** (3.0 * np.sin(xx / 97.0) * np.cos(yy / 61.0))
    img = np.stack([lum * (1.0 + 0.3 * np.sin(yy / 40.0)), lum, lum * (1.0 + 0.3 * np.cos(xx / 53.0))], axis=-1).astype(np.float32)
    img[40:120, 60:200] = (0.18, 0.18, 0.18)      # a flat grey card: every pixel the same cell, dr == dg == db ties
    img[150:220, 250:400] = (0.5, 0.25, 0.125)    # a flat colour patch
    img[:, 300:302] *= 0.02                        # a dark hairline
    img[200:204, 100:104] = 40.0                   # a specular
    img[260:, :] = np.float32(0.0)                 # black border: S < 1e-12 in the 2-D LUT
    p = oracle_inputs(neg, prt, scale)
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, u8 = ctx.render(dev(img), params, want_f32=True, want_u8=True)
    e = assert_close(out.cpu().numpy(), ref, 1e-5, 1e-3, "photograph-like frame")
    print(f"photograph-like frame scale {scale}: max err {e:.2e}")
    diff = np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(ref).astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4
