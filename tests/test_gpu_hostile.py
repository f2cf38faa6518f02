"""Parity battery with hostile tables (Next 3): the four stages whose reference lives in spectral_film_lut (S1, S3 + S4,
S6c) and S8, against the oracle with table contents no analytic stand-in stock produces -- random texels, other sizes than the
defaults (2-D LUT n = 17 / 64 / 128, curves m = 256 / 4096 on NON-uniform axes, 3-D LUT n = 17 / 33 / 65), steps in the grain LUT,
exact 0 / 1 plateaus and blacks below the contract's 1e-3 floor -- and one full render through a BundleStock that went through
save_bundle / load_bundle.

Two kinds of test:
  * per stage, both sides fed IDENTICAL inputs, fully random tables (rough = 1): what is compared is the stage's own arithmetic;
    the bound is the contract's plus, where the stage's input is itself a rounded intermediate, the local slope of the table times
    a few ulp of that input (a table that jumps by 1 across a cell turns one ulp of its argument into 1e-5 of its value: that is
    the table's conditioning, not an error of either implementation);
  * the whole path with tables whose roughness the contract can bear (rough <= 0.25, bounded curve slopes), at the contract's own
    numbers: |hip - oracle| <= 1e-5 max(|oracle|, 1e-3), uint8 <= 1 LSB on <= 1e-4 of the samples.
Layouts: gpu_processor.py:307-409, 565-611; lut_1d.wgsl:43-51; grain.wgsl:78-89; utils.py:247-380."""

import numpy as np
import pytest

from oracle import kernels as ok
from oracle import stages as st

import hostile
from helpers import SEED, oracle_inputs, rel_err, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_parity import dev, from_planes, setup_ctx, to_planes  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from raw2film_amd.context import HipContext

    c = HipContext(0)
    yield c
    c.close()


def _frame(H, W, seed):
    img = synthetic_frame(H, W, seed=seed)
    img[0, :5] = 0.0  # S < 1e-12 -> 0 (lut_2d.wgsl:24)
    img[1, :5] = 1e-9
    img[2, :5] = 60000.0
    return img


@pytest.mark.parametrize("layout", ["hwc3", "chw"])
@pytest.mark.parametrize("n", [17, 64, 128])
def test_random_input_lut_of_any_size(ctx, n, layout):
    """S1 with independent random texels (and one near-black texel), XYZ in: both sides index the table with the same float32
    quotient, so what differs is the rounding of the blend.  n = 128 (256 KB of float4 texels) does not fit LDS: the generic
    front kernel; n <= 64: the LDS kernel.  Weights and texels are non-negative, so the blend has no cancellation: a few ulp.
    (With S0 in front, one ulp of X moves the chromaticity index by 8e-6 of a texel; next to the near-black texel that alone is
    2e-4 of the value -- the table's conditioning, measured here before the matrix was taken out.)"""
    rng = np.random.default_rng(100 + n)
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0, matrix=False)
    p.lut_2d = hostile.lut2d(rng, n, 1.0)
    H, W = 96, 132
    img = _frame(H, W, n)
    ref = st.apply_2d_lut(img, p.lut_2d)
    params = setup_ctx(ctx, p)
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_front(dev(img) if layout == "hwc3" else to_planes(img), params, 0, dst=E, layout=layout)
    got = from_planes(E)
    err = rel_err(got, ref, 1e-6)
    assert err <= 2e-6, err
    assert (got[0, :5] == 0).all()


def _curve_slopes(lut):
    xp = lut[0].astype(np.float64)
    dx = np.diff(xp)
    return [np.where(dx > 0, np.abs(np.diff(lut[1 + c].astype(np.float64))) / np.maximum(dx, 1e-300), 0.0) for c in range(3)]


def _local_slope(lut, x):
    """max |slope| of the curve's cell at x and of its two neighbours, per channel of x (..., 3)."""
    xp = lut[0].astype(np.float64)
    out = np.zeros(x.shape)
    sl = _curve_slopes(lut)
    for c in range(3):
        i = np.clip(np.searchsorted(xp, x[..., c].astype(np.float64), side="right") - 1, 0, len(xp) - 2)
        s = sl[c]
        out[..., c] = np.maximum(np.maximum(s[np.maximum(i - 1, 0)], s[i]), s[np.minimum(i + 1, len(s) - 1)])
    return out


@pytest.mark.parametrize("m,uniform", [(256, False), (4096, False), (1024, True), (2, False), (37, False)])
def test_density_curve_on_a_non_uniform_axis(ctx, m, uniform):
    """S3 + S4 (np.interp semantics, lut_1d.wgsl:43-51 without the half-texel shift) on curves whose axis the device cannot
    index arithmetically: the exact cell walk.  m = 4096 (196 KB of cells) also exceeds LDS.  The input of the curve is
    log10 of an exposure both sides compute (flat 2-D LUT: exposure = X + Y + Z to an ulp); the device's v_log_f32-based log10
    and NumPy's differ by an ulp or two of the logarithm, which the curve's local slope turns into density."""
    rng = np.random.default_rng(200 + m)
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0, matrix=False)
    p.lut_2d = np.ones((8, 8, 3), dtype=np.float32)
    p.lut_1d = hostile.curve(rng, m, max_slope=6.0, uniform=uniform)
    H, W = 128, 164
    img = _frame(H, W, m)
    expo = st.apply_2d_lut(img, p.lut_2d)
    loge = st.log_clip(expo)
    ref = st.multi_channel_interp(loge, p.lut_1d)
    params = setup_ctx(ctx, p)
    D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_front(dev(img), params, 1, dst=D)
    got = from_planes(D)
    bound = 1e-5 * np.maximum(np.abs(ref), 1e-3) + _local_slope(p.lut_1d, loge) * 4.0 * np.spacing(np.abs(loge).astype(np.float32))
    over = np.abs(got.astype(np.float64) - ref) > bound
    assert not over.any(), (int(over.sum()), float(np.max(np.abs(got - ref) / bound)))
    # clamped at both ends like np.interp
    assert np.array_equal(got[0, :5], np.broadcast_to(p.lut_1d[1:, 0], (5, 3)))  # exposure 0 -> log clip -> below the axis


@pytest.mark.parametrize("mode", ["tetrahedral", "trilinear"])
@pytest.mark.parametrize("n", [2, 17, 33, 65, 40])
def test_random_output_lut_with_plateaus(ctx, n, mode):
    """S8 alone on given density planes, independent random texels with exact 0 / 1 entries: both sides index with the same
    float32 product, so this is the interpolation arithmetic itself.  Densities include exact grid points, ties and the upper
    edge (>= 4 -> last cell, d = 1: utils.py:270-289)."""
    rng = np.random.default_rng(300 + n)
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 100.0, halation=False, mtf=False, grain=0)
    p.lut_3d = hostile.lut3d(rng, n, 1.0)
    p.lut3d_mode = mode
    H, W = 96, 128
    dens = rng.uniform(0.0, 4.4, (H, W, 3)).astype(np.float32)
    grid = (np.arange(n) * (4.0 / (n - 1))).astype(np.float32)
    dens[0, :n] = grid[:W, None][:n]
    dens[1, :, 1] = dens[1, :, 0]  # ties dr == dg
    dens[2, :, 2] = dens[2, :, 1]
    dens[3] = 4.0
    ref = st.apply_lut_tetrahedral(dens, p.lut_3d, 0.25) if mode == "tetrahedral" else st.apply_lut_trilinear(dens, p.lut_3d, 0.25)
    params = setup_ctx(ctx, p)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    ctx.stage_tail(to_planes(dens), params, out_f32=out, out_u8=u8, y0=0, y1=H, H_global=H)
    got = out.cpu().numpy()
    # (trilinear, lut_3d.wgsl:27-40, is the optional GPU-twin mode, off the parity path: the device keeps the shader's float32
    # coordinate, the oracle a float64 one -- an ulp of a coordinate of up to 64, i.e. 8e-6 of a cell, times random texel
    # differences of up to 1)
    assert np.max(np.abs(got - ref)) <= (4e-7 if mode == "tetrahedral" else 1e-5)
    d = np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(ref).astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= (2e-4 if mode == "tetrahedral" else 2e-3)
    assert (ref == 0).any() or n == 2 or mode == "trilinear"  # (exact-grid densities return the plateau texels themselves)


@pytest.mark.parametrize("m", [256, 64, 1000])
@pytest.mark.parametrize("mono", [False, True])
def test_grain_lut_with_steps(ctx, m, mono):
    """S6 with a piecewise-constant grain LUT (jumps across single cells, amplitudes up to 0.06: three times the stand-in
    stocks'), planes -> planes: out = max(D + G * lut(D), 0).  Densities from 0.35 up at the contract's bound; the
    field itself is good to 4e-6 (hardware transcendentals), i.e. 2.4e-7 of density at this amplitude, so the row of
    exact-zero densities -- where the clip at 0 (cpu_processor.py:397) decides -- is held to that absolute figure instead."""
    rng = np.random.default_rng(400 + m)
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 166.67, halation=False, mtf=False, grain=1 if mono else 2)
    p.grain_lut = hostile.grain_lut(rng, m)
    H, W = 100, 140
    dens = rng.uniform(0.35, 4.0, (H, W, 3)).astype(np.float32)  # (0.35 - 5 sigma x 0.06 > 0: the contract rows stay off the clip)
    dens[0] = np.linspace(0.35, 4, W, dtype=np.float32)[:, None]
    dens[1] = 0.0
    ref = np.maximum(st.apply_grain(dens, p.grain_lut, p.grain_kernel, p.seed, p.grain_mono), 0)
    params = setup_ctx(ctx, p)
    G = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_grain(to_planes(dens), G, params, y0=0, y1=H, H_global=H)
    got = from_planes(G)
    assert rel_err(np.delete(got, 1, axis=0), np.delete(ref, 1, axis=0), 1e-3) <= 1e-5
    assert (got[1] >= 0).all() and (got[1] == 0).any() and np.max(np.abs(got[1] - ref[1])) <= 3e-7


def _hostile_inputs(rng, neg, prt, scale, n2, m1, n3, **kw):
    return hostile.roughen(rng, oracle_inputs(neg, prt, scale, **kw), n2, m1, n3)


@pytest.mark.parametrize("n2,m1,n3", [(17, 256, 17), (64, 4096, 33), (128, 256, 65), (33, 1000, 24)])
def test_whole_path_with_hostile_tables_at_the_contract(ctx, n2, m1, n3):
    """S0..S8 with halation, MTF and grain on tables of moderate roughness, every size off the defaults, a print-like output LUT
    that falls to exact 0: the contract as written, with samples below its 1e-3 floor (the stand-in stocks never get under 5e-3)."""
    rng = np.random.default_rng(n2 * 1000 + n3)
    neg, prt, _ = stocks()
    scale = 229.33
    p = _hostile_inputs(rng, neg, prt, scale, n2, m1, n3)
    H, W = 200, 280
    img = synthetic_frame(H, W, seed=n2)
    img[40:90, 60:150] *= 0.02  # a shadow region: thin negative -> the output LUT's exact-1 plateau
    img[110:170, 20:120] *= 1000.0  # ... and blown highlights: the black end, down to its exact zeros
    ref = st.render(img, p)
    params = setup_ctx(ctx, p)
    out, u8 = ctx.render(dev(img), params, want_f32=True, want_u8=True)
    got = out.cpu().numpy()
    err = rel_err(got, ref, 1e-3)
    below = int((ref < 1e-3).sum())
    assert below > 100, below
    assert (ref == 0).any() and (ref == 1).any()
    assert err <= 1e-5, (err, below)
    d = np.abs(u8.cpu().numpy().astype(int) - st.to_uint8(ref).astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-4, (d.max(), (d > 0).mean())


@pytest.mark.parametrize("n2,rough2,rough3", [(64, 0.2, 0.1), (33, 1.0, 0.1), (128, 0.5, 0.3)])
def test_whole_path_with_a_near_black_texel_and_full_texel_noise_against_the_float64_truth(ctx, n2, rough2, rough3):
    """The tables round 3 took OUT of the whole-path battery ("a conditioning bomb": one input-LUT texel at 1e-5 among neighbours of
    order 1 -- exposures next to it dive towards the log clip with a relative slope of 1e5 per texel -- and texel noise that does not
    shrink with the table's pitch) are back, under a criterion instead of a tolerance: against the float64 evaluation of the same
    formulas (oracle/truth.py) the kernels may be off by the contract's 1e-5 max(|truth|, 1e-3) plus what the float32 ORACLE is
    off by itself at that pixel.  Where the oracle is exact to rounding the bound is the contract's; next to the black texel both
    sides lose digits, and the device may lose no more than the oracle does."""
    from oracle import truth

    rng = np.random.default_rng(n2 * 7 + int(rough3 * 10))
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 229.33)
    p.lut_2d = hostile.lut2d(rng, n2, rough2, base=p.lut_2d, black_texel=True)
    p.lut_1d = hostile.curve(rng, 512, v_lo=0.08, v_hi=3.6, max_slope=1.5, monotone=True)
    p.lut_3d = hostile.lut3d(rng, 33, rough3)
    H, W = 200, 280
    img = synthetic_frame(H, W, seed=n2)
    img[40:90, 60:150] *= 0.02
    img[110:170, 20:120] *= 1000.0
    ref, exact = st.render(img, p), truth.render(img, p)
    params = setup_ctx(ctx, p)
    out, _ = ctx.render(dev(img), params, want_f32=True)
    got = out.cpu().numpy().astype(np.float64)
    slack = np.abs(ref - exact)
    ratio = np.abs(got - exact) / (1e-5 * np.maximum(np.abs(exact), 1e-3) + slack)
    assert float(ratio.max()) <= 1.0, float(ratio.max())
    # the criterion has teeth: on most of the frame the oracle's own slack is a small part of the bound ...
    assert np.median(slack / (1e-5 * np.maximum(np.abs(exact), 1e-3))) < 0.2


def test_full_render_through_a_round_tripped_bundle(tmp_path):
    """filmstock.save_bundle -> load_bundle -> HipProcessor: the LUT-bundle route by which real spectral_film_lut exports reach this
    backend (filmstock.py), with hostile arrays in the bundle; compared with the oracle fed the same arrays."""
    from raw2film_amd import HipProcessor, filmstock

    rng = np.random.default_rng(77)
    neg, prt, _ = stocks()
    H, W, fw = 160, 240, 1.2
    scale = max(H, W) / fw
    p = _hostile_inputs(rng, neg, prt, scale, 48, 777, 21, grain_size=6.0, grain_sigma=0.4, halation_green_factor=0.3)
    path = str(tmp_path / "hostile_stock.npz")
    filmstock.save_bundle(path, lut_2d=p.lut_2d, lut_1d=p.lut_1d, lut_3d=p.lut_3d, grain_lut=p.grain_lut,
                          mtf_logf=np.stack([m[0] for m in neg.mtf]), mtf_vals=np.stack([m[1] for m in neg.mtf]),
                          rms_density=neg.rms_density, d_ref=np.asarray(neg.d_ref), density_measure="status_m")
    stock = filmstock.load_bundle(path, name="hostile bundle")
    assert stock.mtf is not None and stock.rms_density is not None
    for a, b in ((stock.get_input_lut(), p.lut_2d), (stock.get_density_curve(), p.lut_1d), (stock.get_grain_curve(scale), p.grain_lut),
                 (filmstock.create_lut(stock, None), p.lut_3d)):
        np.testing.assert_array_equal(a, b)
    p.mtf_kernel = ok.mtf_kernel(stock.mtf, scale, 0.0, 1.0)
    img = synthetic_frame(H, W, seed=9)
    proc = HipProcessor(device=0)
    try:
        out = proc.process_array(img, stock, 6, 0.4, colorspace="linear-rec709", seed=SEED, return_float=True, frame_width=fw,
                                 frame_height=fw * H / W, halation_green_factor=0.3)
    finally:
        proc.close()
    ref = st.render(img, p)
    assert rel_err(out, ref, 1e-3) <= 1e-5


@pytest.mark.parametrize("scale,shape", [(341.33, (700, 900)), (60.0, (300, 400))])  # FFT stencils / direct stencils
def test_non_finite_and_absurd_samples_stay_inside_the_stencils_reach(scale, shape):
    """NaN, infinities, negative and 3e38 samples in the frame: no fault, a finite output, and pixels further away than the
    stencils reach are the clean render's (bit for bit in the direct form; to the rounding class of a window that holds a 65 504
    specular in the FFT form, whose pass 1 takes a non-finite sample as 0 instead of handing it to every output of its window)."""
    from raw2film_amd.context import HipContext

    H, W = shape
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, scale, seed=SEED)
    c = HipContext(0)
    try:
        params = setup_ctx(c, p)
        img = synthetic_frame(H, W, seed=3)
        clean, _ = c.render(torch.from_numpy(img).cuda(), params)
        clean = clean.cpu().numpy()
        bad = img.copy()
        cy, cx = H // 2, W // 2
        vals = [np.nan, np.inf, -np.inf, -5.0, 3e38, -3e38, 0.0, -0.0, 1e-45]
        for i, v in enumerate(vals):
            bad[cy + 3 * i, cx, i % 3] = v
            bad[5, 5 + 7 * i, :] = v
        out, u8 = c.render(torch.from_numpy(bad).cuda(), params, want_u8=True)
        o = out.cpu().numpy()
        assert np.isfinite(o).all() and u8.shape == (H, W, 3)
        reach = sum(k.shape[0] // 2 for k in (p.halation_kernel, p.mtf_kernel, p.grain_kernel) if k is not None) + 1
        far = np.ones((H, W), bool)
        far[max(cy - reach, 0):cy + 3 * len(vals) + reach, max(cx - reach, 0):cx + reach] = False
        far[:5 + reach + 1, :5 + 7 * len(vals) + reach] = False
        assert far.mean() > 0.5
        fft = any(s["fft"] for s in c.stencil_stats(0)) or any(s["fft"] for s in c.stencil_stats(1))
        if fft:
            assert np.max(np.abs(o[far] - clean[far]) / np.maximum(np.abs(clean[far]), 1e-3)) <= 5e-6
        else:
            assert np.array_equal(o[far], clean[far])
    finally:
        c.close()


def test_a_non_finite_sample_in_a_stencil_stage_direct_form_against_fft_form():
    """what each form of ONE stencil does with a NaN / infinity in its input plane, pinned at the stage entry (no
    epilogue).  Direct form (like the per-tap loop of convolution.wgsl): NaN in every output whose tap box covers the sample,
    everything else bit-identical to the clean plane's result.  FFT form: the sample enters the correlation as 0 (pass 1; a NaN
    handed to the transforms would come back in every output of its window), so every output is finite, the outputs inside the
    tap box are those of the plane with that sample zeroed, and the rest agrees with the clean result to the transforms' rounding.
    include/r2f.h documents the divergence."""
    from oracle import kernels as ok
    from raw2film_amd.context import HipContext

    rng = np.random.default_rng(8)
    H, W = 300, 420
    img = rng.uniform(0.05, 2.0, (H, W, 3)).astype(np.float32)
    k = ok.compute_halation_kernel(200.0, halation_green_factor=0.3)  # 51 x 51: by FFT by default, direct with stencil_fft = 0
    r = k.shape[0] // 2
    c = HipContext(0)
    try:
        def run(a, fft):
            c.set_option("stencil_fft", fft)
            c.set_kernel(0, k)
            src = torch.from_numpy(np.ascontiguousarray(np.transpose(a, (2, 0, 1)))).cuda()
            dst = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
            c.stage_stencil(0, src, dst, y0=0, y1=H, H_global=H)
            return np.transpose(dst.cpu().numpy(), (1, 2, 0))

        for bad_value in (np.nan, np.inf, -np.inf):
            bad, zeroed = img.copy(), img.copy()
            y, x = 140, 200
            bad[y, x, 0] = bad_value
            zeroed[y, x, 0] = 0.0
            box = np.zeros((H, W), bool)
            box[y - r:y + r + 1, x - r:x + r + 1] = True
            for fft in (0, 1):
                clean, got, zero = run(img, fft), run(bad, fft), run(zeroed, fft)
                assert [s["fft"] for s in c.stencil_stats(0)][:2] == [fft, fft]
                np.testing.assert_array_equal(got[..., 1:], clean[..., 1:])  # the other channels never see the sample
                g, cl, z = got[..., 0], clean[..., 0], zero[..., 0]
                if fft:
                    assert np.isfinite(g).all()
                    np.testing.assert_array_equal(g, z)  # exactly the plane with the sample zeroed
                    assert np.max(np.abs(g[~box] - cl[~box]) / np.maximum(np.abs(cl[~box]), 1e-3)) <= 1e-6
                else:
                    taps = k[::-1, ::-1, 0] != 0  # correlation: output (y - i + r, x - j + r) reads the sample through tap (i, j)
                    hit = np.zeros((H, W), bool)
                    hit[y - r:y + r + 1, x - r:x + r + 1] = taps
                    assert not np.isfinite(g[hit]).any()  # (inf - inf and 0 * inf included: NaN or +-inf, never a finite number)
                    # (the entry list pads the box with zero weights -- columns to a radius of 2 mod 4 for the mirrored pairs, rows to
                    # the four output rows a lane owns per input row -- and 0 * NaN is NaN: up to three more rows / columns are poisoned)
                    wide = np.zeros((H, W), bool)
                    wide[y - r - 3:y + r + 4, x - r - 3:x + r + 4] = True
                    np.testing.assert_array_equal(g[~wide], cl[~wide])
    finally:
        c.close()


@pytest.mark.parametrize("scale,shape", [(341.33, (300, 600)), (60.0, (200, 300))])
def test_non_finite_values_inside_tables_and_stencils_neither_fault_nor_hang(scale, shape):
    """NaN / infinities sprinkled over each table and stencil in turn: the result may be garbage (so is the reference's), the
    call may refuse the table (ValueError: a curve axis must be non-decreasing) -- but every index stays inside its table and every
    loop ends."""
    import copy

    from raw2film_amd.context import HipContext

    H, W = shape
    neg, prt, _ = stocks()
    base = oracle_inputs(neg, prt, scale, seed=SEED)
    rng = np.random.default_rng(0)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=3)).cuda()
    for name in ("lut_2d", "lut_1d", "lut_3d", "grain_lut", "halation_kernel", "mtf_kernel", "grain_kernel"):
        if getattr(base, name) is None or np.ndim(getattr(base, name)) < 2:
            continue
        for v in (np.nan, np.inf, -np.inf):
            p = copy.copy(base)
            t = np.array(getattr(base, name), copy=True)
            flat = t.reshape(-1)
            flat[rng.integers(0, flat.size, size=max(1, flat.size // 50))] = v
            setattr(p, name, t)
            c = HipContext(0)
            try:
                params = setup_ctx(c, p)
                out, u8 = c.render(frame, params, want_u8=True)
                torch.cuda.synchronize()
                assert out.shape == (H, W, 3) and u8.shape == (H, W, 3)
            except ValueError:
                pass
            finally:
                c.close()
