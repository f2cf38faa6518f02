"""Parity at BASELINE.json's full frame size (12288 x 8192, 87 / 35 / 9-tap stencils), where a whole-frame oracle run
would take minutes: (1) windows of the GPU result against the oracle evaluated on window + halo, with the grain hash at
global coordinates, (2) row shards against the whole frame, bit for bit, (3) a constant frame stays constant."""

import numpy as np
import pytest

from oracle import stages as st

from helpers import SEED, oracle_inputs, stocks

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

W_FULL, H_FULL = 12288, 8192
SCALE = W_FULL / 36.0


@pytest.fixture(scope="module")
def full():
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, SCALE, seed=SEED)
    ctx = HipContext(0)
    params = setup_ctx(ctx, p)
    frame = synthetic_frame_device(H_FULL, W_FULL, seed=7)
    out, _ = ctx.render(frame, params)
    torch.cuda.synchronize()
    yield ctx, params, p, frame, out
    ctx.close()


@pytest.mark.parametrize("y0,x0", [(0, 0), (4000, 6000), (H_FULL - 96, W_FULL - 96), (63, 12288 - 200)])
def test_windows_of_the_100mp_render_match_the_oracle(full, y0, x0):
    ctx, params, p, frame, out = full
    n = 96  # window side; the oracle runs on window + halo
    halo = 42 + 17 + 4  # halation + MTF + grain reach
    ya, yb = max(y0 - halo, 0), min(y0 + n + halo, H_FULL)
    xa, xb = max(x0 - halo, 0), min(x0 + n + halo, W_FULL)
    # a crop in the interior has no real border; reflect-101 inside the oracle only touches the discarded halo.  At the
    # frame's own edges the crop edge IS the frame edge, so the reflection is the real one.
    crop = frame[ya:yb, xa:xb].cpu().numpy()
    x = st.apply_2d_lut(st.apply_matrix3x3(crop, p.matrix), p.lut_2d)
    x = st.halation(x, p.halation_kernel)
    x = st.multi_channel_interp(st.log_clip(x), p.lut_1d)
    x = st.film_sharpness(x, p.mtf_kernel)
    x = st.apply_grain(x, p.grain_lut, p.grain_kernel, p.seed, False, row0=ya, H_global=H_FULL, col0=xa, W_global=W_FULL)
    ref = st.apply_lut_tetrahedral(x, p.lut_3d, 0.25)[y0 - ya:y0 - ya + n, x0 - xa:x0 - xa + n]
    got = out[y0:y0 + n, x0:x0 + n].cpu().numpy()
    err = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3))
    assert err <= 1e-5, err


@pytest.mark.parametrize("mode", ["direct", "fft", "fft_dyn"])
def test_row_shards_of_the_100mp_frame_match_the_whole_frame(full, mode):
    """Direct stencils: bit for bit (every pixel sums its taps in the same order whatever the tile or shard).  FFT stencils:
    the windows are anchored at the shard's first row, so a pixel's 256 x 256 window differs between the two renders and
    with it the fp64 rounding noise (~1e-13): after the one rounding to fp32 a handful of pixels may differ by an ulp.
    fft_dyn (round 6): the shards keep the exposure-range record like r2f_render does (R2F_F_TRACK_RANGE on the front call,
    R2F_F_RANGE_VALID on the halation) and are compared with the DEFAULT whole-frame render: both choose per window pair, nearly
    always the 12-byte element on this frame; the element rounds per window, so the two tilings agree to ITS rounding."""
    ctx, params, p, frame, out = full
    fft, dyn = mode != "direct", mode == "fft_dyn"
    ctx.set_option("stencil_fft", int(fft))
    if fft:  # (the fixture's render is not the context's last one any more: render the default again and ask what it chose)
        again, _ = ctx.render(frame, params)
        assert torch.equal(again, out)
        del again
    if not fft:
        out, _ = ctx.render(frame, params)
    elif not dyn:
        # like for like: stage calls without the record keep complex128 scratch for the halation; the whole-frame render chooses its
        # element per WINDOW PAIR from the pair's own range (round 6): the frame's range (max / shadow = 1.4e5) is beyond the
        # element's guard (6.2e4), its windows' ranges (~2e4) are not -- nearly every pair takes the 12-byte element.  The fixture's
        # default render must agree with the complex128 one to the element's own rounding (at most three ulps of a density)
        rng = ctx.frame_exposure_range()
        assert rng["armed"] and rng["pairs"] > 500 and 0.9 * rng["pairs"] <= rng["packed_pairs"] <= rng["pairs"], rng
        assert rng["max_abs"] > rng["bound"] * max(rng["min"], rng["floor"])  # (the FRAME would not qualify)
        ctx.set_option("stencil_fft_scratch96_auto", 0)
        exact, _ = ctx.render(frame, params)
        ctx.set_option("stencil_fft_scratch96_auto", 1)
        d = (exact - out).abs()
        assert float((d / exact.abs().clamp_min(1e-3)).max()) <= 2e-6
        assert 0 < float((d > 0).float().mean()) <= 5e-3  # (it IS the other element: a few pixels in ten thousand differ)
        out = exact
        del d
    else:
        # the shards keep the record and choose per window pair like the whole frame does; their windows are anchored at the shard's
        # first row, so a pixel's window -- and with it the element's rounding -- differs between the two renders
        rng = ctx.frame_exposure_range()
        assert rng["armed"] and rng["packed_pairs"] > 0, rng
    rh, rm = p.halation_kernel.shape[0] // 2, p.mtf_kernel.shape[0] // 2
    bounds = [0, 1000, 4096, 5121, H_FULL]  # uneven shards, each at least a halo tall
    worst = 0.0
    for a, b in zip(bounds[:-1], bounds[1:]):
        d_lo, d_hi = max(a - rm, 0), min(b + rm, H_FULL)
        e_lo, e_hi = max(d_lo - rh, 0), min(d_hi + rh, H_FULL)
        E = torch.empty((3, e_hi - e_lo, W_FULL), dtype=torch.float32, device="cuda")
        if dyn:
            ctx.write_frame_params(params)  # the start of a frame: the record is reset
        ctx.stage_front(frame[e_lo:e_hi], params, 0, in_gy0=e_lo, dst=E, dst_gy0=e_lo, H_global=H_FULL, track_range=dyn)
        D = torch.empty((3, d_hi - d_lo, W_FULL), dtype=torch.float32, device="cuda")
        ctx.stage_halation(E, D, params, src_gy0=e_lo, dst_gy0=d_lo, y0=d_lo, y1=d_hi, H_global=H_FULL, range_valid=dyn)
        if dyn:
            rng = ctx.frame_exposure_range()
            assert rng["armed"] and rng["packed_pairs"] >= 0.9 * rng["pairs"] > 0, rng
        D2 = torch.empty((3, b - a, W_FULL), dtype=torch.float32, device="cuda")
        ctx.stage_mtf(D, D2, params, src_gy0=d_lo, dst_gy0=a, y0=a, y1=b, H_global=H_FULL)
        part = torch.empty((b - a, W_FULL, 3), dtype=torch.float32, device="cuda")
        ctx.stage_tail(D2, params, src_gy0=a, out_f32=part, out_gy0=a, y0=a, y1=b, H_global=H_FULL)
        if fft:
            ref = out[a:b]
            diff = (part - ref).abs()
            worst = max(worst, float((diff / ref.abs().clamp_min(1e-3)).max()))
            # complex128 on both sides: fp64 rounding noise, an fp32 ulp on a handful of pixels.  The 12-byte element on both
            # sides: each side within 1e-6 of the complex128 frame (asserted above for the whole frame: <= 2e-6), windows differ
            assert worst <= (2e-6 if dyn else 5e-7), (a, b, worst)
            assert float((diff > 0).float().mean()) <= (1e-2 if dyn else 1e-3), (a, b)
        else:
            assert torch.equal(part, out[a:b]), (a, b)
        del E, D, D2, part
    print(f"row shards vs whole frame ({mode}): worst relative difference at the 1e-3 floor {worst:.2e}")
    ctx.set_option("stencil_fft", 1)


@pytest.mark.parametrize("rows, cols", [(256, 256), (512, 256), (512, 512), (256, 1024)])
def test_the_100mp_render_does_not_depend_on_the_fft_window_shape(full, rows, cols):
    """The fixture rendered with the window shape the cost model picks (256 x 512 here); any other shape tiles the frame
    differently but computes the same correlation.  With complex128 scratch everywhere the results agree to an fp32 ulp on a
    handful of pixels; with the default complex64 scratch of the MTF passes (two fp32 roundings of the spectrum, whose values
    depend on the window) they agree to the 2 ulp of density those roundings are worth."""
    ctx, params, p, frame, out = full
    assert [c["window"] for c in ctx.stencil_stats(0)][:2] == [(256, 512)] * 2
    try:
        ctx.set_option("stencil_fft_scratch96_auto", 0)  # (complex128 for the halation whatever the window rows: the 12-byte choice
        for s32, tol, frac in ((0, 5e-7, 1e-3), (2, 4e-6, 1.0)):  #  exists for 256-row windows only and has its own test)
            ctx.set_option("stencil_fft_scratch32", s32)
            ctx.set_option("stencil_fft_window_rows", 0)
            ctx.set_option("stencil_fft_window", 0)
            base, _ = ctx.render(frame, params)
            ctx.set_option("stencil_fft_window_rows", rows)
            ctx.set_option("stencil_fft_window", cols)
            other, _ = ctx.render(frame, params)
            assert [c["window"] for c in ctx.stencil_stats(0)][:2] == [(rows, cols)] * 2
            assert [c["window"] for c in ctx.stencil_stats(1)] == [(rows, cols)] * 3
            diff = (other - base).abs()
            assert float((diff / base.abs().clamp_min(1e-3)).max()) <= tol, s32
            assert float((diff > 0).float().mean()) <= frac, s32
            del base, other, diff
    finally:
        ctx.set_option("stencil_fft_scratch96_auto", 1)
        ctx.set_option("stencil_fft_scratch32", 2)
        ctx.set_option("stencil_fft_window_rows", 0)
        ctx.set_option("stencil_fft_window", 0)
        ctx.render(frame, params)  # spectra back to the default shape for the tests that follow


def test_impulse_responses_at_full_size_reproduce_the_stencil(full):
    """Size-independent property of the FFT stencils at the full frame: isolated impulses (at window seams of both window
    widths, mid-frame and 70 px from the corners) come out as the flipped taps, to an fp32 ulp of the largest tap -- no
    leakage from the circular correlation, no seam between windows; and the stage is linear in its input."""
    ctx = full[0]
    rng = np.random.default_rng(11)
    k = np.zeros((61, 45, 1), np.float32)
    k[:, :, 0] = rng.normal(0.0, 1.0, (61, 45))  # nothing symmetric, signed: 2 745 taps -> FFT form
    ctx.set_kernel(1, k)
    src = torch.zeros((3, H_FULL, W_FULL), dtype=torch.float32, device="cuda")
    spots = [(4096, 6144, 1.0), (195, 211, 2.0), (196, 468, -1.5), (70, 70, 0.5), (H_FULL - 71, W_FULL - 71, 3.0),
             (2 * 196 + 3, 3 * 212 - 1, 1.25), (5000, 467, 1.0), (5000, 468, -2.0)]
    def row(y, c):  # the three channels get the impulse on different rows
        return y + 40 * c if y < H_FULL // 2 else y - 40 * c

    for c in range(3):
        for y, x, v in spots:
            src[c, row(y, c), x] = v * (c + 1)
    dst = torch.empty_like(src)
    ctx.stage_stencil(1, src, dst, y0=0, y1=H_FULL, H_global=H_FULL)
    assert [c["fft"] for c in ctx.stencil_stats(1)] == [1, 1, 1]
    flipped = torch.from_numpy(np.ascontiguousarray(k[::-1, ::-1, 0])).cuda()
    tol = 2e-7 * float(np.abs(k).max()) * 6.0  # |v| <= 3 x 2 for the overlapping pair below
    expect = torch.zeros_like(src)
    for c in range(3):
        for y, x, v in spots:
            yy, xx = row(y, c), x
            # out(yy + ay - i, xx + ax - j) = v k[i][j] with the anchor (30, 22): rows yy - 30 .. yy + 30, columns xx - 22 .. xx + 22
            expect[c, yy - 30:yy + 31, xx - 22:xx + 23] += v * (c + 1) * flipped
    assert float((dst - expect).abs().max()) <= tol
    # linearity: a second input and a combination of the two
    other = torch.zeros_like(src)
    other[:, 1000:1200, 3000:3300] = torch.rand((3, 200, 300), device="cuda")
    d2 = torch.empty_like(src)
    ctx.stage_stencil(1, other, d2, y0=0, y1=H_FULL, H_global=H_FULL)
    d3 = torch.empty_like(src)
    ctx.stage_stencil(1, 0.5 * src - 2.0 * other, d3, y0=0, y1=H_FULL, H_global=H_FULL)
    assert float((d3 - (0.5 * dst - 2.0 * d2)).abs().max()) <= 5e-5  # fp32 rounding of the three results (sums of ~2 700 taps of |w| ~ 1)
    del src, dst, expect, other, d2, d3
    ctx.set_kernel(1, full[2].mtf_kernel)


def test_a_constant_100mp_frame_stays_constant(full):
    """Every stencil sums to 1, so with grain off a flat frame must come out flat, at the value the pointwise chain gives."""
    ctx, params, p, frame, out = full
    flat = torch.full((H_FULL, W_FULL, 3), 0.18, dtype=torch.float32, device="cuda")
    q = ctx.make_params(matrix=True, halation=True, mtf=True, grain=False)
    res, _ = ctx.render(flat, q)
    p2 = oracle_inputs(*stocks()[:2], SCALE, grain=0, halation=False, mtf=False)
    expect = st.render(np.full((2, 2, 3), 0.18, np.float32), p2)[0, 0]
    lo, hi = res.amin(dim=(0, 1)).cpu().numpy(), res.amax(dim=(0, 1)).cpu().numpy()
    assert np.all(hi - lo <= 2e-6), (lo, hi)
    np.testing.assert_allclose(lo, expect, rtol=0, atol=3e-6)


def test_bw_stock_with_unsharp_mask_and_mono_grain_at_full_size():
    """The other branches at 100 MP: halation on all three layers (bw stock: no identity plane), MTF with the unsharp-mask
    term (negative taps), monochrome grain."""
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    _, prt, bw = stocks()
    p = oracle_inputs(bw, None, SCALE, seed=SEED + 1, grain=1, sharpening_strength=0.6, halation_green_factor=0.4)
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        frame = synthetic_frame_device(H_FULL, W_FULL, seed=9)
        out, _ = ctx.render(frame, params)
        n, y0, x0 = 64, 5000, 9000
        halo = 42 + 17 + 4
        ya, yb, xa, xb = y0 - halo, y0 + n + halo, x0 - halo, x0 + n + halo
        crop = frame[ya:yb, xa:xb].cpu().numpy()
        x = st.apply_2d_lut(st.apply_matrix3x3(crop, p.matrix), p.lut_2d)
        x = st.halation(x, p.halation_kernel)
        x = st.multi_channel_interp(st.log_clip(x), p.lut_1d)
        x = st.film_sharpness(x, p.mtf_kernel)
        x = st.apply_grain(x, p.grain_lut, p.grain_kernel, p.seed, True, row0=ya, H_global=H_FULL, col0=xa, W_global=W_FULL)
        ref = st.apply_lut_tetrahedral(x, p.lut_3d, 0.25)[halo:halo + n, halo:halo + n]
        got = out[y0:y0 + n, x0:x0 + n].cpu().numpy()
        err = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3))
        assert err <= 1e-5, err
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------- the other BASELINE shapes
def _oracle_window(frame, p, H, W, y0, x0, n, halo, *, mono=False):
    """The oracle on window + halo of a device frame (hash at global coordinates); returns the n x n window."""
    ya, yb = max(y0 - halo, 0), min(y0 + n + halo, H)
    xa, xb = max(x0 - halo, 0), min(x0 + n + halo, W)
    crop = frame[ya:yb, xa:xb].cpu().numpy()
    x = st.apply_2d_lut(st.apply_matrix3x3(crop, p.matrix) if p.matrix is not None else crop, p.lut_2d)
    if p.halation_kernel is not None:
        x = st.halation(x, p.halation_kernel)
    x = st.multi_channel_interp(st.log_clip(x), p.lut_1d)
    if p.mtf_kernel is not None:
        x = st.film_sharpness(x, p.mtf_kernel)
    if p.grain_lut is not None:
        x = st.apply_grain(x, p.grain_lut, p.grain_kernel, p.seed, mono, row0=ya, H_global=H, col0=xa, W_global=W)
    return st.apply_lut_tetrahedral(x, p.lut_3d, 0.25)[y0 - ya:y0 - ya + n, x0 - xa:x0 - xa + n]


def _check_windows(out, u8, frame, p, H, W, spots, n, halo):
    bad = total = 0
    for y0, x0 in spots:
        ref = _oracle_window(frame, p, H, W, y0, x0, n, halo)
        got = out[y0:y0 + n, x0:x0 + n].cpu().numpy()
        err = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3))
        assert err <= 1e-5, (y0, x0, err)
        d = np.abs(u8[y0:y0 + n, x0:x0 + n].cpu().numpy().astype(int) - st.to_uint8(ref).astype(int))
        assert d.max() <= 1, (y0, x0)
        bad += int((d > 0).sum())
        total += d.size
    assert bad / total <= 1e-4, (bad, total)


def test_cfg2_24mp_lut_only_windows_match_the_oracle():
    """BASELINE config 2 at its full size: 6000 x 4000, negative + print LUTs, effects off (one fused pointwise kernel)."""
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    W, H = 6000, 4000
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, W / 36.0, halation=False, mtf=False, grain=0)
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        frame = synthetic_frame_device(H, W, seed=17)
        out, u8 = ctx.render(frame, params, want_f32=True, want_u8=True)
        _check_windows(out, u8, frame, p, H, W, [(0, 0), (1999, 2873), (H - 256, W - 256), (77, W - 300)], 256, 0)
    finally:
        ctx.close()


def test_cfg3_45mp_full_pipeline_windows_match_the_oracle():
    """BASELINE config 3 at its full size: 8256 x 5504, 59 / 23 / 7-tap stencils (halation and, since its scratch is complex64, MTF by FFT rather than in the unrolled direct
    form), grain on."""
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    W, H = 8256, 5504
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, W / 36.0, seed=SEED)
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        frame = synthetic_frame_device(H, W, seed=19)
        out, u8 = ctx.render(frame, params, want_f32=True, want_u8=True)
        halo = p.halation_kernel.shape[0] // 2 + p.mtf_kernel.shape[0] // 2 + p.grain_kernel.shape[0] // 2
        _check_windows(out, u8, frame, p, H, W, [(0, 0), (2750, 4100), (H - 128, W - 128), (200, W - 250), (H - 140, 3)], 128, halo)
    finally:
        ctx.close()


def test_cfg5_24mp_full_pipeline_windows_match_the_oracle():
    """BASELINE config 5's frame at its own size with every effect on: 6000 x 4000, 43 / 17 / 5-tap stencils --
    the halation's 41 x 41 box by FFT (ragged last window row and column on this frame), the 17 x 17 MTF in the unrolled direct
    form --, once through r2f_render on the context and once through the batch surface the reference's export loop calls,
    HipProcessor.extract_image_data_cpu -> process_preloaded (gpu_processor.py:715-783, 1643-1693): uint8, <= 1 LSB on <= 1e-4."""
    from raw2film_amd import HipProcessor
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    W, H = 6000, 4000
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, W / 36.0, seed=SEED)
    assert p.halation_kernel.shape[0] == 43 and p.mtf_kernel.shape[0] == 17
    halo = p.halation_kernel.shape[0] // 2 + p.mtf_kernel.shape[0] // 2 + p.grain_kernel.shape[0] // 2
    frame = synthetic_frame_device(H, W, seed=23)
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        out, u8 = ctx.render(frame, params, want_f32=True, want_u8=True)
        stats = ctx.stencil_stats(0)
        assert [c["fft"] for c in stats] == [1, 1, 0] and all(c["fft"] == 0 and c["unrolled"] == 8 for c in ctx.stencil_stats(1))
        ny, nx = stats[0]["window"]
        vy, vx = ny - 41 + 1, (nx - 41 + 1) & ~3
        ly, lx = (H - 1) // vy * vy, (W - 1) // vx * vx  # first row / column of the last (ragged) window row / column
        assert H - ly < vy and W - lx < vx
        spots = [(0, 0), (1999, 2873), (H - 128, W - 128), (ly - 64, lx - 64), (ly - 64, 77), (300, lx - 64)]
        _check_windows(out, u8, frame, p, H, W, spots, 128, halo)
    finally:
        ctx.close()
    # the batch surface: the same frame as a decoded XYZ host array (no 3 x 3), phase 1 + phase 2
    host = frame.cpu().numpy()
    q = oracle_inputs(neg, prt, W / 36.0, seed=SEED, matrix=False)
    proc = HipProcessor(device=0)
    try:
        payload = proc.extract_image_data_cpu(host, frame_width=36, frame_height=24)
        assert tuple(payload["pipeline_resolution"]) == (W, H)
        got = proc.process_preloaded(payload, neg, 6, 0.4, print_film=prt, frame_width=36, frame_height=24, halation_green_factor=0.3,
                                     exp_kelvin=6000, color_masking=1.0, seed=SEED)
        assert got.dtype == np.uint8 and got.shape == (H, W, 3)
        bad = total = 0
        for y0, x0 in spots:
            ref = st.to_uint8(_oracle_window(frame, q, H, W, y0, x0, 128, halo))
            d = np.abs(got[y0:y0 + 128, x0:x0 + 128].astype(int) - ref.astype(int))
            assert d.max() <= 1, (y0, x0)
            bad += int((d > 0).sum())
            total += d.size
        assert bad / total <= 1e-4, (bad, total)
    finally:
        proc.close()


def test_cfg4_100mp_uint8_windows_match_the_oracle(full):
    """The uint8 output of the 100 MP render (what export writes): <= 1 LSB on <= 1e-4 of the samples of the windows."""
    ctx, params, p, frame, out = full
    _, u8 = ctx.render(frame, params, want_f32=False, want_u8=True)
    halo = 42 + 17 + 4
    _check_windows(out, u8, frame, p, H_FULL, W_FULL, [(0, 0), (4000, 6000), (H_FULL - 96, W_FULL - 96)], 96, halo)


def test_the_100mp_host_frame_streamed_in_row_bands_is_the_whole_frame_render():
    """HipProcessor.process(host array, cache=False) with pinned result buffers at BASELINE's full size: the frame streams through the
    pipeline in 16 bands of 512 rows while it arrives (3 halation window rows of 172 per band, anchored at the band's first row) --
    against the same call with upload, render and download one after the other (r2f_render, whose windows the tests above hold
    against the oracle): uint8, at most one step apart on at most 1e-4 of the samples; and the two pinned buffers come back in turn."""
    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.synthetic import synthetic_frame_device

    stocks_ = filmstock.builtin_stocks()
    neg, prt = stocks_["Kodak Portra 400"], stocks_["Kodak 2383"]
    host = torch.empty((H_FULL, W_FULL, 3), dtype=torch.float32, pin_memory=True)
    host.copy_(synthetic_frame_device(H_FULL, W_FULL, seed=23))
    img = host.numpy()
    kw = dict(print_film=prt, lens_correction=False, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0,
              halation_green_factor=0.3, matrix=REC709_TO_XYZ, seed=SEED, cache=False)
    proc = HipProcessor(device=0, result_buffers=2)
    try:
        assert proc.stream_bands == 16
        got = proc.process(img, neg, 6, 0.4, **kw)
        proc.stream_bands = 0
        want = proc.process(img, neg, 6, 0.4, **kw)
        assert got.shape == want.shape == (H_FULL, W_FULL, 3) and got.dtype == np.uint8 and got.ctypes.data != want.ctypes.data
        bad = 0
        for r0 in range(0, H_FULL, 1024):  # (int16 differences of 0.3 G samples, a slab at a time)
            d = np.abs(got[r0:r0 + 1024].astype(np.int16) - want[r0:r0 + 1024].astype(np.int16))
            assert int(d.max()) <= 1
            bad += int(np.count_nonzero(d))
        assert bad <= 1e-4 * got.size, bad
        assert got.std() > 10  # (a picture, not a constant)
    finally:
        proc.close()


def test_a_frame_with_more_than_2_31_elements_per_buffer():
    """Maximum sizes: 32768 x 21888 x 3 = 2.15e9 floats (8.6 GB) per interleaved buffer -- element indices beyond int32, byte offsets
    beyond 2^33 -- through r2f_render with the headline's stencils; windows at the far end of the buffers against the oracle."""
    from raw2film_amd.context import HipContext
    from raw2film_amd.synthetic import synthetic_frame_device
    from test_gpu_parity import setup_ctx

    if torch.cuda.get_device_properties(0).total_memory < 48 * 2**30:
        pytest.skip("needs ~33 GiB of device memory")
    W, H = 32768, 21888
    assert W * H * 3 > 2**31
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, SCALE, seed=SEED)
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        for y in range(0, H, 2048):  # (the generator's temporaries are several times its output)
            frame[y:y + 2048] = synthetic_frame_device(min(2048, H - y), W, seed=7 + y)
        out, _ = ctx.render(frame, params)
        torch.cuda.synchronize()
        n, halo = 96, 42 + 17 + 4
        for y0, x0 in [(H - 96, W - 96), (21846, 100), (0, 0)]:  # (21846, 100): the first rows past element 2^31
            ya, yb = max(y0 - halo, 0), min(y0 + n + halo, H)
            xa, xb = max(x0 - halo, 0), min(x0 + n + halo, W)
            crop = frame[ya:yb, xa:xb].cpu().numpy()
            x = st.apply_2d_lut(st.apply_matrix3x3(crop, p.matrix), p.lut_2d)
            x = st.halation(x, p.halation_kernel)
            x = st.multi_channel_interp(st.log_clip(x), p.lut_1d)
            x = st.film_sharpness(x, p.mtf_kernel)
            x = st.apply_grain(x, p.grain_lut, p.grain_kernel, p.seed, False, row0=ya, H_global=H, col0=xa, W_global=W)
            ref = st.apply_lut_tetrahedral(x, p.lut_3d, 0.25)[y0 - ya:y0 - ya + n, x0 - xa:x0 - xa + n]
            got = out[y0:y0 + n, x0:x0 + n].cpu().numpy()
            err = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3))
            assert err <= 1e-5, (y0, x0, err)
        del frame, out
    finally:
        ctx.close()
        torch.cuda.empty_cache()


def test_operations_either_side_of_the_path_beyond_2_31_elements():
    """The same maximum size for the steps around the path (histogram, both uint8 scalings, the uint16 hand-off, INTER_AREA, the
    affine warp, chroma NR): a far corner of the 717 MP result equals the same operation on a crop."""
    from raw2film_amd.context import HipContext

    if torch.cuda.get_device_properties(0).total_memory < 48 * 2**30:
        pytest.skip("needs ~24 GiB of device memory")
    W, H = 32768, 21888
    ctx = HipContext(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    try:
        u8 = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)  # > 2^31 bytes
        counts = ctx.histogram_counts(u8)
        want = torch.stack([torch.bincount(u8[..., c].reshape(-1).to(torch.int64), minlength=256) for c in range(3)])
        assert torch.equal(counts.to(torch.int64) & 0xFFFFFFFF, want & 0xFFFFFFFF)
        del want
        small = u8[: H // 2, : W // 2].contiguous()
        up = ctx.resize_lanczos4_u8(small, H, W)  # factor 2: a crop at even offsets, away from the borders, scales to the same pixels
        y0, x0, n = H // 2 - 600, W // 2 - 700, 256
        cu = ctx.resize_lanczos4_u8(small[y0 - 16:y0 + n + 16, x0 - 16:x0 + n + 16].contiguous(), 2 * (n + 32), 2 * (n + 32))
        assert torch.equal(up[2 * y0:2 * (y0 + n), 2 * x0:2 * (x0 + n)], cu[32:32 + 2 * n, 32:32 + 2 * n])
        del up, cu, small
        dn = ctx.resize_area_u8(u8, H // 4, W // 4)
        y0, x0, n = H // 4 - 300, W // 4 - 300, 200
        cd = ctx.resize_area_u8(u8[4 * y0:4 * (y0 + n), 4 * x0:4 * (x0 + n)].contiguous(), n, n)
        assert torch.equal(dn[y0:y0 + n, x0:x0 + n], cd)
        del dn, cd, u8

        u16 = torch.randint(0, 65536, (H, W, 3), dtype=torch.int32, device="cuda", generator=g).to(torch.uint16)
        f = ctx.decode_u16(u16, 1.7)  # > 2^31 elements
        for ya, xa in ((H - 64, W - 64), (21845, 0)):
            src = u16[ya:ya + 64, xa:xa + 64].cpu().numpy().astype(np.float32)
            ref = np.minimum(src / np.float32(65535.0) * np.float32(1.7), np.float32(65504.0))
            assert np.array_equal(f[ya:ya + 64, xa:xa + 64].cpu().numpy(), ref)
        del u16
        a = ctx.resize_area(f, H // 4, W // 4)
        y0, x0, n = H // 4 - 300, W // 4 - 300, 200
        ca = ctx.resize_area(f[4 * y0:4 * (y0 + n), 4 * x0:4 * (x0 + n)].contiguous(), n, n)
        assert torch.allclose(a[:, y0:y0 + n, x0:x0 + n], ca, rtol=1e-6, atol=1e-7)
        del a, ca
        m = [1.0, 0.0, 3.25, 0.0, 1.0, 2.5]  # dst -> src: a sub-pixel shift
        wy, wx, n = H - 400, W - 500, 256
        wa = ctx.warp_affine(f, m, window=(wy, wx, n, n))
        wb = ctx.warp_affine(f[wy - 8:wy + n + 8, wx - 8:wx + n + 8].contiguous(), m, window=(8, 8, n, n))
        assert torch.allclose(wa, wb, rtol=1e-6, atol=1e-7)
        del wa, wb
        nr = ctx.chroma_nr(f, 2)
        y0, x0, n = H - 700, W - 800, 256
        nc = ctx.chroma_nr(f[y0 - 64:y0 + n + 64, x0 - 64:x0 + n + 64].contiguous(), 2)
        assert torch.allclose(nr[:, y0:y0 + n, x0:x0 + n], nc[:, 64:64 + n, 64:64 + n], rtol=2e-6, atol=1e-7)
        del nr, nc, f
    finally:
        ctx.close()
        torch.cuda.empty_cache()
