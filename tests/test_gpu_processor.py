"""The HipProcessor operator surface on a real GPU: process / extract_image_data_cpu / process_preloaded /
process_array against the oracle, with the reference's keyword names."""

import numpy as np
import pytest

from oracle import stages as st

from helpers import SEED, oracle_inputs, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def proc():
    from raw2film_amd import HipProcessor

    p = HipProcessor(cameras={}, lenses={}, device=0)
    yield p
    p.close()


def _xyz(H, W, seed=41):
    return st.apply_matrix3x3(synthetic_frame(H, W, seed=seed), st.REC709_TO_XYZ)


def _u8_close(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    return d.max() <= 1 and (d > 0).mean() <= 1e-4


def test_process_returns_uint8_like_the_cpu_processor(proc):
    neg, prt, _ = stocks()
    H, W, fw = 120, 180, 1.0
    img = _xyz(H, W)
    kw = dict(print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000,
              color_masking=1.0, seed=SEED, profile="Default", film_format="135")  # GUI extras are swallowed
    out = proc.process(img, neg, 6, 0.4, **kw)
    assert out.dtype == np.uint8 and out.shape == (H, W, 3)
    p = oracle_inputs(neg, prt, max(H, W) / fw, matrix=False)
    assert _u8_close(out, st.to_uint8(st.render(img, p)))
    # two-phase batch API gives the same bytes
    payload = proc.extract_image_data_cpu(img, frame_width=fw, frame_height=fw * H / W)
    assert payload["image_array"].shape == (H, W, 4)
    out2 = proc.process_preloaded(payload, neg, 6, 0.4, **kw)
    np.testing.assert_array_equal(out, out2)


def test_process_array_float_and_colorspace(proc):
    neg, prt, _ = stocks()
    H, W, fw = 96, 128, 0.6
    rgb = synthetic_frame(H, W, seed=42)
    kw = dict(print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000,
              color_masking=1.0, seed=SEED)
    out = proc.process_array(rgb, neg, 6, 0.4, colorspace="linear-rec709", return_float=True, **kw)
    ref = st.render(rgb, oracle_inputs(neg, prt, max(H, W) / fw))
    assert out.dtype == np.float32
    assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 0.1)) <= 1e-5
    dev = proc.process_array(torch.from_numpy(rgb).cuda(), neg, 6, 0.4, colorspace="linear-rec709", return_float=True,
                             output="device", **kw)
    assert dev.is_cuda
    np.testing.assert_array_equal(dev.cpu().numpy(), out)


def test_stage_gates_and_negative_only(proc):
    neg, _, _ = stocks()
    H, W = 64, 96
    img = _xyz(H, W, seed=43)
    out = proc.process(img, neg, 6, 0.4, print_film=None, halation=False, sharpness=False, grain=0, exp_kelvin=6000,
                       color_masking=1.0)
    p = oracle_inputs(neg, None, max(H, W) / 36, halation=False, mtf=False, grain=0, matrix=False)
    assert _u8_close(out, st.to_uint8(st.render(img, p)))
    mono = proc.process(img, neg, 6, 0.4, print_film=None, halation=False, sharpness=False, grain=1, seed=7,
                        frame_width=0.375, frame_height=0.25, exp_kelvin=6000, color_masking=1.0)
    pm = oracle_inputs(neg, None, max(H, W) / 0.375, halation=False, mtf=False, grain=1, seed=7, matrix=False)
    assert _u8_close(mono, st.to_uint8(st.render(img, pm)))


def test_crop_zoom_turn_and_canvas(proc):
    from raw2film_amd import geometry

    neg, prt, _ = stocks()
    img = _xyz(100, 180, seed=44)
    kw = dict(print_film=prt, halation=False, sharpness=False, grain=0, exp_kelvin=6000, color_masking=1.0,
              frame_width=36, frame_height=24)
    pre = np.ascontiguousarray(geometry.crop_to_frame(img, 36, 24, 1.5, 1, False))
    p = oracle_inputs(neg, prt, max(pre.shape[:2]) / 36, halation=False, mtf=False, grain=0, matrix=False)
    framed = geometry.add_canvas(st.to_uint8(st.render(pre, p)), "Uniform white", 1.25)
    # the two-phase (GpuProcessor-style) API keeps the canvas at its size ...
    payload = proc.extract_image_data_cpu(img, zoom=1.5, rotate_times=1, canvas_mode="Uniform white", canvas_scale=1.25,
                                          frame_width=36, frame_height=24)
    out = proc.process_preloaded(payload, neg, 6, 0.4, canvas_mode="Uniform white", canvas_scale=1.25, **kw)
    assert out.shape == framed.shape and out[0, 0, 0] == 255
    assert _u8_close(out, framed)
    # ... process() is CpuProcessor.process: the framed result is scaled back INTO the frame's own resolution
    # (cpu_processor.py:119-122, 411-412: cv.resize INTER_AREA of the uint8 canvas)
    from oracle import post

    out = proc.process(img, neg, 6, 0.4, zoom=1.5, rotate_times=1, canvas_mode="Uniform white", canvas_scale=1.25, **kw)
    f = min(pre.shape[0] / framed.shape[0], pre.shape[1] / framed.shape[1])
    ref = post.resize_area_u8(framed, round(framed.shape[0] * f), round(framed.shape[1] * f))
    assert out.shape == ref.shape and out[0, 0, 0] == 255
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 2e-3  # a 1-LSB difference before the shrink survives it in a few samples


@pytest.mark.parametrize("deg,zoom,k,flip", [(2.5, 1.0, 0, False), (-11.0, 1.4, 1, False), (44.0, 1.0, 3, True), (0.3, 1.0, 2, False)])
def test_free_rotation_through_the_processor(proc, deg, zoom, k, flip):
    from raw2film_amd import geometry

    neg, prt, _ = stocks()
    img = _xyz(150, 210, seed=45)
    kw = dict(print_film=prt, halation=False, sharpness=False, grain=0, exp_kelvin=6000, color_masking=1.0,
              frame_width=36, frame_height=24, max_scale=None)
    out = proc.process(img, neg, 6, 0.4, rotation=deg, zoom=zoom, rotate_times=k, flip=flip, **kw)
    # raw_conversion.crop_rotate_zoom on the oracle side
    r0, c0, nr, nc = geometry.crop_box(150, 210, 1, 1.5, flip)
    pre = st.rotate(np.ascontiguousarray(img[r0:r0 + nr, c0:c0 + nc]), deg)
    z = geometry.crop_box(pre.shape[0], pre.shape[1], zoom, 1.5, False)
    pre = np.ascontiguousarray(np.rot90(pre[z[0]:z[0] + z[2], z[1]:z[1] + z[3]], k))
    p = oracle_inputs(neg, prt, max(pre.shape[:2]) / 36, halation=False, mtf=False, grain=0, matrix=False)
    ref = st.to_uint8(st.render(pre, p))
    assert out.shape == ref.shape
    assert _u8_close(out, ref)


def test_warp_affine_matches_oracle_on_every_layout(proc):
    from raw2film_amd import geometry

    rng = np.random.default_rng(8)
    img = rng.uniform(0, 4, (97, 131, 3)).astype(np.float32)
    for deg in (5.0, -33.0, 90.0, 180.0):
        m, win = geometry.rotation_plan(97, 131, deg)
        ref_full = st.warp_affine_linear(img, m)
        ref_win = st.warp_affine_linear(img, m, win[2:], win[:2])
        assert np.array_equal(ref_win, ref_full[win[0]:win[0] + win[2], win[1]:win[1] + win[3]])
        for layout in ("hwc3", "hwc4", "chw"):
            a = img if layout == "hwc3" else np.concatenate([img, np.ones((97, 131, 1), np.float32)], -1) if layout == "hwc4" \
                else np.ascontiguousarray(img.transpose(2, 0, 1))
            t = torch.from_numpy(a).cuda()
            got = proc.ctx.warp_affine(t, m, layout=layout).cpu().numpy().transpose(1, 2, 0)
            # float32 coordinates: a last-bit difference in sx moves the sample by 2^-17 px at x ~ 100 -> ~1e-5 of the local contrast
            assert np.abs(got - ref_full).max() <= 2e-5 * 4
            gotw = proc.ctx.warp_affine(t, m, win, layout=layout).cpu().numpy().transpose(1, 2, 0)
            assert np.array_equal(gotw, got[win[0]:win[0] + win[2], win[1]:win[1] + win[3]])


@pytest.mark.parametrize("shape,out", [((20, 30), (50, 75)), ((97, 131), (200, 270)), ((5, 7), (64, 90)), ((40, 60), (41, 61)),
                                       ((64, 96), (64, 96))])
def test_lanczos4_upscale_bit_exact(proc, shape, out):
    img = np.random.default_rng(shape[0]).integers(0, 256, shape + (3,)).astype(np.uint8)
    got = proc.ctx.resize_lanczos4_u8(torch.from_numpy(img).cuda(), *out).cpu().numpy()
    assert np.array_equal(got, st.resize_lanczos4_u8(img, *out))


def test_max_scale_round_trip_through_the_processor(proc):
    """A small-gauge frame (finer than max_scale px/mm): INTER_AREA down to max_scale, the path, LANCZOS4 back up."""
    from raw2film_amd import geometry

    neg, prt, _ = stocks()
    img = _xyz(160, 240, seed=46)
    kw = dict(print_film=prt, halation=True, sharpness=True, grain=0, exp_kelvin=6000, color_masking=1.0,
              frame_width=5.79, frame_height=3.86, max_scale=20.0)
    out = proc.process(img, neg, 6, 0.4, **kw)
    pre = geometry.crop_to_frame(img, 5.79, 3.86, 1.0, 0, False)
    h, w = pre.shape[:2]
    f = 20.0 / (max(h, w) / 5.79)
    res = [round(h * f), round(w * f)]
    g = min(res[0] / h, res[1] / w)
    small = np.stack([st.resize_area(np.ascontiguousarray(pre[..., c]), round(h * g), round(w * g)) for c in range(3)], axis=-1)
    p = oracle_inputs(neg, prt, max(small.shape[:2]) / 5.79, halation=True, mtf=True, grain=0, matrix=False,
                      halation_green_factor=0.4)  # process()'s default (cpu_processor.py:306), not the GUI's 0.3
    ref = st.resolution_scaling_u8_up(st.to_uint8(st.render(small, p)), (h, w))
    assert out.shape == ref.shape and out.shape[0] > small.shape[0]
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 3 and (d > 0).mean() <= 5e-3  # a 1-LSB truncation flip before the filter spreads over 8 x 8 taps


@pytest.mark.parametrize("mode", ["Uniform white", "Fixed black"])
def test_max_scale_below_the_frame_scale_with_a_canvas(proc, mode):
    """a small-gauge frame rendered at max_scale AND framed by a canvas.  process(): CpuProcessor's order --
    pipeline at max_scale, canvas pasted on the pipeline-size frame (cpu_processor.py:409), then resolution_scaling of the framed
    frame back to the requested size (:411-412, LANCZOS4).  process_preloaded(dst_texture=...): the blit transform comes from
    the payload's output / canvas / pipeline resolutions, which follow gpu_processor.py:764-771 (canvas laid out for the
    UN-shrunk output size; pinned by tests/golden/payload_geometry.npz) -- compared with the oracle's blit of the same frame."""
    from oracle import post
    from raw2film_amd import geometry

    neg, prt, _ = stocks()
    img = _xyz(160, 240, seed=52)
    canvas = dict(canvas_mode=mode, canvas_scale=1.15, canvas_ratio=1.25)
    kw = dict(print_film=prt, halation=True, sharpness=True, grain=0, exp_kelvin=6000, color_masking=1.0,
              frame_width=5.79, frame_height=3.86, max_scale=20.0, **canvas)
    out = proc.process(img, neg, 6, 0.4, **kw)
    pre = geometry.crop_to_frame(img, 5.79, 3.86, 1.0, 0, False)
    h, w = pre.shape[:2]
    f = 20.0 / (max(h, w) / 5.79)
    res = [round(h * f), round(w * f)]
    g = min(res[0] / h, res[1] / w)
    small = np.stack([st.resize_area(np.ascontiguousarray(pre[..., c]), round(h * g), round(w * g)) for c in range(3)], axis=-1)
    p = oracle_inputs(neg, prt, max(small.shape[:2]) / 5.79, halation=True, mtf=True, grain=0, matrix=False, halation_green_factor=0.4)
    rendered = st.render(small, p)
    framed = geometry.add_canvas(st.to_uint8(rendered), **canvas)  # (pinned against effects.add_canvas: tests/golden/add_canvas.npz)
    ref = st.resolution_scaling_u8_up(framed, (h, w))
    assert out.shape == ref.shape and framed.shape[0] > small.shape[0]
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 3 and (d > 0).mean() <= 5e-3
    # the GPU processor's destination branch
    payload = proc.extract_image_data_cpu(img, frame_width=5.79, frame_height=3.86, max_scale=20.0, **canvas)
    ow, oh = payload["output_resolution"]
    assert (oh, ow) == (h, w) and payload["pipeline_resolution"] == (small.shape[1], small.shape[0])
    want_canvas, color, _ = geometry.canvas_layout((h, w), mode, 1.15, 1.25)  # the un-shrunk output size
    assert payload["canvas_resolution"] == (want_canvas[1], want_canvas[0])
    dst = torch.zeros((300, 400, 4), dtype=torch.uint8, device="cuda")
    settings = {k: v for k, v in kw.items() if k not in ("max_scale",)}
    assert proc.process_preloaded(payload, neg, 6, 0.4, dst_texture=dst, **settings) is None
    t = geometry.blit_transform((small.shape[1], small.shape[0]), (400, 300), pipeline_resolution=payload["pipeline_resolution"],
                                output_resolution=payload["output_resolution"], canvas_resolution=payload["canvas_resolution"],
                                canvas_color=color)
    want = post.blit_rgba8(rendered, 300, 400, t)
    got = dst.cpu().numpy()
    np.testing.assert_array_equal(got[..., 3], want[..., 3])
    dd = np.abs(got.astype(int) - want.astype(int))
    assert dd.max() <= 1 and (dd > 0).mean() <= 2e-3
    # the image occupies output / canvas of the canvas area on both axes (the bug this guards against: a canvas laid out for the
    # pipeline size next to an output size in final pixels shrank the image inside its frame)
    inside = (got[..., :3] != np.array(color, dtype=np.uint8)).any(axis=2) & (got[..., 3] == 255)
    rows, cols = np.where(inside.any(axis=1))[0], np.where(inside.any(axis=0))[0]
    area = got[..., 3] == 255
    arows, acols = np.where(area.any(axis=1))[0], np.where(area.any(axis=0))[0]
    assert abs((rows[-1] - rows[0] + 1) / (arows[-1] - arows[0] + 1) - h / want_canvas[0]) < 0.02
    assert abs((cols[-1] - cols[0] + 1) / (acols[-1] - acols[0] + 1) - w / want_canvas[1]) < 0.02


def test_uploads_happen_only_on_change(proc):
    neg, prt, _ = stocks()
    img = _xyz(48, 64, seed=45)
    kw = dict(print_film=prt, exp_kelvin=6000, color_masking=1.0, frame_width=0.5, frame_height=0.375, seed=1)
    proc.process(img, neg, 6, 0.4, **kw)
    n = proc.uploads
    proc.process(img, neg, 6, 0.4, **kw)
    assert proc.uploads == n
    proc.process(img, neg, 6, 0.4, **{**kw, "halation_size": 1.5})
    assert proc.uploads == n + 1


def test_highlight_burn_through_the_processor(proc):
    neg, prt, _ = stocks()
    H, W, fw = 120, 180, 1.5
    img = _xyz(H, W, seed=46)
    img[30:70, 50:120] *= 10.0
    kw = dict(print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000,
              color_masking=1.0, seed=SEED)
    plain = proc.process(img, neg, 6, 0.4, **kw)
    burnt = proc.process(img, neg, 6, 0.4, highlight_burn=0.8, burn_scale=30, **kw)
    p = oracle_inputs(neg, prt, max(H, W) / fw, matrix=False)
    p.highlight_burn, p.burn_scale, p.d_ref = 0.8, 30.0, float(neg.d_ref[1])
    assert _u8_close(burnt, st.to_uint8(st.render(img, p)))
    assert np.abs(burnt.astype(int) - plain.astype(int)).max() > 5


def test_chroma_nr_through_the_processor(proc):
    neg, prt, _ = stocks()
    H, W = 80, 120
    img = _xyz(H, W, seed=47)
    kw = dict(print_film=prt, halation=False, sharpness=False, grain=0, exp_kelvin=6000, color_masking=1.0)
    out = proc.process(img, neg, 6, 0.4, chroma_nr=3, **kw)
    p = oracle_inputs(neg, prt, max(H, W) / 36, halation=False, mtf=False, grain=0, matrix=False)
    ref = st.to_uint8(st.render(st.chroma_nr_filter(img, 3), p))
    assert _u8_close(out, ref)
    assert np.abs(out.astype(int) - proc.process(img, neg, 6, 0.4, **kw).astype(int)).max() > 3


def test_preview_resolution_downscale(proc):
    neg, prt, _ = stocks()
    img = _xyz(200, 300, seed=48)
    kw = dict(print_film=prt, exp_kelvin=6000, color_masking=1.0, seed=SEED, halation_green_factor=0.3)
    out = proc.process(img, neg, 6, 0.4, resolution=(80, 150), **kw)
    assert out.shape == (80, 120, 3)
    small = np.stack([st.resize_area(img[..., c], 80, 120) for c in range(3)], axis=-1)
    p = oracle_inputs(neg, prt, 120 / 36, matrix=False)
    assert _u8_close(out, st.to_uint8(st.render(small, p)))


# ------------------------------------------------------------------------------------- after the path
@pytest.mark.parametrize("shape,out", [((120, 180), (60, 90)), ((120, 180), (40, 60)), ((121, 183), (97, 150)), ((300, 200), (299, 199)),
                                       ((64, 64), (1, 1)), ((90, 120), (30, 120)), ((200, 310), (77, 113))])
def test_uint8_area_shrink_matches_the_oracle_bit_for_bit(proc, shape, out):
    """cv.resize(uint8, INTER_AREA) as the CPU processor applies it to the finished frame: integer factors (2 x 2 and others),
    fractional ones, one axis unchanged."""
    from oracle import post

    rng = np.random.default_rng(shape[0] + out[1])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    got = proc.ctx.resize_area_u8(torch.from_numpy(img).cuda(), *out).cpu().numpy()
    np.testing.assert_array_equal(got, post.resize_area_u8(img, *out))


@pytest.mark.parametrize("canvas", [None, (700, 520)])
@pytest.mark.parametrize("dst", [(200, 300), (333, 250)])
def test_preview_blit_into_a_destination_texture(proc, dst, canvas):
    """copy_to_int.wgsl: bilinear letterbox of the float frame into an RGBA8 tensor, canvas colour, transparent outside."""
    from oracle import post
    from raw2film_amd import geometry

    rng = np.random.default_rng(3)
    H, W = 400, 600
    img = rng.uniform(-0.1, 1.1, (H, W, 3)).astype(np.float32)
    t = geometry.blit_transform((W, H), (dst[1], dst[0]), pipeline_resolution=(W, H), output_resolution=(W, H),
                                canvas_resolution=canvas, canvas_color=(128, 128, 128))
    tex = torch.zeros(dst + (4,), dtype=torch.uint8, device="cuda")
    proc.ctx.blit_rgba8(torch.from_numpy(img).cuda(), tex, t)
    got, ref = tex.cpu().numpy(), post.blit_rgba8(img, dst[0], dst[1], t)
    np.testing.assert_array_equal(got[..., 3], ref[..., 3])  # inside / canvas / transparent regions agree exactly
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 1e-3  # fused vs separate multiply-add in the lerp


def test_histogram_texture_from_device_counts(proc):
    """histogram.wgsl passes 2 + 3 and scale_texture.wgsl on the device counts."""
    from oracle import post
    from raw2film_amd import histogram

    rng = np.random.default_rng(8)
    img = np.clip(rng.normal(120, 40, (200, 300, 3)), 0, 255).astype(np.uint8)
    img[:50] = 255
    counts = proc.ctx.histogram_counts(torch.from_numpy(img).cuda())
    target = torch.zeros((90, 333, 4), dtype=torch.uint8, device="cuda")
    image = proc.ctx.histogram_render(counts, histogram.MIX_TABLE, 256, target=target).cpu().numpy()
    ref_img, ref_tgt, _ = post.histogram_render(counts.cpu().numpy(), histogram.MIX_TABLE, 256, (90, 333))
    # logf on the device and NumPy's float32 log may differ in the last place: a bar may be one pixel taller or shorter
    assert (image != ref_img).any(axis=2).sum(axis=0).max() <= 1
    assert (image != ref_img).any(axis=2).any(axis=0).mean() <= 0.05
    assert (target.cpu().numpy() != ref_tgt).any(axis=2).mean() <= 0.01


def test_process_with_destination_textures_like_the_gpu_processor(proc):
    neg, prt, _ = stocks()
    H, W = 120, 180
    img = _xyz(H, W, seed=47)
    kw = dict(print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0, seed=SEED,
              halation=False, sharpness=False, grain=0)
    dst = torch.zeros((100, 200, 4), dtype=torch.uint8, device="cuda")
    hist = torch.zeros((64, 128, 4), dtype=torch.uint8, device="cuda")
    assert proc.process(img, neg, 6, 0.4, dst_texture=dst, histogram_texture=hist, **kw) is None  # gpu_processor.py:1879-1890
    d = dst.cpu().numpy()
    assert (d[..., 3] == 255).any() and (d[..., 3] == 0).any()  # 3:2 frame letterboxed into a 2:1 texture: transparent bars
    assert d[50, 100, :3].max() > 0 and hist.cpu().numpy().any()
    plain = proc.process(img, neg, 6, 0.4, **kw)
    # no scaling here would be needed at equal size: compare the centre pixel through the blit's own mapping instead
    assert plain.shape == (H, W, 3)
    # GpuProcessor.read_texture (gpu_processor.py:1311-1356): the texture's colour channels on the host
    np.testing.assert_array_equal(proc.read_texture(dst), d[..., :3])
    with pytest.raises(NotImplementedError):
        proc.read_texture(object())
    with pytest.raises(NotImplementedError):
        proc.process(img, neg, 6, 0.4, dst_texture=object(), **kw)


@pytest.mark.parametrize("layout", ["hwc3", "chw"])
def test_float_lanczos4_upscale_before_the_path(proc, layout):
    """A preview larger than the frame: cv.resize(float32, INTER_LANCZOS4) on the device, then the pipeline at that size."""
    from oracle import post

    rng = np.random.default_rng(12)
    img = rng.uniform(0.0, 2.0, (37, 53, 3)).astype(np.float32)
    t = torch.from_numpy(img if layout == "hwc3" else np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
    got = proc.ctx.resize_lanczos4_f32(t, 90, 129, layout=layout).cpu().numpy().transpose(1, 2, 0)
    np.testing.assert_array_equal(got, post.resize_lanczos4_f32(img, 90, 129))
    neg, prt, _ = stocks()
    kw = dict(print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0, seed=SEED,
              halation=False, sharpness=False, grain=0)
    xyz = st.apply_matrix3x3(rng.uniform(0.2, 0.6, (40, 60, 3)).astype(np.float32), st.REC709_TO_XYZ)  # no negative overshoot
    out = proc.process(xyz, neg, 6, 0.4, resolution=(100, 150), **kw)
    assert out.shape == (100, 150, 3)
    p = oracle_inputs(neg, prt, 150 / 36, halation=False, mtf=False, grain=0, matrix=False)
    ref = st.to_uint8(st.render(post.resize_lanczos4_f32(xyz, 100, 150), p))
    assert _u8_close(out, ref)


def test_submitted_frames_equal_processed_frames_and_overlap_is_safe(proc):
    """submit_preloaded + PendingFrame.result() == process_preloaded, also with several frames in flight whose uploads, renders
    and downloads run on three streams (each frame's buffers must survive until its own download has finished)."""
    from raw2film_amd.sharding import BatchSharder

    neg, prt, _ = stocks()
    H, W = 150, 210
    kw = dict(print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0, seed=SEED)
    frames = [_xyz(H, W, seed=60 + i) for i in range(5)]
    payloads = [proc.extract_image_data_cpu(f, frame_width=36, frame_height=24) for f in frames]
    want = [proc.process_preloaded(p, neg, 6, 0.4, **kw) for p in payloads]
    pend = [proc.submit_preloaded(p, neg, 6, 0.4, **kw) for p in payloads]  # all five in flight
    for w, h in zip(want, pend):
        np.testing.assert_array_equal(h.result(), w)
    res, skipped = BatchSharder(0, 1).run(list(range(5)), lambda i: payloads[i],
                                          lambda i, p: proc.submit_preloaded(p, neg, 6, 0.4, **kw), collect=lambda i, h: h.result())
    assert skipped == [] and all(np.array_equal(res[i], want[i]) for i in range(5))


def test_the_frame_stays_on_the_device_between_renders_with_the_same_load_parameters():
    """GpuProcessor.load_image_texture's convention (gpu_processor.py:655-719): process() prepares and uploads a frame only when
    its load parameters (or the source object) change; a re-render with other film settings reads the frame already there."""
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(21)
    img = rng.uniform(0.0, 1.0, (96, 144, 3)).astype(np.float32)
    proc = HipProcessor(device=0)
    calls = []
    extract = proc.extract_image_data_cpu
    proc.extract_image_data_cpu = lambda *a, **k: (calls.append(1), extract(*a, **k))[1]
    kw = dict(print_film=prt, lens_correction=False, seed=3)
    a = proc.process(img, neg, 6, 0.4, **kw)
    b = proc.process(img, neg, 6, 0.4, exp_comp=1.0, **kw)  # film settings only: no second load
    assert len(calls) == 1 and not np.array_equal(a, b)
    c = proc.process(img, neg, 6, 0.4, zoom=1.2, **kw)  # a load parameter: prepared again
    assert len(calls) == 2 and c.shape != a.shape or not np.array_equal(c, a)
    d = proc.process(img.copy(), neg, 6, 0.4, zoom=1.2, **kw)  # another array object, same parameters: loaded again
    assert len(calls) == 3
    np.testing.assert_array_equal(d, c)
    img2 = img.copy()
    e1 = proc.process(img2, neg, 6, 0.4, **kw)
    img2 *= 0.5  # modified in place: the content fingerprint notices, no cache=False needed
    e2 = proc.process(img2, neg, 6, 0.4, **kw)
    assert len(calls) == 5 and not np.array_equal(e1, e2)
    e3 = proc.process(img2, neg, 6, 0.4, cache=False, **kw)  # cache=False always loads
    assert len(calls) == 6
    np.testing.assert_array_equal(e3, e2)
    # a caller-supplied version token replaces the sampled checksum (an edit confined to unsampled rows is not seen by
    # the checksum -- one row of a tall frame here): same token -> the cached frame, new token -> uploaded again
    tall = rng.uniform(0.0, 1.0, (200, 64, 3)).astype(np.float32)
    n0 = len(calls)
    f1 = proc.process(tall, neg, 6, 0.4, src_version=1, **kw)
    tall[99] *= 0.25  # row 99 is not among the 32 sampled rows (every 6th) and lies inside the 3:2 crop
    f2 = proc.process(tall, neg, 6, 0.4, src_version=1, **kw)
    assert len(calls) == n0 + 1
    np.testing.assert_array_equal(f1, f2)  # (the documented limit: the caller said "unchanged")
    f3 = proc.process(tall, neg, 6, 0.4, src_version=2, **kw)
    assert len(calls) == n0 + 2 and not np.array_equal(f3, f1)
    # the processor does not keep the host frame alive
    import gc
    import weakref

    img3 = img.copy()
    ref3 = weakref.ref(img3)
    proc.process(img3, neg, 6, 0.4, **kw)
    del img3
    gc.collect()
    assert ref3() is None
    # results never depend on what was cached: a fresh processor renders the same frames
    fresh = HipProcessor(device=0)
    np.testing.assert_array_equal(fresh.process(img, neg, 6, 0.4, exp_comp=1.0, **kw), b)
    np.testing.assert_array_equal(fresh.process(img, neg, 6, 0.4, **kw), a)
    # a payload from outside replaces the cached frame
    proc.process_preloaded(fresh.extract_image_data_cpu(img * 0.25, lens_correction=False), neg, 6, 0.4, **kw)
    n = len(calls)
    np.testing.assert_array_equal(proc.process(img, neg, 6, 0.4, **kw), a)
    assert len(calls) == n + 1
    proc.close()
    fresh.close()


def test_process_clamps_the_frame_like_the_reference_load():
    """load_raw_image clamps the decoded frame to [0, 65504] (gpu_processor.py:275); process() does it on the uploaded copy."""
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(5)
    img = rng.uniform(0.0, 1.0, (64, 96, 3)).astype(np.float32)
    img[3, 4] = (1e6, -2.0, 70000.0)
    img[10, 20] = (-1e-3, 0.5, -5.0)
    proc = HipProcessor(device=0)
    kw = dict(print_film=prt, lens_correction=False, seed=2)
    got = proc.process(img, neg, 6, 0.4, **kw)
    want = proc.process(np.clip(img, 0, 65504), neg, 6, 0.4, **kw)
    np.testing.assert_array_equal(got, want)
    payload = proc.extract_image_data_cpu(img, lens_correction=False)  # the public payload is clamped on the host, like upstream's
    assert payload["image_array"].max() == 65504.0 and payload["image_array"].min() == 0.0 and not payload["clip_on_device"]
    np.testing.assert_array_equal(proc.process_preloaded(payload, neg, 6, 0.4, **kw), want)
    proc.close()


def test_a_large_host_frame_is_clamped_on_its_way_up_in_chunks():
    """Round 6: a host frame of >= 16.7 M samples goes up in eight row chunks on a copy stream, each clamped on the launch stream
    while the next one travels (the clamp of gpu_processor.py:275 no longer costs a pass of its own behind the upload).  Same
    result as clamping on the host first -- out-of-range samples in the first, a middle and the last chunk, an odd row count;
    pageable and pinned sources; and again when the array is edited and handed in a second time (no stale chunk, no stale frame)."""
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(8)
    H, W = 2403, 2400
    img = rng.uniform(0.0, 1.0, (H, W, 3)).astype(np.float32)
    img[0, 0] = (1e6, -2.0, 70000.0)
    img[H // 2 + 7, 33] = (-1e-3, 0.5, -5.0)
    img[H - 1, W - 1] = (80000.0, 65504.0, -0.0)
    proc = HipProcessor(device=0)
    proc.stream_bands = 0  # (the path a frame takes that does not stream through the pipeline: cache=True, a canvas, a burn ...)
    kw = dict(print_film=prt, lens_correction=False, seed=2, grain=0, halation=False, sharpness=False)
    want = proc.process(np.clip(img, 0, 65504), neg, 6, 0.4, cache=False, **kw)
    got = proc.process(img, neg, 6, 0.4, cache=False, **kw)
    np.testing.assert_array_equal(got, want)
    assert got.ctypes.data != want.ctypes.data  # (arrays of the caller's own: views of two of the pinned buffers the processor lends)
    pinned = torch.from_numpy(img).pin_memory().numpy()
    np.testing.assert_array_equal(proc.process(pinned, neg, 6, 0.4, cache=False, **kw), want)
    img[100:200] *= 0.5  # the same buffer, other content
    want2 = proc.process(np.clip(img, 0, 65504), neg, 6, 0.4, cache=False, **kw)
    got2 = proc.process(img, neg, 6, 0.4, cache=False, **kw)
    np.testing.assert_array_equal(got2, want2)
    assert not np.array_equal(got2, want)
    proc.close()


def test_a_large_host_frame_streams_through_the_pipeline_in_row_bands():
    """Round 6: process(host array, cache=False) with pinned result buffers takes a frame of >= 16.7 M samples through the pipeline
    in row bands (four of 600 rows here) while it arrives (HipProcessor._process_streamed): band k is clamped and fronted as soon as it is on the
    device, the halation of band k - 1 and the MTF / tail / download of band k - 2 follow.  Same frame as the one-after-the-other
    path: bit for bit where nothing but pointwise stages run (LUTs only; with grain: the hash is taken at global coordinates), to
    the FFT form's rounding with the stencil stages (their windows are anchored at each call's first row, like a row shard's) --
    out-of-range samples in the first, a middle and the last band, an odd row count, the same buffer edited and handed in again,
    and the frames that do not qualify (a canvas, a burn, a rotation) taking the other path."""
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(9)
    H, W = 2403, 2400
    img = (0.18 * 2.0 ** rng.normal(0.0, 1.5, (H, W, 1)) * rng.uniform(0.6, 1.4, (H, W, 3))).astype(np.float32)
    img[0, 0] = (1e6, -2.0, 70000.0)
    img[H // 2 + 7, 33] = (-1e-3, 0.5, -5.0)
    img[H - 1, W - 1] = (80000.0, 65504.0, -0.0)
    plain = HipProcessor(device=0, result_buffers=2)
    plain.stream_bands = 0
    banded = HipProcessor(device=0, result_buffers=2)
    base = dict(print_film=prt, lens_correction=False, seed=2, frame_width=36, frame_height=36)
    calls = []
    inner = banded._process_streamed

    def counted(*a, **k):
        res = inner(*a, **k)
        calls.append(res is not None)
        return res

    banded._process_streamed = counted
    for name, kw, exact in (("luts", dict(grain=0, halation=False, sharpness=False), True),
                            ("grain", dict(grain=2, halation=False, sharpness=False), True),
                            ("full", dict(grain=2), False),
                            ("halation", dict(grain=0, sharpness=False), False),
                            ("mtf", dict(grain=0, halation=False), False)):
        want = plain.process(img, neg, 6, 0.4, cache=False, **base, **kw).copy()
        del calls[:]
        got = banded.process(img, neg, 6, 0.4, cache=False, **base, **kw).copy()
        assert calls == [True], (name, calls)
        if exact:
            np.testing.assert_array_equal(got, want, err_msg=name)
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16))
            assert d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size, (name, int(d.max()), int(np.count_nonzero(d)))
    # the same buffer, other content: nothing of the previous frame is left in a band
    img[100:200] *= 0.5
    want2 = plain.process(img, neg, 6, 0.4, cache=False, **base, grain=2).copy()
    got2 = banded.process(img, neg, 6, 0.4, cache=False, **base, grain=2).copy()
    d = np.abs(got2.astype(np.int16) - want2.astype(np.int16))
    assert d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size and not np.array_equal(got2, got)
    # without pinned result buffers the caller gets a FRESH array per call (upstream's ownership): the same frame, other memory
    fresh = HipProcessor(device=0)
    assert fresh.result_buffers == 0 and fresh.stream_bands == 16
    f1 = fresh.process(img, neg, 6, 0.4, cache=False, **base, grain=2)
    f2 = fresh.process(img, neg, 6, 0.4, cache=False, **base, grain=2)
    np.testing.assert_array_equal(f1, got2)
    np.testing.assert_array_equal(f2, got2)
    assert f1.ctypes.data != f2.ctypes.data and f1.flags.writeable and f2.flags.writeable
    # (the arrays are views of pinned buffers the processor lends out -- up to three; a buffer comes back when the caller's last
    # reference to its array is gone, views included, and a caller that keeps more than three results gets freshly allocated arrays)
    view = f1[10:20]
    addr1 = f1.ctypes.data
    del f1
    f3 = fresh.process(img, neg, 6, 0.4, cache=False, **base, grain=2)
    assert f3.ctypes.data not in (addr1, f2.ctypes.data)  # f1's buffer is still held through `view`
    np.testing.assert_array_equal(view, got2[10:20])
    del view
    f4 = fresh.process(img, neg, 6, 0.4, cache=False, **base, grain=2)
    assert f4.ctypes.data == addr1  # ... and now it has come back
    f5 = fresh.process(img, neg, 6, 0.4, cache=False, **base, grain=2)  # f2, f3, f4 are out: a freshly allocated array
    assert f5.flags.owndata and f5.ctypes.data not in (f2.ctypes.data, f3.ctypes.data, f4.ctypes.data)
    for f in (f2, f3, f4, f5):
        np.testing.assert_array_equal(f, got2)
    fresh.close()
    # LibRaw's 16-bit output streams too: converted band by band on the device (raw_conversion.py:50-52) as it arrives
    raw = rng.integers(0, 65536, (H, W, 3), dtype=np.uint16)
    for kw, exact in ((dict(grain=0, halation=False, sharpness=False), True), (dict(grain=2), False)):
        want16 = plain.process(raw, neg, 6, 0.4, cache=False, exposure=0.5, **base, **kw).copy()
        del calls[:]
        got16 = banded.process(raw, neg, 6, 0.4, cache=False, exposure=0.5, **base, **kw).copy()
        assert calls == [True], calls
        d = np.abs(got16.astype(np.int16) - want16.astype(np.int16))
        assert (d.max() == 0) if exact else (d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size), (kw, int(d.max()), int(np.count_nonzero(d)))
    # the two-phase API's device phase streams its payload the same way: upstream's payload with the constant alpha plane (clamped on
    # the host in phase 1), the uint16 payload, serial and "submitted"
    pre = {k: v for k, v in base.items() if k not in ("lens_correction", "frame_width", "frame_height")}
    for src_, ex in ((img, {}), (raw, dict(exposure=0.5))):
        pay = banded.extract_image_data_cpu(src_, lens_correction=False, frame_width=36, frame_height=36, **ex)
        assert pay["image_array"].shape[2] == (4 if src_ is img else 3)
        want5 = plain.process_preloaded(pay, neg, 6, 0.4, frame_width=36, frame_height=36, grain=2, **pre).copy()
        banded.stream_rejected = "not asked"
        got5 = banded.process_preloaded(pay, neg, 6, 0.4, frame_width=36, frame_height=36, grain=2, **pre).copy()
        assert banded.stream_rejected is None
        got6 = banded.submit_preloaded(pay, neg, 6, 0.4, frame_width=36, frame_height=36, grain=2, **pre)
        assert got6.ready()  # (a pageable payload streams and is back at once; a pinned one stays in flight)
        np.testing.assert_array_equal(got6.result(), got5)
        d = np.abs(got5.astype(np.int16) - want5.astype(np.int16))
        assert d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size, (int(d.max()), int(np.count_nonzero(d)))
    # the histogram source is the streamed frame's device copy
    got2 = banded.process(img, neg, 6, 0.4, cache=False, **base, grain=2).copy()
    np.testing.assert_array_equal(banded.last_output.cpu().numpy(), got2)
    # host-side geometry (index arithmetic ahead of the upload: aspect crop, zoom, quarter turns, flip) streams like any other frame
    streamed = 0
    for kw in (dict(zoom=1.25), dict(rotate_times=1, frame_width=36, frame_height=36), dict(flip=True), dict(frame_width=36, frame_height=24),
               dict(rotate_times=2, zoom=1.1)):
        args = dict(base, cache=False, grain=2, halation=False, sharpness=False)
        args.update(kw)
        del calls[:]
        got4 = banded.process(img, neg, 6, 0.4, **args)
        want4 = plain.process(img, neg, 6, 0.4, **args)
        np.testing.assert_array_equal(got4, want4, err_msg=str(kw))
        streamed += calls == [True]
    assert streamed >= 2, streamed  # (a crop below 16.7 M samples takes the other path)
    # an export in between leaves the frame a preview keeps on the device alone: the preview's next re-render uploads nothing
    loads = []
    inner_load = banded.prepare_gpu_textures
    banded.prepare_gpu_textures = lambda p: (loads.append(1), inner_load(p))[1]
    pv = dict(base, resolution=(600, 600), grain=0, halation=False, sharpness=False)
    p1 = banded.process(img, neg, 6, 0.4, **pv)
    assert loads == [1]
    del calls[:]
    banded.process(img, neg, 6, 0.4, cache=False, **base, grain=0, halation=False, sharpness=False)  # the export: streamed
    assert calls == [True] and loads == [1]
    p2 = banded.process(img, neg, 6, 0.4, **pv)
    np.testing.assert_array_equal(p1, p2)
    assert loads == [1], "the preview's frame had to be uploaded again"
    banded.prepare_gpu_textures = inner_load
    # frames that do not qualify take the other path and give its result
    for kw in (dict(canvas_mode="Proportional", canvas_scale=1.1), dict(highlight_burn=0.5), dict(rotation=3.0), dict(cache=True)):
        del calls[:]
        args = dict(cache=False, grain=0, halation=False, sharpness=False)
        args.update(kw)
        got3 = banded.process(img, neg, 6, 0.4, **base, **args)
        want3 = plain.process(img, neg, 6, 0.4, **base, **args)
        np.testing.assert_array_equal(got3, want3, err_msg=str(kw))
        assert not any(calls), (kw, calls)
    plain.close()
    banded.close()


def test_a_lent_result_is_never_written_while_anybody_holds_it():
    """process() without result_buffers hands back the caller's own array -- since round 6 a view of one of up to three pinned
    buffers the processor lends out and takes back when the array and every view of it are gone (hip_processor._lease_result).
    A random walk over what a caller may do with results -- keep them, keep a slice only, drop them, keep more than three --
    while frames keep coming: whatever is still held, array or slice, keeps the content it came with."""
    import gc

    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(21)
    H, W = 1000, 1500  # 4.5 M samples: above the size from which downloads are lent
    base = rng.uniform(0.0, 1.0, (H, W, 3)).astype(np.float32)
    proc = HipProcessor(device=0)
    kw = dict(print_film=prt, lens_correction=False, seed=2, grain=0, halation=False, sharpness=False)
    held = []  # (what the caller kept, a private copy of what it must still hold)
    leases = []
    inner = proc._lease_result
    proc._lease_result = lambda shape: (leases.append(inner(shape)), leases[-1])[1]
    for i in range(60):
        out = proc.process(base * np.float32(0.2 + 0.8 * rng.random()), neg, 6, 0.4, cache=False, **kw)
        what = int(rng.integers(0, 4))
        if what == 0:
            held.append((out, out.copy()))
        elif what == 1:
            a = int(rng.integers(0, H - 50))
            held.append((out[a:a + 50, ::7], out[a:a + 50, ::7].copy()))
        del out
        while held and rng.integers(0, 3) == 0:
            held.pop(int(rng.integers(0, len(held))))
        gc.collect()
        for kept, want in held:
            np.testing.assert_array_equal(kept, want)
    lent, fresh = sum(t is not None for t in leases), sum(t is None for t in leases)
    assert len(leases) == 60 and lent > 20 and fresh > 0, (lent, fresh)  # lent buffers and, with more than three held, fresh arrays
    assert len({t.data_ptr() for t in leases if t is not None}) <= 3
    del leases
    proc.close()
    for kept, want in held:  # (a closed processor does not take its buffers from under the caller)
        np.testing.assert_array_equal(kept, want)


def test_pinned_result_buffers_return_views_in_turn():
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    img = np.random.default_rng(8).uniform(0.0, 1.0, (48, 72, 3)).astype(np.float32)
    kw = dict(print_film=prt, lens_correction=False, seed=2)
    plain = HipProcessor(device=0)
    ring = HipProcessor(device=0, result_buffers=2)
    want = [plain.process(img, neg, 6, 0.4, exp_comp=e, **kw) for e in (0.0, 0.5, 1.0)]
    a = ring.process(img, neg, 6, 0.4, exp_comp=0.0, **kw)
    b = ring.process(img, neg, 6, 0.4, exp_comp=0.5, **kw)
    np.testing.assert_array_equal(a, want[0])  # two buffers: the first result is still intact after the second call
    np.testing.assert_array_equal(b, want[1])
    c = ring.process(img, neg, 6, 0.4, exp_comp=1.0, **kw)
    np.testing.assert_array_equal(c, want[2])
    assert np.shares_memory(a, c) and not np.shares_memory(a, b)  # ... and is overwritten by the third
    fresh = plain.process(img, neg, 6, 0.4, **kw)
    assert not np.shares_memory(fresh, plain.process(img, neg, 6, 0.4, **kw))  # the default: a new array per call
    plain.close()
    ring.close()
