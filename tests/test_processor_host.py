"""Host logic of the HipProcessor operator surface, with the device context replaced by a recorder
(no GPU needed): keyword surface, stage gating, upload caching, out-of-scope errors, settings merge."""

import inspect

import numpy as np
import pytest

from raw2film_amd import _lib, filmstock, settings
from raw2film_amd.hip_processor import REC709_TO_XYZ, HipProcessor

REFERENCE_PROCESS_KWARGS = [  # cpu_processor.py:269-322 (+ dst_texture/histogram_texture of gpu_processor.py:1547-1548)
    "lens_correction", "print_film", "exp_comp", "red_light", "green_light", "blue_light", "projector_kelvin",
    "shadow_comp", "sat_adjust", "gamma_func", "exp_kelvin", "tint", "inversion_gamma", "idealized_curve", "inversion",
    "push_pull", "white_balance", "white_clip", "icc_transform", "resolution", "frame_width", "frame_height", "rotation",
    "zoom", "rotate_times", "flip", "cam", "lens", "canvas_mode", "canvas_scale", "canvas_ratio", "halation_intensity",
    "halation", "halation_size", "halation_green_factor", "sharpness", "sharpening_strength", "sharpening_sigma",
    "chroma_nr", "grain", "highlight_burn", "burn_scale", "half_size", "cache", "color_masking", "max_scale",
]
REFERENCE_DEFAULTS = dict(lens_correction=True, print_film=None, exp_comp=0.0, projector_kelvin=6500, sat_adjust=1.0,
                          gamma_func="sRGB", exp_kelvin=6500, inversion_gamma=4.0, frame_width=36, frame_height=24,
                          halation=True, halation_size=1.0, halation_green_factor=0.4, sharpness=True, grain=2,
                          half_size=True, cache=True, color_masking=None, max_scale=400.0, burn_scale=50.0)


class RecorderContext:
    def __init__(self):
        self.calls = []
        self.device = "fake"

    def __getattr__(self, name):
        if name.startswith("set_"):
            return lambda *a, **k: self.calls.append(name)
        raise AttributeError(name)

    def make_params(self, **kw):
        self.calls.append(("params", kw))
        return kw

    def layout_of(self, image):
        return 0, image.shape[0], image.shape[1]

    def render(self, image, params, want_f32=False, want_u8=True):
        self.calls.append("render")
        return None, np.zeros(image.shape[:2] + (3,), np.uint8)


@pytest.fixture
def proc():
    p = HipProcessor.__new__(HipProcessor)
    p.ctx = RecorderContext()
    p.device = "fake"
    p.cameras = p.lenses = None
    for name in ("input", "curve", "output", "halation", "mtf", "grain_kernel", "grain_lut"):
        setattr(p, f"{name}_param_dict", None)
    p.matrix_key = None
    p.uploads = 0
    return p


def test_process_keeps_the_reference_keyword_surface():
    sig = inspect.signature(HipProcessor.process)
    names = list(sig.parameters)
    assert names[:5] == ["self", "src", "negative_film", "grain_size", "grain_sigma"]
    for kw in REFERENCE_PROCESS_KWARGS:
        assert kw in sig.parameters, kw
    for kw, default in REFERENCE_DEFAULTS.items():
        assert sig.parameters[kw].default == default, kw
    assert any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())  # **_ swallows GUI extras


def test_stage_gating_and_upload_caching(proc):
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    params = proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, seed=5, matrix=REC709_TO_XYZ, profile="x", film_format="135")
    assert params["halation"] and params["mtf"] and params["grain"] and not params["grain_mono"] and params["seed"] == 5
    first = proc.uploads
    assert first == 7  # 2-D LUT, curve, 3-D LUT, halation, MTF, grain LUT, grain kernel
    proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, seed=6, matrix=REC709_TO_XYZ)
    assert proc.uploads == first  # nothing changed -> nothing re-uploaded (cpu_processor.py:157-158 convention)
    proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, exp_comp=0.5)
    assert proc.uploads == first + 1  # only the input LUT depends on exp_comp
    proc.prepare(neg, 6, 0.4, (1200, 800), print_film=prt, exp_comp=0.5)
    assert proc.uploads == first + 1 + 4  # new px/mm: halation, MTF, grain LUT, grain kernel
    p2 = proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, halation=False, sharpness=False, grain=1)
    assert not p2["halation"] and not p2["mtf"] and p2["grain"] and p2["grain_mono"]
    p3 = proc.prepare(prt, 6, 0.4, (600, 400))  # print stock: no MTF table, no granularity -> stages off
    assert not p3["mtf"] and not p3["grain"]


def test_bw_stock_sets_equal_halation_factors(proc, monkeypatch):
    from raw2film_amd import stencils

    seen = {}
    real = stencils.halation_stencil
    monkeypatch.setattr(stencils, "halation_stencil", lambda *a, **k: seen.update(k) or real(*a, **k))
    bw = filmstock.builtin_stocks()["Kodak Tri-X 400"]
    proc.prepare(bw, 6, 0.4, (600, 400))
    assert seen["bw"] is True


def test_highlight_burn_gate_and_parameters(proc):
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    on = proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, highlight_burn=0.5, burn_scale=50)
    assert on["burn_strength"] == 0.5 and on["burn_cell"] == 8 and on["burn_d_ref"] == neg.d_ref[1]
    off = proc.prepare(neg, 6, 0.4, (600, 400), print_film=prt, highlight_burn=0.0)
    assert "burn_strength" not in off
    # cpu_processor.py:399-402: without a print stock only status_m / bw negatives burn
    odd = filmstock.SyntheticStock("odd", density_measure="status_a")
    assert "burn_strength" not in proc.prepare(odd, 6, 0.4, (600, 400), print_film=None, highlight_burn=0.5)
    assert "burn_strength" in proc.prepare(neg, 6, 0.4, (600, 400), print_film=None, highlight_burn=0.5)
    # GpuProcessor.load_highlight_burn (gpu_processor.py:856-878): the low-resolution grid of the frame prepared last
    d = proc.load_highlight_burn(neg, 0.5, 50)
    assert d["lowres_w"] == 600 // 8 and d["lowres_h"] == 400 // 8 and d["d_ref"] == neg.d_ref[1] and d is proc.highlight_burn_param_dict
    big = proc.load_highlight_burn(neg, 0.25, 50.0, (6000, 4000))
    assert (big["cell"], big["lowres_w"], big["lowres_h"]) == (80, 75, 50)


def test_random_seed_when_not_given(proc):
    neg = filmstock.builtin_stocks()["Kodak Portra 400"]
    seeds = {proc.prepare(neg, 6, 0.4, (60, 40))["seed"] for _ in range(4)}
    assert len(seeds) > 1  # upstream draws a new seed every render (gpu_processor.py:591)


def test_out_of_scope_requests_raise(proc):
    neg = filmstock.builtin_stocks()["Kodak Portra 400"]
    img = np.zeros((40, 60, 3), np.float32)
    with pytest.raises(NotImplementedError):
        proc.process("photo.cr3", neg, 6, 0.4)
    up = proc.extract_image_data_cpu(img, resolution=(80, 120))  # a preview larger than the frame: LANCZOS4 on the device
    assert up["resize_to"] == (80, 120) and up["pipeline_resolution"] == (120, 80)
    with pytest.raises(NotImplementedError):
        proc.process(img, neg, 6, 0.4, dst_texture=object())


def test_extract_crops_zooms_turns_and_reports_canvas(proc):
    img = np.random.default_rng(1).uniform(0, 1, (400, 640, 3)).astype(np.float32)
    p = proc.extract_image_data_cpu(img, frame_width=36, frame_height=24, zoom=2.0, rotate_times=1, canvas_mode="Uniform white",
                                    canvas_scale=1.1, max_scale=None)
    assert p["pipeline_resolution"] == (p["image_array"].shape[1], p["image_array"].shape[0])
    assert p["image_array"].shape[0] > p["image_array"].shape[1]  # quarter turn of a landscape crop
    w, h = p["pipeline_resolution"]
    assert p["canvas_resolution"] == (w + int(max(h, w) * 0.1), h + int(max(h, w) * 0.1)) or p["canvas_resolution"][0] > w


def test_free_rotation_is_deferred_to_the_device_with_the_reference_window(proc):
    """Phase 1 only plans the rotation: aspect crop on the host, then a warp into the window effects.rotate + the zoom crop keep."""
    from oracle import stages as st
    from raw2film_amd import geometry

    img = np.random.default_rng(2).uniform(0, 1, (420, 600, 3)).astype(np.float32)
    for deg, zoom, k in ((3.5, 1.0, 0), (-8.0, 1.3, 1), (0.7, 1.0, 2)):
        p = proc.extract_image_data_cpu(img, rotation=deg, zoom=zoom, rotate_times=k, max_scale=None)
        # the oracle's whole-frame route: aspect crop -> rotate (warp + crop) -> zoom crop -> quarter turns
        r0, c0, nr, nc = geometry.crop_box(420, 600, 1, 1.5, False)
        ref = st.rotate(img[r0:r0 + nr, c0:c0 + nc], deg)
        z = geometry.crop_box(ref.shape[0], ref.shape[1], zoom, 1.5, False)
        ref = np.rot90(ref[z[0]:z[0] + z[2], z[1]:z[1] + z[3]], k)
        assert p["pipeline_resolution"] == (ref.shape[1], ref.shape[0])
        assert p["image_array"].shape[:2] == (nr, nc) and p["warp"]["rotate_times"] == k
        # and the planned window reproduces that route when the oracle's warp is evaluated on it
        w = p["warp"]
        direct = st.warp_affine_linear(img[r0:r0 + nr, c0:c0 + nc], w["m_dst_to_src"], w["window"][2:], w["window"][:2])
        assert np.array_equal(np.rot90(direct, k), ref)
    assert proc.extract_image_data_cpu(img)["warp"] is None


def test_max_scale_round_trip_is_planned_like_the_cpu_processor(proc):
    """cpu_processor.py:119-134 + 411-412: a frame finer than max_scale px/mm is rendered at max_scale and scaled back."""
    img = np.zeros((400, 600, 3), np.float32)
    # super-8 gate: 600 px over 5.79 mm = 103.6 px/mm; with max_scale 40 the pipeline runs at 40 px/mm
    p = proc.extract_image_data_cpu(img, frame_width=5.79, frame_height=3.86, max_scale=40.0)
    f = 40.0 / (600 / 5.79)
    # the clamped target is (154, 232); utils.resolution_scaling then fits the frame INSIDE it with one factor (the smaller)
    assert (round(400 * f), round(600 * f)) == (154, 232)
    assert p["resize_to"] == (154, 231) and p["pipeline_resolution"] == (231, 154)
    assert p["upscale_to"] == (400, 600)
    # `output_resolution` is what GpuProcessor.extract_image_data_cpu reports (gpu_processor.py:764: the PIPELINE size divided by
    # the clamp factor, rounded -- not the size of the finished frame, which follows `upscale_to`): pinned for 3 344 cases by
    # tests/golden/payload_geometry.npz
    assert p["output_resolution"] == (round(231 / f), round(154 / f)) == (598, 399)
    # below the limit nothing happens; max_scale=None switches the clamp off
    q = proc.extract_image_data_cpu(img, frame_width=36, frame_height=24, max_scale=40.0)
    assert q["resize_to"] is None and q["upscale_to"] is None and q["output_resolution"] == (600, 400)
    assert proc.extract_image_data_cpu(img, frame_width=5.79, frame_height=3.86, max_scale=None)["upscale_to"] is None
    # a preview resolution above the limit is clamped too, and restored at the end
    r = proc.extract_image_data_cpu(img, frame_width=5.79, frame_height=3.86, max_scale=40.0, resolution=(200, 300))
    assert r["upscale_to"] == (200, 300) and r["resize_to"] == (154, 231)


def test_preview_resolution_becomes_an_area_downscale(proc):
    img = np.zeros((400, 600, 3), np.float32)
    p = proc.extract_image_data_cpu(img, resolution=(100, 200))  # widget 100 x 200: limited by the height factor 0.25
    assert p["resize_to"] == (100, 150) and p["pipeline_resolution"] == (150, 100) and p["output_resolution"] == (150, 100)
    assert p["image_array"].shape == (400, 600, 4)  # the frame itself is resized on the device in phase 2
    assert proc.extract_image_data_cpu(img, resolution=(400, 600))["resize_to"] is None


def test_extract_image_data_cpu_payload(proc):
    img = np.full((40, 60, 3), 70000.0, np.float32)
    payload = proc.extract_image_data_cpu(img)
    assert payload["image_array"].shape == (40, 60, 4) and payload["image_array"].dtype == np.float32
    assert payload["image_array"][..., :3].max() == 65504.0 and np.all(payload["image_array"][..., 3] == 1.0)
    assert payload["output_resolution"] == (60, 40) and payload["pipeline_resolution"] == (60, 40)
    assert payload["canvas_resolution"] is None and payload["chroma_nr"] == 0
    assert proc.extract_image_data_cpu(img, chroma_nr=3)["chroma_nr"] == 3


def test_settings_merge_and_preview_gating():
    stocks = filmstock.builtin_stocks()
    args = settings.build_processing_params(stocks, image_params={"exp_comp": 1.0}, profile_params={"grain_size": 9})
    assert args["negative_film"] is stocks["Kodak Portra 400"] and args["print_film"] is stocks["Fuji Crystal Archive Maxima"]
    assert args["exp_comp"] == 1.0 and args["grain_size"] == 9 and args["halation_green_factor"] == 0.3
    assert args["exp_kelvin"] == 6000 and args["color_masking"] == 1.0
    quick = settings.build_processing_params(stocks, full_preview=False)
    assert quick["sharpness"] is False and quick["grain"] == 0 and quick["halation"] is False
    inv = settings.build_processing_params(stocks, profile_params={"print_film": "Inversion"})
    assert inv["inversion"] is True and inv["print_film"] is None
    assert settings.FORMATS["135"] == (36, 24)


def test_synthetic_stock_lut_shapes():
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    assert neg.get_input_lut(6000, 0, 0).shape == (64, 64, 3)
    c = neg.get_density_curve(0.0, 1.0)
    assert c.shape == (4, 1024) and np.all(np.diff(c[0]) > 0) and np.all(np.diff(c[1:], axis=1) >= 0)
    assert neg.get_grain_curve(341.33).shape == (4, 256)
    lut = filmstock.create_lut(neg, prt)
    assert lut.shape == (33, 33, 33, 3) and lut.min() >= 0 and lut.max() <= 1
    assert filmstock.grain_kernel(1 / 341.33).shape == (9, 9) and filmstock.grain_kernel(1 / 14.22) is None
    assert prt.mtf is None and prt.rms_density is None and len(neg.mtf) == 3
    assert hash(neg) is not None and _lib.F_GRAIN == 8


def test_bundle_roundtrip(tmp_path):
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    path = str(tmp_path / "portra.npz")
    filmstock.save_bundle(path, lut_2d=neg.get_input_lut(), lut_1d=neg.get_density_curve(), lut_3d=filmstock.create_lut(neg, prt),
                          grain_lut=neg.get_grain_curve(100.0), rms_density=neg.rms_density, d_ref=np.array(neg.d_ref),
                          mtf_logf=np.stack([m[0] for m in neg.mtf]), mtf_vals=np.stack([m[1] for m in neg.mtf]))
    b = filmstock.load_bundle(path, "portra-bundle")
    np.testing.assert_array_equal(b.get_input_lut(6500, 0, 0), neg.get_input_lut())
    np.testing.assert_array_equal(filmstock.create_lut(b, None), filmstock.create_lut(neg, prt))
    assert len(b.mtf) == 3 and b.rms_density is not None and b.name == "portra-bundle"


def test_lens_correction_is_refused_only_when_the_reference_would_correct(proc):
    """effects.lens_correction (effects.py:22-30) does something only with a camera AND a lens; cpu_processor.py:107-108 drops
    both when lens_correction is False.  That one case is outside the accelerated path and must not render silently."""
    frame = np.full((40, 60, 3), 0.2, np.float32)
    ok = HipProcessor.extract_image_data_cpu(proc, frame, cam=None, lens=None, lens_correction=True)  # the GUI default
    assert ok["image_array"].shape == (40, 60, 4)
    HipProcessor.extract_image_data_cpu(proc, frame, cam="cam", lens=None, lens_correction=True)
    HipProcessor.extract_image_data_cpu(proc, frame, cam="cam", lens="lens", lens_correction=False)
    with pytest.raises(NotImplementedError, match="lens correction"):
        HipProcessor.extract_image_data_cpu(proc, frame, cam="cam", lens="lens", lens_correction=True)


def test_icc_transform_is_applied_to_the_output_lut_in_8_bits(proc):
    """cpu_processor.py:255-263: with an ICC transform the 3-D LUT itself goes through (lut * 255).astype(uint8) -> PIL
    ImageCms.applyTransform -> / 255 before it is uploaded; the transform is part of the cache key."""
    from PIL import Image, ImageCms

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    uploaded = []
    proc.ctx.set_lut3d = lambda lut: uploaded.append(np.array(lut, copy=True))
    srgb = ImageCms.createProfile("sRGB")
    to_lab = ImageCms.buildTransform(srgb, srgb, "RGB", "RGB", renderingIntent=ImageCms.Intent.PERCEPTUAL)
    proc.load_output_lut(neg, prt, icc_transform=None, color_masking=1.0)
    proc.load_output_lut(neg, prt, icc_transform=to_lab, color_masking=1.0)
    proc.load_output_lut(neg, prt, icc_transform=to_lab, color_masking=1.0)  # unchanged -> cached
    assert len(uploaded) == 2
    plain, icc = uploaded
    # the same steps by hand, as the reference spells them
    lut = (plain * 255).astype(np.uint8)
    img = Image.fromarray(lut.reshape(lut.shape[0], -1, lut.shape[-1]))
    ImageCms.applyTransform(img, to_lab, inPlace=True)
    want = (np.array(img, np.uint8).reshape(lut.shape) / 255.0).astype(np.float32)
    np.testing.assert_array_equal(icc, want)
    assert icc.dtype == np.float32 and len(np.unique(np.round(icc * 255))) <= 256
    assert np.abs(icc - plain).max() <= 1.5 / 255 and not np.array_equal(icc, plain)  # 8-bit quantisation happened


def test_payload_without_the_alpha_plane():
    """HipProcessor(payload_alpha=False): phase 1 hands over (H, W, 3) -- nothing on this backend reads upstream's constant alpha
    (gpu_processor.py:765), and the frame crosses PCIe a quarter smaller; 4-channel sources lose theirs."""
    proc = HipProcessor.__new__(HipProcessor)
    proc.payload_alpha = False
    img = np.random.default_rng(3).uniform(0, 1, (40, 60, 3)).astype(np.float32)
    p = HipProcessor.extract_image_data_cpu(proc, img, lens_correction=False)
    assert p["image_array"].shape == (40, 60, 3) and p["image_array"].dtype == np.float32
    np.testing.assert_array_equal(p["image_array"], img)
    rgba = np.concatenate([img, np.ones_like(img[..., :1])], axis=-1)
    p4 = HipProcessor.extract_image_data_cpu(proc, rgba, lens_correction=False)
    np.testing.assert_array_equal(p4["image_array"], img)
