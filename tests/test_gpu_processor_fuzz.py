"""HipProcessor.process() under randomised load and render settings against the reference's own sequence of steps
(CpuProcessor.load_image_texture + process, cpu_processor.py:70-134, 323-414; raw_conversion.crop_rotate_zoom, raw_conversion.py:56-72;
utils.resolution_scaling, utils.py:226-244) restated from the oracle's pieces, each of which has its own pinned or per-stage test:
aspect crop / free rotation / zoom / quarter turns / flip, chroma NR, the preview resolution and the max_scale round trip (both
directions of resolution_scaling), the pipeline, the canvas, the final scaling of the framed uint8 frame.  Fixed seeds."""

import os

import numpy as np
import pytest

from oracle import post
from oracle import stages as st

from helpers import SEED, oracle_inputs, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _scale_float(image, resolution):
    """utils.resolution_scaling on the float frame before the path."""
    h, w = image.shape[:2]
    f = min(resolution[0] / h, resolution[1] / w)
    if f < 1:
        oh, ow = round(h * f), round(w * f)
        return np.stack([st.resize_area(np.ascontiguousarray(image[..., c]), oh, ow) for c in range(3)], axis=-1)
    if f > 1:
        return post.resize_lanczos4_f32(np.ascontiguousarray(image), round(h * f), round(w * f))
    return image


def _scale_u8(image, resolution):
    """utils.resolution_scaling on the finished uint8 frame."""
    h, w = image.shape[:2]
    f = min(resolution[0] / h, resolution[1] / w)
    if f < 1:
        return post.resize_area_u8(np.ascontiguousarray(image), round(h * f), round(w * f))
    return st.resolution_scaling_u8_up(np.ascontiguousarray(image), resolution)


def reference_process(img, stock, prt, c):
    from raw2film_amd import geometry

    fw, fh = c["frame"]
    aspect = fw / fh
    # raw_conversion.crop_rotate_zoom
    r0, c0, nr, nc = geometry.crop_box(img.shape[0], img.shape[1], 1, aspect, c["flip"])
    x = np.ascontiguousarray(img[r0:r0 + nr, c0:c0 + nc])
    if c["rotation"]:
        x = st.rotate(x, c["rotation"])
    z = geometry.crop_box(x.shape[0], x.shape[1], c["zoom"], aspect, False)
    x = np.ascontiguousarray(np.rot90(x[z[0]:z[0] + z[2], z[1]:z[1] + z[3]], c["turns"]))
    if c["nr"]:
        x = st.chroma_nr_filter(x, c["nr"])
    # cpu_processor.py:115-131
    resolution = c["resolution"]
    if resolution is None and c["max_scale"] is not None:
        resolution = x.shape[:2]
    orig = None if resolution is None else list(resolution)
    if resolution is not None:
        scale = max(resolution) / max(fw, fh)
        if c["max_scale"] is not None and scale > c["max_scale"]:
            resolution = [round(v * c["max_scale"] / scale) for v in resolution]
        x = _scale_float(x, resolution)
    # the path, cpu_processor.py:363-407
    scale = max(x.shape) / max(fw, fh)
    p = oracle_inputs(stock, prt, scale, halation=c["halation"], mtf=c["mtf"], grain=c["grain"], seed=c["seed"], matrix=False,
                      halation_green_factor=c["green"], grain_size=c["grain_size"])
    if c["burn"]:
        p.highlight_burn, p.burn_scale, p.d_ref = c["burn"], 50.0, float(stock.d_ref[1] if len(stock.d_ref) > 1 else stock.d_ref[0])
    out = st.to_uint8(st.render(np.ascontiguousarray(x), p))
    out = geometry.add_canvas(out, c["canvas"], c["canvas_scale"], c["canvas_ratio"])  # cpu_processor.py:409
    if orig is not None:
        out = _scale_u8(out, orig)  # :411-412
    return out, x.shape[:2]


def _cases(n=int(os.environ.get("R2F_PROC_FUZZ_CASES", "16"))):
    rng = np.random.default_rng(int(os.environ.get("R2F_PROC_FUZZ_SEED", "20261003")))
    out = []
    for i in range(n):
        H, W = int(rng.integers(90, 260)), int(rng.integers(90, 340))
        resolution = None
        if rng.integers(0, 3) == 0:
            resolution = (int(rng.integers(40, 300)), int(rng.integers(40, 300)))
        out.append(dict(
            H=H, W=W, frame=[(36.0, 24.0), (24.0, 36.0), (5.79, 3.86), (56.0, 56.0), (21.95, 9.35)][int(rng.integers(0, 5))],
            flip=bool(rng.integers(0, 4) == 0), rotation=float(rng.choice([0.0, 0.0, 1.7, -8.0, 31.0])),
            zoom=float(rng.choice([1.0, 1.0, 1.3, 2.0])), turns=int(rng.integers(0, 4)), nr=int(rng.choice([0, 0, 2])),
            resolution=resolution, max_scale=[None, 400.0, 20.0, 6.0][int(rng.integers(0, 4))],
            halation=bool(rng.integers(0, 2)), mtf=bool(rng.integers(0, 2)), grain=int(rng.integers(0, 3)),
            green=float(rng.choice([0.3, 0.4])), grain_size=float(rng.choice([6.0, 12.0])), burn=float(rng.choice([0.0, 0.0, 0.6])),
            canvas=str(rng.choice(["No", "No", "Uniform white", "Fixed black", "Proportional grey"])),
            canvas_scale=float(rng.choice([1.1, 1.25])), canvas_ratio=float(rng.choice([1.0, 1.5])), seed=int(rng.integers(0, 2**31)),
            bw=bool(rng.integers(0, 5) == 0)))
    return out


@pytest.fixture(scope="module")
def proc():
    from raw2film_amd import HipProcessor

    p = HipProcessor(cameras={}, lenses={}, device=0)
    yield p
    p.close()


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"{c['H']}x{c['W']}-f{c['frame'][0]:g}-r{c['rotation']:g}-z{c['zoom']:g}-t{c['turns']}-res{c['resolution']}-ms{c['max_scale']}")
def test_process_under_random_settings(proc, c):
    neg, prt, bw = stocks()
    stock = bw if c["bw"] else neg
    img = st.apply_matrix3x3(synthetic_frame(c["H"], c["W"], seed=c["seed"] % 997), st.REC709_TO_XYZ)
    # (resolution None with max_scale None: the reference itself raises there -- resolution[:] of None, cpu_processor.py:118 -- and
    # its callers always pass a max_scale; this processor renders at the frame's own size, which is what the restatement does too)
    ref, pipeline_hw = reference_process(img, stock, prt, c)
    out = proc.process(img, stock, c["grain_size"], 0.4, print_film=prt, exp_kelvin=6000, color_masking=1.0, seed=c["seed"],
                       frame_width=c["frame"][0], frame_height=c["frame"][1], rotation=c["rotation"], zoom=c["zoom"],
                       rotate_times=c["turns"], flip=c["flip"], chroma_nr=c["nr"], resolution=c["resolution"], max_scale=c["max_scale"],
                       halation=c["halation"], sharpness=c["mtf"], grain=c["grain"], halation_green_factor=c["green"],
                       highlight_burn=c["burn"], burn_scale=50.0, canvas_mode=c["canvas"], canvas_scale=c["canvas_scale"],
                       canvas_ratio=c["canvas_ratio"], cache=False)
    assert out.shape == ref.shape, (out.shape, ref.shape, pipeline_hw)
    d = np.abs(out.astype(int) - ref.astype(int))
    # a 1-LSB truncation flip of the rendered frame survives an INTER_AREA shrink in a few samples and spreads over 8 x 8 taps of a
    # LANCZOS4 enlargement; the free rotation's bilinear weights and the chroma NR round in float32 on both sides
    scaled = ref.shape[:2] != (pipeline_hw if c["canvas"] == "No" else None)
    lim, frac = (3, 5e-3) if (scaled or c["rotation"] or c["nr"]) else (1, 1e-4)
    # (the contract's fraction has a quantum: on a frame of fewer than 1 / frac samples -- 39 x 59 x 3 here and there -- ONE truncation
    # flip at a float32 rounding boundary is already 1.4e-4 of the samples; one flip is always within the contract)
    assert d.max() <= lim and int((d > 0).sum()) <= max(1, frac * d.size), (int(d.max()), float((d > 0).mean()), c)


@pytest.mark.parametrize("c", _cases(int(os.environ.get("R2F_PROC_FUZZ_CASES", "12"))), ids=lambda c: f"{c['H']}x{c['W']}-f{c['frame'][0]:g}-z{c['zoom']:g}-t{c['turns']}-ms{c['max_scale']}-{c['canvas'].split()[0]}")
def test_two_phase_api_into_a_destination_texture_under_random_settings(proc, c):
    """The GPU processor's preview branch (gui.py:2472-2514; gpu_processor.py:715-783, 1865-1890): extract_image_data_cpu ->
    process_preloaded(dst_texture=...).  The payload's three resolutions are the product's (pinned against the reference by
    tests/golden/payload_geometry.npz); the frame is the oracle's float render of the oracle-prepared frame at the payload's
    pipeline resolution, letterboxed by the oracle's copy_to_int.wgsl restatement with the transform those resolutions give."""
    from raw2film_amd import geometry

    neg, prt, bw = stocks()
    stock = bw if c["bw"] else neg
    img = st.apply_matrix3x3(synthetic_frame(c["H"], c["W"], seed=c["seed"] % 997), st.REC709_TO_XYZ)
    fw, fh = c["frame"]
    load = dict(frame_width=fw, frame_height=fh, rotation=c["rotation"], zoom=c["zoom"], rotate_times=c["turns"], flip=c["flip"],
                chroma_nr=c["nr"], resolution=c["resolution"], max_scale=c["max_scale"], canvas_mode=c["canvas"],
                canvas_scale=c["canvas_scale"], canvas_ratio=c["canvas_ratio"])
    payload = proc.extract_image_data_cpu(img, **load)
    pw, ph = payload["pipeline_resolution"]
    # the frame the pipeline reads, from the oracle's pieces
    aspect = fw / fh
    r0, c0, nr, nc = geometry.crop_box(img.shape[0], img.shape[1], 1, aspect, c["flip"])
    x = np.ascontiguousarray(img[r0:r0 + nr, c0:c0 + nc])
    if c["rotation"]:
        x = st.rotate(x, c["rotation"])
    z = geometry.crop_box(x.shape[0], x.shape[1], c["zoom"], aspect, False)
    x = np.ascontiguousarray(np.rot90(x[z[0]:z[0] + z[2], z[1]:z[1] + z[3]], c["turns"]))
    if c["nr"]:
        x = st.chroma_nr_filter(x, c["nr"])
    resolution = c["resolution"]  # gpu_processor.py:753-762
    if resolution is None and c["max_scale"] is not None:
        resolution = x.shape[:2]
    if resolution is not None:
        scale = max(resolution) / max(fw, fh)
        if c["max_scale"] is not None and scale > c["max_scale"]:
            resolution = [round(v * c["max_scale"] / scale) for v in resolution]
        x = _scale_float(x, resolution)
    assert x.shape[:2] == (ph, pw), (x.shape, (ph, pw))
    p = oracle_inputs(stock, prt, max(x.shape) / max(fw, fh), halation=c["halation"], mtf=c["mtf"], grain=c["grain"], seed=c["seed"],
                      matrix=False, halation_green_factor=c["green"], grain_size=c["grain_size"])
    rendered = st.render(np.ascontiguousarray(x), p)
    dh, dw = 150 + c["seed"] % 200, 180 + (c["seed"] // 7) % 260
    dst = torch.zeros((dh, dw, 4), dtype=torch.uint8, device="cuda")
    out = proc.process_preloaded(payload, stock, c["grain_size"], 0.4, dst_texture=dst, print_film=prt, exp_kelvin=6000,
                                 color_masking=1.0, seed=c["seed"], frame_width=fw, frame_height=fh, halation=c["halation"],
                                 sharpness=c["mtf"], grain=c["grain"], halation_green_factor=c["green"], canvas_mode=c["canvas"],
                                 canvas_scale=c["canvas_scale"], canvas_ratio=c["canvas_ratio"])
    assert out is None
    has_canvas = c["canvas"] != "No"
    color = geometry.canvas_layout((payload["output_resolution"][1], payload["output_resolution"][0]), c["canvas"], c["canvas_scale"],
                                   c["canvas_ratio"])[1] if has_canvas else (255, 255, 255)
    t = geometry.blit_transform((pw, ph), (dw, dh), pipeline_resolution=payload["pipeline_resolution"],
                                output_resolution=payload["output_resolution"],
                                canvas_resolution=payload.get("canvas_resolution") if has_canvas else None, canvas_color=color)
    want = post.blit_rgba8(rendered, dh, dw, t)
    got = dst.cpu().numpy()
    np.testing.assert_array_equal(got[..., 3], want[..., 3])
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 2 and (d > 0).mean() <= 5e-3, (int(d.max()), float((d > 0).mean()), c)
