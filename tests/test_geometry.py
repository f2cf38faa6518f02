"""Host-side crop / canvas arithmetic vs vectors made by the reference's own crop_image / get_canvas_data."""

import os

import numpy as np
import pytest

from raw2film_amd import geometry


@pytest.fixture(scope="module")
def geo(golden_dir):
    return np.load(os.path.join(golden_dir, "geometry.npz"))


def test_crop_box_matches_reference(geo):
    for (h, w, aspect, zoom, flip), box in zip(geo["crop_cases"], geo["crop_boxes"]):
        got = geometry.crop_box(int(h), int(w), zoom=float(zoom), aspect=float(aspect), flip=bool(flip))
        assert got == tuple(int(v) for v in box), (h, w, aspect, zoom, flip)


def test_canvas_layout_matches_reference(geo):
    modes = [str(m) for m in geo["canvas_modes"]]
    for (h, w, mi, scale, ratio), out in zip(geo["canvas_cases"], geo["canvas_out"]):
        res, color, off = geometry.canvas_layout((int(h), int(w), 3), modes[int(mi)], float(scale), float(ratio))
        assert (res[0], res[1], *color, off[0], off[1]) == tuple(int(v) for v in out)


def test_add_canvas_numpy_and_torch_agree():
    torch = pytest.importorskip("torch")
    img = np.random.default_rng(0).integers(0, 256, (40, 60, 3), dtype=np.uint8)
    a = geometry.add_canvas(img, "Uniform black", 1.2)
    b = geometry.add_canvas(torch.from_numpy(img), "Uniform black", 1.2).numpy()
    np.testing.assert_array_equal(a, b)
    out, color, (oy, ox) = geometry.canvas_layout(img.shape, "Uniform black", 1.2)
    assert a.shape == (out[0], out[1], 3) and a[0, 0, 0] == 0 and color == (0, 0, 0)
    np.testing.assert_array_equal(a[oy:oy + 40, ox:ox + 60], img)
    assert geometry.add_canvas(img, "No") is img
    white = geometry.add_canvas(img, "Fixed white", 1.0, 1.0)
    assert white.shape == (60, 60, 3) and white[0, 0, 0] == 255


def test_crop_to_frame_quarter_turns():
    img = np.arange(400 * 600 * 3, dtype=np.float32).reshape(400, 600, 3)
    out = geometry.crop_to_frame(img, 36, 24, zoom=1.0, rotate_times=1)
    assert out.shape == (600, 400, 3)
    sq = geometry.crop_to_frame(img, 56, 56)
    assert sq.shape == (400, 400, 3) and sq[0, 0, 0] == img[0, 100, 0]


def test_rotation_plan_window_matches_reference():
    """tests/golden/rotate_crop.npz: the window effects.rotate keeps, and the centre / angle it hands to OpenCV."""
    import os

    from oracle import stages as st

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rotate_crop.npz"))["cases"]
    assert len(g) == 81
    for H, W, deg, oh, ow, r0, c0, cx, cy, angle, scale in g:
        H, W = int(H), int(W)
        inv, win = geometry.rotation_plan(H, W, deg)
        assert win == (r0, c0, oh, ow)
        assert (cx, cy, angle, scale) == (W / 2, H / 2, -deg, 1.0)
        # the product's inverse matrix and the oracle's (separate code) agree bit for bit, and invert the forward matrix
        fwd = st.rotation_matrix_2d((cx, cy), angle, scale)
        assert np.array_equal(inv, st.invert_affine(fwd))
        full = np.vstack([fwd, [0, 0, 1]]) @ np.vstack([inv, [0, 0, 1]])
        assert np.allclose(full, np.eye(3), atol=1e-9)
        if H <= 600:
            assert st.rotate(np.zeros((H, W, 3), np.float32), deg).shape[:2] == (oh, ow)


def test_oracle_warp_is_the_identity_for_the_identity_matrix_and_zero_outside():
    from oracle import stages as st

    img = np.random.default_rng(3).random((9, 13, 3)).astype(np.float32)
    eye = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(st.warp_affine_linear(img, eye), img)
    shifted = st.warp_affine_linear(img, np.array([[1, 0, 0.5], [0, 1, 0]], float))  # samples half a pixel to the right
    assert np.allclose(shifted[:, :-1], 0.5 * (img[:, :-1] + img[:, 1:]))
    assert np.allclose(shifted[:, -1], 0.5 * img[:, -1])  # the right neighbour is the constant border 0
    assert not st.warp_affine_linear(img, np.array([[1, 0, 100.0], [0, 1, 0]], float)).any()


def test_blit_transform_matches_the_reference_uniform_block(golden_dir):
    """geometry.blit_transform against the twelve floats GpuProcessor._bind_copy_to_dst hands to copy_to_int.wgsl
    (tools/make_golden_blit.py executed the reference's own method): 240 source / destination / output / canvas / colour cases."""
    import os

    import numpy as np

    from raw2film_amd import geometry

    g = np.load(os.path.join(golden_dir, "blit_transform.npz"))
    for case, want in zip(g["cases"], g["uniforms"]):
        src, dst = tuple(case[0:2]), tuple(case[2:4])
        out_res = None if case[4] < 0 else tuple(case[4:6])
        can_res = None if case[6] < 0 else tuple(case[6:8])
        color = None if case[8] < 0 else tuple(case[8:11])
        t = geometry.blit_transform(src, dst, pipeline_resolution=src, output_resolution=out_res, canvas_resolution=can_res,
                                    canvas_color=color)
        got = np.array([t["scale_x"], t["scale_y"], t["offset_x"], t["offset_y"], t["canvas_min_x"], t["canvas_min_y"],
                        t["canvas_max_x"], t["canvas_max_y"], *t["canvas_color"], 0.0], dtype=np.float32)
        np.testing.assert_array_equal(got, want)


def test_add_canvas_matches_the_reference_function():
    """tests/golden/add_canvas.npz: effects.add_canvas itself, run by tools/make_golden_canvas.py, for every canvas mode."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "add_canvas.npz"))
    modes = [str(m) for m in g["modes"]]
    for i in range(int(g["n"])):
        h, w, mi, scale, ratio = g[f"case_{i}"]
        got = geometry.add_canvas(g[f"image_{i}"], modes[int(mi)], float(scale), float(ratio))
        want = g[f"out_{i}"]
        assert got.shape == want.shape and got.dtype == want.dtype, (i, modes[int(mi)])
        np.testing.assert_array_equal(got, want)


def test_payload_geometry_follows_the_reference_phase_one(golden_dir):
    """output / canvas / pipeline resolution and the payload's shape as the reference's own extract_image_data_cpu derives them
    (tools/make_golden_payload.py, gpu_processor.py:715-783): 3 344 cases, among them frames finer than `max_scale` WITH a
    canvas mode -- the canvas is laid out for the un-shrunk output size, not for the pipeline's (:764-771)."""
    from raw2film_amd.hip_processor import HipProcessor

    g = np.load(os.path.join(golden_dir, "payload_geometry.npz"))
    modes = [str(m) for m in g["modes"]]
    proc = HipProcessor.__new__(HipProcessor)  # phase 1 touches no instance state (and needs no GPU)
    shrunk_with_canvas = 0
    frames = {}
    for case, want in zip(g["cases"], g["results"]):
        H, W, fw, fh, r0, r1, ms, mi, cs, cr = case
        H, W = int(H), int(W)
        frame = frames.setdefault((H, W), np.zeros((H, W, 3), dtype=np.float32))
        p = proc.extract_image_data_cpu(frame, frame_width=fw, frame_height=fh, resolution=None if r0 < 0 else (int(r0), int(r1)),
                                        max_scale=None if ms < 0 else ms, canvas_mode=modes[int(mi)], canvas_scale=cs, canvas_ratio=cr)
        got = list(p["output_resolution"]) + list(p["canvas_resolution"] or (-1, -1)) + list(p["pipeline_resolution"])
        assert got == list(want[:6]), (case, got, want)
        # the payload frame itself is scaled on the device in phase 2: its shape there is the pipeline's
        rows, cols = (p["resize_to"] or p["image_array"].shape[:2])
        assert (rows, cols) == (want[6], want[7]) == (p["pipeline_resolution"][1], p["pipeline_resolution"][0])
        assert p["image_array"].shape[2] == want[8] == 4
        shrunk_with_canvas += int(p["canvas_resolution"] is not None and tuple(p["output_resolution"]) != tuple(p["pipeline_resolution"]))
    assert shrunk_with_canvas > 200
