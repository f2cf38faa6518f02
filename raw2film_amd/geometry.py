"""Host-side frame geometry on either side of the render path: aspect crop / zoom / quarter turns
before it, canvas (letterbox border) after it.  Pure index arithmetic, pinned against the
reference's own functions through tests/golden/geometry.npz.

  crop_box / crop_to_frame   <->  effects.crop_image, raw_conversion.crop_rotate_zoom   effects.py:77-111, raw_conversion.py:56-72
  canvas_layout / add_canvas <->  effects.get_canvas_data, effects.add_canvas            effects.py:290-357

  rotation_plan              <->  effects.rotate (matrix handed to cv.warpAffine + the centred crop)  effects.py:46-75
The interpolating resample of the free rotation itself runs on the device (r2f_warp_affine).
"""

from __future__ import annotations

import math

import numpy as np

CANVAS_MODES = ("No", "Proportional white", "Proportional black", "Uniform white", "Uniform black", "Fixed white", "Fixed black")


def crop_box(rows: int, cols: int, zoom: float = 1, aspect: float = 1.5, flip: bool = False) -> tuple[int, int, int, int]:
    """(row0, col0, n_rows, n_cols) that effects.crop_image keeps (effects.py:77-111).

    The longer side is matched to `aspect` (inverted when `flip`), centred with ceil() at both ends;
    then `zoom` > 1 trims ceil((zoom-1)/(2 zoom) * extent) from every side."""
    if flip:
        aspect = 1 / aspect
    r0, r1, c0, c1 = 0, rows, 0, cols
    if rows > cols:
        if rows > aspect * cols:
            r0, r1 = math.ceil(rows / 2 - cols * aspect / 2), math.ceil(rows / 2 + cols * aspect / 2)
        else:
            c0, c1 = math.ceil(cols / 2 - rows / aspect / 2), math.ceil(cols / 2 + rows / aspect / 2)
    elif cols > aspect * rows:
        c0, c1 = math.ceil(cols / 2 - rows * aspect / 2), math.ceil(cols / 2 + rows * aspect / 2)
    else:
        r0, r1 = math.ceil(rows / 2 - cols / aspect / 2), math.ceil(rows / 2 + cols / aspect / 2)
    r0, c0 = max(r0, 0), max(c0, 0)
    r1, c1 = min(r1, rows), min(c1, cols)
    if zoom > 1:
        f = (zoom - 1) / (2 * zoom)
        tr, tc = math.ceil(f * (r1 - r0)), math.ceil(f * (c1 - c0))
        r0, r1, c0, c1 = r0 + tr, r1 - tr, c0 + tc, c1 - tc
    return r0, c0, r1 - r0, c1 - c0


def crop_to_frame(image: np.ndarray, frame_width: float = 36, frame_height: float = 24, zoom: float = 1.0,
                  rotate_times: int = 0, flip: bool = False) -> np.ndarray:
    """raw_conversion.crop_rotate_zoom without the free rotation: aspect crop (honouring `flip`), zoom crop at
    the frame aspect, then `rotate_times` quarter turns (raw_conversion.py:66-70)."""
    aspect = frame_width / frame_height
    r0, c0, nr, nc = crop_box(image.shape[0], image.shape[1], 1, aspect, flip)
    image = image[r0:r0 + nr, c0:c0 + nc]
    r0, c0, nr, nc = crop_box(image.shape[0], image.shape[1], zoom, aspect, False)
    image = image[r0:r0 + nr, c0:c0 + nc]
    return np.rot90(image, k=rotate_times)


def rotation_plan(rows: int, cols: int, degrees: float):
    """What effects.rotate (effects.py:46-75) does to a rows x cols frame, as data:

    (m_dst_to_src, (row0, col0, out_rows, out_cols)) -- the inverse of cv.getRotationMatrix2D((cols/2, rows/2), -degrees, 1)
    (what cv.warpAffine samples with; 2 x 3 float64, OpenCV's invertAffineTransform arithmetic) and the centred window
    kept afterwards: the largest axis-parallel rectangle of the frame's aspect inside the rotated frame."""
    cx, cy = cols / 2, rows / 2  # tuple(np.array(rgb.shape[1::-1]) / 2)
    angle = -degrees * math.pi / 180  # getRotationMatrix2D: degrees -> radians, scale 1
    alpha, beta = math.cos(angle), math.sin(angle)
    m = np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]])
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]  # invertAffineTransform
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    inv = np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]], [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]])
    aspect = rows / cols
    ang = math.fabs(degrees) * math.pi / 180
    if aspect < 1:
        total, aspect, switch = rows, 1 / aspect, True
    else:
        total, switch = cols, False
    w = total / (aspect * math.sin(ang) + math.cos(ang))
    h = w * aspect
    if switch:
        w, h = h, w
    crop_r, crop_c = int((rows - h) // 2), int((cols - w) // 2)
    return inv, (crop_r, crop_c, max(rows - 2 * crop_r, 0), max(cols - 2 * crop_c, 0))


def canvas_layout(shape, canvas_mode: str, canvas_scale: float = 1.0, canvas_ratio: float = 1.0):
    """((out_rows, out_cols), (r, g, b), (row_offset, col_offset)) of effects.get_canvas_data (effects.py:290-333).
    "Proportional" replaces the ratio by the image's own (so it always takes the second branch), "Fixed" uses
    `canvas_ratio`, "Uniform" adds int(max_side * (scale - 1)) to both sides."""
    rows, cols = int(shape[0]), int(shape[1])
    color = (255, 255, 255) if "white" in canvas_mode else (0, 0, 0) if "black" in canvas_mode else (128, 128, 128)
    if "Uniform" in canvas_mode:
        border = int(max(rows, cols) * (canvas_scale - 1))
        out = (rows + border, cols + border)
    elif "Proportional" in canvas_mode or "Fixed" in canvas_mode:
        ratio = cols / rows if "Proportional" in canvas_mode else canvas_ratio
        if cols / rows > ratio:
            out = (int(cols / ratio * canvas_scale), int(cols * canvas_scale))
        else:
            out = (int(rows * canvas_scale), int(rows * ratio * canvas_scale))
    else:
        raise ValueError(f"unknown canvas mode {canvas_mode!r}")
    offset = ((out[0] - rows) // 2, (out[1] - cols) // 2)
    return out, color, offset


def add_canvas(image, canvas_mode: str, canvas_scale: float = 1.0, canvas_ratio: float = 1.0):
    """effects.add_canvas (effects.py:336-357) for a uint8 (H, W, 3) NumPy array or torch tensor (any device):
    the frame pasted at `offset` onto a canvas of the mode's colour."""
    if canvas_mode == "No":
        return image
    out, color, (oy, ox) = canvas_layout(image.shape, canvas_mode, canvas_scale, canvas_ratio)
    h, w = int(image.shape[0]), int(image.shape[1])
    if oy < 0 or ox < 0:
        raise ValueError("canvas smaller than the frame (canvas_scale < 1)")
    if isinstance(image, np.ndarray):
        canvas = np.empty((out[0], out[1], 3), dtype=np.uint8)
        canvas[...] = np.asarray(color, dtype=np.uint8)
        canvas[oy:oy + h, ox:ox + w] = image
        return canvas
    import torch

    canvas = torch.empty((out[0], out[1], 3), dtype=torch.uint8, device=image.device)
    canvas[...] = torch.tensor(color, dtype=torch.uint8, device=image.device)
    canvas[oy:oy + h, ox:ox + w] = image
    return canvas


def blit_transform(src_size, dst_size, pipeline_resolution=None, output_resolution=None, canvas_resolution=None,
                   canvas_color=(255, 255, 255)) -> dict:
    """The uniform block of shaders/copy_to_int.wgsl as GpuProcessor._bind_copy_to_dst computes it
    (gpu_processor.py:1416-1515): where the rendered frame (src_size = (w, h)) lands inside a destination of dst_size = (w, h)
    -- letterboxed to the aspect of the canvas (or of the output / pipeline resolution), the image itself scaled by
    output / canvas inside the canvas area -- plus the canvas bounds and colour.  All sizes are (width, height) like upstream's."""
    src_w, src_h = src_size
    dst_w, dst_h = dst_size
    has_canvas = canvas_resolution is not None and canvas_resolution[0] > 0
    if has_canvas:
        content_w, content_h = canvas_resolution
    elif output_resolution is not None and output_resolution[0] > 0:
        content_w, content_h = output_resolution
    else:
        content_w, content_h = pipeline_resolution if pipeline_resolution is not None else (src_w, src_h)
    src_aspect = content_w / content_h
    dst_aspect = dst_w / dst_h
    if src_aspect > dst_aspect:
        canvas_render_w, canvas_render_h = dst_w, dst_w / src_aspect
        canvas_offset_x, canvas_offset_y = 0.0, (dst_h - canvas_render_h) / 2.0
    else:
        canvas_render_w, canvas_render_h = dst_h * src_aspect, dst_h
        canvas_offset_x, canvas_offset_y = (dst_w - canvas_render_w) / 2.0, 0.0
    if has_canvas:
        cmin_x, cmin_y = canvas_offset_x, canvas_offset_y
        cmax_x, cmax_y = canvas_offset_x + canvas_render_w, canvas_offset_y + canvas_render_h
    else:
        cmin_x = cmin_y = cmax_x = cmax_y = 0.0
    raw = canvas_color if canvas_color is not None else (255, 255, 255)
    color = tuple(c / 255.0 if max(raw) > 1.0 else float(c) for c in raw[:3])
    if canvas_resolution is not None and output_resolution is not None:
        target_w, target_h = output_resolution
        render_w = canvas_render_w * (target_w / content_w)
        render_h = canvas_render_h * (target_h / content_h)
        offset_x = canvas_offset_x + (canvas_render_w - render_w) / 2.0
        offset_y = canvas_offset_y + (canvas_render_h - render_h) / 2.0
    else:
        render_w, render_h = canvas_render_w, canvas_render_h
        offset_x, offset_y = canvas_offset_x, canvas_offset_y
    return {"scale_x": 1.0 / render_w, "scale_y": 1.0 / render_h, "offset_x": offset_x, "offset_y": offset_y,
            "canvas_min_x": cmin_x, "canvas_min_y": cmin_y, "canvas_max_x": cmax_x, "canvas_max_y": cmax_y,
            "canvas_color": color}
