"""HipProcessor: the MI355X backend behind raw2film's processor surface.

Mirrors `CpuProcessor.process` (cpu_processor.py:269-414) / `GpuProcessor.process`,
`.extract_image_data_cpu`, `.process_preloaded` (gpu_processor.py:715-783, 1541-1693): same
keyword names and defaults, unknown keywords swallowed, LUTs / stencils rebuilt and
re-uploaded only when their parameter dict changes (cpu_processor.py:104-105,157-158,...).

Scope: the post-decode per-pixel path plus what sits either side of it: aspect crop, zoom,
quarter turns, free rotation, the preview / `max_scale` down-scale and the pre-path chroma NR before; the highlight
burn (S7) inside; uint8, the LANCZOS4 way back from `max_scale`, canvas and histogram after.  RAW decoding and lens
correction (lensfun) are out of scope: a call that would need them -- `lens_correction=True` WITH a camera and a lens, the
only case in which the reference corrects anything (effects.py:22-30) -- raises NotImplementedError instead of silently
rendering something else.

`src` is therefore a decoded frame: a float32 (H, W, 3|4) array / CUDA tensor in linear CIE XYZ
(what `raw_to_linear` returns, raw_conversion.py:33-53), or the path of a `.npy` file holding one.
"""

from __future__ import annotations

import math
import random

import numpy as np

from . import _lib, filmstock, geometry, stencils
from .context import LOG_EPS, LUT3D_SCALE, HipContext

REC709_TO_XYZ = np.array(  # data.py:128-135
    [[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]],
    dtype=np.float32,
)


class PendingFrame:
    """A frame submitted with HipProcessor.submit_preloaded: `.result()` waits for its download and returns the uint8 array."""

    def __init__(self, host, done, array=None):
        self._host, self._done, self._array = host, done, array

    @classmethod
    def finished(cls, array):
        """A frame that is back already (a payload that streamed through the pipeline: submit_preloaded)."""
        return cls(None, None, array)

    def ready(self) -> bool:
        return self._array is not None or self._done.query()

    def result(self) -> np.ndarray:
        if self._array is not None:
            return self._array
        self._done.synchronize()
        return self._host.numpy()


class HipProcessor:
    """Drop-in for the hot path of CpuProcessor / GpuProcessor (gui.py:1584-1585)."""

    def __init__(self, cameras=None, lenses=None, device: int = 0, payload_alpha: bool = True, result_buffers: int = 0,
                 lib_path: str | None = None):
        """payload_alpha: extract_image_data_cpu appends the constant alpha plane like upstream (gpu_processor.py:765: its wgpu
        texture is rgba32float).  Nothing on this backend reads it -- the device path takes 3- and 4-channel frames alike -- and
        batch export is bound by the upload of the payload: with payload_alpha=False the frame crosses PCIe a quarter smaller.
        result_buffers: see _download (pinned result buffers for interactive use).
        lib_path: another build of libr2f_hip.so for this processor's context (development A/B; default: the in-tree library)."""
        import torch

        self._torch = torch
        self.cameras = cameras
        self.lenses = lenses
        self.payload_alpha = bool(payload_alpha)
        self.result_buffers = int(result_buffers)  # 0: process() returns a fresh array; n: views of n pinned buffers in turn (_download)
        # process(host array, cache=False) with pinned result buffers streams a large frame through the pipeline in row bands while it
        # is still arriving (_process_streamed): at most this many, of at least 512 rows each (100 MP: 16 bands of 512 rows = 23.3 ms
        # against 24.7 with 8, 27.2 with 4, 24.9 with 24 -- tools/stream_bands_probe.py); 0: upload, render, download one after the other
        self.stream_bands = 16
        self.stream_taper = 2  # ... and the last two of them are halved (what runs behind the last byte of the upload gets shorter)
        self.ctx = HipContext(device, lib_path=lib_path)
        self.device = self.ctx.device  # NB: a torch device, not a wgpu device (gui.py:1652 uses bitmap mode)
        # comparison dicts, same role as cpu_processor.py:41-45 / gpu_processor.py
        self.input_param_dict = None
        self.curve_param_dict = None
        self.output_param_dict = None
        self.halation_param_dict = None
        self.mtf_param_dict = None
        self.grain_kernel_param_dict = None
        self.grain_lut_param_dict = None
        self.highlight_burn_param_dict = None
        self.pipeline_resolution = None  # (w, h) of the frame prepared last
        self.matrix_key = None
        self.uploads = 0  # number of table uploads, for the caching tests
        self.last_output = None  # device uint8 (H, W, 3) of the last process()/process_preloaded(): histogram source

    def close(self):
        pool = getattr(self, "_copy_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._copy_pool = None
        self._stream_bufs = None
        self._texture = None  # the frame kept on the device for re-renders
        self._texture_src = None
        self.image_param_dict = None
        self.last_output = None
        self.ctx.close()

    # ------------------------------------------------------------------ cached loaders
    def load_input_lut(self, negative_film, exp_kelvin, tint, exp_comp):
        """cpu_processor.py:142-164"""
        new = {"negative_film": negative_film.name, "exp_kelvin": exp_kelvin, "tint": tint, "exp_comp": exp_comp}
        if new == self.input_param_dict:
            return
        self.ctx.set_lut2d(negative_film.get_input_lut(exp_kelvin, tint, exp_comp))
        self.uploads += 1
        self.input_param_dict = new

    def load_density_curve(self, negative_film, push_pull, color_masking=None):
        """cpu_processor.py:166-188"""
        new = {"negative_film": negative_film.name, "push_pull": push_pull, "color_masking": color_masking}
        if new == self.curve_param_dict:
            return
        self.ctx.set_curve1d(negative_film.get_density_curve(push_pull=push_pull, color_masking=color_masking))
        self.uploads += 1
        self.curve_param_dict = new

    def load_output_lut(self, negative_film, print_film=None, red_light=0.0, green_light=0.0, blue_light=0.0,
                        projector_kelvin=6500, shadow_comp=0.0, sat_adjust=1.0, gamma_func="sRGB",
                        inversion_gamma=4.0, idealized_curve=False, inversion=False, white_balance=False,
                        white_clip=False, icc_transform=None, color_masking=None):
        """cpu_processor.py:190-267"""
        new = {
            "negative_film": negative_film.name,
            "print_film": print_film.name if print_film is not None else None,
            "red_light": red_light, "green_light": green_light, "blue_light": blue_light,
            "projector_kelvin": projector_kelvin, "shadow_comp": shadow_comp, "sat_adjust": sat_adjust,
            "gamma_func": gamma_func, "inversion_gamma": inversion_gamma, "idealized_curve": idealized_curve,
            "inversion": inversion, "white_balance": white_balance, "white_clip": white_clip,
            "icc_transform": icc_transform, "color_masking": color_masking,
        }
        if new == self.output_param_dict:
            return
        lut = filmstock.create_lut(
            negative_film, print_film, mode="print", input_colorspace=None, adx_coding=False, cube=False,
            red_light=red_light, green_light=green_light, blue_light=blue_light, projector_kelvin=projector_kelvin,
            shadow_comp=shadow_comp, sat_adjust=sat_adjust, gamma_func=gamma_func, inversion_gamma=inversion_gamma,
            idealized_curve=idealized_curve, inversion=inversion, white_balance=white_balance, white_clip=white_clip,
            linear_scaling=4.0, color_masking=color_masking,
        )
        if icc_transform is not None:  # cpu_processor.py:255-263: 8-bit ICC pass over the LUT itself
            from PIL import Image, ImageCms

            shape = lut.shape
            img = Image.fromarray((lut * 255).astype(np.uint8).reshape(shape[0], -1, shape[-1]))
            ImageCms.applyTransform(img, icc_transform, inPlace=True)
            lut = (np.array(img, np.uint8).reshape(shape) / 255.0).astype(np.float32)
        self.ctx.set_lut3d(lut)
        self.uploads += 1
        self.output_param_dict = new

    def load_halation_kernel(self, scale, halation_size=1.0, halation_red_factor=1.0, halation_green_factor=0.4,
                             halation_blue_factor=0.0, halation_intensity=1.0, bw=False):
        """gpu_processor.py:818-854"""
        new = {"scale": scale, "halation_size": halation_size, "halation_red_factor": halation_red_factor,
               "halation_green_factor": halation_green_factor, "halation_blue_factor": halation_blue_factor,
               "halation_intensity": halation_intensity, "bw": bw}
        if new == self.halation_param_dict:
            return
        k = stencils.halation_stencil(scale, halation_size, halation_red_factor, halation_green_factor,
                                      halation_blue_factor, halation_intensity, bw=bw)
        self.ctx.set_kernel(_lib.KERNEL_HALATION, k)
        self._halation_reach = stencils.vertical_reach(k)  # (rows above, rows below): what a row band needs of its neighbours
        self.uploads += 1
        self.halation_param_dict = new

    def load_mtf_kernel(self, negative_film, scale, sharpening_strength, sharpening_sigma):
        """gpu_processor.py:792-816"""
        new = {"negative_film": negative_film.name, "scale": scale, "sharpening_strength": sharpening_strength,
               "sharpening_sigma": sharpening_sigma}
        if new == self.mtf_param_dict:
            return
        k = stencils.mtf_stencil(negative_film, scale, sharpening_strength, sharpening_sigma)
        self.ctx.set_kernel(_lib.KERNEL_MTF, k)
        self._mtf_reach = stencils.vertical_reach(k)
        self.uploads += 1
        self.mtf_param_dict = new

    def load_grain(self, negative_film, scale, grain_size_mm=0.01, grain_sigma=0.4, bw_grain=False):
        """gpu_processor.py:904-936 (the seed is an explicit argument of the render here)"""
        new_lut = {"negative_film": negative_film.name, "scale": scale, "bw_grain": bw_grain}
        if new_lut != self.grain_lut_param_dict:
            self.ctx.set_grain_lut(negative_film.get_grain_curve(scale, adx=False, bw_grain=bw_grain))
            self.uploads += 1
            self.grain_lut_param_dict = new_lut
        new = {"scale": scale, "grain_size_mm": grain_size_mm, "grain_sigma": grain_sigma, "bw_grain": bw_grain}
        if new == self.grain_kernel_param_dict:
            return
        k = filmstock.grain_kernel(1 / scale, grain_size_mm=grain_size_mm, grain_sigma=grain_sigma)
        if k is None:
            k = np.ones((1, 1), dtype=np.float32)  # gpu_processor.py:931-932
        self.ctx.set_kernel(_lib.KERNEL_GRAIN, k)
        self.uploads += 1
        self.grain_kernel_param_dict = new

    # ------------------------------------------------------------------ phase 1 (host)
    def load_highlight_burn(self, negative_film, highlight_burn, burn_scale, pipeline_resolution=None):
        """Parameters of the highlight burn for the frame size at hand -- the counterpart of GpuProcessor.load_highlight_burn
        (gpu_processor.py:856-878): reference density, strength, the low-resolution grid.  Nothing is uploaded here: the device
        side takes them by value in r2f_params (the reference keeps them in a uniform buffer).  `pipeline_resolution` = (w, h);
        default: the frame prepared last (the reference reads self.pipeline_resolution).  Returns the dict it keeps in
        `highlight_burn_param_dict`; "cell" is the size in pixels of a low-resolution cell (effects.py:365)."""
        if pipeline_resolution is None:
            pipeline_resolution = self.pipeline_resolution
        if pipeline_resolution is None:
            raise ValueError("load_highlight_burn: no pipeline_resolution (prepare a frame first or pass (w, h))")
        w, h = int(pipeline_resolution[0]), int(pipeline_resolution[1])
        d_ref = negative_film.d_ref[1 if len(negative_film.d_ref) > 1 else 0]  # effects.py:406
        cell = math.ceil(min(w, h) / burn_scale)
        self.highlight_burn_param_dict = {"d_ref": d_ref, "highlight_burn": highlight_burn, "lowres_w": max(1, w // cell),
                                          "lowres_h": max(1, h // cell), "cell": cell}
        return self.highlight_burn_param_dict

    def read_texture(self, texture):
        """A device "texture" (an (h, w, 4) or (h, w, 3) uint8 tensor, e.g. `dst_texture` after process()) as an (h, w, 3) uint8
        array on the host -- GpuProcessor.read_texture (gpu_processor.py:1311-1356)."""
        torch = self._torch
        if not (isinstance(texture, torch.Tensor) and texture.is_cuda and texture.dtype == torch.uint8 and texture.dim() == 3
                and texture.shape[2] in (3, 4)):
            raise NotImplementedError("read_texture: wgpu textures are not supported; pass a uint8 (h, w, 3 or 4) CUDA tensor")
        return self._download(texture[:, :, :3].contiguous())

    def extract_image_data_cpu(self, src, cam=None, lens=None, lens_correction=True, frame_width=36, frame_height=24,
                               rotation=0.0, zoom=1.0, rotate_times=0, flip=False, resolution=None, half_size=True,
                               cache=True, chroma_nr=0, max_scale=400.0, canvas_mode="No", canvas_scale=1.0,
                               canvas_ratio=1.0, exposure=None, metadata=None, **kwargs):
        """PHASE 1 of the two-phase batch API (gpu_processor.py:715-783): pure host work, touches
        no instance state.  Returns the same payload dict; `image_array` is (H, W, 4) float32 ((H, W, 3) with payload_alpha=False)."""
        if lens_correction and cam is not None and lens is not None:
            # the reference corrects only when both are given (cpu_processor.py:107-108, effects.py:22-30); lensfun is not
            # part of the accelerated path, and rendering an uncorrected frame in its place would be a silent difference
            raise NotImplementedError("lens correction (lensfunpy, effects.py:22-43) is outside the accelerated path: "
                                      "pass lens_correction=False or no cam / lens")
        # _internal (load_image_texture: the payload never leaves this object): no alpha plane, and the clamp of
        # gpu_processor.py:275 runs on the device after the upload instead of as a 12 B/px pass over host memory
        internal = bool(kwargs.get("_internal"))
        image = self._load_decoded(src, clip=not internal)
        u16_factor = None
        if image.dtype == np.uint16:
            # raw_to_linear's tail (raw_conversion.py:50-52) moves to the device: the auto exposure is measured here, on the
            # whole decoded frame like upstream (before any crop), unless the caller brings the stops along
            from . import decode

            stops = decode.auto_exposure(image, metadata=metadata) if exposure is None else float(exposure)
            u16_factor = float(decode.exposure_factor(stops))
        aspect = frame_width / frame_height
        warp = None
        if rotation:
            # raw_conversion.crop_rotate_zoom (raw_conversion.py:56-72) with the interpolating part deferred to the device:
            # aspect crop here (a view), then phase 2 warps straight into the window that effects.rotate's centred crop
            # and the zoom crop keep, and applies the quarter turns
            r0, c0, nr, nc = geometry.crop_box(image.shape[0], image.shape[1], 1, aspect, flip)
            image = image[r0:r0 + nr, c0:c0 + nc]
            m_inv, (wr0, wc0, wnr, wnc) = geometry.rotation_plan(nr, nc, rotation)
            zr0, zc0, znr, znc = geometry.crop_box(wnr, wnc, zoom, aspect, False)
            if znr <= 0 or znc <= 0:
                raise ValueError(f"rotation {rotation} / zoom {zoom} leave an empty frame")
            warp = {"m_dst_to_src": m_inv, "window": (wr0 + zr0, wc0 + zc0, znr, znc), "rotate_times": int(rotate_times) % 4}
            h, w = (znc, znr) if warp["rotate_times"] % 2 else (znr, znc)
        else:
            # aspect crop / zoom / quarter turns: index arithmetic of raw_conversion.crop_rotate_zoom (raw_conversion.py:56-72)
            image = geometry.crop_to_frame(image, frame_width, frame_height, zoom, rotate_times, flip)
            h, w = image.shape[:2]
        # cpu_processor.py:119-134: without a preview resolution the frame's own size is the target; a target finer than
        # `max_scale` px/mm is rendered at max_scale and scaled back up at the very end (cpu_processor.py:411-412)
        if resolution is None and max_scale is not None:
            resolution = (h, w)
        resize_to, upscale_to = None, None
        # what CpuProcessor.load_image returns as `orig_resolution` (cpu_processor.py:122) and process() hands to the final
        # resolution_scaling (cpu_processor.py:411-412)
        final_resolution = (int(resolution[0]), int(resolution[1])) if resolution is not None else None
        scale_factor = 1.0
        if resolution is not None:
            resolution = (int(resolution[0]), int(resolution[1]))
            scale = max(resolution) / max(frame_width, frame_height)
            if max_scale is not None and scale > max_scale:
                scale_factor = max_scale / scale
                upscale_to = resolution
                resolution = tuple(round(x * scale_factor) for x in resolution)
            # utils.resolution_scaling (utils.py:226-244), applied on the device in phase 2
            factor = min(resolution[0] / h, resolution[1] / w)
            if factor != 1:
                # cv.resize(dsize=(round(w f), round(h f))): INTER_AREA down, INTER_LANCZOS4 up (a preview larger than the frame)
                resize_to = (round(h * factor), round(w * factor))
                h, w = resize_to
        # gpu_processor.py:764: the size the frame has once it is back from the max_scale pipeline -- the UN-shrunk output size,
        # which is also what the canvas is laid out for (:767-771), not the pipeline's
        out_h, out_w = (round(x / scale_factor) for x in (h, w))
        if upscale_to is not None and min(upscale_to[0] / h, upscale_to[1] / w) <= 1:
            upscale_to = None  # (the uint8 result goes back up with LANCZOS4 only when that enlarges it)
        canvas_res = None
        if canvas_mode != "No":  # gpu_processor.py:767-771
            res, _, _ = geometry.canvas_layout((out_h, out_w), canvas_mode, canvas_scale, canvas_ratio)
            canvas_res = (res[1], res[0])
        alpha = getattr(self, "payload_alpha", True) and not internal
        if u16_factor is not None:
            image = np.ascontiguousarray(image[..., :3])
        elif image.shape[2] == 3 and alpha:
            image = np.concatenate([image, np.ones_like(image[..., :1])], axis=-1)  # gpu_processor.py:765
        elif image.shape[2] == 4 and not alpha:
            image = image[..., :3]
        if u16_factor is None:
            image = np.ascontiguousarray(image, dtype=np.float32)
        return {
            "image_array": image,
            "u16_factor": u16_factor,  # float32 exposure factor of a uint16 payload (converted on the device), else None
            "clip_on_device": internal and u16_factor is None,  # the float frame still has to be clamped to [0, 65504]
            "final_resolution": final_resolution,
            "output_resolution": (out_w, out_h),
            "canvas_resolution": canvas_res,
            "pipeline_resolution": (w, h),
            # upstream filters on the host here (gpu_processor.py:750-751); this backend filters on the device in phase 2
            "chroma_nr": int(chroma_nr),
            "resize_to": resize_to,  # (rows, cols) of the INTER_AREA down-scale still to be applied, or None
            "warp": warp,  # free rotation still to be applied (first of the device pre-path steps), or None
            "upscale_to": upscale_to,  # (rows, cols) the rendered uint8 frame is LANCZOS4-scaled back into, or None
        }

    @staticmethod
    def _load_decoded(src, clip=True):
        if isinstance(src, np.ndarray):
            image = src
        elif isinstance(src, str) and src.lower().endswith(".npy"):
            image = np.load(src)
        elif isinstance(src, str):
            raise NotImplementedError(
                f"{src!r}: RAW decoding (LibRaw/rawpy, raw_conversion.py:33-53) is outside the accelerated path; "
                "pass the decoded linear-XYZ frame (array or .npy)"
            )
        else:
            raise TypeError(f"unsupported src type {type(src)!r}")
        if image.ndim != 3 or image.shape[2] not in (3, 4):
            raise ValueError(f"decoded frame must be (H, W, 3|4), got {image.shape}")
        if image.dtype == np.uint16:  # LibRaw's 16-bit output: converted on the device (decode.py, r2f_decode_u16)
            return image
        image = np.asarray(image, dtype=np.float32)
        return np.clip(image, 0, 65504) if clip else image  # gpu_processor.py:275 (clip=False: clamped after the upload)

    # ------------------------------------------------------------------ the operator surface
    def process(self, src, negative_film, grain_size, grain_sigma, dst_texture=None, histogram_texture=None,
                lens_correction=True, print_film=None, exp_comp=0.0, red_light=0.0, green_light=0.0, blue_light=0.0,
                projector_kelvin=6500, shadow_comp=0.0, sat_adjust=1.0, gamma_func="sRGB", exp_kelvin=6500, tint=0.0,
                inversion_gamma=4.0, idealized_curve=False, inversion=False, push_pull=0.0, white_balance=False,
                white_clip=False, icc_transform=None, resolution=None, frame_width=36, frame_height=24, rotation=0.0,
                zoom=1.0, rotate_times=0, flip=False, cam=None, lens=None, canvas_mode="No", canvas_scale=1.0,
                canvas_ratio=1.0, halation_intensity=1.0, halation=True, halation_size=1.0, halation_green_factor=0.4,
                sharpness=True, sharpening_strength=0.0, sharpening_sigma=1.0, chroma_nr=0, grain=2,
                highlight_burn=0.0, burn_scale=50.0, half_size=True, cache=True, color_masking=None, max_scale=400.0,
                seed=None, exposure=None, metadata=None, src_version=None, **_):
        """Load (decoded) frame and render it: np.uint8 (H, W, 3), like cpu_processor.py:414 -- including the CPU processor's
        last step, resolution_scaling of the finished (canvas-framed) frame to the requested resolution (cpu_processor.py:411-412).
        With `dst_texture` (a uint8 (h, w, 4) CUDA tensor standing in for the preview widget's wgpu texture) the call behaves
        like GpuProcessor.process with a destination (gpu_processor.py:1865-1890): the frame is letterboxed into it on the
        device, `histogram_texture` (same kind of tensor) receives the histogram image, and None is returned.
        A uint16 `src` is LibRaw's 16-bit output before raw_to_linear's last two lines (raw_conversion.py:50-52): those run on
        the device with the auto exposure measured on the host (`metadata`: the EXIF dict calc_exposure reads) or given in stops
        (`exposure`).
        An array `src` that is the array of the previous call is taken for the same frame when a fingerprint of it agrees (shape,
        dtype, address and a checksum of up to 32 evenly spaced rows -- an edit confined to other rows is NOT seen: pass
        cache=False after a partial in-place edit, or `src_version`, any hashable token of the caller's that changes whenever
        the buffer's content does and then replaces the checksum)."""
        if dst_texture is not None:
            self._check_texture(dst_texture, "dst_texture")
        if histogram_texture is not None:
            self._check_texture(histogram_texture, "histogram_texture")
        # profile_stages (bench.py's host_device_copies): wall clock per stage of this call, with a device synchronisation behind
        # each -- a measuring mode (the syncs serialise what otherwise overlaps), off by default
        prof = getattr(self, "profile_stages", False)
        if prof:
            import time

            self._torch.cuda.synchronize(self.device)
            t_start = time.perf_counter()
        elif (not cache and self.stream_bands > 1 and dst_texture is None and isinstance(src, np.ndarray) and src.size >= (1 << 24)
              and not rotation and not chroma_nr and canvas_mode == "No" and not highlight_burn):
            # a large host frame that is uploaded for this one render: streamed through the pipeline in row bands while it arrives
            res = self._process_streamed(
                src, negative_film, grain_size, grain_sigma,
                load=dict(cam=cam, lens=lens, lens_correction=lens_correction, frame_width=frame_width, frame_height=frame_height,
                          rotation=rotation, zoom=zoom, rotate_times=rotate_times, flip=flip, resolution=resolution, half_size=half_size,
                          cache=cache, chroma_nr=chroma_nr, max_scale=max_scale, canvas_mode=canvas_mode, canvas_scale=canvas_scale,
                          canvas_ratio=canvas_ratio, exposure=exposure, metadata=metadata),
                print_film=print_film, exp_comp=exp_comp, red_light=red_light, green_light=green_light, blue_light=blue_light,
                projector_kelvin=projector_kelvin, shadow_comp=shadow_comp, sat_adjust=sat_adjust, gamma_func=gamma_func,
                exp_kelvin=exp_kelvin, tint=tint, inversion_gamma=inversion_gamma, idealized_curve=idealized_curve, inversion=inversion,
                push_pull=push_pull, white_balance=white_balance, white_clip=white_clip, icc_transform=icc_transform,
                frame_width=frame_width, frame_height=frame_height, halation_intensity=halation_intensity, halation=halation,
                halation_size=halation_size, halation_green_factor=halation_green_factor, sharpness=sharpness,
                sharpening_strength=sharpening_strength, sharpening_sigma=sharpening_sigma, grain=grain, highlight_burn=highlight_burn,
                burn_scale=burn_scale, color_masking=color_masking, seed=seed)
            if res is not None:
                return res
        # GpuProcessor.load_image_texture (gpu_processor.py:655-719): the pre-processed frame stays on the device while the
        # load parameters do not change -- a re-render with other film settings neither prepares nor uploads it again
        self.load_image_texture(
            src, cam, lens, lens_correction, frame_width, frame_height, rotation, zoom, rotate_times, flip, resolution,
            half_size, cache, chroma_nr, max_scale, canvas_mode, canvas_scale, canvas_ratio, exposure=exposure, metadata=metadata,
            src_version=src_version,
        )
        image, layout, payload = self._texture
        if prof:
            self._torch.cuda.synchronize(self.device)
            t_loaded = time.perf_counter()
        out_u8 = self._render_prepared(
            image, layout, payload, negative_film, grain_size, grain_sigma, dst_texture, histogram_texture, "cpu",
            print_film=print_film, exp_comp=exp_comp,
            red_light=red_light, green_light=green_light, blue_light=blue_light, projector_kelvin=projector_kelvin,
            shadow_comp=shadow_comp, sat_adjust=sat_adjust, gamma_func=gamma_func, exp_kelvin=exp_kelvin, tint=tint,
            inversion_gamma=inversion_gamma, idealized_curve=idealized_curve, inversion=inversion, push_pull=push_pull,
            white_balance=white_balance, white_clip=white_clip, icc_transform=icc_transform, frame_width=frame_width,
            frame_height=frame_height, halation_intensity=halation_intensity, halation=halation,
            halation_size=halation_size, halation_green_factor=halation_green_factor, sharpness=sharpness,
            sharpening_strength=sharpening_strength, sharpening_sigma=sharpening_sigma, grain=grain,
            highlight_burn=highlight_burn, burn_scale=burn_scale, color_masking=color_masking, seed=seed,
            canvas_mode=canvas_mode, canvas_scale=canvas_scale, canvas_ratio=canvas_ratio,
        )
        if prof:
            self._torch.cuda.synchronize(self.device)
            t_rendered = time.perf_counter()
        res = None if out_u8 is None else self._download(out_u8)  # DEVICE -> HOST, the reference's read_texture/map_sync
        if prof:
            t_end = time.perf_counter()
            self.last_stage_ms = {"load_and_upload": (t_loaded - t_start) * 1e3, "prepare_and_render": (t_rendered - t_loaded) * 1e3,
                                  "download": (t_end - t_rendered) * 1e3, "total": (t_end - t_start) * 1e3,
                                  **{k: v for k, v in (getattr(self, "_load_stage_ms", None) or {}).items()}}
            self._load_stage_ms = None
        return res

    def load_image_texture(self, src, cam=None, lens=None, lens_correction=True, frame_width=36, frame_height=24, rotation=0.0,
                           zoom=1.0, rotate_times=0, flip=False, resolution=None, half_size=True, cache=True, chroma_nr=0,
                           max_scale=400.0, canvas_mode="No", canvas_scale=1.0, canvas_ratio=1.0, exposure=None, metadata=None,
                           src_version=None):
        """GpuProcessor.load_image_texture (gpu_processor.py:655-719): prepare and upload the frame unless the load parameters
        are those of the frame that is already on the device.  A path compares by value like upstream's `src`.  An array
        compares by identity AND by a fingerprint of its content (shape, dtype, address, a checksum of up to 32 evenly spaced
        rows), so a decode buffer that was refilled is uploaded again -- a SAMPLE: an in-place edit that touches none of the sampled
        rows is not seen.  `src_version` (any hashable token that the caller changes with the content) replaces the
        checksum and saves its ~5 MB pass per call; `cache=False` always uploads.  The processor holds the array by weak reference
        only (upstream's cache never sees arrays: this is this backend's own rule, kept conservative)."""
        if isinstance(src, str):
            src_key = src
        elif src_version is not None:
            src_key = (getattr(src, "shape", None), str(getattr(src, "dtype", "")), "version", src_version)
        elif not cache:
            # cache=False uploads whatever the fingerprint says: its pass over 32 rows (4.7 MB of a 100 MP frame, ~4 ms of crc32 --
            # the unexplained part of process(host array)'s 35 ms in round 5) is skipped, and nothing is recorded that a later
            # cache=True call could take for "the same frame"
            src_key = None
        else:
            src_key = self._array_fingerprint(src)
        new_param_dict = {
            "src": src_key, "cam": cam, "lens": lens,
            "lens_correction": lens_correction,
            "frame_width": frame_width, "frame_height": frame_height, "rotation": rotation, "zoom": zoom,
            "rotate_times": rotate_times, "flip": flip, "resolution": resolution, "half_size": half_size, "chroma_nr": chroma_nr,
            "max_scale": max_scale, "canvas_mode": canvas_mode, "canvas_scale": canvas_scale, "canvas_ratio": canvas_ratio,
            "exposure": exposure, "metadata": metadata,
        }
        held = getattr(self, "_texture_src", None)
        same_src = isinstance(src, str) or (held is not None and held() is src)
        if cache and same_src and getattr(self, "_texture", None) is not None and new_param_dict == getattr(self, "image_param_dict", None):
            return
        prof = getattr(self, "profile_stages", False)
        if prof:
            import time

            t0 = time.perf_counter()
        cpu_payload = getattr(self, "_stash_payload", None)  # (made a moment ago by _process_streamed for a frame that did not qualify)
        self._stash_payload = None
        if cpu_payload is None:
            cpu_payload = self.extract_image_data_cpu(
                src, cam, lens, lens_correction, frame_width, frame_height, rotation, zoom, rotate_times, flip, resolution,
                half_size, cache, chroma_nr, max_scale, canvas_mode, canvas_scale, canvas_ratio, exposure=exposure, metadata=metadata,
                _internal=True,
            )
        if prof:
            t1 = time.perf_counter()
        self.prepare_gpu_textures(cpu_payload)
        if prof:
            self._torch.cuda.synchronize(self.device)
            self._load_stage_ms = {"host_phase": (t1 - t0) * 1e3, "upload_and_device_prepath": (time.perf_counter() - t1) * 1e3}
        self.image_param_dict = new_param_dict if (src_key is not None or isinstance(src, str)) else None
        self._texture_src = None
        if not isinstance(src, str) and src_key is not None:
            import weakref

            try:
                self._texture_src = weakref.ref(src)
            except TypeError:  # an object that cannot be weakly referenced is simply never taken for "the same frame"
                self._texture_src = None

    @staticmethod
    def _array_fingerprint(src):
        """(shape, dtype, address, strides, crc32 of a SAMPLE) of a decoded frame handed in as an array: of up to 32 evenly spaced
        rows the first, the middle and the last 512 samples each (<= 192 KB of a frame of any size).  A refilled decode buffer, an
        exposure change, any operation on the whole frame shows in it; an edit confined to other samples does not (cache=False or
        src_version for those).  Until round 6 the whole of the 32 rows went into the checksum: 2.4 ms per call on a 100 MP frame
        -- 87 % of a preview re-render (2.76 ms; tools/preview_latency_probe.py), paid on every slider step."""
        if not isinstance(src, np.ndarray):
            return None
        import zlib

        rows = src.shape[0] if src.ndim else 0
        step = max(1, rows // 32)
        crc = 0
        for r in range(0, rows, step):
            row = src[r].reshape(-1)  # (a view for the usual C-contiguous frame)
            n, k = row.size, min(row.size, 512)
            for a in sorted({0, (n - k) // 2, n - k}):
                crc = zlib.crc32(row[a:a + k].tobytes(), crc)
        return (src.shape, src.dtype.str, src.__array_interface__["data"][0], src.strides, crc)

    def prepare_gpu_textures(self, cpu_payload):
        """PHASE 2's stateful half (gpu_processor.py:785-790): upload the payload's frame and run the device pre-path on it
        (uint16 conversion, free rotation, chroma NR, preview scaling); the result is the frame the pipeline reads."""
        host = self._payload_tensor(cpu_payload)
        if cpu_payload.get("clip_on_device") and not host.is_cuda and host.dim() == 3 and host.numel() >= (1 << 24):
            # A large float frame of this object's own making (process(host array)): the clamp of gpu_processor.py:275 rides on the
            # upload -- the frame goes up in row chunks on a copy stream and every chunk is clamped on the launch stream while the
            # next one travels, so that of its 2.4 GB pass (0.5 ms at 100 MP) only the last chunk's share is left behind the copy
            torch = self._torch
            if getattr(self, "_up_stream", None) is None:
                self._up_stream = torch.cuda.Stream(device=self.device)
                self._down_stream = torch.cuda.Stream(device=self.device)
            compute = torch.cuda.current_stream(self.device)
            image = torch.empty(host.shape, dtype=host.dtype, device=self.device)
            self._up_stream.wait_stream(compute)  # (the block may have been another frame's a moment ago)
            rows = int(host.shape[0])
            step = -(-rows // 8)
            for a0 in range(0, rows, step):
                a1 = min(a0 + step, rows)
                with torch.cuda.stream(self._up_stream):
                    image[a0:a1].copy_(host[a0:a1], non_blocking=True)
                    arrived = self._up_stream.record_event()
                compute.wait_event(arrived)
                image[a0:a1].clamp_(0.0, 65504.0)
            image.record_stream(self._up_stream)
            cpu_payload = dict(cpu_payload, clip_on_device=False)
        else:
            image = host.to(self.device, non_blocking=True)  # HOST -> DEVICE, the reference's write_texture
        image, layout = self._prepare_device_frame(image, cpu_payload)
        # (what is kept next to the device frame is the payload's geometry, not its host frame: the processor must not keep a
        # 1.2 GB decode buffer alive)
        self._texture = (image, layout, {k: v for k, v in cpu_payload.items() if k != "image_array"})
        self.image_param_dict = None  # (a payload from outside: no load parameters to compare the next process() with)
        self._texture_src = None

    def _check_texture(self, t, what):
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 4
                and t.is_contiguous()):
            raise NotImplementedError(f"{what}: wgpu textures are not supported; pass a contiguous uint8 (h, w, 4) CUDA tensor "
                                      "(or use the bitmap branch, gui.py:2219-2228)")

    def process_preloaded(self, cpu_payload, negative_film, grain_size, grain_sigma, dst_texture=None,
                          histogram_texture=None, final_scaling="gpu", **settings):
        """PHASE 2 (gpu_processor.py:1643-1693): upload the payload and run the device pipeline.

        final_scaling: "gpu" -- like GpuProcessor, the canvas keeps its size and only a `max_scale` render is scaled back up;
        "cpu" -- like CpuProcessor.process (cpu_processor.py:411-412), the finished frame, canvas included, is scaled to the
        requested resolution (INTER_AREA down, LANCZOS4 up).  dst_texture / histogram_texture: see process()."""
        if dst_texture is None and histogram_texture is None and self.stream_bands > 1:
            res = self._stream_payload(cpu_payload, negative_film, grain_size, grain_sigma, final_scaling, **settings)
            if res is not None:  # a large frame without a device pre-path: through the pipeline in row bands while it arrives
                return res
        self.prepare_gpu_textures(cpu_payload)
        image, layout, _ = self._texture
        out_u8 = self._render_prepared(image, layout, cpu_payload, negative_film, grain_size, grain_sigma, dst_texture,
                                       histogram_texture, final_scaling, **settings)
        return None if out_u8 is None else self._download(out_u8)  # DEVICE -> HOST, the reference's read_texture/map_sync

    def _process_streamed(self, src, negative_film, grain_size, grain_sigma, *, load, **settings):
        """process() of a large host frame that is uploaded for this one render (cache=False: the GUI's export calls,
        gui.py:2374,2458,2479): the frame goes through the pipeline in `stream_bands` row bands WHILE IT ARRIVES -- band k is clamped and taken through S0 + S1
        as soon as it is on the device, the halation of band k - 1 follows (its stencil reads the first rows of band k), then the
        MTF and the tail of band k - 2, whose uint8 rows start their way back while later bands are still coming up: the stage
        entry points are row-range calls (the ones a row shard makes: grain hashed at global coordinates, reflection at the frame
        edges only, the exposure-range record kept like r2f_render keeps it), PCIe is full duplex, and what remains of the render
        and the download behind the upload is the last two bands' share.  At 100 MP: upload 21.1 ms, render 4.7, download 5.3
        one after the other = 32.5 ms; streamed 23.3 (bench.py host_device_copies, tools/stream_bands_probe.py).
        The FFT stencils' windows are anchored at a call's first row, so a band's outputs agree with the whole-frame render's to the
        FFT form's rounding (an fp32 ulp on a handful of samples, like a row shard's); pointwise configurations agree bit for bit.
        A uint16 frame (LibRaw's 16-bit output: half the upload) is converted band by band on the device as it arrives.
        Returns None when the frame does not qualify (the caller then takes the one-after-the-other path): a device pre-path
        (rotation, chroma NR, scaling), a canvas, a highlight burn (a function of the whole grained frame), a small frame.
        The two-phase API's device phase (process_preloaded, submit_preloaded: batch export) streams its payload the same way."""
        payload = self.extract_image_data_cpu(
            src, load["cam"], load["lens"], load["lens_correction"], load["frame_width"], load["frame_height"], load["rotation"],
            load["zoom"], load["rotate_times"], load["flip"], load["resolution"], load["half_size"], load["cache"], load["chroma_nr"],
            load["max_scale"], load["canvas_mode"], load["canvas_scale"], load["canvas_ratio"], exposure=load["exposure"],
            metadata=load["metadata"], _internal=True)
        res = self._stream_payload(payload, negative_film, grain_size, grain_sigma, "cpu", **settings)
        # (should the frame not qualify, load_image_texture takes the payload from here)
        self._stash_payload = payload if res is None else None
        return res

    def _stream_payload(self, payload, negative_film, grain_size, grain_sigma, final_scaling="cpu", **settings):
        """A phase-1 payload (extract_image_data_cpu) through the pipeline in row bands while it arrives: see _process_streamed.
        Returns the uint8 frame, or None (with `stream_rejected` saying why) when the payload does not qualify."""
        host = self._payload_tensor(payload)
        torch = self._torch
        is_u16 = host.dtype == torch.int16  # LibRaw's 16-bit output: converted band by band on the device (raw_conversion.py:50-52)
        if (payload.get("warp") or payload.get("resize_to") or payload.get("upscale_to") or payload.get("chroma_nr")
                or payload.get("canvas_resolution") or settings.get("canvas_mode", "No") != "No" or host.is_cuda or host.dim() != 3
                or int(host.shape[2]) not in (3, 4) or host.numel() < (1 << 24)
                or (host.dtype not in (torch.float32, torch.int16))
                or (payload.get("u16_factor") is None if is_u16 else payload.get("u16_factor") is not None)):
            self.stream_rejected = ("a device pre-path, a canvas, or a frame below 16.7 M samples: " + ", ".join(
                f"{k} = {payload.get(k)!r}" for k in ("warp", "resize_to", "upscale_to", "chroma_nr", "canvas_resolution", "u16_factor",
                                                      "clip_on_device")) + f", frame {tuple(host.shape)} {host.dtype}")
            return None
        H, W = int(host.shape[0]), int(host.shape[1])
        fr = payload.get("final_resolution") if final_scaling == "cpu" else None  # (cpu_processor.py:411-412: the final scaling)
        if fr is not None and (int(fr[0]), int(fr[1])) != (H, W):
            self.stream_rejected = f"the finished frame is scaled to {fr}"
            return None
        ctx = self.ctx
        params = self.prepare(negative_film, grain_size, grain_sigma, (W, H), **settings)
        flags = int(params.flags)
        hal, mtf, grain = bool(flags & _lib.F_HALATION), bool(flags & _lib.F_MTF), bool(flags & _lib.F_GRAIN)
        if flags & _lib.F_BURN:
            self.stream_rejected = "highlight burn (a function of the whole grained frame)"
            return None
        ha = self._halation_reach if hal else (0, 0)
        ma = self._mtf_reach if mtf else (0, 0)
        n = min(int(self.stream_bands), max(H // 512, 2))
        while n > 1 and H // n < max(2 * max(ha + ma) + 2, 64):  # a band holds its neighbours' halo (and is worth a launch)
            n -= 1
        if n < 2:
            self.stream_rejected = f"{n} band(s) of {H} rows above the stencils' reach {ha} + {ma}"
            return None
        self.stream_rejected = None
        bounds = [H * i // n for i in range(n + 1)]
        # the last bands are the ones nothing hides (their stencil stages, tail and download run behind the last byte of the upload):
        # the final `stream_taper` of them are halved while they stay above the stencils' reach (100 MP: 23.26 -> 22.73 ms with 2)
        floor_rows = max(2 * max(ha + ma) + 2, 64)
        cut = [(bounds[i] + bounds[i + 1]) // 2 for i in range(max(n - int(self.stream_taper), 0), n)
               if bounds[i + 1] - bounds[i] >= 2 * floor_rows]
        bounds = sorted(set(bounds + cut))
        n = len(bounds) - 1
        bufs = getattr(self, "_stream_bufs", None)
        chans = 3 if is_u16 else int(host.shape[2])  # (a payload with upstream's constant alpha plane, gpu_processor.py:765: 4)
        if bufs is None or bufs["shape"] != (H, W, chans) or bufs["mtf"] != mtf:
            def planes():
                return torch.empty((3, H, W), dtype=torch.float32, device=self.device)

            bufs = self._stream_bufs = {"shape": (H, W, chans), "mtf": mtf,
                                        "image": torch.empty((H, W, chans), dtype=torch.float32, device=self.device),
                                        "E": planes(), "D": planes(), "D2": planes() if mtf else None,
                                        "u8": torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)}
        image, E, D, D2, out_u8 = bufs["image"], bufs["E"], bufs["D"], bufs["D2"], bufs["u8"]
        if is_u16 and (bufs.get("u16") is None or tuple(bufs["u16"].shape) != tuple(host.shape)):
            bufs["u16"] = torch.empty(tuple(host.shape), dtype=torch.int16, device=self.device)
        raw16 = bufs.get("u16") if is_u16 else None
        nres = self.result_buffers
        fresh, copies = None, []
        if nres > 0:  # the caller takes views of pinned buffers in turn (_download)
            ring = getattr(self, "_result_ring", None)
            if ring is None or ring[0].shape != out_u8.shape or len(ring) != nres:
                ring = self._result_ring = [torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True) for _ in range(nres)]
                self._result_turn = 0
            result = ring[self._result_turn % nres]
            self._result_turn += 1
        elif (leased := self._lease_result(tuple(out_u8.shape))) is not None:
            # upstream's ownership semantics -- an array of the caller's own per call -- without a fresh allocation: the array is a
            # view of one of up to three pinned buffers this object lends out, and the buffer comes back when the caller's last
            # reference to the array (or to any view of it) is gone.  A caller that drops a result before asking for the next but
            # one -- an export loop, a preview widget -- never meets a fresh page
            result = leased
        else:
            # ... and for a caller that holds on to more results than that: a FRESH array.  Its pages are touched for the first time by whoever writes them
            # (~30 ms for 0.3 GB) -- here by four helper threads that fault them in while the first bands are still on their way up,
            # then copy each band out of a pinned staging buffer as soon as it is back, instead of by one pageable download behind
            # the render
            stage = bufs.get("stage")
            if stage is None:
                stage = bufs["stage"] = torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True)
            result = stage
            fresh = np.empty((H, W, 3), dtype=np.uint8)
            if getattr(self, "_copy_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor

                self._copy_pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix="r2f-result")
            stage_np = stage.numpy()
            flat = fresh.reshape(-1)

            def touch(i0, i1):
                flat[i0:i1:4096] = 0  # one byte per page (NumPy releases the GIL for the strided fill)

            q = -(-flat.size // 4)
            copies += [self._copy_pool.submit(touch, i, min(i + q, flat.size)) for i in range(0, flat.size, q)]

            def copy_out(back, y0, y1):
                back.synchronize()  # (releases the GIL)
                np.copyto(fresh[y0:y1], stage_np[y0:y1])
        if getattr(self, "_up_stream", None) is None:
            self._up_stream = torch.cuda.Stream(device=self.device)
            self._down_stream = torch.cuda.Stream(device=self.device)
        up, down, compute = self._up_stream, self._down_stream, torch.cuda.current_stream(self.device)
        p = _lib.Params.from_buffer_copy(params)
        p.flags |= _lib.F_FRAME_RESIDENT  # the seed is written once, below; the stage calls read it from the frame block
        ctx.write_frame_params(p)          # (and the exposure-range record starts empty)
        up.wait_stream(compute)            # (the buffers may still be read by the previous frame's launches)
        down.wait_stream(compute)
        pointwise = not (hal or mtf or grain)
        state = {"front": 0, "dens": 0, "mtf": 0, "tail": 0, "ident": 0}
        cur = D2 if mtf else D

        def band(b):
            return bounds[b], bounds[b + 1]

        def send_back(b):
            y0, y1 = band(b)
            done = compute.record_event()
            with torch.cuda.stream(down):
                down.wait_event(done)
                result[y0:y1].copy_(out_u8[y0:y1], non_blocking=True)
                if fresh is not None:
                    copies.append(self._copy_pool.submit(copy_out, down.record_event(), y0, y1))

        def advance():
            # every stage runs a band as soon as the rows it reads exist: a stencil stage reads into the band after its own
            while True:
                moved = False
                if hal and state["dens"] < n and state["front"] > min(state["dens"] + 1, n - 1):
                    y0, y1 = band(state["dens"])
                    lo, hi = max(y0 - ha[0], 0), min(y1 + ha[1], H)
                    ctx.stage_halation(E[:, lo:hi], D, p, src_gy0=lo, dst_gy0=0, y0=y0, y1=y1, H_global=H,
                                       identity_done=state["ident"], range_valid=True)
                    state["dens"] += 1
                    moved = True
                if not hal:
                    state["dens"] = state["front"]
                if mtf and state["mtf"] < n and state["dens"] > min(state["mtf"] + 1, n - 1):
                    y0, y1 = band(state["mtf"])
                    lo, hi = max(y0 - ma[0], 0), min(y1 + ma[1], H)
                    ctx.stage_mtf(D[:, lo:hi], D2, p, src_gy0=lo, dst_gy0=0, y0=y0, y1=y1, H_global=H)
                    state["mtf"] += 1
                    moved = True
                ready = state["mtf"] if mtf else state["dens"]
                if not pointwise and state["tail"] < ready:
                    y0, y1 = band(state["tail"])
                    ctx.stage_tail(cur, p, src_gy0=0, out_u8=out_u8, out_gy0=0, y0=y0, y1=y1, H_global=H)
                    send_back(state["tail"])
                    state["tail"] += 1
                    moved = True
                if not moved:
                    return

        landing = raw16 if is_u16 else image

        def send_up(k):
            a0, a1 = band(k)
            with torch.cuda.stream(up):
                landing[a0:a1].copy_(host[a0:a1], non_blocking=True)
                return up.record_event()

        # A copy out of ordinary (pageable) host memory -- any NumPy array that was not made from pinned memory -- returns only when
        # its bytes have left the host: issued from this thread, every band's copy would hold back the launches of the band before
        # it (measured: 35.5 ms against 32.7 one after the other).  So such a source is sent up by a helper thread of its own, band
        # by band, and this thread launches a band when its arrival event comes through the queue.
        arrivals, uploader = None, None
        if not host.is_pinned():
            import queue
            import threading

            arrivals = queue.Queue()

            def upload_all():
                try:
                    with torch.cuda.device(self.device):
                        for k in range(n):
                            arrivals.put(send_up(k))
                except BaseException as e:  # noqa: BLE001 -- handed to the thread that waits for the bands
                    arrivals.put(e)

            uploader = threading.Thread(target=upload_all, name="r2f-upload", daemon=True)
            uploader.start()
        try:
            for k in range(n):
                a0, a1 = band(k)
                arrived = send_up(k) if arrivals is None else arrivals.get()
                if isinstance(arrived, BaseException):
                    raise arrived
                compute.wait_event(arrived)
                rows = image[a0:a1]
                if is_u16:
                    ctx.decode_u16(raw16[a0:a1], payload["u16_factor"], out=rows)
                elif payload.get("clip_on_device"):
                    rows.clamp_(0.0, 65504.0)  # np.clip(image, 0, 65504) of gpu_processor.py:275, band by band
                if pointwise:  # LUTs only: one fused pass per band, straight to uint8
                    ctx.stage_front(rows, p, 2, in_gy0=a0, out_u8=out_u8, out_gy0=0, y0=a0, y1=a1, H_global=H)
                    send_back(k)
                elif hal:
                    state["ident"] = ctx.stage_front_split(rows, p, E, D, in_gy0=a0, y0=a0, y1=a1, H_global=H, track_range=True)
                else:
                    ctx.stage_front(rows, p, 1, in_gy0=a0, dst=D, dst_gy0=0, y0=a0, y1=a1, H_global=H)
                state["front"] = k + 1
                advance()
        except BaseException:
            # a stage call refused (or the caller interrupted): let the queued work drain, hand a lent buffer back, pass it on
            if uploader is not None:
                uploader.join()
            torch.cuda.synchronize(self.device)
            for c in copies:
                c.cancel()
            if nres <= 0 and fresh is None:
                self._lease_pool.append(result)
            raise
        if uploader is not None:
            uploader.join()
        down.synchronize()
        for c in copies:
            c.result()
        # (the frame kept on the device for re-renders -- a preview's, typically -- is left alone: an export in between does not cost
        # the preview its cached frame, which the one-after-the-other path has to overwrite because it works in it)
        self.last_output = out_u8
        if fresh is not None:
            return fresh
        arr = result.numpy()
        if nres <= 0:  # a lent buffer: back into the pool when the caller lets go of the array
            import weakref

            weakref.finalize(arr, self._lease_pool.append, result)
        return arr

    def _lease_result(self, shape):
        """A pinned uint8 buffer of `shape` to lend to the caller of process() as its result (see _process_streamed), or None when
        three are out already."""
        pool = self.__dict__.setdefault("_lease_pool", [])
        for i, t in enumerate(pool):
            if tuple(t.shape) == shape:
                return pool.pop(i)
        del pool[:]  # (buffers of another frame size: let them go)
        if self.__dict__.get("_lease_shape") != shape:
            self._lease_shape, self._lease_count = shape, 0
        if self._lease_count >= 3:
            return None
        self._lease_count += 1
        return self._torch.empty(shape, dtype=self._torch.uint8, pin_memory=True)

    def _download(self, out_u8):
        """The uint8 result as a NumPy array.  Default: an array of the caller's own per call, like upstream.  With result_buffers = n > 0 the
        frame lands in one of n pinned host buffers taken in turn (a 24 MP frame then takes 1.5 instead of 6 ms to come down)
        and the returned array is a VIEW of it: valid until n more frames of the same size have been returned."""
        n = getattr(self, "result_buffers", 0)
        torch = self._torch
        if n <= 0:
            # the caller's own array, like upstream's -- for a frame of a megapixel and more a view of a pinned buffer this object
            # lends out (up to three; see _process_streamed) instead of a freshly allocated pageable array: the download runs at
            # the link's rate and no page is touched for the first time (24 MP: 2.7 instead of 7.8 ms)
            leased = self._lease_result(tuple(out_u8.shape)) if out_u8.numel() >= (3 << 20) else None
            if leased is None:
                return out_u8.cpu().numpy()
            import weakref

            leased.copy_(out_u8, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            arr = leased.numpy()
            weakref.finalize(arr, self._lease_pool.append, leased)
            return arr
        ring = getattr(self, "_result_ring", None)
        if ring is None or ring[0].shape != out_u8.shape or len(ring) != n:
            ring = self._result_ring = [torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True) for _ in range(n)]
            self._result_turn = 0
        host = ring[self._result_turn % n]
        self._result_turn += 1
        host.copy_(out_u8, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return host.numpy()

    def _payload_tensor(self, cpu_payload):
        """The payload's frame as a torch tensor: float32, or the 16 bits of a uint16 frame (as int16: same bytes)."""
        torch = self._torch
        image = cpu_payload["image_array"]
        if isinstance(image, np.ndarray):
            if image.dtype == np.uint16:
                return torch.from_numpy(np.ascontiguousarray(image).view(np.int16))
            return torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        return image

    def submit_preloaded(self, cpu_payload, negative_film, grain_size, grain_sigma, final_scaling="gpu", **settings):
        """process_preloaded without waiting: the upload runs on a copy stream, the render on the current stream, the download
        of the uint8 result into pinned host memory on a second copy stream; returns a `PendingFrame` whose `.result()` is
        process_preloaded's return value.  Keeping one frame pending while the next one is submitted overlaps both PCIe
        directions with the render (raw2film_amd.sharding.BatchSharder.run(..., collect=...) does exactly that) -- what the
        reference's queue.write_texture / read_texture pair serialises."""
        torch = self._torch
        src = cpu_payload.get("image_array")
        if self.stream_bands > 1 and not (isinstance(src, torch.Tensor) and (src.is_cuda or src.is_pinned())):
            # a large frame in ordinary (pageable) host memory: its copy blocks this thread, so nothing of the next frame's host work
            # would overlap it -- it overlaps its own upload, render and download band by band instead (_process_streamed; 24 MP:
            # 7.0 against 7.5 ms, uint16 4.7 against 5.0) and is back when this call returns.  A pinned payload keeps the
            # frame-in-flight scheme below (2.9 against 3.5 ms for uint16: tools/batch_pageable_probe.py)
            res = self._stream_payload(cpu_payload, negative_film, grain_size, grain_sigma, final_scaling, **settings)
            if res is not None:
                return PendingFrame.finished(res)
        if getattr(self, "_up_stream", None) is None:
            self._up_stream = torch.cuda.Stream(device=self.device)
            self._down_stream = torch.cuda.Stream(device=self.device)
        image = self._payload_tensor(cpu_payload)
        # (A pageable source -- an ordinary NumPy array -- makes the "asynchronous" copy below a synchronous one: this thread waits
        # for the frame's bytes to leave the host.  That is fine: the frame before is already queued on the device and its render
        # and download run meanwhile.  Pinning the array first, which this method did until round 6, is a fresh pinned allocation
        # and a single-threaded host copy per frame: 64 instead of 6.3 ms per 24 MP frame, tools/batch_pageable_probe.py.)
        compute = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._up_stream):
            dev = image.to(self.device, non_blocking=True)
            uploaded = self._up_stream.record_event()
        compute.wait_event(uploaded)
        dev.record_stream(compute)
        out_u8 = self._render_preloaded(dev, cpu_payload, negative_film, grain_size, grain_sigma, None, None, final_scaling, **settings)
        rendered = compute.record_event()
        with torch.cuda.stream(self._down_stream):
            self._down_stream.wait_event(rendered)
            host = torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(out_u8, non_blocking=True)
            out_u8.record_stream(self._down_stream)
            done = self._down_stream.record_event()
        return PendingFrame(host, done)

    def _render_preloaded(self, image, cpu_payload, negative_film, grain_size, grain_sigma, dst_texture, histogram_texture,
                          final_scaling, **settings):
        """The device half of process_preloaded on an uploaded frame; returns the uint8 device result (None with a dst_texture)."""
        image, layout = self._prepare_device_frame(image, cpu_payload)
        return self._render_prepared(image, layout, cpu_payload, negative_film, grain_size, grain_sigma, dst_texture,
                                     histogram_texture, final_scaling, **settings)

    def _prepare_device_frame(self, image, cpu_payload):
        """The device pre-path on an uploaded payload frame -> (frame, layout) as the pipeline reads it."""
        torch = self._torch
        if image.dtype in (torch.int16, torch.uint16):  # a decoded 16-bit frame: raw_conversion.py:50-52 on the device
            if cpu_payload.get("u16_factor") is None:
                raise ValueError("a uint16 payload needs its exposure factor (`u16_factor`, extract_image_data_cpu sets it)")
            image = self.ctx.decode_u16(image.contiguous(), cpu_payload["u16_factor"])
        elif cpu_payload.get("clip_on_device"):
            image = image.clamp_(0.0, 65504.0)  # np.clip(image, 0, 65504) of gpu_processor.py:275, on the uploaded copy
        layout = None  # the payload is (H, W, C) like the reference's; the device pre-path hands on (3, H, W) planes
        warp = cpu_payload.get("warp")
        if warp:  # free rotation (effects.rotate) + the crops behind it + quarter turns (np.rot90 on the planes)
            image = self.ctx.warp_affine(image.contiguous(), warp["m_dst_to_src"], warp["window"])
            if warp["rotate_times"]:
                image = torch.rot90(image, warp["rotate_times"], dims=(1, 2)).contiguous()
            layout = "chw"
        if cpu_payload.get("chroma_nr"):  # pre-path chroma NR (effects.py:547-561): XYZ planes out, CHW into the pipeline
            image, layout = self.ctx.chroma_nr(image.contiguous(), cpu_payload["chroma_nr"], layout=layout), "chw"
        if cpu_payload.get("resize_to"):  # preview scaling, after the NR like cpu_processor.py:119-134
            _, h_in, w_in = self.ctx.layout_of(image, layout)
            rt = cpu_payload["resize_to"]
            if rt[0] <= h_in and rt[1] <= w_in:  # cv.INTER_AREA
                image, layout = self.ctx.resize_area(image.contiguous(), *rt, layout=layout), "chw"
            else:  # cv.INTER_LANCZOS4 on the float frame (utils.py:237-242)
                image, layout = self.ctx.resize_lanczos4_f32(image.contiguous(), *rt, layout=layout), "chw"
        return image, layout

    def _render_prepared(self, image, layout, cpu_payload, negative_film, grain_size, grain_sigma, dst_texture, histogram_texture,
                         final_scaling, **settings):
        """The pipeline and the post-path on a prepared device frame (which is only read: it can be rendered again)."""
        if dst_texture is not None:
            self._check_texture(dst_texture, "dst_texture")
        if histogram_texture is not None:
            self._check_texture(histogram_texture, "histogram_texture")
            if dst_texture is None:
                raise ValueError("histogram_texture needs dst_texture (gpu_processor.py:1883: the histogram is only drawn on the "
                                 "destination-texture branch)")
        torch = self._torch  # noqa: F841
        out_f32, out_u8 = self._execute_pipeline(image, negative_film, grain_size, grain_sigma, want_f32=dst_texture is not None,
                                                 want_u8=True, layout=layout, **settings)
        if dst_texture is not None:
            # GpuProcessor's destination branch (gpu_processor.py:1865-1890): letterbox the float frame into the widget's
            # texture with copy_to_int.wgsl's transform, draw the histogram into its own texture, return nothing
            H, W = int(out_f32.shape[0]), int(out_f32.shape[1])
            color = None
            if settings.get("canvas_mode", "No") != "No":
                _, color, _ = geometry.canvas_layout((H, W), settings["canvas_mode"], settings.get("canvas_scale", 1.0),
                                                     settings.get("canvas_ratio", 1.0))
            t = geometry.blit_transform((W, H), (int(dst_texture.shape[1]), int(dst_texture.shape[0])),
                                        pipeline_resolution=cpu_payload.get("pipeline_resolution"),
                                        output_resolution=cpu_payload.get("output_resolution"),
                                        canvas_resolution=cpu_payload.get("canvas_resolution"), canvas_color=color)
            self.ctx.blit_rgba8(out_f32, dst_texture, t)
            self.last_output = out_u8
            if histogram_texture is not None:
                from . import histogram

                self.ctx.histogram_render(self.ctx.histogram_counts(out_u8), histogram.MIX_TABLE, 256, target=histogram_texture)
            return None
        # canvas on the device result (cpu_processor.py:409 / copy_to_int.wgsl): a paste, no arithmetic
        out_u8 = geometry.add_canvas(out_u8, settings.get("canvas_mode", "No"), settings.get("canvas_scale", 1.0),
                                     settings.get("canvas_ratio", 1.0))
        target = cpu_payload.get("final_resolution") if final_scaling == "cpu" else cpu_payload.get("upscale_to")
        if target:  # cpu_processor.py:411-412 -> utils.resolution_scaling (utils.py:226-244) on the uint8 frame
            f = min(target[0] / out_u8.shape[0], target[1] / out_u8.shape[1])
            size = (round(out_u8.shape[0] * f), round(out_u8.shape[1] * f))
            if f > 1:  # cv.INTER_LANCZOS4
                out_u8 = self.ctx.resize_lanczos4_u8(out_u8.contiguous(), *size)
            elif f < 1 and final_scaling == "cpu":  # cv.INTER_AREA: the CPU processor shrinks the canvas-framed frame
                out_u8 = self.ctx.resize_area_u8(out_u8.contiguous(), *size)
        self.last_output = out_u8
        return out_u8

    def generate_histogram(self, image=None, mix_table=None, height=100):
        """utils.generate_histogram (utils.py:145-223) of `image` (uint8 (H, W, 3), NumPy or device) -- or, with no image,
        of the frame the last process() call rendered, counted on the device before it ever left it (the GUI calls this
        right after process(), gui.py:2225).  Returns the (height, 256, 4) uint8 bar image."""
        from . import histogram

        if image is None:
            image = self.last_output
            if image is None:
                raise ValueError("generate_histogram(): no frame has been rendered yet")
        return histogram.generate_histogram(image, histogram.MIX_TABLE if mix_table is None else mix_table, height,
                                            ctx=self.ctx)

    def process_array(self, image, negative_film, grain_size=6, grain_sigma=0.4, *, colorspace="XYZ", seed=None,
                      return_float=False, output="host", out=None, **settings):
        """Render a decoded frame given as an array / CUDA tensor (synthetic benchmark frames).

        colorspace: "XYZ" (S0 skipped) or "linear-rec709" (S0 = data.py:128-135).
        return_float: float32 display-referred (H, W, 3) instead of uint8.  output: "host" | "device".
        out: a device tensor (H, W, 3) of the result's dtype to render into (a stream of frames then keeps its buffers, and
        r2f_render its captured graph).
        """
        torch = self._torch
        if colorspace not in ("XYZ", "linear-rec709"):
            raise ValueError("colorspace must be 'XYZ' or 'linear-rec709'")
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        image = image.to(self.device).contiguous()
        if out is not None:
            _, H, W = self.ctx.layout_of(image, settings.get("layout"))
            want = torch.float32 if return_float else torch.uint8
            if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == want and tuple(out.shape) == (H, W, 3) and out.is_contiguous()):
                raise ValueError(f"out must be a contiguous {want} CUDA tensor of shape {(H, W, 3)}")
        f32, u8 = self._execute_pipeline(
            image, negative_film, grain_size, grain_sigma, want_f32=return_float, want_u8=not return_float,
            matrix=REC709_TO_XYZ if colorspace == "linear-rec709" else None, seed=seed,
            out_f32=out if return_float else None, out_u8=None if return_float else out, **settings,
        )
        out = f32 if return_float else u8
        return out if output == "device" else out.cpu().numpy()

    # ------------------------------------------------------------------ device pipeline
    def prepare(self, negative_film, grain_size, grain_sigma, pipeline_resolution, *, print_film=None, exp_comp=0.0,
                red_light=0.0, green_light=0.0, blue_light=0.0, projector_kelvin=6500, shadow_comp=0.0, sat_adjust=1.0,
                gamma_func="sRGB", exp_kelvin=6500, tint=0.0, inversion_gamma=4.0, idealized_curve=False,
                inversion=False, push_pull=0.0, white_balance=False, white_clip=False, icc_transform=None,
                frame_width=36, frame_height=24, halation_intensity=1.0, halation=True, halation_size=1.0,
                halation_green_factor=0.4, sharpness=True, sharpening_strength=0.0, sharpening_sigma=1.0, grain=2,
                highlight_burn=0.0, burn_scale=50.0, color_masking=None, matrix=None, seed=None, lut3d_mode=0, **_):
        """Upload whatever changed and return the r2f_params for this render: the table half of
        `_execute_gpu_pipeline` (gpu_processor.py:1735-1756, 1772-1825)."""
        self.pipeline_resolution = (int(pipeline_resolution[0]), int(pipeline_resolution[1]))
        self.load_input_lut(negative_film, exp_kelvin, tint, exp_comp)
        self.load_density_curve(negative_film, push_pull, color_masking)
        self.load_output_lut(negative_film, print_film, red_light, green_light, blue_light, projector_kelvin,
                             shadow_comp, sat_adjust, gamma_func, inversion_gamma, idealized_curve, inversion,
                             white_balance, white_clip, icc_transform, color_masking)
        scale = max(pipeline_resolution) / max(frame_width, frame_height)  # px per mm, cpu_processor.py:366
        do_hal = bool(halation)
        do_mtf = bool(sharpness) and negative_film.mtf is not None
        do_grain = bool(grain) and negative_film.rms_density is not None
        if do_hal:
            self.load_halation_kernel(scale, halation_size=halation_size, halation_green_factor=halation_green_factor,
                                      halation_intensity=halation_intensity, bw=negative_film.density_measure == "bw")
        if do_mtf:
            self.load_mtf_kernel(negative_film, scale, sharpening_strength, sharpening_sigma)
        if do_grain:
            self.load_grain(negative_film, scale, grain_size / 1000, grain_sigma, grain == 1)
        mkey = None if matrix is None else np.asarray(matrix, dtype=np.float32).tobytes()
        if mkey != self.matrix_key:
            self.ctx.set_matrix3x3(matrix)
            self.matrix_key = mkey
        if seed is None:
            seed = random.randint(0, 100000000)  # gpu_processor.py:591: a new seed every render
        # S7 gate: cpu_processor.py:399-402
        do_burn = bool(highlight_burn) and (print_film is not None or negative_film.density_measure in ["status_m", "bw"])
        burn_kw = {}
        if do_burn:
            burn = self.load_highlight_burn(negative_film, highlight_burn, burn_scale, pipeline_resolution)
            burn_kw = dict(burn_strength=float(burn["highlight_burn"]), burn_d_ref=float(burn["d_ref"]), burn_cell=burn["cell"])
        return self.ctx.make_params(matrix=matrix is not None, halation=do_hal, mtf=do_mtf, grain=do_grain,
                                    grain_mono=grain == 1, seed=seed, lut3d_mode=lut3d_mode,
                                    log_eps=LOG_EPS, lut3d_scale=LUT3D_SCALE, **burn_kw)

    def _execute_pipeline(self, image, negative_film, grain_size, grain_sigma, want_f32=False, want_u8=True, layout=None,
                          out_f32=None, out_u8=None, **settings):
        """Tables (re-uploaded only when their parameters changed) + ONE r2f_render: the frame's launches are captured into a
        HIP graph the second time the same buffers come by and replayed from then on -- the counterpart of the reference's
        single command encoder and submit (gpu_processor.py:1760, 1877); the per-render seed travels in a device-side block."""
        _, H, W = self.ctx.layout_of(image, layout)
        params = self.prepare(negative_film, grain_size, grain_sigma, (W, H), **settings)
        return self.ctx.render(image, params, out_f32=out_f32, out_u8=out_u8, want_f32=want_f32, want_u8=want_u8, layout=layout)
