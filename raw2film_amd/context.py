"""HipContext: a thin object over the C ABI (include/r2f.h), torch tensors as device buffers.

One context per GPU, entered by one thread at a time (the reference's processors have the
same rule, gui.py:2119-2129).  All launches go to torch's current stream on that device.
torch is used for memory and streams only; every computation is a HIP kernel of libr2f_hip.so.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

LOG_EPS = 1e-6  # lut_1d.wgsl:24
LUT3D_SCALE = 0.25  # cpu_processor.py:405


class R2FError(RuntimeError):
    pass


def _host_f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class HipContext:
    def __init__(self, device: int = 0, lib_path: str | None = None):
        import torch

        self._torch = torch
        # raises ImportError if the HIP library is missing -- no fallback.  lib_path: a development variant of the library bound
        # beside the in-tree one (A/B of two builds in one process)
        self._lib = _lib.load(lib_path)
        if not torch.cuda.is_available():
            raise R2FError("raw2film_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device("cuda", device)
        handle = C.c_void_p()
        rc = self._lib.r2f_create(int(device), C.byref(handle))
        if rc != 0 or not handle.value:
            raise R2FError(f"r2f_create(device={device}) failed with code {rc}")
        self._h = handle
        self._workspace = None

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.r2f_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.r2f_last_error(self._h).decode(errors="replace")
            if rc == _lib.EINVAL:
                raise ValueError(msg)
            raise R2FError(f"r2f error {rc}: {msg}")

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def make_params(self, *, matrix=False, halation=False, mtf=False, grain=False, grain_mono=False, seed=0,
                    lut3d_mode=0, log_eps=LOG_EPS, lut3d_scale=LUT3D_SCALE, burn_strength=0.0, burn_cell=0,
                    burn_d_ref=0.0) -> _lib.Params:
        """burn_strength != 0 switches S7 on; burn_cell = ceil(min(H, W) / burn_scale) (effects.py:365)."""
        flags = (
            (_lib.F_MATRIX if matrix else 0)
            | (_lib.F_HALATION if halation else 0)
            | (_lib.F_MTF if mtf else 0)
            | (_lib.F_GRAIN if grain else 0)
            | (_lib.F_GRAIN_MONO if grain_mono else 0)
            | (_lib.F_BURN if burn_strength else 0)
        )
        return _lib.Params(flags, int(seed) & 0xFFFFFFFF, float(log_eps), float(lut3d_scale), int(lut3d_mode),
                           int(burn_cell), float(burn_strength), float(burn_d_ref))

    def planes(self, t, gy0: int = 0) -> _lib.Planes:
        """Describe a float32 (3, rows, W) device tensor holding global rows gy0..: contiguous, or a ROW SLICE of such a tensor
        (t[:, a:b, :]: rows contiguous, planes a fixed stride apart -- r2f_planes carries the plane stride)."""
        torch = self._torch
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 3 or t.shape[0] != 3 or not t.is_cuda:
            raise ValueError("planes: need a float32 CUDA tensor of shape (3, rows, W)")
        rows, W = int(t.shape[1]), int(t.shape[2])
        if t.is_contiguous():
            stride = rows * W
        elif rows > 0 and t.stride(2) == 1 and t.stride(1) == W and t.stride(0) >= rows * W:
            stride = int(t.stride(0))
        else:
            raise ValueError("planes: need a contiguous (3, rows, W) tensor or a row slice t[:, a:b, :] of one")
        self._same_device(t, "planes")
        return _lib.Planes(t.data_ptr(), stride, int(gy0), rows)

    def _check_out(self, t, dtype, W, what, *, rows=None, gy0=0, y0=None, y1=None):
        """A caller's output tensor before its pointer crosses the C ABI (which cannot check it): (rows, W, 3) of `dtype`,
        contiguous, on this context's device, holding the global rows [gy0, gy0 + rows) and covering [y0, y1)."""
        if t is None:
            return
        torch = self._torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.dim() == 3 and t.shape[2] == 3
                and int(t.shape[1]) == int(W) and t.is_contiguous()):
            raise ValueError(f"{what} must be a contiguous {dtype} CUDA tensor of shape (rows, {int(W)}, 3)")
        self._same_device(t, what)
        if rows is not None and int(t.shape[0]) != int(rows):
            raise ValueError(f"{what} has {int(t.shape[0])} rows, the frame {int(rows)}")
        if y0 is not None and (y0 < gy0 or y1 > gy0 + int(t.shape[0])):
            raise ValueError(f"{what} holds rows [{gy0}, {gy0 + int(t.shape[0])}), asked to write [{y0}, {y1})")

    def generation(self) -> int:
        """Change counter of the context's tables, options and internal buffers (r2f_generation): a captured HIP graph of this
        context's launches is stale once it moves."""
        return int(self._lib.r2f_generation(self._h))

    def render_stats(self) -> dict:
        """Counters of `render` (r2f_render_stats): frames replayed from a captured HIP graph, graphs captured, frames launched
        kernel by kernel, graphs dropped."""
        out = (C.c_uint64 * 4)()
        self._check(self._lib.r2f_render_stats(self._h, out))
        return {"replays": int(out[0]), "captures": int(out[1]), "eager": int(out[2]), "dropped": int(out[3])}

    def frame_exposure_range(self) -> dict:
        """r2f_frame_exposure_range: min / max |.| of the exposure planes the last whole-frame render's halation read, the rule
        (bound, floor) and whether its FFT passes took the 12-byte scratch element (synchronises the device)."""
        out, armed, packed = (C.c_float * 4)(), C.c_int(), C.c_int()
        self._check(self._lib.r2f_frame_exposure_range(self._h, out, C.byref(armed), C.byref(packed)))
        pairs, packed_pairs = C.c_int(), C.c_int()
        self._check(self._lib.r2f_frame_scratch_choice(self._h, C.byref(pairs), C.byref(packed_pairs)))
        # the choice is made per window pair: `twelve_byte_element` = every pair of the last halation call took it; the counts beside it
        return {"min": float(out[0]), "max_abs": float(out[1]), "bound": float(out[2]), "floor": float(out[3]),
                "armed": bool(armed.value), "twelve_byte_element": bool(packed.value),
                "pairs": int(pairs.value), "packed_pairs": int(packed_pairs.value)}

    def frame_scratch_flags(self):
        """r2f_frame_scratch_flags: one flag per window pair (per channel) of the last halation call that chose its scratch element on
        the device -- 1 = the 12-byte element -- as a numpy int32 array (empty when it did not choose).  Synchronises the device."""
        import numpy as np

        n = C.c_int()
        self._check(self._lib.r2f_frame_scratch_flags(self._h, None, 0, C.byref(n)))
        flags = np.zeros(max(int(n.value), 0), dtype=np.int32)
        if flags.size:
            self._check(self._lib.r2f_frame_scratch_flags(self._h, flags.ctypes.data_as(C.POINTER(C.c_int32)), int(flags.size), C.byref(n)))
        return flags

    def write_frame_params(self, params):
        """The per-render uniform write (r2f_write_frame_params): params.seed -> the context's device-side frame block, in
        stream order.  Stage calls whose params carry F_FRAME_RESIDENT read it instead of writing their own seed."""
        self._check(self._lib.r2f_write_frame_params(self._h, C.byref(params), self._stream()))

    def set_option(self, name: str, value: int):
        self._check(self._lib.r2f_set_option(self._h, name.encode(), int(value)))

    # ------------------------------------------------------------------ uploads
    def set_matrix3x3(self, m):
        if m is None:
            self._check(self._lib.r2f_set_matrix3x3(self._h, None))
            return
        m = _host_f32(m)
        if m.shape != (3, 3):
            raise ValueError("matrix must be 3x3")
        self._check(self._lib.r2f_set_matrix3x3(self._h, m.ctypes.data))

    def set_lut2d(self, lut):
        lut = _host_f32(lut)
        if lut.ndim != 3 or lut.shape[0] != lut.shape[1] or lut.shape[2] != 3:
            raise ValueError(f"input LUT must be (n, n, 3), got {lut.shape}")
        self._check(self._lib.r2f_set_lut2d(self._h, lut.ctypes.data, lut.shape[0]))

    def set_curve1d(self, lut):
        lut = _host_f32(lut)
        if lut.ndim != 2 or lut.shape[0] != 4:
            raise ValueError(f"density curve must be (4, m), got {lut.shape}")
        self._check(self._lib.r2f_set_curve1d(self._h, lut.ctypes.data, lut.shape[1]))

    def set_lut3d(self, lut):
        lut = _host_f32(lut)
        if lut.ndim != 4 or lut.shape[3] != 3 or not (lut.shape[0] == lut.shape[1] == lut.shape[2]):
            raise ValueError(f"output LUT must be (n, n, n, 3), got {lut.shape}")
        self._check(self._lib.r2f_set_lut3d(self._h, lut.ctypes.data, lut.shape[0]))

    def set_grain_lut(self, lut):
        lut = _host_f32(lut)
        if lut.ndim != 2 or lut.shape[0] != 4:
            raise ValueError(f"grain LUT must be (4, m), got {lut.shape}")
        self._check(self._lib.r2f_set_grain_lut(self._h, lut.ctypes.data, lut.shape[1]))

    def set_kernel(self, which: int, k):
        k = _host_f32(k)
        if k.ndim == 2:
            k = k[..., None]
        if k.ndim != 3 or k.shape[2] not in (1, 3):
            raise ValueError(f"stencil must be (kh, kw) or (kh, kw, 1|3), got {k.shape}")
        k = np.ascontiguousarray(k)
        self._check(self._lib.r2f_set_kernel(self._h, int(which), k.ctypes.data, k.shape[0], k.shape[1], k.shape[2]))

    # ------------------------------------------------------------------ whole frame
    @staticmethod
    def layout_of(t, layout=None) -> tuple[int, int, int]:
        """(layout, H, W) of an image tensor: (H, W, 3) / (H, W, 4) interleaved or (3, H, W) planar.

        A (3, H, 3) or (3, H, 4) tensor is ambiguous; interleaved wins (the reference only knows (H, W, C) frames),
        so pass ``layout="chw"`` for a planar frame that is 3 or 4 pixels wide."""
        if t.dim() != 3:
            raise ValueError("image must be 3-D")
        if layout is not None:
            want = {"hwc3": (_lib.LAYOUT_HWC3, 2, 3), "hwc4": (_lib.LAYOUT_HWC4, 2, 4), "chw": (_lib.LAYOUT_CHW, 0, 3)}.get(layout)
            if want is None or t.shape[want[1]] != want[2]:
                raise ValueError(f"layout {layout!r} does not fit image shape {tuple(t.shape)}")
            hw = (t.shape[1], t.shape[2]) if layout == "chw" else (t.shape[0], t.shape[1])
            return want[0], int(hw[0]), int(hw[1])
        if t.shape[2] == 3:
            return _lib.LAYOUT_HWC3, int(t.shape[0]), int(t.shape[1])
        if t.shape[2] == 4:
            return _lib.LAYOUT_HWC4, int(t.shape[0]), int(t.shape[1])
        if t.shape[0] == 3:
            return _lib.LAYOUT_CHW, int(t.shape[1]), int(t.shape[2])
        raise ValueError(f"cannot interpret image shape {tuple(t.shape)}")

    def _same_device(self, t, what):
        if t.device.index is not None and self.device.index is not None and t.device.index != self.device.index:
            raise ValueError(f"{what} lives on {t.device}, this context on {self.device}")

    def _check_image(self, t):
        torch = self._torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("image must be a contiguous float32 CUDA tensor")
        self._same_device(t, "image")

    def workspace_bytes(self, params, H, W) -> int:
        return int(self._lib.r2f_workspace_bytes(C.byref(params), H, W))

    def _get_workspace(self, nbytes: int):
        torch = self._torch
        if nbytes == 0:
            return None
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._workspace

    def render(self, image, params, out_f32=None, out_u8=None, want_f32=True, want_u8=False, layout=None):
        """Full pipeline on one frame: S0..S8 (+S9).  Returns (out_f32, out_u8) device tensors (H, W, 3)."""
        torch = self._torch
        self._check_image(image)
        layout, H, W = self.layout_of(image, layout)
        if out_f32 is None and want_f32:
            out_f32 = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        if out_u8 is None and want_u8:
            out_u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        self._check_out(out_f32, torch.float32, W, "out_f32", rows=H)
        self._check_out(out_u8, torch.uint8, W, "out_u8", rows=H)
        nbytes = self.workspace_bytes(params, H, W)
        ws = self._get_workspace(nbytes)
        rc = self._lib.r2f_render(
            self._h, C.byref(params), image.data_ptr(), layout,
            out_f32.data_ptr() if out_f32 is not None else None,
            out_u8.data_ptr() if out_u8 is not None else None,
            H, W, ws.data_ptr() if ws is not None else None, nbytes, self._stream(),
        )
        self._check(rc)
        return out_f32, out_u8

    # ------------------------------------------------------------------ stages (row-shard aware)
    def stage_front(self, image, params, upto, *, in_gy0=0, dst=None, dst_gy0=0, out_f32=None, out_u8=None,
                    out_gy0=0, y0=None, y1=None, H_global=None, layout=None, track_range=False):
        """track_range (upto = exposure): the kernel merges the range of the exposure samples it writes for the halation's FFT
        channels into the context's exposure-range record, a grid of 64 x 256-pixel tiles (R2F_F_TRACK_RANGE), like r2f_render's
        front kernel does for a whole frame."""
        self._check_image(image)
        if track_range:
            params = _lib.Params.from_buffer_copy(params)
            params.flags |= _lib.F_TRACK_RANGE
        layout, rows, W = self.layout_of(image, layout)
        y0 = in_gy0 if y0 is None else y0
        y1 = in_gy0 + rows if y1 is None else y1
        H_global = in_gy0 + rows if H_global is None else H_global
        pl = self.planes(dst, dst_gy0) if dst is not None else None
        self._check_out(out_f32, self._torch.float32, W, "out_f32", gy0=out_gy0, y0=y0, y1=y1)
        self._check_out(out_u8, self._torch.uint8, W, "out_u8", gy0=out_gy0, y0=y0, y1=y1)
        rc = self._lib.r2f_stage_front(
            self._h, C.byref(params), image.data_ptr(), layout, in_gy0, rows, int(upto),
            C.byref(pl) if pl is not None else None,
            out_f32.data_ptr() if out_f32 is not None else None,
            out_u8.data_ptr() if out_u8 is not None else None,
            out_gy0, y0, y1, W, H_global, self._stream(),
        )
        self._check(rc)

    def stage_front_split(self, image, params, exposure, density, *, in_gy0=0, exposure_gy0=0, density_gy0=0, y0=None, y1=None,
                          H_global=None, layout=None, track_range=False) -> int:
        """S0 + S1 for a frame that goes on to stage_halation: channels with a real halation stencil -> `exposure`; channels
        whose halation stencil is one tap at the anchor are finished (tap weight, log, curve) straight into `density`.  Returns
        the mask of channels finished that way: pass it to stage_halation(identity_done=mask).  track_range: see stage_front."""
        self._check_image(image)
        if track_range:
            params = _lib.Params.from_buffer_copy(params)
            params.flags |= _lib.F_TRACK_RANGE
        layout, rows, W = self.layout_of(image, layout)
        y0 = in_gy0 if y0 is None else y0
        y1 = in_gy0 + rows if y1 is None else y1
        H_global = in_gy0 + rows if H_global is None else H_global
        pe, pd = self.planes(exposure, exposure_gy0), self.planes(density, density_gy0)
        mask = C.c_int(0)
        self._check(self._lib.r2f_stage_front_split(self._h, C.byref(params), image.data_ptr(), layout, in_gy0, rows, C.byref(pe),
                                                    C.byref(pd), y0, y1, W, H_global, C.byref(mask), self._stream()))
        return mask.value

    def _stencil_call(self, fn, first, src, src_gy0, dst, dst_gy0, y0, y1, H_global):
        ps, pd = self.planes(src, src_gy0), self.planes(dst, dst_gy0)
        W = int(src.shape[2])
        if int(dst.shape[2]) != W:
            raise ValueError("source and destination widths differ")
        self._check(fn(self._h, first, C.byref(ps), C.byref(pd), y0, y1, W, H_global, self._stream()))

    def stage_exposure_range(self, exposure, *, src_gy0=0, y0, y1, y2=0, y3=0):
        """Merge min / max |.| of rows [y0, y1) and [y2, y3) of the exposure planes (the halation's FFT channels) into the exposure-range record's tiles:
        the halo rows a row shard received from above and below, one launch (r2f_stage_exposure_range)."""
        if y1 <= y0 and y3 <= y2:
            return
        pe = self.planes(exposure, src_gy0)
        self._check(self._lib.r2f_stage_exposure_range(self._h, C.byref(pe), int(y0), int(y1), int(y2), int(y3), int(exposure.shape[2]),
                                                       self._stream()))

    def stage_halation(self, exposure, density, params, *, src_gy0=0, dst_gy0=0, y0, y1, H_global, identity_done=0, range_valid=False):
        """range_valid: the caller has kept the exposure-range record for the rows `exposure` holds this frame (R2F_F_RANGE_VALID): the
        FFT passes then choose their scratch element on the device, window pair by window pair, like r2f_render's."""
        if identity_done or range_valid:
            params = _lib.Params.from_buffer_copy(params)
        if identity_done:  # stage_front_split already wrote the identity channels' density for these rows
            params.flags |= _lib.F_IDENTITY_DONE
        if range_valid:
            params.flags |= _lib.F_RANGE_VALID
        self._stencil_call(self._lib.r2f_stage_halation, C.byref(params), exposure, src_gy0, density, dst_gy0, y0, y1, H_global)

    def stage_mtf(self, density_in, density_out, params, *, src_gy0=0, dst_gy0=0, y0, y1, H_global):
        self._stencil_call(self._lib.r2f_stage_mtf, C.byref(params), density_in, src_gy0, density_out, dst_gy0, y0, y1, H_global)

    def stage_stencil(self, which, src, dst, *, src_gy0=0, dst_gy0=0, y0, y1, H_global):
        self._stencil_call(self._lib.r2f_stage_stencil, int(which), src, src_gy0, dst, dst_gy0, y0, y1, H_global)

    def stage_tail(self, density, params, *, src_gy0=0, out_f32=None, out_u8=None, out_gy0=0, y0, y1, H_global,
                   burn_map=None):
        pd = self.planes(density, src_gy0)
        W = int(density.shape[2])
        self._check_out(out_f32, self._torch.float32, W, "out_f32", gy0=out_gy0, y0=y0, y1=y1)
        self._check_out(out_u8, self._torch.uint8, W, "out_u8", gy0=out_gy0, y0=y0, y1=y1)
        if burn_map is not None:
            self._check_lowres(burn_map, params, H_global, W, "burn_map")
        rc = self._lib.r2f_stage_tail(
            self._h, C.byref(params), C.byref(pd), burn_map.data_ptr() if burn_map is not None else None,
            out_f32.data_ptr() if out_f32 is not None else None,
            out_u8.data_ptr() if out_u8 is not None else None,
            out_gy0, y0, y1, W, H_global, self._stream(),
        )
        self._check(rc)

    def stage_grain_field(self, field, params, *, dst_gy0=0, y0, y1, H_global):
        """The grain field K_g * N for rows [y0, y1) -> (3, rows, W) planes; no image involved."""
        pf = self.planes(field, dst_gy0)
        self._check(self._lib.r2f_stage_grain_field(self._h, C.byref(params), C.byref(pf), y0, y1, int(field.shape[2]), H_global,
                                                    self._stream()))

    def stage_tail_field(self, density, field, params, *, src_gy0=0, field_gy0=0, out_f32=None, out_u8=None, out_gy0=0, y0, y1,
                         H_global):
        """stage_tail with a grain field made by stage_grain_field instead of generating it in the same kernel."""
        pd, pf = self.planes(density, src_gy0), self.planes(field, field_gy0)
        self._check_out(out_f32, self._torch.float32, int(density.shape[2]), "out_f32", gy0=out_gy0, y0=y0, y1=y1)
        self._check_out(out_u8, self._torch.uint8, int(density.shape[2]), "out_u8", gy0=out_gy0, y0=y0, y1=y1)
        self._check(self._lib.r2f_stage_tail_field(
            self._h, C.byref(params), C.byref(pd), C.byref(pf),
            out_f32.data_ptr() if out_f32 is not None else None, out_u8.data_ptr() if out_u8 is not None else None,
            out_gy0, y0, y1, int(density.shape[2]), H_global, self._stream()))

    def stage_grain(self, density_in, density_out, params, *, src_gy0=0, dst_gy0=0, y0, y1, H_global):
        """S6 + clip alone (planes -> planes): first half of the tail when S7 is on."""
        self._stencil_call(self._lib.r2f_stage_grain, C.byref(params), density_in, src_gy0, density_out, dst_gy0, y0, y1, H_global)

    def burn_shape(self, params, H_global, W):
        return H_global // params.burn_cell, W // params.burn_cell

    def _check_lowres(self, t, params, H_global, W, what):
        torch = self._torch
        shape = self.burn_shape(params, H_global, W)
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"{what} must be a contiguous float32 CUDA tensor of shape {shape} (the highlight burn's low-resolution grid)")
        self._same_device(t, what)

    def stage_burn_sums(self, density, params, *, src_gy0=0, y0, y1, H_global):
        """S7 part 1: area-weighted partial sums of the green density over rows [y0, y1) -> (h_lo, w_lo) tensor."""
        torch = self._torch
        W = int(density.shape[2])
        h_lo, w_lo = self.burn_shape(params, H_global, W)
        sums = torch.empty((h_lo, w_lo), dtype=torch.float32, device=self.device)
        pd = self.planes(density, src_gy0)
        self._check(self._lib.r2f_stage_burn_sums(self._h, C.byref(params), C.byref(pd), sums.data_ptr(), y0, y1, W, H_global,
                                                  self._stream()))
        return sums

    def stage_burn_map(self, sums, params, *, W, H_global):
        """S7 part 2: clip(x - d_ref, 0) + Gaussian(sigma 3) on the low-res map."""
        torch = self._torch
        self._check_lowres(sums, params, H_global, W, "sums")
        out = torch.empty_like(sums)
        scratch = torch.empty((2,) + tuple(sums.shape), dtype=torch.float32, device=self.device)
        self._check(self._lib.r2f_stage_burn_map(self._h, C.byref(params), sums.data_ptr(), out.data_ptr(), scratch.data_ptr(), W,
                                                 H_global, self._stream()))
        return out

    def stage_chroma_nr_h(self, image, dst, size, *, in_gy0=0, dst_gy0=0, y0=None, y1=None, layout=None):
        """Pre-path chroma NR pass 1: image rows -> planes (x blurred horizontally, y blurred horizontally, Y)."""
        self._check_image(image)
        layout, rows, W = self.layout_of(image, layout)
        y0 = in_gy0 if y0 is None else y0
        y1 = in_gy0 + rows if y1 is None else y1
        pd = self.planes(dst, dst_gy0)
        self._check(self._lib.r2f_stage_chroma_nr_h(self._h, image.data_ptr(), layout, in_gy0, rows, C.byref(pd), int(size), y0, y1,
                                                    W, self._stream()))

    def stage_chroma_nr_v(self, src, dst, size, *, src_gy0=0, dst_gy0=0, y0, y1, H_global):
        """Pre-path chroma NR pass 2: vertical blur + xyY -> XYZ planes (CHW input for the front stage)."""
        ps, pd = self.planes(src, src_gy0), self.planes(dst, dst_gy0)
        self._check(self._lib.r2f_stage_chroma_nr_v(self._h, C.byref(ps), C.byref(pd), int(size), y0, y1, int(src.shape[2]),
                                                    H_global, self._stream()))

    def chroma_nr(self, image, size, layout=None):
        """effects.chroma_nr_filter on a whole frame: (H, W, 3|4) or (3, H, W) in -> (3, H, W) XYZ planes out."""
        torch = self._torch
        _, H, W = self.layout_of(image, layout)
        tmp = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
        out = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
        self.stage_chroma_nr_h(image, tmp, size, layout=layout)
        self.stage_chroma_nr_v(tmp, out, size, y0=0, y1=H, H_global=H)
        return out

    def resize_area(self, image, out_h, out_w, layout=None):
        """Pre-path INTER_AREA down-scale of a whole frame -> (3, out_h, out_w) planes."""
        torch = self._torch
        self._check_image(image)
        layout, H, W = self.layout_of(image, layout)
        out = torch.empty((3, int(out_h), int(out_w)), dtype=torch.float32, device=self.device)
        pd = self.planes(out, 0)
        self._check(self._lib.r2f_resize_area(self._h, image.data_ptr(), layout, H, W, C.byref(pd), int(out_h), int(out_w),
                                              self._stream()))
        return out

    def resize_lanczos4_f32(self, image, out_h, out_w, layout=None):
        """Pre-path cv.resize(float32 frame, INTER_LANCZOS4) up-scale of a whole frame -> (3, out_h, out_w) planes."""
        torch = self._torch
        self._check_image(image)
        layout, H, W = self.layout_of(image, layout)
        out = torch.empty((3, int(out_h), int(out_w)), dtype=torch.float32, device=self.device)
        pd = self.planes(out, 0)
        self._check(self._lib.r2f_resize_lanczos4_f32(self._h, image.data_ptr(), layout, H, W, C.byref(pd), int(out_h), int(out_w),
                                                      self._stream()))
        return out

    def warp_affine(self, image, m_dst_to_src, window=None, layout=None):
        """cv.warpAffine(image, M, same size, INTER_LINEAR) restricted to `window` = (row0, col0, rows, cols) -> (3, rows, cols)
        planes.  m_dst_to_src: inverse of M, 2 x 3 (geometry.rotation_plan)."""
        torch = self._torch
        self._check_image(image)
        layout, H, W = self.layout_of(image, layout)
        r0, c0, nr, nc = (0, 0, H, W) if window is None else [int(v) for v in window]
        out = torch.empty((3, nr, nc), dtype=torch.float32, device=self.device)
        if nr == 0 or nc == 0:
            return out
        m = np.ascontiguousarray(np.asarray(m_dst_to_src, dtype=np.float64).reshape(6))
        pd = self.planes(out, 0)
        self._check(self._lib.r2f_warp_affine(self._h, image.data_ptr(), layout, H, W, m.ctypes.data, C.byref(pd), nr, nc, r0, c0,
                                              self._stream()))
        return out

    def resize_lanczos4_u8(self, image_u8, out_h: int, out_w: int):
        """cv.resize(uint8 (H, W, 3), (out_w, out_h), interpolation=cv.INTER_LANCZOS4) on the device."""
        torch = self._torch
        if not (image_u8.is_cuda and image_u8.dtype == torch.uint8 and image_u8.is_contiguous() and image_u8.dim() == 3
                and image_u8.shape[2] == 3):
            raise ValueError("resize_lanczos4_u8 needs a contiguous uint8 (H, W, 3) CUDA tensor")
        self._same_device(image_u8, "image")
        out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.uint8, device=self.device)
        self._check(self._lib.r2f_resize_lanczos4_u8(self._h, image_u8.data_ptr(), int(image_u8.shape[0]), int(image_u8.shape[1]),
                                                     out.data_ptr(), int(out_h), int(out_w), self._stream()))
        return out

    def resize_area_u8(self, image_u8, out_h: int, out_w: int):
        """cv.resize(uint8 (H, W, 3), (out_w, out_h), interpolation=cv.INTER_AREA), shrinking, on the device (utils.py:226-236 as
        the CPU processor applies it to the finished frame, cpu_processor.py:411-412)."""
        torch = self._torch
        if not (image_u8.is_cuda and image_u8.dtype == torch.uint8 and image_u8.is_contiguous() and image_u8.dim() == 3
                and image_u8.shape[2] == 3):
            raise ValueError("resize_area_u8 needs a contiguous uint8 (H, W, 3) CUDA tensor")
        self._same_device(image_u8, "image")
        out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.uint8, device=self.device)
        self._check(self._lib.r2f_resize_area_u8(self._h, image_u8.data_ptr(), int(image_u8.shape[0]), int(image_u8.shape[1]),
                                                 out.data_ptr(), int(out_h), int(out_w), self._stream()))
        return out

    def decode_u16(self, image_u16, factor: float, divisor: float = 65535.0, out=None):
        """raw_to_linear's last two lines (raw_conversion.py:50-52) on the device: float32(u) / divisor * float32(factor) for a
        uint16 (H, W, 3 | 4) CUDA tensor (LibRaw's 16-bit output; int16 tensors are read as the same bits) -> float32 (H, W, 3).
        out: a contiguous float32 (H, W, 3) CUDA tensor to write into (e.g. a band of rows of a larger frame)."""
        torch = self._torch
        if not (image_u16.is_cuda and image_u16.dtype in (torch.uint16, torch.int16) and image_u16.is_contiguous()
                and image_u16.dim() == 3 and image_u16.shape[2] in (3, 4)):
            raise ValueError("decode_u16 needs a contiguous uint16 (H, W, 3 or 4) CUDA tensor")
        self._same_device(image_u16, "image")
        shape = (int(image_u16.shape[0]), int(image_u16.shape[1]), 3)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == shape):
            raise ValueError(f"decode_u16: out must be a contiguous float32 CUDA tensor of shape {shape}")
        else:
            self._same_device(out, "out")
        self._check(self._lib.r2f_decode_u16(self._h, image_u16.data_ptr(), int(image_u16.shape[0]), int(image_u16.shape[1]),
                                             int(image_u16.shape[2]), float(np.float32(divisor)), float(np.float32(factor)),
                                             out.data_ptr(), self._stream()))
        return out

    def blit_rgba8(self, image_f32, dst_rgba, transform: dict):
        """shaders/copy_to_int.wgsl: the float (H, W, 3) frame letterboxed into the uint8 (h, w, 4) destination tensor.
        `transform`: the dict of geometry.blit_transform (the shader's uniform block)."""
        torch = self._torch
        if not (image_f32.is_cuda and image_f32.dtype == torch.float32 and image_f32.is_contiguous() and image_f32.dim() == 3
                and image_f32.shape[2] == 3):
            raise ValueError("blit_rgba8 needs a contiguous float32 (H, W, 3) CUDA tensor")
        if not (dst_rgba.is_cuda and dst_rgba.dtype == torch.uint8 and dst_rgba.is_contiguous() and dst_rgba.dim() == 3
                and dst_rgba.shape[2] == 4):
            raise ValueError("blit_rgba8 needs a contiguous uint8 (h, w, 4) CUDA destination")
        self._same_device(image_f32, "image")
        self._same_device(dst_rgba, "destination")
        t = _lib.Blit(transform["scale_x"], transform["scale_y"], transform["offset_x"], transform["offset_y"],
                      transform["canvas_min_x"], transform["canvas_min_y"], transform["canvas_max_x"], transform["canvas_max_y"],
                      (C.c_float * 3)(*transform["canvas_color"]))
        self._check(self._lib.r2f_blit_rgba8(self._h, image_f32.data_ptr(), int(image_f32.shape[0]), int(image_f32.shape[1]),
                                             dst_rgba.data_ptr(), int(dst_rgba.shape[0]), int(dst_rgba.shape[1]), C.byref(t),
                                             self._stream()))
        return dst_rgba

    def histogram_render(self, counts, mix_table, height: int, target=None):
        """histogram.wgsl passes 2 + 3 (+ scale_texture.wgsl into `target`, a uint8 (h, w, 4) device tensor) from the (3, 256)
        device counts of `histogram_counts`; returns the (height, 256, 4) uint8 device bar image."""
        torch = self._torch
        mix = np.ascontiguousarray(np.asarray(mix_table, dtype=np.uint8).reshape(8, 4))
        image = torch.empty((int(height), 256, 4), dtype=torch.uint8, device=self.device)
        if not (isinstance(counts, torch.Tensor) and counts.is_cuda and counts.dtype in (torch.int32, torch.uint32)
                and tuple(counts.shape) == (3, 256)):
            raise ValueError("histogram_render needs the (3, 256) int32 CUDA counts of histogram_counts")
        self._same_device(counts, "counts")
        counts = counts.contiguous()
        if target is not None and not (target.is_cuda and target.dtype == torch.uint8 and target.is_contiguous() and target.dim() == 3
                                       and target.shape[2] == 4):
            raise ValueError("histogram_render needs a contiguous uint8 (h, w, 4) CUDA target")
        if target is not None:
            self._same_device(target, "target")
        self._check(self._lib.r2f_histogram_render(
            self._h, counts.data_ptr(), mix.ctypes.data, int(height), image.data_ptr(),
            target.data_ptr() if target is not None else None, int(target.shape[0]) if target is not None else 0,
            int(target.shape[1]) if target is not None else 0, self._stream()))
        return image

    def stencil_stats(self, which: int):
        """Per channel: dict(entries, rowsteps, phases, sym, unrolled, kh, kw, q, fft, window, real_spectrum) of the device form of
        stencil `which` (bench.py); fft = 1: the channel takes the FFT form, window = (rows, columns) of its last launch or None."""
        out = (C.c_int * 24)()
        self._check(self._lib.r2f_stencil_stats(self._h, int(which), out))
        keys = ("entries", "rowsteps", "phases", "sym", "kh", "kw", "q")
        stats = []
        for c in range(3):
            d = dict(zip(keys, out[8 * c:8 * c + 7]))
            word = out[8 * c + 7]
            d["unrolled"] = (d["sym"] >> 1) & 0x7F  # R of the fully unrolled (2 R + 1)^2 direct form, 0: the entry list
            d["separable"] = (d["sym"] >> 8) & 1  # grain stencil only: two 1-D passes of 2 R + 1 taps
            d["sym"] &= 1
            d["fft"] = word & 1
            dims = (word >> 1) & 0x1FFFFFFF
            d["window"] = (dims // 4096, dims % 4096) if dims else None
            d["real_spectrum"] = (word >> 30) & 1  # its last FFT launch multiplied by a real kernel spectrum (centred symmetric taps)
            stats.append(d)
        return stats

    def stream_copy(self, src, dst):
        """dst <- src with a float4 streaming kernel (bench.py's copy ceiling: 2 x the bytes of HBM traffic, no arithmetic)."""
        nbytes = src.numel() * src.element_size()
        if dst.numel() * dst.element_size() != nbytes or not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("stream_copy needs two contiguous CUDA tensors of the same size in bytes")
        self._same_device(src, "source")
        self._same_device(dst, "destination")
        self._check(self._lib.r2f_stream_copy(self._h, src.data_ptr(), dst.data_ptr(), nbytes, self._stream()))

    def kernel_timing(self, cls: int):
        """(total ms, launches, algorithmic bytes) of FFT pass `cls` since the last call; needs set_option("kernel_timing", 1)."""
        ms, n, b = C.c_double(), C.c_int(), C.c_double()
        self._check(self._lib.r2f_kernel_timing(self._h, int(cls), C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def histogram_counts(self, image_u8):
        """Per-channel bin counts of a uint8 (H, W, 3) device image -> int32 (3, 256) device tensor (utils.py:160-165)."""
        torch = self._torch
        if not (image_u8.is_cuda and image_u8.dtype == torch.uint8 and image_u8.is_contiguous() and image_u8.dim() == 3
                and image_u8.shape[2] == 3):
            raise ValueError("histogram_counts needs a contiguous uint8 (H, W, 3) CUDA tensor")
        self._same_device(image_u8, "image")
        counts = torch.empty((3, 256), dtype=torch.int32, device=self.device)
        self._check(self._lib.r2f_histogram_u8(self._h, image_u8.data_ptr(), int(image_u8.shape[0]), int(image_u8.shape[1]),
                                               counts.data_ptr(), self._stream()))
        return counts

    def stage_noise(self, params, y0, y1, W, want_hash=True, want_noise=True):
        torch = self._torch
        rows = y1 - y0
        h = torch.empty((3, rows, W), dtype=torch.int32, device=self.device) if want_hash else None
        n = torch.empty((3, rows, W), dtype=torch.float32, device=self.device) if want_noise else None
        rc = self._lib.r2f_stage_noise(
            self._h, C.byref(params), h.data_ptr() if h is not None else None,
            n.data_ptr() if n is not None else None, y0, y1, W, self._stream(),
        )
        self._check(rc)
        return h, n
