"""ctypes binding of libr2f_hip.so (C ABI declared in include/r2f.h).

The product path has NO CPU fallback: if the library is missing or a symbol is absent the
import of the binding raises, and every render call goes through these entry points.
"""

from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libr2f_hip.so")

# enums of include/r2f.h
LAYOUT_HWC3, LAYOUT_HWC4, LAYOUT_CHW = 0, 1, 2
KERNEL_HALATION, KERNEL_MTF, KERNEL_GRAIN = 0, 1, 2
F_MATRIX, F_HALATION, F_MTF, F_GRAIN, F_GRAIN_MONO, F_BURN, F_IDENTITY_DONE, F_FRAME_RESIDENT = 1, 2, 4, 8, 16, 32, 64, 128
F_TRACK_RANGE, F_RANGE_VALID = 256, 512
UPTO_EXPOSURE, UPTO_DENSITY, UPTO_OUTPUT = 0, 1, 2
OK, EINVAL, EHIP, ETOOLARGE = 0, -1, -2, -3


class Params(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("seed", C.c_uint32),
        ("log_eps", C.c_float),
        ("lut3d_scale", C.c_float),
        ("lut3d_mode", C.c_int32),
        ("burn_cell", C.c_int32),
        ("burn_strength", C.c_float),
        ("burn_d_ref", C.c_float),
    ]


class Planes(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("plane_stride", C.c_int64),
        ("gy0", C.c_int32),
        ("rows", C.c_int32),
    ]


class FftPlan(C.Structure):  # r2f_fft_plan
    _fields_ = [("ny", C.c_int32), ("nx", C.c_int32), ("vy", C.c_int32), ("vx", C.c_int32), ("gx", C.c_int32), ("ntiles", C.c_int32),
                ("pairs_per_channel", C.c_int32), ("pairs", C.c_int32), ("streams", C.c_int32), ("batch", C.c_int32),
                ("launches", C.c_int32), ("scratch_bytes", C.c_uint64)]


class Blit(C.Structure):  # r2f_blit: the uniform block of shaders/copy_to_int.wgsl
    _fields_ = [
        ("scale_x", C.c_float), ("scale_y", C.c_float), ("offset_x", C.c_float), ("offset_y", C.c_float),
        ("canvas_min_x", C.c_float), ("canvas_min_y", C.c_float), ("canvas_max_x", C.c_float), ("canvas_max_y", C.c_float),
        ("canvas_color", C.c_float * 3),
    ]


_P = C.POINTER
_fp = C.c_void_p  # float* (host numpy or device) passed as an address
_SIGNATURES = {
    "r2f_version": (C.c_char_p, []),
    "r2f_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "r2f_destroy": (None, [C.c_void_p]),
    "r2f_last_error": (C.c_char_p, [C.c_void_p]),
    "r2f_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "r2f_set_matrix3x3": (C.c_int, [C.c_void_p, _fp]),
    "r2f_set_lut2d": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "r2f_set_curve1d": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "r2f_set_lut3d": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "r2f_set_grain_lut": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "r2f_set_kernel": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int, C.c_int, C.c_int]),
    "r2f_workspace_bytes": (C.c_size_t, [_P(Params), C.c_int, C.c_int]),
    "r2f_render": (
        C.c_int,
        [C.c_void_p, _P(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "r2f_stage_front": (
        C.c_int,
        [C.c_void_p, _P(Params), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _P(Planes), C.c_void_p, C.c_void_p,
         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_halation": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_exposure_range": (C.c_int, [C.c_void_p, _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r2f_stage_mtf": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_tail": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_void_p],
    ),
    "r2f_stage_grain": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_burn_sums": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_burn_map": (
        C.c_int,
        [C.c_void_p, _P(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_chroma_nr_h": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_chroma_nr_v": (
        C.c_int,
        [C.c_void_p, _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_resize_area": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _P(Planes), C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_noise": (
        C.c_int,
        [C.c_void_p, _P(Params), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stage_stencil": (
        C.c_int,
        [C.c_void_p, C.c_int, _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_warp_affine": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_resize_lanczos4_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "r2f_lanczos4_table": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "r2f_kernel_timing": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double), _P(C.c_int), _P(C.c_double)]),
    "r2f_stage_grain_field": (C.c_int, [C.c_void_p, _P(Params), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r2f_stage_tail_field": (
        C.c_int,
        [C.c_void_p, _P(Params), _P(Planes), _P(Planes), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_stencil_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "r2f_histogram_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "r2f_stage_front_split": (
        C.c_int,
        [C.c_void_p, _P(Params), C.c_void_p, C.c_int, C.c_int, C.c_int, _P(Planes), _P(Planes), C.c_int, C.c_int, C.c_int, C.c_int,
         _P(C.c_int), C.c_void_p],
    ),
    "r2f_resize_lanczos4_f32": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _P(Planes), C.c_int, C.c_int, C.c_void_p],
    ),
    "r2f_lanczos4_table_f32": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "r2f_resize_area_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "r2f_decode_u16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "r2f_blit_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, _P(Blit), C.c_void_p]),
    "r2f_plan_fft": (C.c_int, [C.c_int] * 11 + [_P(FftPlan)]),
    "r2f_plan_stencil": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r2f_plan_tile_order": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "r2f_generation": (C.c_uint64, [C.c_void_p]),
    "r2f_render_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "r2f_write_frame_params": (C.c_int, [C.c_void_p, _P(Params), C.c_void_p]),
    "r2f_frame_exposure_range": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_int), _P(C.c_int)]),
    "r2f_frame_scratch_choice": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int)]),
    "r2f_frame_scratch_flags": (C.c_int, [C.c_void_p, _P(C.c_int32), C.c_int, _P(C.c_int)]),
    "r2f_stream_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "r2f_histogram_render": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    ),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _bind(path: str) -> C.CDLL:
    # torch FIRST: its wheel ships a HIP runtime of its own (torch/lib/libamdhip64.so), and the dynamic loader keeps whichever
    # libamdhip64 a process met first.  Loaded before torch, this library would pull in the system's runtime (/opt/rocm/lib) and torch
    # would then run on that one instead of the one it was built against -- on the GPU box r2f_create then fails with R2F_EHIP
    # (found with `python __graft_entry__.py --smoke`, i.e. build() -- which binds the library -- and smoke() in ONE process).  The
    # Python layer needs torch anyway (device buffers, streams); a host that only wants the C ABI binds the .so itself.
    try:
        import torch  # noqa: F401
    except ImportError:  # (the plan-only entry points and the symbol tests work without it)
        pass
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def load(path: str | None = None) -> C.CDLL:
    """dlopen the in-tree library and bind every symbol of include/r2f.h; raises if absent.
    path: ANOTHER build of the library (a development variant, tools/build_variant.py) bound beside the in-tree one -- the
    library exports nothing but its C ABI, so two builds in one process do not interpose each other's internals."""
    global _lib
    if path is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing")
        return _bind(os.path.abspath(path))
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m raw2film_amd.build` (hipcc, gfx950). "
            "raw2film_amd has no CPU fallback."
        )
    _lib = _bind(LIB_PATH)
    return _lib
