#!/usr/bin/env python3
"""tests/golden/blit_transform.npz from the reference's own GpuProcessor._bind_copy_to_dst (gpu_processor.py:1416-1539).

    python3 -B tools/make_golden_blit.py        # dev container only; -B: never write __pycache__ into /root/reference

gpu_processor.py is imported with inert stubs for the modules this image lacks (wgpu, spectral_film_lut, cv2, numba, ...: the
same recipe as tools/make_golden.py plus a `wgpu` stub whose enums are attribute bags).  The method is called unbound on a
bare object carrying the attributes it reads; its device calls go to a recorder that keeps the bytes of the uniform buffer,
i.e. the twelve floats the shader receives.  Stored: the inputs of every case and those floats (float32, as packed)."""
import os
import struct
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402  (stub installer)

make_golden._install_stubs()


class _Bag(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        b = _Bag(self.__name__ + "." + name)
        setattr(self, name, b)
        return b

    def __call__(self, *a, **k):
        return None

    def __or__(self, other):
        return self

    __ror__ = __or__


sys.modules["wgpu"] = _Bag("wgpu")
cs = types.ModuleType("spectral_film_lut.color_space")
cs.GAMMA_KEYS = {}
sys.modules["spectral_film_lut.color_space"] = cs
sys.modules["spectral_film_lut.grain_generation"].grain_kernel = None
sys.modules["cv2"].resize = None
sys.path.insert(0, make_golden.REF_SRC)
import raw2film.gpu_processor as gp  # noqa: E402


class _Tex:
    def __init__(self, w, h):
        self.size = (w, h, 1)

    def create_view(self):
        return None


class _Dev:
    def __init__(self):
        self.data = None

    def create_buffer_with_data(self, data, usage):
        self.data = bytes(data)
        return None

    def create_bind_group(self, layout, entries):
        return None


class _Pipe:
    def get_bind_group_layout(self, i):
        return None


def run(src, dst, pipeline_resolution, output_resolution, canvas_resolution, canvas_color):
    self = types.SimpleNamespace(device=_Dev(), pipeline_copy_to_int=_Pipe(), image_sampler=None, canvas_color=canvas_color)
    if pipeline_resolution is not None:
        self.pipeline_resolution = pipeline_resolution
    if output_resolution is not None:
        self.output_resolution = output_resolution
    if canvas_resolution is not None:
        self.canvas_resolution = canvas_resolution
    gp.GpuProcessor._bind_copy_to_dst(self, _Tex(*src), _Tex(*dst))
    return np.array(struct.unpack("ffffffffffff", self.device.data), dtype=np.float32)


cases, outs = [], []
NONE = (-1, -1)
for src in ((600, 400), (400, 600), (1234, 777)):
    for dst in ((300, 200), (200, 300), (512, 512), (1001, 333)):
        for out_res, can_res in ((None, None), (src, None), (src, (int(src[0] * 1.25), int(src[1] * 1.1))),
                                 ((src[0] // 2, src[1] // 2), (src[0], src[1] + 100)), (None, (src[0] + 50, src[1] + 50))):
            for color in ((255, 255, 255), (0, 0, 0), (0.2, 0.4, 0.9), None):
                got = run(src, dst, src, out_res, can_res, color)
                cases.append(list(src) + list(dst) + list(out_res or NONE) + list(can_res or NONE)
                             + list(color if color is not None else (-1, -1, -1)))
                outs.append(got)
out = os.path.join(make_golden.OUT_DIR, "blit_transform.npz")
np.savez_compressed(out, cases=np.array(cases, dtype=np.float64), uniforms=np.array(outs, dtype=np.float32))
print(f"wrote {out}: {len(cases)} cases")
