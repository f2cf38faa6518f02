#!/usr/bin/env python3
"""tests/golden/payload_geometry.npz from the reference's own GpuProcessor.extract_image_data_cpu (gpu_processor.py:715-783).

    python3 -B tools/make_golden_payload.py      # dev container only; -B: never write __pycache__ into /root/reference

What phase 1 derives from a frame's shape and the load settings -- `output_resolution`, `canvas_resolution`,
`pipeline_resolution` and the shape of `image_array` -- for frames coarser and finer than `max_scale` (small-gauge formats),
with and without a preview resolution, in every canvas mode.  The method is called unbound on a bare object whose RAW loader
returns a zero frame of the requested shape; cv2.resize is a stub that returns a zero frame of the requested dsize (only shapes
are recorded), the other absent third-party modules are the inert stubs of tools/make_golden.py / make_golden_blit.py."""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden  # noqa: E402  (stub installer)

make_golden._install_stubs()
from make_golden_blit import _Bag  # noqa: E402,F401  (also installs the wgpu stub and imports gpu_processor)
import raw2film.gpu_processor as gp  # noqa: E402
import raw2film.utils as ru  # noqa: E402


def _resize(image, dsize, interpolation=None):
    return np.zeros((dsize[1], dsize[0]) + image.shape[2:], dtype=image.dtype)


for mod in (sys.modules["cv2"], ):
    mod.resize = _resize
ru.cv.resize = _resize

MODES = ["No", "Proportional white", "Proportional black", "Fixed white", "Fixed grey", "Uniform black", "Uniform white"]
cases, outs = [], []
for (H, W) in ((400, 600), (600, 400), (1001, 1499), (333, 517)):
    for (fw, fh) in ((36.0, 24.0), (5.79, 4.01), (10.26, 7.49), (24.0, 36.0)):  # 135, super 8, 16 mm, portrait
        for resolution in (None, (200, 300), (2000, 3000)):
            for max_scale in (400.0, 40.0, 15.0, None):
                for mi, mode in enumerate(MODES):
                    for (cs, cr) in ((1.0, 1.0), (1.2, 1.5), (1.07, 0.8)):
                        if mode == "No" and (cs, cr) != (1.0, 1.0):
                            continue
                        if resolution is None and max_scale is None:
                            continue
                        self = types.SimpleNamespace(load_raw_image=lambda *a, **k: np.zeros((H, W, 3), dtype=np.float32))
                        self.load_raw_image_cached = self.load_raw_image
                        p = gp.GpuProcessor.extract_image_data_cpu(
                            self, "x", None, None, True, fw, fh, 0.0, 1.0, 0, False, resolution, True, False, 0, max_scale, mode, cs, cr)
                        cases.append([H, W, fw, fh, -1 if resolution is None else resolution[0], -1 if resolution is None else resolution[1],
                                      -1 if max_scale is None else max_scale, mi, cs, cr])
                        can = p["canvas_resolution"] or (-1, -1)
                        outs.append(list(p["output_resolution"]) + list(can) + list(p["pipeline_resolution"]) + list(p["image_array"].shape))
out = os.path.join(make_golden.OUT_DIR, "payload_geometry.npz")
np.savez_compressed(out, cases=np.array(cases, dtype=np.float64), results=np.array(outs, dtype=np.int64), modes=np.array(MODES))
print(f"wrote {out}: {len(cases)} cases")
