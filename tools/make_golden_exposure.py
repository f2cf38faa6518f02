#!/usr/bin/env python3
"""tests/golden/exposure.npz: the reference's own color_processing.calc_exposure (color_processing.py:71-99) -- the auto exposure
raw_to_linear applies to the decoded frame (raw_conversion.py:50-52) -- on small random frames with the metadata variants it
branches on.

    python3 -B tools/make_golden_exposure.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REF_SRC)
    import raw2film.color_processing as cp  # noqa: E402

    rng = np.random.default_rng(20261003)
    metas = [None,
             {"EXIF:FNumber": 8.0, "EXIF:ISO": 100, "EXIF:ExposureTime": 1 / 250},
             {"EXIF:FNumber": 1.8, "EXIF:ISO": 3200, "EXIF:ExposureTime": 1 / 30},
             {"EXIF:FNumber": "undef", "EXIF:ISO": 400, "EXIF:ExposureTime": 0.01},
             {"EXIF:ISO": 200, "EXIF:ExposureTime": 2.0},
             {"EXIF:FNumber": 0, "EXIF:ISO": 800, "EXIF:ExposureTime": 1 / 1000}]
    out = {}
    i = 0
    for si, (h, w) in enumerate([(40, 60), (61, 37), (120, 180), (1, 1), (2, 5)]):
        u16 = rng.integers(0, 65536, (h, w, 3)).astype(np.uint16)
        if h > 10:
            u16[: h // 2] //= 40  # a dark half: the mean of roots is not the root of the mean
        out[f"frame_{si}"] = u16
        rgb = u16.astype(np.float32) / 65535.0  # raw_conversion.py:50
        for mi, meta in enumerate(metas):
            out[f"case_{i}"] = np.array([si, mi])
            out[f"exp_{i}"] = np.array(cp.calc_exposure(rgb, metadata=meta), dtype=np.float64)
            i += 1
    out["n"] = np.array(i)
    out["metas"] = np.array([repr(m) for m in metas])
    path = os.path.join(mg.OUT_DIR, "exposure.npz")
    np.savez_compressed(path, **out)
    print(path, i, "cases; e.g.", [float(out[f"exp_{k}"]) for k in range(3)])


if __name__ == "__main__":
    main()
