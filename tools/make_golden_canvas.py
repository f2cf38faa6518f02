#!/usr/bin/env python3
"""tests/golden/add_canvas.npz: outputs of the reference's own effects.add_canvas (effects.py:338-357) -- the uint8 frame
pasted onto its canvas -- for every canvas mode on small random frames.

    python3 -B tools/make_golden_canvas.py
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
    import raw2film.effects as ref_effects  # noqa: E402
    from typing import get_args

    modes = [m for m in get_args(ref_effects.CANVAS_MODES)]
    rng = np.random.default_rng(20261003)
    out = {"modes": np.array(modes)}
    i = 0
    for (h, w) in [(20, 30), (31, 18)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for mi, mode in enumerate(modes):
            for scale, ratio in [(1.0, 1.0), (1.25, 1.5), (1.1, 0.8)]:
                res = ref_effects.add_canvas(img, mode, scale, ratio)
                out[f"case_{i}"] = np.array([h, w, mi, scale, ratio], dtype=np.float64)
                out[f"image_{i}"] = img
                out[f"out_{i}"] = np.asarray(res)
                i += 1
    out["n"] = np.array(i)
    path = os.path.join(mg.OUT_DIR, "add_canvas.npz")
    np.savez_compressed(path, **out)
    print(path, i, "cases", modes)


if __name__ == "__main__":
    main()
