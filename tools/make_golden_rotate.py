#!/usr/bin/env python3
"""tests/golden/rotate_crop.npz: the crop that the reference's own effects.rotate (effects.py:46-75) applies after its
cv.warpAffine, for a grid of frame shapes and angles.  OpenCV is absent here, so cv.warpAffine is stubbed by the identity
and cv.getRotationMatrix2D by a recorder: what is pinned is the reference's window arithmetic (and the centre / angle it
hands to OpenCV), not the interpolation.

    python3 -B tools/make_golden_rotate.py
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
    import cv2  # the stub module
    import raw2film.effects as ref_effects  # noqa: E402

    calls = []
    cv2.getRotationMatrix2D = lambda center, angle, scale: calls.append((center, angle, scale)) or np.zeros((2, 3))
    cv2.warpAffine = lambda img, m, dsize, flags=None: img
    rows = []
    for (h, w) in [(24, 36), (36, 24), (100, 150), (151, 100), (64, 64), (4000, 6000), (8192, 12288), (333, 517), (517, 333)]:
        for deg in [0.5, -0.5, 1.0, 3.3, -7.25, 15.0, 44.9, -45.0, 89.0]:
            # an index image: the value at (y, x) is y * w + x, so the kept window can be read back from the corners
            idx = np.arange(h * w, dtype=np.int64).reshape(h, w, 1)
            out = ref_effects.rotate(idx, deg)
            center, angle, scale = calls[-1]
            oh, ow = out.shape[:2]
            first = int(out[0, 0, 0]) if oh and ow else -1
            rows.append((h, w, deg, oh, ow, first // w if first >= 0 else -1, first % w if first >= 0 else -1,
                         center[0], center[1], angle, scale))
    out_path = os.path.join(mg.OUT_DIR, "rotate_crop.npz")
    np.savez_compressed(out_path, cases=np.array(rows, dtype=np.float64),
                        columns=np.array(["H", "W", "degrees", "out_h", "out_w", "row0", "col0", "cx", "cy", "angle", "scale"]))
    print(out_path, len(rows), "cases")


if __name__ == "__main__":
    main()
