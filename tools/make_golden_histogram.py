#!/usr/bin/env python3
"""tests/golden/histogram.npz: inputs and outputs of the reference's own utils.generate_histogram / precompute_mix_table
(utils.py:91-223), executed by CPython in the dev container with the stubs of tools/make_golden.py (numba.njit = identity).

    python3 -B tools/make_golden_histogram.py

The default mix-table colours need colour-science (absent), so the table is built with explicit base colours."""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REF_SRC)
    import raw2film.utils as ref_utils  # noqa: E402

    rng = np.random.default_rng(20261002)
    g = {}
    colours = [np.array(c, dtype=np.float64) for c in ((222.4, 61.2, 47.9), (13.1, 155.0, 11.6), (64.3, 118.2, 245.0))]
    table = ref_utils.precompute_mix_table(*colours)
    g["colours"] = np.stack(colours)
    g["mix_table"] = table
    cases = []
    # smooth photographic-ish frame, a flat frame (one bin per channel), an empty-ish frame with clipped highlights, tiny frame
    a = np.clip(rng.normal(110, 45, (48, 64, 3)) * np.array([1.0, 0.9, 1.15]), 0, 255).astype(np.uint8)
    b = np.full((16, 24, 3), (12, 200, 255), dtype=np.uint8)
    c = np.zeros((32, 40, 3), dtype=np.uint8)
    c[::3, ::5] = 255
    c[1::7, 2::3, 1] = 128
    d = rng.integers(0, 256, (3, 5, 3)).astype(np.uint8)
    for img, h in ((a, 80), (a, 100), (b, 80), (c, 37), (d, 80)):
        cases.append((img, h))
    g["n"] = np.array(len(cases))
    for i, (img, h) in enumerate(cases):
        g[f"image_{i}"] = img
        g[f"height_{i}"] = np.array(h)
        g[f"hist_{i}"] = ref_utils.generate_histogram(img, table, h)
    out = os.path.join(mg.OUT_DIR, "histogram.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
