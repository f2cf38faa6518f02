"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the caller-side histogram (utils.py:91-223).

Pinned: tests/golden/histogram.npz holds inputs and outputs of the reference's own `generate_histogram` and
`precompute_mix_table` (explicit colours) run by CPython here (tools/make_golden_histogram.py).  `numba_semantics=False`
reproduces those vectors bit for bit; `True` is what the compiled (numba) reference computes: numba types `height` as int64
and promotes `float32_array * int64` to float64 (NumPy 2 keeps float32), which can move a bar by one pixel on a rare bin.
The default mix-table colours come from colour-science (absent here): only the explicit-colour form is pinned.
"""

import numpy as np


def precompute_mix_table(red, green, blue):
    """utils.py:102-138 with the three base colours given (8-bit sRGB triples, float)."""
    lin = [(np.asarray(c).astype(np.float32) / 255.0) ** 2.2 for c in (red, green, blue)]
    t = np.zeros((2, 2, 2, 4), dtype=np.uint8)
    for idx in np.ndindex(2, 2, 2):
        if not any(idx):
            continue
        mix = np.clip(sum(i * c for i, c in zip(idx, lin)), 0.0, 1.0)  # utils.py:121-124
        t[idx][0:3] = np.round(mix ** (1.0 / 2.2) * 255.0).astype(np.uint8)
        t[idx][3] = 255
    peak_rgb = (t[1, 1, 1, :3] / 255.0) ** 2.2  # utils.py:134-136
    t[1, 1, 1, :3] = peak_rgb.mean() ** (1.0 / 2.2) * 255.0
    return t


def counts(image):
    """utils.py:156-165: (3, 256) int32 counts of a uint8 (H, W, 3) image."""
    image = np.asarray(image)
    return np.stack([np.bincount(image[..., c].ravel(), minlength=256) for c in range(3)]).astype(np.int32)


def generate_histogram(image, mix_table, height=100, numba_semantics=True):
    """utils.py:145-223, loop for loop (256 bins and `height` rows only; the pixel loop is `counts`)."""
    hist = counts(image).astype(np.float32)  # :168-170
    max_val = max(hist[0].max(), hist[1].max(), hist[2].max())  # :172-174
    if max_val == 0:
        max_val = 1
    f = np.empty_like(hist)
    for c in range(3):
        for i in range(256):
            f[c, i] = np.log1p(hist[c, i] / max_val)  # :176-179
    smoothed = np.empty_like(f)
    for c in range(3):
        for i in range(256):
            left = i - 1 if i > 0 else i
            right = i + 1 if i < 255 else i
            smoothed[c, i] = (f[c, left] + f[c, i] + f[c, right]) / 3  # :186-191
    max_val = max(smoothed[0].max(), smoothed[1].max(), smoothed[2].max())  # :194-196
    if max_val == 0:
        max_val = 1
    if numba_semantics:
        final = ((smoothed.astype(np.float64) * height) / np.float64(max_val)).astype(np.int32)
    else:
        final = ((smoothed * height) / max_val).astype(np.int32)  # :198-200
    img = np.zeros((height, 256, 4), dtype=np.uint8)
    for x in range(256):
        lim = [height - final[c, x] for c in range(3)]
        for y in range(height):
            img[y, x] = mix_table[int(y >= lim[0]), int(y >= lim[1]), int(y >= lim[2])]  # :208-221
    return img
