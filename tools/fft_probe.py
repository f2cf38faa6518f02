"""Where the FFT form overtakes the direct form: square random stencils of n x n taps on a 24 MP plane set, both forms."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd.context import HipContext
H, W = 4000, 6000
ctx = HipContext(0)
src = torch.rand((3, H, W), dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
rng = np.random.default_rng(0)
ctx.set_option("stencil_fft_min_taps", 1)
for n in (17, 19, 21, 23, 25, 27):
    k = rng.uniform(0.0, 1.0, (n, n, 3)).astype(np.float32)
    k = (k + k[:, ::-1]) / 2  # left-right mirror symmetric like the MTF stencils (the direct form's fast path)
    k /= k.sum(axis=(0, 1), keepdims=True)
    ctx.set_kernel(1, k)
    t = {}
    for fft in (0, 1):
        ctx.set_option("stencil_fft", fft)
        t[fft] = timeit(lambda: ctx.stage_stencil(1, src, dst, y0=0, y1=H, H_global=H))
    ctx.set_option("stencil_fft", 0)
    ctx.set_option("stencil_fixed", 0)  # the generic entry list instead of the unrolled small-stencil form
    tg = timeit(lambda: ctx.stage_stencil(1, src, dst, y0=0, y1=H, H_global=H))
    ctx.set_option("stencil_fixed", 1)
    print(f"{n:2d} x {n:2d} ({n * n:4d} taps): direct {t[0]:.3f} ms (entry list {tg:.3f})   fft {t[1]:.3f} ms   {'FFT' if t[1] < t[0] else 'direct'}")
