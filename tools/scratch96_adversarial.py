"""The 12-byte halation scratch element on patterns with far more energy per window than isolated speculars (a bright block, a
half-bright frame, bright stripes next to a dark field): exposure error / (hi / lo) -- the coefficient the guard of r2f_render is built
on (profiles/r05_scratch96_probe.txt).  Development aid: python tools/scratch96_adversarial.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import stocks  # noqa: E402
from oracle import kernels as ok  # noqa: E402
from oracle import stages as st  # noqa: E402
from raw2film_amd.context import HipContext  # noqa: E402

ctx = HipContext(0)
neg, prt, _ = stocks()
curve = neg.get_density_curve(0.0, 1.0)
k = ok.compute_halation_kernel(341.33, halation_green_factor=0.3)
ctx.set_curve1d(curve); ctx.set_kernel(0, k)
ctx.set_option("stencil_fft_window_rows", 256); ctx.set_option("stencil_fft_window", 512)
params = ctx.make_params(halation=True)
H, W = 600, 1100
rng = np.random.default_rng(3)
print("pattern            lo      hi    ratio | c128 D err | 12-byte D err | 12-byte E rel err (floor lo)")
def run(img, name, lo, hi):
    expo = st.halation(img, k)
    ref = st.multi_channel_interp(st.log_clip(expo), curve)
    t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
    res = []
    for s96 in (0, 1):
        ctx.set_option("stencil_fft_scratch96", s96)
        D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_halation(t, D, params, y0=0, y1=H, H_global=H)
        d = D.cpu().numpy().transpose(1, 2, 0)
        res.append(float(np.max(np.abs(d - ref)[..., :2] / np.maximum(np.abs(ref[..., :2]), 1e-3))))
    ctx.set_option("stencil_fft_scratch96", 1)
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_stencil(0, t, E, y0=0, y1=H, H_global=H)
    e = E.cpu().numpy().transpose(1, 2, 0)
    ee = float(np.max(np.abs(e - expo)[..., :2] / np.maximum(np.abs(expo[..., :2]), lo)))
    ctx.set_option("stencil_fft_scratch96", 0)
    print(f"{name:16s} {lo:7.0e} {hi:7.0f} {hi/lo:8.1e} | {res[0]:9.2e} | {res[1]:9.2e} | {ee:9.2e}   coefficient {ee/(hi/lo):.2e}")
for lo, hi in ((1e-3, 100.0), (1e-2, 1000.0), (1e-2, 4000.0), (1e-3, 400.0)):
    base = (lo * rng.uniform(1.0, 3.0, (H, W, 3))).astype(np.float32)
    a = base.copy(); a[::97, ::131] = hi; a[300:340, 500:560] = hi * rng.uniform(0.5, 1.0, (40, 60, 3)); run(a, "isolated+patch", lo, hi)
    b = base.copy(); b[:, :200] = hi * rng.uniform(0.5, 1.0, (H, 200, 3)); run(b, "left 200 cols", lo, hi)
    c = base.copy(); c[:, ::2] = 0; c[:, :W//2] = hi * rng.uniform(0.5, 1.0, (H, W//2, 3)); run(c, "half frame", lo, hi)
    d = base.copy(); d[::2] = hi * rng.uniform(0.5, 1.0, d[::2].shape); d[:, 600:] = base[:, 600:]; run(d, "stripes+dark", lo, hi)
