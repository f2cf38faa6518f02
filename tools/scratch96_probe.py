"""How far can the halation's 12-byte scratch element (stencil_fft_scratch96) be trusted?  Density error (halation + log + curve,
against the fp64 oracle) of complex128 and 12-byte scratch on frames with a dark field at level `lo` and bright pixels at level
`hi`, as a function of hi / lo.  Development aid: python tools/scratch96_probe.py"""
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
ctx.set_curve1d(curve)
ctx.set_kernel(0, k)
params = ctx.make_params(halation=True)
H, W = 600, 1100
rng = np.random.default_rng(3)
print(f"{'lo':>8s} {'hi':>9s} {'ratio':>9s} | complex128: max rel err (density, floor 1e-3) | 12-byte scratch | exposure-only 12-byte (floor = lo)")
for lo in (1e-4, 1e-3, 1e-2):
    for hi in (1.0, 16.0, 100.0, 1000.0, 16000.0, 65504.0):
        img = (lo * rng.uniform(1.0, 3.0, (H, W, 3))).astype(np.float32)
        img[::97, ::131] = hi  # isolated bright pixels, every window has some; most dark pixels are farther than the kernel's reach
        img[300:340, 500:560] = hi * rng.uniform(0.5, 1.0, (40, 60, 3))  # and a bright patch
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
        print(f"{lo:8.0e} {hi:9.0f} {hi / lo:9.1e} | {res[0]:10.2e} | {res[1]:10.2e} | {ee:10.2e}")
