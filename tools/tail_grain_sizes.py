"""Tail stage time at 100 MP for several grain sizes (the grain stencil grows with the grain: 9 x 9 at 6 um, 15 x 15 at 12 um at
341 px/mm).  Development aid: python tools/tail_grain_sizes.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

H, W = 8192, 12288
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
img = synthetic_frame_device(H, W)
D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for gs in (3, 6, 9, 12, 15):
    params = proc.prepare(neg, gs, 0.4, (W, H), seed=1, matrix=REC709_TO_XYZ, print_film=prt, halation=False, sharpness=False)
    ctx.stage_front(img, params, 1, dst=D)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.stage_tail(D, params, out_f32=out, y0=0, y1=H, H_global=H)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    st = ctx.stencil_stats(2)[0]
    print(f"grain {gs:2d} um: stencil {st['kh']} x {st['kw']} unrolled R {st['unrolled']} separable {st['separable']}: tail {np.median(ts):.3f} ms")
