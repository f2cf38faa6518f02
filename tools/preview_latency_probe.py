"""Latency of a preview re-render through the drop-in call: process(src, resolution=preview, cache=True) on a frame that is already on
the device, one film setting changed per call (what a slider does) -- wall clock per call, and the host-side fingerprint of the
source array alone.    python tools/preview_latency_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
for (H, W) in ((4000, 6000), (8192, 12288)):
    img = synthetic_frame_device(H, W, seed=3).cpu().numpy()
    proc = HipProcessor(device=0)
    kw = dict(print_film=prt, lens_correction=False, frame_width=36, frame_height=24, resolution=(1000, 1500), seed=1)
    proc.process(img, neg, 6, 0.4, **kw)
    proc.process(img, neg, 6, 0.4, exp_comp=0.01, **kw)
    ts = []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = proc.process(img, neg, 6, 0.4, exp_comp=0.02 + 0.01 * i, **kw)
        ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(5):
        proc._array_fingerprint(img)
    fp = (time.perf_counter() - t0) / 5 * 1e3
    ts2 = []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = proc.process(img, neg, 6, 0.4, exp_comp=0.5 + 0.01 * i, src_version=1, **kw)
        ts2.append((time.perf_counter() - t0) * 1e3)
    print(f"{W}x{H} source, preview {out.shape[1]}x{out.shape[0]}: re-render {np.median(ts):.2f} ms per call (min {min(ts):.2f}); the fingerprint "
          f"of the source alone {fp:.2f} ms; with src_version {np.median(ts2):.2f} ms (min {min(ts2):.2f})", flush=True)
    proc.close()

# which setting a slider changes matters: each rebuilds another table on the host before the render
H, W = 4000, 6000
img = synthetic_frame_device(H, W, seed=3).cpu().numpy()
proc = HipProcessor(device=0)
kw = dict(print_film=prt, lens_correction=False, frame_width=36, frame_height=24, resolution=(1000, 1500), seed=1)
proc.process(img, neg, 6, 0.4, **kw)
for name, make in (("exp_comp (input LUT)", lambda i: dict(exp_comp=0.01 * i)), ("red_light (print LUT)", lambda i: dict(red_light=0.01 * i)),
                   ("halation_size (halation stencil)", lambda i: dict(halation_size=1.0 + 0.01 * i)),
                   ("sharpening_strength (MTF stencil)", lambda i: dict(sharpening_strength=0.01 * i)),
                   ("grain_size (grain stencil)", lambda i: dict()), ("push_pull (density curve)", lambda i: dict(push_pull=0.01 * i)),
                   ("nothing (same settings again)", lambda i: dict())):
    ts = []
    for i in range(1, 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proc.process(img, neg, 6 + (0.05 * i if name.startswith("grain") else 0), 0.4, **kw, **make(i))
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"slider {name:36s}: {np.median(ts):7.2f} ms per step (min {min(ts):.2f})", flush=True)
proc.close()
