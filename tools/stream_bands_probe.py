"""process(host array, cache=False, result_buffers=2) at cfg 4 against the number of row bands the frame streams through the pipeline in
(HipProcessor.stream_bands; 0 = upload, render, download one after the other; "16:2" = 16 bands with the last two halved,
HipProcessor.stream_taper).    python tools/stream_bands_probe.py [bands[:taper] ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

H, W = 8192, 12288
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0, result_buffers=2)
host = torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True)
host.copy_(synthetic_frame_device(H, W, seed=1234))
host_np = host.numpy()
kw = dict(print_film=prt, lens_correction=False, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0,
          halation_green_factor=0.3, matrix=REC709_TO_XYZ)
for arg in sys.argv[1:] or ["0", "4", "8", "12", "16", "24", "32"]:
    bands, _, taper = arg.partition(":")  # "16:2" = 16 bands, the last two halved
    bands, proc.stream_taper = int(bands), int(taper or 0)
    proc.stream_bands = bands
    ts = []
    for i in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proc.process(host_np, neg, 6, 0.4, cache=False, seed=100 + i, **kw)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"stream_bands {bands:2d} taper {proc.stream_taper}: best {min(ts[2:]):.2f} ms   (all: {' '.join(f'{t:.2f}' for t in ts)})", flush=True)
proc.close()
