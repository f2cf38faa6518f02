"""process(host array, cache=False, result_buffers=2) at cfg 4 against the number of row bands the frame streams through the pipeline in
(HipProcessor.stream_bands; 0 = upload, render, download one after the other; "16:2" = 16 bands with the last two halved,
HipProcessor.stream_taper).    python tools/stream_bands_probe.py [bands[:taper] ...]
R2F_PROBE_SOURCES="pinned f32,pageable f32,pinned u16,pageable u16": which source arrays (default: the pinned float frame)."""
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
SRC = {"pinned f32": host_np, "pageable f32": np.array(host_np)}
u16 = (np.clip(host_np, 0, 1) * 65535).astype(np.uint16)
pin16 = torch.from_numpy(u16.view(np.int16)).pin_memory().numpy().view(np.uint16)
SRC.update({"pinned u16": pin16, "pageable u16": u16})
which = os.environ.get("R2F_PROBE_SOURCES", "pinned f32").split(",")
for arg in [a for a in sys.argv[1:]] or ["0", "4", "8", "12", "16", "24", "32"]:
    bands, _, taper = arg.partition(":")  # "16:2" = 16 bands, the last two halved
    bands, proc.stream_taper = int(bands), int(taper or 0)
    proc.stream_bands = bands
    for name in which:
        src = SRC[name.strip()]
        extra = dict(exposure=0.0) if src.dtype == np.uint16 else {}
        ts = []
        for i in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            proc.process(src, neg, 6, 0.4, cache=False, seed=100 + i, **kw, **extra)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"stream_bands {bands:2d} taper {proc.stream_taper} {name.strip():13s}: best {min(ts[2:]):.2f} ms   (all: {' '.join(f'{t:.2f}' for t in ts)})", flush=True)
proc.close()
