"""Batch export (BASELINE config 5: 24 MP frames through BatchSharder + the two-phase API) with payload frames in PINNED host memory
(what bench.py's pcie_inclusive leg hands over) against ordinary PAGEABLE NumPy arrays (what extract_image_data_cpu makes of a decoded
frame).    python tools/batch_pageable_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.sharding import BatchSharder  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

H, W = 4000, 6000
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
frame = synthetic_frame_device(H, W, seed=1234)
kw = dict(print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0, halation_green_factor=0.3,
          matrix=REC709_TO_XYZ, seed=1)
host3 = frame.cpu()
srcs = {
    "pinned fp32 RGB": host3.pin_memory(),
    "pageable fp32 RGB (NumPy)": host3.numpy().copy(),
    "pinned uint16": (frame.clamp(0, 1) * 65535).to(torch.int32).to(torch.int16).cpu().pin_memory(),
    "pageable uint16 (NumPy)": (frame.clamp(0, 1) * 65535).to(torch.int32).cpu().numpy().astype(np.uint16),
}
N = 16
for name, img in srcs.items():
    pay = {"image_array": img, "output_resolution": (W, H), "canvas_resolution": None, "pipeline_resolution": (W, H)}
    if "uint16" in name:
        pay["u16_factor"] = 1.0
    for mode, execute, collect in (
            ("serial    ", lambda t, pl: int(proc.process_preloaded(pl, neg, 6, 0.4, **kw)[0, 0, 0]), None),
            ("overlapped", lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw), lambda t, h: int(h.result()[0, 0, 0]))):
        BatchSharder(0, 1).run([0, 1], lambda t: pay, execute, collect=collect)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, _ = BatchSharder(0, 1).run(list(range(N)), lambda t: pay, execute, collect=collect)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:28s} {mode}: {dt / N * 1e3:7.2f} ms per 24 MP frame = {H * W / 1e6 * N / dt:7.0f} MP/s", flush=True)
proc.close()
