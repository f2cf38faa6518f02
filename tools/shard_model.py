"""Device time of a 1/N row shard of the 100 MP frame on ONE GPU (what each of N GPUs would do between the exchanges),
eager launches vs HIP graph replay: the measured inputs of DESIGN.md section 5's modelled 1 -> 8 GPU curve.

    python tools/shard_model.py            # on the GPU box
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock, stencils  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

W, H = 12288, 8192
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
scale = W / 36.0
hal_k = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
mtf_k = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
ha, ma = stencils.vertical_reach(hal_k), stencils.vertical_reach(mtf_k)
print(f"halo rows: halation {ha}, MTF {ma}; exchange per boundary and direction: {(ha[0] + ma[0]) * W * 3 * 4 / 1e6:.2f} MB")
for n in (1, 2, 4, 8):
    rows = H // n
    # a shard in the middle of the frame carries halo rows on both sides: the renderer computes halation for rows + 2 * r_m
    ext = rows + (2 * (ha[0] + ma[0]) if n > 1 else 0)
    fh = 36.0 * rows / W  # keep px/mm: frame of `rows` rows at 341.33 px/mm
    params = proc.prepare(neg, 6, 0.4, (W, rows), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=fh,
                          exp_kelvin=6000, color_masking=1.0, halation_green_factor=0.3)
    be = HipStageBackend(proc.ctx, params, ha, ma)
    frame = synthetic_frame_device(rows, W, seed=n)
    out = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    res = {}
    for graph in (False, True):
        rr = RowShardedRenderer(be, rows, W, halation=True, mtf=True, rank=0, world=1, graph=graph)
        for _ in range(3):
            rr.render(frame, out_f32=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters):
            rr.render(frame, out_f32=out)
        t_host = (time.perf_counter() - t0) / iters * 1e3
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / iters * 1e3
        res[graph] = (t_all, t_host)
        del rr
    print(f"N = {n}: shard {W} x {rows}: eager {res[False][0]:.3f} ms/frame (host issue {res[False][1]:.3f}), "
          f"graph replay {res[True][0]:.3f} ms/frame (host issue {res[True][1]:.3f})  -> ideal {5.6 / n:.2f}")
