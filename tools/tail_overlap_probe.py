"""Does the tail kernel gain from desynchronised workgroups?  The 100 MP tail as ONE launch vs as 2 / 4 row bands launched
on separate streams at the same time (workgroups of different bands then sit on the same CUs in different phases)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402

W, H = 12288, 8192
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, halation_green_factor=0.3,
                      exp_kelvin=6000, color_masking=1.0)
D = torch.rand((3, H, W), dtype=torch.float32, device="cuda") * 2.5 + 0.3
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(8)]


def run(nb):
    if nb == 1:
        ctx.stage_tail(D, params, out_f32=out, y0=0, y1=H, H_global=H)
        return
    main = torch.cuda.current_stream()
    ev = main.record_event()
    step = H // nb
    for b in range(nb):
        with torch.cuda.stream(streams[b]):
            streams[b].wait_event(ev)
            ctx.stage_tail(D, params, out_f32=out, y0=b * step, y1=(b + 1) * step, H_global=H)
            main.wait_event(streams[b].record_event())


for nb in (1, 2, 4, 8, 1):
    for _ in range(2):
        run(nb)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(nb)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"tail in {nb} concurrent band(s): {np.median(ts):.3f} ms")
