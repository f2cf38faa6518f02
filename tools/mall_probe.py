"""Do the FFT passes run faster when the PLANES they read and write are Infinity-Cache resident too (not only the scratch)?
The halation stencil (no epilogue) on a 12288 x H frame, H = 8192 (planes 2 x 403 MB: streamed from HBM) against H = 1024 and 512
(planes 2 x 50 / 2 x 25 MB: resident between repeated calls together with a small scratch batch), per-pass kernel time per
megapixel, one internal stream.  Input of the band-pipelining question.

    python tools/mall_probe.py          # on the GPU box
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402

W = 12288
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
proc.prepare(neg, 6, 0.4, (W, 8192), seed=1, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000,
             color_masking=1.0, halation_green_factor=0.3)
ctx.set_option("stencil_fft_streams", 1)
for batch in (32, 192):
    ctx.set_option("stencil_fft_batch", batch)
    for H in (8192, 2048, 1024, 512):
        E = torch.rand((3, H, W), device="cuda") * 2.0 + 0.01
        D = torch.empty_like(E)
        for _ in range(3):
            ctx.stage_stencil(0, E, D, y0=0, y1=H, H_global=H)
        res = []
        for mask in (1, 2, 4):
            ctx.set_option("kernel_timing", mask)
            for c in range(6):
                ctx.kernel_timing(c)
            for _ in range(4):
                ctx.stage_stencil(0, E, D, y0=0, y1=H, H_global=H)
            torch.cuda.synchronize()
            res.append(sum(ctx.kernel_timing(c)[0] for c in range(6)) / 4)
            ctx.set_option("kernel_timing", 0)
        mp = H * W / 1e6
        print(f"batch {batch:3d} MiB, {W} x {H:4d} (planes {2 * H * W * 4 / 1e6:5.0f} MB in + out each): rows fwd {res[0] / mp * 100:.3f}  cols {res[1] / mp * 100:.3f}  "
              f"rows inv {res[2] / mp * 100:.3f}  ms per 100 MP   (window rows {-(-H // 172)} for {H / 172:.2f})")
        del E, D
