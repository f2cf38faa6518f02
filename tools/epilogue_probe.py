"""What does the log + density-curve epilogue of the halation's FFT pass 3 cost at 100 MP?  r2f_stage_halation (epilogue) against
r2f_stage_stencil on the same stencil and planes (no epilogue), device time per call, one and two internal streams.

    python tools/epilogue_probe.py          # on the GPU box
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock, stencils  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402

if "--lib" in sys.argv:  # a development build of the library (tools/_var/...), e.g. with -DR2F_FFT_EPI_ABLATE=n
    from raw2film_amd import _lib

    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
W, H = 12288, 8192
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000,
                      color_masking=1.0, halation_green_factor=0.3)
E = torch.rand((3, H, W), device="cuda") * 2.0 + 0.01
D = torch.empty_like(E)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


for streams in (2, 1):
    ctx.set_option("stencil_fft_streams", streams)
    a = timed(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
    b = timed(lambda: ctx.stage_stencil(0, E, D, y0=0, y1=H, H_global=H))
    print(f"{streams} internal stream(s): halation with log + curve epilogue {a:.3f} ms, the same stencil without {b:.3f} ms  -> epilogue {a - b:+.3f} ms")
    for mask, name in ((1, "rows fwd"), (2, "cols"), (4, "rows inv")):
        for label, fn in (("epilogue", lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)),
                          ("plain", lambda: ctx.stage_stencil(0, E, D, y0=0, y1=H, H_global=H))):
            ctx.set_option("kernel_timing", mask)
            for c in range(6):
                ctx.kernel_timing(c)
            fn()
            torch.cuda.synchronize()
            ms = sum(ctx.kernel_timing(c)[0] for c in range(6))
            ctx.set_option("kernel_timing", 0)
            print(f"    {name:9s} {label:9s} summed kernel time {ms:.3f} ms")
