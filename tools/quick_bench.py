"""Per-stage timing of the pipeline on one GPU (development aid; bench.py is the contract)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=12288)
ap.add_argument("--h", type=int, default=8192)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--variants", type=str, default="0")
ap.add_argument("--lds-kb", dest="lds_kb", type=str, default="160,80,53,40")
args = ap.parse_args()

H, W = args.h, args.w
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
img = synthetic_frame_device(H, W)
settings = dict(print_film=prt, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
settings["matrix"] = REC709_TO_XYZ
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, **settings)
scale = max(H, W) / 36
print(f"frame {W}x{H} = {H*W/1e6:.1f} MP, scale {scale:.2f} px/mm")


def timeit(fn, iters=args.iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))


E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
D2 = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
mp = H * W / 1e6
t = timeit(lambda: ctx.stage_front(img, params, 0, dst=E))
print(f"front->E        {t[0]:8.3f} ms  ({24*H*W/t[0]/1e6:.0f} GB/s)")
t = timeit(lambda: ctx.stage_front(img, params, 1, dst=D))
print(f"front->D        {t[0]:8.3f} ms  ({24*H*W/t[0]/1e6:.0f} GB/s)")
p0 = ctx.make_params(matrix=True)
t = timeit(lambda: ctx.stage_front(img, p0, 2, out_f32=out))
print(f"front->out      {t[0]:8.3f} ms  ({24*H*W/t[0]/1e6:.0f} GB/s)")
ctx.set_option("xcd_remap", 0)
for v in [int(x) for x in args.variants.split(",")]:
    ctx.set_option("stencil_variant", v)
    for kb in [int(x) for x in args.lds_kb.split(",")]:
        ctx.set_option("stencil_lds_kb", kb)
        try:
            t = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
        except Exception as ex:
            t = (float("nan"),)
        try:
            t2 = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
        except Exception as ex:
            t2 = (float("nan"),)
        print(f"v{v} lds {kb:3d} KB: halation {t[0]:8.3f} ms   mtf {t2[0]:8.3f} ms")
ctx.set_option("stencil_lds_kb", 80)
for v in [int(x) for x in args.variants.split(",")]:
    ctx.set_option("stencil_variant", v)
    for ab in (1, 2, 3):
        ctx.set_option("stencil_ablate", ab)
        t = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
        print(f"halation v{v} ablate={ab} ({['', 'no fill', 'no accumulate', 'weights always entry 0'][ab]}) {t[0]:8.3f} ms")
ctx.set_option("stencil_ablate", 0)
ctx.set_option("stencil_variant", -1)
ctx.set_option("xcd_remap", 2)
t = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
print(f"halation auto, xcd remap 2 (default) {t[0]:8.3f} ms")
ctx.set_option("stencil_sym", 0)
t = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
t2 = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
print(f"symmetric path OFF: halation {t[0]:8.3f} ms   mtf {t2[0]:8.3f} ms")
ctx.set_option("stencil_sym", 1)
t = timeit(lambda: ctx.stage_tail(D2, params, out_f32=out, y0=0, y1=H, H_global=H))
print(f"tail(grain)     {t[0]:8.3f} ms")
pn = ctx.make_params(matrix=True, halation=True, mtf=True)
t = timeit(lambda: ctx.stage_tail(D2, pn, out_f32=out, y0=0, y1=H, H_global=H))
print(f"tail(lut3d)     {t[0]:8.3f} ms  ({24*H*W/t[0]/1e6:.0f} GB/s)")
t = timeit(lambda: ctx.render(img, params, out_f32=out))
print(f"render full     {t[0]:8.3f} ms  -> {mp/t[0]*1e3:.0f} MP/s")
