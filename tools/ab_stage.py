"""Stage times of one frame under a set of context options (development aid for A/B runs on the GPU box).

    python tools/ab_stage.py [--config cfg4_100mp] [--opt name=value ...] [--lib path/to/variant.so] [--iters 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg4_100mp")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--lib", default=None)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--frame", default="noise")
ap.add_argument("--effects", action="store_true", help="all stages on, whatever the configuration (cfg2_24mp + --effects = one frame of cfg 5)")
args = ap.parse_args()
from raw2film_amd import _lib  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import CONFIGS, synthetic_frame_device  # noqa: E402

W, H = CONFIGS[args.config]
effects = args.effects or args.config != "cfg2_24mp"
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
for o in args.opt:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
img = synthetic_frame_device(H, W, kind=args.frame)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, halation_green_factor=0.3,
                      exp_kelvin=6000, color_masking=1.0, halation=effects, sharpness=effects, grain=2 if effects else 0)


def timeit(fn, iters=args.iters):
    for _ in range(2):
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
    return float(np.median(ts))


out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
line = [f"{os.path.basename(_lib.LIB_PATH)} {args.config} {' '.join(args.opt)}:"]
if effects:
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    D2 = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    line.append(f"front {timeit(lambda: ctx.stage_front(img, params, 0, dst=E)):.3f}")
    line.append(f"halation {timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)):.3f}")
    line.append(f"mtf {timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)):.3f}")
    line.append(f"tail {timeit(lambda: ctx.stage_tail(D2, params, out_f32=out, y0=0, y1=H, H_global=H)):.3f}")
t = timeit(lambda: ctx.render(img, params, out_f32=out))
line.append(f"render {t:.3f} ms -> {H * W / 1e6 / t * 1e3:.0f} MP/s")
torch.cuda.synchronize()
line.append(f"sum {out.double().sum().item():.6f} crc {int(out.view(torch.int32).long().sum().item()) & 0xffffffff:08x}")
print("  ".join(line), flush=True)
