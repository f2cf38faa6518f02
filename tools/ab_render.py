"""Interleaved A/B of the rendered frame under several option sets, in ONE process on one box (no box-to-box spread, drift shared):

    python tools/ab_render.py [--lib variant.so] [--config cfg4_100mp] [--rounds 6] [--iters 20] --set a=1,b=2 --set a=0 ...

Every round renders each option set `iters` times (after 3 untimed frames, so that the set's graph is captured and replayed) and
notes the median; the table at the end gives per set the median of its round medians, the spread, and the stage times of the
last round.  An empty --set "" is the library's defaults.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--config", default="cfg4_100mp")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--reset", default="", help="options (name=value,...) that restore the defaults before each set")
args = ap.parse_args()
from raw2film_amd import _lib  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import CONFIGS, synthetic_frame_device  # noqa: E402

W, H = CONFIGS[args.config]
effects = args.config != "cfg2_24mp"
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, halation_green_factor=0.3,
                      exp_kelvin=6000, color_masking=1.0, halation=effects, sharpness=effects, grain=2 if effects else 0)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")


def parse(sx):
    return [(kv.split("=")[0], int(kv.split("=")[1])) for kv in sx.split(",") if kv]


sets = [parse(s) for s in (args.set or [""])]
reset = parse(args.reset)


def apply(opts):
    for k, v in reset:
        ctx.set_option(k, v)
    for k, v in opts:
        ctx.set_option(k, v)


def frames(n):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ctx.render(img, params, out_f32=out)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return ts


meds = [[] for _ in sets]
for r in range(args.rounds):
    for i, opts in enumerate(sets):
        apply(opts)
        frames(3)
        meds[i].append(float(np.median(frames(args.iters))))
print(f"{os.path.basename(_lib.LIB_PATH)} {args.config}: {args.rounds} rounds x {args.iters} frames, ms (median of round medians; min..max of them)")
for i, opts in enumerate(sets):
    m = meds[i]
    name = ",".join(f"{k}={v}" for k, v in opts) or "(defaults)"
    print(f"  {name:<48} {np.median(m):.3f}   {min(m):.3f}..{max(m):.3f}   rounds: " + " ".join(f"{x:.3f}" for x in m), flush=True)
