"""Interleaved A/B of two (or more) BUILDS of the library in ONE process on one box: the library exports nothing but its C ABI
(csrc/r2f_exports.map), so several builds bind side by side (HipProcessor(lib_path=...)) and take turns on the same frame --
no box-to-box spread, clock and thermal drift shared.

    python tools/ab_libs.py [--config cfg4_100mp] [--rounds 6] [--iters 20] [--opt name=value ...] lib_a.so lib_b.so ...

("default" names the in-tree library.)  Per round every build renders `iters` frames (after 3 untimed ones: capture + replay) and
notes the median of its event-timed frames; the table gives per build the median of the round medians and their range.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--config", default="cfg4_100mp")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--frame", default="noise")
ap.add_argument("--clamp", default="", help="lo,hi: clamp the synthetic frame (a range the 12-byte scratch element's guard accepts)")
args = ap.parse_args()
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import CONFIGS, synthetic_frame_device  # noqa: E402

W, H = CONFIGS[args.config]
effects = args.config != "cfg2_24mp"
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
img = synthetic_frame_device(H, W, kind=args.frame)
if args.clamp:
    lo, hi = (float(v) for v in args.clamp.split(","))
    img = img.clamp_(lo, hi)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
procs = []
for lib in args.libs:
    p = HipProcessor(device=0, lib_path=None if lib == "default" else lib)
    for o in args.opt:
        k, v = o.split("=")
        p.ctx.set_option(k, int(v))
    params = p.prepare(neg, 6, 0.4, (W, H), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, halation_green_factor=0.3,
                       exp_kelvin=6000, color_masking=1.0, halation=effects, sharpness=effects, grain=2 if effects else 0)
    procs.append((lib, p, params))
ref = None
med = {lib: [] for lib, _, _ in procs}
for r in range(args.rounds):
    for lib, p, params in (procs if r % 2 == 0 else procs[::-1]):
        for _ in range(3):
            p.ctx.render(img, params, out_f32=out)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
        ev[0].record()
        for i in range(args.iters):
            p.ctx.render(img, params, out_f32=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        med[lib].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters)])))
        if r == 0:
            cs = int(out.view(torch.int32).to(torch.int64).sum().item())
            same = "" if ref is None else ("  (frame bit-identical to the first build's)" if cs == ref else "  (frame DIFFERS from the first build's)")
            ref = cs if ref is None else ref
            print(f"# {lib}: {p.ctx._lib.r2f_version().decode()} checksum {cs}{same}", flush=True)
print(f"# tools/ab_libs.py --config {args.config} --rounds {args.rounds} --iters {args.iters} {' '.join('--opt ' + o for o in args.opt)}"
      + (f" --clamp {args.clamp}" if args.clamp else "") + ": frame ms, median of round medians (min..max)")
for lib, _, _ in procs:
    m = med[lib]
    print(f"  {lib:40s} {np.median(m):7.3f}   {min(m):.3f}..{max(m):.3f}   rounds: " + " ".join(f"{x:.3f}" for x in m))
for _, p, _ in procs:
    p.close()
