"""Occupancy experiment: halation with the 512-thread tile (4 waves/SIMD) vs the 256-thread tile (2 waves/SIMD)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 8192, 12288
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
E = torch.empty((3, H, W), dtype=torch.float32, device="cuda"); D = torch.empty_like(E); D2 = torch.empty_like(E)
ctx.stage_front(img, params, 0, dst=E)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for v in (0, 2):
    ctx.set_option("stencil_variant", v)
    for kb in (80, 53, 40):
        ctx.set_option("stencil_lds_kb", kb)
        row = []
        for ab in (0, 1, 2):
            ctx.set_option("stencil_ablate", ab)
            try:
                row.append(timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)))
            except Exception as ex:
                row.append(float("nan"))
        ctx.set_option("stencil_ablate", 0)
        tm = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
        print(f"variant {v} lds {kb} KB: halation {row[0]:7.3f}  no-fill {row[1]:7.3f}  no-accumulate {row[2]:7.3f}   mtf {tm:7.3f}")
