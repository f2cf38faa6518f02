"""Sweep the band width of the XCD tile order (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = (int(sys.argv[2]), int(sys.argv[1])) if len(sys.argv) > 2 else (8192, 12288)
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
for mode, band in [(0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (2, 3), (2, 4), (2, 6), (2, 8), (2, 12), (2, 16)]:
    ctx.set_option("xcd_remap", mode); ctx.set_option("xcd_band", band)
    th = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
    tm = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
    print(f"xcd_remap {mode} band {band:2d}: halation {th:7.3f} ms   mtf {tm:7.3f} ms")
