"""Frame time vs FFT batch size (window pairs per launch triple): does the scratch stay in the 256 MB Infinity Cache?"""
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
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for batch in (64, 96, 128, 160, 192, 224, 256, 320, 384, 512):
    ctx.set_option("stencil_fft_batch", batch)
    print(f"batch {batch:5d} ({batch} MB of scratch): render {timeit(lambda: ctx.render(img, params, out_f32=out)):.3f} ms")
