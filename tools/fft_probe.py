import sys
import torch
sys.path.insert(0, "/root/repo")
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 8192, 12288
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
E = torch.empty((3, H, W), dtype=torch.float32, device="cuda"); D = torch.empty_like(E); D2 = torch.empty_like(E)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ctx.stage_front(img, params, 0, dst=E)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for mt in (2000, 1000):
    ctx.set_option("stencil_fft_min_taps", mt)
    for batch in (256, 512, 1024, 2048):
        ctx.set_option("stencil_fft_batch", batch)
        th = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
        tm = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
        tr = timeit(lambda: ctx.render(img, params, out_f32=out))
        print(f"min_taps {mt} batch {batch}: halation {th:.3f}  mtf {tm:.3f}  render {tr:.3f}")
