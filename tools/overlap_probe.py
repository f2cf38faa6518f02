"""Can the VALU-bound tail hide behind the memory-bound MTF FFT passes?  Runs both stages of the 100 MP frame back to back on
one stream, then concurrently on two (the tail reads an older density buffer: timing only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import _lib
if os.environ.get('R2F_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['R2F_LIB'])
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 8192, 12288
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
E = torch.empty((3, H, W), dtype=torch.float32, device="cuda"); D = torch.empty_like(E); D2 = torch.empty_like(E); D3 = torch.empty_like(E)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ctx.stage_front(img, params, 0, dst=E)
ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)
ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
D3.copy_(D2)
torch.cuda.synchronize()
side = torch.cuda.Stream()
def seq():
    ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
    ctx.stage_tail(D3, params, out_f32=out, y0=0, y1=H, H_global=H)
def par(bands=1):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
    with torch.cuda.stream(side):
        for b in range(bands):
            ctx.stage_tail(D3, params, out_f32=out, y0=H * b // bands, y1=H * (b + 1) // bands, H_global=H)
    main.wait_stream(side)
def timeit(fn, iters=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
print("mtf alone          %.3f ms" % timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)))
print("tail alone         %.3f ms" % timeit(lambda: ctx.stage_tail(D3, params, out_f32=out, y0=0, y1=H, H_global=H)))
print("mtf then tail      %.3f ms" % timeit(seq))
for bands in (1, 4, 8):
    print("mtf || tail (%d bands) %.3f ms" % (bands, timeit(lambda: par(bands))))
def par_hal():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)
    with torch.cuda.stream(side):
        ctx.stage_tail(D3, params, out_f32=out, y0=0, y1=H, H_global=H)
    main.wait_stream(side)
print("halation alone     %.3f ms" % timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)))
print("halation || tail   %.3f ms" % timeit(par_hal))
