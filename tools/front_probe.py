"""LUT-only fused kernel (BASELINE config 2, 24 MP): time vs workgroups per CU of its persistent grid, noise and smooth frames."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 4000, 6000
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation=False, sharpness=False, grain=0)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
def timeit(fn, iters=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for kind in ("noise", "smooth"):
    img = synthetic_frame_device(H, W, kind=kind)
    for bpc in (3, 6, 12):
        ctx.set_option("front_blocks_per_cu", bpc)
        print(f"{kind:6s} blocks/CU {bpc:2d}: %.3f ms" % timeit(lambda: ctx.render(img, params, out_f32=out)))
