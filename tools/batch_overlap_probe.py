"""Batch export (config 5): 24 MP frames one after the other on one context, against two contexts on two streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 4000, 6000
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
procs = [HipProcessor(device=0) for _ in range(2)]
img = synthetic_frame_device(H, W)
params = [p.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3) for p in procs]
outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
N = 64
def one():
    for i in range(N):
        procs[0].ctx.render(img, params[0], out_f32=outs[0])
def two():
    for i in range(N):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            procs[k].ctx.render(img, params[k], out_f32=outs[k])
for name, fn in (("one context", one), ("two contexts / streams", two), ("one context", one), ("two contexts / streams", two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:24s}: {dt * 1e3:.1f} ms per {N} frames = {dt / N * 1e3:.3f} ms per frame")
