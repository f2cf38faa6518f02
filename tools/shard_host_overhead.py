"""How long does the host take to issue one shard-sized step (1/8 of the 100 MP frame), and how long does the GPU take?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock, stencils
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer
from raw2film_amd.synthetic import synthetic_frame_device
from raw2film_amd.tracing import TimedBackend
W, H = 12288, 1024
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
fw = 36.0
params = proc.prepare(neg, 6, 0.4, (W, 8192), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
scale = 12288 / 36
hal_k = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3); mtf_k = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
be = HipStageBackend.for_stencils(proc.ctx, params, hal_k, mtf_k)
r = RowShardedRenderer(be, H, W, halation=True, mtf=True, grain=True, rank=0, world=1)
r.backend = TimedBackend(be)
frame = synthetic_frame_device(H, W); out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(3): r.render(frame, out_f32=out)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): r.render(frame, out_f32=out)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"1/8 shard ({W}x{H}): host issue {t_issue/N*1e3:.3f} ms/step, wall {t_all/N*1e3:.3f} ms/step")
