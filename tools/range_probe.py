"""Device + issue time of the exposure-range record's pieces at a 1/8 row shard's sizes (development aid, round 6):
    python tools/range_probe.py
"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from raw2film_amd import HipProcessor, filmstock, stencils
from raw2film_amd.hip_processor import REC709_TO_XYZ
W, H = 12288, 8192
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0, halation_green_factor=0.3)
E = torch.rand((3, 1200, W), device="cuda") + 0.01
frame = torch.rand((1200, W, 3), device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
ctx.write_frame_params(params)
print("range kernel 59+59 rows: %.1f us" % t(lambda: ctx.stage_exposure_range(E, src_gy0=4000, y0=4000, y1=4059, y2=5100, y3=5159)))
print("range kernel 1024 rows: %.1f us" % t(lambda: ctx.stage_exposure_range(E, src_gy0=4000, y0=4000, y1=5024)))
for track in (False, True):
    print("front 59 rows track=%s: %.1f us" % (track, t(lambda: ctx.stage_front(frame[:59], params, 0, in_gy0=4000, dst=E, dst_gy0=4000, y0=4000, y1=4059, H_global=H, track_range=track))))
    print("front 906 rows track=%s: %.1f us" % (track, t(lambda: ctx.stage_front(frame[:906], params, 0, in_gy0=4059, dst=E, dst_gy0=4000, y0=4059, y1=4965, H_global=H, track_range=track))))
# with a reset between (as a frame does)
def frame_like(track):
    ctx.write_frame_params(params)
    ctx.stage_front(frame[:59], params, 0, in_gy0=4000, dst=E, dst_gy0=4000, y0=4000, y1=4059, H_global=H, track_range=track)
def reset_range():
    ctx.write_frame_params(params)
    ctx.stage_exposure_range(E, src_gy0=4000, y0=4000, y1=4059, y2=5100, y3=5159)
print("reset alone: %.1f us" % t(lambda: ctx.write_frame_params(params)))
print("reset + range kernel 59+59 rows: %.1f us" % t(reset_range))
print("reset + front 59 rows track=False: %.1f us" % t(lambda: frame_like(False)))
print("reset + front 59 rows track=True: %.1f us" % t(lambda: frame_like(True)))
