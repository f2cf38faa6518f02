"""Run one stage a few times (target for rocprofv3).  usage: prof_stage.py {halation|mtf|tail|front|render} [W H] [variant]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "halation"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
H = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
variant = int(sys.argv[4]) if len(sys.argv) > 4 else -1
iters = int(os.environ.get("ITERS", "3"))
fw = 36.0 * W / 12288  # keep the 100 MP frame's px/mm (341.33) whatever the crop size
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
img = synthetic_frame_device(H, W, kind=os.environ.get("FRAME", "noise"))  # FRAME=smooth: photograph-like
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, print_film=prt, halation_green_factor=0.3, exp_kelvin=6000,
                      color_masking=1.0, matrix=REC709_TO_XYZ, frame_width=fw, frame_height=fw * H / W)
ctx.set_option("stencil_variant", variant)
ctx.set_option("xcd_remap", int(os.environ.get("XCD", "2")))
E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
D = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
D2 = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ctx.stage_front(img, params, 0, dst=E)
ctx.stage_front(img, params, 1, dst=D)
torch.cuda.synchronize()
for _ in range(iters):
    if stage == "halation":
        ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)
    elif stage == "mtf":
        ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
    elif stage == "tail":
        ctx.stage_tail(D, params, out_f32=out, y0=0, y1=H, H_global=H)
    elif stage == "front":
        ctx.stage_front(img, params, 1, dst=D)
    else:
        ctx.render(img, params, out_f32=out)
torch.cuda.synchronize()
print("done", stage, W, H)
