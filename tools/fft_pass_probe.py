"""Per-pass device time of the FFT stencils for a (development) build of the library: python tools/fft_pass_probe.py [lib.so]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raw2film_amd import HipProcessor, filmstock
from raw2film_amd.hip_processor import REC709_TO_XYZ
from raw2film_amd.synthetic import synthetic_frame_device
H, W = 8192, 12288
stocks = filmstock.builtin_stocks(); neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0); ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(2): ctx.render(img, params, out_f32=out)
torch.cuda.synchronize()
ctx.set_option("kernel_timing", 7)
for c in range(3): ctx.kernel_timing(c)
N = 5
for _ in range(N): ctx.render(img, params, out_f32=out)
t = [ctx.kernel_timing(c) for c in range(3)]
print(f"{os.path.basename(_lib.LIB_PATH):>16}: rows_fwd {t[0][0]/N:.3f}  cols {t[1][0]/N:.3f}  rows_inv {t[2][0]/N:.3f} ms per frame")
