"""Frame time and per-pass device time of the FFT stencils vs window width (256 / 512 columns), 100 MP frame.

    python tools/fft_window_probe.py [path/to/lib.so]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import _lib
if len(sys.argv) > 1:  # a development build of the library (tools/ablate/*.so)
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
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
print(os.path.basename(_lib.LIB_PATH))
for window in ((256, 256), (256, 512), (512, 256), (512, 512), (0, 0)):
    ctx.set_option("stencil_fft_window_rows", window[0])
    ctx.set_option("stencil_fft_window", window[1])
    for streams in (2,):
        ctx.set_option("stencil_fft_streams", streams)
        t = timeit(lambda: ctx.render(img, params, out_f32=out))
        print(f"window {window} streams {streams}: render {t:.3f} ms", [c['window'] for c in ctx.stencil_stats(0)][:1],
              [c['window'] for c in ctx.stencil_stats(1)][:1])
    ctx.set_option("kernel_timing", 7)
    for cls in range(3): ctx.kernel_timing(cls)
    ctx.render(img, params, out_f32=out); torch.cuda.synchronize()
    print("   passes (1 stream, ms, launches, GB):", [(round(ms, 3), n, round(b / 1e9, 2)) for ms, n, b in (ctx.kernel_timing(c) for c in range(3))])
    ctx.set_option("kernel_timing", 0)
