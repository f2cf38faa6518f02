"""Time the halation / MTF stencils with a development build of the library (tools/ablate/lib_expN.so, built with
-DR2F_EXP=N: bit 0 no LDS reads, bit 1 no weight loads, bit 2 no FMAs in the symmetric inner loop; or -DR2F_TAIL_EXP=N:
bit 0 no noise generation, bit 1 no grain stencil, bit 2 no 3-D LUT in the tail kernel).  Results are wrong by
construction; only the timings mean anything.

    python tools/ablate_stencil.py --build          # here (hipcc): builds tools/ablate/lib_exp{1,2,3,4}.so, lib_tail{1,2,4,7}.so
    python tools/ablate_stencil.py <path/to/lib.so>  # on the GPU box (the .so files travel with gpurun)
"""
import os
import sys

if "--build" in sys.argv:
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "tools", "ablate"), exist_ok=True)
    src = [os.path.join(root, "raw2film_amd", "csrc", f) for f in ("r2f_kernels.hip", "r2f_fft.hip", "r2f_api.hip")]
    for macro, values, stem in (("R2F_EXP", (1, 2, 3, 4), "lib_exp"), ("R2F_TAIL_EXP", (1, 2, 4, 7), "lib_tail")):
        for n in values:
            out = os.path.join(root, "tools", "ablate", f"{stem}{n}.so")
            subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", f"-D{macro}={n}", "-o", out] + src,
                           check=True)
            print(out)
    sys.exit(0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

H, W = 8192, 12288
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
ctx = proc.ctx
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3)
E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
D = torch.empty_like(E)
D2 = torch.empty_like(E)
ctx.stage_front(img, params, 1, dst=E)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


th = timeit(lambda: ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H))
tm = timeit(lambda: ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H))
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
tt = timeit(lambda: ctx.stage_tail(D2, params, out_f32=out, y0=0, y1=H, H_global=H))
print(f"{os.path.basename(_lib.LIB_PATH):>16}: halation {th:7.3f} ms   mtf {tm:7.3f} ms   tail {tt:7.3f} ms")
