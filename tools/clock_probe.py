"""Sample engine clock / power with rocm-smi while the halation stencil runs back to back (development aid)."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402

H, W = 8192, 12288
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
img = synthetic_frame_device(H, W)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, print_film=prt, matrix=REC709_TO_XYZ)
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "junction", "fclk"))]
        print(time.strftime("%H:%M:%S"), " | ".join(keep), flush=True)
        time.sleep(0.5)


mode = sys.argv[1] if len(sys.argv) > 1 else "render"
t = threading.Thread(target=sampler)
t.start()
time.sleep(1.5)
print("--- load starts:", mode, flush=True)
t0 = time.time()
n = 0
while time.time() - t0 < 8:
    for _ in range(10):
        proc.ctx.render(img, params, want_f32=True)
        n += 1
    torch.cuda.synchronize()
dt = time.time() - t0
print(f"--- load ends: {n} frames, {dt / n * 1e3:.2f} ms/frame", flush=True)
time.sleep(1.0)
stop = True
t.join()
