"""Wall-clock latency of HipProcessor.process() for an interactive re-render: the first call prepares and uploads the frame,
the following ones (other film settings, same load parameters) read it on the device (GpuProcessor.load_image_texture's
convention).  24 MP float frame; full pipeline at full size, and the GUI's simplified preview (LUTs only) at 1620 x 1080."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2film_amd import HipProcessor, filmstock  # noqa: E402

stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
rng = np.random.default_rng(0)
img = (0.18 * 2.0 ** rng.normal(0.0, 1.5, (4000, 6000, 1)) * rng.uniform(0.6, 1.4, (4000, 6000, 3))).astype(np.float32)
proc = HipProcessor(device=0, result_buffers=int(os.environ.get("RESULT_BUFFERS", "0")))


def timed(**kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = proc.process(img, neg, 6, 0.4, print_film=prt, lens_correction=False, seed=1, **kw)
    return (time.perf_counter() - t0) * 1e3, out.shape


for name, kw in (("full pipeline, 6000 x 4000", {}),
                 ("simplified preview, 1620 x 1080", dict(resolution=(1080, 1620), halation=False, sharpness=False, grain=0))):
    proc.process(img[:64, :96], neg, 6, 0.4, print_film=prt, lens_correction=False, **kw)  # warm the tables
    first = timed(exp_comp=0.0, **kw)
    again = [timed(exp_comp=0.1 * i, **kw)[0] for i in range(1, 6)]
    fresh = timed(exp_comp=0.0, cache=False, **kw)
    print(f"{name}: first call {first[0]:.1f} ms -> {first[1]}, re-render {np.median(again):.1f} ms, cache=False {fresh[0]:.1f} ms", flush=True)
