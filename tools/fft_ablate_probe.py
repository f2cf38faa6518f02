"""Per-pass device time of the FFT stencils (cfg 4) for one or more builds of the library, each in its own process:

    python tools/fft_ablate_probe.py lib_a.so [lib_b.so ...] [--opt name=value ...] [--iters 5]

For every library: event-bracketed launch times of the six pass classes summed per frame with ONE internal stream (every kernel
alone on the GPU: clean per-pass times), then halation / MTF stage times and the rendered frame with the product's two streams.
Development variants render wrong frames by construction; only the timings mean anything there.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, opts, iters):
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from raw2film_amd import _lib

    _lib.LIB_PATH = os.path.abspath(lib)
    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.synthetic import synthetic_frame_device

    H, W = 8192, 12288
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    proc = HipProcessor(device=0)
    ctx = proc.ctx
    for o in opts:
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    img = synthetic_frame_device(H, W)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, print_film=prt, matrix=REC709_TO_XYZ, halation_green_factor=0.3,
                          exp_kelvin=6000, color_masking=1.0)
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    D = torch.empty_like(E)
    D2 = torch.empty_like(E)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ctx.stage_front(img, params, 0, dst=E)

    def hal():
        ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)

    def mtf():
        ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)

    def med(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    hal(), mtf()
    torch.cuda.synchronize()
    # one internal stream, per-launch events
    ctx.set_option("stencil_fft_streams", 1)
    hal(), mtf()
    torch.cuda.synchronize()
    ctx.set_option("kernel_timing", 7)
    for c in range(6):
        ctx.kernel_timing(c)
    for _ in range(iters):
        hal(), mtf()
    torch.cuda.synchronize()
    t = [ctx.kernel_timing(c) for c in range(6)]
    ctx.set_option("kernel_timing", 0)
    alone = [x[0] / iters for x in t]
    h1, m1 = med(hal), med(mtf)
    ctx.set_option("stencil_fft_streams", 2)
    h2, m2 = med(hal), med(mtf)
    r = med(lambda: ctx.render(img, params, out_f32=out))
    name = os.path.basename(lib)
    print(f"{name:>22}: 1 stream c128 fwd {alone[0]:.3f} cols {alone[1]:.3f} inv {alone[2]:.3f} | c64 fwd {alone[3]:.3f} cols {alone[4]:.3f} "
          f"inv {alone[5]:.3f} | hal {h1:.3f} mtf {m1:.3f} || 2 streams hal {h2:.3f} mtf {m2:.3f} render {r:.3f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], [a for a in sys.argv[4:]], int(sys.argv[3]))
        sys.exit(0)
    args = sys.argv[1:]
    opts, libs, iters = [], [], 5
    i = 0
    while i < len(args):
        if args[i] == "--opt":
            opts.append(args[i + 1])
            i += 2
        elif args[i] == "--iters":
            iters = int(args[i + 1])
            i += 2
        else:
            libs.append(args[i])
            i += 1
    for lib in libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, str(iters)] + opts, check=False)
