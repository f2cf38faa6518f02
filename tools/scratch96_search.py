"""Search for the worst error coefficient of the halation's 12-byte FFT scratch element: random frames of the
families in tests/hostile.py (dark holes in a bright field, blocks, stripes, checkers, gradients, half-bright frames, isolated
speculars) x bright-region statistics x max / min ratios of 2e5 .. 3e6 (the coefficient does not depend on the ratio -- the error is
linear in hi -- and below 1e5 the fp32 rounding of the two results, 1.2e-7, hides it), 256 x 512 windows forced like cfg 4's.  Per frame:
    coefficient = max |E_12byte - E_complex128| / max(|E_complex128|, lo) / (hi / lo)
over the FFT channels of the halation stencil -- the constant the guard of r2f_render is built on (r2f_api.hip dyn_rule).
    python tools/scratch96_search.py [--budget 400] [--seed 1] [--shape 600x1100]
Prints the worst frames and the per-family maxima; tests/test_gpu_fft.py runs the same function on a fixed, smaller budget."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(ctx, torch, img, channels=(0, 1)):
    """(coefficient, exposure rel err at the floor lo, hi / lo) of one frame; the halation stencil and the forced window are set."""
    H, W = img.shape[:2]
    t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
    res = []
    for s96 in (0, 1):
        ctx.set_option("stencil_fft_scratch96", s96)
        E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_stencil(0, t, E, y0=0, y1=H, H_global=H)
        res.append(E.cpu().numpy().astype(np.float64))
    ctx.set_option("stencil_fft_scratch96", 0)
    lo, hi = float(img[..., list(channels)].min()), float(np.abs(img[..., list(channels)]).max())
    # (both results are fp32 roundings of fp64 values, so the difference carries up to one fp32 ulp -- 1.2e-7 relative -- that is not
    # the element's doing: the search therefore runs at ratios where the element's own error is 16 .. 160 ulps; the coefficient
    # does not depend on the ratio, the error is linear in hi)
    e = 0.0
    for c in channels:
        e = max(e, float(np.max(np.abs(res[1][c] - res[0][c]) / np.maximum(np.abs(res[0][c]), lo))))
    return e / (hi / lo), e, hi / lo


def search(ctx, torch, budget, seed, H=600, W=1100, log=None, ratios=(2e5, 3e6)):
    import hostile

    rng = np.random.default_rng(seed)
    rows = []
    for i in range(budget):
        kind = hostile.SCRATCH96_KINDS[i % len(hostile.SCRATCH96_KINDS)]
        fill = hostile.SCRATCH96_FILLS[int(rng.integers(0, len(hostile.SCRATCH96_FILLS)))]
        ratio = float(np.exp(rng.uniform(np.log(ratios[0]), np.log(ratios[1]))))
        lo = float(10.0 ** rng.uniform(-4, -2))
        img = hostile.scratch96_frame(rng, H, W, kind, fill, lo, lo * ratio)
        coef, err, r = measure(ctx, torch, img)
        rows.append((coef, kind, fill, lo, r, err))
        if log:
            log(f"{i:4d} {kind:10s} {fill:9s} lo {lo:8.2e} ratio {r:8.2e}  E err {err:8.2e}  coefficient {coef:8.2e}")
    return rows


def setup(ctx):
    from helpers import stocks
    from oracle import kernels as ok

    neg, _, _ = stocks()
    ctx.set_curve1d(neg.get_density_curve(0.0, 1.0))
    ctx.set_kernel(0, ok.compute_halation_kernel(341.33, halation_green_factor=0.3))  # the 100 MP pitch: 87 x 87
    ctx.set_option("stencil_fft_window_rows", 256)
    ctx.set_option("stencil_fft_window", 512)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--shape", default="600x1100")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    import torch

    from raw2film_amd.context import HipContext

    H, W = (int(v) for v in args.shape.split("x"))
    ctx = HipContext(0)
    setup(ctx)
    rows = search(ctx, torch, args.budget, args.seed, H, W, log=print if args.verbose else None)
    rows.sort(reverse=True)
    print(f"# tools/scratch96_search.py --budget {args.budget} --seed {args.seed} --shape {args.shape}: 87 x 87 halation stencil, 256 x 512 windows forced")
    print("# worst 12 frames: coefficient = exposure error of the 12-byte element against complex128, relative at the floor lo, / (hi / lo)")
    for coef, kind, fill, lo, r, err in rows[:12]:
        print(f"  {kind:10s} {fill:9s} lo {lo:8.2e} ratio {r:8.2e}  E err {err:8.2e}  coefficient {coef:8.2e}")
    print("# maximum per family / per bright-region statistics")
    for key, idx in (("family", 1), ("fill", 2)):
        names = sorted({r[idx] for r in rows})
        print("  " + key + ": " + "  ".join(f"{n} {max(r[0] for r in rows if r[idx] == n):.2e}" for n in names))
    print(f"# worst coefficient over {len(rows)} frames: {rows[0][0]:.3e}")
    ctx.close()
