"""The per-window-pair choice of the halation's FFT scratch element against a HOST MODEL of what each pair's windows hold.

The guard (r2f_api.hip dyn_rule, r2f_fft.hip fft_decide_kernel): a pair may take the 12-byte element when
    max |x| <= bound * max(min x, floor)
over the exposure samples whose rounding can reach an output the pair keeps.  The device decides from a grid of 64 x 256-pixel tiles
(a superset of those samples); this tool reads the flags back (r2f_frame_scratch_flags) and checks the one thing that must hold whatever
the tiles are: a pair that took the element satisfies the rule on ITS OWN samples, gathered here straight from the exposure planes --
    rows: the rows within the stencil's reach of an output the window keeps (both roundings are row-local), reflected (101) into the
          frame and clamped to the rows the call's source buffer holds, exactly like pass 1 reads them;
    columns: all nx columns of the window, reflected into the frame (along a row the transforms mix everything the window holds);
    both windows of the pair (they share one complex image), the FFT channels of the stencil.
Random frames (tests/hostile.py families, tile-aligned bright regions, NaN sprinkles), random frame sizes (so that edges and window
pairings fall everywhere), whole-frame calls and row-shard calls with a source buffer of their own, the record filled by the range
kernel or by the front kernel, and whole frames through r2f_render itself (eager and from its captured graph).  It also reports how many pairs the exact per-pair range WOULD allow (what the tiles' granularity
costs).  Numerics play no part: this is the soundness of the decision, which no parity soak sees (the right-edge hole of round 6 -- a
window that reflects in columns its tiles did not cover -- passed 2 700 fuzz cases; this tool finds it in 5 of 1 500 random calls:
profiles/r06_scratch_choice_model.txt).
    python tools/scratch_choice_model.py [--budget 300] [--seed 1] [--verbose] [--lib tools/_var/lib_owncols.so]
(--lib: another build of the library, e.g. `python tools/build_variant.py owncols=-DR2F_DECIDE_OWN_COLUMNS_ONLY=1`, the decide kernel
as it was before the reflected columns counted: the tool must FAIL on it.)
tests/test_gpu_fft.py runs the same function on a fixed, smaller budget."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NY, NX = 256, 512  # the windows the choice exists for (cfg 4's), forced below


def reflect101(i, n):
    i = np.asarray(i, dtype=np.int64)
    if n == 1:
        return np.zeros_like(i)
    period = 2 * n - 2
    i = np.mod(i, period)
    return np.where(i < n, i, period - i)


def geometry(kernel, fft_channels):
    """Tap box of the FFT channels and its anchor (cv.filter2D's default anchor at (kh // 2, kw // 2))."""
    k = np.asarray(kernel)
    nz = np.nonzero(np.any(k[..., list(fft_channels)] != 0, axis=2))
    b0, b1, b2, b3 = int(nz[0].min()), int(nz[0].max()), int(nz[1].min()), int(nz[1].max())
    bh, bw = b1 - b0 + 1, b3 - b2 + 1
    return bh, bw, k.shape[0] // 2 - b0, k.shape[1] // 2 - b2


def model(E, geo, y0, y1, H, W, buf0, buf1, bound, floor_, fft_channels, NY=NY, NX=NX):
    """Per pair: (allowed by its own samples, lo, hi).  E: (3, H, W) float32 host copy of the WHOLE frame's exposure planes; the call
    covers rows [y0, y1) from a source buffer that holds rows [buf0, buf1).  NY x NX: the call's window shape."""
    bh, bw, ay, ax = geo
    vy, vx = NY - bh + 1, (NX - bw + 1) & ~3
    gx = (W + vx - 1) // vx
    ntiles = gx * ((y1 - y0 + vy - 1) // vy)
    ppc = (ntiles + 1) // 2
    out = []
    bound32, floor32 = np.float32(bound), np.float32(floor_)
    for pc in range(ppc):
        lo, hi, nan = np.float32(np.inf), np.float32(0.0), False
        for t in (2 * pc, 2 * pc + 1):
            if t >= ntiles:
                continue
            ty, tx = y0 + (t // gx) * vy, (t % gx) * vx
            kept_hi = min(ty + vy, y1)  # the window keeps outputs [ty, kept_hi) x [tx, tx + vx) (cropped to the frame)
            rows = np.arange(ty - ay, kept_hi - 1 + (bh - 1 - ay) + 1)
            rows = np.unique(np.clip(reflect101(rows, H), buf0, buf1 - 1))
            cols = np.unique(reflect101(np.arange(tx - ax, tx - ax + NX), W))
            S = E[np.ix_(list(fft_channels), rows, cols)]
            nan = nan or bool(np.isnan(S).any())
            with np.errstate(invalid="ignore"):
                lo = min(lo, np.float32(np.nanmin(S))) if not np.all(np.isnan(S)) else lo
                hi = max(hi, np.float32(np.nanmax(np.abs(S)))) if not np.all(np.isnan(S)) else hi
        if nan:  # pass 1 takes a NaN as 0: the outputs around it fall below the samples that are left
            lo = np.float32(-np.inf)
        allowed = bool(np.isfinite(hi) and hi <= bound32 * max(lo, floor32))
        out.append((allowed, float(lo), float(hi)))
    return out


def random_frame(rng, H, W, hostile):
    """An exposure-like (H, W, 3) float32 frame; bright regions partly aligned to the record's 64 x 256 tiles."""
    lo = float(10.0 ** rng.uniform(-4.5, -1.5))
    ratio = float(np.exp(rng.uniform(np.log(1e2), np.log(1e7))))
    kind = int(rng.integers(0, len(hostile.SCRATCH96_KINDS) + 3))
    if kind < len(hostile.SCRATCH96_KINDS):
        fill = hostile.SCRATCH96_FILLS[int(rng.integers(0, len(hostile.SCRATCH96_FILLS)))]
        img = hostile.scratch96_frame(rng, H, W, hostile.SCRATCH96_KINDS[kind], fill, lo, lo * ratio)
    else:  # shadows with a few bright tile rows / tile columns / tiles (the granularity the device decides at)
        img = (lo * rng.uniform(1.0, 3.0, (H, W, 3))).astype(np.float32)
        for _ in range(int(rng.integers(1, 5))):
            ty, tx = int(rng.integers(0, (H + 63) // 64)), int(rng.integers(0, (W + 255) // 256))
            ys = slice(ty * 64, ty * 64 + 64) if kind != len(hostile.SCRATCH96_KINDS) + 1 else slice(None)
            xs = slice(tx * 256, tx * 256 + 256) if kind != len(hostile.SCRATCH96_KINDS) + 2 else slice(None)
            img[ys, xs] = (lo * ratio * rng.uniform(0.5, 1.0, img[ys, xs].shape)).astype(np.float32)
    if rng.integers(0, 8) == 0:  # a few NaN samples
        for _ in range(int(rng.integers(1, 4))):
            img[int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(0, 2))] = np.nan
    return np.ascontiguousarray(img, dtype=np.float32)


def run_case(ctx, torch, rng, params, kernel, hostile, front=None, log=None):
    """One random case; returns (pairs, packed, allowed_by_model, violations)."""
    fft_channels = (0, 1)  # (the stand-in halation stencil leaves blue to a single tap)
    geo = geometry(kernel, fft_channels)
    bh, bw, ay, ax = geo
    H = int(rng.integers(130, 1100))
    W = int(rng.integers(75, 650)) * 4 if rng.integers(0, 5) else int(rng.integers(300, 2600))
    img = random_frame(rng, H, W, hostile)
    shard = bool(rng.integers(0, 2)) and H > 260
    if shard:
        y0 = int(rng.integers(0, H - 100))
        y1 = int(rng.integers(y0 + 50, H + 1))
    else:
        y0, y1 = 0, H
    below = bh - 1 - ay
    buf0 = max(y0 - ay - int(rng.integers(0, 120)) * int(rng.integers(0, 2)), 0) if y0 - ay > 0 else 0
    buf1 = min(y1 + below + int(rng.integers(0, 120)) * int(rng.integers(0, 2)), H) if y1 + below < H else H
    # reflection at a frame edge needs the rows it lands on (stencil_source_rows): keep the buffer generous there
    if y0 - ay < 0:
        buf1 = max(buf1, min(ay - y0 + 1, H))
    if y1 + below > H:
        buf0 = min(buf0, max(2 * (H - 1) - (y1 - 1 + below), 0))
    use_front = front is not None and bool(rng.integers(0, 3) == 0)
    use_render = front is not None and not shard and not use_front and bool(rng.integers(0, 3) == 0)
    ctx.write_frame_params(params)
    if use_render:
        # the whole-frame entry itself: r2f_render (front kernel recording, halation choosing, MTF, tail) -- launched kernel by kernel
        # on its first call, from a captured graph on the third; the exposure planes for the model come from an untracked front call
        src = torch.from_numpy(img).cuda()
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        for _ in range(int(rng.integers(1, 4))):
            ctx.render(src, front, out_f32=out)
        info = ctx.frame_exposure_range()
        flags = ctx.frame_scratch_flags()
        Efull = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_front(src, front, 0, dst=Efull)
        if not info["armed"]:
            return 0, 0, 0, []
        m = model(Efull.cpu().numpy(), geo, 0, H, H, W, 0, H, info["bound"], info["floor"], fft_channels)
        assert len(m) == len(flags), (len(m), len(flags), H, W)
        bad = [(pc, m[pc]) for pc in range(len(m)) if flags[pc] and not m[pc][0]]
        if log:
            log(f"H {H:4d} W {W:4d} r2f_render: pairs {len(m)} packed {int(flags.sum())} allowed by their own samples {sum(a for a, _, _ in m)}"
                f"{'  VIOLATIONS ' + str(bad) if bad else ''}")
        return len(m), int(flags.sum()), sum(a for a, _, _ in m), [(H, W, 0, H, 0, H, pc, v) for pc, v in bad]
    if use_front:
        # the exposure planes are made by the front kernel from `img` as a linear frame, which records what it writes; rows of the
        # buffer it does not write (none here: it writes the whole buffer) would stay unknown
        src = torch.from_numpy(img).cuda()
        Efull = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
        ctx.stage_front(src, front, 0, dst=Efull)  # rows outside the buffer: written, not recorded (never read by the call)
        ctx.write_frame_params(params)
        ctx.stage_front(src[buf0:buf1], front, 0, in_gy0=buf0, dst=Efull[:, buf0:buf1], dst_gy0=buf0, y0=buf0, y1=buf1, H_global=H,
                        track_range=True)
    else:
        Efull = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
        # (every row of the buffer is recorded: that is what R2F_F_RANGE_VALID vouches for -- a tile counts as known once ANY of its
        # rows was recorded, so a caller that vouches for a buffer whose halo rows it did not record breaks the contract, not the rule)
        ctx.stage_exposure_range(Efull[:, buf0:buf1], src_gy0=buf0, y0=buf0, y1=buf1)
    D = torch.empty((3, y1 - y0, W), dtype=torch.float32, device="cuda")
    ctx.stage_halation(Efull[:, buf0:buf1], D, params, src_gy0=buf0, dst_gy0=y0, y0=y0, y1=y1, H_global=H, range_valid=True)
    info = ctx.frame_exposure_range()
    flags = ctx.frame_scratch_flags()
    if not info["armed"]:
        return 0, 0, 0, []
    E = Efull.cpu().numpy()
    m = model(E, geo, y0, y1, H, W, buf0, buf1, info["bound"], info["floor"], fft_channels)
    assert len(m) == len(flags), (len(m), len(flags), H, W, y0, y1)
    bad = [(pc, m[pc]) for pc in range(len(m)) if flags[pc] and not m[pc][0]]
    if log:
        log(f"H {H:4d} W {W:4d} rows [{y0}, {y1}) buffer [{buf0}, {buf1}) {'front' if use_front else 'range'}: pairs {len(m)} packed {int(flags.sum())} "
            f"allowed by their own samples {sum(a for a, _, _ in m)}{'  VIOLATIONS ' + str(bad) if bad else ''}")
    return len(m), int(flags.sum()), sum(a for a, _, _ in m), [(H, W, y0, y1, buf0, buf1, pc, v) for pc, v in bad]


def setup(ctx):
    """The halation stencil of the 100 MP pitch (87 x 87 taps, an 85 x 85 box), the stand-in stock's tables, 256 x 512 windows."""
    from helpers import SEED, oracle_inputs, stocks
    from test_gpu_parity import setup_ctx

    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 341.33, seed=SEED)
    front = setup_ctx(ctx, p)
    ctx.set_option("stencil_fft_window_rows", NY)
    ctx.set_option("stencil_fft_window", NX)
    return ctx.make_params(halation=True), front, np.asarray(p.halation_kernel)


def soak(ctx, torch, budget, seed, log=None):
    import hostile

    params, front, kernel = setup(ctx)
    rng = np.random.default_rng(seed)
    tot = np.zeros(3, dtype=np.int64)
    bad = []
    for _ in range(budget):
        n, packed, allowed, v = run_case(ctx, torch, rng, params, kernel, hostile, front=front, log=log)
        tot += (n, packed, allowed)
        bad += v
    return tot, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import torch

    from raw2film_amd.context import HipContext

    ctx = HipContext(0, lib_path=args.lib)
    tot, bad = soak(ctx, torch, args.budget, args.seed, log=print if args.verbose else None)
    print(f"# tools/scratch_choice_model.py --budget {args.budget} --seed {args.seed}{' --lib ' + args.lib if args.lib else ''}: {tot[0]} window pairs in {args.budget} random calls "
          f"(whole frames and row shards; record by the range kernel, the front kernel, r2f_render)")
    print(f"#   took the 12-byte element: {tot[1]}   allowed by their own samples (exact per-pair range): {tot[2]}   "
          f"-> the 64 x 256 tiles cost {tot[2] - tot[1]} pairs ({100.0 * (tot[2] - tot[1]) / max(tot[2], 1):.1f} % of the allowed ones)")
    print(f"#   pairs that took the element AGAINST their own samples: {len(bad)}")
    for b in bad[:20]:
        print("   ", b)
    ctx.close()
    sys.exit(1 if bad else 0)
