"""Row shards through the stage entry points against the whole frame, under random frame sizes, scales, stage sets and UNEVEN shard
bounds (shards down to a few rows: smaller than the stencils' reach, so a shard's halo spans several neighbours' rows -- which the
stage calls do not care about: they see global rows).  Direct stencils: bit for bit.  FFT stencils (windows anchored at the call's
first row): one float32 ulp on a handful of pixels.  Fixed seeds; R2F_SHARD_FUZZ_CASES for a soak."""

import os

import numpy as np
import pytest

from helpers import oracle_inputs, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cases(n=int(os.environ.get("R2F_SHARD_FUZZ_CASES", "10"))):
    rng = np.random.default_rng(int(os.environ.get("R2F_SHARD_FUZZ_SEED", "77")))
    out = []
    for _ in range(n):
        H, W = int(rng.integers(40, 700)), int(rng.integers(8, 600))
        k = int(rng.integers(1, 7))
        cuts = sorted(set(int(v) for v in rng.integers(1, H, size=k)))
        out.append(dict(H=H, W=W, scale=float(rng.choice([40.0, 97.3, 166.67, 341.33])), bounds=[0] + cuts + [H],
                        halation=bool(rng.integers(0, 4) > 0), mtf=bool(rng.integers(0, 4) > 0), grain=int(rng.integers(0, 3)),
                        fft=int(rng.integers(0, 2)), bw=bool(rng.integers(0, 5) == 0), seed=int(rng.integers(0, 2**31)),
                        burn=float(rng.choice([0.0, 0.0, 0.5])),
                        # round 6: the shard's calls keep the exposure-range record (front records, halation vouches), so its FFT
                        # passes may choose the 12-byte scratch element from the SHARD's range where the whole frame chose from its own
                        dyn=bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"{c['H']}x{c['W']}-s{c['scale']:.0f}-h{int(c['halation'])}m{int(c['mtf'])}g{c['grain']}-fft{c['fft']}-{len(c['bounds']) - 1}shards")
def test_uneven_row_shards_equal_the_whole_frame(c):
    from raw2film_amd.context import HipContext
    from test_gpu_parity import setup_ctx

    neg, prt, bw = stocks()
    stock = bw if c["bw"] else neg
    H, W = c["H"], c["W"]
    p = oracle_inputs(stock, prt, c["scale"], halation=c["halation"], mtf=c["mtf"], grain=c["grain"], seed=c["seed"])
    ctx = HipContext(0)
    try:
        params = setup_ctx(ctx, p)
        ctx.set_option("stencil_fft", c["fft"])
        if c["burn"]:
            params.flags |= 32
            params.burn_cell, params.burn_strength, params.burn_d_ref = max(1, int(np.ceil(min(H, W) / 20.0))), c["burn"], float(stock.d_ref[1] if len(stock.d_ref) > 1 else stock.d_ref[0])
            if H // params.burn_cell < 1 or W // params.burn_cell < 1:
                params.flags &= ~32
        frame = torch.from_numpy(synthetic_frame(H, W, seed=c["seed"] % 991)).cuda()
        whole, _ = ctx.render(frame, params)
        rh = p.halation_kernel.shape[0] // 2 if p.halation_kernel is not None else 0
        rm = p.mtf_kernel.shape[0] // 2 if p.mtf_kernel is not None else 0
        burn = bool(params.flags & 32)
        parts, sums = [], None
        for a, b in zip(c["bounds"][:-1], c["bounds"][1:]):
            d_lo, d_hi = max(a - rm, 0), min(b + rm, H)
            e_lo, e_hi = max(d_lo - rh, 0), min(d_hi + rh, H)
            cur = torch.empty((3, e_hi - e_lo, W), dtype=torch.float32, device="cuda")
            lo = e_lo
            if p.halation_kernel is not None:
                if c["dyn"]:
                    ctx.write_frame_params(params)  # the start of a shard's frame: the record is reset
                ctx.stage_front(frame[e_lo:e_hi], params, 0, in_gy0=e_lo, dst=cur, dst_gy0=e_lo, H_global=H, track_range=c["dyn"])
                D = torch.empty((3, d_hi - d_lo, W), dtype=torch.float32, device="cuda")
                ctx.stage_halation(cur, D, params, src_gy0=e_lo, dst_gy0=d_lo, y0=d_lo, y1=d_hi, H_global=H, range_valid=c["dyn"])
                cur, lo = D, d_lo
            else:
                cur = torch.empty((3, d_hi - d_lo, W), dtype=torch.float32, device="cuda")
                ctx.stage_front(frame[d_lo:d_hi], params, 1, in_gy0=d_lo, dst=cur, dst_gy0=d_lo, H_global=H)
                lo = d_lo
            if p.mtf_kernel is not None:
                D2 = torch.empty((3, b - a, W), dtype=torch.float32, device="cuda")
                ctx.stage_mtf(cur, D2, params, src_gy0=lo, dst_gy0=a, y0=a, y1=b, H_global=H)
                cur, lo = D2, a
            if burn:
                if p.grain_lut is not None:
                    G = torch.empty((3, b - a, W), dtype=torch.float32, device="cuda")
                    ctx.stage_grain(cur, G, params, src_gy0=lo, dst_gy0=a, y0=a, y1=b, H_global=H)
                    cur, lo = G, a
                s = ctx.stage_burn_sums(cur, params, src_gy0=lo, y0=a, y1=b, H_global=H)
                sums = s if sums is None else sums + s
                parts.append((a, b, cur, lo))
            else:
                part = torch.empty((b - a, W, 3), dtype=torch.float32, device="cuda")
                ctx.stage_tail(cur, params, src_gy0=lo, out_f32=part, out_gy0=a, y0=a, y1=b, H_global=H)
                parts.append((a, b, part, None))
        got = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        if burn:
            m = ctx.stage_burn_map(sums, params, W=W, H_global=H)
            q = type(params).from_buffer_copy(params)
            q.flags &= ~8  # the grain is in the planes already
            for a, b, cur, lo in parts:
                ctx.stage_tail(cur, q, src_gy0=lo, out_f32=got[a:b], out_gy0=a, y0=a, y1=b, H_global=H, burn_map=m)
        else:
            for a, b, part, _ in parts:
                got[a:b] = part
        fft_used = c["fft"] and ((p.halation_kernel is not None and any(s["fft"] for s in ctx.stencil_stats(0)))
                                 or (p.mtf_kernel is not None and any(s["fft"] for s in ctx.stencil_stats(1))))
        if fft_used or burn:  # (the burn's cell sums add up in another order across shards)
            diff = (got - whole).abs()
            assert float((diff / whole.abs().clamp_min(1e-3)).max()) <= (2e-6 if (burn or c["dyn"]) else 1.5e-6), c
        else:
            assert torch.equal(got, whole), c
    finally:
        ctx.close()
