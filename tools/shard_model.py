"""Device time of a 1/N row shard of the 100 MP frame on ONE GPU (what each of N GPUs would do per frame), eager launches vs HIP
graph replay: the measured inputs of DESIGN.md section 5's modelled 1 -> 8 GPU curve.

Round 5: every schedule RowShardedRenderer can run (one exchange with the halation in one call or as interior + bands, two
exchanges) is measured for the middle rank, then the renderer's own measured choice ("auto").
Round 4: the shard measured is a MIDDLE rank's -- RowShardedRenderer(rank = N // 2, world = N) with the exchange itself stubbed out
(nothing travels: the halo rows keep whatever the buffer held), so the launches are exactly a real rank's: front kernels on the
own rows, the interior halation, the two boundary bands, the halation of the MTF's halo rows, MTF, tail.  Round 3 measured a
(rows x W) frame of its own, i.e. left the halo work to the model.

    python tools/shard_model.py            # on the GPU box
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raw2film_amd import HipProcessor, filmstock, stencils  # noqa: E402
from raw2film_amd.hip_processor import REC709_TO_XYZ  # noqa: E402
from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer  # noqa: E402
from raw2film_amd.synthetic import synthetic_frame_device  # noqa: E402


class NoTransport(RowShardedRenderer):
    """A rank whose neighbours never answer: every launch of a real rank, no bytes on any wire."""

    def _exchange(self, buf, buf_gy0, above, below, wait=True):
        return None


W, H = 12288, 8192
stocks = filmstock.builtin_stocks()
neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
proc = HipProcessor(device=0)
scale = W / 36.0
hal_k = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
mtf_k = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
params = proc.prepare(neg, 6, 0.4, (W, H), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=24,
                      exp_kelvin=6000, color_masking=1.0, halation_green_factor=0.3)
be = HipStageBackend.for_stencils(proc.ctx, params, hal_k, mtf_k)
ha, ma = be.halation_taps, be.mtf_taps
per = be.halation_taps_per_channel
mb_all = (ha[0] + ma[0]) * W * 3 * 4 / 1e6
mb_per = sum(a + ma[0] for a, _ in per) * W * 4 / 1e6
print(f"halo rows: halation {ha} (per plane {per}), MTF {ma}; exchange per boundary and direction: {mb_per:.2f} MB "
      f"(all planes with the full halo: {mb_all:.2f} MB)")
SCHEDULES = (("1 exchange, halation in one call", dict(exchanges=1, split_halation=False)),
             ("1 exchange, interior halation + bands", dict(exchanges=1, split_halation=True)),
             ("2 exchanges, halation on the own rows", dict(exchanges=2)),
             ("measured on the first frames (auto)", dict()))


def run(n, rank, graph, kw):
    # (R2F_SHARD_DYN=0: A/B without the exposure-range record -- the halation keeps complex128 scratch whatever the rows hold)
    rr = NoTransport(be, H, W, halation=True, mtf=True, rank=rank, world=n, graph=graph,
                     dyn_scratch=os.environ.get("R2F_SHARD_DYN", "1") != "0", **kw)
    rows = rr.plan.rows
    frame = synthetic_frame_device(rows, W, seed=n)
    out = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    i = 0
    while rr.tuning:  # the measuring phase of "auto"
        rr.render(frame, out_f32=out, seed=100 + i)
        i += 1
    for _ in range(3):  # ... then the frames that capture the graphs
        rr.render(frame, out_f32=out, seed=100 + i)
        i += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 10
    for i in range(iters):
        rr.render(frame, out_f32=out, seed=200 + i)
    t_host = (time.perf_counter() - t0) / iters * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / iters * 1e3
    info = (rows, rr.schedule, rr.split, rr.d_lo, rr.d_hi, rr.tuned_ms)
    del rr, frame, out
    return t_all, t_host, info


base = None
ONLY = os.environ.get("R2F_SHARD_ONLY")  # "8": just the middle rank of 8 on its two-exchange schedule (for a run under rocprofv3)
for n in ((int(ONLY),) if ONLY else (1, 2, 4, 8)):
    rank = n // 2 if n > 2 else 0
    for name, kw in ((SCHEDULES[2:3] if ONLY else SCHEDULES) if n > 1 else SCHEDULES[:1]):
        res = {g: run(n, rank, g, kw) for g in (False, True)}
        rows, sched, split, dl, dh, tuned = res[True][2]
        if n == 1 or base is None:
            base = res[True][0] * n
        what = f"schedule {sched}" + (f", halation interior [{split[0]}, {split[1]}) of [{dl}, {dh})" if split else "")
        if tuned:
            what += ", measured " + " / ".join(f"{t:.3f}" for t in tuned) + " ms per candidate"
        print(f"N = {n}: rank {rank}, shard {W} x {rows}, {name} ({what}): eager {res[False][0]:.3f} ms/frame (host issue {res[False][1]:.3f}), "
              f"graph replay {res[True][0]:.3f} ms/frame (host issue {res[True][1]:.3f})  -> 1/N of the N = 1 time would be {base / n:.3f}", flush=True)
for which, name in ((0, "halation"), (1, "MTF")):
    print(name, "windows of the last call:", [c["window"] for c in proc.ctx.stencil_stats(which)])

# ... and round 3's measure for comparison: a (rows x W) frame of its own (no halo rows, one halation call)
for n in (() if ONLY else (8,)):
    rows = H // n
    fh = 36.0 * rows / W
    params1 = proc.prepare(neg, 6, 0.4, (W, rows), seed=20260630, matrix=REC709_TO_XYZ, print_film=prt, frame_width=36, frame_height=fh,
                           exp_kelvin=6000, color_masking=1.0, halation_green_factor=0.3)
    be1 = HipStageBackend.for_stencils(proc.ctx, params1, hal_k, mtf_k)
    rr = RowShardedRenderer(be1, rows, W, halation=True, mtf=True, rank=0, world=1, graph=True)
    frame = synthetic_frame_device(rows, W, seed=n)
    out = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    for i in range(3):
        rr.render(frame, out_f32=out, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        rr.render(frame, out_f32=out, seed=10 + i)
    torch.cuda.synchronize()
    print(f"a {W} x {rows} frame of its own (round 3's measure), graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/frame")
