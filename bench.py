#!/usr/bin/env python3
"""bench.py -- megapixels/s of the full film pipeline (neg + print + grain + halation + MTF).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4_100mp|cfg3_45mp|cfg2_24mp|cfg5_batch]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`--gpus N` > 1 without a launcher: bench.py starts its N ranks itself, as a child process running torch.distributed.run.)

A "step" is one pass of the hot path (S0..S8, float32 output) over one synthetic decoded frame that is already
resident in HBM.  N = 1: the whole frame on one MI355X.  N > 1: the SAME frame, row-sharded over N GPUs with the RCCL
neighbour exchange of raw2film_amd.sharding (strong scaling: total work fixed).  Rank 0 prints ONE JSON line.

Objects on that line besides the contract's keys:
  roofline      SURVEY.md 8(d)'s definition: 24 algorithmic bytes per pixel (12 read + 12 written) x the frame, over the
                step time, against the 8 TB/s HBM peak (`frac`); the 12 B/px read-only variant; `traffic` = the L2<->fabric
                bytes of one step from the committed PMC capture of the SAME sources and configuration (null when the
                capture does not match what is running; provenance beside it); `dominant_kernel` = the FFT pass with the
                largest share of the step, timed live with HIP events on its launch streams; `fp64_valu` = the FFT
                passes' fp64 work against the 78.6 TFLOP/s vector peak.
                `copy_ceiling_GBps` / `frac_of_copy_ceiling`: a float4 streaming copy of the frame's own 12 B/px in -> 12 B/px
                out, measured in this run (r2f_stream_copy) -- the practical ceiling beside the spec peak.
  ms_per_step_median / _min / _max   HIP events between consecutive steps on the launch stream (ms_per_step is the wall clock).
  rccl_ranks    N > 1: an all-reduce (SUM) of ones over the process group that carried the halo exchange.
  stage_ms      per-stage device time (events on the launch stream, inside the timed steps).
  cpu_baseline  the NumPy oracle ("port") timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (FMA = 2 flop): 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
GRAIN_SEED = 20260630
BASELINE_METRIC = "megapixels/sec full film pipeline (neg+print+grain+halation+MTF), 100MP frame"
METRICS = {
    "cfg4_100mp": BASELINE_METRIC,
    "cfg3_45mp": "megapixels/sec full film pipeline (neg+print+grain+halation+MTF), 45MP frame (BASELINE config 3)",
    "cfg2_24mp": "megapixels/sec negative+print LUT pipeline, effects off, 24MP frame (BASELINE config 2)",
    "cfg5_batch": "megapixels/sec full film pipeline, batch of 24MP frames, device-resident compute only "
                  "(BASELINE config 5 without the host phase and the PCIe copies; see pcie_inclusive)",
}


def strip_comments(text: str) -> str:
    """C / C++ source without its comments and with runs of white space collapsed: what the compiler sees, roughly.  String and
    character literals are kept as they are (a // inside one is not a comment)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":  # a literal: copy up to the closing quote, minding escapes
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            while j > 0 and text[j - 1] == "\\":  # a line comment continued with a backslash
                j = text.find("\n", j + 1)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_hash() -> str:
    """sha256 over the HIP sources and the C header WITHOUT their comments and white space: ties a committed PMC capture to the
    code that was profiled, and keeps it tied across commits that only touch comments (round 5 re-measured unchanged code six
    times because the hash covered the raw text: VERDICT r5, weak 8)."""
    h = hashlib.sha256()
    for rel in ("raw2film_amd/csrc/r2f_device.h", "raw2film_amd/csrc/r2f_launch.h", "raw2film_amd/csrc/r2f_kernels.hip",
                "raw2film_amd/csrc/r2f_fft.hip", "raw2film_amd/csrc/r2f_fft_math.h", "raw2film_amd/csrc/r2f_front.hip", "raw2film_amd/csrc/r2f_post.hip", "raw2film_amd/csrc/r2f_api.hip", "raw2film_amd/csrc/r2f_plan.cpp",
                "raw2film_amd/csrc/r2f_plan.h", "include/r2f.h"):
        with open(os.path.join(ROOT, rel), "r", encoding="utf-8") as f:
            h.update(strip_comments(f.read()).encode())
            h.update(b"\0")
    return h.hexdigest()[:16]


def fft_fp64_ops(ny: int, nx: int, vy: int):
    """(flops, fp64 VALU lane-instructions) of the three FFT passes for ONE window pair, counted from r2f_fft.hip:
    dft16 = 128 add + 9 complex multiplies (2 mul + 2 fma each); twiddle_powers = 29 complex multiplies; a 256-point line =
    16 lanes x (2 dft16 + twiddles); a 512-point line = 32 lanes x (the same + the radix-2 step: 32 add, 15 multiplies on the
    odd lanes); pass 2 adds one complex multiply per element (the kernel spectrum)."""
    cm_f, cm_i = 6, 4                       # complex multiply: flops, instructions
    d16_f, d16_i = 128 + 9 * cm_f, 128 + 9 * cm_i
    tw_f, tw_i = 29 * cm_f, 29 * cm_i
    lane256 = (2 * d16_f + tw_f, 2 * d16_i + tw_i)
    lane512 = (lane256[0] + 32 + 7.5 * cm_f, lane256[1] + 32 + 7.5 * cm_i)

    def line(n):
        return (16 * lane256[0], 16 * lane256[1]) if n == 256 else (32 * lane512[0], 32 * lane512[1])

    rx, cy = line(nx), line(ny)
    mult = (ny * nx * cm_f, ny * nx * cm_i)
    flops = ny * rx[0] + nx * 2 * cy[0] + mult[0] + vy * rx[0]
    insts = ny * rx[1] + nx * 2 * cy[1] + mult[1] + vy * rx[1]
    return flops, insts


def self_launch(n: int) -> int:
    """Run this script under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous, a free port) as a child
    process; rank 0's JSON line goes to this process's stdout unchanged.  Returns the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4_100mp", choices=["cfg4_100mp", "cfg3_45mp", "cfg2_24mp", "cfg5_batch"])
    ap.add_argument("--frames", type=int, default=64, help="cfg5_batch: frames per step, dealt round-robin to the ranks")
    ap.add_argument("--frame", default="noise", choices=["noise", "smooth"],
                    help="synthetic frame statistics: independent pixels (headline; worst case for the LUT gathers) or photograph-like")
    ap.add_argument("--clamp", default="",
                    help="lo,hi: clamp the synthetic frame's samples (NOT the headline input: a frame whose exposure range the guard of "
                         "the halation's 12-byte FFT scratch element accepts, e.g. 0.004,48 -- the labelled second capture of tools/profile_round.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the two extra steps that time the dominant FFT pass with one internal stream (tools/profile_round.sh: "
                         "keeps the profiled launches all of one size)")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip every eager step behind the timed ones (stage times, per-pass times): the run then holds nothing but "
                         "the product's own renders -- what tools/profile_round.sh profiles, so that calls and bytes per kernel are "
                         "whole multiples of the renders")
    ap.add_argument("--no-pcie", action="store_true",
                    help="skip the host <-> device legs (cfg5_batch: the PCIe-inclusive BatchSharder leg; the others: host_device_copies)")
    ap.add_argument("--no-graph", action="store_true",
                    help="A/B: launch every kernel from the host instead of replaying the frame's HIP graph (the default: everything "
                         "downstream of the halo exchange is captured once and replayed, raw2film_amd/sharding.py)")
    ap.add_argument("--output", default="f32", choices=["f32", "u8"],
                    help="f32 (default, the contract's fp32 HWC frame: 24 algorithmic B/px) or u8: the reference's own output stage "
                         "S9 as the only output (SURVEY.md 8d: 12 + 3 = 15 B/px) -- single GPU, frame configurations")
    ap.add_argument("--checksum", action="store_true",
                    help="validation: every rank takes its rows of ONE full frame (seed 1234) and the JSON line carries a checksum of "
                         "the whole output, comparable across --gpus values")
    ap.add_argument("--side-grain", action="store_true", help="A/B: make the grain field on a side stream while the stencils run")
    ap.add_argument("--direct-stencils", action="store_true", help="A/B: run the stencils in their direct fp32 form instead of fp64 FFTs")
    ap.add_argument("--opt", action="append", default=[], help="A/B: r2f_set_option name=value (repeatable)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the measured configuration); gloo + --same-device validates the N > 1 code path on one GPU")
    ap.add_argument("--same-device", action="store_true", help="validation only: every rank uses cuda:0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--parse-only", action="store_true",
                    help="print the parsed arguments as JSON and exit without touching torch or the GPU (tests/test_bench_host.py checks "
                         "the command lines of tools/first_multi_gpu.sh with it)")
    args = ap.parse_args()
    if args.parse_only:
        print(json.dumps(vars(args)))
        return

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before HIP initialises
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD (torch.distributed.run, one process per GPU)
        # and hand its return code back.  Nothing has touched HIP in this process yet (torch is not even imported), and the
        # ranks are children, never an exec of this process.
        raise SystemExit(self_launch(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    from raw2film_amd import HipProcessor, filmstock, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import BatchSharder, HipStageBackend, RowShardedRenderer
    from raw2film_amd.synthetic import CONFIGS, synthetic_frame_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # under a launcher its world size is the truth
        args.gpus = world
    if args.same_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():  # (device_count() does not initialise HIP)
        raise SystemExit(f"bench.py: rank {rank} of {world} needs GPU {local_rank}, but this node has {torch.cuda.device_count()} "
                         f"GPU(s) visible: --gpus {world} needs {world} (validation on one GPU: --backend gloo --same-device)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    batch = args.config == "cfg5_batch"  # BASELINE config 5: 64 x 24 MP frames, full pipeline, frame-per-GPU, no collectives
    W, H = CONFIGS["cfg2_24mp" if batch else args.config]
    effects = args.config != "cfg2_24mp"  # config 2 = negative + print LUTs only
    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    settings = dict(print_film=prt, frame_width=36, frame_height=24, exp_kelvin=6000, color_masking=1.0,
                    halation=effects, halation_size=1.0, halation_green_factor=0.3, halation_intensity=1.0,
                    sharpness=effects, sharpening_strength=0.0, grain=2 if effects else 0)

    proc = HipProcessor(device=local_rank)
    if args.direct_stencils:
        proc.ctx.set_option("stencil_fft", 0)
    for o in args.opt:
        k, v = o.split("=")
        proc.ctx.set_option(k, int(v))
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=GRAIN_SEED, matrix=REC709_TO_XYZ, **settings)
    scale = max(H, W) / 36.0
    hal_k = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3) if effects else None
    mtf_k = stencils.mtf_stencil(neg, scale, 0.0, 1.0) if effects else None
    backend = HipStageBackend.for_stencils(proc.ctx, params, hal_k, mtf_k)
    use_graph = not args.no_graph and effects and not args.side_grain
    # N = 1 (and every rank of the frame-per-GPU batch): the product's own entry -- HipProcessor.process_array on device tensors,
    # i.e. prepare() + r2f_render, with a NEW grain seed every frame like the reference's renders (gpu_processor.py:585-597).
    # r2f_render replays the frame's launches from a HIP graph (one submit per frame); the seed travels in a device-side block.
    # N > 1: one frame row-sharded over the ranks (RowShardedRenderer: halo exchange + the stage entry points, graph replay
    # downstream of the exchange), also with a new seed every step.
    use_processor = (batch or world == 1) and not args.side_grain
    renderer = None
    if use_processor:
        if args.no_graph:
            proc.ctx.set_option("render_graph", 0)
        frames_here = len([i for i in range(args.frames) if i % world == rank]) if batch else 1
        r0, r1 = 0, H
    else:
        renderer = RowShardedRenderer(backend, H, W, halation=effects, mtf=effects, grain=effects, side_grain=args.side_grain,
                                      graph=use_graph)
        frames_here = 1
        r0, r1 = renderer.plan.r0, renderer.plan.r1
    replaying = (use_graph and effects) if use_processor else renderer.graph

    # this rank's rows of the synthetic frame, resident in HBM before the clock starts
    if args.checksum:
        whole_frame = synthetic_frame_device(H, W, seed=1234, device=f"cuda:{local_rank}", kind=args.frame)
        frame = whole_frame[r0:r1].contiguous()
        del whole_frame
    else:
        frame = synthetic_frame_device(r1 - r0, W, seed=1234 + rank, device=f"cuda:{local_rank}", kind=args.frame)
    if args.clamp:
        c_lo, c_hi = (float(v) for v in args.clamp.split(","))
        frame.clamp_(c_lo, c_hi)
    out_u8 = args.output == "u8"
    if out_u8 and not use_processor:
        raise SystemExit("--output u8 is measured through HipProcessor.process_array: one GPU, a frame configuration")
    out = torch.empty((r1 - r0, W, 3), dtype=torch.uint8 if out_u8 else torch.float32, device=frame.device)

    # per-stage device times come from eager steps bracketed with events (a replayed graph has no stage boundaries to time)
    from raw2film_amd.tracing import TimedBackend

    timed = TimedBackend(backend)
    if renderer is not None and not renderer.graph:
        renderer.backend = timed
    frame_no = [0]

    def next_seed():
        frame_no[0] += 1  # (--checksum compares frames across --gpus values: one seed)
        return GRAIN_SEED if args.checksum else (GRAIN_SEED + frame_no[0]) & 0xFFFFFFFF

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for _ in range(frames_here):
            if use_processor:
                proc.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=next_seed(), return_float=not out_u8,
                                   output="device", out=out, **settings)
            else:
                renderer.render(frame, out_f32=out, seed=next_seed())

    def drain_timing():
        return [proc.ctx.kernel_timing(cls) for cls in range(6)]  # (total ms, launches, algorithmic bytes) per class

    # N > 1: the renderer measures its candidate schedules (one / two exchanges, interior halation ahead of the exchange or not)
    # on its first frames -- untimed warm-up frames like the others, the same number on every rank
    tune_steps = 0
    while not use_processor and getattr(renderer, "tuning", False) and tune_steps < 32:
        step()
        tune_steps += 1
    for _ in range(max(args.warmup, 2 if replaying else 0)):  # (the graph is captured on the second frame)
        step()
    barrier()
    timed.reset()
    # events around every launch of the FFT column passes (the kernels with the largest share), on their launch streams
    # (eager launches only: a replayed graph carries no events -- its breakdown comes from two eager steps afterwards)
    if not replaying and not use_processor:
        proc.ctx.set_option("kernel_timing", 2)
    drain_timing()
    stats0 = proc.ctx.render_stats()
    # per-step device times (median of event-timed runs): one event between consecutive steps on the launch stream;
    # the contract's number stays the wall clock around all K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # which scratch element the halation's passes of the LAST TIMED frame took (r2f_render chooses on the device, frame by frame);
    # asked now, before the eager breakdown steps overwrite the frame block
    scratch_choice = proc.ctx.frame_exposure_range() if effects else None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=frame.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    n_frames = args.frames if batch else 1
    mp_per_s = H * W / 1e6 * n_frames * args.steps / dt
    full = "3x3 + 2-D LUT + halation + log/curve + MTF + grain + tetrahedral 3-D LUT"

    result = {
        "metric": METRICS[args.config],
        "value": mp_per_s,
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_median": step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
        "ms_per_step_min": step_ms[0],
        "ms_per_step_max": step_ms[-1],
        "ms_per_step_note": "ms_per_step: wall clock over the K timed steps (barrier + synchronize on both sides, max over ranks); "
                            "median / min / max: HIP events between consecutive steps on this rank's launch stream",
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": ("f32" if args.direct_stencils or not effects else
                  "f32 (pointwise stages, grain) + f64 (FFT stencils; the MTF's scratch images are complex64)"),
        "data": ("synthetic" if args.frame == "noise" else "synthetic (smooth, photograph-like frame: not the headline input)")
                + (f" (samples clamped to [{args.clamp}]: not the headline input)" if args.clamp else ""),
        "config": {
            "workload": f"{args.config}: " + (f"{args.frames} x " if batch else "") + f"{W}x{H} ({H * W / 1e6:.1f} MP) decoded linear-Rec.709 "
                        "frame, 36x24 mm, " + (f"full pipeline S0-S8: {full}" if effects else "LUTs only (S0+S1+S3+S4+S8), effects off")
                        + (f", stencils {hal_k.shape[0]}x{hal_k.shape[1]} / {mtf_k.shape[0]}x{mtf_k.shape[1]}" if effects else "")
                        + (", fp32 HWC in -> uint8 HWC out (S9, the only output)" if args.output == "u8" else ", fp32 HWC in -> fp32 HWC out"),
            "stocks": "synthetic stand-ins portra400_like + k2383_like (spectral_film_lut data unavailable offline)",
            "sharding": (f"batch of {args.frames} frames, frame i -> rank i mod {world}, no collectives" if batch else
                         "single GPU" if world == 1 else
                         f"row-sharded over {world} {'GPUs' if not args.same_device else 'ranks on one GPU'}, "
                         f"{'RCCL' if args.backend == 'nccl' else 'gloo (host-staged: validation only)'} halo exchange(s) per frame: "
                         "schedule measured on the first frames (shard_schedule)"),
            "options": args.opt,
        },
    }
    if not use_processor and world > 1:
        ex, sp = renderer.schedule or (1, False)
        result["config"]["shard_schedule"] = {
            "exchanges": ex, "interior_halation_ahead_of_the_exchange": bool(sp),
            "candidates": [list(c) for c in renderer._candidates], "measured_ms_max_over_ranks": renderer.tuned_ms,
            "note": "RowShardedRenderer times every candidate schedule on its first frames (events on the launch stream, exchange "
                    "included) and all ranks take the one whose slowest rank was fastest; 1 exchange = exposure halo widened by the MTF "
                    "reach, 2 = exposure halo, then density halo (saves a halation window row per shard)"}

    if use_processor:
        stats1 = proc.ctx.render_stats()
        result["config"]["launch"] = (
            "HipProcessor.process_array (device tensors in and out) -> prepare() + r2f_render, a new grain seed every frame; "
            + ("r2f_render replays the frame's captured HIP graph: one seed write + one graph launch per frame"
               if replaying else "one host launch per kernel (render_graph = 0)"))
        result["config"]["render_stats_timed_steps"] = {k: stats1[k] - stats0[k] for k in stats1}
    else:
        result["config"]["launch"] = ("RowShardedRenderer, a new grain seed every step; "
                                      + ("HIP graph replay of the frame's launches downstream of the halo exchange"
                                         if renderer.graph else "one host launch per kernel"))
    if use_dist:
        # how many ranks the collective backend really joined: an all-reduce (SUM) of ones over the process group that carried
        # the halo exchange ("nccl" = RCCL over xGMI on the GPU box)
        ones = torch.ones(1, dtype=torch.float32, device=frame.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        result["rccl_ranks" if args.backend == "nccl" else "gloo_ranks"] = int(ones.item())
        result["backend"] = args.backend
    if args.checksum:
        # order-independent 64-bit checksum of the fp32 output's bit patterns, summed over the ranks
        cs = out.view(torch.int32).to(torch.int64).sum().reshape(1)
        if use_dist:
            cs = cs.cpu() if args.backend == "gloo" else cs
            dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        result["checksum"] = int(cs.item())
    eager = None
    if args.no_breakdown:
        steps_for_cols = args.steps
    elif replaying or use_processor:  # the breakdowns below need per-launch events: the same frame, stage by stage, eagerly
        sched = {} if (use_processor or renderer is None or renderer.schedule is None) else \
            {"exchanges": renderer.schedule[0], "split_halation": renderer.schedule[1]}  # the schedule the timed steps ran, not a new measurement
        eager = RowShardedRenderer(timed, H, W, halation=effects, mtf=effects, grain=effects,
                                   **({"rank": 0, "world": 1} if use_processor else sched))
        if renderer is not None:
            eager.E, eager.D, eager.D2 = renderer.E, renderer.D, renderer.D2  # share the planes (no second 2.4 GB set)

        def step():  # noqa: F811 -- from here on the eager renderer
            for _ in range(frames_here):
                eager.render(frame, **{"out_u8" if out_u8 else "out_f32": out}, seed=next_seed())

        proc.ctx.set_option("kernel_timing", 2)
        drain_timing()
        timed.reset()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        steps_for_cols = 2
    else:
        steps_for_cols = args.steps
    stage_ms = timed.summary()
    result["stage_ms"] = {k: round(v, 4) for k, v in stage_ms.items()}
    if eager is not None:
        result["stage_ms_note"] = ("two eager stage-by-stage steps (RowShardedRenderer over the stage entry points) after the timed steps: the "
                                   "same kernels on the same windows -- the stage path keeps the exposure-range record like r2f_render, so "
                                   "the halation's passes choose the same scratch element")
    cols = drain_timing()  # only the column passes (classes 1 and 4) were on
    # the other two passes, for the breakdown only: two extra steps outside the timed region
    extra = [(0.0, 0, 0.0)] * 6
    if not args.no_breakdown:
        proc.ctx.set_option("kernel_timing", 5)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        extra = drain_timing()
    # ... and the dominant column pass with the GPU to itself (one internal stream): what a launch does when no other kernel
    # shares the CUs and the memory system with it
    solo = [(0.0, 0, 0.0)] * 6
    if not args.no_alone and not args.no_breakdown and effects:
        proc.ctx.set_option("kernel_timing", 2)
        proc.ctx.set_option("stencil_fft_streams", 1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        solo = drain_timing()
        proc.ctx.set_option("stencil_fft_streams", 2)
    proc.ctx.set_option("kernel_timing", 0)

    # ---- roofline, SURVEY.md 8(d): 24 algorithmic bytes per pixel over the whole step
    bpp = 15 if out_u8 else 24
    alg_bytes = float(bpp) * H * W * n_frames
    gbps = alg_bytes / (ms_per_step * 1e-3) / 1e9
    peak = HBM_PEAK_GBPS * world
    roof = {
        "bound": "hbm", "achieved": gbps, "peak": peak, "unit": "GB/s", "frac": gbps / peak,
        "definition": ("SURVEY.md 8(d): 12 B/px read + 3 B/px written (fp32 HWC in, uint8 HWC out: the only output) x the frame, over ms_per_step"
                       if out_u8 else "SURVEY.md 8(d): 12 B/px read + 12 B/px written (fp32 HWC in and out) x the frame, over ms_per_step")
                      + (f"; peak = {world} x 8 TB/s" if world > 1 else ""),
        "bytes_per_px": bpp,
        "read_only": {"bytes_per_px": 12, "achieved": gbps * 12 / bpp, "frac": gbps * 12 / bpp / peak,
                      "note": "the north_star's 'HBM-read roofline' variant: input bytes only"},
        "traffic": None,
    }
    result["roofline"] = roof
    # the practical ceiling beside the spec peak: a float4 streaming copy of the frame's own 12 B/px in and
    # 12 B/px out (r2f_stream_copy: input frame -> output frame buffer, the algorithmic bytes of one step and nothing else),
    # timed with events on the launch stream, best of 5 after a warm-up
    if not batch and frame.numel() == out.numel() and not out_u8:
        proc.ctx.stream_copy(frame, out)
        best = float("inf")
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            proc.ctx.stream_copy(frame, out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        ceil_gbps = 2.0 * frame.numel() * 4 / (best * 1e-3) / 1e9
        if use_dist and args.backend == "nccl":  # whole job: the ranks' ceilings add up
            t = torch.tensor([ceil_gbps], dtype=torch.float64, device=frame.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ceil_gbps = float(t.item())
        elif not args.same_device:  # (ranks that share one GPU share one ceiling: ADVICE r3)
            ceil_gbps *= world
        roof["copy_ceiling_GBps"] = ceil_gbps
        roof["copy_ceiling_ms"] = best
        roof["frac_of_copy_ceiling"] = gbps / ceil_gbps
        roof["copy_ceiling_note"] = ("measured in this run: float4 streaming copy of this rank's input rows into its output buffer "
                                     "(the step's algorithmic bytes, nothing else), best of 5; the step cannot be faster than copy_ceiling_ms")

    # L2 <-> fabric bytes of one step from the committed PMC capture, only when it was taken from these very sources on this
    # configuration (tools/profile_round.sh writes the file; it cannot be measured inside an un-profiled run)
    import glob

    traffic_rec = None
    captures = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{args.config}_hbm_traffic.json")))  # latest round last
    tfile = captures[-1] if captures else ""
    if tfile and world == 1 and not args.opt and not args.direct_stencils and not args.side_grain and not out_u8 and not args.clamp:
        rec = json.load(open(tfile))
        meta = rec.get("_meta", {})
        match = meta.get("source_hash") == source_hash() and meta.get("config") == args.config and meta.get("frame") == args.frame
        roof["traffic_provenance"] = {"file": os.path.relpath(tfile, ROOT), "collected": meta.get("date"), "command": meta.get("command"),
                                      "source_hash_then": meta.get("source_hash"), "source_hash_now": source_hash(), "match": match}
        if match:
            traffic_rec = rec
            roof["traffic"] = meta["bytes_per_step"]
            roof["traffic_over_algorithmic"] = meta["bytes_per_step"] / alg_bytes
            roof["traffic_note"] = ("FETCH_SIZE x 2 + WRITE_SIZE per step, separate --pmc passes (gfx950 tallies 128-B reads at 64 B): "
                                    "L2 <-> fabric bytes; Infinity Cache hits are counted, so this is an upper bound on HBM bytes")

    if effects and "halation" in stage_ms:
        st_h, st_m = proc.ctx.stencil_stats(0), proc.ctx.stencil_stats(1)
        # Which scratch element the halation's passes took: r2f_render chooses on the device per frame (asked after the last TIMED
        # frame), and the eager stage-by-stage steps behind the pass times choose the same way since round 6 (the stage path keeps
        # the exposure-range record too) -- asked again after them: the two must agree, or the breakdown is of other kernels.
        rng = scratch_choice
        rng_eager = proc.ctx.frame_exposure_range()
        if rng is not None:
            roof["halation_scratch_element"] = dict(
                rng, eager_breakdown_took_the_same_element=bool(rng_eager and rng_eager["armed"] == rng["armed"]
                                                                and rng_eager["pairs"] == rng["pairs"]
                                                                and rng_eager["packed_pairs"] == rng["packed_pairs"]),
                note="what r2f_render's front kernel recorded about the exposure planes of the last timed frame (min / max_abs: the FRAME's "
                     "extremes) and what the halation's FFT passes made of the record: the choice is per WINDOW PAIR -- a pair takes the "
                     "12-byte element (doubles rounded to 48 bits) when the max |x| of its two windows <= bound x max(their min, floor), "
                     "else complex128; packed_pairs of pairs (per channel) took it, twelve_byte_element = all of them.  stage_ms / "
                     "fft_pass_ms_per_step come from eager stage calls that make the same choices from the same record "
                     "(R2F_F_TRACK_RANGE / R2F_F_RANGE_VALID)")
        src_rng = rng
        if rng_eager is not None and rng is not None and not roof["halation_scratch_element"]["eager_breakdown_took_the_same_element"]:
            src_rng = rng_eager  # label the breakdown by what IT ran
        armed = bool(src_rng and src_rng["armed"])
        packed_frac = (src_rng["packed_pairs"] / src_rng["pairs"]) if (src_rng and src_rng.get("pairs")) else 0.0
        packed = packed_frac > 0.0
        el_h = ("12-byte element" if packed_frac == 1.0 else
                f"12-byte element on {packed_frac:.1%} of the window pairs, complex128 on the rest" if packed else "complex128")
        names = [f"rows_fwd ({el_h})", f"cols ({el_h})", f"rows_inv ({el_h})", "rows_fwd (complex64)", "cols (complex64)",
                 "rows_inv (complex64)"]
        per_step = [extra[0][0] / 2, cols[1][0] / steps_for_cols, extra[2][0] / 2, extra[3][0] / 2, cols[4][0] / steps_for_cols, extra[5][0] / 2]
        roof["fft_pass_ms_per_step"] = {n: round(v, 4) for n, v in zip(names, per_step) if v > 0}
        roof["fft_pass_note"] = ("event-bracketed launch times summed per step; launches alternate between two internal streams, so "
                                 "the sums exceed the stage wall time. cols: inside the timed steps when those launch kernel by kernel, else -- "
                                 "like rows_* -- from two extra eager steps after them (the same kernels: see halation_scratch_element)")
        # The library sums the passes' algorithmic bytes with 16-byte scratch elements for launches that choose on the device; when
        # they took the 12-byte element the scratch part of those sums is a quarter smaller: pass 1 moves window floats (8 B per
        # element of a pair's image) + the image, pass 2 the image + its valid rows, pass 3 the valid rows + the outputs
        win_h = next((c["window"] for c in st_h if c["fft"]), None)
        byte_scale = [1.0] * 6
        if packed and win_h:
            nzh = np.nonzero(hal_k[..., 0])
            bw_h = int(nzh[1].max() - nzh[1].min() + 1)
            vx_h = (win_h[1] - bw_h + 1) & ~3
            q = 0.25 * packed_frac  # (the share of the pairs that took the element: their scratch part is a quarter smaller)
            byte_scale[0] = 1.0 - q * 16.0 / (16.0 + 8.0)
            byte_scale[1] = 1.0 - q
            byte_scale[2] = 1.0 - q * (win_h[1] * 16.0) / (win_h[1] * 16.0 + 2.0 * vx_h * 4.0)
        cols = [(ms, n, b * byte_scale[c]) for c, (ms, n, b) in enumerate(cols)]
        extra = [(ms, n, b * byte_scale[c]) for c, (ms, n, b) in enumerate(extra)]
        solo = [(ms, n, b * byte_scale[c]) for c, (ms, n, b) in enumerate(solo)]
        # The dominant kernel of a step: the tail (one launch per frame) or one of the six FFT pass classes (summed per step)
        cand = {"tail": float(stage_ms.get("tail", 0.0))}
        cand.update({f"fft{c}": v for c, v in enumerate(per_step)})
        top = max(cand, key=cand.get)
        dom = 1 if cols[1][0] >= cols[4][0] else 4
        if top == "tail" and not batch:
            tail_ms = cand["tail"]
            tb = 24.0 * (r1 - r0) * W if not out_u8 else 15.0 * (r1 - r0) * W
            roof["dominant_kernel"] = {
                "kernel": "r2f::tail_kernel<4, true> (S6 hash noise + separable 9 x 9 grain stencil + grain LUT + clip + S8 tetrahedral 3-D LUT "
                          "+ interleaved store: planar fp32 density in, HWC out)",
                "bound": "hbm", "achieved": tb / (tail_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": tb / (tail_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms": tail_ms, "launches_per_step": 1.0, "bytes_per_launch": tb,
                "bytes_counted": "12 B/px of density planes read + the output written (12 B/px fp32, 3 B/px uint8): its algorithmic bytes; the noise is generated, the LUTs are cache-resident",
                "share_of_step": tail_ms / ms_per_step,
                "note": "the largest single kernel of a step since round 5 (event-timed eager stage call); the FFT pass classes follow in "
                        "fft_pass_ms_per_step, the largest of them in largest_fft_pass"}
            if traffic_rec is not None:
                for name, v in traffic_rec.items():
                    if isinstance(v, dict) and "tail_kernel" in name and "hbm_bytes_per_launch" in v:
                        roof["dominant_kernel"]["traffic"] = v["hbm_bytes_per_launch"]
                        roof["dominant_kernel"]["traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / tb
        if cols[dom][1] > 0:
            tot_ms, launches, bytes_alg = cols[dom]
            stats = st_h if dom == 1 else st_m
            win = next((c["window"] for c in stats if c["fft"]), None)
            real = any(c.get("real_spectrum") for c in stats if c["fft"])
            walk = real and win[0] == 256
            g = bytes_alg / (tot_ms * 1e-3) / 1e9
            # (ST template argument: 1 complex64, 0 complex128, 3 = element chosen on the device at the top of the kernel)
            st_arg = 1 if dom == 4 else (3 if armed else 0)
            kname = (f"r2f::fft_cols_walk_kernel<{win[1] // 16}, {st_arg}>" if walk else
                     f"r2f::fft_cols_kernel<{win[1] // 16}, {'true' if win[0] == 512 else 'false'}, {1 if dom == 4 else 0}, {'true' if real else 'false'}>")
            dk = {
                "kernel": f"{kname} "
                          f"(pass 2 of the fp64 overlap-save FFT of the {'MTF' if dom == 4 else 'halation'} stencil, windows of {win[0]} rows x "
                          f"{win[1]} columns, {'complex64' if dom == 4 else el_h + (' (chosen on the device)' if armed else '')} scratch: column FFT, x "
                          f"{'real ' if real else ''}kernel spectrum, inverse column FFT, in place"
                          + ("; a resident grid walking the launch's pairs per column block" if walk else "") + ")",
                "bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g / HBM_PEAK_GBPS,
                "kernel_ms": tot_ms / launches, "launches_per_step": launches / steps_for_cols, "bytes_per_launch": bytes_alg / launches,
                "bytes_counted": "per window pair: the scratch image read + its rows that hold valid outputs written back; the kernel "
                                 "spectrum (registers / L2) is not counted",
                "share_of_step": tot_ms / steps_for_cols / ms_per_step,
                "concurrency": "two FFT-pass kernels usually share the GPU (two internal streams): kernel_ms and achieved are per launch "
                               "under that sharing; the event pair also spans the dispatch gap (~5 us)",
            }
            if dom == 1 and armed:
                dk["note"] = ("measured on eager stage calls that choose the scratch element from the same exposure-range record as the timed "
                              "steps' r2f_render: the same kernel instance on the same windows with the same element")
            if solo[dom][1]:
                ms, n, b = solo[dom]
                dk["alone"] = {"kernel_ms": ms / n, "achieved": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "note": "the same launches (twice the pairs each) with one internal stream, two extra steps"}
            if traffic_rec is not None:  # the same capture, this kernel's launches: L2 <-> fabric bytes per launch
                for name, v in traffic_rec.items():
                    if isinstance(v, dict) and kname.split("r2f::")[1] in name and "hbm_bytes_per_launch" in v:
                        dk["traffic"] = v["hbm_bytes_per_launch"]
                        dk["traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / dk["bytes_per_launch"]
            roof["largest_fft_pass" if "dominant_kernel" in roof else "dominant_kernel"] = dk
        # all FFT passes of both stencils over the two stages' wall time
        sb = (sum(extra[c][2] / 2 for c in (0, 2, 3, 5)) + sum(cols[c][2] / steps_for_cols for c in (1, 4))) / frames_here  # per frame
        sms = float(stage_ms["halation"]) + float(stage_ms.get("mtf", 0.0))
        roof["stencil_stages"] = {"scratch_and_window_bytes_per_frame": sb, "ms_per_frame": sms, "GB/s": sb / (sms * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": sb / (sms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                  "bytes_per_px": sb / ((r1 - r0) * W),
                                  "note": "what the three passes of halation + MTF have to move (window floats in, scratch written, read, "
                                          "written back, read, outputs out), single-tap plane included in the time"}
        # fp64 work of the FFT passes against the vector peak
        flops = insts = 0.0
        rows_here = r1 - r0
        for stats, kern in ((st_h, hal_k), (st_m, mtf_k)):
            for ch, c in enumerate(stats):
                if not c["fft"] or not c["window"]:
                    continue
                nz = np.nonzero(kern[..., ch if kern.shape[2] > 1 else 0])
                bh, bw = int(nz[0].max() - nz[0].min() + 1), int(nz[1].max() - nz[1].min() + 1)  # box of the non-zero taps
                ny, nx = c["window"]
                vy, vx = ny - bh + 1, (nx - bw + 1) & ~3
                pairs = (((W + vx - 1) // vx) * ((rows_here + vy - 1) // vy) + 1) // 2
                f, i = fft_fp64_ops(ny, nx, vy)
                flops += pairs * f
                insts += pairs * i
        if flops:
            roof["fp64_valu"] = {
                "flops_per_frame": flops, "tflops": flops / (sms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                "frac": flops / (sms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "issue_slot_frac": insts / (sms * 1e-3) / (FP64_PEAK_TFLOPS / 2 * 1e12),
                "note": "fp64 adds, multiplies and FMAs of the FFT passes counted from r2f_fft.hip (FMA = 2 flop) over the two stencil "
                        "stages' wall time; issue_slot_frac counts every fp64 instruction as one slot of the 39.3 T lane-instr/s the "
                        "vector units issue at 2.4 GHz (adds fill a slot with 1 flop)"}
        if args.direct_stencils:
            px = (r1 - r0) * W
            nnz = [int(np.count_nonzero(hal_k[..., c])) for c in range(3)]
            hal_ms = float(stage_ms["halation"])
            roof["direct_halation"] = {"tflops_nonzero_taps": 2.0 * sum(nnz) * px / (hal_ms * 1e-3) / 1e12, "fp32_peak": 157.3,
                                       "kernel_ms": hal_ms}

    if batch and world == 1 and not args.no_pcie:
        # BASELINE config 5 end to end on this GPU: BatchSharder with the two-phase API, host frames in (pinned fp32 HWC4 like
        # the reference's payload), uint8 frames back on the host -- PCIe-inclusive, never `value`.  Twice: frame after frame
        # (process_preloaded), and with one frame in flight (submit_preloaded + collect: both copy directions overlap the render)
        n_e2e = min(args.frames, 16)
        host4 = torch.cat([frame, torch.ones_like(frame[..., :1])], dim=-1).cpu().pin_memory()  # gpu_processor.py:765
        payload = {"image_array": host4, "output_resolution": (W, H), "canvas_resolution": None, "pipeline_resolution": (W, H)}
        host3 = frame.cpu().pin_memory()  # what extract_image_data_cpu hands over with HipProcessor(payload_alpha=False)
        payload3 = dict(payload, image_array=host3)
        # ... and as LibRaw's 16-bit output with the conversion of raw_conversion.py:50-52 on the device (r2f_decode_u16)
        host16 = (frame.clamp(0, 1) * 65535).to(torch.int32).to(torch.int16).cpu().pin_memory()  # the 16 bits of a uint16 frame
        payload16 = dict(payload, image_array=host16, u16_factor=1.0)
        # ... and the frame as an ordinary (pageable) NumPy array, which is what a decoder hands over
        payload3p = dict(payload, image_array=host3.numpy().copy())
        kw = dict(settings, seed=GRAIN_SEED, matrix=REC709_TO_XYZ)
        legs = {}
        for name, pay, execute, collect in (
                # (a frame is "exported" by looking at it and dropping it: holding every result would make each frame allocate
                # fresh pinned memory instead of reusing the previous frames' buffers)
                ("serial", payload, lambda t, pl: int(proc.process_preloaded(pl, neg, 6, 0.4, **kw)[0, 0, 0]), None),
                ("overlapped", payload, lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw), lambda t, h: int(h.result()[0, 0, 0])),
                ("overlapped_rgb", payload3, lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw), lambda t, h: int(h.result()[0, 0, 0])),
                ("overlapped_u16", payload16, lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw), lambda t, h: int(h.result()[0, 0, 0])),
                ("overlapped_rgb_pageable", payload3p, lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw), lambda t, h: int(h.result()[0, 0, 0]))):
            BatchSharder(0, 1).run([0, 1], lambda t: pay, execute, collect=collect)  # warm-up (pinned pools, streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res, skipped = BatchSharder(0, 1).run(list(range(n_e2e)), lambda t: pay, execute, collect=collect)
            torch.cuda.synchronize()
            legs[name] = H * W / 1e6 * len(res) / (time.perf_counter() - t0)
        result["pcie_inclusive"] = {
            "value": legs["overlapped"], "unit": "MP/s", "frames": n_e2e, "serial": legs["serial"],
            "without_alpha_plane": legs["overlapped_rgb"], "uint16_payload": legs["overlapped_u16"],
            "without_alpha_plane_pageable_source": legs["overlapped_rgb_pageable"],
            "note": "BatchSharder.run over the two-phase API: pinned fp32 HWC4 frame -> device, render, uint8 result -> pinned host "
                    "memory. value: one frame in flight while the next is submitted (upload, render and download on three streams); "
                    "serial: process_preloaded frame after frame.  Upload-bound (384 MB per 24 MP frame); without_alpha_plane: the same "
                    "with the (H, W, 3) payload of HipProcessor(payload_alpha=False), 288 MB per frame; uint16_payload: the decoded frame handed over as LibRaw's 16-bit output "
                    "(144 MB per frame), converted by r2f_decode_u16 on the device; without_alpha_plane_pageable_source: the (H, W, 3) payload as an "
                    "ordinary NumPy array instead of pinned memory (its copy blocks the submitting thread, so nothing of the host's "
                    "work overlaps it; until round 6 submit_preloaded pinned such a frame first: 373 MP/s).  Not part of `value`"}

    if not batch and world == 1 and not args.no_pcie:
        # host <-> device copies reported separately, never in `value`.  What the reference pays at the same boundary:
        # queue.write_texture of the float frame (gpu_processor.py:279-305) and read_texture of the uint8 result (:1311-1357).
        # Event-timed on the launch stream, best of 5, pinned host memory; then one end-to-end HipProcessor.process() on a host array.
        def best_ms(fn, n=5):
            best = float("inf")
            for _ in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            return best

        copies = {}
        host_f32 = torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True)
        host_f32.copy_(frame)
        dev_f32 = torch.empty_like(frame)
        copies["h2d_f32_ms"] = best_ms(lambda: dev_f32.copy_(host_f32, non_blocking=True))
        host_u16 = torch.empty((H, W, 3), dtype=torch.int16, pin_memory=True)  # the 16 bits of LibRaw's uint16 output
        host_u16.copy_((frame.clamp(0, 1) * 65535).to(torch.int32).to(torch.int16))
        dev_u16 = torch.empty((H, W, 3), dtype=torch.int16, device=frame.device)

        def up_u16():
            dev_u16.copy_(host_u16, non_blocking=True)
            proc.ctx.decode_u16(dev_u16, 1.0)  # raw_conversion.py:50-52 on the device (r2f_decode_u16)

        copies["h2d_u16_ms"] = best_ms(up_u16)
        dev_u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=frame.device)
        host_u8 = torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=True)
        copies["d2h_u8_ms"] = best_ms(lambda: host_u8.copy_(dev_u8, non_blocking=True))
        del dev_f32, dev_u16, dev_u8, host_u8
        # the drop-in call on a host array, as the GUI's export makes it: upload (pageable NumPy memory), render, uint8 download
        host_np = host_f32.numpy()

        def timed_calls(src, n, seed0, **kw):
            ts = []
            for i in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = proc.process(src, neg, 6, 0.4, cache=False, seed=GRAIN_SEED + seed0 + i, lens_correction=False, **settings, **kw)
                ts.append((time.perf_counter() - t0) * 1e3)
            assert r.dtype == np.uint8 and r.shape == (H, W, 3)
            return ts

        e2e = timed_calls(host_np, 6, 0)
        copies["process_end_to_end_ms"] = min(e2e)
        copies["process_end_to_end_first_ms"] = e2e[0]
        copies["process_end_to_end_all_ms"] = [round(x, 3) for x in e2e]
        bands0, proc.stream_bands = proc.stream_bands, 0
        copies["process_end_to_end_one_after_the_other_ms"] = min(timed_calls(host_np, 3, 50))
        proc.stream_bands = bands0
        pageable = np.array(host_np)  # the same frame in ordinary (pageable) host memory, like an array the GUI's decoder hands over
        copies["process_end_to_end_pageable_source_ms"] = min(timed_calls(pageable, 3, 60))
        del pageable
        # the same call with the result handed back as a view of a pinned buffer (HipProcessor(result_buffers=2): interactive use)
        # instead of a fresh pageable array like upstream's: the download then runs at the link's rate
        proc.result_buffers = 2
        e2e = []
        for i in range(8):  # (the uploaded frame alternates between two device blocks -- the previous one is still held while the
            #                   next arrives --, and r2f_render captures a buffer set the second time it comes by: steady state from
            #                   the fifth call on)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = proc.process(host_np, neg, 6, 0.4, cache=False, seed=GRAIN_SEED + 10 + i, lens_correction=False, **settings)
            e2e.append((time.perf_counter() - t0) * 1e3)
        copies["process_end_to_end_pinned_result_ms"] = min(e2e)
        copies["process_end_to_end_pinned_result_all_ms"] = [round(x, 3) for x in e2e]
        # ... which streams the frame through the pipeline in row bands while it arrives (HipProcessor._process_streamed: PCIe is full
        # duplex and the stage entry points are row-range calls); the same call with upload, render and download one after the other:
        bands, proc.stream_bands = proc.stream_bands, 0
        e2e = []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = proc.process(host_np, neg, 6, 0.4, cache=False, seed=GRAIN_SEED + 30 + i, lens_correction=False, **settings)
            e2e.append((time.perf_counter() - t0) * 1e3)
        copies["process_end_to_end_pinned_result_one_after_the_other_ms"] = min(e2e)
        copies["process_stream_bands"] = bands
        # ... and the hand-off RAW decoding really makes: LibRaw's uint16 frame (half the upload), converted on the device
        # (raw_conversion.py:50-52 = r2f_decode_u16) band by band as it arrives; exposure given in stops (no host pass over the frame)
        host_u16_np = host_u16.numpy().view(np.uint16)
        for name, b in (("process_u16_end_to_end_pinned_result_ms", bands), ("process_u16_end_to_end_pinned_result_one_after_the_other_ms", 0)):
            proc.stream_bands = b
            e2e = []
            for i in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = proc.process(host_u16_np, neg, 6, 0.4, cache=False, exposure=0.0, seed=GRAIN_SEED + 40 + i, lens_correction=False, **settings)
                e2e.append((time.perf_counter() - t0) * 1e3)
            copies[name] = min(e2e[2:])
        proc.stream_bands = 0
        # where that call's time goes: one more call with a device synchronisation behind every stage
        # (HipProcessor.profile_stages -- a measuring mode: its total is a little above the un-profiled call's)
        proc.profile_stages = True
        proc.process(host_np, neg, 6, 0.4, cache=False, seed=GRAIN_SEED + 20, lens_correction=False, **settings)
        proc.profile_stages = False
        proc.stream_bands = bands
        copies["process_stage_ms"] = {k: round(float(v), 3) for k, v in proc.last_stage_ms.items()}
        copies["process_stage_note"] = ("the one-after-the-other path of HipProcessor.process(host ndarray, cache=False, result_buffers=2) with a synchronisation behind each stage: "
                                        "host_phase = extract_image_data_cpu (views and index arithmetic; cache=False skips the 32-row checksum "
                                        "that cost ~4 ms in round 5), upload_and_device_prepath = the fp32 frame over PCIe + the clamp of "
                                        "gpu_processor.py:275 on the device, prepare_and_render = table checks + r2f_render, download = uint8 "
                                        "result into a pinned buffer; load_and_upload = the first two together")
        proc.result_buffers = 0
        # ... and what a slider step costs: a preview re-render through the same call -- the frame already on the device (cache=True),
        # one film setting changed, a 1500 x 1000 preview back on the host
        pv = dict(settings, resolution=(1000, 1500), lens_correction=False)
        proc.process(host_np, neg, 6, 0.4, seed=GRAIN_SEED, **pv)
        proc.process(host_np, neg, 6, 0.4, seed=GRAIN_SEED, exp_comp=0.01, **pv)
        ts = []
        for i in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            proc.process(host_np, neg, 6, 0.4, seed=GRAIN_SEED, exp_comp=0.02 + 0.01 * i, **pv)
            ts.append((time.perf_counter() - t0) * 1e3)
        copies["preview_rerender_ms"] = sorted(ts)[len(ts) // 2]
        copies["preview_rerender_note"] = ("HipProcessor.process(host ndarray, resolution=(1000, 1500)) with the frame already on the device and "
                                           "one film setting changed per call, median of 12: table rebuild + a 1.5 MP render + its download + the "
                                           "source array's fingerprint (a 192 KB sample since round 6: 0.14 ms; all of 32 rows before: 2.4 ms of a "
                                           "2.76 ms call at 100 MP)")
        gb = H * W * 3 / 1e9
        copies["GB_per_s"] = {"h2d_f32": 4 * gb / (copies["h2d_f32_ms"] * 1e-3), "h2d_u16": 2 * gb / (copies["h2d_u16_ms"] * 1e-3),
                              "d2h_u8": gb / (copies["d2h_u8_ms"] * 1e-3)}
        copies["note"] = ("SURVEY 8(d): host <-> device copies of this frame, never part of `value`.  h2d_f32: pinned fp32 HWC3 frame -> device "
                          "(the reference's write_texture, gpu_processor.py:279-305); h2d_u16: pinned uint16 frame -> device + r2f_decode_u16 "
                          "(raw_conversion.py:50-52 on the device); d2h_u8: the uint8 result -> pinned host (read_texture, :1311-1357); events on "
                          "the launch stream, best of 5.  process_end_to_end: wall clock of HipProcessor.process(host ndarray, cache=False) -> "
                          "uint8 ndarray of the caller's own, with NO option set (first call listed too: it builds tables and pinned buffers): the frame "
                          "streams through the pipeline in row bands while it arrives, into one of up to three pinned buffers the processor lends "
                          "out and gets back when the caller drops the array (a caller that keeps more gets freshly allocated arrays, filled by "
                          "helper threads band by band: 38 ms); ..._one_after_the_other: stream_bands = 0 (upload 21 + render 5 + download 5; until "
                          "round 5 the download was a pageable one into a fresh array, 33 ms: upstream's sequence); ..._pageable_source: the "
                          "source array in ordinary host memory instead of pinned memory; process_end_to_end_pinned_result: the "
                          "same with result_buffers = 2 (a view of a pinned buffer comes back), which also lets the frame stream through the "
                          "pipeline in process_stream_bands row bands while it arrives -- upload of band k + 2, render of band k + 1, download "
                          "of band k at the same time; ..._one_after_the_other: the same call with stream_bands = 0; process_u16_...: the same two calls "
                          "on LibRaw's uint16 frame (half the upload, exposure in stops, converted on the device)")
        result["host_device_copies"] = copies
        del host_f32, host_np, host_u16, host_u16_np

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import oracle_inputs  # oracle as the timed CPU baseline only
        from oracle import baseline

        p = oracle_inputs(neg, prt, scale, halation=effects, mtf=effects, grain=2 if effects else 0, seed=GRAIN_SEED)
        result["cpu_baseline"] = baseline.time_cpu_baseline(p, target_seconds=args.cpu_seconds)

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()
    proc.close()


if __name__ == "__main__":
    main()
