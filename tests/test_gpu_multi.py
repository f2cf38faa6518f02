"""The multi-GPU layers on REAL devices: two ranks over `nccl` (= RCCL over xGMI), one GPU each.  Skipped on a box with a single
GPU (tests/test_gpu_sharding.py covers the same code there over gloo with both ranks on cuda:0); the first 2-GPU box that runs
`pytest -m gpu` exercises RCCL's send/recv pairs, the checksum equality of bench.py's N = 2 line and one processor per device.
SURVEY.md 8(e); batch semantics: gui_objects.py:65-115."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import SEED, stocks, synthetic_frame  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _render_rows(rank, world, device, H, W, fw, direct):
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    neg, prt, _ = stocks()
    proc = HipProcessor(device=device)
    if direct:
        proc.ctx.set_option("stencil_fft", 0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    scale = max(H, W) / fw
    hal = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
    mtf = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
    rr = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, graph=world > 1)
    frame = synthetic_frame(H, W, seed=31)
    frame[60:150, 40:200] *= 8.0
    img = torch.from_numpy(frame[rr.plan.r0:rr.plan.r1]).to(f"cuda:{device}").contiguous()
    out = torch.empty((rr.plan.rows, W, 3), dtype=torch.float32, device=img.device)
    while rr.tuning:  # (world > 1: the first frames measure the candidate schedules over RCCL; every one of them is a correct frame)
        out.zero_()
        rr.render(img, out_f32=out)
    for _ in range(3 if world > 1 else 1):  # eager, capture + replay, replay: all three have to give the same rows
        out.zero_()
        rr.render(img, out_f32=out)
    torch.cuda.synchronize(img.device)
    res = out.cpu().numpy()
    proc.close()
    return res


def _nccl_worker(rank, world, port, H, W, fw, direct, path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        ones = torch.ones(1, device=f"cuda:{rank}")
        dist.all_reduce(ones)
        assert int(ones.item()) == world
        np.save(f"{path}.{rank}.npy", _render_rows(rank, world, rank, H, W, fw, direct))
    finally:
        dist.destroy_process_group()


@needs_two
@pytest.mark.parametrize("direct", [True, False], ids=["direct", "fft"])
def test_two_ranks_over_rccl_reproduce_the_single_gpu_frame(tmp_path, direct):
    """Direct stencils: bit-identical (tile-independent tap order).  FFT stencils: a shard anchors its windows at its own first
    row, so a handful of pixels may differ by one fp32 ulp of the density -- asserted as <= 2e-6 relative on the output."""
    import torch.multiprocessing as mp

    H, W, fw = 420, 512, 2.0  # 256 px/mm: 65-tap halation, 27-tap MTF; shards of 210 rows
    path = str(tmp_path / "shard")
    mp.spawn(_nccl_worker, args=(2, _free_port(), H, W, fw, direct, path), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    whole = _render_rows(0, 1, 0, H, W, fw, direct)
    if direct:
        np.testing.assert_array_equal(sharded, whole)
    else:
        err = np.max(np.abs(sharded - whole) / np.maximum(np.abs(whole), 1e-3))
        assert err <= 2e-6, err
        assert np.mean(sharded != whole) < 1e-3


def _bench(args, timeout=1200):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env,
                         cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])


@needs_two
def test_bench_two_gpus_launches_itself_and_matches_the_one_gpu_checksum():
    """`python bench.py --gpus 2` (no launcher): bench.py starts its two ranks as a child, RCCL carries the halo exchange, and with
    the direct stencils the sharded frame's checksum equals the single-GPU frame's."""
    common = ["--config", "cfg3_45mp", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alone", "--checksum", "--direct-stencils"]
    two = _bench(["--gpus", "2"] + common)
    one = _bench(["--gpus", "1"] + common)
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["backend"] == "nccl" and two["scaling"] == "strong"
    assert two["checksum"] == one["checksum"]
    assert two["roofline"]["peak"] == 2 * one["roofline"]["peak"]
    assert two["value"] > 0 and two["ms_per_step_min"] <= two["ms_per_step_median"] <= two["ms_per_step_max"]


@needs_two
def test_batch_sharder_with_one_processor_per_device():
    """Config 5's shape: frame i -> GPU i mod 2, one HipProcessor (context, stream, tables) per device, no collectives; every frame
    equals the same frame rendered on GPU 0 alone, byte for byte."""
    import threading

    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd.sharding import BatchSharder

    stocks_ = filmstock.builtin_stocks()
    neg, prt = stocks_["Kodak Portra 400"], stocks_["Kodak 2383"]
    kw = dict(print_film=prt, exp_kelvin=6000, color_masking=1.0)
    frames = [synthetic_frame(160 + 8 * i, 240 + 12 * i, seed=70 + i) for i in range(6)]
    tasks = [dict(src=f, seed=200 + i) for i, f in enumerate(frames)]
    procs = [HipProcessor(device=d) for d in range(2)]
    results = [None, None]

    def run(rank):
        proc = procs[rank]
        results[rank] = BatchSharder(rank, 2).run(
            tasks, lambda t: proc.extract_image_data_cpu(t["src"], **kw),
            lambda t, payload: proc.process_preloaded(payload, neg, 6, 0.4, seed=t["seed"], **kw))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    merged = {}
    for res, skipped in results:
        assert skipped == []
        merged.update(res)
    assert sorted(merged) == list(range(6))
    for i, t in enumerate(tasks):
        ref = procs[0].process(t["src"], neg, 6, 0.4, seed=t["seed"], **kw)
        assert np.array_equal(merged[i], ref), i
    for p in procs:
        p.close()


def test_a_launcherless_multi_gpu_bench_fails_for_lack_of_devices_not_for_usage():
    """On a box with fewer GPUs than ranks `python bench.py --gpus N` must get as far as its ranks: the launch is bench.py's own
    job, and what stops it is the missing device."""
    n = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--config",
                          "cfg2_24mp", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode != 0
    assert "must be launched with" not in res.stderr
    assert f"needs GPU {n - 1}" in res.stderr, res.stderr[-2000:]
