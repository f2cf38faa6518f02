"""RowShardedRenderer with the HIP stage backend.  One GPU is enough: world_size 2 over gloo with both
ranks on cuda:0 (RCCL refuses two ranks on one device; the exchange code path is the same
`batch_isend_irecv`, only the transport differs)."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from helpers import SEED, stocks, synthetic_frame  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render(rank, world, H, W, fw, burn=0.0):
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    neg, prt, _ = stocks()
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0,
                          highlight_burn=burn, burn_scale=20)
    scale = max(H, W) / fw
    hal = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
    mtf = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    be = HipStageBackend(proc.ctx, params, stencils.vertical_reach(hal), stencils.vertical_reach(mtf))
    rr = RowShardedRenderer(be, H, W, halation=True, mtf=True, burn=bool(burn), rank=rank, world=world)
    frame = synthetic_frame(H, W, seed=31)
    frame[60:150, 40:200] *= 8.0
    img = torch.from_numpy(frame).cuda()
    out = torch.empty((rr.plan.rows, W, 3), dtype=torch.float32, device="cuda")
    rr.render(img[rr.plan.r0:rr.plan.r1].contiguous(), out_f32=out)
    torch.cuda.synchronize()
    return out.cpu().numpy(), proc, params, img


def _worker(rank, world, port, H, W, fw, path, burn=0.0):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out, _, _, _ = _render(rank, world, H, W, fw, burn)
        np.save(f"{path}.{rank}.npy", out)
    finally:
        dist.destroy_process_group()


def test_two_rank_hip_row_shards_bit_identical_to_single_gpu(tmp_path):
    import torch.multiprocessing as mp

    H, W, fw = 210, 256, 1.0  # 256 px/mm -> 65-tap halation, 27-tap MTF; shards of 105 rows
    path = str(tmp_path / "shard")
    mp.spawn(_worker, args=(2, _free_port(), H, W, fw, path), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    whole, proc, params, img = _render(0, 1, H, W, fw)
    np.testing.assert_array_equal(sharded, whole)
    ref, _ = proc.ctx.render(img, params)
    np.testing.assert_array_equal(whole, ref.cpu().numpy())


def test_two_rank_hip_row_shards_with_highlight_burn(tmp_path):
    """S7 adds one all-reduce of the low-res cell sums; partial sums are added in a different order than on one
    GPU, so the result agrees to rounding instead of bit for bit."""
    import torch.multiprocessing as mp

    H, W, fw = 210, 256, 1.0
    path = str(tmp_path / "shard")
    mp.spawn(_worker, args=(2, _free_port(), H, W, fw, path, 0.8), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    whole, proc, params, img = _render(0, 1, H, W, fw, 0.8)
    assert params.flags & 32
    np.testing.assert_allclose(sharded, whole, rtol=0, atol=2e-6)
    ref, _ = proc.ctx.render(img, params)
    np.testing.assert_allclose(whole, ref.cpu().numpy(), rtol=0, atol=1e-6)


def test_timed_backend_reports_every_stage():
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer
    from raw2film_amd.tracing import TimedBackend

    neg, prt, _ = stocks()
    H, W = 128, 192
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=2.0,
                          frame_height=2.0 * H / W, highlight_burn=0.5)
    be = TimedBackend(HipStageBackend(proc.ctx, params, (23, 23), (9, 9)))
    rr = RowShardedRenderer(be, H, W, halation=True, mtf=True, burn=True, rank=0, world=1)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    rr.render(torch.from_numpy(synthetic_frame(H, W, seed=3)).cuda(), out_f32=out)
    ms = be.summary()
    assert set(ms) == {"front", "halation", "mtf", "grain", "burn_sums", "burn_map", "tail"}
    assert all(v > 0 for v in ms.values())
    proc.close()


def test_batch_export_through_the_two_phase_api():
    """Config 5's shape in small: frames dealt to ranks, host phase one frame ahead on a producer thread, a frame whose host
    phase fails is skipped (gui_objects.py:65-115) -- here with the real HipProcessor on one rank."""
    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd.sharding import BatchSharder

    stocks_ = filmstock.builtin_stocks()
    neg, prt = stocks_["Kodak Portra 400"], stocks_["Kodak 2383"]
    proc = HipProcessor(device=0)
    try:
        frames = [synthetic_frame(96 + 8 * i, 144 + 12 * i, seed=60 + i) for i in range(5)]
        tasks = [dict(src=f, seed=100 + i) for i, f in enumerate(frames)]
        tasks.insert(2, dict(src="missing_frame.cr3", seed=0))  # RAW decoding is not ours: the host phase raises, the frame is skipped
        kw = dict(print_film=prt, exp_kelvin=6000, color_masking=1.0, chroma_nr=1)

        def prepare(t):
            return proc.extract_image_data_cpu(t["src"], **kw)

        def execute(t, payload):
            return proc.process_preloaded(payload, neg, 6, 0.4, seed=t["seed"], **kw)

        results, skipped = BatchSharder(0, 1).run(tasks, prepare, execute)
        assert skipped == [2] and sorted(results) == [0, 1, 3, 4, 5]
        for idx, out in results.items():
            t = tasks[idx]
            direct = proc.process(t["src"], neg, 6, 0.4, seed=t["seed"], **kw)
            assert out.dtype == np.uint8 and np.array_equal(out, direct)
        # two ranks split the same list without overlap
        mine0 = [i for i, _ in BatchSharder(0, 2).my_tasks(tasks)]
        mine1 = [i for i, _ in BatchSharder(1, 2).my_tasks(tasks)]
        assert sorted(mine0 + mine1) == list(range(6)) and not set(mine0) & set(mine1)
    finally:
        proc.close()
