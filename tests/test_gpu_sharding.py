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


def _render(rank, world, H, W, fw, burn=0.0, direct=False, dyn=False):
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    neg, prt, _ = stocks()
    proc = HipProcessor(device=0)
    if direct:
        proc.ctx.set_option("stencil_fft", 0)  # tile-independent tap order: shards and whole frame agree bit for bit
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0,
                          highlight_burn=burn, burn_scale=20)
    scale = max(H, W) / fw
    hal = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
    mtf = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
    # dyn = False: complex128 scratch for the halation whatever the rows hold -- like for like with a whole-frame render under
    # stencil_fft_scratch96_auto = 0, which is what the bit-identity tests below compare; dyn = True is the product default
    if dyn:
        proc.ctx.set_option("stencil_fft_window_rows", 256)  # (the device-side choice exists for the 256-row passes large frames take)
    rr = RowShardedRenderer(be, H, W, halation=True, mtf=True, burn=bool(burn), rank=rank, world=world, dyn_scratch=dyn)
    frame = synthetic_frame(H, W, seed=31)
    frame[60:150, 40:200] *= 8.0
    if dyn:  # a frame whose range the 12-byte element's guard accepts (max / shadow <= 6e4 with the stand-in Portra curve)
        frame = np.clip(frame, 8e-3, None)
    img = torch.from_numpy(frame).cuda()
    out = torch.empty((rr.plan.rows, W, 3), dtype=torch.float32, device="cuda")
    rr.render(img[rr.plan.r0:rr.plan.r1].contiguous(), out_f32=out)
    torch.cuda.synchronize()
    return out.cpu().numpy(), proc, params, img


def _worker(rank, world, port, H, W, fw, path, burn=0.0, direct=False, dyn=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # no hostname lookup (the box's name may not resolve)
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out, proc, _, _ = _render(rank, world, H, W, fw, burn, direct, dyn)
        np.save(f"{path}.{rank}.npy", out)
        if dyn:
            rng = proc.ctx.frame_exposure_range()
            np.save(f"{path}.{rank}.packed.npy", np.array([rng["armed"], rng["twelve_byte_element"], rng["min"], rng["max_abs"]], dtype=np.float64))
    finally:
        dist.destroy_process_group()


def _worker_graph(rank, world, port, H, W, fw, path):
    """Four frames with four seeds through a graph-replaying renderer and through an eager one (same rank, same buffers)."""
    import torch.distributed as dist

    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        neg, prt, _ = stocks()
        proc = HipProcessor(device=0)
        params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                              frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
        scale = max(H, W) / fw
        hal, mtf = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3), stencils.mtf_stencil(neg, scale, 0.0, 1.0)
        be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
        eager = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, split_halation=True, exchanges=1)
        graphed = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, graph=True, split_halation=True, exchanges=1)
        assert graphed.graph and graphed.split is not None
        # per plane: the blue layer (a single halation tap) travels with the MTF's halo only
        assert graphed.halo_e_ch[2] == (be.mtf_taps[0], be.mtf_taps[1]) and graphed.halo_e_ch[0][0] == be.halation_taps[0] + be.mtf_taps[0]
        r0, r1 = graphed.plan.r0, graphed.plan.r1
        img = torch.from_numpy(synthetic_frame(H, W, seed=31)).cuda()[r0:r1].contiguous()
        out_e = torch.empty((r1 - r0, W, 3), dtype=torch.float32, device="cuda")
        out_g = torch.empty_like(out_e)
        frames = []
        for k, seed in enumerate((SEED, 7, 0xFFFFFFFF, 12345)):
            eager.render(img, out_f32=out_e, seed=seed)
            graphed.trace = []
            graphed.render(img, out_f32=out_g, seed=seed)
            assert torch.equal(out_g, out_e), (rank, k)
            if k >= 1:  # captured on the second frame, replayed from then on -- a new seed costs no graph
                # (the received halo rows join the exposure-range record INSIDE the graph downstream of the exchange, ahead of the
                # halation launches that choose their scratch element from it)
                assert graphed.trace == ["exchange_start", "replay:halation_interior", "exchange_finish", "replay:after_exchange"] or \
                    graphed.trace[:1] == ["exchange_start"] and graphed.trace[-3:] == ["replay:halation_interior", "exchange_finish", "replay:after_exchange"], graphed.trace
            frames.append(out_g.cpu().numpy().copy())
        assert [sum(g is not None for g in v[1].values()) for v in graphed._graphs.values() if v[1] is not None] == [2]
        assert not np.array_equal(frames[0], frames[1])  # the seed reaches the grain
        np.save(f"{path}.{rank}.npy", frames[1])
        proc.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_graph_replay_with_a_new_seed_per_frame_and_the_interior_halation_ahead_of_the_exchange(tmp_path):
    """world = 2 over gloo on one GPU, graph = True: the interior halation is one captured graph replayed BEFORE the exchange is
    waited for (it overlaps the host-staged gloo transfer here, RCCL's own stream on a real node), everything downstream a
    second one; the per-frame seed lives in the device-side frame block, so four seeds replay the same two graphs.  Each frame
    equals the eager launches bit for bit, and the sharded frame the single-GPU one to an fp32 ulp (FFT windows are anchored
    per call)."""
    import torch.multiprocessing as mp

    H, W, fw = 1400, 512, 1400 / 341.33  # 341 px/mm: 87-tap halation (85 non-zero) and 35-tap MTF by FFT; shards of 700 rows, one band each
    path = str(tmp_path / "graph")
    mp.spawn(_worker_graph, args=(2, _free_port(), H, W, fw, path), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    from raw2film_amd import HipProcessor
    from raw2film_amd.hip_processor import REC709_TO_XYZ

    neg, prt, _ = stocks()
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=7, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    whole, _ = proc.ctx.render(torch.from_numpy(synthetic_frame(H, W, seed=31)).cuda(), params)
    whole = whole.cpu().numpy()
    assert np.max(np.abs(sharded - whole) / np.maximum(np.abs(whole), 1e-3)) <= 2e-6
    proc.close()


def _worker_schedules(rank, world, port, H, W, fw, path):
    """The two-exchange schedule eager and under graph replay, and a renderer that measures its schedule on its first frames."""
    import torch.distributed as dist

    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        neg, prt, _ = stocks()
        proc = HipProcessor(device=0)
        params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                              frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
        scale = max(H, W) / fw
        hal, mtf = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3), stencils.mtf_stencil(neg, scale, 0.0, 1.0)
        be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
        eager = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, exchanges=2)
        graphed = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, graph=True, exchanges=2)
        assert graphed.graph and graphed.schedule == (2, False) and not graphed.tuning
        ha, ma = be.halation_taps[0], be.mtf_taps[0]
        assert graphed.halo_e_ch[0] == (ha, ha) and graphed.halo_e_ch[2] == (0, 0) and graphed.plan.halo_d == (ma, ma)
        r0, r1 = graphed.plan.r0, graphed.plan.r1
        img = torch.from_numpy(synthetic_frame(H, W, seed=31)).cuda()[r0:r1].contiguous()
        out_e = torch.empty((r1 - r0, W, 3), dtype=torch.float32, device="cuda")
        out_g = torch.empty_like(out_e)
        for k, seed in enumerate((SEED, 7, 0xFFFFFFFF, 12345)):
            eager.render(img, out_f32=out_e, seed=seed)
            graphed.trace = []
            graphed.render(img, out_f32=out_g, seed=seed)
            assert torch.equal(out_g, out_e), (rank, k)
            if k >= 1:
                assert graphed.trace[-5:] == ["exchange_start", "exchange_finish", "replay:density", "exchange_density", "replay:finish"] or \
                    graphed.trace[-4:] == ["exchange_finish", "replay:density", "exchange_density", "replay:finish"], graphed.trace
            if k == 1:
                np.save(f"{path}.two.{rank}.npy", out_g.cpu().numpy())
        # measured schedule: every frame of the measuring phase is a correct frame; afterwards the choice is one of the candidates,
        # the same on both ranks, and the frames replay graphs
        auto = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=rank, world=world, graph=True)
        assert auto.tuning and len(auto._candidates) >= 2
        frames = 0
        while auto.tuning:
            auto.render(img, out_f32=out_g, seed=7)
            frames += 1
            assert float(((out_g - out_e.new_tensor(np.load(f"{path}.two.{rank}.npy"))).abs()
                          / out_e.new_tensor(np.load(f"{path}.two.{rank}.npy")).abs().clamp_min(1e-3)).max()) <= 2e-6, frames
        assert frames == (1 + auto.tune_frames) * len(auto._candidates)  # (one untimed frame ahead of every candidate's timed ones)
        assert auto.schedule[0] in (1, 2) and len(auto.tuned_ms) == len(auto._candidates) and all(t > 0 for t in auto.tuned_ms)
        chosen = torch.tensor([auto.schedule[0] * 2 + int(auto.schedule[1])])
        both = [torch.zeros_like(chosen) for _ in range(world)]
        dist.all_gather(both, chosen)
        assert all(int(b) == int(chosen) or int(b) // 2 == int(chosen) // 2 for b in both)  # (the split itself is rank-local)
        for seed in (1, 2, 3):
            auto.render(img, out_f32=out_g, seed=seed)
        assert any(v[1] is not None for v in auto._graphs.values())
        proc.close()
    finally:
        dist.destroy_process_group()


def test_two_exchange_schedule_and_the_measured_choice(tmp_path):
    """Round 5.  `exchanges = 2`: exposure halo for the halation only, halation on exactly the own rows (one
    FFT window row fewer per 1/8 shard of the 100 MP frame), then the MTF's density halo -- eager and as two captured graphs
    either side of the second exchange, bit for bit; the sharded frame equals the whole one to the FFT form's rounding.  With
    `exchanges` / `split_halation` left on "auto" the renderer TIMES its candidate schedules on its first frames (the modelled
    link constant of round 4 is gone) and the ranks agree on one."""
    import torch.multiprocessing as mp

    H, W, fw = 1400, 512, 1400 / 341.33
    path = str(tmp_path / "sched")
    mp.spawn(_worker_schedules, args=(2, _free_port(), H, W, fw, path), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.two.{r}.npy") for r in range(2)])
    from raw2film_amd import HipProcessor
    from raw2film_amd.hip_processor import REC709_TO_XYZ

    neg, prt, _ = stocks()
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=7, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    whole, _ = proc.ctx.render(torch.from_numpy(synthetic_frame(H, W, seed=31)).cuda(), params)
    whole = whole.cpu().numpy()
    assert np.max(np.abs(sharded - whole) / np.maximum(np.abs(whole), 1e-3)) <= 2e-6
    proc.close()


def test_two_rank_hip_row_shards_bit_identical_to_single_gpu(tmp_path):
    import torch.multiprocessing as mp

    H, W, fw = 210, 256, 1.0  # 256 px/mm -> 65-tap halation, 27-tap MTF; shards of 105 rows
    path = str(tmp_path / "shard")
    mp.spawn(_worker, args=(2, _free_port(), H, W, fw, path), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    whole, proc, params, img = _render(0, 1, H, W, fw)
    np.testing.assert_array_equal(sharded, whole)
    proc.ctx.set_option("stencil_fft_scratch96_auto", 0)  # (the stage entry points keep complex128 for the halation: like for like)
    ref, _ = proc.ctx.render(img, params)
    np.testing.assert_array_equal(whole, ref.cpu().numpy())
    proc.ctx.set_option("stencil_fft_scratch96_auto", 1)
    ref12, _ = proc.ctx.render(img, params)  # ... and r2f_render's own choice agrees to the 12-byte element's rounding
    assert np.max(np.abs(ref12.cpu().numpy() - whole) / np.maximum(np.abs(whole), 1e-3)) <= 2e-6


def test_row_shards_choose_the_halation_scratch_element_like_the_whole_frame(tmp_path):
    """Round 6: the row-sharded renderer IS the headline renderer.  Its front calls record the range of the
    exposure rows they write, the received halo rows are added by r2f_stage_exposure_range, and the halation calls vouch for the
    record (R2F_F_RANGE_VALID) -- so the FFT passes choose their scratch element on the device exactly like r2f_render's.
    One rank: the same kernels on the same windows with the same element -- bit-identical to r2f_render's default frame.
    Two ranks: each chooses from ITS rows' range (rank-local: the bound is per window); the frames agree to the element's rounding."""
    import torch.multiprocessing as mp

    H, W, fw = 420, 256, 1.0  # 65-tap halation, 27-tap MTF; shards of 210 rows
    whole, proc, params, img = _render(0, 1, H, W, fw, dyn=True)
    rng = proc.ctx.frame_exposure_range()
    assert rng["armed"] and rng["twelve_byte_element"], rng
    ref, _ = proc.ctx.render(img, params)  # r2f_render's own choice
    assert proc.ctx.frame_exposure_range()["twelve_byte_element"]
    np.testing.assert_array_equal(whole, ref.cpu().numpy())
    proc.ctx.set_option("stencil_fft_scratch96_auto", 0)
    exact, _ = proc.ctx.render(img, params)
    assert not np.array_equal(exact.cpu().numpy(), whole)  # (it is the other element)
    proc.close()
    path = str(tmp_path / "shard")
    mp.spawn(_worker, args=(2, _free_port(), H, W, fw, path, 0.0, False, True), nprocs=2, join=True)
    sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(2)])
    assert np.max(np.abs(sharded - whole) / np.maximum(np.abs(whole), 1e-3)) <= 2e-6
    for r in range(2):
        armed, packed, lo, hi = np.load(f"{path}.{r}.packed.npy")
        assert armed and packed, (r, lo, hi)
        assert lo >= rng["min"] and hi <= rng["max_abs"]  # a rank's record is a sub-range of the frame's


def test_a_rank_vouches_for_exactly_the_rows_its_halation_reads_this_frame():
    """The row-sharded renderer's side of the per-window-pair choice, checked without numerics (tools/scratch_choice_model.py): every
    rank of a 3-rank world, on each of its three schedules, renders a sequence of hostile frames (extremes anywhere, halo rows
    included; graphs captured and replayed on the way) with the halo exchange looped back from the whole frame's exposure planes --
    and after every frame the flags its last vouched halation call left are held against a host model of the samples that call's
    windows hold: own rows as the front calls wrote them, halo rows as they arrived THIS frame, nothing of the rows the buffer keeps
    from an earlier frame (under two exchanges the planes are wider than what travels).  A pair may take the 12-byte element only
    if its own samples allow it."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hostile
    import scratch_choice_model as scm
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    class Loopback(RowShardedRenderer):
        """Neighbours that answer at once: the exposure halo rows come from the whole frame's planes (full_E); the density halo of
        the two-exchange schedule does not travel (its rows feed outputs nobody looks at here)."""
        full_E = None

        def _exchange(self, buf, buf_gy0, above, below, wait=True):
            p = self.plan
            if buf is not self.E:
                return None
            above = [above] * 3 if isinstance(above, int) else list(above)
            below = [below] * 3 if isinstance(below, int) else list(below)
            for c in range(3):
                if p.rank > 0 and above[c]:
                    buf[c, p.r0 - above[c] - buf_gy0:p.r0 - buf_gy0].copy_(self.full_E[c, p.r0 - above[c]:p.r0])
                if p.rank < p.world - 1 and below[c]:
                    buf[c, p.r1 - buf_gy0:p.r1 + below[c] - buf_gy0].copy_(self.full_E[c, p.r1:p.r1 + below[c]])
            return None

    neg, prt, _ = stocks()
    # 420 px/mm: the halation reaches 52 rows; three shards of 310 rows -- the middle rank's upper halo (rows 258 .. 309) lies inside the
    # range tile its own first rows belong to (rows 256 .. 319): a tile the record knows whether or not the halo rows were added
    H, W, fw = 930, 1300, 3.1
    proc = HipProcessor(device=0)
    proc.ctx.set_option("stencil_fft_window_rows", 256)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    scale = max(H, W) / fw
    hal = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3)
    mtf = stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    geo = scm.geometry(hal, (0, 1))
    rng = np.random.default_rng(20261005)
    frames = [scm.random_frame(rng, H, W, hostile) for _ in range(4)]
    frames = [np.nan_to_num(f, nan=1.0) for f in frames]  # (NaN pixels have their own test; here they would blank whole LUT cells)
    # ... and the two frames this is about: shadows everywhere / shadows with a bright band of 30 rows next to every shard boundary
    # (in the halo its neighbour sends the middle rank, outside that rank's own rows): rendered alternately, so that what a buffer keeps
    # from the frame before is always the opposite of what this frame holds
    dark = (1e-3 * rng.uniform(1.0, 3.0, (H, W, 3))).astype(np.float32)
    band = dark.copy()
    for b0, b1 in ((280, 310), (620, 650)):  # (either side of the middle rank's rows [310, 620): in its halo, not in its own rows)
        band[b0:b1] = (1e3 * rng.uniform(0.5, 1.0, (b1 - b0, W, 3))).astype(np.float32)
    frames += [dark, band]
    E_full = torch.empty((3, H, W), dtype=torch.float32, device="cuda")

    def run(cls, rank, kw, order):
        be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
        calls = []
        inner = be.halation

        def recording(E, e_gy0, D, d_gy0, y0, y1, Hh, identity_done=0, range_valid=False):
            calls.append((int(e_gy0), int(E.shape[1]), int(y0), int(y1), bool(range_valid)))
            return inner(E, e_gy0, D, d_gy0, y0, y1, Hh, identity_done=identity_done, range_valid=range_valid)

        be.halation = recording
        rr = cls(be, H, W, halation=True, mtf=True, rank=rank, world=3, graph=True, dyn_scratch=True, **kw)
        out = torch.empty((rr.plan.rows, W, 3), dtype=torch.float32, device="cuda")
        pairs = packed = allowed = 0
        bad = []
        for i in order:  # (the graphs are captured on the first frames and replayed on the others)
            img = torch.from_numpy(frames[i]).cuda()
            proc.ctx.stage_front(img, params, 0, dst=E_full)
            rr.full_E = E_full
            rr.render(img[rr.plan.r0:rr.plan.r1].contiguous(), out_f32=out, seed=i)
            info = proc.ctx.frame_exposure_range()
            flags = proc.ctx.frame_scratch_flags()
            vouched = [c for c in calls if c[4]]
            assert info["armed"] and vouched, (rank, kw, info)
            e_gy0, e_rows, y0, y1, _ = vouched[-1]
            ny, nx = proc.ctx.stencil_stats(0)[0]["window"]
            assert ny == 256
            m = scm.model(E_full.cpu().numpy(), geo, y0, y1, H, W, e_gy0, e_gy0 + e_rows, info["bound"], info["floor"], (0, 1), NY=ny, NX=nx)
            assert len(m) == len(flags), (rank, kw, len(m), len(flags))
            bad += [(rank, kw, i, y0, y1, e_gy0, e_rows, pc, m[pc]) for pc in range(len(m)) if flags[pc] and not m[pc][0]]
            if os.environ.get("R2F_DEBUG_SHARD_MODEL"):
                print(cls.__name__, rank, kw, "frame", i, "call", (y0, y1), "buffer", (e_gy0, e_gy0 + e_rows), "window", (ny, nx), "flags", flags.tolist(), "model", [(a, f"{lo:.2g}", f"{hi:.2g}") for a, lo, hi in m])
            pairs, packed, allowed = pairs + len(m), packed + int(flags.sum()), allowed + sum(a for a, _, _ in m)
        return pairs, packed, allowed, bad

    tot = np.zeros(3, dtype=np.int64)
    for rank in range(3):
        for kw in (dict(exchanges=1, split_halation=False), dict(exchanges=1, split_halation=True), dict(exchanges=2)):
            n, k, a, bad = run(Loopback, rank, kw, (0, 5, 4, 5, 1, 4, 5, 2, 4, 3))
            assert not bad, bad[:3]
            tot += (n, k, a)
    print(f"row shards against the host model: {tot[0]} pairs, {tot[1]} took the element, {tot[2]} allowed by their own samples")
    assert tot[0] > 400 and 0 < tot[1] <= tot[2] < tot[0]

    # the control: a middle rank that does NOT add its halo rows to the record still has their tiles "known" through the own rows
    # that share them -- and takes the element on windows whose halo rows hold the bright band: the model must object
    class NoHaloRange(Loopback):
        def _halo_range(self):
            pass

    _, _, _, bad = run(NoHaloRange, 1, dict(exchanges=2), (4, 5, 4, 5))
    assert bad, "the host model did not notice halo rows missing from the record"
    proc.close()


def test_four_rank_hip_row_shards_bit_identical_to_single_gpu(tmp_path):
    """Four ranks (gloo, all on cuda:0): ranks 1 and 2 have a neighbour on BOTH sides, which a 2-rank world never exercises --
    two sends and two receives in one batch, halos above and below.  Direct stencils: bit-identical to the whole frame."""
    import torch.multiprocessing as mp

    H, W, fw = 280, 192, 1.0  # 192 px/mm -> 49-tap halation, 19-tap MTF; shards of 70 rows, halos of 24 + 9
    path = str(tmp_path / "shard")
    for direct in (True, False):
        mp.spawn(_worker, args=(4, _free_port(), H, W, fw, path, 0.0, direct), nprocs=4, join=True)
        sharded = np.concatenate([np.load(f"{path}.{r}.npy") for r in range(4)])
        whole, proc, params, img = _render(0, 1, H, W, fw, direct=direct)
        if direct:
            np.testing.assert_array_equal(sharded, whole)
        else:  # FFT windows are anchored at a shard's first row: an fp32 ulp on a handful of pixels
            assert np.max(np.abs(sharded - whole) / np.maximum(np.abs(whole), 1e-3)) <= 2e-6
        proc.close()


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
    # (world = 1: the front kernel also finishes the halation's identity channel -> "front_split")
    assert set(ms) == {"front_split", "halation", "mtf", "grain", "burn_sums", "burn_map", "tail"}
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


def test_graph_replay_of_a_frame_is_bit_identical_to_eager_launches():
    """RowShardedRenderer(graph=True) captures everything downstream of the (absent, world = 1) exchange on its second frame and
    replays it afterwards: same bits as the eager path, for new input CONTENT in the same buffers, and again after the
    output buffer changes (a new capture)."""
    from raw2film_amd import HipProcessor, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    neg, prt, _ = stocks()
    H, W, fw = 300, 640, 2.0  # 320 px/mm: 81-tap halation and 33-tap MTF by FFT, 9 x 9 grain
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw,
                          frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    scale = max(H, W) / fw
    hal, mtf = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3), stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
    eager = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=0, world=1)
    graphed = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=0, world=1, graph=True)
    assert graphed.graph and not eager.graph
    img = torch.from_numpy(synthetic_frame(H, W, seed=5)).cuda()
    out_e = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    out_g = torch.zeros_like(out_e)
    u8_g = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    eager.render(img, out_f32=out_e)
    for k in range(4):  # eager, capture + replay, replay, replay
        out_g.zero_()
        graphed.render(img, out_f32=out_g, out_u8=u8_g)
        assert torch.equal(out_g, out_e), k
    assert [v[1] is not None for v in graphed._graphs.values()] == [True]
    img.copy_(torch.from_numpy(synthetic_frame(H, W, seed=6)).cuda())  # new content, same buffers: the graph reads it
    eager.render(img, out_f32=out_e)
    graphed.render(img, out_f32=out_g, out_u8=u8_g)
    assert torch.equal(out_g, out_e)
    other = torch.zeros_like(out_e)  # another output buffer: eager once, then its own graph
    for _ in range(3):
        graphed.render(img, out_f32=other)
        assert torch.equal(other, out_e)
    assert len(graphed._graphs) == 2
    proc.close()


def test_graph_replay_follows_parameter_table_and_buffer_changes():
    """A captured graph freezes by-value launch arguments and device pointers into the context's tables.  The renderer keys its
    graphs on the context's change counter (r2f_generation) and the parameter block: a new seed, a new output LUT, a new stencil
    and a re-allocated internal scratch buffer (a larger frame on the same context) must each show up in the next frame --
    compared with an eager renderer on the same backend after every change."""
    from raw2film_amd import HipProcessor, filmstock, stencils
    from raw2film_amd.hip_processor import REC709_TO_XYZ
    from raw2film_amd.sharding import HipStageBackend, RowShardedRenderer

    neg, prt, _ = stocks()
    H, W, fw = 300, 640, 2.0
    proc = HipProcessor(device=0)
    kw = dict(matrix=REC709_TO_XYZ, print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3,
              exp_kelvin=6000, color_masking=1.0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, **kw)
    scale = max(H, W) / fw
    hal, mtf = stencils.halation_stencil(scale, 1.0, halation_green_factor=0.3), stencils.mtf_stencil(neg, scale, 0.0, 1.0)
    be = HipStageBackend.for_stencils(proc.ctx, params, hal, mtf)
    eager = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=0, world=1)
    graphed = RowShardedRenderer(be, H, W, halation=True, mtf=True, rank=0, world=1, graph=True)
    img = torch.from_numpy(synthetic_frame(H, W, seed=5)).cuda()
    out_e, out_g = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"), torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def check(what, frames=3):
        eager.render(img, out_f32=out_e)
        for k in range(frames):  # (eager or replay), capture + replay, replay
            out_g.zero_()
            graphed.render(img, out_f32=out_g)
            assert torch.equal(out_g, out_e), (what, k)

    check("first")
    assert any(v[1] is not None for v in graphed._graphs.values())
    before = out_e.clone()
    be.params.seed = SEED + 1  # by-value kernel argument
    check("seed")
    assert not torch.equal(out_e, before)
    before = out_e.clone()
    lut = filmstock.create_lut(neg, prt, color_masking=1.0)
    proc.ctx.set_lut3d(np.ascontiguousarray(lut[..., ::-1]))  # table contents (same size: same address, new bytes -- and a new generation)
    check("lut3d")
    assert not torch.equal(out_e, before)
    before = out_e.clone()
    proc.ctx.set_kernel(1, stencils.mtf_stencil(neg, scale * 0.8, 0.0, 1.0))  # another MTF stencil: new spectra
    check("mtf stencil")
    assert not torch.equal(out_e, before)
    # a larger frame on the same context grows the FFT scratch: the old graphs hold its freed address
    H2, W2 = 900, 1400
    big = RowShardedRenderer(be, H2, W2, halation=True, mtf=True, rank=0, world=1)
    gen = proc.ctx.generation()
    big.render(torch.from_numpy(synthetic_frame(H2, W2, seed=7)).cuda(), out_f32=torch.empty((H2, W2, 3), dtype=torch.float32, device="cuda"))
    assert proc.ctx.generation() > gen
    check("scratch re-allocated")
    # fresh output buffers every frame: the bookkeeping stays bounded
    for _ in range(20):
        graphed.render(img, out_f32=torch.empty_like(out_g))
    assert len(graphed._graphs) <= 8
    proc.close()


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_runs_two_ranks_on_one_gpu_and_reproduces_the_single_rank_frame(tmp_path, ranks):
    """bench.py's N > 1 path end to end on a one-GPU box: torch.distributed.run with two ranks over gloo that share cuda:0 (RCCL
    refuses two ranks on one device; only the transport differs from the real run).  The line must say n_gpus 2 and strong
    scaling, and with the direct stencils (tile-independent tap order) the sharded frame's checksum equals the whole frame's."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "cfg3_45mp", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alone", "--checksum",
              "--direct-stencils"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # no launcher: `python bench.py --gpus 2` starts its two ranks itself (as a child running torch.distributed.run)
    # (4 ranks: the middle two have a neighbour on both sides)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--same-device"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900,
                         env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert l2["n_gpus"] == ranks and l2["scaling"] == "strong" and l1["n_gpus"] == 1
    assert l2["metric"] == l1["metric"] and l2["unit"] == "MP/s" and l2["value"] > 0
    # (the label says what carried the halos: RCCL only with --backend nccl; here gloo, host-staged, ranks sharing one GPU)
    assert f"row-sharded over {ranks} ranks on one GPU" in l2["config"]["sharding"] and "gloo" in l2["config"]["sharding"]
    assert "RCCL" not in l2["config"]["sharding"]
    assert l2["checksum"] == l1["checksum"]
    assert l2["roofline"]["peak"] == ranks * l1["roofline"]["peak"]
    assert l2["gloo_ranks"] == ranks and "rccl_ranks" not in l2  # (RCCL's count appears with --backend nccl: tests/test_gpu_multi.py)
    for line in (l1, l2):
        assert line["ms_per_step_min"] <= line["ms_per_step_median"] <= line["ms_per_step_max"]
        assert line["roofline"]["copy_ceiling_GBps"] > 1000 and 0 < line["roofline"]["frac_of_copy_ceiling"] < 1
