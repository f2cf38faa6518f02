"""Row-sharded rendering and batch sharding, exercised on CPU: 2 processes, gloo backend.
The exchange logic of raw2film_amd.sharding is compute-agnostic; here the stage backend is the
NumPy oracle (tests may use it), so the test checks halo bookkeeping, reflect handling at the
global edges and the neighbour exchange -- not the HIP kernels (tests/test_gpu_parity.py does)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import stages as st
from raw2film_amd import sharding, stencils

from helpers import SEED, oracle_inputs, stocks, synthetic_frame


class OracleStageBackend:
    """Same protocol as sharding.HipStageBackend, on CPU tensors, computed by the oracle."""

    def __init__(self, p: st.RenderInputs):
        self.p = p
        self.halation_taps = stencils.vertical_reach(p.halation_kernel) if p.halation_kernel is not None else (0, 0)
        self.mtf_taps = stencils.vertical_reach(p.mtf_kernel) if p.mtf_kernel is not None else (0, 0)
        self.halation_taps_per_channel = (stencils.vertical_reach_per_channel(p.halation_kernel)
                                          if p.halation_kernel is not None else None)

    def empty(self, rows, W):
        return torch.full((3, rows, W), float("nan"), dtype=torch.float32)

    @staticmethod
    def _hwc(t):
        return t.numpy() if t.shape[-1] in (3, 4) else np.transpose(t.numpy(), (1, 2, 0))

    def _front(self, image_rows, in_gy0, upto, y0, y1):
        x = self._hwc(image_rows)[y0 - in_gy0:y1 - in_gy0, :, :3]
        if self.p.matrix is not None:
            x = st.apply_matrix3x3(x, self.p.matrix)
        x = st.apply_2d_lut(x, self.p.lut_2d)
        if upto >= 1:
            x = st.multi_channel_interp(st.log_clip(x), self.p.lut_1d)
        return x

    def front(self, image_rows, in_gy0, upto, dst, dst_gy0, y0, y1, H):
        x = self._front(image_rows, in_gy0, upto, y0, y1)
        dst[:, y0 - dst_gy0:y1 - dst_gy0, :] = torch.from_numpy(np.transpose(x, (2, 0, 1)).copy())

    @staticmethod
    def _stencil(src, src_gy0, kernel, y0, y1, H):
        above, below = kernel.shape[0] // 2, kernel.shape[0] - 1 - kernel.shape[0] // 2
        rows = np.arange(y0 - above, y1 + below)
        period = max(2 * H - 2, 1)
        rows = np.abs(((rows % period) + period) % period)
        rows = np.where(rows >= H, period - rows, rows)  # reflect-101 on the global frame
        idx = np.clip(rows - src_gy0, 0, src.shape[1] - 1)
        gathered = np.transpose(src.numpy()[:, idx, :], (1, 2, 0))
        for c in range(3):  # per plane: a single-tap channel (the halation's blue layer) needs no halo rows of its own
            used = kernel[..., c if kernel.shape[2] > 1 else 0].any(axis=1)
            if not used.any():
                continue
            lo, hi = np.nonzero(used)[0][[0, -1]]
            need = rows[lo:len(rows) - (kernel.shape[0] - 1 - hi)]
            assert need.min() >= src_gy0 and need.max() < src_gy0 + src.shape[1], "halo rows missing"
            assert not np.isnan(gathered[lo:len(rows) - (kernel.shape[0] - 1 - hi), :, c]).any(), "halo rows were never filled"
        gathered = np.nan_to_num(gathered)
        full = st.convolve_2d(gathered, kernel)
        return full[above:above + (y1 - y0)]

    def halation(self, E, e_gy0, D, d_gy0, y0, y1, H):
        x = self._stencil(E, e_gy0, self.p.halation_kernel, y0, y1, H)
        x = st.multi_channel_interp(st.log_clip(x), self.p.lut_1d)
        D[:, y0 - d_gy0:y1 - d_gy0, :] = torch.from_numpy(np.transpose(x, (2, 0, 1)).copy())

    def mtf(self, D, d_gy0, D2, d2_gy0, y0, y1, H):
        x = self._stencil(D, d_gy0, self.p.mtf_kernel, y0, y1, H)
        D2[:, y0 - d2_gy0:y1 - d2_gy0, :] = torch.from_numpy(np.transpose(x, (2, 0, 1)).copy())

    def _grained(self, D, d_gy0, y0, y1, H):
        x = np.transpose(D.numpy()[:, y0 - d_gy0:y1 - d_gy0, :], (1, 2, 0))
        if self.p.grain_lut is not None:
            gk = self.p.grain_kernel if self.p.grain_kernel is not None else np.ones((1, 1), np.float32)
            x = st.apply_grain(x, self.p.grain_lut, gk, self.p.seed, self.p.grain_mono, row0=y0, H_global=H)
        return x

    def tail(self, D, d_gy0, out_f32, out_u8, out_gy0, y0, y1, H, burn_map=None):
        if burn_map is None:
            x = self._grained(D, d_gy0, y0, y1, H)
        else:  # grain already applied by grain()
            x = np.transpose(D.numpy()[:, y0 - d_gy0:y1 - d_gy0, :], (1, 2, 0))
            cell = st.burn_geometry(H, x.shape[1], self.p.burn_scale)[0]
            x = st.burn_apply(x, burn_map.numpy(), cell, self.p.highlight_burn, row0=y0, H_global=H)
        x = st.apply_lut_tetrahedral(x, self.p.lut_3d, 0.25)
        out_f32[y0 - out_gy0:y1 - out_gy0] = torch.from_numpy(x)

    def grain(self, D, d_gy0, G, g_gy0, y0, y1, H):
        x = self._grained(D, d_gy0, y0, y1, H)
        G[:, y0 - g_gy0:y1 - g_gy0, :] = torch.from_numpy(np.transpose(x, (2, 0, 1)).copy())

    def burn_sums(self, D, d_gy0, y0, y1, H):
        green = D.numpy()[1, y0 - d_gy0:y1 - d_gy0, :].astype(np.float64)
        W = green.shape[1]
        _, h_lo, w_lo = st.burn_geometry(H, W, self.p.burn_scale)
        wy, wx = st.area_table(H, h_lo)[:, y0:y1], st.area_table(W, w_lo)
        return torch.from_numpy((wy @ green @ wx.T).astype(np.float32))

    def burn_map(self, sums, W, H):
        from scipy import ndimage

        down = np.clip(sums.numpy() - np.float32(self.p.d_ref), 0, None)
        return torch.from_numpy(ndimage.gaussian_filter(down, sigma=3, truncate=2))

    def front_to_output(self, image_rows, in_gy0, out_f32, out_u8, out_gy0, y0, y1, H):
        x = self._front(image_rows, in_gy0, 1, y0, y1)
        out_f32[y0 - out_gy0:y1 - out_gy0] = torch.from_numpy(st.apply_lut_tetrahedral(x, self.p.lut_3d, 0.25))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs(H, W, scale, burn=0.0, **kw):
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, scale, seed=SEED, **kw)
    img = synthetic_frame(H, W, seed=5)
    if burn:
        p.highlight_burn, p.burn_scale, p.d_ref = burn, 10.0, float(neg.d_ref[1])
        img[10:30, 8:40] *= 10.0
    return p, img


def _worker(rank, world, port, H, W, scale, flags, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # no hostname lookup (the box's name may not resolve)
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flags = dict(flags)
        flags_ex = flags.pop("exchanges", "auto")
        p, img = _inputs(H, W, scale, **flags)
        be = OracleStageBackend(p)
        rr = sharding.RowShardedRenderer(be, H, W, halation=p.halation_kernel is not None, mtf=p.mtf_kernel is not None,
                                         grain=p.grain_lut is not None, burn=bool(p.highlight_burn), split_halation=True, exchanges=flags_ex)
        rr.trace = []
        r0, r1 = rr.plan.r0, rr.plan.r1
        out = torch.zeros((r1 - r0, W, 3), dtype=torch.float32)
        rr.render(torch.from_numpy(img[r0:r1].copy()), out_f32=out)
        if world > 1 and rr.halation and rr.mtf and flags_ex == 2:
            # two exchanges: exposure halo for the halation alone, the halation on exactly the own rows, then the density halo
            ha, ma = be.halation_taps[0], be.mtf_taps[0]
            assert rr.schedule == (2, False) and not rr.single_exchange
            assert rr.halo_e_ch[0] == (ha, ha) and rr.halo_e_ch[2] == (0, 0), rr.halo_e_ch
            assert rr.trace == ["exchange_start", "exchange_finish", "exchange_density"], rr.trace
        elif world > 1 and rr.halation and rr.mtf:
            # one exchange; the blue plane (a single halation tap) travels with the MTF's halo only; the interior halation is
            # issued while the halos are in flight, the boundary bands after they arrived
            ha, ma = be.halation_taps[0], be.mtf_taps[0]
            assert rr.halo_e_ch[0] == (ha + ma, ha + ma) and rr.halo_e_ch[2] == (ma, ma), rr.halo_e_ch
            if world == 2:
                assert rr.split is not None
            if rr.split is not None:
                assert rr.trace == ["exchange_start", "halation_interior", "exchange_finish", "halation_bands"], rr.trace
            else:  # (shards too short to have rows that need no halo: one halation call after the exchange)
                assert rr.trace == ["exchange_start", "exchange_finish"], rr.trace
        gathered = [torch.zeros((b - a, W, 3), dtype=torch.float32) for a, b in sharding.shard_rows(H, world)]
        if world > 1:
            # shards may differ by one row: gather through a padded buffer
            rows_max = max(t.shape[0] for t in gathered)
            pad = torch.zeros((rows_max, W, 3), dtype=torch.float32)
            pad[: out.shape[0]] = out
            bufs = [torch.zeros_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad)
            gathered = [b[: g.shape[0]] for b, g in zip(bufs, gathered)]
        else:
            gathered = [out]
        if rank == 0:
            np.save(result_path, torch.cat(gathered).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize(
    "H,W,scale,flags",
    [
        (97, 64, 120.0, dict()),  # halation + MTF + grain, odd split (49 / 48 rows)
        (64, 48, 100.0, dict(mtf=False)),
        (64, 48, 160.0, dict(halation=False, grain=0)),
        (40, 32, 60.0, dict(halation=False, mtf=False)),
        (40, 32, 60.0, dict(halation=False, mtf=False, grain=0)),  # LUTs only: fused pointwise pass
        (97, 64, 120.0, dict(burn=0.7)),  # S7 on top of everything: all-reduce of the low-res cell sums
        (64, 48, 100.0, dict(burn=0.7, halation=False, mtf=False, grain=0)),
        (97, 64, 120.0, dict(exchanges=2)),  # the two-exchange schedule: exposure halo, halation on the own rows, density halo
        (97, 64, 120.0, dict(exchanges=2, burn=0.7)),
    ],
)
def test_two_rank_row_shards_match_whole_frame(tmp_path, H, W, scale, flags):
    world = 2
    path = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, scale, flags, path), nprocs=world, join=True)
    got = np.load(path)
    flags = {k: v for k, v in flags.items() if k != "exchanges"}
    p, img = _inputs(H, W, scale, **flags)
    ref = st.render(img, p)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("world,H,W,scale,flags", [
    (4, 131, 48, 100.0, dict()),            # interior ranks talk to BOTH neighbours (a 2-rank world never does); shards of 33 / 33 / 33 / 32
    (3, 100, 40, 80.0, dict(burn=0.7)),     # odd world, S7's all-reduce over three ranks
    (8, 163, 40, 100.0, dict()),            # the node's full width: eight ranks, six of them interior, shards of 21 / 21 / 21 / 20 x 5 rows
    (4, 131, 48, 100.0, dict(exchanges=2)),  # two exchanges with interior ranks
])
def test_interior_ranks_exchange_with_both_neighbours(tmp_path, world, H, W, scale, flags):
    path = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, scale, flags, path), nprocs=world, join=True)
    got = np.load(path)
    flags = {k: v for k, v in flags.items() if k != "exchanges"}
    p, img = _inputs(H, W, scale, **flags)
    ref = st.render(img, p)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_shard_rows_partition():
    for H in (1, 7, 8192, 8191):
        for world in (1, 2, 3, 8):
            parts = sharding.shard_rows(H, world)
            assert parts[0][0] == 0 and parts[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_shards_shorter_than_halo_are_rejected():
    p, _ = _inputs(32, 32, 341.33)  # 87-tap halation: 42-row halo
    be = OracleStageBackend(p)
    with pytest.raises(ValueError, match="shorter than"):
        sharding.RowShardedRenderer(be, 32, 32, halation=True, mtf=True, rank=0, world=2)


def test_batch_sharder_round_robin_skip_and_order():
    tasks = [f"frame{i}" for i in range(11)]
    seen = {}
    for rank in range(4):
        bs = sharding.BatchSharder(rank, 4)

        def prepare(t):
            if t == "frame5":
                raise RuntimeError("decode failed")
            return t.upper()

        results, skipped = bs.run(tasks, prepare, lambda t, payload: (t, payload))
        for i, r in results.items():
            assert i % 4 == rank and r == (tasks[i], tasks[i].upper())
            seen[i] = rank
        assert all(i % 4 == rank for i in skipped)
        if rank == 1:
            assert skipped == [5]
    assert sorted(seen) == [i for i in range(11) if i != 5]


def test_batch_sharder_cancel():
    bs = sharding.BatchSharder(0, 1)
    done = []

    def execute(t, payload):
        done.append(t)
        if len(done) == 2:
            bs.cancel()
        return t

    bs.run(list(range(100)), lambda t: t, execute)
    assert 2 <= len(done) < 100


def test_batch_sharder_stops_its_producer_when_execute_raises_or_the_run_is_cancelled():
    """The producer thread must not outlive run(): neither blocked on the depth-1 queue with a payload in its hands, nor
    decoding the rest of the batch."""
    import threading
    import time

    before = threading.active_count()
    prepared = []

    def prepare(t):
        prepared.append(t)
        return bytearray(1024)

    def execute(t, payload):
        if t == 3:
            raise RuntimeError("device phase failed")
        return t

    bs = sharding.BatchSharder(0, 1)
    with pytest.raises(RuntimeError, match="device phase failed"):
        bs.run(list(range(1000)), prepare, execute)
    time.sleep(0.3)
    assert threading.active_count() == before  # producer gone
    assert len(prepared) < 20  # it did not go on decoding the batch
    # the object is still usable: an exception ends that run, it is not a cancel request
    results, skipped = bs.run([10, 11], lambda t: t, lambda t, p: p)
    assert results == {0: 10, 1: 11} and skipped == []

    bs = sharding.BatchSharder(0, 1)
    n = []

    def slow_execute(t, payload):
        n.append(t)
        if len(n) == 2:
            bs.cancel()
        return t

    bs.run(list(range(1000)), prepare, slow_execute)
    time.sleep(0.3)
    assert threading.active_count() == before


def test_batch_sharder_pipelined_collect_keeps_one_frame_in_flight():
    """With `collect`, execute() only submits: frame k is collected after frame k + 1 went in, the last one at the end, results
    keyed like the serial form's."""
    log = []
    bs = sharding.BatchSharder(0, 1)

    def execute(t, payload):
        log.append(("submit", t))
        return ("handle", t)

    def collect(t, handle):
        assert handle == ("handle", t)
        log.append(("collect", t))
        return t * 10

    def prepare(t):
        if t == 2:
            raise RuntimeError("decode failed")
        return t

    results, skipped = bs.run([0, 1, 2, 3], prepare, execute, collect=collect)
    assert results == {0: 0, 1: 10, 3: 30} and skipped == [2]
    assert log == [("submit", 0), ("submit", 1), ("collect", 0), ("submit", 3), ("collect", 1), ("collect", 3)]


def test_batch_sharder_progress_counts_finished_frames_and_a_failure_keeps_the_frame_in_flight():
    """in collect mode `progress` used to fire when a frame was SUBMITTED (a GUI bar ran one frame ahead); it now
    fires when the frame has been collected.  And when execute() raises, the frame already in flight is still collected (its
    export runs) before the error leaves run()."""
    log = []
    bs = sharding.BatchSharder(0, 1)
    results, skipped = bs.run([0, 1, 2], lambda t: t, lambda t, p: (log.append(("submit", t)), t)[1],
                              progress=lambda i, n: log.append(("done", i, n)), collect=lambda t, h: (log.append(("collect", t)), t)[1])
    assert results == {0: 0, 1: 1, 2: 2} and skipped == []
    assert log == [("submit", 0), ("submit", 1), ("collect", 0), ("done", 0, 3), ("submit", 2), ("collect", 1), ("done", 1, 3),
                   ("collect", 2), ("done", 2, 3)]
    # serial form: after each execute
    log.clear()
    bs.run([0, 1], lambda t: t, lambda t, p: (log.append(("exec", t)), t)[1], progress=lambda i, n: log.append(("done", i, n)))
    assert log == [("exec", 0), ("done", 0, 2), ("exec", 1), ("done", 1, 2)]

    log.clear()

    def execute(t, payload):
        if t == 2:
            raise RuntimeError("submit failed")
        log.append(("submit", t))
        return t

    with pytest.raises(RuntimeError, match="submit failed"):
        bs.run([0, 1, 2, 3], lambda t: t, execute, collect=lambda t, h: log.append(("collect", t)))
    assert log == [("submit", 0), ("submit", 1), ("collect", 0), ("collect", 1)]  # frame 1 was in flight when frame 2 failed


class _BandedBackend(OracleStageBackend):
    """... that also tells the row tiler how many rows one window row of its (imaginary) FFT form yields, so that the two-exchange
    schedule becomes a candidate."""

    def halation_band_rows(self, W, rows):
        return 10


def _tuning_worker(rank, world, port, H, W, scale, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, img = _inputs(H, W, scale)
        be = _BandedBackend(p)
        # a clock per rank: locally rank 0 and 2 would pick (1, True), rank 1 (2, False); the slowest rank of (1, True) is slower
        # than the slowest rank of (2, False), so every rank has to end on (2, False)
        clock = {(1, False): [5.0, 5.0, 5.0], (1, True): [3.0, 6.0, 3.0], (2, False): [4.0, 4.5, 4.2]}
        seen = []

        def timer(cand, render):
            seen.append(cand)
            render()
            return clock[cand][rank] + 0.01 * len(seen)  # (later frames of a candidate a little slower: the best one counts)

        rr = sharding.RowShardedRenderer(be, H, W, halation=True, mtf=True, grain=p.grain_lut is not None, frame_timer=timer, tune_frames=2)
        assert rr.tuning and rr._candidates == [(1, False), (1, True), (2, False)], rr._candidates
        r0, r1 = rr.plan.r0, rr.plan.r1
        out = torch.zeros((r1 - r0, W, 3), dtype=torch.float32)
        ref = st.render(img, p)[r0:r1]
        frames = 0
        while rr.tuning:  # every measuring frame is a correct frame, whatever the candidate
            rr.render(torch.from_numpy(img[r0:r1].copy()), out_f32=out)
            frames += 1
            np.testing.assert_allclose(out.numpy(), ref, rtol=0, atol=2e-6)
        # (one untimed frame ahead of every candidate's two timed ones: its first frame may still build a spectrum or grow the scratch)
        assert frames == 3 * (1 + 2) and seen == [(1, False)] * 2 + [(1, True)] * 2 + [(2, False)] * 2, (frames, seen)
        assert rr.schedule == (2, False) and not rr.single_exchange, rr.schedule
        assert [round(t, 2) for t in rr.tuned_ms] == [5.01, 6.03, 4.55], rr.tuned_ms  # the MAX over the ranks of each rank's best frame
        rr.trace = []
        rr.render(torch.from_numpy(img[r0:r1].copy()), out_f32=out)
        assert rr.trace == ["exchange_start", "exchange_finish", "exchange_density"], rr.trace
        np.testing.assert_allclose(out.numpy(), ref, rtol=0, atol=2e-6)
        if rank == 0:
            np.save(result_path, np.asarray(rr.tuned_ms))
    finally:
        dist.destroy_process_group()


def test_the_schedule_is_measured_and_agreed_across_ranks(tmp_path):
    """Round 5: with `exchanges` / `split_halation` on "auto" the ranks walk the candidate schedules in lockstep on their first
    frames, time each (here: an injected clock per rank), all-reduce the times (MAX) and take the schedule whose SLOWEST rank was
    fastest -- the same one everywhere, whatever each rank would have picked on its own (the two-exchange form needs its neighbours
    to send density rows)."""
    path = str(tmp_path / "tuned.npy")
    mp.spawn(_tuning_worker, args=(3, _free_port(), 120, 40, 80.0, path), nprocs=3, join=True)
    assert np.load(path).shape == (3,)
