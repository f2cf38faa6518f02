"""One submit per frame on the product surface: r2f_render captures a frame's launches into a HIP graph
and replays it; the grain seed lives in a device-side block (the reference's uniform buffer, gpu_processor.py:585-597), so a new
seed per render -- the reference's semantics and HipProcessor.prepare()'s default -- does not cost the graph.  Everything here
goes through HipProcessor / HipContext.render, i.e. the C ABI."""

import numpy as np
import pytest

from helpers import SEED, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _settings(H, W, fw):
    neg, prt, _ = stocks()
    return neg, dict(print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000,
                     color_masking=1.0)


@pytest.mark.parametrize("shape,fw", [((192, 288), 36.0 * 288 / 6000.0),   # FFT halation + unrolled MTF, like a 24 MP frame
                                       ((160, 256), 36.0 * 256 / 12288.0)])  # both stencils by FFT (100 MP pixel pitch)
def test_four_frames_four_seeds_replay_and_equal_eager_launches(shape, fw):
    from raw2film_amd import HipProcessor

    H, W = shape
    neg, kw = _settings(H, W, fw)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=5)).cuda()
    seeds = [SEED, 1, 0xFFFFFFFF, 123456789]

    eager = HipProcessor(device=0)
    eager.ctx.set_option("render_graph", 0)
    want = [eager.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=s, return_float=True, output="device", **kw).clone()
            for s in seeds]
    assert eager.ctx.render_stats()["replays"] == 0 and eager.ctx.render_stats()["eager"] == len(seeds)
    eager.close()
    assert not torch.equal(want[0], want[1])  # the seeds do reach the grain

    proc = HipProcessor(device=0)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    got = []
    for s in seeds:  # the same buffers every frame: eager, capture + replay, replay, replay
        params = proc.prepare(neg, 6, 0.4, (W, H), seed=s, matrix=_rec709(), **kw)
        proc.ctx.render(frame, params, out_f32=out)
        got.append(out.clone())
    stats = proc.ctx.render_stats()
    assert stats["eager"] == 1 and stats["captures"] == 1 and stats["replays"] == 3, stats
    for g, w in zip(got, want):
        assert torch.equal(g, w)

    # the operator surface itself (fresh output tensors per call: the caching allocator hands the same block back)
    before = proc.ctx.render_stats()["replays"]
    for s, w in zip(seeds, want):
        o = proc.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=s, return_float=True, output="device", **kw)
        assert torch.equal(o, w)
        del o
    assert proc.ctx.render_stats()["replays"] > before
    proc.close()


def _rec709():
    from raw2film_amd.hip_processor import REC709_TO_XYZ

    return REC709_TO_XYZ


def test_process_replays_with_a_random_seed_per_render_and_uint8_output():
    """HipProcessor.process(): device-resident frame (load_image_texture), seed=None -> a new random seed per render like
    upstream; from the third render on the frame is one graph launch."""
    from raw2film_amd import HipProcessor

    H, W = 128, 192
    neg, kw = _settings(H, W, 0.8)
    img = synthetic_frame(H, W, seed=9)
    proc = HipProcessor(device=0)
    outs = [proc.process(img, neg, 6, 0.4, **kw) for _ in range(6)]
    stats = proc.ctx.render_stats()
    assert stats["replays"] >= 3, stats
    assert all(o.dtype == np.uint8 and o.shape == (H, W, 3) for o in outs)
    assert any(not np.array_equal(outs[0], o) for o in outs[1:])  # different seeds, different grain
    # a fixed seed reproduces the eager render bit for bit, replayed or not
    a = proc.process(img, neg, 6, 0.4, seed=77, **kw)
    proc.ctx.set_option("render_graph", 0)
    b = proc.process(img, neg, 6, 0.4, seed=77, **kw)
    np.testing.assert_array_equal(a, b)
    proc.close()


def test_table_changes_and_new_shapes_drop_the_graphs():
    from raw2film_amd import HipProcessor

    H, W = 160, 224
    neg, kw = _settings(H, W, 0.7)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=11)).cuda()
    proc = HipProcessor(device=0)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def render(**over):
        params = proc.prepare(neg, 6, 0.4, (W, H), seed=3, matrix=_rec709(), **dict(kw, **over))
        proc.ctx.render(frame, params, out_f32=out)
        return out.clone()

    base = [render() for _ in range(3)]
    assert proc.ctx.render_stats()["replays"] == 2
    assert torch.equal(base[0], base[1]) and torch.equal(base[0], base[2])
    warm = render(exp_kelvin=5000)          # a new input LUT: generation moves, graphs dropped, this frame eager
    s = proc.ctx.render_stats()
    assert s["dropped"] >= 1 and s["eager"] == 2
    assert not torch.equal(warm, base[0])
    again = [render(exp_kelvin=5000) for _ in range(2)]
    assert torch.equal(again[0], warm) and torch.equal(again[1], warm)
    assert proc.ctx.render_stats()["replays"] == 4
    # a different stage set is a different structure: never served by the other structure's graph
    plain = render(exp_kelvin=5000, halation=False)
    ref = HipProcessor(device=0)
    ref.ctx.set_option("render_graph", 0)
    p2 = ref.prepare(neg, 6, 0.4, (W, H), seed=3, matrix=_rec709(), **dict(kw, exp_kelvin=5000, halation=False))
    want, _ = ref.ctx.render(frame, p2)
    assert torch.equal(plain, want)
    ref.close()
    proc.close()


def test_stage_calls_write_their_own_seed_unless_told_the_block_is_resident():
    """r2f_stage_tail honours p->seed (it writes the frame block first); with F_FRAME_RESIDENT it reads what
    r2f_write_frame_params wrote."""
    from raw2film_amd import HipProcessor, _lib

    H, W = 96, 128
    neg, kw = _settings(H, W, 0.5)
    proc = HipProcessor(device=0)
    ctx = proc.ctx
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=5, **kw)
    D = torch.rand((3, H, W), device="cuda") * 2.0
    o1 = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    o2 = torch.empty_like(o1)
    o3 = torch.empty_like(o1)
    ctx.stage_tail(D, params, out_f32=o1, y0=0, y1=H, H_global=H)
    p9 = _lib.Params.from_buffer_copy(params)
    p9.seed = 9
    ctx.stage_tail(D, p9, out_f32=o2, y0=0, y1=H, H_global=H)
    assert not torch.equal(o1, o2)
    resident = _lib.Params.from_buffer_copy(params)  # seed 5 in the struct, but the block says 9
    resident.flags |= _lib.F_FRAME_RESIDENT
    ctx.write_frame_params(p9)
    ctx.stage_tail(D, resident, out_f32=o3, y0=0, y1=H, H_global=H)
    assert torch.equal(o3, o2)
    proc.close()


def test_replay_on_a_side_stream_with_burn_uint8_output_and_planar_input():
    """The graph is captured on a stream of the context's own and launched on whatever stream the caller is on (here a torch side
    stream); the highlight burn's extra stages (grain to planes, cell sums, map, tail with the map) are part of it; uint8-only
    output and a planar (3, H, W) input are keys like any other."""
    from raw2film_amd import HipProcessor

    H, W = 180, 240
    neg, kw = _settings(H, W, 0.9)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=21)).cuda().permute(2, 0, 1).contiguous()  # (3, H, W)
    ref = HipProcessor(device=0)
    ref.ctx.set_option("render_graph", 0)
    proc = HipProcessor(device=0)
    side = torch.cuda.Stream()
    seeds = [3, 4, 5, 6]
    out_u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    for s in seeds:
        p_ref = ref.prepare(neg, 6, 0.4, (W, H), seed=s, highlight_burn=0.6, burn_scale=20.0, **kw)
        assert p_ref.flags & 32
        _, want = ref.ctx.render(frame, p_ref, want_f32=False, want_u8=True, layout="chw")
        with torch.cuda.stream(side):
            params = proc.prepare(neg, 6, 0.4, (W, H), seed=s, highlight_burn=0.6, burn_scale=20.0, **kw)
            proc.ctx.render(frame, params, out_u8=out_u8, want_f32=False, layout="chw")
        side.synchronize()
        assert torch.equal(out_u8, want), s
    stats = proc.ctx.render_stats()
    assert stats["replays"] == 3 and stats["captures"] == 1 and stats["eager"] == 1, stats
    ref.close()
    proc.close()


def test_more_than_eight_buffer_sets_evict_the_least_recently_used_graph():
    from raw2film_amd import HipProcessor

    H, W = 96, 160
    neg, kw = _settings(H, W, 0.5)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=2)).cuda()
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=_rec709(), **kw)
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(10)]
    for o in outs:  # first sight of every buffer set: kernel by kernel (a set is captured the SECOND time it comes by)
        proc.ctx.render(frame, params, out_f32=o)
    s = proc.ctx.render_stats()
    assert s["captures"] == 0 and s["eager"] == 10 and s["replays"] == 0, s
    for o in outs:  # ten distinct buffer sets seen before: ten captures, the two oldest evicted on the way
        proc.ctx.render(frame, params, out_f32=o)
    s = proc.ctx.render_stats()
    assert s["captures"] == 10 and s["dropped"] == 2 and s["replays"] == 10, s
    proc.ctx.render(frame, params, out_f32=outs[9])  # still cached
    proc.ctx.render(frame, params, out_f32=outs[0])  # evicted: captured again
    s = proc.ctx.render_stats()
    assert s["captures"] == 11 and s["replays"] == 12, s
    assert all(torch.equal(o, outs[0]) for o in outs)
    proc.close()


def test_an_eviction_does_not_wait_for_other_work_on_the_device():
    """evicting a captured graph used to hipDeviceSynchronize() inside r2f_render -- a caller cycling through
    more than eight buffer sets (a preview with a ring of destination textures) stalled every stream of the device per eviction.
    The executable graph is now parked behind an event recorded at its last launch and destroyed once that has completed.  Here
    a long spin kernel runs on another stream (standing in for a second context's work) while this context evicts: the renders
    return while the spin is still running, the results are the eager ones', and the parked graphs go away afterwards."""
    import time

    from raw2film_amd import HipProcessor

    H, W = 96, 160
    neg, kw = _settings(H, W, 0.5)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=2)).cuda()
    proc = HipProcessor(device=0)
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=_rec709(), **kw)
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(12)]
    for _ in range(2):  # seen, then captured: 12 captures through 8 slots, 4 evictions already
        for o in outs:
            proc.ctx.render(frame, params, out_f32=o)
    torch.cuda.synchronize()
    want = outs[0].clone()
    side = torch.cuda.Stream()
    busy = torch.cuda.Event()
    # a spin kernel of ~3 s on the other stream (its cycle counter's rate is calibrated first)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        a.record()
        torch.cuda._sleep(20_000_000)
        b.record()
    side.synchronize()
    per_ms = 20_000_000 / max(a.elapsed_time(b), 1e-3)
    with torch.cuda.stream(side):
        for _ in range(6):
            torch.cuda._sleep(int(min(per_ms * 500, 2_000_000_000)))  # 6 x 0.5 s (each count stays inside 31 bits)
        busy.record()
    t0 = time.perf_counter()
    for o in outs:  # every one of these evicts (the cache holds the 8 most recent sets, these come round-robin)
        proc.ctx.render(frame, params, out_f32=o)
    dt = time.perf_counter() - t0  # HOST time of the twelve calls: a device-wide synchronisation inside one would sit out the spin
    still_running = not busy.query()
    s = proc.ctx.render_stats()
    torch.cuda.synchronize()
    # (whether the renders also EXECUTE beside the spin depends on how the runtime maps streams to hardware queues; what the
    # library owes is that its calls do not wait for other streams' work)
    assert still_running, "the spin kernel was meant to outlast the calls: lengthen it"
    assert dt < 0.5, dt
    assert s["captures"] == 24 and s["dropped"] >= 16, s
    assert all(torch.equal(o, want) for o in outs)
    proc.ctx.render(frame, params, out_f32=outs[0])  # (polls the parked graphs: all of their events have completed by now)
    torch.cuda.synchronize()
    proc.close()


def test_fresh_buffers_every_frame_are_never_captured_and_two_alternating_sets_are():
    """Results a caller keeps alive arrive in new buffers every frame: capturing those would cost a graph per frame that is never
    replayed.  Double buffering (A, B, A, B, ...) is captured from the second round on."""
    from raw2film_amd import HipProcessor

    H, W = 96, 160
    neg, kw = _settings(H, W, 0.5)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=2)).cuda()
    proc = HipProcessor(device=0)
    kept = [proc.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=s, return_float=True, output="device", **kw)
            for s in range(12)]  # all alive: twelve different allocations
    s = proc.ctx.render_stats()
    assert s["captures"] == 0 and s["replays"] == 0 and s["eager"] == 12, s
    assert len({int(k.data_ptr()) for k in kept}) == 12
    a, b = torch.empty_like(kept[0]), torch.empty_like(kept[0])
    for i in range(8):
        proc.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=i % 12, return_float=True, output="device",
                           out=a if i % 2 == 0 else b, **kw)
        assert torch.equal(a if i % 2 == 0 else b, kept[i % 12])
    s = proc.ctx.render_stats()
    assert s["captures"] == 2 and s["replays"] == 6 and s["eager"] == 14, s
    proc.close()


def test_wrong_output_tensors_raise_before_anything_is_launched():
    """The C ABI takes raw pointers; the binding checks a caller's output tensors (dtype, shape, rows covered) so that a mistake
    is a ValueError, not a GPU memory fault."""
    from raw2film_amd import HipProcessor

    H, W = 64, 96
    neg, kw = _settings(H, W, 0.5)
    proc = HipProcessor(device=0)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=4)).cuda()
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=_rec709(), **kw)
    for bad in (dict(out_f32=torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")),        # a quarter of the bytes
                dict(out_f32=torch.empty((H - 1, W, 3), dtype=torch.float32, device="cuda")),
                dict(out_u8=torch.empty((H, W, 3), dtype=torch.float32, device="cuda"), want_f32=False),
                dict(out_f32=torch.empty((H, W, 4), dtype=torch.float32, device="cuda")),
                dict(out_f32=torch.empty((H, W, 3), dtype=torch.float32))):                        # host memory
        with pytest.raises(ValueError):
            proc.ctx.render(frame, params, **bad)
    D = torch.rand((3, H, W), device="cuda")
    with pytest.raises(ValueError):  # rows [0, H) asked of a buffer that holds [0, H / 2)
        proc.ctx.stage_tail(D, params, out_f32=torch.empty((H // 2, W, 3), dtype=torch.float32, device="cuda"), y0=0, y1=H, H_global=H)
    with pytest.raises(ValueError):
        proc.ctx.stage_tail(D, params, out_u8=torch.empty((H, W, 3), dtype=torch.float32, device="cuda"), y0=0, y1=H, H_global=H)
    ok = torch.empty((H // 2, W, 3), dtype=torch.float32, device="cuda")
    proc.ctx.stage_tail(D, params, out_f32=ok, out_gy0=H // 2, y0=H // 2, y1=H, H_global=H)  # a shard's own rows: fine
    burn = proc.prepare(neg, 6, 0.4, (W, H), seed=1, matrix=_rec709(), highlight_burn=0.5, burn_scale=8.0, **kw)
    with pytest.raises(ValueError):  # a burn map that is not the low-resolution grid of this frame
        proc.ctx.stage_tail(D, burn, out_f32=torch.empty((H, W, 3), dtype=torch.float32, device="cuda"), y0=0, y1=H, H_global=H,
                            burn_map=torch.zeros((3, 3), device="cuda"))
    with pytest.raises(ValueError):
        proc.ctx.stage_burn_map(torch.zeros((2, 2), device="cuda"), burn, W=W, H_global=H)
    with pytest.raises(ValueError):
        proc.ctx.histogram_render(torch.zeros((3, 255), dtype=torch.int32, device="cuda"), [0] * 32, 40)
    torch.cuda.synchronize()
    proc.close()


def test_a_resident_seed_caller_gets_a_fresh_exposure_range_every_frame():
    """a caller that keeps the seed resident (R2F_F_FRAME_RESIDENT: it wrote the frame block itself) used to keep the
    exposure RANGE of earlier frames too -- only r2f_write_frame_params reset it -- so the union of the frames so far decided frame
    N's scratch element.  Now every whole-frame render starts its record empty, eager or replayed: a benign frame after a hostile
    one takes the 12-byte element again, and its bits are those of the same frame rendered by a non-resident caller."""
    from helpers import oracle_inputs
    from raw2film_amd import _lib
    from raw2film_amd.context import HipContext
    from test_gpu_parity import setup_ctx

    H, W = 700, 1100
    neg, prt, _ = stocks()
    p = oracle_inputs(neg, prt, 341.33, seed=SEED)
    benign = np.clip(synthetic_frame(H, W, seed=9), 0.01, 16.0)
    hostile = benign.copy()
    hostile[300:304, 500:504] = 65504.0
    hostile[50:150, 60:200] = 1e-5
    c = HipContext(0)
    try:
        params = setup_ctx(c, p)
        c.set_option("stencil_fft_window_rows", 256)
        c.set_option("stencil_fft_window", 512)
        buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        want = {}
        for name, frame in (("benign", benign), ("hostile", hostile)):  # a non-resident caller: r2f_render writes the block itself
            buf.copy_(torch.from_numpy(frame))
            for _ in range(3):
                c.render(buf, params, out_f32=out)
            want[name] = (out.cpu().numpy(), c.frame_exposure_range()["twelve_byte_element"])
        assert want["benign"][1] and not want["hostile"][1]
        resident = _lib.Params.from_buffer_copy(params)
        resident.flags |= _lib.F_FRAME_RESIDENT
        c.write_frame_params(params)  # the caller's own seed write, once
        for mode in ("replay", "eager"):
            c.set_option("render_graph", 1 if mode == "replay" else 0)
            for name, frame in (("hostile", hostile), ("benign", benign), ("benign", benign), ("hostile", hostile), ("benign", benign)):
                buf.copy_(torch.from_numpy(frame))
                c.render(buf, resident, out_f32=out)
                rng = c.frame_exposure_range()
                assert rng["armed"] and rng["twelve_byte_element"] == want[name][1], (mode, name, rng)
                np.testing.assert_array_equal(out.cpu().numpy(), want[name][0])
    finally:
        c.close()


def test_a_pending_error_of_another_runtime_user_is_not_reported_as_a_launch_failure():
    """every launch wrapper used to return hipGetLastError(), i.e. whatever error was pending on the calling thread --
    one PyTorch or RCCL left there -- as its own status (R2F_EHIP), consuming it on the way.  Launches now go through
    hipLaunchKernel and report that call's own return value: with an error pending (an invalid hipFree through ctypes), a stage
    call and a whole-frame render succeed and produce the same bits as without."""
    import ctypes

    from raw2film_amd import HipProcessor

    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        pytest.skip("no libamdhip64 to provoke an error with")
    hip.hipFree.argtypes = [ctypes.c_void_p]
    H, W = 160, 256
    neg, kw = _settings(H, W, 36.0 * W / 12288.0)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=5)).cuda()
    proc = HipProcessor(device=0)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    params = proc.prepare(neg, 6, 0.4, (W, H), seed=SEED, matrix=_rec709(), **kw)
    for _ in range(3):
        proc.ctx.render(frame, params, out_f32=out)
    want = out.clone()
    proc.ctx.set_option("render_graph", 0)  # kernel by kernel: ~20 launches, each of which used to read the pending error
    out.zero_()
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    assert hip.hipFree(ctypes.c_void_p(0x10)) != 0  # an error of "somebody else's", left pending on this thread
    proc.ctx.render(frame, params, out_f32=out)     # raised R2FError before
    # ... and the error is still THEIRS to read: the library neither reported nor consumed it (torch's next check would raise it)
    assert hip.hipGetLastError() != 0
    assert hip.hipGetLastError() == 0
    assert torch.equal(out, want)
    assert hip.hipFree(ctypes.c_void_p(0x10)) != 0
    proc.ctx.stage_front(frame, params, 0, dst=E)
    assert hip.hipGetLastError() != 0
    torch.cuda.synchronize()
    proc.close()
