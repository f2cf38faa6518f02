"""uint16 hand-off on the device: r2f_decode_u16 against the NumPy expressions of raw_to_linear (raw_conversion.py:50-52), bit for
bit, and a uint16 source through HipProcessor against the same frame decoded on the host."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import post  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    from raw2film_amd.context import HipContext

    c = HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("shape", [(64, 96, 3), (33, 47, 3), (33, 47, 4), (1, 1, 3), (2, 3, 4), (400, 600, 3)])
@pytest.mark.parametrize("stops", [0.0, 0.657791852173465, -2.3])
def test_decode_kernel_is_bit_identical_to_numpy(ctx, shape, stops):
    from raw2film_amd import decode

    rng = np.random.default_rng(shape[0] * 7 + shape[2])
    u16 = rng.integers(0, 65536, shape).astype(np.uint16)
    u16.flat[:3] = (0, 65535, 1)
    dev = torch.from_numpy(u16.view(np.int16)).cuda()
    got = ctx.decode_u16(dev, decode.exposure_factor(stops)).cpu().numpy()
    want = post.decode_u16(u16, stops)
    assert got.shape == shape[:2] + (3,) and got.dtype == np.float32
    np.testing.assert_array_equal(got, want)
    # an unaligned view of the same data takes the scalar path
    if shape[2] == 3 and shape[1] > 2:
        flat = torch.empty(u16.size + 1, dtype=torch.int16, device="cuda")
        flat[1:] = dev.reshape(-1)
        got2 = ctx.decode_u16(flat[1:].view(shape), decode.exposure_factor(stops)).cpu().numpy()
        np.testing.assert_array_equal(got2, want)


def test_clamp_and_argument_checks(ctx):
    top = torch.from_numpy(np.full((2, 4, 3), 65535, np.uint16).view(np.int16)).cuda()
    assert float(ctx.decode_u16(top, 2.0**17).max()) == 65504.0  # gpu_processor.py:275
    with pytest.raises(ValueError):
        ctx.decode_u16(torch.zeros((4, 4, 3), dtype=torch.float32, device="cuda"), 1.0)
    with pytest.raises(ValueError):
        ctx.decode_u16(torch.zeros((4, 4, 2), dtype=torch.int16, device="cuda"), 1.0)


def test_uint16_source_renders_like_its_host_decoded_float_frame():
    from raw2film_amd import HipProcessor, decode, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(11)
    H, W = 120, 180
    u16 = (rng.uniform(0, 1, (H, W, 3)) ** 3 * 20000).astype(np.uint16)
    meta = {"EXIF:FNumber": 5.6, "EXIF:ISO": 200, "EXIF:ExposureTime": 1 / 125}
    stops = decode.auto_exposure(u16, metadata=meta)
    host_frame = decode.decode_u16_host(u16, stops)
    proc = HipProcessor(device=0, payload_alpha=False)
    kw = dict(print_film=prt, lens_correction=False, seed=7)
    want = proc.process(host_frame, neg, 6, 0.4, **kw)
    got = proc.process(u16, neg, 6, 0.4, metadata=meta, **kw)
    np.testing.assert_array_equal(got, want)
    assert got.shape[2] == 3 and got.dtype == np.uint8 and got.std() > 5
    # the payload stays uint16 across the two-phase API (6 bytes per pixel over PCIe) and carries its exposure factor
    payload = proc.extract_image_data_cpu(u16, lens_correction=False, metadata=meta)
    assert payload["image_array"].dtype == np.uint16 and payload["image_array"].shape[2] == 3
    assert payload["u16_factor"] == float(decode.exposure_factor(stops))
    np.testing.assert_array_equal(proc.process_preloaded(payload, neg, 6, 0.4, **kw), proc.process_preloaded(
        proc.extract_image_data_cpu(host_frame, lens_correction=False), neg, 6, 0.4, **kw))
    # the stops can come from the caller instead (upstream's value, or a manual one)
    fixed = proc.process(u16, neg, 6, 0.4, exposure=0.5, **kw)
    np.testing.assert_array_equal(fixed, proc.process(decode.decode_u16_host(u16, 0.5), neg, 6, 0.4, **kw))
    proc.close()


def test_uint16_payloads_through_the_pipelined_batch_path():
    """BatchSharder.run(..., collect=...) with HipProcessor.submit_preloaded on uint16 payloads (upload, device conversion, render
    and download overlapped across frames) returns what process_preloaded returns frame by frame."""
    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd.sharding import BatchSharder

    stocks = filmstock.builtin_stocks()
    neg, prt = stocks["Kodak Portra 400"], stocks["Kodak 2383"]
    rng = np.random.default_rng(12)
    frames = [(rng.uniform(0, 1, (90, 132, 3)) ** 2 * 30000).astype(np.uint16) for _ in range(5)]
    proc = HipProcessor(device=0)
    kw = dict(print_film=prt, seed=9)
    prepare = lambda i: proc.extract_image_data_cpu(frames[i], lens_correction=False, exposure=0.25 * i)  # noqa: E731
    want = [proc.process_preloaded(prepare(i), neg, 6, 0.4, **kw) for i in range(len(frames))]
    res, skipped = BatchSharder(0, 1).run(list(range(len(frames))), prepare,
                                          lambda t, pl: proc.submit_preloaded(pl, neg, 6, 0.4, **kw),
                                          collect=lambda t, h: h.result().copy())
    assert not skipped and sorted(res) == list(range(len(frames)))
    for i in range(len(frames)):
        np.testing.assert_array_equal(res[i], want[i])
    assert not np.array_equal(want[0], want[1])
    proc.close()
