"""HipProcessor.process(host array, cache=False) streamed in row bands against the same call with upload, render and download one
after the other (stream_bands = 0), under random frame sizes, stage sets, grain modes, host-side geometry, float / uint16 sources,
band counts, tapers, pinned-ring / lent results.  The one-after-the-other path is r2f_render, which the parity tests hold against the
oracle; here the two paths must agree: bit for bit where no stencil stage runs, to the FFT form's rounding where one does (windows
anchored at each band's first row, like a row shard's): uint8, at most one step apart on at most 1e-4 of the samples (direct-form
stencils -- small frames' MTF -- bit for bit again).  Fixed seed; R2F_STREAM_FUZZ_CASES / R2F_STREAM_FUZZ_SEED for a soak."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cases(n=int(os.environ.get("R2F_STREAM_FUZZ_CASES", "10"))):
    rng = np.random.default_rng(int(os.environ.get("R2F_STREAM_FUZZ_SEED", "20261006")))
    out = []
    for _ in range(n):
        W = int(rng.integers(550, 1100)) * 4 + int(rng.integers(0, 4)) * int(rng.integers(0, 3) == 0)  # (a third of them: no multiple of 4)
        H = int(rng.integers(max((1 << 24) // (3 * (W - 8)) + 9, 1400), 3600))  # (the reference's crop arithmetic may trim a row and a column)
        out.append(dict(H=H, W=W, seed=int(rng.integers(0, 2 ** 31)), u16=bool(rng.integers(0, 4) == 0),
                        halation=bool(rng.integers(0, 3) > 0), sharpness=bool(rng.integers(0, 3) > 0), grain=int(rng.integers(0, 3)),
                        flip=bool(rng.integers(0, 4) == 0), turns=int(rng.choice([0, 0, 0, 2])),
                        px_per_mm=int(rng.choice([32, 64, 128, 256])), bands=int(rng.choice([2, 5, 8, 16, 16, 23])),
                        taper=int(rng.integers(0, 4)), ring=int(rng.choice([0, 0, 2])), bw=bool(rng.integers(0, 6) == 0)))
    return out


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"{c['H']}x{c['W']}-{'u16' if c['u16'] else 'f32'}-h{int(c['halation'])}m{int(c['sharpness'])}g{c['grain']}-b{c['bands']}t{c['taper']}-r{c['ring']}")
def test_streamed_and_one_after_the_other_agree(c):
    from raw2film_amd import HipProcessor, filmstock

    stocks = filmstock.builtin_stocks()
    neg, prt = (stocks["Kodak Tri-X 400"], None) if c["bw"] and "Kodak Tri-X 400" in stocks else (stocks["Kodak Portra 400"], stocks["Kodak 2383"])
    rng = np.random.default_rng(c["seed"])
    H, W = c["H"], c["W"]
    if c["u16"]:
        img = rng.integers(0, 65536, (H, W, 3), dtype=np.uint16)
        extra = dict(exposure=float(rng.uniform(-1.0, 1.0)))
    else:
        img = (0.18 * 2.0 ** rng.normal(0.0, 1.5, (H, W, 1)) * rng.uniform(0.6, 1.4, (H, W, 3))).astype(np.float32)
        img[rng.integers(0, H, 50), rng.integers(0, W, 50)] = 16.0
        img[0, 0], img[H - 1, W - 1] = (1e6, -2.0, 70000.0), (-1.0, 65504.0, 3.0)
        extra = {}
    # (the frame's own aspect, in numbers that divide exactly: no aspect crop -- a cropped width need not be a multiple of 4)
    # (`flip` -- the frame format turned against the sensor -- always crops, mostly to below the size that streams: not drawn here;
    # tests/test_gpu_processor.py has it)
    # (the frame format is long side x short side whichever way the image lies, effects.py:77-111)
    kw = dict(print_film=prt, lens_correction=False, seed=c["seed"] & 0xFFFF, frame_width=max(H, W) / c["px_per_mm"],
              frame_height=min(H, W) / c["px_per_mm"], halation=c["halation"], sharpness=c["sharpness"], grain=c["grain"],
              rotate_times=c["turns"], cache=False, **extra)
    proc = HipProcessor(device=0, result_buffers=c["ring"])
    try:
        proc.stream_bands, proc.stream_taper = c["bands"], c["taper"]
        took = []
        inner = proc._process_streamed
        proc._process_streamed = lambda *a, **k: (took.append(inner(*a, **k)), took[-1])[1]
        got = proc.process(img, neg, 6, 0.4, **kw).copy()
        assert len(took) == 1 and took[0] is not None, f"the frame did not stream: {proc.stream_rejected}"
        proc.stream_bands = 0
        want = proc.process(img, neg, 6, 0.4, **kw)
        assert got.shape == want.shape and got.dtype == want.dtype == np.uint8
        if not (c["halation"] or c["sharpness"]):
            np.testing.assert_array_equal(got, want)
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16))
            assert int(d.max()) <= 1 and np.count_nonzero(d) <= 1e-4 * d.size, (int(d.max()), int(np.count_nonzero(d)), d.size)
    finally:
        proc.close()
