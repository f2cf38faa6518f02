"""Device counting pass of the caller-side histogram against the oracle (bit-exact integers)."""

import numpy as np
import pytest

from oracle import histogram as oh

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from raw2film_amd.context import HipContext

    c = HipContext(0)
    yield c
    c.close()


def frames():
    rng = np.random.default_rng(5)
    yield "noise", rng.integers(0, 256, (257, 1031, 3)).astype(np.uint8)
    yield "flat", np.full((512, 768, 3), (7, 7, 250), dtype=np.uint8)  # every pixel into 2 bins: worst case for atomics
    yield "one pixel", np.array([[[1, 2, 3]]], dtype=np.uint8)
    yield "5 px", rng.integers(0, 256, (1, 5, 3)).astype(np.uint8)  # 15 bytes: only the tail loop
    yield "17 bytes past a chunk", rng.integers(0, 256, (1, 4096 // 3 * 16 + 11, 3)).astype(np.uint8)
    img = np.clip(rng.normal(120, 40, (1200, 1800, 3)), 0, 255).astype(np.uint8)
    yield "photo-like", img


@pytest.mark.parametrize("name,img", list(frames()), ids=[n for n, _ in frames()])
def test_counts_bit_exact(ctx, name, img):
    got = ctx.histogram_counts(torch.from_numpy(img).cuda()).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (3, 256)
    assert np.array_equal(got, oh.counts(img))
    assert got.sum() == img.size


def test_counts_are_overwritten_not_accumulated(ctx):
    img = torch.from_numpy(np.full((64, 64, 3), 9, dtype=np.uint8)).cuda()
    a = ctx.histogram_counts(img).cpu().numpy()
    b = ctx.histogram_counts(img).cpu().numpy()
    assert np.array_equal(a, b) and a[:, 9].tolist() == [4096] * 3


def test_empty_frame(ctx):
    got = ctx.histogram_counts(torch.empty((0, 8, 3), dtype=torch.uint8, device="cuda")).cpu().numpy()
    assert not got.any()


def test_generate_histogram_matches_oracle(ctx):
    from raw2film_amd import histogram as ph

    rng = np.random.default_rng(11)
    img = np.clip(rng.normal(100, 50, (333, 500, 3)) * np.array([1.0, 0.8, 1.2]), 0, 255).astype(np.uint8)
    for h in (80, 100):
        got = ph.generate_histogram(img, ph.MIX_TABLE, h, ctx=ctx)
        assert np.array_equal(got, oh.generate_histogram(img, ph.MIX_TABLE, h))
        got_dev = ph.generate_histogram(torch.from_numpy(img).cuda(), ph.MIX_TABLE, h, ctx=ctx)
        assert np.array_equal(got_dev, got)


def test_full_size_counts_sum_and_checksum(ctx):
    """100 MP: every byte lands in exactly one bin; channel sums agree with torch's own reduction."""
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randint(0, 256, (8192, 12288, 3), dtype=torch.uint8, device="cuda", generator=g)
    counts = ctx.histogram_counts(img)
    assert int(counts.sum()) == img.numel()
    bins = torch.arange(256, device="cuda", dtype=torch.int64)
    for c in range(3):
        assert int((counts[c].to(torch.int64) * bins).sum()) == int(img[..., c].to(torch.int64).sum())


def test_processor_histogram_of_the_last_frame(ctx):
    from raw2film_amd import HipProcessor, filmstock
    from raw2film_amd import histogram as ph
    from helpers import synthetic_frame

    stocks = filmstock.builtin_stocks()
    proc = HipProcessor(device=0)
    try:
        with pytest.raises(ValueError):
            proc.generate_histogram()
        img = synthetic_frame(120, 180, seed=3)
        out = proc.process(img, stocks["Kodak Portra 400"], 6, 0.4, print_film=stocks["Kodak 2383"], seed=5)
        hist = proc.generate_histogram(height=80)
        assert hist.shape == (80, 256, 4)
        assert np.array_equal(hist, oh.generate_histogram(out, ph.MIX_TABLE, 80))
        assert np.array_equal(proc.generate_histogram(out, height=80), hist)
    finally:
        proc.close()
