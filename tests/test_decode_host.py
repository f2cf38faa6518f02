"""The hand-off from RAW decoding (raw_conversion.py:50-52): the auto-exposure statistic of raw2film_amd.decode and of the oracle
against the reference's own calc_exposure (tests/golden/exposure.npz, tools/make_golden_exposure.py), and the host form of the
uint16 -> float32 conversion the device kernel reproduces."""

import os

import numpy as np
import pytest

from oracle import post
from raw2film_amd import decode

METAS = [None,  # the variants tools/make_golden_exposure.py ran the reference with, in its order
         {"EXIF:FNumber": 8.0, "EXIF:ISO": 100, "EXIF:ExposureTime": 1 / 250},
         {"EXIF:FNumber": 1.8, "EXIF:ISO": 3200, "EXIF:ExposureTime": 1 / 30},
         {"EXIF:FNumber": "undef", "EXIF:ISO": 400, "EXIF:ExposureTime": 0.01},
         {"EXIF:ISO": 200, "EXIF:ExposureTime": 2.0},
         {"EXIF:FNumber": 0, "EXIF:ISO": 800, "EXIF:ExposureTime": 1 / 1000}]


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "exposure.npz"))
    assert [repr(m) for m in METAS] == [str(m) for m in g["metas"]]
    return g


def test_auto_exposure_matches_the_reference_bit_for_bit(golden):
    for i in range(int(golden["n"])):
        si, mi = golden[f"case_{i}"]
        u16 = golden[f"frame_{si}"]
        want = float(golden[f"exp_{i}"])
        assert decode.auto_exposure(u16, metadata=METAS[mi]) == want, (i, si, mi)  # LibRaw's uint16 frame
        rgb = u16.astype(np.float32) / 65535.0
        assert decode.auto_exposure(rgb, metadata=METAS[mi]) == want  # ... and the float frame upstream measures
        assert post.calc_exposure(rgb, metadata=METAS[mi]) == want  # the oracle's restatement


def test_exposure_root_follows_the_exif_branches():
    assert decode.exposure_root(None) == 3
    assert decode.exposure_root(METAS[1]) == pytest.approx(np.sqrt(64 / 100 * 250) + 1)
    for m in METAS[3:]:  # "undef", missing and zero apertures all fall back to f/4
        assert decode.exposure_root(m) == pytest.approx(np.sqrt(16 / m["EXIF:ISO"] / m["EXIF:ExposureTime"]) + 1)


def test_host_decode_is_the_two_lines_of_raw_to_linear():
    rng = np.random.default_rng(4)
    u16 = rng.integers(0, 65536, (33, 47, 4)).astype(np.uint16)
    for stops in (0.0, 0.657791852173465, -1.25, 3.0):
        want = u16[..., :3].astype(np.float32) / 65535.0
        want *= 2**stops
        got = decode.decode_u16_host(u16, stops)
        assert got.dtype == np.float32 and np.array_equal(got, want) and np.array_equal(post.decode_u16(u16, stops), want)
    assert decode.decode_u16_host(np.full((1, 1, 3), 65535, np.uint16), 17.0).max() == 65504.0  # the upload clamp
    assert decode.exposure_factor(0.5).dtype == np.float32
