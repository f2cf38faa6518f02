"""LANCZOS4 up-scale of the uint8 result (utils.resolution_scaling -> cv.resize INTER_LANCZOS4): the host-side weight tables
of the library against the oracle's restatement (no GPU needed), and properties of the oracle itself."""

import numpy as np
import pytest

from oracle import stages as st
from raw2film_amd import _lib


@pytest.mark.parametrize("ssize,dsize", [(10, 25), (100, 333), (2316, 6000), (1544, 4000), (7, 7), (5, 100), (4000, 4001), (1, 9)])
def test_library_tables_match_the_oracle_bit_for_bit(ssize, dsize):
    lib = _lib.load()
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros((dsize, 8), np.int16)
    assert lib.r2f_lanczos4_table(ssize, dsize, ofs.ctypes.data, coef.ctypes.data) == 0
    o, c = st.lanczos4_table(ssize, dsize)
    assert np.array_equal(o, ofs) and np.array_equal(c, coef)


def test_coefficients_are_a_partition_of_unity_up_to_fixed_point_rounding():
    for x in (0.0, 1e-7, 0.125, 0.5, 0.75, 0.999999):
        c = st.lanczos4_coeffs(x)
        assert abs(float(c.sum()) - 1.0) < 1e-6
    assert np.argmax(st.lanczos4_coeffs(0.0)) == 3 and np.argmax(st.lanczos4_coeffs(0.999999)) == 4
    _, coef = st.lanczos4_table(100, 333)
    assert np.abs(coef.astype(int).sum(axis=1) - 2048).max() <= 3


def test_flat_frames_stay_flat_and_the_size_rule_is_the_references():
    flat = np.full((9, 14, 3), (0, 137, 255), np.uint8)
    up = st.resize_lanczos4_u8(flat, 31, 47)
    assert up.shape == (31, 47, 3) and (up == flat[0, 0]).all()
    # utils.resolution_scaling: factor = min over both axes, dsize = (round(w f), round(h f)); only up-scaling here
    img = np.random.default_rng(0).integers(0, 256, (20, 30, 3)).astype(np.uint8)
    assert st.resolution_scaling_u8_up(img, (50, 90)).shape == (50, 75, 3)
    assert st.resolution_scaling_u8_up(img, (20, 30)) is img


def test_identity_size_reproduces_the_image():
    img = np.random.default_rng(1).integers(0, 256, (12, 17, 3)).astype(np.uint8)
    assert np.array_equal(st.resize_lanczos4_u8(img, 12, 17), img)


@pytest.mark.parametrize("ssize,dsize", [(10, 25), (100, 333), (400, 601), (7, 7), (1, 9)])
def test_float_tables_of_the_pre_path_upscale_match_the_oracle(ssize, dsize):
    from oracle import post

    lib = _lib.load()
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros((dsize, 8), np.float32)
    assert lib.r2f_lanczos4_table_f32(ssize, dsize, ofs.ctypes.data, coef.ctypes.data) == 0
    o, c = post.lanczos4_table_f32(ssize, dsize)
    assert np.array_equal(o, ofs) and np.array_equal(c, coef)
    flat = np.full((5, ssize, 3), 0.375, np.float32)
    assert np.abs(post.resize_lanczos4_f32(flat, 5, dsize) - 0.375).max() < 1e-6
