"""HipProcessor as a GUI session uses it: one long-lived processor, the frame kept on the device between renders, tables re-uploaded
only when their parameters change, graphs replayed -- under random sequences of slider moves, geometry changes, other frames, in-place
edits of the decode buffer (announced by a version token or caught by the content fingerprint).  Every result must equal that of a
second processor that reloads everything on every call (cache=False, render_graph = 0).  Fixed seeds; R2F_PSEQ_FUZZ_CASES / _STEPS."""

import os

import numpy as np
import pytest

from oracle import stages as st

from helpers import stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("R2F_PSEQ_FUZZ_CASES", "2")))))
def test_a_session_equals_reloading_everything_every_time(seed):
    from raw2film_amd import HipProcessor

    rng = np.random.default_rng(500 + seed + 104729 * int(os.environ.get("R2F_PSEQ_FUZZ_SEED", "0")))
    neg, prt, bw = stocks()
    srcs = [st.apply_matrix3x3(synthetic_frame(h, w, seed=h), st.REC709_TO_XYZ) for h, w in ((120, 180), (200, 150), (90, 260))]
    versions = [0, 0, 0]
    a, b = HipProcessor(cameras={}, lenses={}, device=0), HipProcessor(cameras={}, lenses={}, device=0)
    b.ctx.set_option("render_graph", 0)
    try:
        film = dict(print_film=prt, exp_kelvin=6000, color_masking=1.0)
        load = dict(frame_width=36, frame_height=24, zoom=1.0, rotate_times=0, resolution=None, max_scale=400.0, canvas_mode="No")
        k, stock, use_version = 0, neg, bool(rng.integers(0, 2))
        for step in range(int(os.environ.get("R2F_PSEQ_FUZZ_STEPS", "40"))):
            op = int(rng.integers(0, 14))
            if op == 0:
                k = int(rng.integers(0, 3))
            elif op == 1:
                film["exp_kelvin"] = float(rng.choice([5000, 6000, 6500]))
            elif op == 2:
                film["halation"] = bool(rng.integers(0, 2))
            elif op == 3:
                film["grain"] = int(rng.integers(0, 3))
            elif op == 4:
                film["print_film"] = prt if rng.integers(0, 3) else None
            elif op == 5:
                load["zoom"] = float(rng.choice([1.0, 1.5]))
            elif op == 6:
                load["rotate_times"] = int(rng.integers(0, 4))
            elif op == 7:
                load["resolution"] = None if rng.integers(0, 2) else (int(rng.integers(50, 220)), int(rng.integers(50, 220)))
            elif op == 8:
                load["canvas_mode"] = str(rng.choice(["No", "Uniform white"]))
                load["canvas_scale"] = 1.2
            elif op == 9:  # the decode buffer is refilled in place: row 0 is always among the fingerprint's rows
                srcs[k][0] *= np.float32(rng.uniform(0.5, 1.5))
                versions[k] += 1
            elif op == 10:
                stock = bw if rng.integers(0, 3) == 0 else neg
            elif op == 11:
                film["sharpness"] = bool(rng.integers(0, 2))
            elif op == 12:
                film["highlight_burn"] = float(rng.choice([0.0, 0.5]))
            sd = int(rng.integers(0, 100000000))
            kw = dict(film, **load, seed=sd)
            ver = dict(src_version=(k, versions[k])) if use_version else {}
            got = a.process(srcs[k], stock, 6, 0.4, **kw, **ver)
            want = b.process(srcs[k], stock, 6, 0.4, cache=False, **kw)
            assert got.shape == want.shape and np.array_equal(got, want), (seed, step, op, k, kw)
        assert a.uploads < b.uploads or a.uploads == b.uploads  # (the session never uploads MORE tables than reloading does)
    finally:
        a.close()
        b.close()
