"""r2f_render's graph cache under random SEQUENCES of what a session does: frames of a few sizes into a pool of output buffers,
new seeds, stage sets switched on and off, tables and stencils re-uploaded, options changed, buffers that come by once or many
times.  The same sequence runs on a second context with render_graph = 0; every frame must be bit-identical -- a stale graph (a
table that moved, a scratch buffer that grew, a seed frozen at capture time, an evicted entry launched again) shows up as a frame
that differs.  Fixed seeds; R2F_SEQ_FUZZ_CASES / _STEPS for a soak."""

import os

import numpy as np
import pytest

from helpers import oracle_inputs, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("R2F_SEQ_FUZZ_CASES", "3")))))
def test_random_session_replays_equal_eager_launches(seed):
    from raw2film_amd import _lib, filmstock, stencils
    from raw2film_amd.context import HipContext
    from test_gpu_parity import setup_ctx

    rng = np.random.default_rng(1000 + seed + 7919 * int(os.environ.get("R2F_SEQ_FUZZ_SEED", "0")))
    neg, prt, _ = stocks()
    shapes = [(96, 160), (150, 222), (301, 417)]
    scale = float(rng.choice([97.3, 341.33]))
    p = oracle_inputs(neg, prt, scale, seed=1)
    ctxs = [HipContext(0), HipContext(0)]
    ctxs[1].set_option("render_graph", 0)
    try:
        params = [setup_ctx(c, p) for c in ctxs]
        frames = {s: torch.from_numpy(synthetic_frame(s[0], s[1], seed=s[0])).cuda() for s in shapes}
        pools = [{s: [torch.empty((s[0], s[1], 3), dtype=torch.float32, device="cuda") for _ in range(10)] for s in shapes} for _ in ctxs]
        luts = [np.ascontiguousarray(p.lut_3d), np.ascontiguousarray(p.lut_3d[..., ::-1])]
        mtfs = [stencils.mtf_stencil(neg, scale, 0.0, 1.0), stencils.mtf_stencil(neg, scale * 0.8, 0.0, 1.0)]
        shape, flags_off = shapes[0], 0
        hot = [0, 1]  # buffer indices that keep coming back
        replays0 = 0
        for step in range(int(os.environ.get("R2F_SEQ_FUZZ_STEPS", "70"))):
            op = rng.integers(0, 20)
            if op == 0:
                shape = shapes[int(rng.integers(0, len(shapes)))]
            elif op == 1:
                k = int(rng.integers(0, 2))
                for c in ctxs:
                    c.set_lut3d(luts[k])
            elif op == 2:
                k = int(rng.integers(0, 2))
                for c in ctxs:
                    c.set_kernel(1, mtfs[k])
            elif op == 3:
                v = int(rng.integers(0, 2))
                for c in ctxs:
                    c.set_option("stencil_fft", v)
            elif op == 4:
                flags_off = int(rng.choice([0, 2, 4, 8, 2 | 4]))  # halation / MTF / grain switched off
            elif op == 5:
                hot = [int(v) for v in rng.integers(0, 10, size=2)]
            elif op == 6:
                v = int(rng.choice([0, 256, 512]))
                for c in ctxs:
                    c.set_option("stencil_fft_window", v)
            # a frame: mostly into the buffers that keep coming back, sometimes into one seen once
            buf = hot[int(rng.integers(0, 2))] if rng.integers(0, 4) else int(rng.integers(0, 10))
            sd = int(rng.integers(0, 2**32))
            outs = []
            for c, prm, pool in zip(ctxs, params, pools):
                q = _lib.Params.from_buffer_copy(prm)
                q.seed = sd
                q.flags &= ~flags_off
                o = pool[shape][buf]
                o.zero_()
                c.render(frames[shape], q, out_f32=o)
                outs.append(o)
            assert torch.equal(outs[0], outs[1]), (seed, step, int(op), shape, buf, flags_off)
        stats = ctxs[0].render_stats()
        assert stats["replays"] > 0 and ctxs[1].render_stats()["replays"] == 0, stats
    finally:
        for c in ctxs:
            c.close()
