"""Two builds of libr2f_hip.so in one process: the library's dynamic symbol table is the C ABI of
include/r2f.h and nothing else (tests/test_lib_symbols.py), so a second copy -- a development variant beside the in-tree
library, the tools/ab_render.py use -- keeps its own launchers, planners and kernels instead of resolving them to the first
copy's.  Both copies render the same frames here, interleaved, and agree bit for bit with a single-library render."""

import shutil

import pytest

from helpers import SEED, stocks, synthetic_frame

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_two_copies_of_the_library_render_side_by_side(tmp_path):
    from raw2film_amd import HipProcessor, _lib

    copy = tmp_path / "libr2f_hip_copy.so"
    shutil.copyfile(_lib.LIB_PATH, copy)  # another file -> dlopen maps a second image of the code
    H, W = 160, 256
    fw = 36.0 * W / 12288.0  # both stencils by FFT (100 MP pixel pitch)
    neg, prt, _ = stocks()
    kw = dict(print_film=prt, frame_width=fw, frame_height=fw * H / W, halation_green_factor=0.3, exp_kelvin=6000, color_masking=1.0)
    frame = torch.from_numpy(synthetic_frame(H, W, seed=11)).cuda()

    a = HipProcessor(device=0)
    b = HipProcessor(device=0, lib_path=str(copy))
    assert a.ctx._lib is not b.ctx._lib and a.ctx._lib._handle != b.ctx._lib._handle
    # the two images of the code really are two: the same entry point sits at two addresses
    import ctypes as C

    addr = lambda lib: C.cast(lib.r2f_render, C.c_void_p).value  # noqa: E731
    assert addr(a.ctx._lib) != addr(b.ctx._lib)
    outs = {id(a): [], id(b): []}
    bufs = {id(p): torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for p in (a, b)}  # the same buffers every frame
    for i in range(4):  # interleaved: eager, capture + replay, replay, replay in each copy
        for p in (a, b):
            p.process_array(frame, neg, 6, 0.4, colorspace="linear-rec709", seed=SEED + i, return_float=True, output="device",
                            out=bufs[id(p)], **kw)
            outs[id(p)].append(bufs[id(p)].clone())
    for x, y in zip(outs[id(a)], outs[id(b)]):
        assert torch.equal(x, y)
    assert not torch.equal(outs[id(a)][0], outs[id(a)][1])
    assert a.ctx.render_stats()["replays"] >= 2 and b.ctx.render_stats()["replays"] >= 2
    b.close()
    a.close()
