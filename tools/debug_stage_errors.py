"""Per-stage error breakdown HIP vs oracle (development aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_inputs, rel_err, stocks, synthetic_frame  # noqa: E402
from oracle import stages as st  # noqa: E402
from raw2film_amd.context import HipContext  # noqa: E402
from test_gpu_parity import dev, from_planes, setup_ctx, to_planes  # noqa: E402

ctx = HipContext(0)
neg, prt, _ = stocks()
for (H, W), scale in (((131, 203), 341.33), ((256, 384), 341.33), ((160, 240), 166.67)):
    p = oracle_inputs(neg, prt, scale)
    img = synthetic_frame(H, W, seed=21)
    ref = st.render(img, p, keep_stages=True)
    params = setup_ctx(ctx, p)
    S = p.stages
    print(f"--- {H}x{W} scale {scale}")
    # isolated: feed the oracle's stage input to each HIP stage
    E = torch.empty((3, H, W), dtype=torch.float32, device="cuda")
    ctx.stage_front(dev(img), params, 0, dst=E)
    print("front->E      isolated rel err (floor 1e-4):", rel_err(from_planes(E), S["exposure"], 1e-4))
    Hh = torch.empty_like(E)
    ctx.stage_stencil(0, to_planes(S["exposure"]), Hh, y0=0, y1=H, H_global=H)
    print("halation conv isolated rel err (floor 1e-4):", rel_err(from_planes(Hh), S["halation"], 1e-4))
    D = torch.empty_like(E)
    ctx.stage_halation(to_planes(S["exposure"]), D, params, y0=0, y1=H, H_global=H)
    print("hal+log+curve isolated abs err:", np.abs(from_planes(D) - S["density"]).max())
    D2 = torch.empty_like(E)
    ctx.stage_mtf(to_planes(S["density"]), D2, params, y0=0, y1=H, H_global=H)
    print("mtf           isolated abs err:", np.abs(from_planes(D2) - S["mtf"]).max())
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    ctx.stage_tail(to_planes(S["mtf"]), params, out_f32=out, y0=0, y1=H, H_global=H)
    print("tail          isolated abs err:", np.abs(out.cpu().numpy() - ref).max())
    # chained
    ctx.stage_halation(E, D, params, y0=0, y1=H, H_global=H)
    print("chained D     abs err:", np.abs(from_planes(D) - S["density"]).max())
    ctx.stage_mtf(D, D2, params, y0=0, y1=H, H_global=H)
    print("chained mtf   abs err:", np.abs(from_planes(D2) - S["mtf"]).max())
    ctx.stage_tail(D2, params, out_f32=out, y0=0, y1=H, H_global=H)
    o = out.cpu().numpy()
    print("chained out   abs err:", np.abs(o - ref).max(), " rel(floor .1):", rel_err(o, ref, 0.1))
    # sensitivity of the output LUT
    g = np.abs(np.diff(p.lut_3d, axis=0)).max() * (p.lut_3d.shape[0] - 1) / 4.0
    print("max |d out / d density| of the 3-D LUT:", g)
