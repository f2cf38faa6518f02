"""Synthetic decoded frames for tests and benchmarks: linear
scene-referred Rec.709 RGB, log-normal luminance (sigma 1.5 stops around 18 % grey), per-channel
colour variation, 0.1 % specular pixels at 16.0 so that halation has something to spread."""

from __future__ import annotations

import numpy as np

CONFIGS = {  # name -> (W, H) of BASELINE.json's configs, 3:2 frames on 36 x 24 mm
    "cfg1_512": (512, 512),
    "cfg2_24mp": (6000, 4000),
    "cfg3_45mp": (8256, 5504),
    "cfg4_100mp": (12288, 8192),
}
BENCH_SEED = 20260630


def synthetic_frame(H: int, W: int, seed: int = 1234) -> np.ndarray:
    """float32 (H, W, 3) linear Rec.709, clipped to [0, 65504] (gpu_processor.py:275)."""
    rng = np.random.default_rng(seed)
    img = (0.18 * 2.0 ** rng.normal(0.0, 1.5, (H, W, 1))).astype(np.float32)
    img = img * rng.uniform(0.6, 1.4, (H, W, 3)).astype(np.float32)
    img[rng.uniform(size=(H, W)) < 0.001] = 16.0
    return np.clip(img, 0.0, 65504.0).astype(np.float32)


def synthetic_frame_device(H: int, W: int, seed: int = 1234, device="cuda", layout="hwc", kind="noise"):
    """Same distribution generated on the GPU (for frames too large to build on the host quickly).
    Not bit-identical to `synthetic_frame`; used for throughput runs only.

    kind="noise": every pixel independent (the headline frames: the worst case for the LUT gathers, neighbouring
    pixels share no texel).  kind="smooth": the same luminance / colour statistics drawn on a 64 x coarser grid and
    interpolated, plus 2 % per-pixel noise -- what the LUT stages see on a photograph (bench.py --frame smooth)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if kind == "smooth":
        h, w = max(H // 64, 2), max(W // 64, 2)
        stops = torch.nn.functional.interpolate(1.5 * torch.randn((1, 1, h, w), generator=g, device=device), size=(H, W),
                                                mode="bicubic", align_corners=False)[0].permute(1, 2, 0)
        tint = torch.nn.functional.interpolate(0.6 + 0.8 * torch.rand((1, 3, h, w), generator=g, device=device), size=(H, W),
                                               mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        img = 0.18 * torch.exp2(stops) * tint * (1.0 + 0.02 * torch.randn((H, W, 3), generator=g, device=device))
    elif kind == "noise":
        lum = 0.18 * torch.exp2(1.5 * torch.randn((H, W, 1), generator=g, device=device))
        img = lum * (0.6 + 0.8 * torch.rand((H, W, 3), generator=g, device=device))
    else:
        raise ValueError(f"unknown synthetic frame kind {kind!r}")
    spec = torch.rand((H, W), generator=g, device=device) < 0.001
    img[spec] = 16.0
    img = img.clamp_(0.0, 65504.0).contiguous()
    if layout == "chw":
        img = img.permute(2, 0, 1).contiguous()
    return img
