"""raw2film_amd -- MI355X-native (gfx950) backend for raw2film's post-decode film-emulation
render path: hand-written HIP kernels behind a C ABI (include/r2f.h), called through ctypes
with torch-ROCm tensors as device buffers.  See DESIGN.md.

Importing the package does not need a GPU; constructing `HipProcessor` / `HipContext` does,
and raises if the HIP library or the GPU is missing (there is no CPU fallback).
"""

from . import filmstock, histogram, settings, stencils  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name == "HipProcessor":
        from .hip_processor import HipProcessor

        return HipProcessor
    if name == "HipContext":
        from .context import HipContext

        return HipContext
    if name in ("RowShardedRenderer", "BatchSharder"):
        from . import sharding

        return getattr(sharding, name)
    raise AttributeError(name)
