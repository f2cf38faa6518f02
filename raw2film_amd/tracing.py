"""Per-stage device timing for the render path (the reference only prints wall-clock seconds around an export,
gui.py:2339-2352).  `TimedBackend` wraps a stage backend of raw2film_amd.sharding and brackets every stage call with
events on the launch stream; `summary()` gives average milliseconds per stage.  Used by bench.py for the live
roofline of the dominant kernel and handy for "where did the frame time go" questions:

    backend = TimedBackend(HipStageBackend(ctx, params, ...))
    RowShardedRenderer(backend, H, W, halation=True, mtf=True, rank=0, world=1).render(frame, out_f32=out)
    print(backend.summary())          # {'front': 0.43, 'halation': 15.6, 'mtf': 5.5, 'tail': 1.8}
"""

from __future__ import annotations

STAGES = ("front", "front_split", "halation", "mtf", "grain", "burn_sums", "burn_map", "tail", "tail_field", "front_to_output")


class TimedBackend:
    def __init__(self, backend):
        import torch

        self._torch = torch
        self._be = backend
        self.events: dict[str, list] = {}

    def __getattr__(self, name):
        attr = getattr(self._be, name)
        if name not in STAGES or not callable(attr):
            return attr

        def timed(*args, **kwargs):
            e0 = self._torch.cuda.Event(enable_timing=True)
            e1 = self._torch.cuda.Event(enable_timing=True)
            e0.record()
            result = attr(*args, **kwargs)
            e1.record()
            self.events.setdefault(name, []).append((e0, e1))
            return result

        return timed

    def reset(self):
        self.events.clear()

    def summary(self) -> dict[str, float]:
        """Average milliseconds per call of each stage (synchronises the device)."""
        self._torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items() if v}
