"""CPU oracle for the raw2film post-decode film-emulation path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it, and only as the checker / the timed CPU baseline.  Nothing under
`raw2film_amd/` imports `oracle`; the product path fails loudly when the HIP
extension is missing instead of falling back to this code.

What it is: a NumPy restatement of the reference's per-pixel pipeline
(`/root/reference/src/raw2film/cpu_processor.py:363-407`) stage by stage; every
function cites the reference file:line it follows.

Pinning status:

* PINNED against the reference's own code, executed in the dev container by
  `tools/make_golden.py` (fixtures committed under `tests/golden/`):
    - `oracle.kernels.exponential_blur_kernel`, `compute_halation_kernel`
    - `oracle.kernels.compute_kernel_from_function`, `mtf_kernel_layer`, `mtf_kernel`
    - `oracle.stages.apply_lut_tetrahedral`
* PARITY UNPINNED: `apply_2d_lut`, `log_clip`, `multi_channel_interp`,
  `generate_grain`/`grain_kernel`/`grain_transform` live in the third-party
  package `spectral-film-lut` (`>=0.8.0`, `pyproject.toml:28`, no lockfile, not
  vendored, not installed, no network).  The reference holds no test or golden
  vector for them.  Their restatements here follow the reference's in-tree WGSL
  twins (`shaders/lut_2d.wgsl`, `lut_1d.wgsl`, `noise.wgsl`, `noise_bw.wgsl`,
  `grain.wgsl`) and the LUT layouts in `gpu_processor.py:307-409,565-611`.
  `tools/make_golden_sfl.py` turns that into a one-command job wherever sfl is
  installed: it runs sfl's own functions on this repo's fixture frame and writes
  `tests/golden/sfl.npz`; `tests/test_oracle_golden.py -k sfl` (skipping while the
  file is absent) then pins S1 / S3 / S4 and the grain factor of S6 to it.
* `cv.filter2D` (OpenCV is not installed): restated from its documented
  semantics -- correlation, centred anchor, BORDER_REFLECT_101, float32 in/out --
  and cross-checked against `scipy.ndimage.correlate(mode="mirror")`.
"""

from . import kernels, stages  # noqa: F401
