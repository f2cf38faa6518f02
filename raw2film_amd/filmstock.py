"""Film-stock containers: the counterpart of `spectral_film_lut.FilmSpectral` for this path.

The render path consumes LUTs and a few scalars of a stock, never its spectral model:
  .name, .density_measure, .mtf, .rms_density, .d_ref                     (attributes)
  .get_input_lut(exp_kelvin, tint, exp_comp)      -> (n, n, 3)   cpu_processor.py:160
  .get_density_curve(push_pull, color_masking)    -> (4, m)      cpu_processor.py:182
  .get_grain_curve(scale, adx=False, bw_grain)    -> (4, m)      gpu_processor.py:913
and module-level `create_lut(...)` -> (n, n, n, 3) (cpu_processor.py:232) and
`grain_kernel(pixel_size_mm, grain_size_mm, grain_sigma)` (gpu_processor.py:927).

`spectral_film_lut` (the package that owns the real film data and these generators) is not
available offline, and LUT *generation* is outside the accelerated path, so this module ships

  * `SyntheticStock` -- clearly-labelled ANALYTIC stand-ins ("portra400_like", "k2383_like",
    "bw400_like") that produce LUTs of the right shapes, ranges and smoothness, and
  * `BundleStock` / `load_bundle` -- LUT bundles (.npz) exported on a machine that has
    spectral_film_lut, so real stocks can be rendered without that package on the GPU box.

Nothing here touches the GPU; it is host-side input preparation (NumPy).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32

LUT2D_SIZE = 64
CURVE_SIZE = 1024
GRAIN_LUT_SIZE = 256
LUT3D_SIZE = 33

_XYZ_TO_LIN709 = np.array(
    [[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]]
)


def _cct_to_xy(kelvin: float) -> tuple[float, float]:
    """Planckian-locus approximation (Kim et al.), valid 1667..25000 K."""
    t = float(np.clip(kelvin, 1667.0, 25000.0))
    if t <= 4000:
        x = -0.2661239e9 / t**3 - 0.2343589e6 / t**2 + 0.8776956e3 / t + 0.179910
    else:
        x = -3.0258469e9 / t**3 + 2.1070379e6 / t**2 + 0.2226347e3 / t + 0.240390
    if t <= 2222:
        y = -1.1063814 * x**3 - 1.34811020 * x**2 + 2.18555832 * x - 0.20219683
    elif t <= 4000:
        y = -0.9549476 * x**3 - 1.37418593 * x**2 + 2.09137015 * x - 0.16748867
    else:
        y = 3.0817580 * x**3 - 5.87338670 * x**2 + 3.75112997 * x - 0.37001483
    return x, y


def _smoothstep_curve(loge: np.ndarray, toe: float, gamma: float, dmin: float, dmax: float) -> np.ndarray:
    """Soft-clipped straight line: slope `gamma` through (toe, dmin) with smooth toe and shoulder."""
    span = dmax - dmin
    u = gamma * (loge - toe) / span
    s = np.log1p(np.exp(6.0 * u)) / 6.0  # softplus toe
    s = s - np.log1p(np.exp(6.0 * (s - 1.0))) / 6.0  # softplus shoulder at 1
    return dmin + span * np.clip(s, 0.0, 1.0)


@dataclass(eq=False)  # hashable by identity, like FilmSpectral (lru_cache key at effects.py:165)
class SyntheticStock:
    """Analytic stand-in for a film stock.  NOT real film data."""

    name: str
    density_measure: str = "status_m"  # "bw" switches halation to equal factors (cpu_processor.py:375)
    gamma: tuple = (0.62, 0.65, 0.68)
    dmin: tuple = (0.20, 0.55, 0.85)
    dmax: tuple = (2.6, 3.0, 3.3)
    toe_log_exposure: tuple = (-2.6, -2.6, -2.6)
    crosstalk: float = 0.08
    mtf_f50: tuple | None = (55.0, 75.0, 45.0)
    rms_granularity: tuple | None = (9.0, 8.0, 13.0)  # x1000, 48 um aperture, per layer
    d_ref: tuple = (0.8, 1.2, 1.5)
    is_print: bool = False
    _cache: dict = field(default_factory=dict, repr=False)

    # -- attributes the reference reads ----------------------------------------------------
    @property
    def mtf(self):
        """Iterable of (log1p(cycles/mm) grid, response) per layer, or None (effects.py:174)."""
        if self.mtf_f50 is None:
            return None
        f = np.array([0.0, 1, 2, 5, 10, 20, 30, 40, 50, 60, 80, 100, 150, 200, 400])
        logf = np.log1p(f)
        out = []
        for f50, bump in zip(self.mtf_f50, (0.06, 0.08, 0.03)):
            vals = (1.0 + bump * np.exp(-(((f - 12.0) / 10.0) ** 2))) / (1.0 + (f / f50) ** 2) ** 0.75
            out.append((logf, vals))
        return out

    @property
    def rms_density(self):
        return None if self.rms_granularity is None else np.asarray(self.rms_granularity, dtype=np.float64)

    # -- LUT generators ----------------------------------------------------------------------
    def _layer_matrix(self) -> np.ndarray:
        c = self.crosstalk
        mix = np.array([[1 - 2 * c, c, c], [c, 1 - 2 * c, c], [c, c, 1 - 2 * c]])
        if self.density_measure == "bw":
            mix = np.tile(np.array([[0.25, 0.55, 0.20]]), (3, 1))
        return mix @ _XYZ_TO_LIN709

    def get_input_lut(self, exp_kelvin: float = 6500, tint: float = 0.0, exp_comp: float = 0.0) -> np.ndarray:
        """(n, n, 3) float32: layer exposure per unit (X+Y+Z) as a function of CIE (x, y);
        first axis = x chromaticity (layout note in oracle.stages.apply_2d_lut)."""
        n = LUT2D_SIZE
        g = np.arange(n) / (n - 1)
        x, y = np.meshgrid(g, g, indexing="ij")
        xyz = np.stack([x, y, 1.0 - x - y], axis=-1)
        m = self._layer_matrix()
        wx, wy = _cct_to_xy(exp_kelvin)
        wy = wy * (1.0 + 0.1 * float(tint))
        white = m @ np.array([wx, wy, 1.0 - wx - wy])
        gains = (2.0 ** float(exp_comp)) * wy / np.maximum(white, 1e-6)  # illuminant Y=1 -> equal layer exposure 1
        lut = np.einsum("ij,xyj->xyi", m, xyz) * gains
        return np.maximum(lut, 1e-5).astype(F32)

    def get_density_curve(self, push_pull: float = 0.0, color_masking: float | None = None) -> np.ndarray:
        """(4, m) float32: row 0 = log10 exposure axis (uniform), rows 1..3 = layer densities."""
        m = CURVE_SIZE
        xp = np.linspace(-4.0, 1.5, m)
        mask = 1.0 if color_masking is None else float(color_masking)
        rows = [xp]
        for c in range(3):
            gamma = self.gamma[c] * (1.0 + 0.18 * float(push_pull))
            dmin = 0.08 + (self.dmin[c] - 0.08) * (0.5 + 0.5 * mask)
            rows.append(_smoothstep_curve(xp, self.toe_log_exposure[c] - 0.1 * float(push_pull), gamma, dmin, self.dmax[c]))
        return np.stack(rows).astype(F32)

    def get_grain_curve(self, scale: float, adx: bool = False, bw_grain: bool = False) -> np.ndarray:
        """(4, m) float32: row 0 = density axis, rows 1..3 = grain amplitude per unit Gaussian
        field at `scale` px/mm (Selwyn: sigma_D ~ 1/sqrt(aperture area); 48 um reference)."""
        if self.rms_granularity is None:
            raise ValueError(f"{self.name}: stock has no granularity data")
        m = GRAIN_LUT_SIZE
        xp = np.linspace(0.0, 4.0, m)
        aperture_px = 0.048 * float(scale) * math.sqrt(math.pi) / 2.0
        rows = [xp]
        rms = np.asarray(self.rms_granularity, dtype=np.float64)
        if bw_grain:
            rms = np.full(3, rms.mean())
        for c in range(3):
            shape = np.sqrt(np.clip(xp - 0.5 * self.dmin[c], 0.02, None) / self.d_ref[c])
            rows.append(0.4 * rms[c] / 1000.0 * max(aperture_px, 1.0) * shape)
        return np.stack(rows).astype(F32)

    def __repr__(self):
        return f"SyntheticStock({self.name!r})"


def grain_kernel(pixel_size_mm: float, grain_size_mm: float = 0.006, grain_sigma: float = 0.4):
    """Stand-in for sfl `grain_kernel` (call site gpu_processor.py:927-932): a unit-energy blob
    whose radius follows the mean grain diameter, widened by the log-normal spread
    `grain_sigma`.  Returns None when the grain is finer than a pixel (caller then uses 1x1 ones)."""
    sigma_px = 0.5 * grain_size_mm / pixel_size_mm * math.exp(0.5 * grain_sigma**2) / math.exp(0.5 * 0.4**2)
    if sigma_px < 0.3:
        return None
    r = int(math.ceil(3.0 * sigma_px))
    ax = np.arange(-r, r + 1)
    k = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2.0 * sigma_px**2))
    k /= math.sqrt(float((k**2).sum()))
    return k.astype(F32)


def _srgb_oetf(v: np.ndarray) -> np.ndarray:
    v = np.clip(v, 0.0, 1.0)
    return np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055)


def create_lut(
    negative_film,
    print_film=None,
    mode: str = "print",
    lut_size: int = LUT3D_SIZE,
    red_light: float = 0.0,
    green_light: float = 0.0,
    blue_light: float = 0.0,
    projector_kelvin: float = 6500,
    shadow_comp: float = 0.0,
    sat_adjust: float = 1.0,
    gamma_func: str = "sRGB",
    inversion_gamma: float = 4.0,
    idealized_curve: bool = False,
    inversion: bool = False,
    white_balance: bool = False,
    white_clip: bool = False,
    linear_scaling: float = 4.0,
    color_masking: float | None = None,
    **_,
) -> np.ndarray:
    """Stand-in for sfl `create_lut(negative, print, mode="print", ..., linear_scaling=4)`
    (call site cpu_processor.py:232-253): (n, n, n, 3) float32 in [0, 1], indexed [r, g, b] by
    negative density / linear_scaling, returning display-referred RGB."""
    if isinstance(negative_film, BundleStock):
        return negative_film.output_lut(print_film)
    n = int(lut_size)
    axis = np.linspace(0.0, linear_scaling, n)
    d = np.stack(np.meshgrid(axis, axis, axis, indexing="ij"), axis=-1)  # negative densities
    neg_curve = negative_film.get_density_curve(0.0, color_masking)
    grey = math.log10(0.18)
    d_mid = np.array([np.interp(grey, neg_curve[0], neg_curve[1 + c]) for c in range(3)])  # 18 % grey on the negative
    lights = np.array([red_light, green_light, blue_light], dtype=np.float64)
    if print_film is not None and not inversion:
        # printer exposure through the negative; calibrated so that 18 % grey prints to 18 % transmittance
        curve = print_film.get_density_curve()
        pdmin = np.array(print_film.dmin) * 0.9
        dp = []
        for c in range(3):
            target = -grey + pdmin[c]
            log_h_mid = np.interp(target, curve[1 + c], curve[0])  # curve is monotone in log H
            log_h = -(d[..., c] - d_mid[c]) + log_h_mid + 0.025 * lights[c]
            dp.append(np.interp(log_h, curve[0], curve[1 + c]))
        dp = np.stack(dp, axis=-1) - pdmin
        lin = 10.0 ** (-dp) * (1.0 + float(shadow_comp))
        wx, wy = _cct_to_xy(projector_kelvin)
        w65x, w65y = _cct_to_xy(6500)
        tint = (_XYZ_TO_LIN709 @ np.array([wx / wy, 1.0, (1 - wx - wy) / wy])) / (
            _XYZ_TO_LIN709 @ np.array([w65x / w65y, 1.0, (1 - w65x - w65y) / w65y])
        )
        lin = lin * tint
    else:
        # negative-only / "Inversion": undo each layer's straight-line gamma around 18 % grey
        g = np.array(negative_film.gamma) * (4.0 / float(inversion_gamma) if inversion else 1.0)
        lin = 0.18 * 10.0 ** ((d - d_mid) / g)
    luma = lin @ np.array([0.2126729, 0.7151522, 0.0721750])
    lin = luma[..., None] + float(sat_adjust) * (lin - luma[..., None])
    if white_clip:
        lin = lin / max(float(lin.max()), 1e-6)
    out = _srgb_oetf(lin) if gamma_func == "sRGB" else np.clip(lin, 0, 1) ** (1 / 2.2)
    return np.clip(out, 0.0, 1.0).astype(F32)


class BundleStock:
    """A stock backed by LUT arrays exported from spectral_film_lut (see `save_bundle`).
    Parameters of the get_* calls are accepted for interface parity but the stored arrays
    are returned as they are (they were generated for one set of parameters)."""

    def __init__(self, name: str, arrays: dict):
        self.name = name
        self._a = {k: np.asarray(v) for k, v in arrays.items()}
        self.density_measure = str(self._a.get("density_measure", "status_m"))
        self.d_ref = tuple(np.asarray(self._a.get("d_ref", (1.0, 1.0, 1.0)), dtype=float).tolist())
        self.rms_density = self._a.get("rms_density")
        if "mtf_logf" in self._a:
            self.mtf = [(self._a["mtf_logf"][c], self._a["mtf_vals"][c]) for c in range(3)]
        else:
            self.mtf = None

    def get_input_lut(self, *_, **__):
        return self._a["lut_2d"].astype(F32)

    def get_density_curve(self, *_, **__):
        return self._a["lut_1d"].astype(F32)

    def get_grain_curve(self, *_, **__):
        return self._a["grain_lut"].astype(F32)

    def output_lut(self, print_film=None):
        return self._a["lut_3d"].astype(F32)

    def __repr__(self):
        return f"BundleStock({self.name!r})"


def save_bundle(path: str, **arrays) -> None:
    """Keys: lut_2d (n,n,3), lut_1d (4,m), lut_3d (n,n,n,3), grain_lut (4,m), optional
    grain_kernel, mtf_logf (3,k), mtf_vals (3,k), rms_density, d_ref, density_measure."""
    np.savez_compressed(path, **arrays)


def load_bundle(path: str, name: str | None = None) -> BundleStock:
    with np.load(path, allow_pickle=False) as z:
        return BundleStock(name or str(path), {k: z[k] for k in z.files})


def builtin_stocks() -> dict:
    """The `filmstocks` dict MainWindow receives from sfl's load_ui (gui.py:194), with the GUI's
    default names (gui.py:486-488) mapped to synthetic stand-ins."""
    portra = SyntheticStock("Kodak Portra 400 (synthetic stand-in)")
    k2383 = SyntheticStock(
        "Kodak 2383 (synthetic stand-in)",
        density_measure="status_a",
        gamma=(2.9, 3.0, 3.1),
        dmin=(0.05, 0.05, 0.06),
        dmax=(3.8, 3.9, 4.0),
        toe_log_exposure=(-1.15, -1.15, -1.15),
        mtf_f50=None,
        rms_granularity=None,
        is_print=True,
    )
    maxima = SyntheticStock(
        "Fuji Crystal Archive Maxima (synthetic stand-in)",
        density_measure="status_a",
        gamma=(2.6, 2.7, 2.8),
        dmin=(0.06, 0.06, 0.07),
        dmax=(2.5, 2.6, 2.6),
        toe_log_exposure=(-1.1, -1.1, -1.1),
        mtf_f50=None,
        rms_granularity=None,
        is_print=True,
    )
    bw = SyntheticStock(
        "Kodak Tri-X 400 (synthetic stand-in)",
        density_measure="bw",
        gamma=(0.6, 0.6, 0.6),
        dmin=(0.25, 0.25, 0.25),
        dmax=(2.4, 2.4, 2.4),
        mtf_f50=(60.0, 60.0, 60.0),
        rms_granularity=(17.0, 17.0, 17.0),
        d_ref=(1.0, 1.0, 1.0),
    )
    return {
        "Kodak Portra 400": portra,
        "Kodak 2383": k2383,
        "Fuji Crystal Archive Maxima": maxima,
        "Kodak Tri-X 400": bw,
        "portra400_like": portra,
        "k2383_like": k2383,
        "bw400_like": bw,
    }
