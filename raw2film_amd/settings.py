"""Render-settings surface of the reference's GUI, kept verbatim so callers can switch
backends: the two default dicts of MainWindow (gui.py:486-531), their merge order
(gui.py:2183: defaults < image < profile), the preview gating (gui.py:2206-2209) and the
frame formats (data.py:104-125)."""

from __future__ import annotations

DEFAULT_PROFILE_PARAMS = {  # gui.py:486-515 `dflt_prf_params`
    "negative_film": "Kodak Portra 400",
    "print_film": "Fuji Crystal Archive Maxima",
    "red_light": 0,
    "green_light": 0,
    "blue_light": 0,
    "halation": True,
    "sharpness": True,
    "grain": 2,
    "film_format": "135",
    "frame_width": 36,
    "frame_height": 24,
    "grain_size": 6,
    "halation_size": 1.0,
    "halation_green_factor": 0.3,
    "projector_kelvin": 6500,
    "inversion_gamma": 4.0,
    "idealized_curve": False,
    "halation_intensity": 1,
    "shadow_comp": 0,
    "white_clip": False,
    "white_balance": False,
    "sat_adjust": 1,
    "grain_sigma": 0.4,
    "gamma_func": "sRGB",
    "push_pull": 0.0,
    "sharpening_strength": 0.0,
    "sharpening_sigma": 1.0,
    "color_masking": 1.0,
}

DEFAULT_IMAGE_PARAMS = {  # gui.py:516-531 `dflt_img_params`
    "exp_comp": 0,
    "zoom": 1,
    "rotate_times": 0,
    "rotation": 0,
    "exp_kelvin": 6000,
    "profile": "Default",
    "canvas_mode": "No",
    "canvas_scale": 1.0,
    "canvas_ratio": 0.8,
    "highlight_burn": 0,
    "burn_scale": 50,
    "flip": False,
    "tint": 0,
    "chroma_nr": 0,
}

FORMATS = {  # data.py:104-125, (width, height) in mm
    "110": (17, 13), "135-half": (24, 18), "135": (36, 24), "xpan": (65, 24), "120-4.5": (56, 42),
    "120-6": (56, 56), "120": (70, 56), "120-9": (83, 56), "4x5": (127, 101.6), "5x7": (177.8, 127),
    "8x10": (254, 203.2), "11x14": (355.6, 279.4), "super16": (12.42, 7.44), "scope": (24.89, 10.4275),
    "flat": (24.89, 13.454), "academy": (24.89, 18.7), "super8": (5.79, 4.01), "8mm": (4.5, 3.3),
    "65mm": (48.56, 22.1), "IMAX": (70.41, 52.63),
}


def build_processing_params(filmstocks: dict, image_params: dict | None = None, profile_params: dict | None = None,
                            full_preview: bool = True) -> dict:
    """The kwargs `update_preview` hands to `process` (gui.py:2183-2209): merge, resolve stock
    names to objects, "Inversion" print stock -> inversion=True, simplified preview gating."""
    args = {**DEFAULT_PROFILE_PARAMS, **DEFAULT_IMAGE_PARAMS, **(image_params or {}), **(profile_params or {})}
    neg = args["negative_film"]
    args["negative_film"] = filmstocks[neg] if isinstance(neg, str) else neg
    prt = args.get("print_film")
    if isinstance(prt, str):
        if prt == "Inversion":  # gui.py:2191-2192
            args["inversion"] = True
            args["print_film"] = None
        elif prt in ("None", ""):
            args["print_film"] = None
        else:
            args["print_film"] = filmstocks[prt]
    if not full_preview:  # gui.py:2206-2209
        args["sharpness"] = False
        args["grain"] = 0
        args["halation"] = False
    return args
