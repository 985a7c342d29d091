"""ctypes binding of libgoofer_hip.so (the C ABI declared in include/goofer_hip.h).

There is no CPU fallback: a missing or unloadable library raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgoofer_hip.so")

# numpy mirror of goofer_note_params (C layout, 88 bytes)
NOTE_PARAMS = np.dtype({
    "names": ["pitch_shift", "formant_shift", "f_shift", "uv_strength", "breath_strength", "normalize",
              "apply_brightness", "cut_below_f0", "mix_harm", "mix_breath", "mix_unvoiced", "volume", "seed"],
    "formats": ["<f4", "<f4", ("<f8", 4), "<f4", "<f4", "<f4", "<i4", "<i4", "<f4", "<f4", "<f4", "<f4", ("<u4", 2)],
    "offsets": [0, 4, 8, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76],
    "itemsize": 88,
})


class Batch(C.Structure):
    """goofer_batch"""
    _fields_ = [
        ("n_notes", C.c_int32), ("n_bins", C.c_int32), ("ld", C.c_int32), ("reserved", C.c_int32),
        ("total_frames", C.c_int64), ("total_samples", C.c_int64), ("total_env_rows", C.c_int64),
        ("sample_off", C.c_void_p), ("frame_off", C.c_void_p), ("env_off", C.c_void_p),
        ("env", C.c_void_p), ("formants", C.c_void_p), ("f0", C.c_void_p), ("mask", C.c_void_p),
        ("phi", C.c_void_p), ("params", C.c_void_p), ("seed", C.c_uint64),
        ("transition_sigma", C.c_float), ("reserved2", C.c_float),
        ("harm", C.c_void_p), ("uv", C.c_void_p), ("bre", C.c_void_p), ("rec", C.c_void_p), ("mix", C.c_void_p),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "goofer_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "goofer_destroy": (None, [C.c_void_p]),
    "goofer_last_error": (C.c_char_p, [C.c_void_p]),
    "goofer_version": (C.c_char_p, []),
    "goofer_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "goofer_reserve": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "goofer_rfft_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "goofer_irfft_ola": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_void_p]),
    "goofer_pulse_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "goofer_gauss_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "goofer_warp_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_void_p]),
    "goofer_knot_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p]),
    "goofer_synth_batch": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_void_p]),
    "goofer_debug_table": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "goofer_debug_fetch": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "goofer_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "goofer_profile_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "goofer_profile_stage_name": (C.c_char_p, [C.c_int]),
}

_lib = None


def load(path: str = LIB_PATH):
    """Load the shared library and type every export.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension has not been built (run `python -m goofer_amd.build`). "
            "goofer_amd has no CPU fallback.")
    # torch first: it carries its own libamdhip64; loading ours afterwards binds to that same runtime
    # (the other order puts two HIP runtimes in the process and hipGetDeviceCount fails)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
