"""ctypes binding of libgoofer_hip.so (the C ABI declared in include/goofer_hip.h).

There is no CPU fallback: a missing or unloadable library raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOOFER_HIP_LIB") or os.path.join(HERE, "libgoofer_hip.so")   # override: kernel experiments only

# numpy mirror of goofer_note_params (C layout, 112 bytes; checked against goofer_sizeof)
NOTE_PARAMS = np.dtype({
    "names": ["pitch_shift", "formant_shift", "f_shift", "uv_strength", "breath_strength", "normalize",
              "apply_brightness", "cut_below_f0", "mix_harm", "mix_breath", "mix_unvoiced", "volume", "seed",
              "vol_jitter_harm", "vol_jitter_breath", "subharm_weight", "f0_jitter", "subharm_f0_jitter"],
    "formats": ["<f4", "<f4", ("<f8", 4), "<f4", "<f4", "<f4", "<i4", "<i4", "<f4", "<f4", "<f4", "<f4", ("<u4", 2),
                "<f4", "<f4", "<f4", "<f8", "<f8"],
    "offsets": [0, 4, 8, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 84, 88, 92, 96, 104],
    "itemsize": 112,
})


# numpy mirror of goofer_note_plan (align=True reproduces the C layout; checked against goofer_sizeof)
NOTE_PLAN = np.dtype([
    ("knot_off", "<i8"), ("edit_off", "<i8"), ("tap_off", "<i8"), ("env_off", "<i8"), ("src_sample_off", "<i8"),
    ("out_sample_off", "<i8"), ("ylen", "<i8"), ("bend_off", "<i8"),
    ("es_amount", "<f8"), ("vel_factor", "<f8"), ("pitch_m", "<f8"), ("pitch_t", "<f8"), ("tick_dt", "<f8"),
    ("fst", "<f8", 4),
    ("K", "<i4"), ("lerp_plan", "<i4"), ("n_src_rows", "<i4"), ("reverse", "<i4"), ("row_lo", "<i4"), ("n_edit", "<i4"),
    ("tilt", "<i4"), ("es_mode", "<i4"), ("es_taps_off", "<i4"), ("es_radius", "<i4"), ("fw_plan", "<i4"),
    ("n_out_rows", "<i4"), ("env_f64", "<i4"), ("n_out", "<i4"), ("n_pre", "<i4"), ("s_pre", "<i4"), ("s_tail", "<i4"),
    ("tail_len", "<i4"), ("want_samples", "<i4"), ("n_before_vel", "<i4"), ("pre_new", "<i4"), ("vel_active", "<i4"),
    ("force_voiced", "<i4"), ("n_bend", "<i4"), ("reserved", "<i4"),
    ("fry_hz", "<f8"), ("fry_dir", "<i4"), ("fry_const_lo", "<i4"), ("fry_const_hi", "<i4"), ("fry_glide_lo", "<i4"),
    ("fry_glide_hi", "<i4"), ("fry_a", "<i4"), ("fry_b", "<i4"), ("fry_fade", "<i4"), ("pd_on", "<i4"), ("reserved4", "<i4"),
    ("pd_base", "<f8"),
], align=True)

# goofer_plan_request / goofer_plan_geometry (host-side planner)
PLAN_REQUEST = np.dtype([
    ("offset", "<f8"), ("length", "<f8"), ("consonant", "<f8"), ("cutoff", "<f8"), ("vel_factor", "<f8"), ("fry", "<f8"),
    ("fry_glide", "<f8"), ("ylen", "<i8"), ("sr", "<i4"), ("n_src_frames", "<i4"), ("loop_mode", "<i4"), ("reverse", "<i4"),
    ("tracks", "<u8", 4), ("track_len", "<i4", 4), ("fst_skip", "<i4", 4),
], align=True)
PLAN_GEOMETRY = np.dtype([
    ("start_sample", "<i8"), ("consonant_sample", "<i8"), ("end_sample", "<i8"), ("tap_off", "<i8"), ("vel_factor", "<f8"),
    ("status", "<i4"), ("start_frame", "<i4"), ("consonant_frame", "<i4"), ("end_frame", "<i4"), ("n_rows", "<i4"),
    ("n_out_rows", "<i4"), ("row_lo", "<i4"), ("row_hi", "<i4"), ("env_f64", "<i4"),
    ("n_out", "<i4"), ("n_pre", "<i4"), ("s_pre", "<i4"), ("s_tail", "<i4"), ("tail_len", "<i4"), ("want_samples", "<i4"),
    ("n_before_vel", "<i4"), ("pre_new", "<i4"), ("vel_active", "<i4"),
    ("fry_dir", "<i4"), ("fry_const_lo", "<i4"), ("fry_const_hi", "<i4"), ("fry_glide_lo", "<i4"), ("fry_glide_hi", "<i4"),
    ("fry_a", "<i4"), ("fry_b", "<i4"), ("fry_fade", "<i4"), ("reserved", "<i4"),
], align=True)

# goofer_onepole_job / goofer_post_note
ONEPOLE_JOB = np.dtype([
    ("src_off", "<i8"), ("dst_off", "<i8"), ("f0_off", "<i8"), ("n", "<i4"), ("order", "<i4"), ("highpass", "<i4"),
    ("f0_mode", "<i4"), ("cutoff_factor", "<f4"), ("reserved", "<i4"),
], align=True)
POST_NOTE = np.dtype([
    ("su_off", "<i8"), ("sj_off", "<i8"), ("sa_off", "<i8"), ("su_gain", "<f4"), ("sj_mix", "<f4"), ("sa_mix", "<f4"),
    ("sd_strength", "<f4"), ("tension", "<f4"), ("pitch_dyn", "<f4"), ("fry_a", "<i4"), ("fry_b", "<i4"), ("fry_fade", "<i4"),
    ("reserved", "<i4"),
], align=True)


class Assembly(C.Structure):
    """goofer_assembly"""
    _fields_ = [
        ("n_notes", C.c_int32), ("n_bins", C.c_int32), ("ld", C.c_int32), ("sr", C.c_int32), ("max_K", C.c_int32),
        ("reserved", C.c_int32),
        ("total_edit_rows", C.c_int64), ("total_out_rows", C.c_int64), ("total_samples", C.c_int64),
        ("notes", C.c_void_p), ("knots", C.c_void_p), ("lerp_idx", C.c_void_p), ("lerp_w0", C.c_void_p), ("lerp_w1", C.c_void_p),
        ("tilts", C.c_void_p), ("es_taps", C.c_void_p), ("fw_lo", C.c_void_p), ("fw_hi", C.c_void_p), ("fw_frac", C.c_void_p),
        ("tap_idx", C.c_void_p), ("tap_w", C.c_void_p), ("fst_tracks", C.c_void_p), ("mask_src", C.c_void_p), ("bend", C.c_void_p),
        ("edit_rows", C.c_void_p), ("env_out", C.c_void_p), ("f0_out", C.c_void_p), ("mask_out", C.c_void_p),
        ("bend_out", C.c_void_p), ("any_fry", C.c_int32), ("reserved5", C.c_int32),
        ("f0_mul", C.c_void_p), ("f0_mul_out", C.c_void_p),
    ]


class Post(C.Structure):
    """goofer_post"""
    _fields_ = [
        ("n_notes", C.c_int32), ("reserved", C.c_int32), ("total_samples", C.c_int64),
        ("sample_off", C.c_void_p), ("sample_off_host", C.c_void_p), ("params", C.c_void_p), ("notes", C.c_void_p),
        ("f0", C.c_void_p), ("mask", C.c_void_p), ("bend", C.c_void_p),
        ("harm", C.c_void_p), ("uv", C.c_void_p), ("bre", C.c_void_p), ("su_harm", C.c_void_p), ("sj_harm", C.c_void_p),
        ("sa_uv", C.c_void_p), ("sa_bre", C.c_void_p), ("mix", C.c_void_p),
    ]


class Batch(C.Structure):
    """goofer_batch"""
    _fields_ = [
        ("n_notes", C.c_int32), ("n_bins", C.c_int32), ("ld", C.c_int32), ("mix_only", C.c_int32),
        ("total_frames", C.c_int64), ("total_samples", C.c_int64), ("total_env_rows", C.c_int64),
        ("sample_off", C.c_void_p), ("frame_off", C.c_void_p), ("env_off", C.c_void_p),
        ("env", C.c_void_p), ("formants", C.c_void_p), ("f0", C.c_void_p), ("mask", C.c_void_p),
        ("phi", C.c_void_p), ("env_noise", C.c_void_p), ("params", C.c_void_p), ("seed", C.c_uint64),
        ("transition_sigma", C.c_float), ("vol_jitter_speed", C.c_float),
        ("noise_f0", C.c_void_p), ("noise_vol_h", C.c_void_p), ("noise_vol_b", C.c_void_p), ("noise_subharm", C.c_void_p),
        ("f0_jitter_sigma", C.c_float), ("vol_jitter_sigma", C.c_float),
        ("subharm_ratio", C.c_double), ("subharm_more", C.c_double * 15), ("subharm_vib_rate", C.c_double), ("subharm_vib_depth", C.c_double),
        ("subharm_vib_delay", C.c_double), ("subharm_vibrato", C.c_int32), ("volume_vibrato", C.c_int32),
        ("unit_pitch_shift", C.c_int32), ("no_warp", C.c_int32),
        ("harm", C.c_void_p), ("uv", C.c_void_p), ("bre", C.c_void_p), ("rec", C.c_void_p), ("mix", C.c_void_p),
        ("f0_64", C.c_void_p),
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
    "goofer_pulse_model": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "goofer_pulse_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "goofer_gauss_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "goofer_warp_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, C.c_void_p]),
    "goofer_knot_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p]),
    "goofer_synth_batch": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_void_p]),
    "goofer_mag_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "goofer_gauss_bins_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                        C.c_int, C.c_void_p]),
    "goofer_knot_fit_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "goofer_knot_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "goofer_debug_table": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "goofer_debug_fetch": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "goofer_sizeof": (C.c_int, [C.c_int]),
    "goofer_gauss_rows_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "goofer_stretch_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "goofer_onepole_cascade": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "goofer_vocal_roughness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "goofer_pcm16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "goofer_post_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "goofer_assemble_batch": (C.c_int, [C.c_void_p, C.POINTER(Assembly), C.c_void_p]),
    "goofer_render_batch": (C.c_int, [C.c_void_p, C.POINTER(Assembly), C.POINTER(Batch), C.c_void_p]),
    "goofer_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "goofer_profile_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "goofer_profile_stage_name": (C.c_char_p, [C.c_int]),
    "goofer_profile_stage_name_ex": (C.c_char_p, [C.c_void_p, C.c_int]),
    "goofer_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "goofer_check": (C.c_int, [C.c_void_p]),
    "goofer_counter": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "goofer_host_gauss_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "goofer_host_plan_notes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "goofer_host_plan_into": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "goofer_host_plans_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "goofer_host_plans_free": (None, [C.c_void_p]),
    "goofer_host_parse_floats": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "goofer_host_pack": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "goofer_host_decode_bends": (C.c_int64, [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "goofer_smooth_mask_ds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p,
                                        C.c_void_p]),
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
