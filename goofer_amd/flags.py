"""The resampler's flag vocabulary as one table: what `sampler.decode_request` understands, the ranges a host application may
offer, and the OpenUtau expression each flag is exposed as.  `manifest()` / `python -m goofer_amd.flags` emit the OpenUtau
resampler manifest (the YAML next to the entry script) from this table, so the manifest cannot drift from the decoder.

Reference: SillySampler.py:286-411 (scaling of every flag), README.md:6-41 (ranges), SillySampler.yaml (expression ids, labels
and abbreviations OpenUtau shows — interface data that has to stay what users' projects already reference).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Flag:
    flag: str                 # letters as typed in the flag string; for `options` flags the prefix before the digit
    lo: int
    hi: int
    default: int
    drives: str               # field of sampler.Request it sets
    doc: str
    # OpenUtau expression (None: not offered by the manifest; still decoded when typed)
    key: str | None = None
    label: str | None = None
    abbr: str | None = None
    options: tuple = ()       # non-empty: an `Options` expression, its values are flag + digit


TAG = " (SillySampler)"

FLAGS = (
    Flag("t", -100, 100, 0, "flags['t']", "pitch offset in cents", "cent", "Pitch Offset", "foff"),
    Flag("g", -100, 100, 0, "formant_shift", "global formant shift (gender): ratio 1 + g / 200"),
    Flag("fw", -100, 100, 0, "formant_width", "formant width: bins stretched about the centre by 1 + 0.001 fw", "fmwd",
         "Formant Width" + TAG, "S_FW"),
    Flag("fst", -100, 100, 0, "formant_strength", "formant-band strength, all four", "fmst", "Formant Strength Global" + TAG, "S_FT"),
    Flag("fa", -100, 100, 0, "f_shift[0]", "scale formant 1: ratio 1 + fa / 100", "SF1", "Scale Formant (F1)" + TAG, "S_F1"),
    Flag("fb", -100, 100, 0, "f_shift[1]", "scale formant 2", "SF2", "Scale Formant (F2)" + TAG, "S_F2"),
    Flag("fc", -100, 100, 0, "f_shift[2]", "scale formant 3", "SF3", "Scale Formant (F3)" + TAG, "S_F3"),
    Flag("fd", -100, 100, 0, "f_shift[3]", "scale formant 4", "SF4", "Scale Formant (F4)" + TAG, "S_F4"),
    Flag("fsta", -100, 100, 0, "formant_strength[0]", "strength of formant 1", "STF1", "Strength Formant (F1)" + TAG, "STF1"),
    Flag("fstb", -100, 100, 0, "formant_strength[1]", "strength of formant 2", "STF2", "Strength Formant (F2)" + TAG, "STF2"),
    Flag("fstc", -100, 100, 0, "formant_strength[2]", "strength of formant 3", "STF3", "Strength Formant (F3)" + TAG, "STF3"),
    Flag("fstd", -100, 100, 0, "formant_strength[3]", "strength of formant 4", "STF4", "Strength Formant (F4)" + TAG, "STF4"),
    Flag("V", 0, 100, 100, "harmonic_mix", "harmonic (voiced) level", "Hvoi", "Voiced Harmonics" + TAG, "S_V"),
    Flag("B", -100, 100, 0, "breathiness_mix", "breath level: (B + 100) / 100"),
    Flag("U", -100, 100, 0, "unvoiced_mix", "unvoiced (fricative) level: (U + 100) / 100", "cons", "Unvoiced Consonant Gain" + TAG, "S_C"),
    Flag("sh", 0, 100, 0, "f0_jitter_strength", "f0 jitter (harsh)", "grit", "Grittiness" + TAG, "S_G"),
    Flag("sr", 0, 100, 0, "volume_jitter_strength", "volume jitter (rough)", "dist", "Distortion" + TAG, "S_D"),
    Flag("st", -100, 100, 0, "tension", "tension", "tens", "Tension" + TAG, "S_T"),
    Flag("sg", 0, 100, 0, "subharm_weight", "growl: +12 semitone pulse layer", "grwl", "Growl" + TAG, "S_GW"),
    Flag("vf", -100, 100, 0, "fry", "vocal fry amount (positive: start, negative: end)", "vfry", "Vocal Fry" + TAG, "S_VF"),
    Flag("vh", 0, 100, 50, "fry_hz", "vocal fry base pitch in Hz", "vfhz", "Vocal Fry Base Hz" + TAG, "S_VZ"),
    Flag("vl", 0, 100, 15, "fry_glide", "vocal fry pitch slide amount", "vfsl", "Vocal Fry Slide Amount" + TAG, "S_VL"),
    Flag("sd", 0, 100, 0, "sd_strength", "noise jitter (dry throat)", "thdr", "Dryness" + TAG, "S_DR"),
    Flag("sj", 0, 100, 0, "growl_mix", "rasp: blend of an f0-jittered layer", "rasp", "Rasp" + TAG, "S_SJ"),
    Flag("sa", 0, 100, 0, "aperiodic_mix", "whisper growl: blend of a full-noise layer", "wgwl", "Whisper Growl" + TAG, "S_WG"),
    Flag("su", 0, 100, 0, "subharm_gain", "sub-harmonic layer strength", "subh", "Subharmonics" + TAG, "S_SH"),
    Flag("br", -100, 100, 0, "brightness_env", "spectral-envelope tilt", "brig", "Brightness", "BRI"),
    Flag("es", -100, 100, 0, "env_shape", "envelope smoothing (< 0) / sharpening (> 0)", "evsh", "Envelope Shaping" + TAG, "EVSH"),
    Flag("pd", -100, 100, 0, "pitch_dyn", "dynamics from the pitch curve", "pdyn", "Dynamic from Pitch" + TAG, "PDYN"),
    Flag("P", 0, 100, 100, "normalize", "peak normalisation amount (absent: full)"),
    Flag("L", 0, 2, 0, "loop_mode", "sustain: L0 concat loop, L1 averaged mirror loop, L2 stretch", "sust", "Sustain Behavior" + TAG, "S_SS",
         ("L0", "L1", "L2")),
    Flag("FV", 0, 1, 0, "force_voiced", "force the whole note voiced", "fvoi", "Force Voicing" + TAG, "FVOI", ("FV0", "FV1")),
    Flag("R", 0, 1, 0, "reverse", "reverse the sample", "rev", "Reverse", "REV", ("R0", "R1")),
    Flag("SE", 0, 1, 0, "use_editor", "SillyEditor voicing regions (parsed; the Tk editor is not part of this backend)", "edit",
         "SillyEditor", "SEDI", ("SE0", "SE1")),
)

BY_FLAG = {f.flag: f for f in FLAGS}


def manifest() -> dict:
    """The OpenUtau resampler manifest as data: {'expressions': {id: {...}}} in the table's order."""
    ex = {}
    for f in FLAGS:
        if f.key is None:
            continue
        e = {"name": f.label, "abbr": f.abbr}
        if f.options:
            e.update(type="Options", min=0, max=1, default_value=f.default, is_flag=True, options=list(f.options))
        else:
            e.update(type="Numerical", min=f.lo, max=f.hi, default_value=f.default, is_flag=True, flag=f.flag)
        ex[f.key] = e
    return {"expressions": ex}


def manifest_yaml() -> str:
    out = ["expressions:"]
    for key, e in manifest()["expressions"].items():
        out.append(f"  {key}:")
        for k in ("name", "abbr", "type", "min", "max", "default_value", "is_flag", "flag", "options"):
            if k not in e:
                continue
            v = e[k]
            if k == "options":
                out.append("    options:")
                out += [f"    - {o}" for o in v]
            else:
                out.append(f"    {k}: {str(v).lower() if isinstance(v, bool) else v}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    import sys
    sys.stdout.write(manifest_yaml())
