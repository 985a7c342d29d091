#!/usr/bin/env python3
"""Per-phase instruction tables of the two stem walkers from the compiler's assembly (no GPU needed).

    python scripts/isa_phases.py [--out profiles/r03_isa_table.txt]

Compiles goofer_amd/csrc/stems.hip for gfx950 with line tables, attributes every instruction of the frame loop of
k_noise_stems<512,false> and k_harm_stem<512> to the line of the KERNEL BODY it was inlined into (scripts/isa_table.py
--by-root) and buckets those lines into phases.  The phase boundaries are found from marker comments in the source, so the
table follows the code when it moves.  Counts are STATIC (instructions of the kernel, every branch counted once, the set-up in front of the loop in its own row):
next to each phase the table says when it runs on the bench workload (BASELINE config 3: voiced after 8 %).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "goofer_amd", "csrc", "stems.hip")


def find(lines, text, start=0):
    for i in range(start, len(lines)):
        if text in lines[i]:
            return i + 1
    raise SystemExit("marker not found: " + text)


def phases_of(lines):
    n0 = find(lines, "void k_noise_stems(const noise_args A)")
    h0 = find(lines, "void k_harm_stem(")
    fin = find(lines, "constexpr int FIN_THREADS")
    L = lambda t, s=0: find(lines, t, s)
    noise = [
        ("set-up outside the frame loop: tables into LDS, run bounds, first frame records", [(n0, L("    float2 carry_u[R - G], carry_b[R - G];", n0) - 1)]),
        ("frame records, note entry, flatness checks, knot prefetch (every frame; note entry once per note)", [(L("    float2 carry_u[R - G], carry_b[R - G];", n0), L("// 1. noise envelope: sigma-1.75 blur", n0) - 1)]),
        ("1. row staging through LDS + sigma-1.75 blur (every frame)", [(L("// 1. noise envelope: sigma-1.75 blur", n0), L("if (f + 1 < f1) {                                     // the row registers are consumed", n0) - 1)]),
        ("next row's fetch + frame-record refill (refill every 64 frames)", [(L("if (f + 1 < f1) {                                     // the row registers are consumed", n0), L("// 2. U * env_n", n0) - 1)]),
        ("2. Philox, sin/cos, U*env, high-pass, brightness (every frame; brightness on voiced frames)", [(L("// 2. U * env_n", n0), L("        const bool td_blur = (mode & 4) != 0;", n0) - 1)]),
        ("2b. 5-tap blur of the breath spectrum (voiced frames, only with td_blur off: the default folds it into the window)", [(L("        const bool td_blur = (mode & 4) != 0;", n0), L("// 3. inverse transforms + overlap-add", n0) - 1)]),
        ("3. irFFT + overlap-add, breath stem (skipped where the stem gain is exactly 0: unvoiced stretches)", [(L("if (voiced && td_blur) w.blur_edges(sb, ec, t5[0], t5[1]);", n0), L("w.inverse_ola(sb, t, carry_b, ob, (voiced && td_blur) ? w.wsv : w.wsc);", n0))]),
        ("3. irFFT + overlap-add, unvoiced stem (skipped where exactly 0: voiced stretches = the bench workload)", [(L("else w.inverse_ola(su, t, carry_u, ou, w.wsc);", n0), L("else w.inverse_ola(su, t, carry_u, ou, w.wsc);", n0))]),
        ("3. skip bookkeeping (ring rotation of a skipped transform, flatness bits)", [(L("// 3. inverse transforms + overlap-add", n0), L("// 4. hop t -> window-sum quotient", n0) - 2)]),
        ("4. output: window-sum quotient, mask gain, stores (every emitted frame; smooth_mask_at32 only on non-flat hops)", [(L("// 4. hop t -> window-sum quotient", n0) - 1, h0 - 1)]),
    ]
    harm = [
        ("set-up outside the frame loop: tables into LDS, run bounds, first frame records", [(h0, L("    float2 raw[R];", h0) - 1)]),
        ("frame records, note entry (every frame)", [(L("    float2 raw[R];", h0), L("    auto fetch = [&](int idx) {", h0) - 1), (L("    float2 carry[R - G];", h0), L("        fetch(idx);", h0) - 1),
                                                     (L("        fetch(idx);", h0) + 1, L("// 1. windowed frame -> complex FFT", h0) - 1)]),
        ("1. window + forward FFT (every frame)", [(L("// 1. windowed frame -> complex FFT", h0), L("// 2. even/odd split", h0) - 1)]),
        ("2. even/odd split through LDS (every frame)", [(L("// 2. even/odd split", h0), L("// 3. shaping (GOOFER.py:1102-1144)", h0) - 1)]),
        ("the frame's fetch at its head: 8 sample pairs + envelope row (the reflect padding of a note's end frames is most of the static count)", [(L("    auto fetch = [&](int idx) {", h0), L("    float2 carry[R - G];", h0) - 1), (L("        fetch(idx);", h0), L("        fetch(idx);", h0))]),
        ("3. shaping: high-pass, max|S|, env * boost, brightness (every frame)", [(L("// 3. shaping (GOOFER.py:1102-1144)", h0), L("        if (voiced && !td_blur) {\n", h0) - 1)]),
        ("3b. 5-tap blur (voiced frames, only with td_blur off)", [(L("        if (voiced && !td_blur) {\n", h0), L("// 4. inverse transform + overlap-add; hop t leaves", h0) - 1)]),
        ("4. blur edge correction + irFFT + overlap-add (every frame)", [(L("// 4. inverse transform + overlap-add; hop t leaves", h0), L("w.inverse_ola(X, t, carry, e, (voiced && td_blur) ? w.wsv : w.wsc);", h0))]),
        ("4b. output: window-sum quotient, stores (every emitted frame)", [(L("w.inverse_ola(X, t, carry, e, (voiced && td_blur) ? w.wsv : w.wsc);", h0) + 1, fin - 1)]),
    ]
    mk = lambda ph: [{"name": n, "file": "stems.hip", "lines": [list(r) for r in rs]} for n, rs in ph]
    return mk(noise), mk(harm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    lines = open(SRC).read().split("\n")
    lines = [l + "\n" for l in lines]
    noise, harm = phases_of(lines)
    out = []
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", "-save-temps", "-c", SRC, "-o", os.path.join(td, "stems.o")]
        subprocess.run(cmd, cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = os.path.join(td, "stems-hip-amdgcn-amd-amdhsa-gfx950.s")
        for kern, ph in (("k_noise_stemsILi512ELb0", noise), ("k_harm_stemILi512", harm)):
            pj = os.path.join(td, kern + ".json")
            json.dump(ph, open(pj, "w"))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_table.py"), asm, kern, "--by-root", "--phases", pj], capture_output=True, text=True, check=True)
            out.append(r.stdout)
    text = __doc__.split("\n\n", 1)[1] + "\n" + "\n".join(out)
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
