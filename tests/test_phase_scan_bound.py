"""CPU: the rounding band of the parallel phase scan (goofer_amd/csrc/pulse.hip, k_pulse_onsets_par) against data.

The kernel takes a note's pulse onsets from a blocked fp64 scan S_i instead of the reference's sequential sum p_i
(GOOFER.py:491) whenever no integer lies within tol_i = 2.3e-16 (c0 + 512) S_i of S_i.  That is sound if |p_i - S_i| <= tol_i
for every sample; the bound is Higham's gamma_{i-1} sum|x| for either order.  Here the same two sums are formed in numpy
(np.cumsum is the sequential loop; the scan is restated with the kernel's association: 8 samples per lane, a Hillis-Steele
scan of the 64 lane totals, a carry per 512-sample round) and the worst observed |p_i - S_i| / tol_i is reported: it must stay
below 1 (measured: 0.11, on constant f0 where the roundings of equal terms do not cancel).
"""
import numpy as np


def scan_like_kernel(x):
    n = len(x)
    pad = (-n) % 512
    xp = np.concatenate([x, np.zeros(pad)])
    S = np.empty_like(xp)
    tol = np.empty_like(xp)
    carry = 0.0
    for c0 in range(0, len(xp), 512):
        blk = xp[c0:c0 + 512].reshape(64, 8)
        l = np.cumsum(blk, axis=1)                              # sequential inside a lane
        incl = l[:, 7].copy()
        o = 1
        while o < 64:                                           # wave_scan_add_f64
            nxt = incl.copy()
            nxt[o:] = incl[o:] + incl[:-o]
            incl = nxt
            o *= 2
        excl = np.concatenate([[0.0], incl[:-1]])
        p0 = carry + excl
        S[c0:c0 + 512] = (p0[:, None] + l).reshape(-1)
        tol[c0:c0 + 512] = S[c0:c0 + 512] * (2.3e-16 * (c0 + 512))
        carry = carry + incl[63]
    return S[:n], tol[:n]


def test_scan_stays_inside_its_band():
    rng = np.random.default_rng(5)
    worst = 0.0
    for k in range(24):
        n = int(rng.integers(3000, 140000))
        t = np.arange(n) / 44100
        f0 = (rng.uniform(60, 1200) * 2 ** (rng.uniform(-0.5, 0.5) * np.sin(2 * np.pi * rng.uniform(0.2, 9) * t))).astype(np.float32)
        f0[rng.uniform(size=n) < 0.05] = 0
        if k % 4 == 0:
            f0[:] = [441.0, 220.5, 882.0, 97.3, 1000.0, 55.0][k // 4]
        x = f0.astype(np.float64) / 44100.0
        p = np.cumsum(x)                                        # the reference's order
        S, tol = scan_like_kernel(x)
        live = tol > 0
        assert np.all(S[~live] == p[~live])                     # nothing added yet: both exactly 0
        worst = max(worst, float(np.max(np.abs(p[live] - S[live]) / tol[live])))
        # where the band holds no integer, the floors agree (what the kernel relies on)
        sure = live & (np.abs(S - np.rint(S)) > tol)
        assert np.array_equal(np.floor(S[sure]), np.floor(p[sure]))
    assert worst < 0.5, worst                                   # measured 0.11 (constant f0: correlated roundings in the first round)
