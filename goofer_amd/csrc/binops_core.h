// Bin-axis device helpers shared by binops.hip, assemble.hip and stems.hip (gfx950): numpy-exact interpolation,
// envelope warps on an LDS row, high-pass mask, 5-tap blur, frame picks, Philox.
#pragma once

#include "common.h"

// numpy's binary_search_with_guess (numpy/_core/src/multiarray/compiled_base.c) restated for the
// short anchor arrays of warp_env_by_formants: np.interp is called there with a possibly
// NON-monotone xp (anchors are not sorted, GOOFER.py:855-870), so the index it returns depends on
// the guess carried over from the previous (ascending) query.  We reproduce that exactly.
__device__ __forceinline__ int np_search_guess(double key, const double *arr, int len, int guess)
{
    if (key > arr[len - 1]) return len;
    if (key < arr[0]) return -1;
    if (len <= 4) {
        int i = 1;
        while (i < len && key >= arr[i]) ++i;
        return i - 1;
    }
    if (guess > len - 3) guess = len - 3;
    if (guess < 1) guess = 1;
    int imin = 0, imax = len;
    if (key < arr[guess]) {
        if (key < arr[guess - 1]) {
            imax = guess - 1;   // (the LIKELY_IN_CACHE_SIZE=8 refinement can never trigger for len <= 6)
        } else {
            return guess - 1;
        }
    } else {
        if (key < arr[guess + 1]) return guess;
        if (key < arr[guess + 2]) return guess + 1;
        imin = guess + 2;
    }
    while (imin < imax) {
        int imid = imin + ((imax - imin) >> 1);
        if (key >= arr[imid]) imin = imid + 1; else imax = imid;
    }
    return imin - 1;
}

// np_search_guess as a table: for the short anchor arrays (len <= 6) the result depends on the key only through the six
// comparisons key >= arr[k]; NP_SEARCH_TAB[len][mask][guess] is the function above evaluated on that bit mask (valid
// for arr[0] <= key <= arr[len-1], which holds for the bin frequencies: arr[0] = 0, arr[len-1] = nyquist).
struct np_search_tab_t {
    signed char v[7][64][4];
};
constexpr int np_search_from_mask(unsigned m, int len, int guess)
{
    if (len <= 4) {
        int i = 1;
        while (i < len && ((m >> i) & 1u)) ++i;
        return i - 1;
    }
    if (guess > len - 3) guess = len - 3;
    if (guess < 1) guess = 1;
    int imin = 0, imax = len;
    if (!((m >> guess) & 1u)) {
        if (!((m >> (guess - 1)) & 1u)) imax = guess - 1;
        else return guess - 1;
    } else {
        if (!((m >> (guess + 1)) & 1u)) return guess;
        if (!((m >> (guess + 2)) & 1u)) return guess + 1;
        imin = guess + 2;
    }
    while (imin < imax) {
        int imid = imin + ((imax - imin) >> 1);
        if ((m >> imid) & 1u) imin = imid + 1; else imax = imid;
    }
    return imin - 1;
}
constexpr np_search_tab_t make_np_search_tab()
{
    np_search_tab_t t{};
    for (int len = 2; len <= 6; ++len)
        for (unsigned m = 0; m < 64; ++m)
            for (int g = 0; g < 4; ++g) t.v[len][m][g] = (signed char)np_search_from_mask(m, len, g);
    return t;
}
static __constant__ np_search_tab_t NP_SEARCH_TAB = make_np_search_tab();

// np.interp value for index j (arr_interp inner body), xp/fp short arrays
__device__ __forceinline__ double np_interp_eval(double x, int j, const double *xp, const double *fp, int len)
{
    if (j == -1) return fp[0];
    if (j == len) return fp[len - 1];
    if (j == len - 1) return fp[j];
    if (x == xp[j]) return fp[j];
    double slope = (fp[j + 1] - fp[j]) * fast_rcp(xp[j + 1] - xp[j]);
    double v = slope * (x - xp[j]) + fp[j];
    if (isnan(v)) {
        v = slope * (x - xp[j + 1]) + fp[j + 1];
        if (isnan(v) && fp[j] == fp[j + 1]) v = fp[j];
    }
    return v;
}

// linear interpolation of an fp32 row sampled on the uniform grid b*step, at x in [0, nyq]
// (np.interp with sorted xp: largest j with xp[j] <= x), plus gf.interp1d's linear extrapolation.
// Index: multiply estimate, then the exact residual d = x - j*step (one fma, so its sign is exact) moves j by at
// most one; d is also the reference's (x - xp[j]) to within an ulp of xp[j].  The slope uses inv_step = 1/step
// instead of a division (the grid spacing is exactly `step`).
__device__ __forceinline__ double row_interp(const float *r, int n_bins, double step, double inv_step, double nyq, double x)
{
    if (x < 0.0) {
        double sl = (double)(r[1] - r[0]) / (step + 1e-10);           // fp32 difference, like the reference
        return (double)r[0] + sl * (x - 0.0);
    }
    if (x > nyq) {
        double xl = (double)(n_bins - 2) * step;
        double sl = (double)(r[n_bins - 1] - r[n_bins - 2]) / (nyq - xl + 1e-10);
        return (double)r[n_bins - 1] + sl * (x - nyq);
    }
    int j = (int)(x * inv_step);
    double d = fma(-(double)j, step, x);
    if (d < 0.0) { --j; d += step; }
    else if (d >= step) { ++j; d -= step; }
    if (j >= n_bins - 1 || x >= nyq) return (double)r[n_bins - 1];   // linspace pins the last grid point to nyq
    const double r0 = (double)r[j];
    const double slope = ((double)r[j + 1] - r0) * inv_step;
    return slope * d + r0;
}

// Formant-anchored warp (if `warp` and formants given) then uniform warp (if ratio != 1) of the fp32 row in
// `ra`, ping-ponging with `rb` (both LDS, n_bins floats).  Returns the buffer holding the result.
// `seg` = 18 doubles of per-wave LDS: the sorted-anchor path parks (x_k, y_k, slope_k) there so the per-bin
// segment lookup is three LDS reads instead of a select chain over register arrays.
// GOOFER.py:840-875, 618-627; each stage rounds to fp32 like the reference.
constexpr int WARP_SEG_DOUBLES = 18;
// fp32 segment record of the sorted-anchor warp (8-byte aligned like the table it overlays: one ds_read2_b64 per bin)
struct alignas(8) warp_seg_f32 {
    float d, s, c, pad;
};

// The bin grid of a plan, made once on the host (three fp64 divisions per row otherwise): np.linspace(0, nyq, n_bins)'s
// spacing and its reciprocal.
struct warp_grid {
    double nyq, step, inv_step;
};
static inline warp_grid make_warp_grid(int sr, int n_bins)
{
    warp_grid g;
    g.nyq = (double)sr / 2.0;
    g.step = g.nyq / (double)(n_bins - 1);
    g.inv_step = 1.0 / g.step;
    return g;
}

// Which notes' harmonic rows differ from their assembled rows: some fa..fd shift with formant tracks at hand, or a
// uniform shift (GOOFER.py:1004-1017).  One definition for the kernel that writes the warped copies and the walker that
// chooses which row to read.
__device__ __forceinline__ bool note_shifts_formants(const goofer_note_params &q)
{
    return q.f_shift[0] != 1.0 || q.f_shift[1] != 1.0 || q.f_shift[2] != 1.0 || q.f_shift[3] != 1.0;
}
__device__ __forceinline__ bool note_warps(const goofer_note_params &q, bool have_formants)
{
    return (have_formants && note_shifts_formants(q)) || (double)q.formant_shift != 1.0;
}

// One output bin of an fp32 warp stage: displacement dl (bins) of bin b (bf = (float)b) -> the source position clamped to
// [0, topf] -> 2-tap lerp of the LDS row `cur` (n_bins + 1 floats: see warp_row).
__device__ __forceinline__ float warp_lerp_f32(const float *cur, int b, float bf, float dl, float topf)
{
    dl = __builtin_amdgcn_fmed3f(dl, -bf, topf - bf);
    const float fl = __builtin_floorf(dl);
    const int j2 = b + (int)fl;
    const float d = dl - fl;                                           // exact (Sterbenz / same binade), in [0, 1)
    const float r0 = cur[j2], r1 = cur[j2 + 1];                        // j2 = n_bins - 1 only with d == 0: the pad element
    return __builtin_fmaf(r1 - r0, d, r0);
}

// The segment of the lane's bin in chunk c (bin lane + 64 c), for every chunk at once: anchor k counts from the lane's first
// chunk ck = ceil((t_k - lane) / 64) on, so with 3-bit fields per chunk the packed segment indices are the sum over k of the
// all-ones-fields constant shifted up by 3 ck (a field holds at most 5).  Five thresholds cost ~28 vector instructions per
// row; a bin then takes its segment with one bit-field extract instead of five compares, selects, shifts and adds (17
// instructions as compiled).  Nine chunks per 32-bit word.
template <int NW>
__device__ __forceinline__ void warp_seg_words(const int (&tk)[5], int lane, uint32_t (&jw)[NW])
{
#pragma unroll
    for (int q = 0; q < NW; ++q) jw[q] = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int ck = (tk[k] + 63 - lane) >> 6;                       // >= 0 for thresholds >= 0 (they are)
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            int sh = ck - 9 * q;
            sh = sh < 0 ? 0 : (sh > 10 ? 10 : sh);
            jw[q] += 0x09249249u << (3 * sh);                          // fields past the word's nine fall off the top
        }
    }
}
template <int NW>
__device__ __forceinline__ int warp_seg_at(const uint32_t (&jw)[NW], int c) { return (int)((jw[c / 9] >> (3 * (c % 9))) & 7u); }

// CH: 64-bin chunks of a row when known at compile time (the per-bin loop of the common sorted-anchor case is then
// unrolled, so its LDS round trips overlap), 0 = any width.
//
// F32 (the default): the per-bin VALUE arithmetic of the sorted-anchor warp and of the uniform warp runs in fp32 — the
// anchors, slopes, segment thresholds (everything that decides an index or a comparison, once per row) stay fp64.  The
// source position of bin b is kept as the DISPLACEMENT delta(b) = pos(b) - b = A_k + (s_k - 1) b, whose magnitude is the
// size of the formant shift in bins (a few), not the bin index (hundreds): its fp32 rounding is ~1e-6 bins against ~5e-5
// for pos itself, j = b + floor(delta) is exact integer arithmetic and the fraction is delta - floor(delta), exact.  The
// lerp is one subtraction and one FMA.  Error against the fp64 path: ~1e-6 bins of position (x the row's slope: far below
// an ulp on the smooth envelopes of this path; the uniform warp's displacement grows to the bin index, 3e-5 bins) plus
// 1.5 ulp of fp32 — DESIGN.md 4 (error budget); option "value_f64" keeps the fp64
// arithmetic for A/B.  In F32 mode BOTH row buffers must have n_bins + 1 readable floats: the pad element makes the
// right neighbour of the last bin a finite value (weight exactly 0) instead of a bounds test per bin.
template <int CH = 0, bool F32 = true>
__device__ __forceinline__ float *warp_row(float *ra, float *rb, int n_bins, const warp_grid &grid, const double *formants, const double *fs,
                                           bool warp, double ratio, int lane, double *seg)
{
    const double nyq = grid.nyq, step = grid.step, inv_step = grid.inv_step;
    float *cur = ra, *nxt = rb;
    const float topf = (float)(n_bins - 1);
    // one output bin of either fp32 stage: displacement dl (bins) -> clamp the position to [0, n_bins - 1] -> 2-tap lerp
    auto lerp_f32 = [&](int b, float bf, float dl) { nxt[b] = warp_lerp_f32(cur, b, bf, dl, topf); };

    if (warp && formants) {
        // anchors: (0,0), valid (shifted -> orig) in formant order, (nyq, nyq)      GOOFER.py:850-865
        // The row's anchor set is wave-uniform data, so it is built ONCE per row across lanes instead of once per lane:
        // lane i < 4 tests formant i, a ballot compacts the valid ones (rank = anchors in front), and the table lives in
        // the wave's 18 doubles of LDS — x (shifted) at seg[k], y (original) at seg[6 + k].  (Per-lane register arrays
        // indexed by the running length cost ~400 select instructions per row.)
        const int li = lane & 3;
        const double fo = formants[li];
        const double fsl = li == 0 ? fs[0] : (li == 1 ? fs[1] : (li == 2 ? fs[2] : fs[3]));
        const double fsft = fo * fsl;
        const bool valid = lane < 4 && fo > 50.0 && fo < nyq && fsft > 50.0;
        const unsigned vm = (unsigned)__ballot(valid) & 15u;
        const int len = __popc(vm) + 2;
        if (valid) {
            const int rank = 1 + __popc(vm & ((1u << li) - 1u));
            seg[rank] = fsft;
            seg[6 + rank] = fo;
        }
        if (lane == 4) { seg[0] = 0.0; seg[6] = 0.0; }
        if (lane == 5) { seg[len - 1] = nyq; seg[6 + len - 1] = nyq; }
        wave_lds_sync();
        const double *dst = seg, *sp = seg + 6;
        // lane k < len holds anchor k and its successor
        const int kk = lane < len ? lane : len - 1, kn = kk + 1 < len ? kk + 1 : kk;
        const double xk = dst[kk], yk = sp[kk], xn = dst[kn], yn = sp[kn];
        const bool sorted = __all(!(lane < len - 1) || xk <= xn);
        if (sorted || len <= 4) {
            // monotone anchors (or numpy's guess-free linear search for len <= 4): the answer does not depend on the
            // guess chain — index = number of anchors (after the first) that are <= x
            if (sorted) {
                // The warp of a sorted anchor set is piecewise linear in the bin index: source position (in bins)
                // pos(b) = A_k + s_k b on segment k, with s_k the np.interp slope and A_k = (y_k - s_k x_k) / step.
                // Folding the two interpolations into that form moves the fp64 intermediates by ~1e-13 bins — far
                // below the fp32 rounding of the result — and leaves ~25 vector instructions per bin instead of ~85.
                // segment of bin b = number of anchors d_k (k >= 1) with d_k <= x(b), x(b) = b step (nyq for the last bin) — x is
                // increasing in b, so anchor k contributes from the first bin thr_k on: lane k finds thr_k once per row with the
                // very comparison the per-bin search would make, and the bins compare integers (five fp64 compares, a
                // conversion and a multiply per bin less)
                int thr = n_bins;
                if (lane >= 1 && lane < len) {
                    const double dk = xk;
                    auto xb = [&](int c) { return c >= n_bins - 1 ? nyq : (double)c * step; };
                    // ceil(dk / step) through the reciprocal is within one of the first bin with dk <= x(bin) (both are a
                    // few 1e-16 relative away from the real quotient), so one conditional step either way settles it
                    const double est = ceil(dk * inv_step);
                    int c = est < 0.0 ? 0 : (est > (double)(n_bins - 1) ? n_bins - 1 : (int)est);
                    if (c > 0 && dk <= xb(c - 1)) --c;
                    else if (c < n_bins && !(dk <= xb(c))) ++c;
                    thr = c;
                }
                wave_lds_sync();                                       // every lane holds its anchors: the table may be overwritten
                warp_seg_f32 *segf = reinterpret_cast<warp_seg_f32 *>(seg);
                if (lane < len) {
                    const double sl = lane < len - 1 ? (yn - yk) * fast_rcp(xn - xk) : 0.0;
                    if constexpr (F32) {
                        // displacement on segment k, centred on the segment's first bin c_k: delta(b) = D_k + (s_k - 1)(b - c_k) with
                        // D_k = pos(c_k) - c_k.  Both terms are of the size of the formant shift in bins, whatever |A_k| is (two
                        // anchors shifted close together give slopes of 10 and intercepts of hundreds of bins)
                        const double ck = lane >= 1 ? (double)thr : 0.0;
                        const double A = (yk - sl * xk) * inv_step;
                        segf[lane] = warp_seg_f32{(float)(A + (sl - 1.0) * ck), (float)(sl - 1.0), (float)ck, 0.f};
                    } else {
                        seg[3 * lane] = (yk - sl * xk) * inv_step;         // A_k
                        seg[3 * lane + 1] = sl;                            // s_k
                        seg[3 * lane + 2] = xk;
                    }
                }
                if constexpr (F32) {
                    if (lane == 0) cur[n_bins] = cur[n_bins - 1];      // pad element (see above)
                }
                wave_lds_sync();
                // lanes >= len keep n_bins, so absent anchors never count
                const int t1 = __builtin_amdgcn_readlane(thr, 1), t2 = __builtin_amdgcn_readlane(thr, 2), t3 = __builtin_amdgcn_readlane(thr, 3),
                          t4 = __builtin_amdgcn_readlane(thr, 4), t5 = __builtin_amdgcn_readlane(thr, 5);
                const double top = (double)(n_bins - 1);
                auto bin = [&](int b, int j) {
                    if constexpr (F32) {
                        const warp_seg_f32 as = segf[j];
                        const float bf = (float)b;
                        lerp_f32(b, bf, __builtin_fmaf(as.s, bf - as.c, as.d));
                        return;
                    }
                    double pos = fma(seg[3 * j + 1], (double)b, seg[3 * j]);
                    pos = __builtin_fmax(0.0, __builtin_fmin(pos, top));          // (finite: the anchors are)
                    int j2 = (int)pos;
                    if (j2 > n_bins - 2) j2 = n_bins - 2;
                    const double d = pos - (double)j2;
                    const double r0 = (double)cur[j2];
                    nxt[b] = (float)(((double)cur[j2 + 1] - r0) * d + r0);
                };
                auto seg_of = [&](int b) { return (b >= t1) + (b >= t2) + (b >= t3) + (b >= t4) + (b >= t5); };
                if constexpr (CH > 0 && CH <= 18 && F32) {
                    constexpr int NW = (CH + 8) / 9;
                    uint32_t jw[NW];
                    const int tk[5] = {t1, t2, t3, t4, t5};
                    warp_seg_words<NW>(tk, lane, jw);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int b = lane + WAVE * c;
                        if (c < CH - 1 || b < n_bins) bin(b, warp_seg_at<NW>(jw, c));
                    }
                } else if constexpr (CH > 0) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int b = lane + WAVE * c;
                        if (c < CH - 1 || b < n_bins) bin(b, seg_of(b));
                    }
                } else {
                    for (int b = lane; b < n_bins; b += WAVE) bin(b, seg_of(b));
                }
            } else {
                for (int b = lane; b < n_bins; b += WAVE) {
                    const double x = b >= n_bins - 1 ? nyq : (double)b * step;
                    const int j = np_search_guess(x, dst, len, 1);
                    const double wf = np_interp_eval(x, j, dst, sp, len);
                    nxt[b] = (float)row_interp(cur, n_bins, step, inv_step, nyq, wf);
                }
            }
        } else {
        // Resolve np.interp's guess chain over the ascending bin frequencies.  The clamped guess
        // takes at most 3 values (1..len-3), so each bin is a map state->state; lanes own
        // contiguous chunks, compose their maps, scan across the wave, then replay.
        // The anchors already sit in LDS (x at seg[k], y at seg[6 + k]); the slopes of np.interp's segments join them.
        if (lane < len) seg[12 + lane] = lane < len - 1 ? (yn - yk) * fast_rcp(xn - xk) : 0.0;
        wave_lds_sync();
        const double *xp = seg, *fp = seg + 6;                // run-time indexed copies (np_interp_eval)
        const int per = (n_bins + WAVE - 1) / WAVE;
        const int b0 = lane * per;
        const double a1 = dst[1], a2 = dst[2], a3 = dst[3], a4 = dst[4], a5 = len > 5 ? dst[5] : 0.0;   // len is 5 or 6 here
        auto cmp_mask = [&](double x) {
            unsigned m = 1u;                                  // x >= arr[0] = 0
            m |= (unsigned)(x >= a1) << 1;
            m |= (unsigned)(x >= a2) << 2;
            m |= (unsigned)(x >= a3) << 3;
            m |= (unsigned)(x >= a4) << 4;
            m |= (unsigned)((len > 5) & (x >= a5)) << 5;
            return m;
        };
        auto clampg = [&](int g) { int hi = len - 3; if (g > hi) g = hi; if (g < 1) g = 1; return g; };
        // lane m keeps the table row of comparison mask m (four results, one per guess, packed in a dword); a bin's row
        // is then one cross-lane read away
        const uint32_t my_row = reinterpret_cast<const uint32_t *>(NP_SEARCH_TAB.v[len])[lane];
        auto lookup = [&](unsigned mask, int g) { return (int)(signed char)((__shfl((int)my_row, (int)mask, WAVE) >> (8 * g)) & 0xff); };
        int m1 = 1, m2 = 2, m3 = 3;                 // composed map of this lane's chunk: state s -> m_s
        for (int q = 0; q < per; ++q) {
            int b = b0 + q;
            double x = b >= n_bins - 1 ? nyq : (double)b * step;
            const int row = __shfl((int)my_row, (int)cmp_mask(x), WAVE);          // every lane takes part in the exchange
            if (b < n_bins) {
                m1 = clampg((int)(signed char)((row >> (8 * m1)) & 0xff));
                m2 = clampg((int)(signed char)((row >> (8 * m2)) & 0xff));
                m3 = clampg((int)(signed char)((row >> (8 * m3)) & 0xff));
            }
        }
        // inclusive scan of map composition (earlier lanes apply first)
        for (int off = 1; off < WAVE; off <<= 1) {
            int p1 = __shfl_up(m1, off, WAVE), p2 = __shfl_up(m2, off, WAVE), p3 = __shfl_up(m3, off, WAVE);
            if (lane >= off) {
                int a1_ = p1 == 1 ? m1 : (p1 == 2 ? m2 : m3);
                int a2_ = p2 == 1 ? m1 : (p2 == 2 ? m2 : m3);
                int a3_ = p3 == 1 ? m1 : (p3 == 2 ? m2 : m3);
                m1 = a1_; m2 = a2_; m3 = a3_;
            }
        }
        // state entering this lane's chunk = inclusive result of lane-1 applied to the initial guess
        int incoming = __shfl_up(m1, 1, WAVE);       // initial j = 0 clamps to state 1
        int guess = lane == 0 ? 1 : incoming;
        for (int q = 0; q < per; ++q) {
            int b = b0 + q;
            double x = b >= n_bins - 1 ? nyq : (double)b * step;
            const int j = lookup(cmp_mask(x), guess);
            if (b < n_bins) {
                guess = clampg(j);
                // np_interp_eval with the slope read from the per-segment table (0 <= j <= len - 1 for these keys)
                double wf;
                if (j >= len - 1) {
                    wf = fp[len - 1];
                } else {
                    const double xj = xp[j], yj = fp[j];
                    wf = x == xj ? yj : seg[12 + j] * (x - xj) + yj;
                    if (isnan(wf)) wf = np_interp_eval(x, j, xp, fp, len);
                }
                nxt[b] = (float)row_interp(cur, n_bins, step, inv_step, nyq, wf);
            }
        }
        }
        wave_lds_sync();
        float *t = cur; cur = nxt; nxt = t;
    }
    if (ratio != 1.0) {
        // env(f) <- env(clip(f / ratio, 0, nyq)): in bin units the source position is b / ratio (GOOFER.py:618-627)
        const double inv_ratio = fast_rcp(ratio);
        const double top = (double)(n_bins - 1);
        if constexpr (F32) {
            if (lane == 0) cur[n_bins] = cur[n_bins - 1];
            wave_lds_sync();
            const float irm1 = (float)(inv_ratio - 1.0);                  // displacement per bin: b / ratio - b
            if (CH > 0) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int b = lane + WAVE * c;
                    const float bf = (float)b;
                    if (c < CH - 1 || b < n_bins) lerp_f32(b, bf, irm1 * bf);
                }
            } else {
                for (int b = lane; b < n_bins; b += WAVE) {
                    const float bf = (float)b;
                    lerp_f32(b, bf, irm1 * bf);
                }
            }
        } else
        for (int b = lane; b < n_bins; b += WAVE) {
            double pos = (double)b * inv_ratio;
            pos = pos < 0.0 ? 0.0 : (pos > top ? top : pos);
            int j2 = (int)pos;
            if (j2 > n_bins - 2) j2 = n_bins - 2;
            const double d = pos - (double)j2;
            const double r0 = (double)cur[j2];
            nxt[b] = (float)(((double)cur[j2 + 1] - r0) * d + r0);
        }
        wave_lds_sync();
        float *t = cur; cur = nxt; nxt = t;
    }
    return cur;
}

// frame -> per-frame picks of the per-sample arrays: x[::hop] edge-padded to T (GOOFER.py:1104-1106)
__device__ __forceinline__ int64_t pick_index(int64_t t, int64_t n, int hop)
{
    if (n <= 0) return 0;
    const int64_t at = t * hop;
    if (at < n) return at;                        // t < len(x[::hop]): the usual case, no division
    return ((n - 1) / hop) * hop;                 // edge-padded: the last pick
}

// 1 / (1 + exp(-clip((f - f0) / 5, -60, 60)))   GOOFER.py:1110-1111.  Hardware exp2 / rcp (1 ulp) and a multiply by
// 0.2f stand in for the reference's exp and two divisions: <= 3e-6 relative on the mask, far inside the bound.
__device__ __forceinline__ float hp_mask(float freq, float f0f)
{
    float z = (freq - f0f) * 0.2f;
    z = fminf(fmaxf(z, -60.0f), 60.0f);
    const float e = __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// 5-tap sigma=0.5 blur of a complex row held in LDS (reflect padded).  The reference accumulates in complex128 and
// rounds to complex64; here the five products are fp32 FMAs in the same tap order (<= 2e-7 relative).
__device__ __forceinline__ float2 blur5(const float2 *r, int k, int n_bins, const double *t5)
{
    const float t0 = (float)t5[0], t1 = (float)t5[1], t2 = (float)t5[2], t3 = (float)t5[3], t4 = (float)t5[4];
    float2 v0, v1, v2, v3, v4;
    if (k >= 2 && k + 2 < n_bins) {
        v0 = r[k - 2]; v1 = r[k - 1]; v2 = r[k]; v3 = r[k + 1]; v4 = r[k + 2];
    } else {
        auto at = [&](int q) { return r[q < 0 ? -q : (q >= n_bins ? 2 * (n_bins - 1) - q : q)]; };
        v0 = at(k - 2); v1 = at(k - 1); v2 = at(k); v3 = at(k + 1); v4 = at(k + 2);
    }
    float re = t0 * v0.x, im = t0 * v0.y;
    re = fmaf(t1, v1.x, re); im = fmaf(t1, v1.y, im);
    re = fmaf(t2, v2.x, re); im = fmaf(t2, v2.y, im);
    re = fmaf(t3, v3.x, re); im = fmaf(t3, v3.y, im);
    re = fmaf(t4, v4.x, re); im = fmaf(t4, v4.y, im);
    return make_float2(re, im);
}

// Philox-4x32 with 7 rounds (the Crush-resistant minimum of the Random123 paper; its 32-bit multiplies only feed noise
// phases) keyed by (seed), counter (frame, slot): four 32-bit words per block = EIGHT 16-bit phases (2 pi / 65536: the
// reference's float32 uniforms resolve finer, no ear or statistic does).  The phase of bin k comes from slot (k & 63) + 64 (k >> 9),
// word (k >> 7) & 3, half (k >> 6) & 1 — so a lane that owns bins lane, lane + 64, ... needs ONE block per eight of its bins (one
// block per frame for n_fft 1024; it was two with 24-bit phases).  philox_u16 is the same mapping evaluated for a single bin.
__device__ __forceinline__ uint4 philox_4x32(uint64_t seed, uint64_t ctr_hi, uint32_t ctr_lo)
{
    uint32_t c0 = ctr_lo, c1 = (uint32_t)ctr_hi, c2 = (uint32_t)(ctr_hi >> 32), c3 = 0x9E3779B9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ uint32_t philox_half(const uint4 &v, int i)       // i = index of the bin among its lane's eight
{
    const int w = (i >> 1) & 3;
    const uint32_t word = w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w));
    return (i & 1) ? (word >> 16) : (word & 0xffffu);
}

__device__ __forceinline__ uint32_t philox_u16(uint64_t seed, uint64_t frame, uint32_t bin)
{
    const uint4 v = philox_4x32(seed, frame, (bin & 63u) + 64u * (bin >> 9));
    return philox_half(v, (int)((bin >> 6) & 7u));
}
