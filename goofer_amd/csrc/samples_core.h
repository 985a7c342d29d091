// Per-sample device helpers shared by samples.hip and stems.hip (gfx950): the smoothed voicing-mask upsampler
// (GOOFER.py:563-567) and the corrected-reciprocal quotient.
#pragma once

#include "common.h"

#define MASK_DS 4
#define KNOT_MARGIN 12   // knots staged per hop beyond hop / MASK_DS (k_irfft_ola3)

// short-array slot of a note: floor(sample_off/4) + note  (capacity >= ceil(n/4))
__device__ __forceinline__ int64_t short_base(const int64_t *sample_off, int note) { return sample_off[note] / MASK_DS + note; }

// np.linspace(0, 1, num, dtype=float32)[i] as a double; step = 1/(num-1) precomputed per note
__device__ __forceinline__ double lin01_f32(int64_t i, int64_t num, double step)
{
    if (num <= 1) return 0.0;
    if (i >= num - 1) return 1.0;
    return (double)(float)((double)i * step);
}

// np.interp of the smoothed decimated mask (knots at float32 linspace(0,1,ns)) at float32
// linspace(0,1,n)[i]  (GOOFER.py:564-567).  Index search is exact (integer estimate + compare fix-up on
// the true knot positions); the slope uses a 1e-16-accurate reciprocal instead of a division.
__device__ __forceinline__ float smooth_mask_at(const double *__restrict__ ss, int64_t ns, int64_t i, int64_t n, double step_n,
                                                double step_s)
{
    if (ns <= 1) return (float)ss[0];                 // single knot: constant (GOOFER.py:183-191)
    const double x = lin01_f32(i, n, step_n);
    int64_t j = (int64_t)(x * (double)(ns - 1));
    if (j > ns - 1) j = ns - 1;
    if (j < 0) j = 0;
    while (j + 1 <= ns - 1 && lin01_f32(j + 1, ns, step_s) <= x) ++j;
    while (j > 0 && lin01_f32(j, ns, step_s) > x) --j;
    if (j >= ns - 1) return (float)ss[ns - 1];
    const double xj = lin01_f32(j, ns, step_s);
    if (x == xj) return (float)ss[j];
    const double s0 = ss[j], s1 = ss[j + 1];
    const double slope = (s1 - s0) * fast_rcp(lin01_f32(j + 1, ns, step_s) - xj);
    return (float)(slope * (x - xj) + s0);
}

// 32-bit flavour for the hot kernels (note lengths are far below 2^31): same comparisons, half the integer work
__device__ __forceinline__ double lin01_f32_i(int i, int num, double step)
{
    if (num <= 1) return 0.0;
    if (i >= num - 1) return 1.0;
    return (double)(float)((double)i * step);
}

// `knot(k)` returns smoothed-mask knot k: a global array, or the window of it a wave has staged in LDS
template <typename Knot>
__device__ __forceinline__ float smooth_mask_at32(Knot knot, int ns, int i, int n, double step_n, double step_s,
                                                  float knots_per_sample)
{
    if (ns <= 1) return (float)knot(0);
    // Almost everywhere the smoothed mask is flat (0, or the tap sum): when the four knots around a cheap index
    // estimate (good to +-1) are equal, any of the candidate intervals interpolates to exactly that value
    // (slope 0), and the exact index search below is not needed.
    {
        int je = (int)((float)i * knots_per_sample);
        je = je < 1 ? 1 : (je > ns - 3 ? ns - 3 : je);
        if (ns >= 4) {
            const double a = knot(je - 1), b = knot(je), c = knot(je + 1), d = knot(je + 2);
            if (a == b && b == c && c == d) return (float)b;
        }
    }
    const double x = lin01_f32_i(i, n, step_n);
    int j = (int)(x * (double)(ns - 1));
    if (j > ns - 1) j = ns - 1;
    if (j < 0) j = 0;
    // the estimate is within one knot of the answer (both grids are fp32 roundings of k / (num - 1)): one step
    // either way replaces the search loops of smooth_mask_at
    double xj = lin01_f32_i(j, ns, step_s), xn = lin01_f32_i(j + 1, ns, step_s);
    if (j + 1 <= ns - 1 && xn <= x) {
        ++j;
        xj = xn;
        xn = lin01_f32_i(j + 1, ns, step_s);
    } else if (j > 0 && xj > x) {
        --j;
        xn = xj;
        xj = lin01_f32_i(j, ns, step_s);
    }
    if (j >= ns - 1) return (float)knot(ns - 1);
    if (x == xj) return (float)knot(j);
    const double s0 = knot(j), s1 = knot(j + 1);
    const double slope = (s1 - s0) * fast_rcp(xn - xj);
    return (float)(slope * (x - xj) + s0);
}

// x / d for a divisor whose correctly rounded reciprocal r = RN(1 / d) is at hand: q = RN(x r) is within an ulp,
// the FMA residual x - q d is exact, and RN(q + residual r) is the correctly rounded quotient (Markstein 1990) —
// for finite operands and a quotient in the normal range, which is where audio samples over a window sum live
// (a zero stays a zero; a subnormal quotient may differ from the division in its last subnormal bit).
__device__ __forceinline__ float div_by(float x, float d, float r)
{
    const float q = x * r;
    return fmaf(fmaf(-q, d, x), r, q);
}

// gain = (1 / (peak + 1e-12))^normalize (GOOFER.py:1208-1214), `pk12` = the fp32 sum.  numpy's float64 power returns x for an
// exponent of exactly 1 and 1 for 0 (C99 pow), which is what the sampler always passes (normalize = 1, or 0 with the 'P0' flag);
// those two skip the library routine — 300 instructions and some forty live registers that, inlined between the two passes of
// k_note_finish, pushed the values it keeps across them out to scratch.
__device__ __forceinline__ float peak_gain(float pk12, float normalize)
{
    const double amt = (double)fminf(fmaxf(normalize, 0.f), 1.f);
    const double x = 1.0 / (double)pk12;
    if (amt == 1.0) return (float)x;
    if (amt == 0.0) return 1.0f;
    return (float)pow(x, amt);
}
