// LF glottal pulse train for gfx950 — replaces gf.pulse_train_numba (GOOFER.py:473-554).
//
// The onset positions are a discontinuous function of a strictly sequential fp64 sum
// (`total_phase += f0[i]/sr`, GOOFER.py:491-493): re-associating it moves onsets by a sample
// (SURVEY.md §7.3-1).  So the work is split three ways:
//   k_phase_inc     fully parallel: inc[i] = (double)f0[i] / sr  (true IEEE division)
//   k_pulse_onsets  one LANE per note walks its increments in order (fp64 add + compare only) and
//                   emits a compact onset list (sample, T0, period, running max of sample+T0)
//   k_pulse_place   fully parallel gather: every output sample sums, in ascending onset order, the
//                   LF shapes that cover it — no atomics, same fp32 accumulation order as the
//                   reference's `pulse[j] += cache[k]`
// Shapes are evaluated on the fly in fp64 (numba's typing) and normalised by a per-T0 peak table.
#include "common.h"

#define PT_RA 0.02
#define PT_RG 1.7
#define PT_RK 0.8
#define PT_PI 3.141592653589793

// un-normalised LF shape sample k of a T0-sample pulse with period T (GOOFER.py:509-519), rounded
// to fp32 like the reference's `buf[j] = ...` store
__device__ __forceinline__ float lf_raw(int k, int T0, double T)
{
    double ti = ((double)k * T) / (double)T0;
    double Tp = PT_RA * T;
    double Tc = Tp + PT_RK * (T - Tp);
    double v;
    if (ti < Tp) {
        double s = sin(PT_PI * ti / (2.0 * Tp + 1e-12));
        v = s * s;
    } else if (ti < Tc) {
        double tau = (ti - Tp) / (Tc - Tp + 1e-12);
        v = exp(-PT_RG * tau) * cos(PT_PI * tau / 2.0);
    } else {
        v = 0.0;
    }
    return (float)v;
}

// peak[T0] = max_k |buf[k]| for the nominal period T = T0/sr; one block per T0
__global__ __launch_bounds__(256) void k_pulse_peak(float *__restrict__ peak, double sr)
{
    __shared__ float red[4];
    int T0 = blockIdx.x;
    float m = 0.f;
    if (T0 >= 3) {
        double T = (double)T0 / sr;
        for (int k = threadIdx.x; k < T0; k += blockDim.x) m = fmaxf(m, fabsf(lf_raw(k, T0, T)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) peak[T0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int launch_pulse_peak(goofer_ctx *ctx, float *peak, double sr, hipStream_t st)
{
    hipLaunchKernelGGL(k_pulse_peak, dim3(8193), dim3(256), 0, st, peak, sr);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

__global__ __launch_bounds__(256) void k_phase_inc(const float *__restrict__ f0, int64_t n, double sr, float scale,
                                                   double *__restrict__ inc)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) inc[g] = (double)(f0[g] * scale) / sr;
}

struct onset_t {
    int32_t i;        // sample index inside the note
    int32_t T0;       // pulse length in samples
    int32_t end_max;  // max over onsets <= this one of (i + T0): monotone, bounds the look-back
    int32_t pad;
    double T;         // period used for the shape
};

// One lane per note.  cap_off[note] = first slot of the note's onset list; capacity = cap_off[note+1]-cap_off[note].
__global__ __launch_bounds__(64) void k_pulse_onsets(const float *__restrict__ f0, float f0_scale, const double *__restrict__ inc,
                                                     const int64_t *__restrict__ sample_off, int n_notes, double sr,
                                                     onset_t *__restrict__ onsets, int32_t *__restrict__ onset_cnt,
                                                     int32_t *__restrict__ overflow)
{
    int note = blockIdx.x * 64 + threadIdx.x;
    if (note >= n_notes) return;
    const int64_t base = sample_off[note];
    const int64_t n = sample_off[note + 1] - base;
    const int64_t obase = base / 2 + 16 * (int64_t)note;
    const int64_t cap = (sample_off[note + 1] / 2 + 16 * (int64_t)(note + 1)) - obase;
    const float *f = f0 + base;
    const double *a = inc + base;
    double phase = 0.0, next_k = 1.0, last = 160.0;
    int32_t cnt = 0, end_max = 0;
    for (int64_t i = 0; i < n; ++i) {
        float fi = f[i] * f0_scale;
        if (fi > 1e-6f) last = (double)fi;
        phase += a[i];
        while (phase >= next_k) {
            double T = 1.0 / fmax(last, 1e-6);
            long T0 = (long)rint(sr * T);       // round-half-even, like Python round()
            T0 = T0 < 3 ? 3 : (T0 > 8192 ? 8192 : T0);
            if (cnt < cap) {
                int32_t e = (int32_t)i + (int32_t)T0;
                end_max = e > end_max ? e : end_max;
                onset_t o;
                o.i = (int32_t)i; o.T0 = (int32_t)T0; o.end_max = end_max; o.pad = 0; o.T = T;
                onsets[obase + cnt] = o;
            } else {
                *overflow = 1;
            }
            ++cnt;
            next_k += 1.0;
        }
    }
    onset_cnt[note] = cnt < cap ? cnt : (int32_t)cap;
}

__global__ __launch_bounds__(256) void k_pulse_place(const onset_t *__restrict__ onsets, const int32_t *__restrict__ onset_cnt,
                                                     const float *__restrict__ peak, const int64_t *__restrict__ sample_off,
                                                     int n_notes, int64_t total_samples, float *__restrict__ pulse)
{
    __shared__ int s_lo;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) s_lo = csr_find(sample_off, n_notes, g0);
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    if (g >= total_samples) return;
    int note = s_lo;
    while (sample_off[note + 1] <= g) ++note;
    const int32_t j = (int32_t)(g - sample_off[note]);
    const onset_t *ol = onsets + (sample_off[note] / 2 + 16 * (int64_t)note);
    const int cnt = onset_cnt[note];
    float acc = 0.f;
    // last onset with i <= j
    int lo = -1, hi = cnt;   // ol[lo].i <= j < ol[hi].i
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ol[mid].i <= j) lo = mid; else hi = mid;
    }
    if (lo >= 0) {
        int first = lo;
        while (first > 0 && ol[first - 1].end_max > j) --first;
        for (int k = first; k <= lo; ++k) {
            onset_t o = ol[k];
            int d = j - o.i;
            if (d < o.T0) {
                float raw = lf_raw(d, o.T0, o.T);
                double m = (double)peak[o.T0];
                float v = m > 0.0 ? (float)((double)raw / m) : raw;
                acc += v;
            }
        }
    }
    pulse[g] = acc;
}

int launch_phase_inc(goofer_ctx *ctx, const float *f0, float f0_scale, int64_t total_samples, double *inc, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_phase_inc, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, f0, total_samples,
                       (double)ctx->plan.sr, f0_scale, inc);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_pulse_onsets(goofer_ctx *ctx, const float *f0, float f0_scale, const double *inc, const int64_t *sample_off,
                        int n_notes, onset_t *onsets, int32_t *onset_cnt, int32_t *overflow, hipStream_t st)
{
    if (n_notes <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_pulse_onsets, dim3((n_notes + 63) / 64), dim3(64), 0, st, f0, f0_scale, inc, sample_off, n_notes,
                       (double)ctx->plan.sr, onsets, onset_cnt, overflow);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_pulse_place(goofer_ctx *ctx, const onset_t *onsets, const int32_t *onset_cnt, const int64_t *sample_off, int n_notes,
                       int64_t total_samples, float *pulse, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_pulse_place, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, onsets, onset_cnt,
                       ctx->plan.pulse_peak, sample_off, n_notes, total_samples, pulse);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_pulse_train(goofer_ctx *ctx, const float *f0, float f0_scale, const int64_t *sample_off, int n_notes,
                       int64_t total_samples, float *pulse, double *inc, onset_t *onsets, int32_t *onset_cnt,
                       int32_t *overflow, hipStream_t st)
{
    if (total_samples <= 0 || n_notes <= 0) return GOOFER_OK;
    int rc;
    if ((rc = launch_phase_inc(ctx, f0, f0_scale, total_samples, inc, st))) return rc;
    if ((rc = launch_pulse_onsets(ctx, f0, f0_scale, inc, sample_off, n_notes, onsets, onset_cnt, overflow, st))) return rc;
    return launch_pulse_place(ctx, onsets, onset_cnt, sample_off, n_notes, total_samples, pulse, st);
}
