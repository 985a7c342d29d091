// Jitter oscillators of gf.synthesize (flags sh / sr) for gfx950.
//
//   k_gauss_samples   gaussian_filter1d along the SAMPLE axis of every note (numpy 'reflect' padding), fp64
//                     accumulate: smoothed noise (sigma = sr/(6*speed): 393-589 taps) and the sigma-20
//                     voicing mask of the volume jitter                      GOOFER.py:654, 667, 1189
//   k_note_absmax     per-note max(|x| + 1e-6)  ("noise /= np.max(np.abs(noise) + 1e-6)")   GOOFER.py:655, 668
//   k_f0_jitter       f0 *= 1 + (jitter - 1) * mask,  jitter = 1 + noise/max * strength    GOOFER.py:669, 1071
//   k_volume_jitter   harm *= 1 + (jh - 1) * vjm ;  bre *= 1 + (jb - 1) * vjm              GOOFER.py:1187-1191
// The random draws themselves come from the host (the reference uses the legacy global np.random stream);
// each wave stages its window in LDS so every input sample is fetched once per 64 outputs.
#include "common.h"


// The taps are consumed GS_CHUNK at a time: a chunk of taps and the 256 + GS_CHUNK window values it meets are staged in LDS,
// every thread adds its products in tap order, and the running sum stays in its register across chunks — the same additions
// in the same order as one long FIR, for any radius (the 'pd' smoothing at 96 kHz is sigma 960, gf.synthesize's roughness
// slew sigma 1920: radii of 3 840 and 7 680 taps).
#define GS_CHUNK 2048

template <typename Tin>
__global__ __launch_bounds__(256) void k_gauss_samples(const Tin *__restrict__ in, const int64_t *__restrict__ sample_off, int n_notes,
                                                       int64_t total, const double *__restrict__ taps, int radius,
                                                       const unsigned char *__restrict__ note_on, double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_taps = reinterpret_cast<double *>(smem);      // [chunk]
    const int nt = 2 * radius + 1;
    const int chunk = nt < GS_CHUNK ? nt : GS_CHUNK;
    double *s_win = s_taps + chunk;                           // [256 + chunk]
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);   // includes a __syncthreads
    const int64_t g = g0 + threadIdx.x;
    if (lo == hi) {
        if (note_on && !note_on[lo]) return;                  // whole block: nothing to do for this note
        const int64_t base = sample_off[lo], n = sample_off[lo + 1] - base;
        const int64_t i0 = g0 - base;
        double acc = 0.0;
        for (int j0 = 0; j0 < nt; j0 += chunk) {
            const int cn = nt - j0 < chunk ? nt - j0 : chunk;
            for (int i = threadIdx.x; i < cn; i += blockDim.x) s_taps[i] = taps[j0 + i];
            const int win = (int)blockDim.x + cn - 1;           // window values the chunk's taps meet for this tile
            for (int w = threadIdx.x; w < win; w += blockDim.x)
                s_win[w] = (double)in[base + reflect_index(i0 - radius + j0 + w, n)];
            __syncthreads();
            if (g < total) {
                const double *x = s_win + threadIdx.x;
                for (int j = 0; j < cn; ++j) acc += s_taps[j] * x[j];
            }
            __syncthreads();
        }
        if (g < total) out[g] = acc;
        return;
    }
    if (g >= total) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    if (note_on && !note_on[note]) return;
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base, i = g - base;
    double acc = 0.0;
    for (int j = 0; j < nt; ++j) acc += taps[j] * (double)in[base + reflect_index(i + j - radius, n)];
    out[g] = acc;
}

__global__ __launch_bounds__(256) void k_note_absmax(const double *__restrict__ x, const int64_t *__restrict__ sample_off, int n_notes,
                                                     int64_t total, const unsigned char *__restrict__ note_on,
                                                     unsigned long long *__restrict__ max_bits)
{
    __shared__ int s_pair[2];
    __shared__ double s_red[4];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    double v = 0.0;
    int note = lo;
    if (g < total) {
        while (sample_off[note + 1] <= g) ++note;
        if (!note_on || note_on[note]) v = fabs(x[g]) + 1e-6;
    }
    if (lo == hi) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
            if (m > 0.0) atomicMax(max_bits + lo, (unsigned long long)__double_as_longlong(m));
        }
    } else if (g < total && v > 0.0) {
        atomicMax(max_bits + note, (unsigned long long)__double_as_longlong(v));
    }
}

// f0 (fp32, in place) *= 1 + ((1 + noise/max*strength) - 1) * mask, evaluated in fp64 then rounded to fp32.
// f0_64 (gf.synthesize behind its time stretch: f0_interp is a float64 array there, GOOFER.py:1053): that array takes the product,
// and the fp32 one becomes its cast — what pulse_train_numba is handed (:1074).
__global__ __launch_bounds__(256) void k_f0_jitter(float *__restrict__ f0, double *__restrict__ f0_64, const float *__restrict__ mask,
                                                   const double *__restrict__ noise_s,
                                                   const unsigned long long *__restrict__ max_bits, const int64_t *__restrict__ sample_off,
                                                   int n_notes, int64_t total, const goofer_note_params *__restrict__ params, int which)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    const double strength = which == 0 ? params[note].f0_jitter                                   // :1071 / :1080
                                       : (params[note].subharm_weight > 0.f ? params[note].subharm_f0_jitter : 0.0);
    if (!(strength > 0.0)) return;
    const double mx = __longlong_as_double((long long)max_bits[note]);
    const double jit = 1.0 + (noise_s[g] / mx) * strength;     // python-float strength: fp64 like the reference
    const double fac = 1.0 + ((jit - 1.0) * (double)mask[g]);
    if (f0_64) {
        const double v = f0_64[g] * fac;
        f0_64[g] = v;
        f0[g] = (float)v;
    } else {
        f0[g] = (float)((double)f0[g] * fac);
    }
}

__global__ __launch_bounds__(256) void k_volume_jitter(float *__restrict__ harm, float *__restrict__ bre, const double *__restrict__ nh,
                                                       const double *__restrict__ nb, const double *__restrict__ vjm,
                                                       const unsigned long long *__restrict__ max_h,
                                                       const unsigned long long *__restrict__ max_b, const int64_t *__restrict__ sample_off,
                                                       int n_notes, int64_t total, const goofer_note_params *__restrict__ params,
                                                       int vibrato, double speed, double sr)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    const float sh = params[note].vol_jitter_harm, sb = params[note].vol_jitter_breath;
    if (!(sh > 0.f) && !(sb > 0.f)) return;
    double jh, jb;
    if (vibrato) {   // volume_vibrato: zero-phase sinusoid, 0.1 s fade-in, clip [0.5, 1.5]   GOOFER.py:643-660
        const int64_t i = g - sample_off[note], n = sample_off[note + 1] - sample_off[note];
        double z = sin(((2.0 * 3.141592653589793) * speed) * ((double)i / sr) + 0.0);
        const int fade = (int)(0.1 * sr);
        if (fade < n && i < fade) z *= fade > 1 ? (i == fade - 1 ? 1.0 : (double)i * (1.0 / (double)(fade - 1))) : 0.0;
        jh = fmin(fmax(1.0 + z * (double)sh, 0.5), 1.5);
        jb = fmin(fmax(1.0 + z * (double)sb, 0.5), 1.5);
    } else {
        jh = 1.0 + (nh[g] / __longlong_as_double((long long)max_h[note])) * (double)sh;
        jb = 1.0 + (nb[g] / __longlong_as_double((long long)max_b[note])) * (double)sb;
    }
    const double m = vjm[g];
    harm[g] = (float)((double)harm[g] * (1.0 + (jh - 1.0) * m));
    bre[g] = (float)((double)bre[g] * (1.0 + (jb - 1.0) * m));
}

template <typename Tin>
int launch_gauss_samples(goofer_ctx *ctx, const Tin *in, const int64_t *sample_off, int n_notes, int64_t total, const double *d_taps,
                         int radius, const unsigned char *note_on, double *out, hipStream_t st)
{
    if (total <= 0) return GOOFER_OK;
    if (radius < 0) return goofer_fail(ctx, GOOFER_EINVAL, "negative gaussian radius");
    const int nt = 2 * radius + 1, chunk = nt < GS_CHUNK ? nt : GS_CHUNK;
    size_t lds = sizeof(double) * ((size_t)chunk + 256 + chunk);              // at most 36 KB
    hipLaunchKernelGGL(k_gauss_samples<Tin>, dim3((unsigned)((total + 255) / 256)), dim3(256), lds, st, in, sample_off, n_notes, total,
                       d_taps, radius, note_on, out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
template int launch_gauss_samples<double>(goofer_ctx *, const double *, const int64_t *, int, int64_t, const double *, int,
                                          const unsigned char *, double *, hipStream_t);
template int launch_gauss_samples<float>(goofer_ctx *, const float *, const int64_t *, int, int64_t, const double *, int,
                                         const unsigned char *, double *, hipStream_t);

int launch_note_absmax(goofer_ctx *ctx, const double *x, const int64_t *sample_off, int n_notes, int64_t total,
                       const unsigned char *note_on, unsigned long long *max_bits, hipStream_t st)
{
    if (total <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_note_absmax, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, sample_off, n_notes, total, note_on,
                       max_bits);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_f0_jitter(goofer_ctx *ctx, float *f0, double *f0_64, const float *mask, const double *noise_s, const unsigned long long *max_bits,
                     const int64_t *sample_off, int n_notes, int64_t total, const goofer_note_params *params, int which, hipStream_t st)
{
    if (total <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_f0_jitter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, f0, f0_64, mask, noise_s, max_bits, sample_off,
                       n_notes, total, params, which);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_volume_jitter(goofer_ctx *ctx, float *harm, float *bre, const double *nh, const double *nb, const double *vjm,
                         const unsigned long long *max_h, const unsigned long long *max_b, const int64_t *sample_off, int n_notes,
                         int64_t total, const goofer_note_params *params, int vibrato, double speed, hipStream_t st)
{
    if (total <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_volume_jitter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, harm, bre, nh, nb, vjm, max_h, max_b,
                       sample_off, n_notes, total, params, vibrato, speed, (double)ctx->plan.sr);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
