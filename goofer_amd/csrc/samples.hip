// Per-sample kernels for gfx950: voicing-mask smoothing, stem gains, peak normalisation, V/B/U mix.
//
//   k_mask_short   mask[::4] -> Gaussian sigma/4 (fp64)              GOOFER.py:556-562 (smooth_mask_ds)
//   k_stem_gains   lerp-upsample the smoothed mask, scale the three stems, per-note peak
//                                                                    GOOFER.py:563-567, 1179-1193, 1210
//   k_apply_gain   gain = (1/peak)^normalize, reconstruct, V/B/U mix GOOFER.py:1208-1218, SillySampler.py:1142-1151
#include "common.h"

#define MASK_DS 4

// short-array slot of a note: floor(sample_off/4) + note  (capacity >= ceil(n/4))
__device__ __forceinline__ int64_t short_base(const int64_t *sample_off, int note) { return sample_off[note] / MASK_DS + note; }

__global__ __launch_bounds__(256) void k_mask_short(const float *__restrict__ mask, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total_short, const double *__restrict__ taps, int radius,
                                                    double *__restrict__ short_s)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_taps = reinterpret_cast<double *>(smem);
    __shared__ int s_lo;
    for (int i = threadIdx.x; i < 2 * radius + 1; i += blockDim.x) s_taps[i] = taps[i];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) {
        // largest note with short_base(note) <= g0
        int lo = 0, hi = n_notes;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (short_base(sample_off, mid) <= g0) lo = mid; else hi = mid;
        }
        s_lo = lo;
    }
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    if (g >= total_short) return;
    int note = s_lo;
    while (note + 1 < n_notes && short_base(sample_off, note + 1) <= g) ++note;
    const int64_t q = g - short_base(sample_off, note);
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
    const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
    if (q >= ns) return;
    const float *m = mask + base;
    double acc = 0.0;
    if (q >= radius && q + radius < ns) {
        for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)m[MASK_DS * (q + j - radius)];
    } else {
        for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)m[MASK_DS * reflect_index(q + j - radius, ns)];
    }
    short_s[g] = acc;
}

// np.linspace(0, 1, num, dtype=float32)[i] as a double
__device__ __forceinline__ double lin01_f32(int64_t i, int64_t num)
{
    if (num <= 1) return 0.0;
    if (i >= num - 1) return 1.0;
    double step = 1.0 / (double)(num - 1);
    return (double)(float)((double)i * step);
}

__device__ __forceinline__ float smooth_mask_at(const double *__restrict__ ss, int64_t ns, int64_t i, int64_t n)
{
    if (ns <= 1) return (float)ss[0];                 // single knot: constant (GOOFER.py:183-191)
    double x = lin01_f32(i, n);
    double step = 1.0 / (double)(ns - 1);
    int64_t j = (int64_t)floor(x / step);
    if (j > ns - 1) j = ns - 1;
    if (j < 0) j = 0;
    while (j + 1 <= ns - 1 && lin01_f32(j + 1, ns) <= x) ++j;
    while (j > 0 && lin01_f32(j, ns) > x) --j;
    if (j >= ns - 1) return (float)ss[ns - 1];
    double xj = lin01_f32(j, ns);
    if (x == xj) return (float)ss[j];
    double slope = (ss[j + 1] - ss[j]) / (lin01_f32(j + 1, ns) - xj);
    return (float)(slope * (x - xj) + ss[j]);
}

// In place on the three OLA outputs: harm already divided by the per-note spectrum max.
__global__ __launch_bounds__(256) void k_stem_gains(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    const double *__restrict__ short_s, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total_samples,
                                                    const goofer_note_params *__restrict__ params, float *__restrict__ note_peak)
{
    __shared__ int s_lo;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) s_lo = csr_find(sample_off, n_notes, g0);
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    float pk = 0.f;
    int note = -1;
    if (g < total_samples) {
        note = s_lo;
        while (sample_off[note + 1] <= g) ++note;
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
        const goofer_note_params p = params[note];
        float ms = smooth_mask_at(short_s + short_base(sample_off, note), ns, g - base, n);
        float b = (bre[g] * ms) * p.breath_strength;
        float u = (uv[g] * (1.0f - ms)) * p.uv_strength;
        float h = harm[g];
        bre[g] = b;
        uv[g] = u;
        pk = fabsf((h + u) + b);
    }
    // wave-level max when the whole wave sits in one note, else per-lane atomics
    int note0 = __shfl(note, 0, WAVE);
    bool uniform = __all(note == note0);
    if (uniform) {
        pk = wave_max(pk);
        if ((threadIdx.x & 63) == 0 && note >= 0) atomic_max_pos(note_peak + note, pk);
    } else if (note >= 0) {
        atomic_max_pos(note_peak + note, pk);
    }
}

__global__ __launch_bounds__(256) void k_apply_gain(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    float *__restrict__ rec, float *__restrict__ mix,
                                                    const int64_t *__restrict__ sample_off, int n_notes, int64_t total_samples,
                                                    const goofer_note_params *__restrict__ params, const float *__restrict__ note_peak)
{
    __shared__ int s_lo;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) s_lo = csr_find(sample_off, n_notes, g0);
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    if (g >= total_samples) return;
    int note = s_lo;
    while (sample_off[note + 1] <= g) ++note;
    const goofer_note_params p = params[note];
    float peak = note_peak[note] + 1e-12f;                     // fp32 add, like np.float32 + 1e-12
    double amt = (double)fminf(fmaxf(p.normalize, 0.f), 1.f);
    float gain = (float)pow(1.0 / (double)peak, amt);
    float h = harm[g], u = uv[g], b = bre[g];
    float comb = (h + u) + b;
    h *= gain; u *= gain; b *= gain;
    harm[g] = h; uv[g] = u; bre[g] = b;
    if (rec) rec[g] = comb * gain;
    if (mix) mix[g] = ((h * p.mix_harm + b * p.mix_breath) + u * p.mix_unvoiced) * p.volume;
}

int launch_mask_short(goofer_ctx *ctx, const float *mask, const int64_t *sample_off, int n_notes, int64_t total_samples,
                      const double *d_taps, int radius, double *short_s, hipStream_t st)
{
    int64_t total_short = total_samples / MASK_DS + n_notes;
    if (total_short <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_mask_short, dim3((unsigned)((total_short + 255) / 256)), dim3(256), sizeof(double) * (2 * radius + 1), st,
                       mask, sample_off, n_notes, total_short, d_taps, radius, short_s);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_stem_gains(goofer_ctx *ctx, float *harm, float *uv, float *bre, const double *short_s, const int64_t *sample_off,
                      int n_notes, int64_t total_samples, const goofer_note_params *params, float *note_peak, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_stem_gains, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, harm, uv, bre, short_s,
                       sample_off, n_notes, total_samples, params, note_peak);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_apply_gain(goofer_ctx *ctx, float *harm, float *uv, float *bre, float *rec, float *mix, const int64_t *sample_off,
                      int n_notes, int64_t total_samples, const goofer_note_params *params, const float *note_peak, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_apply_gain, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, harm, uv, bre, rec, mix,
                       sample_off, n_notes, total_samples, params, note_peak);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
