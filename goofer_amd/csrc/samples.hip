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

// Each wave owns 64 consecutive short samples.  When they all belong to one note (the common case) the
// wave first parks the 64 + 2*radius decimated mask values it needs in LDS (each value fetched once
// instead of 2*radius+1 times); taps are shared by the block.  fp64 accumulate in tap order.
#define MS_MAXWIN 1152     // floats per wave: 64 + 2*radius, radius <= 544

__global__ __launch_bounds__(256) void k_mask_short(const float *__restrict__ mask, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total_short, const double *__restrict__ taps, int radius,
                                                    double *__restrict__ short_s)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_taps = reinterpret_cast<double *>(smem);
    float *s_win = reinterpret_cast<float *>(s_taps + (2 * radius + 1)) + (threadIdx.x >> 6) * MS_MAXWIN;
    __shared__ int s_lo;
    for (int i = threadIdx.x; i < 2 * radius + 1; i += blockDim.x) s_taps[i] = taps[i];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_notes;      // largest note with short_base(note) <= g0
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (short_base(sample_off, mid) <= g0) lo = mid; else hi = mid;
        }
        s_lo = lo;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t g = g0 + threadIdx.x;
    int note = s_lo;
    while (note + 1 < n_notes && short_base(sample_off, note + 1) <= g) ++note;
    const int64_t q = g - short_base(sample_off, note);
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
    const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
    const bool live = g < total_short && q < ns;
    const float *m = mask + base;

    // wave-level staging: valid when every lane of the wave sits in the same note
    const int note0 = __shfl(note, 0, WAVE);
    const int64_t q0 = __shfl((long long)q, 0, WAVE);
    const bool same = __all(note == note0) && (2 * radius + WAVE <= MS_MAXWIN);
    if (same) {
        const int win = WAVE + 2 * radius;
        for (int w = lane; w < win; w += WAVE) {
            const int64_t idx = reflect_index(q0 - radius + w, ns);
            s_win[w] = ns > 0 ? m[MASK_DS * idx] : 0.f;
        }
        wave_lds_sync();
        if (live) {
            double acc = 0.0;
            const float *x = s_win + (int)(q - q0);
            for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)x[j];
            short_s[g] = acc;
        }
        return;
    }
    if (!live) return;
    double acc = 0.0;
    for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)m[MASK_DS * reflect_index(q + j - radius, ns)];
    short_s[g] = acc;
}

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

// per-note constants of the mask upsampler: 1/(n-1) and 1/(ns-1) as true divisions (numpy's linspace step)
__global__ void k_note_steps(const int64_t *__restrict__ sample_off, int n_notes, double *__restrict__ steps)
{
    int note = blockIdx.x * blockDim.x + threadIdx.x;
    if (note >= n_notes) return;
    const int64_t n = sample_off[note + 1] - sample_off[note];
    const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
    steps[2 * note] = n > 1 ? 1.0 / (double)(n - 1) : 0.0;
    steps[2 * note + 1] = ns > 1 ? 1.0 / (double)(ns - 1) : 0.0;
}

// In place on the three OLA outputs: harm already divided by the per-note spectrum max.
// Each thread owns 4 consecutive samples (16-byte loads/stores, four independent interpolations in
// flight); a block of 256 threads covers 1024 samples and issues ONE atomic when it lies in one note.
#define SPT 4

__global__ __launch_bounds__(256) void k_stem_gains(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    const double *__restrict__ short_s, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total_samples,
                                                    const goofer_note_params *__restrict__ params, float *__restrict__ note_peak,
                                                    const double *__restrict__ steps)
{
    __shared__ int s_pair[2];
    __shared__ float s_red[4];
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x * SPT);
    int lo, hi;
    if (threadIdx.x == 0) {
        s_pair[0] = csr_find(sample_off, n_notes, g0);
        int64_t gl = g0 + (int64_t)blockDim.x * SPT - 1;
        if (gl > total_samples - 1) gl = total_samples - 1;
        s_pair[1] = csr_find(sample_off, n_notes, gl);
    }
    __syncthreads();
    lo = __builtin_amdgcn_readfirstlane(s_pair[0]);
    hi = __builtin_amdgcn_readfirstlane(s_pair[1]);
    const int64_t g = g0 + (int64_t)threadIdx.x * SPT;

    auto one = [&](int note, int64_t gi, float h, float u_in, float b_in, float &u_out, float &b_out) -> float {
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
        const float ms = smooth_mask_at(short_s + short_base(sample_off, note), ns, gi - base, n, steps[2 * note], steps[2 * note + 1]);
        b_out = (b_in * ms) * params[note].breath_strength;
        u_out = (u_in * (1.0f - ms)) * params[note].uv_strength;
        return fabsf((h + u_out) + b_out);
    };

    if (lo == hi && g + SPT <= total_samples) {
        const float4 h4 = *reinterpret_cast<const float4 *>(harm + g);
        const float4 u4 = *reinterpret_cast<const float4 *>(uv + g);
        const float4 b4 = *reinterpret_cast<const float4 *>(bre + g);
        float4 uo, bo;
        float pk = one(lo, g, h4.x, u4.x, b4.x, uo.x, bo.x);
        pk = fmaxf(pk, one(lo, g + 1, h4.y, u4.y, b4.y, uo.y, bo.y));
        pk = fmaxf(pk, one(lo, g + 2, h4.z, u4.z, b4.z, uo.z, bo.z));
        pk = fmaxf(pk, one(lo, g + 3, h4.w, u4.w, b4.w, uo.w, bo.w));
        *reinterpret_cast<float4 *>(uv + g) = uo;
        *reinterpret_cast<float4 *>(bre + g) = bo;
        pk = wave_max(pk);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pk;
    } else {
        float pk_lo = 0.f;
        for (int k = 0; k < SPT; ++k) {
            const int64_t gi = g + k;
            if (gi >= total_samples) break;
            int note = lo;
            while (sample_off[note + 1] <= gi) ++note;
            float uo, bo;
            const float pk = one(note, gi, harm[gi], uv[gi], bre[gi], uo, bo);
            uv[gi] = uo;
            bre[gi] = bo;
            if (lo == hi) pk_lo = fmaxf(pk_lo, pk); else atomic_max_pos(note_peak + note, pk);
        }
        pk_lo = wave_max(pk_lo);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pk_lo;
    }
    __syncthreads();
    if (lo == hi && threadIdx.x == 0)
        atomic_max_pos(note_peak + lo, fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
}

// Overlap-add of the three stems + mask upsample + stem gains + per-note peak in ONE pass over the samples
// (k_ola_gather x3 + k_stem_gains): each output sample gathers its covering windowed frames of the three
// stems in ascending frame order (the reference's fp32 accumulation order), normalises by the summed
// squared window, then applies the gains.  Saves one full write + read of the three stems.
__global__ __launch_bounds__(256) void k_ola3_gains(const float *__restrict__ fr_h, const float *__restrict__ fr_u,
                                                    const float *__restrict__ fr_b, const float *__restrict__ win_sq,
                                                    const float *__restrict__ note_mag, const double *__restrict__ short_s,
                                                    const int64_t *__restrict__ sample_off, const int64_t *__restrict__ frame_off,
                                                    int n_notes, int64_t total_samples, int n_fft, int hop,
                                                    const goofer_note_params *__restrict__ params, const double *__restrict__ steps,
                                                    float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    float *__restrict__ note_peak)
{
    __shared__ int s_pair[2];
    __shared__ float s_red[4];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo_n, hi_n;
    block_note_range(sample_off, n_notes, g0, total_samples, s_pair, lo_n, hi_n);
    const int64_t g = g0 + threadIdx.x;
    const bool live = g < total_samples;

    auto body = [&](int note) -> float {
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t i = g - base;
        const int64_t fbase = frame_off[note];
        const int64_t T = frame_off[note + 1] - fbase;
        float h = 0.f, u = 0.f, b = 0.f;
        if (i < (int64_t)hop * (T - 1)) {
            const int64_t p = i + n_fft / 2;
            int64_t lo = p - n_fft + 1;
            lo = lo <= 0 ? 0 : (lo + hop - 1) / hop;
            int64_t hi = p / hop;
            if (hi > T - 1) hi = T - 1;
            float ws = 0.f;
            for (int64_t fr = lo; fr <= hi; ++fr) {
                const int j = (int)(p - fr * hop);
                const int64_t at = (fbase + fr) * n_fft + j;
                h += fr_h[at];
                u += fr_u[at];
                b += fr_b[at];
                ws += win_sq[j];
            }
            if (ws > 1e-9f) { h /= ws; u /= ws; b /= ws; }
            h = h / note_mag[note];
        }
        const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
        const float ms = smooth_mask_at(short_s + short_base(sample_off, note), ns, i, n, steps[2 * note], steps[2 * note + 1]);
        b = (b * ms) * params[note].breath_strength;
        u = (u * (1.0f - ms)) * params[note].uv_strength;
        harm[g] = h;
        uv[g] = u;
        bre[g] = b;
        return fabsf((h + u) + b);
    };
    if (lo_n == hi_n) {
        float pk = live ? body(lo_n) : 0.f;
        pk = wave_max(pk);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pk;
        __syncthreads();
        if (threadIdx.x == 0) atomic_max_pos(note_peak + lo_n, fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
    } else if (live) {
        int note = lo_n;
        while (sample_off[note + 1] <= g) ++note;
        atomic_max_pos(note_peak + note, body(note));
    }
}

__global__ __launch_bounds__(256) void k_apply_gain(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    float *__restrict__ rec, float *__restrict__ mix,
                                                    const int64_t *__restrict__ sample_off, int n_notes, int64_t total_samples,
                                                    const goofer_note_params *__restrict__ params, const float *__restrict__ note_peak)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x * SPT);
    if (threadIdx.x == 0) {
        s_pair[0] = csr_find(sample_off, n_notes, g0);
        int64_t gl = g0 + (int64_t)blockDim.x * SPT - 1;
        if (gl > total_samples - 1) gl = total_samples - 1;
        s_pair[1] = csr_find(sample_off, n_notes, gl);
    }
    __syncthreads();
    const int lo = __builtin_amdgcn_readfirstlane(s_pair[0]), hi = __builtin_amdgcn_readfirstlane(s_pair[1]);
    const int64_t g = g0 + (int64_t)threadIdx.x * SPT;
    if (g >= total_samples) return;

    auto gain_of = [&](int note) -> float {
        const float peak = note_peak[note] + 1e-12f;                 // fp32 add, like np.float32 + 1e-12
        const double amt = (double)fminf(fmaxf(params[note].normalize, 0.f), 1.f);
        return (float)pow(1.0 / (double)peak, amt);
    };
    auto mixdown = [&](int note, float h, float u, float b) -> float {
        return ((h * params[note].mix_harm + b * params[note].mix_breath) + u * params[note].mix_unvoiced) * params[note].volume;
    };
    if (lo == hi && g + SPT <= total_samples) {
        const float gain = gain_of(lo);
        float4 h = *reinterpret_cast<const float4 *>(harm + g);
        float4 u = *reinterpret_cast<const float4 *>(uv + g);
        float4 b = *reinterpret_cast<const float4 *>(bre + g);
        const float4 comb = make_float4((h.x + u.x) + b.x, (h.y + u.y) + b.y, (h.z + u.z) + b.z, (h.w + u.w) + b.w);
        h.x *= gain; h.y *= gain; h.z *= gain; h.w *= gain;
        u.x *= gain; u.y *= gain; u.z *= gain; u.w *= gain;
        b.x *= gain; b.y *= gain; b.z *= gain; b.w *= gain;
        *reinterpret_cast<float4 *>(harm + g) = h;
        *reinterpret_cast<float4 *>(uv + g) = u;
        *reinterpret_cast<float4 *>(bre + g) = b;
        if (rec) *reinterpret_cast<float4 *>(rec + g) = make_float4(comb.x * gain, comb.y * gain, comb.z * gain, comb.w * gain);
        if (mix)
            *reinterpret_cast<float4 *>(mix + g) = make_float4(mixdown(lo, h.x, u.x, b.x), mixdown(lo, h.y, u.y, b.y),
                                                               mixdown(lo, h.z, u.z, b.z), mixdown(lo, h.w, u.w, b.w));
        return;
    }
    for (int k = 0; k < SPT; ++k) {
        const int64_t gi = g + k;
        if (gi >= total_samples) break;
        int note = lo;
        while (sample_off[note + 1] <= gi) ++note;
        const float gain = gain_of(note);
        float h = harm[gi], u = uv[gi], b = bre[gi];
        const float comb = (h + u) + b;
        h *= gain; u *= gain; b *= gain;
        harm[gi] = h; uv[gi] = u; bre[gi] = b;
        if (rec) rec[gi] = comb * gain;
        if (mix) mix[gi] = mixdown(note, h, u, b);
    }
}

// per-note max |harm + uv + bre| (used when the volume jitter changes stems after the fused OLA/gain pass)
__global__ __launch_bounds__(256) void k_stem_peak(const float *__restrict__ harm, const float *__restrict__ uv, const float *__restrict__ bre,
                                                   const int64_t *__restrict__ sample_off, int n_notes, int64_t total,
                                                   float *__restrict__ note_peak)
{
    __shared__ int s_pair[2];
    __shared__ float s_red[4];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    float pk = g < total ? fabsf((harm[g] + uv[g]) + bre[g]) : 0.f;
    if (lo == hi) {
        pk = wave_max(pk);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pk;
        __syncthreads();
        if (threadIdx.x == 0) atomic_max_pos(note_peak + lo, fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
    } else if (g < total) {
        int note = lo;
        while (sample_off[note + 1] <= g) ++note;
        atomic_max_pos(note_peak + note, pk);
    }
}

int launch_stem_peak(goofer_ctx *ctx, const float *harm, const float *uv, const float *bre, const int64_t *sample_off, int n_notes,
                     int64_t total, float *note_peak, hipStream_t st)
{
    if (total <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_stem_peak, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, harm, uv, bre, sample_off, n_notes, total,
                       note_peak);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_mask_short(goofer_ctx *ctx, const float *mask, const int64_t *sample_off, int n_notes, int64_t total_samples,
                      const double *d_taps, int radius, double *short_s, hipStream_t st)
{
    int64_t total_short = total_samples / MASK_DS + n_notes;
    if (total_short <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_mask_short, dim3((unsigned)((total_short + 255) / 256)), dim3(256),
                       sizeof(double) * (2 * radius + 1) + sizeof(float) * 4 * MS_MAXWIN, st, mask, sample_off, n_notes, total_short,
                       d_taps, radius, short_s);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_stem_gains(goofer_ctx *ctx, float *harm, float *uv, float *bre, const double *short_s, const int64_t *sample_off,
                      int n_notes, int64_t total_samples, const goofer_note_params *params, float *note_peak, double *steps,
                      hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_stem_gains, dim3((unsigned)((total_samples + 1023) / 1024)), dim3(256), 0, st, harm, uv, bre, short_s,
                       sample_off, n_notes, total_samples, params, note_peak, steps);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_apply_gain(goofer_ctx *ctx, float *harm, float *uv, float *bre, float *rec, float *mix, const int64_t *sample_off,
                      int n_notes, int64_t total_samples, const goofer_note_params *params, const float *note_peak, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_apply_gain, dim3((unsigned)((total_samples + 1023) / 1024)), dim3(256), 0, st, harm, uv, bre, rec, mix,
                       sample_off, n_notes, total_samples, params, note_peak);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_ola3_gains(goofer_ctx *ctx, const float *fr_h, const float *fr_u, const float *fr_b, const float *note_mag,
                      const double *short_s, const int64_t *sample_off, const int64_t *frame_off, int n_notes, int64_t total_samples,
                      const goofer_note_params *params, double *steps, float *harm, float *uv, float *bre, float *note_peak,
                      hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_ola3_gains, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, fr_h, fr_u, fr_b, p.win_sq,
                       note_mag, short_s, sample_off, frame_off, n_notes, total_samples, p.n_fft, p.hop, params, steps, harm, uv, bre,
                       note_peak);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
