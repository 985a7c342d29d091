// Sample-domain post chain of the resampler for gfx950 (SillySampler.py:95-174, 857-881, 1037-1182).
//
//   k_onepole_cascade  dynamic_butter_filter: cascades of time-varying one-pole sections.  y_i = A_i y_{i-1} + B_i is
//                      an affine recurrence, so a 2048-sample tile is solved by a workgroup scan of (A, B) pairs
//                      (fp64), all sections of the cascade chained in registers: one read + one write of the signal.
//   k_post_layers      su: harm += h_su * gain;  sj: harm = (1 - m) harm + m h_sj
//   k_post_fry         harm / bre blended with their 200 Hz high-passed copies under the fry mask
//   k_post_sd          bre *= 1 + (vibrato - 1) * gauss(mask, 20);  bre *= 1 + sd / 10
//   k_note_sumsq       per-note sum of (harm + bre)^2 (gf.rms)
//   k_post_tension     st > 0: harm += hp * (1 + 20 t), bre *= 1 - t;  k_post_scale: both *= rms_before / rms_after
//   k_post_mix         ((harm V + bre B) + uv U) * volume, then the sa blend and the pd gain
//   k_percentile95     np.percentile(|x|, 95) per note by radix select on the fp64 bit patterns
//   k_dyn_gain         pd: 10^(12 |pd| clip(bend_s / ref, -1, 1) / 20) -> fp32 -> clip -> 1 + (g - 1) * gauss(mask)
#include "common.h"

constexpr int OP_PER = 8;
constexpr int OP_TILE = 256 * OP_PER;
constexpr int OP_MAXORD = 12;

struct aff {
    double A, B;
};
// later o earlier
__device__ __forceinline__ aff aff_after(aff later, aff earlier) { return {later.A * earlier.A, later.A * earlier.B + later.B}; }

__global__ __launch_bounds__(256) void k_onepole_cascade(const float *__restrict__ src, float *__restrict__ dst,
                                                         const float *__restrict__ f0, const goofer_onepole_job *__restrict__ jobs,
                                                         double sr)
{
    __shared__ double s_tot[4][2];
    __shared__ float s_wlast[4];
    const goofer_onepole_job job = jobs[blockIdx.x];
    const int n = job.n;
    if (n <= 0) return;
    const float *__restrict__ x0 = src + job.src_off;
    float *__restrict__ y0 = dst + job.dst_off;
    const float *__restrict__ fr = f0 + job.f0_off;
    const int order = job.order < 1 ? 1 : (job.order > OP_MAXORD ? OP_MAXORD : job.order);
    const bool hp = job.highpass != 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double cf = (double)job.cutoff_factor;
    const float floor_hz = hp ? 20.0f : 60.0f;
    const double ceil_hz = 0.45 * sr;
    const double two_pi = 2.0 * 3.141592653589793;

    double cy[OP_MAXORD];
    float cx[OP_MAXORD];
#pragma unroll
    for (int s = 0; s < OP_MAXORD; ++s) { cy[s] = 0.0; cx[s] = 0.f; }

    for (int t0 = 0; t0 < n; t0 += OP_TILE) {
        const int base = t0 + tid * OP_PER;
        float x[OP_PER];
        double al[OP_PER];
        // per-sample coefficient: 5-tap box over the edge-padded f0 reference, fp32 like np.convolve on fp32
        {
            float v[OP_PER + 4];
#pragma unroll
            for (int k = 0; k < OP_PER + 4; ++k) {
                int idx = base + k - 2;
                idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
                float f = job.f0_mode == 2 ? 1.0f : fr[idx];
                if (job.f0_mode == 1) f = fmaxf(f, 120.0f);
                v[k] = f;
            }
#pragma unroll
            for (int k = 0; k < OP_PER; ++k) {
                const float c = 0.2f;
                float f0s = v[k] * c;
                f0s = f0s + v[k + 1] * c;
                f0s = f0s + v[k + 2] * c;
                f0s = f0s + v[k + 3] * c;
                f0s = f0s + v[k + 4] * c;
                float fc = f0s > 0.0f ? (float)((double)f0s * cf) : (float)cf;
                fc = fmaxf(fc, floor_hz);
                fc = (float)fmin((double)fc, ceil_hz);
                const double w = two_pi * (double)fc;
                al[k] = (double)(float)(hp ? sr / (w + sr) : w / (w + sr));
                x[k] = base + k < n ? x0[base + k] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < OP_MAXORD; ++s) {
            if (s < order) {
            // the sample before this thread's first one (high-pass only): neighbour lane, previous wave, or carry
            float xprev = 0.f;
            if (hp) {
                if (lane == 63) s_wlast[wv] = x[OP_PER - 1];
                __syncthreads();
                xprev = __shfl_up(x[OP_PER - 1], 1, 64);
                if (lane == 0) xprev = wv > 0 ? s_wlast[wv - 1] : cx[s];
                if (base == 0) xprev = x[0];                 // prev_x starts at y[0]   SillySampler.py:165
                cx[s] = s_wlast[3];
            }
            aff loc[OP_PER];
            aff acc = {1.0, 0.0};
#pragma unroll
            for (int k = 0; k < OP_PER; ++k) {
                aff e;
                if (base + k < n) {
                    if (hp) {
                        const float xp = k == 0 ? xprev : x[k - 1];
                        e = {al[k], al[k] * ((double)x[k] - (double)xp)};
                    } else {
                        e = {1.0 - al[k], al[k] * (double)x[k]};
                    }
                } else {
                    e = {1.0, 0.0};
                }
                loc[k] = e;
                acc = aff_after(e, acc);
            }
            // inclusive scan of the per-thread composites across the wave
            aff inc = acc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                aff o = {__shfl_up(inc.A, off, 64), __shfl_up(inc.B, off, 64)};
                if (lane >= off) inc = aff_after(inc, o);
            }
            aff exc = {__shfl_up(inc.A, 1, 64), __shfl_up(inc.B, 1, 64)};
            if (lane == 0) exc = {1.0, 0.0};
            if (lane == 63) { s_tot[wv][0] = inc.A; s_tot[wv][1] = inc.B; }
            __syncthreads();
            aff pre = {1.0, 0.0}, tot = {1.0, 0.0};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                aff tw = {s_tot[w][0], s_tot[w][1]};
                if (w < wv) pre = aff_after(tw, pre);
                tot = aff_after(tw, tot);
            }
            const aff in = aff_after(exc, pre);
            double y = in.A * cy[s] + in.B;
            cy[s] = tot.A * cy[s] + tot.B;
#pragma unroll
            for (int k = 0; k < OP_PER; ++k) {
                y = loc[k].A * y + loc[k].B;
                x[k] = (float)y;                              // each section's output is an fp32 array in the reference
            }
            __syncthreads();                                  // s_tot / s_wlast are rewritten by the next section
            }
        }
#pragma unroll
        for (int k = 0; k < OP_PER; ++k)
            if (base + k < n) y0[base + k] = x[k];
    }
}

int launch_onepole(goofer_ctx *ctx, const float *src, float *dst, const float *f0, const goofer_onepole_job *jobs, int n_jobs,
                   hipStream_t st)
{
    if (n_jobs <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_onepole_cascade, dim3(n_jobs), dim3(256), 0, st, src, dst, f0, jobs, (double)ctx->plan.sr);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// elementwise stages; each thread owns one sample and exits unless its note carries the flag
#define POST_PROLOGUE()                                                                       \
    __shared__ int s_pair[2];                                                                 \
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;                                      \
    int lo_, hi_;                                                                             \
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo_, hi_);                       \
    const int64_t g = g0 + threadIdx.x;                                                       \
    if (g >= total) return;                                                                   \
    int note = lo_;                                                                           \
    while (sample_off[note + 1] <= g) ++note;                                                 \
    const goofer_post_note pn = notes[note];                                                  \
    const int64_t i = g - sample_off[note];                                                   \
    (void)i

__device__ __forceinline__ float fry_mask_at(const goofer_post_note &pn, int64_t i)
{
    const int a = pn.fry_a, b = pn.fry_b, fade = pn.fry_fade;
    if (i < a || i >= b) return 0.f;
    float v = 1.0f;
    if (fade > 0) {
        const int a1 = b < a + fade ? b : a + fade;
        if (i < a1) {
            const int m = a1 - a, k = (int)(i - a);
            const double w = m > 1 ? (k == m - 1 ? 1.0 : (double)k * (1.0 / (double)(m - 1))) : 0.0;   // np.linspace(0, 1, m)
            v = (float)((double)v * w);
        }
        const int b0 = a > b - fade ? a : b - fade;
        if (i >= b0) {
            const int m = b - b0, k = (int)(i - b0);
            const double w = m > 1 ? (k == m - 1 ? 0.0 : (double)k * (-1.0 / (double)(m - 1)) + 1.0) : 1.0;   // np.linspace(1, 0, m)
            v = (float)((double)v * w);
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void k_post_layers(float *__restrict__ harm, const float *__restrict__ su, const float *__restrict__ sj,
                                                     const goofer_post_note *__restrict__ notes, const int64_t *__restrict__ sample_off,
                                                     int n_notes, int64_t total)
{
    POST_PROLOGUE();
    if (pn.su_off < 0 && pn.sj_off < 0) return;
    float h = harm[g];
    if (pn.su_off >= 0) h = h + su[pn.su_off + i] * pn.su_gain;                               // :1059
    if (pn.sj_off >= 0) h = (1.0f - pn.sj_mix) * h + pn.sj_mix * sj[pn.sj_off + i];          // :1081
    harm[g] = h;
}

__global__ __launch_bounds__(256) void k_post_fry(float *__restrict__ harm, float *__restrict__ bre, const float *__restrict__ harm_hp,
                                                  const float *__restrict__ bre_hp, const goofer_post_note *__restrict__ notes,
                                                  const int64_t *__restrict__ sample_off, int n_notes, int64_t total)
{
    POST_PROLOGUE();
    if (pn.fry_a >= pn.fry_b) return;
    const float m = fry_mask_at(pn, i);
    harm[g] = harm[g] * (1.0f - m) + harm_hp[g] * m;                                          // :1097-1098
    bre[g] = bre[g] * (1.0f - m) + bre_hp[g] * m;
}

__global__ __launch_bounds__(256) void k_post_sd(float *__restrict__ bre, const double *__restrict__ vmask_s,
                                                 const goofer_post_note *__restrict__ notes, const int64_t *__restrict__ sample_off,
                                                 int n_notes, int64_t total, double sr)
{
    POST_PROLOGUE();
    if (!(pn.sd_strength > 0.f)) return;
    const int64_t n = sample_off[note + 1] - sample_off[note];
    // gf.create_volume_jitter(vibrato=True): zero-phase 150 Hz sinusoid, 0.1 s fade-in, clip [0.5, 1.5]   GOOFER.py:638-660
    double z = sin(((2.0 * 3.141592653589793) * 150.0) * ((double)i / sr) + 0.0);
    const int fade = (int)(0.1 * sr);
    if (fade < n && i < fade) z *= fade > 1 ? (i == fade - 1 ? 1.0 : (double)i * (1.0 / (double)(fade - 1))) : 0.0;
    double env = 1.0 + z * ((double)pn.sd_strength / 200.0);
    env = fmin(fmax(env, 0.5), 1.5);
    float b = (float)((double)bre[g] * (1.0 + (env - 1.0) * vmask_s[g]));                     // :1110
    b = b * (float)(1.0 + ((double)pn.sd_strength / 100.0) * 10.0);                           // :1112
    bre[g] = b;
}

__global__ __launch_bounds__(256) void k_note_sumsq(const float *__restrict__ harm, const float *__restrict__ bre,
                                                    const goofer_post_note *__restrict__ notes, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total, double *__restrict__ sums)
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
        if (notes[note].tension != 0.f) {
            const float s = harm[g] + bre[g];
            v = (double)s * (double)s;
        }
    }
    if (lo == hi) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
            if (t != 0.0) atomicAdd(sums + lo, t);
        }
    } else if (g < total && v != 0.0) {
        atomicAdd(sums + note, v);
    }
}

__global__ __launch_bounds__(256) void k_post_tension(float *__restrict__ harm, float *__restrict__ bre, const float *__restrict__ harm_hp,
                                                      const goofer_post_note *__restrict__ notes, const int64_t *__restrict__ sample_off,
                                                      int n_notes, int64_t total)
{
    POST_PROLOGUE();
    if (!(pn.tension > 0.f)) return;
    const float t = pn.tension;
    harm[g] = harm[g] + harm_hp[g] * (float)(1.0 + (double)t * 20.0);                         // :1130-1131
    bre[g] = bre[g] * (float)(1.0 - (double)t);                                              // :1135
}

__global__ __launch_bounds__(256) void k_post_scale(float *__restrict__ harm, float *__restrict__ bre, const double *__restrict__ before,
                                                    const double *__restrict__ after, const goofer_post_note *__restrict__ notes,
                                                    const int64_t *__restrict__ sample_off, int n_notes, int64_t total)
{
    POST_PROLOGUE();
    if (pn.tension == 0.f) return;
    const double n = (double)(sample_off[note + 1] - sample_off[note]);
    const double r0 = sqrt(before[note] / n + 1e-12), r1 = sqrt(after[note] / n + 1e-12);   // gf.rms   GOOFER.py:171
    if (!(r1 > 0.0)) return;
    const float gain = (float)(r0 / r1);
    harm[g] = harm[g] * gain;                                                                 // :1138-1140
    bre[g] = bre[g] * gain;
}

__global__ __launch_bounds__(256) void k_post_mix(const float *__restrict__ harm, const float *__restrict__ uv, const float *__restrict__ bre,
                                                  const float *__restrict__ sa_uv, const float *__restrict__ sa_bre,
                                                  const double *__restrict__ dyn, const goofer_post_note *__restrict__ notes,
                                                  const unsigned char *__restrict__ note_on, const goofer_note_params *__restrict__ params,
                                                  const int64_t *__restrict__ sample_off, int n_notes, int64_t total, float *__restrict__ mix)
{
    POST_PROLOGUE();
    if (!note_on[note]) return;
    const goofer_note_params p = params[note];
    float o = ((harm[g] * p.mix_harm + bre[g] * p.mix_breath) + uv[g] * p.mix_unvoiced) * p.volume;    // :1142-1151
    if (pn.sa_off >= 0) {
        const float ap = sa_uv[pn.sa_off + i] + sa_bre[pn.sa_off + i];
        o = o * (1.0f - pn.sa_mix) + (ap * p.volume) * pn.sa_mix;                             // :1172
    }
    if (pn.pitch_dyn != 0.f && dyn) o = (float)((double)o * dyn[g]);                          // :1182
    mix[g] = o;
}

// ---------------------------------------------------------------------------------------------
// pd: reference level = np.percentile(|bend_s|, 95) (linear interpolation between order statistics)
__global__ __launch_bounds__(256) void k_percentile95(const double *__restrict__ x, const int64_t *__restrict__ sample_off,
                                                      const unsigned char *__restrict__ note_on, double *__restrict__ ref)
{
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned int s_rank;
    __shared__ unsigned long long s_min;
    __shared__ unsigned int s_cnt;
    const int note = blockIdx.x;
    if (!note_on[note]) return;
    const int64_t base = sample_off[note];
    const int n = (int)(sample_off[note + 1] - base);
    if (n <= 0) return;
    const double vidx = 0.95 * (double)(n - 1);
    const int k_lo = (int)floor(vidx);
    const double frac = vidx - (double)k_lo;
    const double *__restrict__ v = x + base;
    auto key = [&](int i) { return (unsigned long long)__double_as_longlong(fabs(v[i])); };
    if (threadIdx.x == 0) { s_prefix = 0ull; s_rank = (unsigned)k_lo; }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << ((pass + 1) * 8));
        for (int i = threadIdx.x; i < n; i += 256) {
            const unsigned long long kk = key(i);
            if ((kk & hi_mask) == prefix) atomicAdd(&hist[(kk >> (pass * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned r = s_rank, b = 0;
            for (; b < 256; ++b) {
                if (r < hist[b]) break;
                r -= hist[b];
            }
            s_rank = r;
            s_prefix = prefix | ((unsigned long long)b << (pass * 8));
        }
        __syncthreads();
    }
    const unsigned long long klo = s_prefix;
    if (threadIdx.x == 0) { s_min = ~0ull; s_cnt = 0; }
    __syncthreads();
    unsigned cnt = 0;
    unsigned long long mn = ~0ull;
    for (int i = threadIdx.x; i < n; i += 256) {
        const unsigned long long kk = key(i);
        if (kk <= klo) ++cnt;
        else if (kk < mn) mn = kk;
    }
    atomicAdd(&s_cnt, cnt);
    atomicMin(&s_min, mn);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = __longlong_as_double((long long)klo);
        double b = a;
        if (k_lo + 1 <= n - 1 && s_cnt < (unsigned)(k_lo + 2)) b = __longlong_as_double((long long)s_min);
        const double diff = b - a;                            // numpy's _lerp
        double r = a + diff * frac;
        if (frac >= 0.5) r = b - diff * (1.0 - frac);
        if (diff == 0.0) r = a;
        ref[note] = r;
    }
}

__global__ __launch_bounds__(256) void k_dyn_gain(const double *__restrict__ bend_s, const double *__restrict__ vmask_s,
                                                  const double *__restrict__ ref, const goofer_post_note *__restrict__ notes,
                                                  const int64_t *__restrict__ sample_off, int n_notes, int64_t total,
                                                  double *__restrict__ dyn)
{
    POST_PROLOGUE();
    if (pn.pitch_dyn == 0.f) return;
    const double r = ref[note] + 1e-8;
    double v = bend_s[g] / r;
    v = fmin(fmax(v, -1.0), 1.0);
    const double pd = (double)pn.pitch_dyn;
    const double db = (12.0 * fabs(pd)) * (pd > 0.0 ? v : -v);
    float gq = (float)pow(10.0, db / 20.0);
    gq = fminf(fmaxf(gq, 1e-3f), 1e3f);
    dyn[g] = 1.0 + ((double)gq - 1.0) * vmask_s[g];
}

// ---------------------------------------------------------------------------------------------
#define ELEMENTWISE_GRID dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st

int launch_post_layers(goofer_ctx *ctx, float *harm, const float *su, const float *sj, const goofer_post_note *notes,
                       const int64_t *sample_off, int n_notes, int64_t total, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_layers, ELEMENTWISE_GRID, harm, su, sj, notes, sample_off, n_notes, total);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_post_fry(goofer_ctx *ctx, float *harm, float *bre, const float *harm_hp, const float *bre_hp, const goofer_post_note *notes,
                    const int64_t *sample_off, int n_notes, int64_t total, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_fry, ELEMENTWISE_GRID, harm, bre, harm_hp, bre_hp, notes, sample_off, n_notes, total);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_post_sd(goofer_ctx *ctx, float *bre, const double *vmask_s, const goofer_post_note *notes, const int64_t *sample_off,
                   int n_notes, int64_t total, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_sd, ELEMENTWISE_GRID, bre, vmask_s, notes, sample_off, n_notes, total, (double)ctx->plan.sr);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_note_sumsq(goofer_ctx *ctx, const float *harm, const float *bre, const goofer_post_note *notes, const int64_t *sample_off,
                      int n_notes, int64_t total, double *sums, hipStream_t st)
{
    hipLaunchKernelGGL(k_note_sumsq, ELEMENTWISE_GRID, harm, bre, notes, sample_off, n_notes, total, sums);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_post_tension(goofer_ctx *ctx, float *harm, float *bre, const float *harm_hp, const goofer_post_note *notes,
                        const int64_t *sample_off, int n_notes, int64_t total, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_tension, ELEMENTWISE_GRID, harm, bre, harm_hp, notes, sample_off, n_notes, total);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_post_scale(goofer_ctx *ctx, float *harm, float *bre, const double *before, const double *after,
                      const goofer_post_note *notes, const int64_t *sample_off, int n_notes, int64_t total, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_scale, ELEMENTWISE_GRID, harm, bre, before, after, notes, sample_off, n_notes, total);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_post_mix(goofer_ctx *ctx, const float *harm, const float *uv, const float *bre, const float *sa_uv, const float *sa_bre,
                    const double *dyn, const goofer_post_note *notes, const unsigned char *note_on, const goofer_note_params *params,
                    const int64_t *sample_off, int n_notes, int64_t total, float *mix, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_mix, ELEMENTWISE_GRID, harm, uv, bre, sa_uv, sa_bre, dyn, notes, note_on, params, sample_off, n_notes,
                       total, mix);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_dyn_gain(goofer_ctx *ctx, const double *bend_s, const double *vmask_s, const unsigned char *note_on, double *ref,
                    const goofer_post_note *notes, const int64_t *sample_off, int n_notes, int64_t total, double *dyn, hipStream_t st)
{
    hipLaunchKernelGGL(k_percentile95, dim3(n_notes), dim3(256), 0, st, bend_s, sample_off, note_on, ref);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_dyn_gain, ELEMENTWISE_GRID, bend_s, vmask_s, ref, notes, sample_off, n_notes, total, dyn);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// apply_vocal_roughness (GOOFER.py:901-940), the `roughness_on` layer of gf.synthesize: amplitude modulation of the
// harmonic stem at f0/k (k = 2, 3, 4 ...) with noisy modulation rates, the difference high-passed and faded in by the
// slewed voicing mask.  Two strictly sequential recurrences per note — the fp64 running sum of each modulator's
// frequency (np.cumsum) and the one-pole high-pass with its python-float state — so one lane walks one note; notes run
// side by side.  A rarely used switch (no caller of the resampler sets it): written for exactness, not speed.
//   nz     [n_k][total] fp64 smoothed noises (make_smooth_noise: legacy RNG re-seeded 1337 + idx, fp32 draw, fp64 Gaussian)
//   aslew  [total]      fp32 gaussian_filter1d(alpha * mask, sigma = alpha_slew_ms sr / 6000)
struct rough_cfg {
    int n_k;
    double k[8];
    float h[8];          // hk * float32 array: the weight is applied in fp32
    double noise_amp, hp_a, sr;
};

__global__ __launch_bounds__(64) void k_vocal_roughness(const float *__restrict__ y, const float *__restrict__ f0,
                                                        const float *__restrict__ mask, const double *__restrict__ nz,
                                                        const float *__restrict__ aslew, const int64_t *__restrict__ sample_off,
                                                        int n_notes, int64_t total, rough_cfg cfg, float *__restrict__ out)
{
    const int note = blockIdx.x * blockDim.x + threadIdx.x;
    if (note >= n_notes) return;
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
    double c[8];
    for (int q = 0; q < 8; ++q) c[q] = 0.0;
    double px = 0.0, py = 0.0;
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t g = base + i;
        const float f = f0[g], vm = mask[g], yv = y[g];
        float mod = 0.f;
        for (int q = 0; q < cfg.n_k; ++q) {
            const float fk = f / (float)cfg.k[q];                                   // f0 / float(k): fp32 array / python float
            double fm = (double)fk * (1.0 + cfg.noise_amp * nz[(int64_t)q * total + g]);
            fm = (fm > 0.0 ? fm : 0.0) * (double)vm;                                // np.maximum(f_mod, 0.0) * vmask
            c[q] += fm;                                                             // np.cumsum, fp64, in order
            const double ph = (two_pi * c[q]) / cfg.sr;
            mod += cfg.h[q] * (float)cos(ph);                                       // mod_sum += hk * cos(phase).astype(f32)
        }
        const float ym = yv * (1.0f + mod);
        const float ys = ym - yv;
        const double xn = (double)ys;
        const double yn = cfg.hp_a * (py + xn - px);                                // one_pole_highpass: python-float state
        px = xn;
        py = yn;
        out[g] = yv + aslew[g] * (float)yn;
    }
}

int launch_vocal_roughness(goofer_ctx *ctx, const float *y, const float *f0, const float *mask, const double *nz, int n_k,
                           const double *k_list, const double *h_list, double noise_amp, double hp_fc, const float *aslew,
                           const int64_t *sample_off, int n_notes, int64_t total, float *out, hipStream_t st)
{
    if (n_notes <= 0 || total <= 0) return GOOFER_OK;
    if (n_k < 0 || n_k > 8) return goofer_fail(ctx, GOOFER_EINVAL, "%d roughness modulators (0..8 supported)", n_k);
    rough_cfg cfg;
    cfg.n_k = n_k;
    for (int q = 0; q < 8; ++q) {
        cfg.k[q] = q < n_k ? k_list[q] : 1.0;
        cfg.h[q] = q < n_k ? (float)h_list[q] : 0.f;
    }
    cfg.noise_amp = noise_amp;
    cfg.sr = (double)ctx->plan.sr;
    if (hp_fc > 0.0) {
        const double rc = 1.0 / (2.0 * 3.14159265358979323846 * hp_fc);
        cfg.hp_a = rc / (rc + 1.0 / cfg.sr);
    } else {
        cfg.hp_a = 0.0;                                        // fc <= 0: one_pole_highpass returns zeros
    }
    hipLaunchKernelGGL(k_vocal_roughness, dim3((n_notes + 63) / 64), dim3(64), 0, st, y, f0, mask, nz, aslew, sample_off, n_notes,
                       total, cfg, out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// The wav the reference writes is PCM16 (soundfile's default WAV subtype, SillySampler.py:1184-1185): clip to
// [-1, 1 - 2^-15], times 32768, round half to even.  Done here the finished audio crosses PCIe at two bytes per sample — the
// download of the fp32 mix is what bounds a long job (199 MB per 1024 notes, 3.6 ms against a 2.3 ms device step).
__global__ __launch_bounds__(256) void k_pcm16(const float *__restrict__ x, int64_t n, int16_t *__restrict__ out)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i0 >= n) return;
    auto q = [](float v) {
        double d = (double)v;
        d = d < -1.0 ? -1.0 : (d > 1.0 - 1.0 / 32768.0 ? 1.0 - 1.0 / 32768.0 : d);
        return (int16_t)(int)rint(d * 32768.0);
    };
    if (i0 + 8 <= n && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)out) & 15) == 0)) {
        const float4 a = *reinterpret_cast<const float4 *>(x + i0), b = *reinterpret_cast<const float4 *>(x + i0 + 4);
        union { int16_t s[8]; uint4 v; } o;
        o.s[0] = q(a.x); o.s[1] = q(a.y); o.s[2] = q(a.z); o.s[3] = q(a.w);
        o.s[4] = q(b.x); o.s[5] = q(b.y); o.s[6] = q(b.z); o.s[7] = q(b.w);
        *reinterpret_cast<uint4 *>(out + i0) = o.v;
    } else {
        for (int64_t i = i0; i < n && i < i0 + 8; ++i) out[i] = q(x[i]);
    }
}

extern "C" int goofer_pcm16(goofer_ctx *ctx, const float *x, int64_t n, int16_t *out, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (n <= 0) return GOOFER_OK;
    if (!x || !out) return goofer_fail(ctx, GOOFER_EINVAL, "null pointer");
    hipLaunchKernelGGL(k_pcm16, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
