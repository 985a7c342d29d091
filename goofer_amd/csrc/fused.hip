// Fused per-frame kernels of goofer_synth_batch (gfx950).  One wave owns one frame from the first HBM
// load to the windowed time frame; spectra and envelopes never leave LDS/registers.
//
//   k_harm_frames   stft(pulse) -> HP mask -> per-note max -> * warped env * boost -> brightness + blur
//                   -> irfft * window                         GOOFER.py:1099-1146 (+ :840-875, 618-627)
//   k_noise_frames  sigma-1.75 blurred env -> random-phase spectra (uv, breath) -> brightness + blur
//                   -> 2x irfft * window                       GOOFER.py:993, 1148-1176
//
// HBM traffic per frame: harm 1-4 KB pulse + 2 KB env in, 4 KB frame out; noise 2 KB env (+2 KB phases when
// injected) in, 8 KB frames out — versus ~45 KB for the same work as separate kernels.
#include "binops_core.h"
#include "fft_core.h"

constexpr int FZ_WAVES = 4;
constexpr int FZ_FRAMES = 16;      // contiguous frames per workgroup (4 per wave)

template <int M> struct fz_cfg {
    static constexpr int B = M + 1;                    // bins
    static constexpr int XS = (B + 3) & ~1;            // float2 slots per spectrum buffer (even, >= B)
    static constexpr int PER = (B + WAVE - 1) / WAVE;  // bins per lane
    static constexpr size_t wave_f2 = XS + fft_cfg<M>::BUF;        // one spectrum + the FFT exchange buffer
    static constexpr size_t lds_bytes = sizeof(float2) * (M + M / 2 + 1 + FZ_WAVES * wave_f2) + 16;
};

struct fz_frame {
    int note;
    int64_t t, base, n, src;
    float f0f;
    bool voiced;
};

__device__ __forceinline__ fz_frame frame_info(int64_t f, const int *frame_note, const int64_t *frame_off, const int64_t *sample_off,
                                               const int64_t *row_src, const float *f0, const float *mask,
                                               const goofer_note_params *params, int hop)
{
    fz_frame q;
    q.note = frame_note[f];
    q.t = f - frame_off[q.note];
    q.base = sample_off[q.note];
    q.n = sample_off[q.note + 1] - q.base;
    q.src = row_src[f];
    const int64_t pk = pick_index(q.t, q.n, hop);
    q.f0f = q.n > 0 ? f0[q.base + pk] : 0.f;                   // f0 already carries pitch_shift
    q.voiced = params[q.note].apply_brightness && q.n > 0 && mask[q.base + pk] > 0.f;
    return q;
}

// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256, 3) void k_harm_frames(const float *__restrict__ pulse, const float *__restrict__ env, int ld,
                                                     const double *__restrict__ formants, const float *__restrict__ f0,
                                                     const float *__restrict__ mask, const int64_t *__restrict__ sample_off,
                                                     const int64_t *__restrict__ frame_off, const int *__restrict__ frame_note,
                                                     const int64_t *__restrict__ row_src,
                                                     const goofer_note_params *__restrict__ params, int64_t total_frames,
                                                     float *__restrict__ frames_out, float *__restrict__ note_mag, int hop, double nyq,
                                                     const float2 *__restrict__ g_tw, const float2 *__restrict__ g_twh,
                                                     const float *__restrict__ g_win, const float *__restrict__ freqs,
                                                     const float *__restrict__ boost, const float *__restrict__ bright,
                                                     const double *__restrict__ taps5)
{
    using C = fz_cfg<M>;
    constexpr int R = fft_cfg<M>::R, B = C::B;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_seg[4][WARP_SEG_DOUBLES];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *wbase = twh + (M / 2 + 1);
    const float *win = g_win;                                  // 4-8 KB table: lives in L1, keeps LDS for occupancy
    for (int i = threadIdx.x; i < M; i += blockDim.x) tw[i] = g_tw[i];
    for (int i = threadIdx.x; i < M / 2 + 1; i += blockDim.x) twh[i] = g_twh[i];
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *X1 = wbase + wave * C::wave_f2, *fbuf = X1 + C::XS;
    double t5[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) t5[j] = taps5[j];

    const int64_t f_begin = (int64_t)blockIdx.x * FZ_FRAMES;
    for (int i = wave; i < FZ_FRAMES; i += FZ_WAVES) {
        const int64_t f = f_begin + i;
        if (f >= total_frames) break;                          // wave-uniform
        const fz_frame q = frame_info(f, frame_note, frame_off, sample_off, row_src, f0, mask, params, hop);
        const goofer_note_params &p = params[q.note];

        // 1. STFT of the pulse frame, spectrum kept in LDS
        float2 v[R];
        rfft_load<M>(v, pulse + q.base, q.t * hop - M, q.n, win, lane);
        wave_fft<M>(v, fbuf, tw, lane);
        rfft_split<M>(fbuf, twh, lane, [&](int k, float2 val) { X1[k] = val; });

        // 2. harmonic envelope: formant-anchored + uniform warp of the source row (rows alias the FFT buffer)
        wave_lds_sync();
        float *ra = reinterpret_cast<float *>(fbuf), *rb = ra + ((B + 1) & ~1);
        const float *er = env + q.src * (int64_t)ld;
        for (int b = lane; b < B; b += WAVE) ra[b] = er[b];
        wave_lds_sync();
        double fs[4];
        bool warp = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fs[k] = p.f_shift[k];
            warp |= fs[k] != 1.0;
        }
        const float *cur = warp_row(ra, rb, B, nyq, formants ? formants + q.src * 4 : nullptr, fs, warp, (double)p.formant_shift, lane, s_seg[wave]);
        float eh[C::PER];
#pragma unroll
        for (int r = 0; r < C::PER; ++r) {
            int b = lane + WAVE * r;
            eh[r] = b < B ? cur[b] : 0.f;
        }
        wave_lds_sync();                                       // rows dead, FFT buffer free

        // 3. shaping in place in X1; per-note max(|S| + 1e-8) (applied after the OLA: it commutes with the linear chain)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < C::PER; ++r) {
            int k = lane + WAVE * r;
            if (k < B) {
                float2 s = X1[k];
                if (p.cut_below_f0) {
                    float h = hp_mask(freqs[k], q.f0f);
                    s.x *= h; s.y *= h;
                }
                mx = fmaxf(mx, cabs_fast(s) + 1e-8f);
                s.x = (s.x * eh[r]) * boost[k];
                s.y = (s.y * eh[r]) * boost[k];
                if (q.voiced) { s.x *= bright[k]; s.y *= bright[k]; }
                X1[k] = s;
            }
        }
        mx = wave_max(mx);
        if (lane == 0) atomic_max_pos(note_mag + q.note, mx);
        wave_lds_sync();
        const float2 *spec = X1;
        if (q.voiced) {                                        // blurred copy goes to the (free) FFT buffer
            for (int k = lane; k < B; k += WAVE) fbuf[k] = blur5(X1, k, B, t5);
            wave_lds_sync();
            spec = fbuf;
        }

        // 4. inverse transform, window, out
        irfft_load<M>(v, [&](int k) { return spec[k]; }, twh, lane);
        wave_lds_sync();                                       // all reads of spec done before the FFT reuses fbuf
        wave_fft<M>(v, fbuf, tw, lane);
        irfft_store<M>(fbuf, win, frames_out + f * (int64_t)(2 * M), lane);
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256, 3) void k_noise_frames(const float *__restrict__ env, int ld, const float *__restrict__ phi,
                                                      const float *__restrict__ f0, const float *__restrict__ mask,
                                                      const int64_t *__restrict__ sample_off, const int64_t *__restrict__ frame_off,
                                                      const int *__restrict__ frame_note, const int64_t *__restrict__ row_src,
                                                      const goofer_note_params *__restrict__ params, uint64_t seed,
                                                      int64_t total_frames, float *__restrict__ frames_uv,
                                                      float *__restrict__ frames_br, int hop, const float2 *__restrict__ g_tw,
                                                      const float2 *__restrict__ g_twh, const float *__restrict__ g_win,
                                                      const float *__restrict__ freqs, const float *__restrict__ bright,
                                                      const double *__restrict__ taps5, const double *__restrict__ taps175)
{
    using C = fz_cfg<M>;
    constexpr int R = fft_cfg<M>::R, B = C::B;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *wbase = twh + (M / 2 + 1);
    const float *win = g_win;
    for (int i = threadIdx.x; i < M; i += blockDim.x) tw[i] = g_tw[i];
    for (int i = threadIdx.x; i < M / 2 + 1; i += blockDim.x) twh[i] = g_twh[i];
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *X1 = wbase + wave * C::wave_f2, *fbuf = X1 + C::XS;
    double t5[5], t15[15];
#pragma unroll
    for (int j = 0; j < 5; ++j) t5[j] = taps5[j];
#pragma unroll
    for (int j = 0; j < 15; ++j) t15[j] = taps175[j];

    const int64_t f_begin = (int64_t)blockIdx.x * FZ_FRAMES;
    for (int i = wave; i < FZ_FRAMES; i += FZ_WAVES) {
        const int64_t f = f_begin + i;
        if (f >= total_frames) break;
        const fz_frame q = frame_info(f, frame_note, frame_off, sample_off, row_src, f0, mask, params, hop);
        const goofer_note_params &p = params[q.note];

        // 1. noise envelope = sigma-1.75 blur of the UN-warped source row (GOOFER.py:993); row aliases fbuf
        float *ra = reinterpret_cast<float *>(fbuf);
        const float *er = env + q.src * (int64_t)ld;
        for (int b = lane; b < B; b += WAVE) ra[b] = er[b];
        wave_lds_sync();
        float en[C::PER];
#pragma unroll
        for (int r = 0; r < C::PER; ++r) {
            int b = lane + WAVE * r;
            float acc = 0.f;                                   // fp32 FMAs in tap order, like k_noise_spectra
            if (b < B) {
                if (b >= 7 && b + 7 < B) {
                    acc = (float)t15[0] * ra[b - 7];
#pragma unroll
                    for (int j = 1; j < 15; ++j) acc = fmaf((float)t15[j], ra[b + j - 7], acc);
                } else {
                    acc = (float)t15[0] * ra[reflect_index(b - 7, B)];
#pragma unroll
                    for (int j = 1; j < 15; ++j) acc = fmaf((float)t15[j], ra[reflect_index(b + j - 7, B)], acc);
                }
            }
            en[r] = acc;
        }
        wave_lds_sync();                                       // row dead, fbuf free

        // 2. U * env_n kept in registers; breath spectrum S_br = U env_n HP (* brightness) -> X1
        const uint64_t key = seed ^ ((uint64_t)p.seed[0] | ((uint64_t)p.seed[1] << 32));
        float2 uv2[C::PER];
#pragma unroll
        for (int r = 0; r < C::PER; ++r) {
            int k = lane + WAVE * r;
            uv2[r] = make_float2(0.f, 0.f);
            if (k < B) {
                float c, s;
                if (phi) {
                    const float ph = phi[f * (int64_t)ld + k];
                    c = cosf(ph);
                    s = sinf(ph);
                } else {
                    const uint32_t u = philox_u32(key, (uint64_t)q.t, (uint32_t)k);
                    const float rev = (float)(u >> 8) * (1.0f / 16777216.0f);
                    c = __builtin_amdgcn_cosf(rev);
                    s = __builtin_amdgcn_sinf(rev);
                }
                uv2[r] = make_float2(c * en[r], s * en[r]);
                float h = hp_mask(freqs[k], q.f0f);
                float2 bb = make_float2(uv2[r].x * h, uv2[r].y * h);
                if (q.voiced) { bb.x *= bright[k]; bb.y *= bright[k]; }
                X1[k] = bb;
            }
        }
        wave_lds_sync();

        // 3. breath stem
        float2 v[R];
        const float2 *spec = X1;
        if (q.voiced) {
            for (int k = lane; k < B; k += WAVE) fbuf[k] = blur5(X1, k, B, t5);
            wave_lds_sync();
            spec = fbuf;
        }
        irfft_load<M>(v, [&](int k) { return spec[k]; }, twh, lane);
        wave_lds_sync();
        wave_fft<M>(v, fbuf, tw, lane);
        irfft_store<M>(fbuf, win, frames_br + f * (int64_t)(2 * M), lane);
        wave_lds_sync();

        // 4. unvoiced stem: S_uv from registers -> X1
#pragma unroll
        for (int r = 0; r < C::PER; ++r) {
            int k = lane + WAVE * r;
            if (k < B) X1[k] = uv2[r];
        }
        wave_lds_sync();
        irfft_load<M>(v, [&](int k) { return X1[k]; }, twh, lane);
        wave_fft<M>(v, fbuf, tw, lane);
        irfft_store<M>(fbuf, win, frames_uv + f * (int64_t)(2 * M), lane);
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
template <int M>
static int harm_impl(goofer_ctx *ctx, const float *pulse, const goofer_batch *b, const float *f0s, const int *frame_note,
                     const int64_t *row_src, float *frames, float *note_mag, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    if (int arc = kernel_allow_max_lds(ctx, (const void *)k_harm_frames<M>, 159 * 1024)   /* + 576 B static */) return arc;
    unsigned blocks = (unsigned)((b->total_frames + FZ_FRAMES - 1) / FZ_FRAMES);
    hipLaunchKernelGGL(k_harm_frames<M>, dim3(blocks), dim3(256), fz_cfg<M>::lds_bytes, st, pulse, b->env, b->ld, b->formants, f0s,
                       b->mask, b->sample_off, b->frame_off, frame_note, row_src, b->params, b->total_frames, frames, note_mag, p.hop,
                       (double)p.sr / 2.0, p.tw_full, p.tw_half, p.window, p.freqs, p.boost, p.bright_h, p.blur5);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

template <int M>
static int noise_impl(goofer_ctx *ctx, const goofer_batch *b, const float *f0s, const int *frame_note, const int64_t *row_src,
                      float *frames_uv, float *frames_br, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    if (int arc = kernel_allow_max_lds(ctx, (const void *)k_noise_frames<M>)) return arc;
    unsigned blocks = (unsigned)((b->total_frames + FZ_FRAMES - 1) / FZ_FRAMES);
    hipLaunchKernelGGL(k_noise_frames<M>, dim3(blocks), dim3(256), fz_cfg<M>::lds_bytes, st, b->env, b->ld, b->phi, f0s, b->mask,
                       b->sample_off, b->frame_off, frame_note, row_src, b->params, b->seed, b->total_frames, frames_uv, frames_br,
                       p.hop, p.tw_full, p.tw_half, p.window, p.freqs, p.bright_b, p.blur5, p.blur175);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_harm_frames(goofer_ctx *ctx, const float *pulse, const goofer_batch *b, const float *f0s, const int *frame_note,
                       const int64_t *row_src, float *frames, float *note_mag, hipStream_t st)
{
    if (b->total_frames <= 0) return GOOFER_OK;
    switch (ctx->plan.n_fft) {
    case 512: return harm_impl<256>(ctx, pulse, b, f0s, frame_note, row_src, frames, note_mag, st);
    case 1024: return harm_impl<512>(ctx, pulse, b, f0s, frame_note, row_src, frames, note_mag, st);
    case 2048: return harm_impl<1024>(ctx, pulse, b, f0s, frame_note, row_src, frames, note_mag, st);
    }
    return goofer_fail(ctx, GOOFER_EINVAL, "unsupported n_fft %d", ctx->plan.n_fft);
}

int launch_noise_frames(goofer_ctx *ctx, const goofer_batch *b, const float *f0s, const int *frame_note, const int64_t *row_src,
                        float *frames_uv, float *frames_br, hipStream_t st)
{
    if (b->total_frames <= 0) return GOOFER_OK;
    switch (ctx->plan.n_fft) {
    case 512: return noise_impl<256>(ctx, b, f0s, frame_note, row_src, frames_uv, frames_br, st);
    case 1024: return noise_impl<512>(ctx, b, f0s, frame_note, row_src, frames_uv, frames_br, st);
    case 2048: return noise_impl<1024>(ctx, b, f0s, frame_note, row_src, frames_uv, frames_br, st);
    }
    return goofer_fail(ctx, GOOFER_EINVAL, "unsupported n_fft %d", ctx->plan.n_fft);
}
