// Bin-axis kernels on the [frames x bins] matrix for gfx950: one 64-lane wave owns one frame row,
// the row is staged in LDS, neighbours along the bin axis come from LDS, reductions are wavefront
// shuffles.  All HBM traffic is row-contiguous (coalesced); no MFMA (2-tap lerps and <=57-tap FIRs).
//
//   k_gauss_bins     gf.gaussian_filter1d(axis=0)                      GOOFER.py:241-261
//   k_warp_bins      gf.warp_env_by_formants + gf.shift_formants       GOOFER.py:840-875, 618-627
//   k_knot_decode    gf.decode_env_from_knots                          GOOFER.py:149-168
//   k_harm_shape     high-pass mask, env*boost, brightness + 5-tap blur GOOFER.py:1102-1144
//   k_noise_spectra  random-phase noise stems                          GOOFER.py:1148-1173
#include <hip/hip_fp16.h>

#include "binops_core.h"

constexpr int ROWS_PER_BLOCK = 4;   // one wave per row, 256-thread workgroups

// ---------------------------------------------------------------------------------------------
// Gaussian FIR along bins; taps fp64 [2r+1]; numpy 'reflect' padding; fp64 accumulate, fp32 store.
__global__ __launch_bounds__(256) void k_gauss_bins(const float *__restrict__ in, float *__restrict__ out, int64_t rows,
                                                    int n_bins, int ld, const double *__restrict__ taps, int radius,
                                                    const int64_t *__restrict__ row_src)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_taps = reinterpret_cast<double *>(smem);
    float *s_rows = reinterpret_cast<float *>(s_taps + (2 * radius + 1));
    for (int i = threadIdx.x; i < 2 * radius + 1; i += blockDim.x) s_taps[i] = taps[i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    float *r = s_rows + wave * n_bins;
    if (row < rows) {
        const int64_t src = row_src ? row_src[row] : row;
        for (int b = lane; b < n_bins; b += WAVE) r[b] = in[src * ld + b];
    }
    __syncthreads();
    if (row >= rows) return;
    for (int b = lane; b < n_bins; b += WAVE) {
        double acc = 0.0;
        if (b >= radius && b + radius < n_bins) {
            for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)r[b + j - radius];
        } else {
            for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)r[reflect_index(b + j - radius, n_bins)];
        }
        out[row * ld + b] = (float)acc;
    }
}

int launch_gauss_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld, const double *d_taps,
                      int radius, const int64_t *row_src, hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    size_t lds = sizeof(double) * (2 * radius + 1) + sizeof(float) * ROWS_PER_BLOCK * n_bins;
    if (lds > 160 * 1024) return goofer_fail(ctx, GOOFER_EINVAL, "Gaussian radius %d with %d bins needs more than 160 KiB of LDS", radius, n_bins);
    if (lds > 64 * 1024)
        if (int arc = kernel_allow_max_lds(ctx, (const void *)k_gauss_bins)) return arc;
    hipLaunchKernelGGL(k_gauss_bins, dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), lds, st, in, out,
                       rows, n_bins, ld, d_taps, radius, row_src);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-row formant-anchored warp then uniform warp.  f_shift == nullptr skips the first stage,
// ratio == 1 the second; each stage rounds to fp32 (np.zeros_like(env) in the reference).
// rows may be addressed through row_map (source row per output row), or identity when nullptr.
__global__ __launch_bounds__(256) void k_warp_bins(const float *__restrict__ in, float *__restrict__ out, int64_t rows,
                                                   int n_bins, int ld, const double *__restrict__ formants,
                                                   const double *__restrict__ f_shift_global,
                                                   const goofer_note_params *__restrict__ params,
                                                   const int *__restrict__ row_note, const int64_t *__restrict__ row_src,
                                                   double ratio_global, const warp_grid grid)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_seg[ROWS_PER_BLOCK][WARP_SEG_DOUBLES];
    float *s_all = reinterpret_cast<float *>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (row >= rows) return;                       // no block-level barrier below
    float *ra = s_all + (2 * wave) * (n_bins + 1);       // (+ 1: the pad element of warp_row's fp32 lerp)
    float *rb = ra + n_bins + 1;
    const int64_t src = row_src ? row_src[row] : row;
    for (int b = lane; b < n_bins; b += WAVE) ra[b] = in[src * ld + b];
    wave_lds_sync();

    double fs[4];
    double ratio = ratio_global;
    bool warp = false;
    if (params) {
        const goofer_note_params &p = params[row_note[row]];
        for (int i = 0; i < 4; ++i) fs[i] = p.f_shift[i];
        ratio = (double)p.formant_shift;
        warp = (fs[0] != 1.0) || (fs[1] != 1.0) || (fs[2] != 1.0) || (fs[3] != 1.0);
    } else if (f_shift_global) {
        for (int i = 0; i < 4; ++i) fs[i] = f_shift_global[i];
        warp = true;   // the caller decides (gf.synthesize tests any(shift != 1))
    }
    float *cur = warp_row(ra, rb, n_bins, grid, formants ? formants + src * 4 : nullptr, fs, warp, ratio, lane, s_seg[wave]);
    for (int b = lane; b < n_bins; b += WAVE) out[row * ld + b] = cur[b];
}

int launch_warp_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld, const double *formants,
                     const double *d_f_shift, const goofer_note_params *params, const int *row_note,
                     const int64_t *row_src, double ratio, hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    size_t lds = sizeof(float) * 2 * ROWS_PER_BLOCK * (n_bins + 1);
    hipLaunchKernelGGL(k_warp_bins, dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), lds, st, in, out,
                       rows, n_bins, ld, formants, d_f_shift, params, row_note, row_src, ratio, make_warp_grid(ctx->plan.sr, n_bins));
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// env[row][b] = exp(w0[b]*knot[idx[b]] + w1[b]*knot[idx[b]+1]); the reference's dense W @ knots
// has exactly these two non-zeros per row.
__global__ __launch_bounds__(256) void k_knot_decode(const __half *__restrict__ knots, int K, int64_t rows,
                                                     const int *__restrict__ idx, const float *__restrict__ w0,
                                                     const float *__restrict__ w1, float *__restrict__ env, int n_bins, int ld)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_k = reinterpret_cast<float *>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (row >= rows) return;
    float *kv = s_k + wave * K;
    for (int k = lane; k < K; k += WAVE) kv[k] = __half2float(knots[row * K + k]);
    wave_lds_sync();
    for (int b = lane; b < n_bins; b += WAVE) {
        int i = idx[b];
        float v = w0[b] * kv[i] + w1[b] * kv[i + 1];
        env[row * ld + b] = expf(v);
    }
}

int launch_knot_decode(goofer_ctx *ctx, const uint16_t *knots, int K, int64_t rows, const int *idx, const float *w0,
                       const float *w1, float *env, int n_bins, int ld, hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_knot_decode, dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256),
                       sizeof(float) * ROWS_PER_BLOCK * K, st, reinterpret_cast<const __half *>(knots), K, rows, idx, w0, w1, env,
                       n_bins, ld);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// In place on S: optional high-pass, per-note max(|S| + 1e-8), then * env * boost, and on voiced
// frames * brightness followed by the 5-tap blur.  The 1/max normalisation commutes with the
// (linear) rest of the chain and is applied after the overlap-add.
// LDS (round 5): the three per-bin tables (bin frequency, boost, brightness) once per workgroup — as register preloads they were
// 3 x ITERS registers per lane and held the n_fft 2048 instantiation at 177 registers / two waves per SIMD — then per wave the
// complex row of the 5-tap blur and, only when the batch may warp (`warp_rows`; the caller's goofer_batch.no_warp hint turns
// them off), the warp's two fp32 rows: 45 KB instead of 66 KB per workgroup at n_fft 2048.
__host__ __device__ static inline int harm_shape_tab_floats(int n_bins) { return (n_bins + 4) & ~3; }
__host__ __device__ static inline size_t harm_shape_lds(int n_bins, bool warp_rows)
{
    const int rowf = (n_bins + 1) & ~1;
    return sizeof(float) * 3 * harm_shape_tab_floats(n_bins) + sizeof(float2) * ROWS_PER_BLOCK * (n_bins + (warp_rows ? rowf : 0));
}

template <int ITERS, bool NT>
__global__ __launch_bounds__(256, ITERS <= 9 ? 4 : 3) void k_harm_shape(float2 *__restrict__ S, int ldc, int64_t total_frames,
                                                    const int *__restrict__ frame_note, const int64_t *__restrict__ frame_off,
                                                    const int64_t *__restrict__ sample_off, const float *__restrict__ f0,
                                                    const float *__restrict__ mask, const float *__restrict__ env, int ld,
                                                    const goofer_note_params *__restrict__ params, float *__restrict__ note_mag,
                                                    const float *__restrict__ freqs, const float *__restrict__ boost,
                                                    const float *__restrict__ bright, const double *__restrict__ taps5,
                                                    int n_bins, int hop, const int64_t *__restrict__ row_src,
                                                    const double *__restrict__ formants, const warp_grid grid,
                                                    const float2 *__restrict__ picks, int warp_rows)
{
    // env is either the already-warped [frames x ld] matrix (row_src == nullptr) or the source rows, in which
    // case the formant-anchored + uniform warp (GOOFER.py:1004-1017) runs here on the LDS row.
    // Latency plan: the frame index is made wave-uniform so every per-note scalar is a scalar load (its own
    // counter), and all of the row's vector loads (spectrum, envelope) are issued before the first wait.
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_seg[ROWS_PER_BLOCK][WARP_SEG_DOUBLES];
    const int tabf = harm_shape_tab_floats(n_bins);
    float *t_fq = reinterpret_cast<float *>(smem), *t_bo = t_fq + tabf, *t_br = t_bo + tabf;
    for (int k = threadIdx.x; k < n_bins; k += blockDim.x) {
        t_fq[k] = freqs[k];
        t_bo[k] = boost[k];
        t_br[k] = bright[k];
    }
    __syncthreads();                                          // (every wave of the workgroup: the frame test comes after it)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t f = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (f >= total_frames) return;
    const int rowf = (n_bins + 1) & ~1;                           // floats per fp32 row (even)
    float2 *r = reinterpret_cast<float2 *>(t_br + tabf) + (size_t)wave * (n_bins + (warp_rows ? rowf : 0));
    float *ra = reinterpret_cast<float *>(r + n_bins), *rb = ra + rowf;   // (only with warp_rows)

    float2 *row = S + f * (int64_t)ldc;
    float2 sv[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int k = lane + WAVE * i;
        sv[i] = k < n_bins ? row[k] : make_float2(0.f, 0.f);
    }
    const int64_t src = row_src ? row_src[f] : f;
    const float *er = env + src * (int64_t)ld;
    float ev[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int k = lane + WAVE * i;
        ev[i] = er[k < n_bins ? k : n_bins - 1];
    }
    const int note = frame_note[f];
    const goofer_note_params &p = params[note];
    float f0f;                                       // f0 already carries pitch_shift
    bool voiced;
    if (picks) {                                     // (f0, mask) record of the frame, written by the map kernel
        const float2 pv = picks[f];
        f0f = pv.x;
        voiced = p.apply_brightness && pv.y > 0.f;
    } else {
        const int64_t t = f - frame_off[note];
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t pk = pick_index(t, n, hop);
        f0f = n > 0 ? f0[base + pk] : 0.f;
        voiced = p.apply_brightness && n > 0 && mask[base + pk] > 0.f;
    }
    double t5[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) t5[j] = taps5[j];

    const float *eg = nullptr;                       // LDS row of the warped envelope, or registers when no warp ran
    if (row_src && warp_rows) {
        double fs[4];
        bool warp = false;
        for (int k = 0; k < 4; ++k) {
            fs[k] = p.f_shift[k];
            warp |= fs[k] != 1.0;
        }
        if ((warp && formants) || p.formant_shift != 1.0f) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int k = lane + WAVE * i;
                if (k < n_bins) ra[k] = ev[i];
            }
            wave_lds_sync();
            eg = warp_row(ra, rb, n_bins, grid, formants ? formants + src * 4 : nullptr, fs, warp, (double)p.formant_shift, lane,
                          s_seg[wave]);
        }
    }
    const int cut = p.cut_below_f0;
    // bins 64 and up sit a whole 64-bin stride above the lowest bins: when f0 + 100 Hz is still below bin 64 their high-pass
    // factor is exactly 1.0f (1 + exp(-z) rounds to 1 for z > 18, i.e. 90 Hz above f0; rcp(1) = 1) and is not evaluated
    const bool hp_low_only = n_bins > WAVE && t_fq[WAVE] - f0f > 100.0f;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int k = lane + WAVE * i;
        if (k < n_bins) {
            float2 s = sv[i];
            if (cut && (i == 0 || !hp_low_only)) {
                float h = hp_mask(t_fq[k], f0f);
                s.x *= h; s.y *= h;
            }
            mx = fmaxf(mx, s.x * s.x + s.y * s.y);                 // |s|^2: the square root is taken once, of the maximum
            const float g = eg ? eg[k] : ev[i];
            const float bo = t_bo[k];
            s.x = (s.x * g) * bo;
            s.y = (s.y * g) * bo;
            if (voiced) {
                const float br = t_br[k];
                s.x *= br; s.y *= br;
                r[k] = s;
            } else {
                store_f2(row + k, s, NT);
            }
        }
    }
    mx = __builtin_amdgcn_sqrtf(wave_max(mx)) + 1e-8f;              // max(|s| + 1e-8) = sqrt(max |s|^2) + 1e-8: sqrt is monotone
    if (lane == 0) atomic_max_pos(note_mag + note, mx);
    if (voiced) {
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int k = lane + WAVE * i;
            if (k < n_bins) store_f2(row + k, blur5(r, k, n_bins, t5), NT);
        }
    }
}

int launch_harm_shape(goofer_ctx *ctx, float2 *S, int ldc, int64_t total_frames, const int *frame_note, const int64_t *frame_off,
                      const int64_t *sample_off, const float *f0, const float *mask, const float *env, int ld,
                      const goofer_note_params *params, float *note_mag, const int64_t *row_src, const double *formants,
                      bool no_warp, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    const goofer_plan_t &pl = ctx->plan;
    const dim3 grid((unsigned)((total_frames + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    const int warp_rows = (row_src && !no_warp) ? 1 : 0;       // the batch may warp rows here: the kernel needs its two fp32 LDS rows
    const size_t lds = harm_shape_lds(pl.n_bins, warp_rows != 0);
#define HARM_SHAPE_NT(IT, NT)                                                                                                      \
    hipLaunchKernelGGL((k_harm_shape<IT, NT>), grid, dim3(256), lds, st, S, ldc, total_frames, frame_note, frame_off, sample_off, f0, \
                       mask, env, ld, params, note_mag, pl.freqs, pl.boost, pl.bright_h, pl.blur5, pl.n_bins, pl.hop, row_src,    \
                       formants, make_warp_grid(pl.sr, pl.n_bins), ctx->frame_picks, warp_rows)
    // (the shaped rows are written once and read once, by the inverse transform: non-temporal stores)
#define HARM_SHAPE(IT) HARM_SHAPE_NT(IT, true)
    // bins per lane: the instantiation with the smallest count that covers the row (the kernels test k < n_bins per bin)
    const int chunks = (pl.n_bins + WAVE - 1) / WAVE;
    if (chunks <= 5) HARM_SHAPE(5);
    else if (chunks <= 7) HARM_SHAPE(7);
    else if (chunks <= 9) HARM_SHAPE(9);
    else if (chunks <= 13) HARM_SHAPE(13);
    else if (chunks <= 17) HARM_SHAPE(17);
    else return goofer_fail(ctx, GOOFER_EINVAL, "unsupported bin count %d", pl.n_bins);
#undef HARM_SHAPE
#undef HARM_SHAPE_NT
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// float2 slots of LDS per wave of k_noise_spectra: an fp32 row padded to 16 bytes + a complex row, an even count (16-byte waves)
__host__ __device__ static inline int noise_spectra_wave_f2(int n_bins, bool shared_rows)
{
    return shared_rows ? ((n_bins + 2) & ~1) : (((((n_bins + 4) & ~3) / 2 + n_bins) + 1) & ~1);
}

// S_uv = U * env_noise ; S_br = (U * env_noise) * HP, brightened + blurred on voiced frames.
// PHI: injected phases (parity runs; accurate libm sin / cos) instead of Philox + hardware sin / cos — a template parameter: the
// phases of a row are ITERS registers the production kernel does not need.
// reg_blur (decided by the launcher): the sigma-1.75 blur runs in registers (below); the fp32 row then shares its LDS with the
// complex row of the 5-tap blur — the two are never live together on that path — and a workgroup takes 33 KB instead of 49 KB at
// n_fft 2048 (four per CU instead of three).
// RB: reg_blur_ok as a template parameter — the LDS version of the blur, compiled beside the register one, cost the kernel 30
// registers it never used (ITERS 17: 144 -> 114, three -> four waves per SIMD; ITERS 9: 96 -> 62, five -> eight)
template <int ITERS, bool NT, bool PHI, bool RB>
__global__ __launch_bounds__(256, (ITERS <= 9 || RB) ? 4 : 3) void k_noise_spectra(float2 *__restrict__ S_uv, float2 *__restrict__ S_br, int ldc,
                                                       int64_t total_frames, const int *__restrict__ frame_note,
                                                       const int64_t *__restrict__ frame_off, const int64_t *__restrict__ sample_off,
                                                       const float *__restrict__ f0, const float *__restrict__ mask,
                                                       const float *__restrict__ env_noise, const float *__restrict__ phi, int ld,
                                                       const goofer_note_params *__restrict__ params, uint64_t seed,
                                                       const float *__restrict__ freqs, const float *__restrict__ bright,
                                                       const double *__restrict__ taps5, int n_bins, int hop,
                                                       const int64_t *__restrict__ row_src, const double *__restrict__ taps175,
                                                       const float2 *__restrict__ picks, const unsigned char *__restrict__ frame_skip)
{
    constexpr int reg_blur_ok = RB ? 1 : 0;
    // env_noise is either the already-blurred [frames x ld] matrix (row_src == nullptr) or the source rows, in
    // which case the sigma-1.75 bin blur (GOOFER.py:993) runs here from the LDS row.
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t f = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    if (f >= total_frames) return;
    // bit 0 / 1: the frame's unvoiced / breath spectrum cannot reach a non-zero sample (k_frame_skip): not written, and the
    // overlap-add kernel will not read it
    const unsigned sk = frame_skip ? (unsigned)frame_skip[f] : 0u;
    if (sk == 3u) return;
    // per wave: the fp32 row (16-byte aligned: the register blur stores it in 16-byte pieces), then the complex row of the 5-tap blur
    const int rowf4 = (n_bins + 4) & ~3;
    float *ra = reinterpret_cast<float *>(reinterpret_cast<float2 *>(smem) + (size_t)wave * noise_spectra_wave_f2(n_bins, reg_blur_ok != 0));
    float2 *r = reg_blur_ok ? reinterpret_cast<float2 *>(ra) : reinterpret_cast<float2 *>(ra + rowf4);

    const int64_t src = row_src ? row_src[f] : f;
    const float *er = env_noise + src * (int64_t)ld;
    float ev[ITERS], br[ITERS], ph[PHI ? ITERS : 1];
    const float fq0 = freqs[lane], fq64 = freqs[n_bins > WAVE ? WAVE : 0];
    if constexpr (!RB) {                                     // (the register blur loads the row in its own layout)
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int k = lane + WAVE * i;
            ev[i] = er[k < n_bins ? k : n_bins - 1];
        }
    }
    if constexpr (PHI) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int k = lane + WAVE * i;
            ph[i] = phi[f * (int64_t)ld + (k < n_bins ? k : n_bins - 1)];
        }
    }
    const int note = frame_note[f];
    const goofer_note_params p = params[note];
    const int64_t t = f - frame_off[note];
    float f0f;                                       // f0 already carries pitch_shift
    bool voiced;
    if (picks) {                                     // (f0, mask) record of the frame, written by the map kernel
        const float2 pv = picks[f];
        f0f = pv.x;
        voiced = p.apply_brightness && pv.y > 0.f;
    } else {
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t pk = pick_index(t, n, hop);
        f0f = n > 0 ? f0[base + pk] : 0.f;
        voiced = p.apply_brightness && n > 0 && mask[base + pk] > 0.f;
    }
    double t5[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) t5[j] = taps5[j];

    float2 *ru = S_uv + f * (int64_t)ldc;
    float2 *rb = S_br + f * (int64_t)ldc;
    const bool do_blur = row_src && taps175;                 // taps175 == nullptr: the rows are the noise envelope already
    float t175[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) t175[j] = do_blur ? (float)taps175[j] : 0.f;
    // Rows of 64 NB + 1 bins (n_fft 1024 / 2048): the sigma-1.75 blur as register arithmetic.  A lane takes NB CONSECUTIVE bins
    // (16-byte loads), the eight bins on either side come from the neighbouring lanes through DPP wave shifts, lane 63 also
    // blurs the Nyquist bin; numpy's 'reflect' at the two ends of the row is a select on lanes 0 and 63.  The same products in
    // the same ascending tap order as the LDS version below (bit-identical), for 16 vector moves and NB / 4 + 1 loads instead of
    // 15 LDS reads per bin (255 per frame at n_fft 2048) and a bounds test per bin.  The blurred row then goes through LDS once
    // into the layout of the spectrum rows, bin lane + 64 i.
    constexpr int NB = ITERS - 1;
    const bool reg_blur = do_blur && reg_blur_ok != 0;      // (the launcher checked: 64 NB + 1 bins, 16-byte aligned rows)
    bool blurred = false;
    if constexpr (RB && (NB == 8 || NB == 16)) {
        if (reg_blur) {
            float x[NB + 16];
#pragma unroll
            for (int q = 0; q < NB / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(er + NB * lane + 4 * q);
                x[8 + 4 * q] = v.x; x[8 + 4 * q + 1] = v.y; x[8 + 4 * q + 2] = v.z; x[8 + 4 * q + 3] = v.w;
            }
            const float e_ny = er[n_bins - 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[NB + j]), 0x138, 0xf, 0xf, false));           // wave_shr:1 <- lane - 1: its last eight
                x[8 + NB + j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[8 + j]), 0x130, 0xf, 0xf, false));   // wave_shl:1 <- lane + 1: its first eight
            }
            // bins -1..-7 are bins 1..7, bins B..B+6 are bins B-2..B-8 (B - 1 is the Nyquist bin, lane 63's extra one)
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                x[8 - j] = lane == 0 ? x[8 + j] : x[8 - j];
                x[8 + NB + j] = lane == 63 ? x[8 + NB - j] : x[8 + NB + j];
            }
            x[8 + NB] = lane == 63 ? e_ny : x[8 + NB];
            float o[NB + 1];
#pragma unroll
            for (int j = 0; j <= NB; ++j) {
                float acc = t175[0] * x[j + 1];
#pragma unroll
                for (int q = 1; q < 15; ++q) acc = fmaf(t175[q], x[j + 1 + q], acc);
                o[j] = acc;
            }
#pragma unroll
            for (int q = 0; q < NB / 4; ++q)
                *reinterpret_cast<float4 *>(ra + NB * lane + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            if (lane == 63) ra[n_bins - 1] = o[NB];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int k = lane + WAVE * i;
                ev[i] = ra[k < n_bins ? k : n_bins - 1];
            }
            blurred = true;
        }
    }
    if constexpr (!RB) {
        if (do_blur && !blurred) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int k = lane + WAVE * i;
                if (k < n_bins) ra[k] = ev[i];
            }
            wave_lds_sync();
        }
    }
    // (the brightness curve after the blur: behind it the row's registers are free again)
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int k = lane + WAVE * i;
        br[i] = bright[k < n_bins ? k : n_bins - 1];
    }
    const uint64_t key = seed ^ ((uint64_t)p.seed[0] | ((uint64_t)p.seed[1] << 32));
    uint4 rnd = make_uint4(0, 0, 0, 0);
    const bool hp_low_only = n_bins > WAVE && fq64 - f0f > 100.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int k = lane + WAVE * i;
        if (k >= n_bins) continue;
        float c, s;
        if constexpr (PHI) {
            c = cosf(ph[i]);
            s = sinf(ph[i]);
        } else {
            // one Philox-4x32-7 block feeds eight bins of this lane (bins lane + 64 i, i = 8q..8q+7), 16 bits each
            if ((i & 7) == 0) rnd = philox_4x32(key, (uint64_t)t, (uint32_t)(lane + WAVE * (i >> 3)));
            const float rev = (float)philox_half(rnd, i & 7) * (1.0f / 65536.0f);      // phase / 2 pi, uniform in [0, 1)
            c = __builtin_amdgcn_cosf(rev);
            s = __builtin_amdgcn_sinf(rev);
        }
        float e;
        if (!RB && do_blur && !blurred) {
            // sigma-1.75 blur of the envelope row (GOOFER.py:993): the reference accumulates in fp64 and the product with
            // the unit phasor is rounded to complex64; fp32 FMAs in tap order stay within ~2e-7 relative
            float acc;
            if (k >= 7 && k + 7 < n_bins) {
                acc = t175[0] * ra[k - 7];
#pragma unroll
                for (int j = 1; j < 15; ++j) acc = fmaf(t175[j], ra[k + j - 7], acc);
            } else {
                // single reflection is enough for a 7-bin reach; the general periodic map costs a 64-bit modulo
                auto refl = [&](int q) { return q < 0 ? -q : (q >= n_bins ? 2 * (n_bins - 1) - q : q); };   // n_bins >= 257 (goofer_plan)
                acc = t175[0] * ra[refl(k - 7)];
#pragma unroll
                for (int j = 1; j < 15; ++j) acc = fmaf(t175[j], ra[refl(k + j - 7)], acc);
            }
            e = acc;
        } else {
            e = ev[i];
        }
        float2 u2 = make_float2(c * e, s * e);
        if (!(sk & 1u)) store_f2(ru + k, u2, NT);
        if (sk & 2u) continue;
        // bins 64 and up: the factor is exactly 1.0f when f0 + 100 Hz lies below bin 64 (see k_harm_shape)
        const float h = i == 0 ? hp_mask(fq0, f0f) : (hp_low_only ? 1.0f : hp_mask(freqs[k], f0f));
        float2 b = make_float2(u2.x * h, u2.y * h);
        if (voiced) { b.x *= br[i]; b.y *= br[i]; r[k] = b; }
        else store_f2(rb + k, b, NT);
    }
    if (voiced && !(sk & 2u)) {
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int k = lane + WAVE * i;
            if (k < n_bins) store_f2(rb + k, blur5(r, k, n_bins, t5), NT);
        }
    }
}

int launch_noise_spectra(goofer_ctx *ctx, float2 *S_uv, float2 *S_br, int ldc, int64_t total_frames, const int *frame_note,
                         const int64_t *frame_off, const int64_t *sample_off, const float *f0, const float *mask,
                         const float *env_noise, const float *phi, int ld, const goofer_note_params *params, uint64_t seed,
                         const int64_t *row_src, bool preblurred, const unsigned char *frame_skip, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    const goofer_plan_t &pl = ctx->plan;
    const dim3 grid((unsigned)((total_frames + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    const int chunks0 = (pl.n_bins + WAVE - 1) / WAVE;
    // the register blur: rows of 64 NB + 1 bins with NB = 8 / 16 (the instantiations 9 / 17), 16-byte aligned envelope rows, and a blur to run
    const int reg_blur = (row_src && !preblurred && (chunks0 == 9 || chunks0 == 17) && pl.n_bins == WAVE * (chunks0 - 1) + 1 && (ld & 3) == 0 &&
                          (((uintptr_t)env_noise) & 15) == 0) ? 1 : 0;
    const size_t lds = sizeof(float2) * ROWS_PER_BLOCK * noise_spectra_wave_f2(pl.n_bins, reg_blur != 0);
#define NOISE_SPECTRA_RB(IT, NT, PH, RBV)                                                                                          \
    hipLaunchKernelGGL((k_noise_spectra<IT, NT, PH, RBV>), grid, dim3(256), lds, st, S_uv, S_br, ldc, total_frames, frame_note, frame_off,  \
                       sample_off, f0, mask, env_noise, phi, ld, params, seed, pl.freqs, pl.bright_b, pl.blur5, pl.n_bins, pl.hop, \
                       row_src, preblurred ? (const double *)nullptr : pl.blur175, ctx->frame_picks, frame_skip)
#define NOISE_SPECTRA_P(IT, NT, PH)                                                                                                \
    do {                                                                                                                           \
        if constexpr (IT == 9 || IT == 17) {                                                                                       \
            if (reg_blur) NOISE_SPECTRA_RB(IT, NT, PH, true);                                                                      \
            else NOISE_SPECTRA_RB(IT, NT, PH, false);                                                                              \
        } else {                                                                                                                   \
            NOISE_SPECTRA_RB(IT, NT, PH, false);                                                                                   \
        }                                                                                                                          \
    } while (0)
#define NOISE_SPECTRA(IT)                                                                                                          \
    do {                                                                                                                           \
        if (phi) NOISE_SPECTRA_P(IT, true, true);                                                                                  \
        else NOISE_SPECTRA_P(IT, true, false);                                                                                     \
    } while (0)
    // bins per lane: the instantiation with the smallest count that covers the row (the kernels test k < n_bins per bin)
    const int chunks = (pl.n_bins + WAVE - 1) / WAVE;
    if (chunks <= 5) NOISE_SPECTRA(5);
    else if (chunks <= 7) NOISE_SPECTRA(7);
    else if (chunks <= 9) NOISE_SPECTRA(9);
    else if (chunks <= 13) NOISE_SPECTRA(13);
    else if (chunks <= 17) NOISE_SPECTRA(17);
    else return goofer_fail(ctx, GOOFER_EINVAL, "unsupported bin count %d", pl.n_bins);
#undef NOISE_SPECTRA
#undef NOISE_SPECTRA_P
#undef NOISE_SPECTRA_RB
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
