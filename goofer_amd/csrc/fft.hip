// Framewise real FFT / inverse FFT / overlap-add for gfx950.
//
//   k_rfft_frames   replaces gf.stft        (GOOFER.py:355-370)
//   k_irfft_frames  replaces np.fft.irfft in gf.istft (GOOFER.py:399-400) and the `frames*window`
//                   product of _overlap_add (GOOFER.py:383)
//   k_ola_gather    replaces _overlap_add + trim/pad of istft (GOOFER.py:372-390, 402-413)
//
// One 64-lane wave owns one frame.  A real n_fft-point transform is a complex M = n_fft/2 point
// Stockham autosort FFT (lane holds M/64 points; first radix-(M/64) pass in registers straight from
// global memory, then two radix-8 passes exchanged through a padded per-wave LDS buffer) followed by
// the even/odd split that recovers the n_fft/2+1 real-input bins.  No MFMA: ~5 flop/B, HBM-bound.
#include "fft_core.h"

// ---------------------------------------------------------------------------------------------
// register budget: two frames of M / 64 sample pairs in flight; the 2048-point frame takes the 256-VGPR budget
template <int M, bool NT = false>
__global__ __launch_bounds__(256, M <= 512 ? 4 : 2) void k_rfft_frames(const float *__restrict__ x, const int64_t *__restrict__ sample_off,
                                                     const int64_t *__restrict__ frame_off, const int *__restrict__ frame_note,
                                                     int64_t total_frames, float2 *__restrict__ S, int ldc, int hop,
                                                     const float2 *__restrict__ g_tw, const float2 *__restrict__ g_twh,
                                                     const float *__restrict__ g_win)
{
    constexpr int R = fft_cfg<M>::R;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WAVES_PER_BLOCK * fft_cfg<M>::BUF);
    load_tables<M>(tw, twh, win, g_tw, g_twh, g_win);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *buf = bufs + wave * fft_cfg<M>::BUF;
    const int64_t f_begin = (int64_t)blockIdx.x * FRAMES_PER_BLOCK;
    const bool wide = (ldc & 1) == 0 && ((uintptr_t)S & 15) == 0;      // rows 16-byte aligned: two bins per store

    // raw sample pairs of a frame (reflect-padded at the note ends); the next frame's are in flight during the FFT
    auto fetch = [&](int64_t f, float2 (&raw)[R]) {
        const int note = frame_note[f];
        const int64_t base = sample_off[note];
        const int64_t n = sample_off[note + 1] - base;
        const int64_t t = f - frame_off[note];
        const int64_t start = t * hop - M;            // first sample of the frame, un-padded coordinates
        const float *xs = x + base;
        if (start >= 0 && start + 2 * M <= n) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int m = lane + WAVE * r;
                raw[r] = make_float2(xs[start + 2 * m], xs[start + 2 * m + 1]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int m = lane + WAVE * r;
                float a = n > 0 ? xs[reflect_index(start + 2 * m, n)] : 0.f;
                float b = n > 0 ? xs[reflect_index(start + 2 * m + 1, n)] : 0.f;
                raw[r] = make_float2(a, b);
            }
        }
    };
    // One frame: window in place, transform, split, store.  `nf` = the frame whose sample pairs go into the same registers as
    // soon as the transform has consumed them — issued BEFORE this frame's stores.  Loads and stores share one in-order
    // counter (vmcnt): a wave that waits for a load also waits for every store issued before it, and a store completes
    // microseconds after issue.  With two frames in flight and the loads ahead of the stores, the wait for frame i + 1's
    // samples covers only stores that are two frames old — the kernel then runs at the larger of its load / compute time
    // and its store time instead of their sum (0.33 -> 0.2x ms on the 1024-note batch).
    auto frame = [&](int64_t f, float2 (&v)[R], int64_t nf) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = lane + WAVE * r;
            v[r] = make_float2(v[r].x * win[2 * m], v[r].y * win[2 * m + 1]);
        }
        wave_fft<M>(v, buf, tw, lane);
        if (nf >= 0) fetch(nf, v);

        // even/odd split: X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k])
        float2 *row = S + f * (int64_t)ldc;
        auto split = [&](int k) {
            const float2 zk = buf[lds_pad(k)];
            const float2 zm = buf[lds_pad(k == 0 ? 0 : M - k)];
            const float2 w = (k <= M / 2) ? twh[k] : make_float2(-twh[M - k].x, twh[M - k].y);
            const float2 A = make_float2(zk.x + zm.x, zk.y - zm.y);
            const float2 B = make_float2(zk.x - zm.x, zk.y + zm.y);
            const float2 C = cmul(w, B);
            return make_float2(0.5f * (A.x + C.y), 0.5f * (A.y - C.x));
        };
        if (wide) {
            // The split reads stay conflict-free (bin k = lane + 64 r per lane); neighbouring lanes then trade one bin each
            // (a 2 x 2 transpose over two r values, one quad-permute DPP move per dword), so that an even lane holds bins
            // (k, k + 1) of row segment r and the odd lane beside it bins (k - 1, k) of segment r + 1: every lane issues
            // 16-byte stores, on 128-byte aligned rows.
            const bool odd = lane & 1;
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const float2 xa = split(lane + WAVE * r), xb = split(lane + WAVE * (r + 1));
                const float2 send = odd ? xa : xb;
                float2 recv;
                recv.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send.x), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                recv.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send.y), 0xB1, 0xF, 0xF, false));
                const float4 o = odd ? make_float4(recv.x, recv.y, xb.x, xb.y) : make_float4(xa.x, xa.y, recv.x, recv.y);
                const int k0 = odd ? lane - 1 + WAVE * (r + 1) : lane + WAVE * r;
                typedef float v4f_t __attribute__((ext_vector_type(4)));
                if (NT) __builtin_nontemporal_store(v4f_t{o.x, o.y, o.z, o.w}, reinterpret_cast<v4f_t *>(row + k0));   // write-once rows: not kept in L2
                else *reinterpret_cast<float4 *>(row + k0) = o;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + WAVE * r;
                row[k] = split(k);
            }
        }
        if (lane == 0) {
            float2 z0 = buf[0];
            row[M] = make_float2(z0.x - z0.y, 0.f);
        }
        wave_lds_sync();
    };

    // frames f_begin + wave + 4 q of this wave, q = 0 .. FRAMES_PER_BLOCK / 4 - 1, two at a time in registers A and B
    constexpr int PER_WAVE = FRAMES_PER_BLOCK / WAVES_PER_BLOCK;
    static_assert(PER_WAVE % 2 == 0, "frames per wave are processed in pairs");
    auto frame_id = [&](int q) -> int64_t {
        const int64_t f = f_begin + wave + (int64_t)WAVES_PER_BLOCK * q;
        return (q < PER_WAVE && f < total_frames) ? f : -1;
    };
    float2 A[R], B[R];
    if (frame_id(0) >= 0) fetch(frame_id(0), A);
    if (frame_id(1) >= 0) fetch(frame_id(1), B);
    for (int q = 0; q < PER_WAVE; q += 2) {
        if (frame_id(q) < 0) break;                   // wave-uniform
        frame(frame_id(q), A, frame_id(q + 2));
        if (frame_id(q + 1) < 0) break;
        frame(frame_id(q + 1), B, frame_id(q + 3));
    }
}

// S row -> windowed time frame (fp32 irfft value times window[j], the `val` of _overlap_add).
template <int M>
__global__ __launch_bounds__(256) void k_irfft_frames(const float2 *__restrict__ S, int ldc, int64_t total_frames,
                                                      float *__restrict__ frames, const float2 *__restrict__ g_tw,
                                                      const float2 *__restrict__ g_twh, const float *__restrict__ g_win)
{
    constexpr int R = fft_cfg<M>::R;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WAVES_PER_BLOCK * fft_cfg<M>::BUF);
    load_tables<M>(tw, twh, win, g_tw, g_twh, g_win);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *buf = bufs + wave * fft_cfg<M>::BUF;
    const int64_t f_begin = (int64_t)blockIdx.x * FRAMES_PER_BLOCK;
    const float inv_m = 0.5f / (float)M;                    // 1/M of the transform and the 1/2 of the input stage (irfft_pre)

    for (int i = wave; i < FRAMES_PER_BLOCK; i += WAVES_PER_BLOCK) {
        const int64_t f = f_begin + i;
        if (f >= total_frames) break;
        const float2 *row = S + f * (int64_t)ldc;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int k = lane + WAVE * r;
            float2 xk = row[k];
            float2 xm = row[M - k];
            if (k == 0) { xk.y = 0.f; xm.y = 0.f; }   // irfft ignores Im of DC and Nyquist
            float2 wc = (k <= M / 2) ? cconj(twh[k]) : make_float2(-twh[M - k].x, -twh[M - k].y);
            // Z = (A + i C)/2 ; inverse FFT = conj(FFT(conj Z))
            v[r] = irfft_pre(xk, xm, wc);
        }
        wave_fft<M>(v, buf, tw, lane);
        float2 *out = reinterpret_cast<float2 *>(frames + f * (int64_t)(2 * M));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int m = lane + WAVE * r;
            float2 z = buf[lds_pad(m)];
            float a = z.x * inv_m, b = -z.y * inv_m;
            out[m] = make_float2(a * win[2 * m], b * win[2 * m + 1]);
        }
        wave_lds_sync();
    }
}

// y[i] = (sum over covering frames, ascending, of frames[fr][p - fr*hop]) / (same sum of w^2), fp32,
// p = i + n_fft/2; samples past hop*(T-1) are the zero padding of istft.  Optional per-note scale.
__global__ __launch_bounds__(256) void k_ola_gather(const float *__restrict__ frames, const float *__restrict__ win_sq,
                                                    const int64_t *__restrict__ sample_off, const int64_t *__restrict__ frame_off,
                                                    int n_notes, int64_t total_samples, int n_fft, int hop,
                                                    float *__restrict__ y, const float *__restrict__ divisor)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo_n, hi_n;
    block_note_range(sample_off, n_notes, g0, total_samples, s_pair, lo_n, hi_n);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total_samples) return;

    auto body = [&](int note) {
        const int64_t i = g - sample_off[note];
        const int64_t fbase = frame_off[note];
        const int64_t T = frame_off[note + 1] - fbase;
        float out = 0.f;
        if (i < (int64_t)hop * (T - 1)) {
            const int64_t p = i + n_fft / 2;
            int64_t lo = p - n_fft + 1;
            lo = lo <= 0 ? 0 : (lo + hop - 1) / hop;
            int64_t hi = p / hop;
            if (hi > T - 1) hi = T - 1;
            float acc = 0.f, ws = 0.f;
            for (int64_t fr = lo; fr <= hi; ++fr) {
                int j = (int)(p - fr * hop);
                acc += frames[(fbase + fr) * n_fft + j];
                ws += win_sq[j];
            }
            if (ws > 1e-9f) acc /= ws;
            out = acc;
            if (divisor) out = out / divisor[note];
        }
        y[g] = out;
    };
    if (lo_n == hi_n) {
        body(lo_n);
    } else {
        int note = lo_n;
        while (sample_off[note + 1] <= g) ++note;
        body(note);
    }
}

// ---------------------------------------------------------------------------------------------
// Transform sizes without a radix plan (any even n_fft up to 2048): Bluestein's chirp-z form of the M = n_fft / 2 point
// complex DFT, Z_k = conj(c_k) sum_n (z_n conj(c_n)) c_{k-n} with c_n = exp(i pi n^2 / M) — a circular convolution of length
// L >= 2 M - 1 (a power of two up to 2048: wave_fft<L>) with the wrapped chirp, whose transform goofer_plan made in fp64.  Two L-point
// transforms and three complex products per point; fp32 error ~4e-7 relative.  The real-input split / conj-trick stages around
// it are the ones of the native kernels.
//
// v[r] = z[lane + 64 r] (anything for indices >= M) on entry; Z[k], k < M, in natural order in buf[k] (un-padded) on exit.
template <int L>
__device__ __forceinline__ void bluestein_dft(float2 (&v)[L / 64], int M, float2 *buf, const float2 *twl, const float2 *chirp,
                                              const float2 *bhat, int lane)
{
    constexpr int R = L / 64;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = lane + WAVE * r;
        v[r] = n < M ? cmul(v[r], cconj(chirp[n])) : make_float2(0.f, 0.f);
    }
    wave_fft<L>(v, buf, twl, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = lane + WAVE * r;
        v[r] = cconj(cmul(buf[lds_pad(n)], bhat[n]));          // inverse transform = conj(FFT(conj .)) / L
    }
    wave_lds_sync();
    wave_fft<L>(v, buf, twl, lane);
    const float inv_l = 1.0f / (float)L;
    float2 z[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = lane + WAVE * r;
        const float2 y = buf[lds_pad(k < M ? k : 0)];
        z[r] = cmul(make_float2(y.x * inv_l, -(y.y * inv_l)), cconj(chirp[k < M ? k : 0]));
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = lane + WAVE * r;
        if (k < M) buf[k] = z[r];
    }
    wave_lds_sync();
}

template <int L> struct bluestein_lds {
    static constexpr size_t bytes = sizeof(float2) * (2 * L + L / 2 + (L / 2 + 2) + WAVES_PER_BLOCK * fft_cfg<L>::BUF) + sizeof(float) * L;
    float2 *twl, *bhat, *chirp, *twh, *bufs;
    float *win;
    __device__ __forceinline__ void carve(unsigned char *smem, int M, const float2 *g_twl, const float2 *g_bhat, const float2 *g_chirp,
                                          const float2 *g_twh, const float *g_win)
    {
        twl = reinterpret_cast<float2 *>(smem);
        bhat = twl + L;
        chirp = bhat + L;
        twh = chirp + L / 2;
        bufs = twh + (L / 2 + 2);
        win = reinterpret_cast<float *>(bufs + WAVES_PER_BLOCK * fft_cfg<L>::BUF);
        for (int i = threadIdx.x; i < L; i += blockDim.x) { twl[i] = g_twl[i]; bhat[i] = g_bhat[i]; }
        for (int i = threadIdx.x; i < M; i += blockDim.x) chirp[i] = g_chirp[i];
        for (int i = threadIdx.x; i <= M; i += blockDim.x) twh[i] = g_twh[i];
        for (int i = threadIdx.x; i < 2 * M; i += blockDim.x) win[i] = g_win[i];
        __syncthreads();
    }
};

template <int L>
__global__ __launch_bounds__(256) void k_rfft_bluestein(const float *__restrict__ x, const int64_t *__restrict__ sample_off,
                                                        const int64_t *__restrict__ frame_off, const int *__restrict__ frame_note,
                                                        int64_t total_frames, float2 *__restrict__ S, int ldc, int hop, int M,
                                                        const float2 *__restrict__ g_twl, const float2 *__restrict__ g_bhat,
                                                        const float2 *__restrict__ g_chirp, const float2 *__restrict__ g_twh,
                                                        const float *__restrict__ g_win)
{
    constexpr int R = L / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    bluestein_lds<L> t;
    t.carve(smem, M, g_twl, g_bhat, g_chirp, g_twh, g_win);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *buf = t.bufs + wave * fft_cfg<L>::BUF;
    const int64_t f_begin = (int64_t)blockIdx.x * FRAMES_PER_BLOCK;
    for (int i = wave; i < FRAMES_PER_BLOCK; i += WAVES_PER_BLOCK) {
        const int64_t f = f_begin + i;
        if (f >= total_frames) break;                         // wave-uniform; no block barrier below
        const int note = frame_note[f];
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t start = (f - frame_off[note]) * hop - M;   // first sample of the frame, un-padded coordinates
        const float *xs = x + base;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = lane + WAVE * r;
            float a = 0.f, b = 0.f;
            if (m < M && n > 0) {
                a = xs[reflect_index(start + 2 * m, n)];
                b = xs[reflect_index(start + 2 * m + 1, n)];
            }
            v[r] = m < M ? make_float2(a * t.win[2 * m], b * t.win[2 * m + 1]) : make_float2(0.f, 0.f);
        }
        bluestein_dft<L>(v, M, buf, t.twl, t.chirp, t.bhat, lane);
        // even/odd split: X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k]); X[M] from Z[0]
        float2 *row = S + f * (int64_t)ldc;
        for (int k = lane; k <= M; k += WAVE) {
            float2 X;
            if (k == M) {
                const float2 z0 = buf[0];
                X = make_float2(z0.x - z0.y, 0.f);
            } else {
                const float2 zk = buf[k], zm = buf[k == 0 ? 0 : M - k], w = t.twh[k];
                const float2 A = make_float2(zk.x + zm.x, zk.y - zm.y), B = make_float2(zk.x - zm.x, zk.y + zm.y);
                const float2 C = cmul(w, B);
                X = make_float2(0.5f * (A.x + C.y), 0.5f * (A.y - C.x));
            }
            row[k] = X;
        }
        wave_lds_sync();
    }
}

template <int L>
__global__ __launch_bounds__(256) void k_irfft_bluestein(const float2 *__restrict__ S, int ldc, int64_t total_frames, float *__restrict__ frames,
                                                         int M, const float2 *__restrict__ g_twl, const float2 *__restrict__ g_bhat,
                                                         const float2 *__restrict__ g_chirp, const float2 *__restrict__ g_twh,
                                                         const float *__restrict__ g_win)
{
    constexpr int R = L / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    bluestein_lds<L> t;
    t.carve(smem, M, g_twl, g_bhat, g_chirp, g_twh, g_win);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *buf = t.bufs + wave * fft_cfg<L>::BUF;
    const int64_t f_begin = (int64_t)blockIdx.x * FRAMES_PER_BLOCK;
    const float inv_m = 0.5f / (float)M;                     // 1/M of the transform and the 1/2 of the input stage (irfft_pre)
    for (int i = wave; i < FRAMES_PER_BLOCK; i += WAVES_PER_BLOCK) {
        const int64_t f = f_begin + i;
        if (f >= total_frames) break;
        const float2 *row = S + f * (int64_t)ldc;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = lane + WAVE * r;
            v[r] = make_float2(0.f, 0.f);
            if (k < M) {
                float2 xk = row[k], xm = row[M - k];
                if (k == 0) { xk.y = 0.f; xm.y = 0.f; }       // irfft ignores Im of DC and Nyquist
                v[r] = irfft_pre(xk, xm, cconj(t.twh[k]));     // Z = (A + i C)/2 ; inverse FFT = conj(FFT(conj Z))
            }
        }
        bluestein_dft<L>(v, M, buf, t.twl, t.chirp, t.bhat, lane);
        float2 *out = reinterpret_cast<float2 *>(frames + f * (int64_t)(2 * M));
        for (int m = lane; m < M; m += WAVE) {
            const float2 z = buf[m];
            const float a = z.x * inv_m, b = -z.y * inv_m;
            out[m] = make_float2(a * t.win[2 * m], b * t.win[2 * m + 1]);
        }
        wave_lds_sync();
    }
}

template <int L>
static int rfft_bluestein_impl(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off, const int *frame_note,
                               int64_t total_frames, float2 *S, int ldc, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    const unsigned blocks = (unsigned)((total_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK);
    if (bluestein_lds<L>::bytes > 64 * 1024)
        if (int rc = kernel_allow_max_lds(ctx, (const void *)k_rfft_bluestein<L>)) return rc;
    hipLaunchKernelGGL(k_rfft_bluestein<L>, dim3(blocks), dim3(256), bluestein_lds<L>::bytes, st, x, sample_off, frame_off, frame_note,
                       total_frames, S, ldc, p.hop, p.n_fft / 2, p.bl_tw, p.bl_bhat, p.bl_chirp, p.bl_twh, p.window);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

template <int L>
static int irfft_bluestein_impl(goofer_ctx *ctx, const float2 *S, int ldc, int64_t total_frames, float *frames, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    const unsigned blocks = (unsigned)((total_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK);
    if (bluestein_lds<L>::bytes > 64 * 1024)
        if (int rc = kernel_allow_max_lds(ctx, (const void *)k_irfft_bluestein<L>)) return rc;
    hipLaunchKernelGGL(k_irfft_bluestein<L>, dim3(blocks), dim3(256), bluestein_lds<L>::bytes, st, S, ldc, total_frames, frames,
                       p.n_fft / 2, p.bl_tw, p.bl_bhat, p.bl_chirp, p.bl_twh, p.window);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_frame_note(const int64_t *__restrict__ frame_off, int n_notes, int64_t total_frames, int *__restrict__ frame_note)
{
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < total_frames) frame_note[f] = csr_find(frame_off, n_notes, f);
}

int launch_frame_note(goofer_ctx *ctx, const int64_t *frame_off, int n_notes, int64_t total_frames, int *frame_note, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_frame_note, dim3((unsigned)((total_frames + 255) / 256)), dim3(256), 0, st, frame_off, n_notes,
                       total_frames, frame_note);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

template <int M>
static int rfft_impl(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off,
                     const int *frame_note, int64_t total_frames, float2 *S, int ldc, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    unsigned blocks = (unsigned)((total_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK);
    // (the spectrum rows — 80 % of the kernel's bytes, written once — leave as non-temporal stores: 0.40 -> 0.51 of the HBM peak)
    hipLaunchKernelGGL((k_rfft_frames<M, true>), dim3(blocks), dim3(256), fft_lds_bytes<M>(), st, x, sample_off, frame_off, frame_note,
                       total_frames, S, ldc, p.hop, p.tw_full, p.tw_half, p.window);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_rfft_frames_mapped(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off,
                              const int *frame_note, int64_t total_frames, float2 *S, int ldc, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    switch (ctx->plan.bl_L) {
    case 0: break;
    case 256: return rfft_bluestein_impl<256>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 512: return rfft_bluestein_impl<512>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 1024: return rfft_bluestein_impl<1024>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 2048: return rfft_bluestein_impl<2048>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    default: return goofer_fail(ctx, GOOFER_EINVAL, "bad Bluestein length %d", ctx->plan.bl_L);
    }
    switch (ctx->plan.n_fft) {
    case 512: return rfft_impl<256>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 1024: return rfft_impl<512>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 2048: return rfft_impl<1024>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 768: return rfft_impl<384>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    case 1536: return rfft_impl<768>(ctx, x, sample_off, frame_off, frame_note, total_frames, S, ldc, st);
    }
    return goofer_fail(ctx, GOOFER_EINVAL, "unsupported n_fft %d", ctx->plan.n_fft);
}

template <int M>
static int irfft_impl(goofer_ctx *ctx, const float2 *S, int ldc, int64_t total_frames, float *frames, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    unsigned blocks = (unsigned)((total_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK);
    hipLaunchKernelGGL(k_irfft_frames<M>, dim3(blocks), dim3(256), fft_lds_bytes<M>(), st, S, ldc, total_frames, frames,
                       p.tw_full, p.tw_half, p.window);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_irfft_frames(goofer_ctx *ctx, const float2 *S, int ldc, int64_t total_frames, float *frames, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    switch (ctx->plan.bl_L) {
    case 0: break;
    case 256: return irfft_bluestein_impl<256>(ctx, S, ldc, total_frames, frames, st);
    case 512: return irfft_bluestein_impl<512>(ctx, S, ldc, total_frames, frames, st);
    case 1024: return irfft_bluestein_impl<1024>(ctx, S, ldc, total_frames, frames, st);
    case 2048: return irfft_bluestein_impl<2048>(ctx, S, ldc, total_frames, frames, st);
    default: return goofer_fail(ctx, GOOFER_EINVAL, "bad Bluestein length %d", ctx->plan.bl_L);
    }
    switch (ctx->plan.n_fft) {
    case 512: return irfft_impl<256>(ctx, S, ldc, total_frames, frames, st);
    case 1024: return irfft_impl<512>(ctx, S, ldc, total_frames, frames, st);
    case 2048: return irfft_impl<1024>(ctx, S, ldc, total_frames, frames, st);
    case 768: return irfft_impl<384>(ctx, S, ldc, total_frames, frames, st);
    case 1536: return irfft_impl<768>(ctx, S, ldc, total_frames, frames, st);
    }
    return goofer_fail(ctx, GOOFER_EINVAL, "unsupported n_fft %d", ctx->plan.n_fft);
}

int launch_ola_gather(goofer_ctx *ctx, const float *frames, const int64_t *sample_off, const int64_t *frame_off,
                      int n_notes, int64_t total_samples, float *y, const float *inv_scale, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    hipLaunchKernelGGL(k_ola_gather, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, frames, p.win_sq,
                       sample_off, frame_off, n_notes, total_samples, p.n_fft, p.hop, y, inv_scale);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
