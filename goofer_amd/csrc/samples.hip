// Per-sample kernels for gfx950: voicing-mask smoothing, stem gains, peak normalisation, V/B/U mix.
//
//   k_mask_short   mask[::4] -> Gaussian sigma/4 (fp64)              GOOFER.py:556-562 (smooth_mask_ds)
//   k_stem_gains   lerp-upsample the smoothed mask, scale the three stems, per-note peak
//                                                                    GOOFER.py:563-567, 1179-1193, 1210
//   k_apply_gain   gain = (1/peak)^normalize, reconstruct, V/B/U mix GOOFER.py:1208-1218, SillySampler.py:1142-1151
#include <type_traits>
#include "fft_core.h"

#include "samples_core.h"

// A workgroup owns MS_TILE consecutive short samples of the concatenated axis.  For every note that tile touches (one,
// almost always) it parks the decimated mask values of the segment plus 2*radius in LDS — each fetched once instead of
// 2*radius+1 times, and the note lookup / tap staging paid once per 1024 outputs: the kernel is a chain of dependent
// loads in front of very little arithmetic, so fewer, longer workgroups are what makes it fast — then the waves take
// the segment's 64-sample chunks in turn.  A chunk whose 64 + 2*radius window is all zeros (ones) answers 0 (the running
// sum of the taps: the same additions in the same order) without the 2*radius+1-tap loop — voicing masks are 0/1
// almost everywhere.  fp64 accumulate in tap order.
#define MS_MAXWIN 1152     // LDS floats per workgroup = 4 * MS_MAXWIN >= MS_TILE + 2*radius
#define MS_TILE 1024

__global__ __launch_bounds__(256) void k_mask_short(const float *__restrict__ mask, const int64_t *__restrict__ sample_off,
                                                    int n_notes, int64_t total_short, const double *__restrict__ taps, int radius,
                                                    double *__restrict__ short_s, double tap_sum)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_taps = reinterpret_cast<double *>(smem);
    float *s_all = reinterpret_cast<float *>(s_taps + (2 * radius + 1));          // 4 * MS_MAXWIN floats
    __shared__ int s_lo[2];
    for (int i = threadIdx.x; i < 2 * radius + 1; i += blockDim.x) s_taps[i] = taps[i];
    const int64_t g0 = (int64_t)blockIdx.x * MS_TILE;
    int64_t g1 = g0 + MS_TILE;                               // one past the tile
    if (g1 > total_short) g1 = total_short;
    if (threadIdx.x < WAVE) {                               // largest note with short_base(note) <= g, first wave cooperatively
        auto key = [&](int k) { return short_base(sample_off, k); };
        const int a = wave_find(n_notes, g0, (int)threadIdx.x, key);
        const int b = wave_find(n_notes, g1 - 1, (int)threadIdx.x, key);
        if (threadIdx.x == 0) { s_lo[0] = a; s_lo[1] = b; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_lo = __builtin_amdgcn_readfirstlane(s_lo[0]), n_hi = __builtin_amdgcn_readfirstlane(s_lo[1]);

    if (2 * radius + MS_TILE <= 4 * MS_MAXWIN) {
        for (int note = n_lo; note <= n_hi; ++note) {
            const int64_t sb = short_base(sample_off, note);
            const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
            const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
            const int64_t s0 = g0 > sb ? g0 : sb, s1 = g1 < sb + ns ? g1 : sb + ns;    // the tile's part of this note's live range
            if (s1 <= s0) continue;                          // workgroup-uniform
            const int64_t q0 = s0 - sb;
            const int len = (int)(s1 - s0);
            const float *m = mask + base;
            if (note != n_lo) __syncthreads();               // the previous note's window is no longer read
            for (int w = threadIdx.x; w < len + 2 * radius; w += blockDim.x) s_all[w] = m[MASK_DS * reflect_index(q0 - radius + w, ns)];
            __syncthreads();
            for (int c0 = wv * WAVE; c0 < len; c0 += 4 * WAVE) {
                const float *x0 = s_all + c0;                // this chunk's window: cl + 2r values
                const int cl = len - c0 < WAVE ? len - c0 : WAVE;
                const bool live = lane < cl;
                bool all0 = true, all1 = true;
                for (int w = lane; w < cl + 2 * radius; w += WAVE) {
                    const float v = x0[w];
                    all0 &= v == 0.0f;
                    all1 &= v == 1.0f;
                }
                double acc;
                if (__all(all0)) {
                    acc = 0.0;
                } else if (__all(all1)) {
                    acc = tap_sum;
                } else {
                    acc = 0.0;
                    const float *x = x0 + lane;
                    if (live)
                        for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)x[j];
                }
                if (live) short_s[s0 + c0 + lane] = acc;
            }
        }
        return;
    }
    for (int64_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
        int note = n_lo;
        while (note + 1 < n_notes && short_base(sample_off, note + 1) <= g) ++note;
        const int64_t q = g - short_base(sample_off, note);
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
        if (q >= ns) continue;
        const float *m = mask + base;
        double acc = 0.0;
        for (int j = 0; j <= 2 * radius; ++j) acc += s_taps[j] * (double)m[MASK_DS * reflect_index(q + j - radius, ns)];
        short_s[g] = acc;
    }
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
    {
        int64_t gl = g0 + (int64_t)blockDim.x * SPT - 1;
        if (gl > total_samples - 1) gl = total_samples - 1;
        block_note_range_last(sample_off, n_notes, g0, gl, s_pair, lo, hi);
    }
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

// ---------------------------------------------------------------------------------------------
// irFFT of the three stems + overlap-add + gains + per-note peak in one pass over the spectra (replaces three
// k_irfft_frames launches and k_ola3_gains: the windowed frames never go to HBM).
//
// A wave walks `run` consecutive frames of the concatenated frame axis in order.  Per stem it keeps a ring of
// n_fft accumulators in LDS indexed by (padded sample position mod n_fft): frame t adds its windowed samples to
// positions [t hop, t hop + n_fft); the last `hop` of them are first contributions and are stored, the rest are
// added — so every output sample receives its covering frames in ascending frame order, the reference's fp32
// order (GOOFER.py:379-385).  After frame t, hop t is complete: it is normalised by the summed squared window,
// scaled by the stem gains and written.  A run that starts inside a note first replays the `halo` preceding
// frames (accumulate only); the wave that owns a note's last frame also flushes the hops behind it and the
// zero-filled tail (GOOFER.py:402-413).
template <int M>
__global__ __launch_bounds__(256, M <= 512 ? 2 : 1) void k_irfft_ola3(const float2 *__restrict__ S_h, const float2 *__restrict__ S_u,
                                                    const float2 *__restrict__ S_b, int ldc, int64_t total_frames,
                                                    const int *__restrict__ frame_note, const int64_t *__restrict__ frame_off,
                                                    const int64_t *__restrict__ sample_off, int hop, int run, int halo,
                                                    const float *__restrict__ note_mag, const double *__restrict__ short_s,
                                                    const double *__restrict__ steps, const goofer_note_params *__restrict__ params,
                                                    float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    float *__restrict__ note_peak, const float2 *__restrict__ g_tw,
                                                    const float2 *__restrict__ g_twh, const float *__restrict__ g_win)
{
    constexpr int R = fft_cfg<M>::R, NF = 2 * M, BUF = fft_cfg<M>::BUF;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WAVES_PER_BLOCK * BUF);
    float *rings = win + NF;
    double *knots = reinterpret_cast<double *>(rings + (size_t)WAVES_PER_BLOCK * 3 * NF);
    load_tables<M>(tw, twh, win, g_tw, g_twh, g_win);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *buf = bufs + wave * BUF;
    float *ring = rings + (size_t)wave * 3 * NF;
    const int64_t f0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * run;
    if (f0 >= total_frames) return;                          // no block barrier below
    const int64_t f1 = f0 + run < total_frames ? f0 + run : total_frames;
    const float inv_m = 0.5f / (float)M;                     // 1/M of the transform and the 1/2 of the input stage (irfft_pre)

    int64_t fs = f0;
    {
        const int nt = frame_note[f0];
        const int64_t t0 = f0 - frame_off[nt];
        fs = f0 - (t0 < halo ? t0 : halo);
    }

    // spectrum rows of the next (frame, stem) job are fetched while the current one is transformed
    float2 nk[R], nm[R];
    auto fetch = [&](int64_t f, int stem) {
        const float2 *row = (stem == 0 ? S_h : (stem == 1 ? S_u : S_b)) + f * (int64_t)ldc;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = lane + WAVE * r;
            nk[r] = row[k];
            nm[r] = row[M - k];
        }
    };
    fetch(fs, 0);

    int note = -1;
    int64_t base = 0, fbase = 0;
    int n = 0, T = 0, ns = 0, out_len = 0;
    float mag = 1.f, rmag = 1.f, g_b = 0.f, g_u = 0.f, pk = 0.f, kps = 0.f;
    double step_n = 0.0, step_s = 0.0;
    const double *ss = nullptr;

    // per-lane constants of the output stage for its samples j = lane + 64 u of a hop: the summed squared window
    // when all covering frames exist (interior hops), and its reciprocal
    constexpr int SLOTS = M >= 1024 ? 8 : 4;                  // hop <= 64 SLOTS; larger hops recompute per sample.  Eight slots
                                                              // in the 1024-point kernel would cost it its second wave per SIMD
    float ws_u[SLOTS], rws_u[SLOTS];
    const int max_back = (NF - 1) / hop;                      // every position of a hop >= max_back has all its frames
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int j = lane + WAVE * u;
        ws_u[u] = 0.f;
        rws_u[u] = 0.f;
        if (j < hop) {
            const int back = (NF - 1 - j) / hop;
            float ws = 0.f;
            for (int q = back; q >= 0; --q) {                 // ascending frame order = descending offset
                const float w = win[j + q * hop];
                ws += w * w;
            }
            ws_u[u] = ws;
            rws_u[u] = 1.0f / ws;
        }
    }
    // per-lane constants of the transform's two ends: the conj-trick twiddle of bin k and the synthesis window of
    // sample pair m, k = m = lane + 64 r
    // (kept in registers up to M = 512; the 2048-point frame has no registers to spare and re-reads them from LDS)
    constexpr bool HOIST = M <= 512;
    auto wc_of = [&](int k) { return (k <= M / 2) ? cconj(twh[k]) : make_float2(-twh[M - k].x, -twh[M - k].y); };
    // (z / M) * w == z * (w / M) exactly (M is a power of two), and the conjugate's sign rides along
    auto win_of = [&](int k) { return make_float2(win[2 * k] * inv_m, -(win[2 * k + 1] * inv_m)); };
    float2 tw1_r[7], tw2_r[7];                                // radix-8 pass twiddles of this lane (M == 512 only)
    if constexpr (M == 512) fft_lane_twiddles<M>(tw, lane, tw1_r, tw2_r);
    float2 wc_r[HOIST ? R : 1], win_r[HOIST ? R : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            wc_r[r] = wc_of(lane + WAVE * r);
            win_r[r] = win_of(lane + WAVE * r);
        }
    }

    // The smoothed-mask knots a hop needs (hop / MASK_DS of them, plus the +-3 the exact index search may reach) are
    // fetched lane-parallel a frame ahead and parked in LDS, so the output stage reads them with LDS latency instead of
    // issuing dependent global loads per sample.
    constexpr int KPL = (WAVE * SLOTS / MASK_DS + KNOT_MARGIN + WAVE - 1) / WAVE;
    const int KN = hop / MASK_DS + KNOT_MARGIN;
    double *kbuf = knots + (size_t)wave * KN;
    double kn_r[KPL];
    int kn_lo = 0;
    const bool slots_ok = hop <= WAVE * SLOTS;
    auto knots_fetch = [&](int h) {
        int i0 = h * hop - M;
        i0 = i0 < 0 ? 0 : i0;
        int lo = (int)((float)i0 * kps) - 4;
        lo = lo < 0 ? 0 : lo;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = lane + WAVE * c;
            const int k = lo + e < ns - 1 ? lo + e : ns - 1;
            kn_r[c] = (e < KN && ns > 0) ? ss[k] : 0.0;
        }
        kn_lo = lo;
    };
    auto knots_park = [&]() {
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = lane + WAVE * c;
            if (e < KN) kbuf[e] = kn_r[c];
        }
        wave_lds_sync();
    };

    // finished hop h of the current note -> gains -> stems; also used for the flush hops and the zero tail.
    // NS slots of 64 samples cover the hop; loads first, then the arithmetic, then the stores.
    auto emit = [&](auto ns_tag, int h) {
        constexpr int NS = decltype(ns_tag)::value;
        const int p0 = h * hop - M;                           // output index of the hop's first sample
        const int e_hi = KN - 1, lo = kn_lo;
        auto knot = [&](int k) {
            const int e = k - lo;
            return kbuf[e < e_hi ? e : e_hi];
        };
        float vh[NS], vu[NS], vb[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int q = (h * hop + lane + WAVE * u) & (NF - 1);
            vh[u] = ring[q];
            vu[u] = ring[NF + q];
            vb[u] = ring[2 * NF + q];
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int j = lane + WAVE * u;
            const int i = p0 + j;
            if (j >= hop || i < 0 || i >= n) continue;
            float xh = 0.f, xu = 0.f, xb = 0.f;
            if (i < out_len) {
                xh = vh[u]; xu = vu[u]; xb = vb[u];
                if (h >= max_back && h <= T - 1) {
                    // interior hop: the divisor is a per-lane constant; with r = RN(1 / ws) the quotient correction
                    // q + fma(-q, ws, x) * r is the correctly rounded x / ws (Markstein) in three instructions
                    const float ws = ws_u[u], rw = rws_u[u];
                    if (ws > 1e-9f) { xh = div_by(xh, ws, rw); xu = div_by(xu, ws, rw); xb = div_by(xb, ws, rw); }
                } else {
                    const int back = (NF - 1 - j) / hop;
                    const int flo = h - back < 0 ? 0 : h - back, fhi = h > T - 1 ? T - 1 : h;
                    float ws = 0.f;
                    for (int fr = flo; fr <= fhi; ++fr) {
                        const float w = win[j + (h - fr) * hop];
                        ws += w * w;
                    }
                    if (ws > 1e-9f) { xh /= ws; xu /= ws; xb /= ws; }
                }
                xh = div_by(xh, mag, rmag);
            }
            const float ms = smooth_mask_at32(knot, ns, i, n, step_n, step_s, kps);
            xb = (xb * ms) * g_b;
            xu = (xu * (1.0f - ms)) * g_u;
            harm[base + i] = xh;
            uv[base + i] = xu;
            bre[base + i] = xb;
            pk = fmaxf(pk, fabsf((xh + xu) + xb));
        }
    };
    auto emit_any = [&](int h) {                              // hops wider than the cached slots
        auto knot = [&](int k) { return ss[k]; };
        for (int j = lane; j < hop; j += WAVE) {
            const int i = h * hop + j - M;
            if (i < 0 || i >= n) continue;
            float vh = 0.f, vu = 0.f, vb = 0.f;
            if (i < out_len) {
                const int back = (NF - 1 - j) / hop;
                const int lo = h - back < 0 ? 0 : h - back, hi = h > T - 1 ? T - 1 : h;
                float ws = 0.f;
                for (int fr = lo; fr <= hi; ++fr) {
                    const float w = win[j + (h - fr) * hop];
                    ws += w * w;
                }
                const int q = (h * hop + j) & (NF - 1);
                vh = ring[q]; vu = ring[NF + q]; vb = ring[2 * NF + q];
                if (ws > 1e-9f) { vh /= ws; vu /= ws; vb /= ws; }
                vh = vh / mag;
            }
            const float ms = smooth_mask_at32(knot, ns, i, n, step_n, step_s, kps);
            vb = (vb * ms) * g_b;
            vu = (vu * (1.0f - ms)) * g_u;
            harm[base + i] = vh;
            uv[base + i] = vu;
            bre[base + i] = vb;
            pk = fmaxf(pk, fabsf((vh + vu) + vb));
        }
    };
    const int nslot = (hop + WAVE - 1) / WAVE;

    for (int64_t f = fs; f < f1; ++f) {
        const int nt = frame_note[f];
        if (nt != note) {
            if (note >= 0) {
                const float m = wave_max(pk);
                if (lane == 0) atomic_max_pos(note_peak + note, m);
            }
            note = nt;
            pk = 0.f;
            base = sample_off[note];
            n = (int)(sample_off[note + 1] - base);
            fbase = frame_off[note];
            T = (int)(frame_off[note + 1] - fbase);
            ns = (n + MASK_DS - 1) / MASK_DS;
            out_len = hop * (T - 1);
            mag = note_mag[note];
            rmag = 1.0f / mag;
            g_b = params[note].breath_strength;
            g_u = params[note].uv_strength;
            step_n = steps[2 * note];
            step_s = steps[2 * note + 1];
            kps = n > 1 ? (float)(ns - 1) / (float)(n - 1) : 0.f;
            ss = short_s + short_base(sample_off, note);
        }
        const int t = (int)(f - fbase);
        if (f >= f0 && slots_ok) knots_fetch(t);              // lands during the three transforms below
#pragma unroll
        for (int stem = 0; stem < 3; ++stem) {
            float2 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + WAVE * r;
                float2 xk = nk[r], xm = nm[r];
                if (k == 0) { xk.y = 0.f; xm.y = 0.f; }       // irfft ignores Im of DC and Nyquist
                v[r] = irfft_pre(xk, xm, HOIST ? wc_r[HOIST ? r : 0] : wc_of(k));
            }
            // next job's rows: same frame next stem, or the next frame's first stem
            if (stem < 2) fetch(f, stem + 1);
            else if (f + 1 < f1) fetch(f + 1, 0);
            float *rg = ring + stem * NF;
            const int shift = (t * hop) & (NF - 1);
            // all ring reads, then all ring writes: the R slots of a lane are distinct, and a branch-free body lets the
            // LDS reads overlap instead of paying one round trip per slot
            float2 z[R], o[R];
            if constexpr (M == 512) {
                wave_fft_keep_tw<M>(v, buf, tw1_r, tw2_r, lane, z);   // twiddles and the lane's output points in registers
            } else if constexpr (M >= 512) {
                wave_fft_keep<M>(v, buf, tw, lane, z);        // the lane's output points stay in registers
            } else {
                wave_fft<M>(v, buf, tw, lane);
#pragma unroll
                for (int r = 0; r < R; ++r) z[r] = buf[lds_pad(lane + WAVE * r)];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                o[r] = *reinterpret_cast<const float2 *>(rg + ((2 * m + shift) & (NF - 1)));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                const float2 wn = HOIST ? win_r[HOIST ? r : 0] : win_of(m);
                const float a = z[r].x * wn.x, b = z[r].y * wn.y;
                const bool first = t == 0 || 2 * m >= NF - hop;   // first contribution: y starts from zero
                *reinterpret_cast<float2 *>(rg + ((2 * m + shift) & (NF - 1))) =
                    make_float2(first ? a : o[r].x + a, first ? b : o[r].y + b);
            }
            wave_lds_sync();
        }
        if (f >= f0) {
            // hop t; behind a note's last frame also the hops up to the one holding sample out_len - 1 and the zero tail
            for (int h = t;;) {
                if (slots_ok) {
                    knots_park();
                    if (nslot <= 2) emit(std::integral_constant<int, 2>{}, h);
                    else if (nslot <= 4) emit(std::integral_constant<int, 4>{}, h);
                    else if constexpr (SLOTS >= 8) emit(std::integral_constant<int, 8>{}, h);
                } else {
                    emit_any(h);
                }
                ++h;
                if (t != T - 1 || h * hop - M >= n) break;
                if (slots_ok) knots_fetch(h);
            }
        }
        wave_lds_sync();
    }
    if (note >= 0) {
        const float m = wave_max(pk);
        if (lane == 0) atomic_max_pos(note_peak + note, m);
    }
}

// ---------------------------------------------------------------------------------------------
// The same pass with ONE stem per wave, for geometries whose three rings do not leave room for a second wave per SIMD
// (n_fft 2048: 24 KB of rings + 8 KB of exchange buffer per wave, 1 wave per SIMD, 0.17-0.22 of the HBM roofline).  A wave
// walks a run of frames of one stem — job = 3 * run + stem — with a single n_fft-float ring (16.6 KB per wave at 2048:
// eight waves per CU), and leaves its stem the way the stem walkers of stems.hip do: the harmonic stem divided by the
// window sum only (1 / max|S| is applied by k_note_finish, which also takes the peak), the noise stems with their
// mask gains.  Same transforms, same ascending-frame accumulation: bit-identical to k_irfft_ola3 + k_apply_gain (tested).
// Arguments needed once per note, once per 64 frames or at set-up are read from the kernarg segment where they are used
// (cold_arg, as in the stem walkers): the frame loop holds ~60 scalars of wave state, and with 22 arguments live beside them
// the compiler spilled 142 SGPRs (a v_writelane / v_readlane pair each) and a vector register to scratch.
struct ola1_args {
    // every frame
    const float2 *S_h, *S_u, *S_b;
    const int *frame_note;
    int ldc, hop, run, halo;
    // cold
    int64_t total_frames;
    const int64_t *frame_off, *sample_off;
    const double *short_s, *steps;
    const goofer_note_params *params;
    float *harm, *uv, *bre;
    const float2 *g_tw, *g_twh;
    const float *g_win;
    const unsigned char *frame_skip;
};
template <typename T>
__device__ __forceinline__ T ola1_cold(size_t offset)
{
    const char __attribute__((address_space(4))) *ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<const T __attribute__((address_space(4))) *>(ka + offset);
}
#define OCOLD(field) ola1_cold<decltype(ola1_args::field)>(offsetof(ola1_args, field))

template <int M, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void k_irfft_ola1(const ola1_args A)
{
    const int ldc = A.ldc, hop = A.hop, run = A.run, halo = A.halo;
    const int *__restrict__ frame_note = A.frame_note;
    constexpr int R = fft_cfg<M>::R, NF = 2 * M, BUF = fft_cfg<M>::BUF;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WPB * BUF);
    float *rings = win + NF;
    double *knots = reinterpret_cast<double *>(rings + (size_t)WPB * NF);
    load_tables<M>(tw, twh, win, OCOLD(g_tw), OCOLD(g_twh), OCOLD(g_win));

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *buf = bufs + wave * BUF;
    float *ring = rings + (size_t)wave * NF;
    // A workgroup holds runs of ONE stem (stems interleaved workgroup by workgroup): where a stem's transforms are skipped
    // (frame_skip: its gain is exactly zero over everything the frame reaches) whole workgroups retire early and the
    // dispatcher hands their CU to the next one — with the three stems of a run side by side in one workgroup the skipped
    // waves only idled beside the others.
    const int stem = (int)(blockIdx.x % 3);                   // workgroup-uniform
    const int64_t f0 = ((int64_t)(blockIdx.x / 3) * WPB + wave) * run;
    if (f0 >= OCOLD(total_frames)) return;                    // no block barrier below
    const unsigned skip_bit = OCOLD(frame_skip) ? (unsigned)stem : 0u;   // bit 0 (1): unvoiced stem, bit 1 (2): breath stem; harmonic: never
    // the skip bits of 64 consecutive frames as one ballot (a byte load per frame would sit on the loop's critical path)
    uint64_t skip_mask = 0;
    int64_t skip_base = -(int64_t)WAVE;
    auto skipped = [&](int64_t f) {
        if (skip_bit == 0u) return false;
        if (f >= skip_base + WAVE || f < skip_base) {
            skip_base = f;
            const int64_t g = f + lane;
            skip_mask = __ballot(g < OCOLD(total_frames) && (OCOLD(frame_skip)[g] & skip_bit) != 0);
        }
        return ((skip_mask >> (int)(f - skip_base)) & 1ull) != 0;
    };
    const int64_t f1 = f0 + run < OCOLD(total_frames) ? f0 + run : OCOLD(total_frames);
    const float inv_m = 0.5f / (float)M;                     // 1/M of the transform and the 1/2 of the input stage (irfft_pre)
    const float2 *S = stem == 0 ? A.S_h : (stem == 1 ? A.S_u : A.S_b);
    float *out = stem == 0 ? OCOLD(harm) : (stem == 1 ? OCOLD(uv) : OCOLD(bre));

    int64_t fs = f0;
    {
        const int nt = frame_note[f0];
        const int64_t t0 = f0 - OCOLD(frame_off)[nt];
        fs = f0 - (t0 < halo ? t0 : halo);
    }
    // spectrum row of the next frame, fetched during the transform: bins lane + 64 r and the Nyquist bin; the mirrored bins
    // X[M - k] of the inverse transform's input stage come from the row itself through the (idle) exchange buffer — a second
    // set of loads would hold 32 more registers across the whole frame
    float2 nk[R], ny;
    auto fetch = [&](int64_t f) {
        const float2 *row = S + f * (int64_t)ldc;
#pragma unroll
        for (int r = 0; r < R; ++r) nk[r] = row[lane + WAVE * r];
        ny = row[M];
    };
    bool skip_cur = skipped(fs);
    if (!skip_cur) fetch(fs);

    int note = -1;
    int64_t base = 0, fbase = 0;
    int n = 0, T = 0, ns = 0, out_len = 0;
    float gain = 0.f, kps = 0.f;
    double step_n = 0.0, step_s = 0.0;
    const double *ss = nullptr;
    const int max_back = (NF - 1) / hop;
    auto wc_of = [&](int k) { return (k <= M / 2) ? cconj(twh[k]) : make_float2(-twh[M - k].x, -twh[M - k].y); };
    auto win_of = [&](int k) { return make_float2(win[2 * k] * inv_m, -(win[2 * k + 1] * inv_m)); };

    // summed squared window of this lane's hop samples j = lane + 64 u for interior hops (every covering frame exists): a
    // constant per lane, with its correctly rounded reciprocal (div_by); the first two slots only — wider hops take the loop
    float ws_c[2] = {0.f, 0.f}, rws_c[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = lane + WAVE * u;
        if (j < hop) {
            float ws = 0.f;
            for (int q = (NF - 1 - j) / hop; q >= 0; --q) {   // ascending frame order = descending offset
                const float w = win[j + q * hop];
                ws += w * w;
            }
            ws_c[u] = ws;
            rws_c[u] = 1.0f / ws;
        }
    }
    const int KN = hop / MASK_DS + KNOT_MARGIN;
    double *kbuf = knots + (size_t)wave * (512 / MASK_DS + KNOT_MARGIN);
    constexpr int KPL = (512 / MASK_DS + KNOT_MARGIN + WAVE - 1) / WAVE;
    double kn_r[KPL];
    int kn_lo = 0;
    const bool slots_ok = hop <= 512 && stem != 0;            // the harmonic stem has no mask gain
    auto knots_fetch = [&](int h) {
        int i0 = h * hop - M;
        i0 = i0 < 0 ? 0 : i0;
        int lo = (int)((float)i0 * kps) - 4;
        lo = lo < 0 ? 0 : lo;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = lane + WAVE * c;
            const int k = lo + e < ns - 1 ? lo + e : ns - 1;
            kn_r[c] = (e < KN && ns > 0) ? ss[k] : 0.0;
        }
        kn_lo = lo;
    };
    // finished hop h of the current note -> window-sum quotient -> (noise stems) mask gain -> out; also the flush hops and the zero tail
    auto emit = [&](int h) {
        const int e_hi = KN - 1, lo = kn_lo;
        auto knot_l = [&](int k) {
            const int e = k - lo;
            return kbuf[e < e_hi ? e : e_hi];
        };
        auto knot_g = [&](int k) { return ss[k]; };
        const bool inner = h >= max_back && h <= T - 1;       // every covering frame exists
        for (int j = lane, u = 0; j < hop; j += WAVE, ++u) {
            const int i = h * hop + j - M;
            if (i < 0 || i >= n) continue;
            float x = 0.f;
            if (i < out_len) {
                x = ring[(h * hop + j) & (NF - 1)];
                if (inner && u < 2) {
                    const float ws = u == 0 ? ws_c[0] : ws_c[1], rw = u == 0 ? rws_c[0] : rws_c[1];
                    if (ws > 1e-9f) x = div_by(x, ws, rw);
                } else {
                    const int back = (NF - 1 - j) / hop;
                    const int flo = h - back < 0 ? 0 : h - back, fhi = h > T - 1 ? T - 1 : h;
                    float ws = 0.f;
                    for (int fr = flo; fr <= fhi; ++fr) {
                        const float w = win[j + (h - fr) * hop];
                        ws += w * w;
                    }
                    if (ws > 1e-9f) x /= ws;
                }
            }
            if (stem != 0) {
                const float ms = slots_ok ? smooth_mask_at32(knot_l, ns, i, n, step_n, step_s, kps)
                                          : smooth_mask_at32(knot_g, ns, i, n, step_n, step_s, kps);
                x = (x * (stem == 1 ? 1.0f - ms : ms)) * gain;
            }
            out[base + i] = x;
        }
    };

    for (int64_t f = fs; f < f1; ++f) {
        const int nt = frame_note[f];
        if (nt != note) {
            note = nt;
            const int64_t *sample_off = OCOLD(sample_off), *frame_off = OCOLD(frame_off);
            base = sample_off[note];
            n = (int)(sample_off[note + 1] - base);
            fbase = frame_off[note];
            T = (int)(frame_off[note + 1] - fbase);
            ns = (n + MASK_DS - 1) / MASK_DS;
            out_len = hop * (T - 1);
            const goofer_note_params *params = OCOLD(params);
            const double *steps = OCOLD(steps);
            gain = stem == 1 ? params[note].uv_strength : params[note].breath_strength;
            step_n = steps[2 * note];
            step_s = steps[2 * note + 1];
            kps = n > 1 ? (float)(ns - 1) / (float)(n - 1) : 0.f;
            ss = OCOLD(short_s) + short_base(sample_off, note);
        }
        const int t = (int)(f - fbase);
        const int shift = (t * hop) & (NF - 1);
        const bool skip_this = skip_cur;
        if (f >= f0 && slots_ok && !skip_this) knots_fetch(t);
        skip_cur = f + 1 < f1 && skipped(f + 1);
        if (skip_this) {
            // no transform: the slots this frame would have started from zero are zeroed, the others keep their sums
            if (f + 1 < f1 && !skip_cur) fetch(f + 1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                if (t == 0 || 2 * m >= NF - hop) *reinterpret_cast<float2 *>(ring + ((2 * m + shift) & (NF - 1))) = make_float2(0.f, 0.f);
            }
        } else {
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) buf[lane + WAVE * r] = nk[r];
        if (lane == 0) buf[M] = ny;
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = lane + WAVE * r;
            float2 xk = nk[r], xm = buf[M - k];
            if (k == 0) { xk.y = 0.f; xm.y = 0.f; }           // irfft ignores Im of DC and Nyquist
            v[r] = irfft_pre(xk, xm, wc_of(k));
        }
        wave_lds_sync();                                      // the row is read before the transform reuses buf
        if (f + 1 < f1 && !skip_cur) fetch(f + 1);
        float2 z[R];
        wave_fft_keep<M>(v, buf, tw, lane, z);                // the lane's output points stay in registers
        // ring reads ahead of ring writes, eight slots at a time (the R slots of a lane are distinct)
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 8) {
            float2 o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = lane + WAVE * (r0 + q);
                o[q] = *reinterpret_cast<const float2 *>(ring + ((2 * m + shift) & (NF - 1)));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = lane + WAVE * (r0 + q);
                const float2 wn = win_of(m);
                const float a = z[r0 + q].x * wn.x, b = z[r0 + q].y * wn.y;
                const bool first = t == 0 || 2 * m >= NF - hop;   // first contribution: y starts from zero
                *reinterpret_cast<float2 *>(ring + ((2 * m + shift) & (NF - 1))) = make_float2(first ? a : o[q].x + a, first ? b : o[q].y + b);
            }
        }
        }
        wave_lds_sync();
        if (f >= f0) {
            for (int h = t;;) {
                if (slots_ok && !skip_this) {
#pragma unroll
                    for (int c = 0; c < KPL; ++c) {
                        const int e = lane + WAVE * c;
                        if (e < KN) kbuf[e] = kn_r[c];
                    }
                    wave_lds_sync();
                }
                if (skip_this) {
                    // every hop this frame reaches has a gain of exactly zero: the hop (and the flush hops behind a last frame) is zeros
                    for (int j = lane; j < hop; j += WAVE) {
                        const int i = h * hop + j - M;
                        if (i >= 0 && i < n) out[base + i] = 0.f;
                    }
                } else {
                    emit(h);
                }
                ++h;
                if (t != T - 1 || h * hop - M >= n) break;
                if (slots_ok && !skip_this) knots_fetch(h);
            }
        }
        wave_lds_sync();
    }
}

template <int M>
static int irfft_ola1_impl(goofer_ctx *ctx, const float2 *S_h, const float2 *S_u, const float2 *S_b, int ldc, int64_t total_frames,
                           const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, const double *short_s,
                           const double *steps, const goofer_note_params *params, float *harm, float *uv, float *bre,
                           const unsigned char *frame_skip, hipStream_t st)
{
    constexpr int WPB = 8;
    const goofer_plan_t &p = ctx->plan;
    const int halo = (p.n_fft + p.hop - 1) / p.hop - 1;
    const size_t lds = sizeof(float2) * (M + M / 2 + 1 + WPB * fft_cfg<M>::BUF) + sizeof(float) * 2 * M + sizeof(float) * WPB * 2 * M + 16 +
                       sizeof(double) * WPB * (512 / MASK_DS + KNOT_MARGIN);
    if (lds > 160 * 1024) return goofer_fail(ctx, GOOFER_EINVAL, "one-stem overlap-add: %zu bytes of LDS", lds);
    const void *fn = (const void *)k_irfft_ola1<M, WPB>;
    int rc = kernel_allow_max_lds(ctx, fn);
    if (rc) return rc;
    int cus = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const int64_t slots = (int64_t)(cus > 0 ? cus : 256) * WPB;          // one workgroup per CU
    // jobs = 3 stems x runs; runs sized so that the jobs fill the device a whole number of times, a run >= 8 halos
    const int min_run = halo <= 4 ? 32 : 8 * halo;
    const int64_t rounds = (3 * total_frames + slots * 256 - 1) / (slots * 256);
    int64_t fit = (3 * total_frames + rounds * slots - 1) / (rounds * slots);
    const int run = (int)(fit > min_run ? fit : min_run);
    const int64_t runs = (total_frames + run - 1) / run;
    ola1_args A;
    A.S_h = S_h; A.S_u = S_u; A.S_b = S_b; A.frame_note = frame_note; A.ldc = ldc; A.hop = p.hop; A.run = run; A.halo = halo;
    A.total_frames = total_frames; A.frame_off = frame_off; A.sample_off = sample_off; A.short_s = short_s; A.steps = steps;
    A.params = params; A.harm = harm; A.uv = uv; A.bre = bre; A.g_tw = p.tw_full; A.g_twh = p.tw_half; A.g_win = p.window;
    A.frame_skip = frame_skip;
    hipLaunchKernelGGL((k_irfft_ola1<M, WPB>), dim3((unsigned)(3 * ((runs + WPB - 1) / WPB))), dim3(64 * WPB), lds, st, A);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// one stem per wave (n_fft 2048); needs k_note_finish behind it (1 / max|S|, peak, gain)
bool ola_split_supported(const goofer_plan_t &p) { return p.n_fft == 2048 && (p.hop & 1) == 0; }

int launch_irfft_ola1(goofer_ctx *ctx, const float2 *S_h, const float2 *S_u, const float2 *S_b, int ldc, int64_t total_frames,
                      const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, int n_notes, const double *short_s,
                      double *steps, const goofer_note_params *params, float *harm, float *uv, float *bre,
                      const unsigned char *frame_skip, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    if (!ola_split_supported(ctx->plan)) return goofer_fail(ctx, GOOFER_EINVAL, "one-stem overlap-add is built for n_fft 2048");
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    return irfft_ola1_impl<1024>(ctx, S_h, S_u, S_b, ldc, total_frames, frame_note, frame_off, sample_off, short_s, steps, params,
                                 harm, uv, bre, frame_skip, st);
}

// ---------------------------------------------------------------------------------------------
// Exact sparsity of the noise stems for the LDS-ring pipeline (the walkers of stems.hip decide the same thing per hop inside
// their frame loop): where the smoothed mask is flat at (float) 1 over every knot a hop's samples can read, the unvoiced stem
// of that hop is multiplied by exactly 0 (GOOFER.py:1181), where it is flat at 0 the breath stem is (:1180).  A frame whose
// every hop is such a hop cannot reach a non-zero sample of that stem: its spectrum is neither written nor transformed.
// hop_flat[frame_off[note] + reach * note + h], h < T + reach: bit 0 flat at one, bit 1 flat at zero (hops without an
// output sample count as both); the knot window is the one k_irfft_ola1 stages for the hop (knots_fetch).
// eq[k] = (short_s[k] == short_s[k + 1]) over the whole smoothed-mask array: a hop's knots are one constant iff the eq bytes of
// its window are all set, and a hop then reads 35 bytes instead of 36 doubles (its neighbours' windows overlap it by a third:
// read as doubles, thread by thread, they cost eight times the unique bytes)
__global__ void k_knot_eq(const double *__restrict__ short_s, int64_t count, unsigned char *__restrict__ eq)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) eq[k] = k + 1 < count && short_s[k] == short_s[k + 1] ? 1 : 0;
}

__global__ void k_hop_flat(const double *__restrict__ short_s, const unsigned char *__restrict__ eq, const int64_t *__restrict__ sample_off,
                           const int64_t *__restrict__ frame_off, int n_notes, int hop, int M, int reach, unsigned char *__restrict__ hop_flat)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= frame_off[n_notes] + (int64_t)reach * n_notes) return;
    int lo = 0, hi = n_notes;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] + (int64_t)reach * mid <= g) lo = mid; else hi = mid;
    }
    const int note = lo;
    const int h = (int)(g - (frame_off[note] + (int64_t)reach * note));
    const int n = (int)(sample_off[note + 1] - sample_off[note]);
    const int ns = (n + MASK_DS - 1) / MASK_DS;
    const int i_lo = h * hop - M, i_hi = i_lo + hop - 1;
    unsigned flags = 3u;
    if (!(i_hi < 0 || i_lo >= n || ns <= 0)) {
        const int64_t sb = short_base(sample_off, note);
        const double *ss = short_s + sb;
        const unsigned char *eqn = eq + sb;
        const float kps = n > 1 ? (float)(ns - 1) / (float)(n - 1) : 0.f;
        int k0 = (int)((float)(i_lo < 0 ? 0 : i_lo) * kps) - 4;
        k0 = k0 < 0 ? 0 : (k0 > ns - 1 ? ns - 1 : k0);
        int k1 = k0 + hop / MASK_DS + KNOT_MARGIN - 1;
        k1 = k1 > ns - 1 ? ns - 1 : k1;
        const double c = ss[k0];
        bool same = true;
        for (int k = k0; k < k1; ++k) same = same && eqn[k] != 0;      // ss[k0] == ss[k0 + 1] == ... == ss[k1]
        const float cf = (float)c;
        flags = same ? ((1.0f - cf == 0.0f ? 1u : 0u) | (cf == 0.0f ? 2u : 0u)) : 0u;
    }
    hop_flat[g] = (unsigned char)flags;
}

__global__ void k_frame_skip(const unsigned char *__restrict__ hop_flat, const int *__restrict__ frame_note, const int64_t *__restrict__ frame_off,
                             int64_t total_frames, int reach, unsigned char *__restrict__ frame_skip)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total_frames) return;
    const int note = frame_note[f];
    const unsigned char *hf = hop_flat + f + (int64_t)reach * note;      // (= frame_off[note] + reach * note + t)
    unsigned acc = 3u;
    for (int q = 0; q < reach; ++q) acc &= hf[q];
    frame_skip[f] = (unsigned char)acc;
}

int launch_frame_skip(goofer_ctx *ctx, const double *short_s, int64_t short_count, const int64_t *sample_off, const int64_t *frame_off,
                      const int *frame_note, int n_notes, int64_t total_frames, unsigned char *knot_eq, unsigned char *hop_flat,
                      unsigned char *frame_skip, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    const int reach = (p.n_fft + p.hop - 1) / p.hop;
    const int64_t hops = total_frames + (int64_t)reach * n_notes;
    hipLaunchKernelGGL(k_knot_eq, dim3((unsigned)((short_count + 255) / 256)), dim3(256), 0, st, short_s, short_count, knot_eq);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_hop_flat, dim3((unsigned)((hops + 255) / 256)), dim3(256), 0, st, short_s, knot_eq, sample_off, frame_off, n_notes, p.hop,
                       p.n_fft / 2, reach, hop_flat);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_frame_skip, dim3((unsigned)((total_frames + 255) / 256)), dim3(256), 0, st, hop_flat, frame_note, frame_off,
                       total_frames, reach, frame_skip);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

template <int M>
static int irfft_ola3_impl(goofer_ctx *ctx, const float2 *S_h, const float2 *S_u, const float2 *S_b, int ldc, int64_t total_frames,
                           const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, const float *note_mag,
                           const double *short_s, const double *steps, const goofer_note_params *params, float *harm, float *uv,
                           float *bre, float *note_peak, hipStream_t st)
{
    const goofer_plan_t &p = ctx->plan;
    const int halo = (p.n_fft + p.hop - 1) / p.hop - 1;
    const size_t lds = sizeof(float2) * (M + M / 2 + 1 + WAVES_PER_BLOCK * fft_cfg<M>::BUF) + sizeof(float) * 2 * M +
                       sizeof(float) * WAVES_PER_BLOCK * 3 * 2 * M + 16 +
                       (p.hop <= (M >= 1024 ? 512 : 256) ? sizeof(double) * WAVES_PER_BLOCK * (p.hop / MASK_DS + KNOT_MARGIN) : 0);
    int rc = kernel_allow_max_lds(ctx, (const void *)k_irfft_ola3<M>);
    if (rc) return rc;
    int slots = 0;                                            // waves of this kernel the device holds at once
    if ((rc = kernel_resident_waves(ctx, (const void *)k_irfft_ola3<M>, lds, &slots))) return rc;
    // Frames per wave.  A run replays `halo` frames it does not emit, so longer runs waste less; and every wave does the
    // same work, so the grid should fill the device a whole number of times: with k rounds of `slots` waves,
    // run = ceil(frames / (k slots)), k the smallest that keeps a run at or below 128 frames (0.92 -> 0.85 ms on the
    // 1024-note batch: 95 frames per wave in one round instead of 32 in three).
    const int min_run = halo <= 4 ? 32 : 8 * halo;
    const int64_t rounds = (total_frames + (int64_t)slots * 128 - 1) / ((int64_t)slots * 128);
    int64_t fit = (total_frames + rounds * slots - 1) / (rounds * slots);
    const int run = (int)(fit > min_run ? fit : min_run);
    const int64_t runs = (total_frames + run - 1) / run;
    hipLaunchKernelGGL(k_irfft_ola3<M>, dim3((unsigned)((runs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), dim3(256), lds, st, S_h, S_u,
                       S_b, ldc, total_frames, frame_note, frame_off, sample_off, p.hop, run, halo, note_mag, short_s, steps, params,
                       harm, uv, bre, note_peak, p.tw_full, p.tw_half, p.window);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// smooth_mask_ds's last step on its own (GOOFER.py:563-567): np.interp of the smoothed decimated mask back to the sample
// grid.  `fast` selects the walkers' 32-bit interpolant (smooth_mask_at32: flat-knot shortcut, one-step index fix-up)
// instead of the search-loop form the separate stem-gain kernels use; the two must agree bit for bit.
__global__ __launch_bounds__(256) void k_mask_upsample(const double *__restrict__ short_s, const int64_t *__restrict__ sample_off,
                                                       int n_notes, int64_t total_samples, const double *__restrict__ steps, int fast,
                                                       float *__restrict__ out)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total_samples, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total_samples) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
    const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
    const double *ss = short_s + short_base(sample_off, note);
    if (fast) {
        const float kps = n > 1 ? (float)(ns - 1) / (float)(n - 1) : 0.f;
        auto knot = [&](int k) { return ss[k]; };
        out[g] = smooth_mask_at32(knot, (int)ns, (int)(g - base), (int)n, steps[2 * note], steps[2 * note + 1], kps);
    } else {
        out[g] = smooth_mask_at(ss, ns, g - base, n, steps[2 * note], steps[2 * note + 1]);
    }
}

int launch_mask_upsample(goofer_ctx *ctx, const double *short_s, const int64_t *sample_off, int n_notes, int64_t total_samples,
                         double *steps, bool fast, float *out, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_mask_upsample, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, short_s, sample_off, n_notes,
                       total_samples, steps, fast ? 1 : 0, out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_note_steps(goofer_ctx *ctx, const int64_t *sample_off, int n_notes, double *steps, hipStream_t st)
{
    if (n_notes <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_irfft_ola3(goofer_ctx *ctx, const float2 *S_h, const float2 *S_u, const float2 *S_b, int ldc, int64_t total_frames,
                      const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, int n_notes, const float *note_mag,
                      const double *short_s, double *steps, const goofer_note_params *params, float *harm, float *uv, float *bre,
                      float *note_peak, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    if (p.hop & 1) return goofer_fail(ctx, GOOFER_EINVAL, "the fused overlap-add needs an even hop");
    hipLaunchKernelGGL(k_note_steps, dim3((n_notes + 255) / 256), dim3(256), 0, st, sample_off, n_notes, steps);
    LAUNCH_CHECK(ctx);
    switch (p.n_fft) {
    case 512: return irfft_ola3_impl<256>(ctx, S_h, S_u, S_b, ldc, total_frames, frame_note, frame_off, sample_off, note_mag, short_s, steps, params, harm, uv, bre, note_peak, st);
    case 1024: return irfft_ola3_impl<512>(ctx, S_h, S_u, S_b, ldc, total_frames, frame_note, frame_off, sample_off, note_mag, short_s, steps, params, harm, uv, bre, note_peak, st);
    case 2048: return irfft_ola3_impl<1024>(ctx, S_h, S_u, S_b, ldc, total_frames, frame_note, frame_off, sample_off, note_mag, short_s, steps, params, harm, uv, bre, note_peak, st);
    }
    return goofer_fail(ctx, GOOFER_EINVAL, "unsupported n_fft %d", p.n_fft);
}

__global__ __launch_bounds__(256) void k_apply_gain(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                    float *__restrict__ rec, float *__restrict__ mix,
                                                    const int64_t *__restrict__ sample_off, int n_notes, int64_t total_samples,
                                                    const goofer_note_params *__restrict__ params, const float *__restrict__ note_peak,
                                                    int write_stems)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x * SPT);
    int lo, hi;
    {
        int64_t gl = g0 + (int64_t)blockDim.x * SPT - 1;
        if (gl > total_samples - 1) gl = total_samples - 1;
        block_note_range_last(sample_off, n_notes, g0, gl, s_pair, lo, hi);
    }
    const int64_t g = g0 + (int64_t)threadIdx.x * SPT;
    if (g >= total_samples) return;

    auto gain_of = [&](int note) -> float {
        const float peak = note_peak[note] + 1e-12f;                 // fp32 add, like np.float32 + 1e-12
        return peak_gain(peak, params[note].normalize);
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
        if (write_stems) {
            *reinterpret_cast<float4 *>(harm + g) = h;
            *reinterpret_cast<float4 *>(uv + g) = u;
            *reinterpret_cast<float4 *>(bre + g) = b;
        }
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
        if (write_stems) { harm[gi] = h; uv[gi] = u; bre[gi] = b; }
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
                      const double *d_taps, int radius, double tap_sum, double *short_s, hipStream_t st)
{
    int64_t total_short = total_samples / MASK_DS + n_notes;
    if (total_short <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_mask_short, dim3((unsigned)((total_short + MS_TILE - 1) / MS_TILE)), dim3(256),
                       sizeof(double) * (2 * radius + 1) + sizeof(float) * 4 * MS_MAXWIN, st, mask, sample_off, n_notes, total_short,
                       d_taps, radius, short_s, tap_sum);
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
                      int n_notes, int64_t total_samples, const goofer_note_params *params, const float *note_peak, bool write_stems,
                      hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_apply_gain, dim3((unsigned)((total_samples + 1023) / 1024)), dim3(256), 0, st, harm, uv, bre, rec, mix,
                       sample_off, n_notes, total_samples, params, note_peak, write_stems ? 1 : 0);
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
