// Stem walkers with the overlap-add in an LDS ring: n_fft 2048 with any even hop (gfx950).
//
// stems.hip keeps the open sums of the overlap-add in registers, which needs hop == n_fft / 4.  BASELINE config 5 is
// n_fft 2048 / hop 96 at 96 kHz — 21 frames over every sample — and ran the one-kernel-per-reference-step pipeline: three
// [frames x 1025] complex spectra through HBM between the framewise rFFT, the two shaping kernels and the inverse
// transforms (25 KB per frame and stem written and read back: 70 of the step's 87 GB).  Here a wave walks a run of frames
// of ONE stem (k_irfft_ola1's layout: a 2048-float ring and an exchange buffer per wave, eight waves per CU) and makes
// the frame's spectrum itself:
//
//   harmonic stem   stft(pulse) frame -> high-pass -> per-note max|S| -> envelope warps, * env * boost, brightness + 5-tap
//                   blur -> irfft -> ring                                         GOOFER.py:1099-1146 (+ :840-875, 618-627)
//   noise stems     sigma-1.75 blur of the envelope row -> random-phase spectrum (unvoiced), * high-pass, brightness + blur
//                   (breath) -> irfft -> ring -> mask gains                      GOOFER.py:993, 1148-1183
//
// No spectrum reaches HBM.  Every expression is the one k_rfft_frames, k_harm_shape, k_noise_spectra and k_irfft_ola1
// evaluate, in the same order, so the stems are bit-identical to that path (tested; option "ring_walkers" 0 runs it).
// k_note_finish follows (1 / max|S|, peak, gain, mix).
#include <type_traits>
#include <utility>

#include "binops_core.h"
#include "fft_core.h"
#include "samples_core.h"
#include "stems_core.h"

// blur5 (binops_core.h) for bin k = lane + 64 i: the same five FMAs; rows 1 .. R - 2 of a lane's bins are interior for every
// lane, so the reflect variant is compiled for three of the seventeen bins only (the frame loop has to fit the instruction
// cache: with both variants for every bin it was 60 KB of code and ran four times slower).
template <int I, int R>
__device__ __forceinline__ float2 blur5_bin(const float2 *r, int k, int n_bins, const double *t5)
{
    if constexpr (I >= 1 && I <= R - 2) {
        const float t0 = (float)t5[0], t1 = (float)t5[1], t2 = (float)t5[2], t3 = (float)t5[3], t4 = (float)t5[4];
        const float2 v0 = r[k - 2], v1 = r[k - 1], v2 = r[k], v3 = r[k + 1], v4 = r[k + 2];
        float re = t0 * v0.x, im = t0 * v0.y;
        re = fmaf(t1, v1.x, re); im = fmaf(t1, v1.y, im);
        re = fmaf(t2, v2.x, re); im = fmaf(t2, v2.y, im);
        re = fmaf(t3, v3.x, re); im = fmaf(t3, v3.y, im);
        re = fmaf(t4, v4.x, re); im = fmaf(t4, v4.y, im);
        return make_float2(re, im);
    } else {
        return blur5(r, k, n_bins, t5);
    }
}
template <int R, int... Is>
__device__ __forceinline__ void blur5_row(float2 (&X)[R + 1], const float2 *r, int lane, int n_bins, const double *t5, std::integer_sequence<int, Is...>)
{
    ((lane + WAVE * Is < n_bins ? (void)(X[Is] = blur5_bin<Is, R>(r, lane + WAVE * Is, n_bins, t5)) : (void)0), ...);
}

// ---------------------------------------------------------------------------------------------
// The harmonic stem's spectrum of one frame, in registers: window, forward transform, even/odd split (k_rfft_frames), then
// high-pass, max|S|, envelope (warped here when the note warps), boost, brightness and the 5-tap blur (k_harm_shape) — the
// same expressions in the same order as those two kernels.  Shared by the ring walker and by k_rfft_shape.
struct harm_note {
    int cut_below = 0;
    bool warp_f = false, warp_row_on = false;   // some f_shift != 1; the row is warped in the kernel (f_shift with formants, or formant_shift)
    double fsh[4] = {1.0, 1.0, 1.0, 1.0}, ratio = 1.0;
    __device__ __forceinline__ void load(const goofer_note_params &p, bool have_formants, bool have_row_src)
    {
        cut_below = p.cut_below_f0;
        warp_f = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fsh[k] = p.f_shift[k];
            warp_f |= fsh[k] != 1.0;
        }
        ratio = (double)p.formant_shift;
        warp_row_on = have_row_src && ((warp_f && have_formants) || p.formant_shift != 1.0f);
    }
};

// raw sample pairs m = lane + 64 r of the frame that starts at sample `start` of a note of n samples (numpy 'reflect' padding at
// the note's ends, GOOFER.py:358-360).  Frames over an end take a rolled loop through the exchange buffer: unrolled, the index
// maps are a tenth of a walker's loop for a frame in fifty.
template <int M>
__device__ __forceinline__ void fetch_frame_pairs(float2 (&v)[M / 64], const float *xs, int start, int n, float2 *buf, int lane)
{
    constexpr int R = M / 64;
    if (start >= 0 && start + 2 * M <= n) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = lane + WAVE * r;
            v[r] = make_float2(xs[start + 2 * m], xs[start + 2 * m + 1]);
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            const int m = lane + WAVE * r;
            const float a = n > 0 ? xs[reflect_index(start + 2 * m, n)] : 0.f;
            const float b = n > 0 ? xs[reflect_index(start + 2 * m + 1, n)] : 0.f;
            buf[m] = make_float2(a, b);
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = buf[lane + WAVE * r];
        wave_lds_sync();
    }
}

// K: where the cold kernel arguments come from (freqs, boost, bright_h, taps5, formants, note_mag)
template <int M, typename K>
__device__ __forceinline__ void harm_spectrum(float2 (&X)[M / 64 + 1], float2 (&v)[M / 64], const float (&ev)[M / 64 + 1], float2 *buf,
                                              const float2 *tw, const float2 *twh, const float *win, int lane, const harm_note &hn,
                                              bool voiced, float f0f, int src, const warp_grid &wg, double *seg, int note)
{
    constexpr int R = M / 64, B = M + 1, PER = R + 1, ROWF = (B + 1) & ~1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = lane + WAVE * r;
        v[r] = make_float2(v[r].x * win[2 * m], v[r].y * win[2 * m + 1]);
    }
    wave_fft<M>(v, buf, tw, lane);
    // even/odd split: X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k])
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = lane + WAVE * r;
        const float2 zk = buf[lds_pad(k)];
        const float2 zm = buf[lds_pad(k == 0 ? 0 : M - k)];
        const float2 w = (k <= M / 2) ? twh[k] : make_float2(-twh[M - k].x, twh[M - k].y);
        const float2 Aa = make_float2(zk.x + zm.x, zk.y - zm.y);
        const float2 Bb = make_float2(zk.x - zm.x, zk.y + zm.y);
        const float2 Cc = cmul(w, Bb);
        X[r] = make_float2(0.5f * (Aa.x + Cc.y), 0.5f * (Aa.y - Cc.x));
    }
    {
        const float2 z0 = buf[0];
        X[R] = make_float2(z0.x - z0.y, 0.f);
    }
    wave_lds_sync();                              // the transform is read: buf is free
    // envelope gain of the frame: the row, or its formant-anchored + uniform warp (GOOFER.py:1004-1017)
    float gv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) gv[i] = ev[i];
    if (hn.warp_row_on) {
        float *ra = reinterpret_cast<float *>(buf), *rb = ra + ROWF;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            if (k < B) ra[k] = ev[i];
        }
        wave_lds_sync();
        const double *fm = K::formants();
        const float *eg = warp_row(ra, rb, B, wg, fm ? fm + (int64_t)src * 4 : nullptr, hn.fsh, hn.warp_f, hn.ratio, lane, seg);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            gv[i] = eg[k < B ? k : B - 1];
        }
        wave_lds_sync();                          // the warped row is read before the blur row overwrites it
    }
    const float *freqs = K::freqs(), *boost = K::boost(), *bright = K::bright_h();
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = lane + WAVE * i;
        if (k < B) {
            float2 s = X[i];
            if (hn.cut_below) {
                const float h = hp_mask(freqs[k], f0f);
                s.x *= h; s.y *= h;
            }
            mx = fmaxf(mx, s.x * s.x + s.y * s.y);     // |s|^2: the square root is taken once, of the maximum
            const float g = gv[i], bo = boost[k];
            s.x = (s.x * g) * bo;
            s.y = (s.y * g) * bo;
            if (voiced) {
                const float br = bright[k];
                s.x *= br; s.y *= br;
                buf[k] = s;
            }
            X[i] = s;
        }
    }
    mx = __builtin_amdgcn_sqrtf(wave_max(mx)) + 1e-8f;  // max(|s| + 1e-8) = sqrt(max |s|^2) + 1e-8: sqrt is monotone
    if (lane == 0) atomic_max_pos(K::note_mag() + note, mx);
    if (voiced) {
        const double *t5 = K::taps5();
        wave_lds_sync();
        blur5_row<R>(X, buf, lane, B, t5, std::make_integer_sequence<int, PER>{});
        wave_lds_sync();
    }
}

struct ring_args {
    // every frame
    const float *pulse, *env_h, *env_n, *phi;
    float *out[3];                      // harm, uv, bre
    int ld, hop, run, halo, stem_first, stem_count;
    // every 64 frames, once per note, at set-up: read where they are used (cold_arg)
    int64_t total_frames;
    uint64_t seed;
    const int *frame_note;
    const int64_t *frame_off, *sample_off, *row_src;
    const float2 *picks;
    const double *short_s, *steps, *formants, *taps5, *taps175;
    const goofer_note_params *params;
    const float *freqs, *boost, *bright_h, *bright_b;
    float *note_mag;
    const float2 *g_tw, *g_twh;
    const float *g_win;
    const unsigned char *frame_skip;
    double grid_nyq, grid_step, grid_inv;   // warp_grid
};
#define RCOLD(field) COLD(ring_args, field)
struct ring_cold {
    static __device__ __forceinline__ const double *formants() { return RCOLD(formants); }
    static __device__ __forceinline__ const double *taps5() { return RCOLD(taps5); }
    static __device__ __forceinline__ const float *freqs() { return RCOLD(freqs); }
    static __device__ __forceinline__ const float *boost() { return RCOLD(boost); }
    static __device__ __forceinline__ const float *bright_h() { return RCOLD(bright_h); }
    static __device__ __forceinline__ float *note_mag() { return RCOLD(note_mag); }
};

// LDS of a workgroup of WPB waves: FFT twiddles, half-bin twiddles, the window, and per wave the exchange buffer (which
// also stages the envelope row, the warp's two rows and the complex row of the 5-tap blur, one after the other), the ring,
// the mask knots of a hop and the warp's anchor table
template <int M, int WPB> static size_t ring_lds_bytes(int hop)
{
    const int kns = ((hop < 512 ? hop : 512) / MASK_DS + KNOT_MARGIN + 1) & ~1;
    return sizeof(float2) * (M + M / 2 + 1 + WPB * fft_cfg<M>::BUF) + sizeof(float) * 2 * M + sizeof(float) * WPB * 2 * M + 16 +
           sizeof(double) * WPB * (kns + WARP_SEG_DOUBLES);
}

// HARM: the workgroups of this launch are the harmonic stem's (stem 0), else the noise stems' (1 unvoiced, 2 breath)
template <int M, int WPB, bool HARM, bool PHI>
__global__ __launch_bounds__(64 * WPB, 2) void k_stem_ring(const ring_args A)
{
    constexpr int R = fft_cfg<M>::R, NF = 2 * M, BUF = fft_cfg<M>::BUF, B = M + 1, PER = R + 1, ROWF = (B + 1) & ~1;
    static_assert(2 * ROWF * sizeof(float) <= BUF * sizeof(float2) && B <= BUF, "the exchange buffer stages the rows");
    const int ld = A.ld, hop = A.hop, run = A.run, halo = A.halo;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WPB * BUF);
    float *rings = win + NF;
    const int kns = ((hop < 512 ? hop : 512) / MASK_DS + KNOT_MARGIN + 1) & ~1;
    double *knots = reinterpret_cast<double *>(rings + (size_t)WPB * NF + 4);
    double *segs = knots + (size_t)WPB * kns;
    load_tables<M>(tw, twh, win, RCOLD(g_tw), RCOLD(g_twh), RCOLD(g_win));

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;                              // (not const: see the frame loop)
    float2 *buf = bufs + wave * BUF;
    float *ring = rings + (size_t)wave * NF;
    double *kbuf = knots + (size_t)wave * kns;
    double *seg = segs + (size_t)wave * WARP_SEG_DOUBLES;
    // a workgroup holds runs of ONE stem (see k_irfft_ola1): 0 harmonic, 1 unvoiced, 2 breath
    const int stem = HARM ? 0 : A.stem_first + (int)(blockIdx.x % (unsigned)A.stem_count);
    const int64_t f0 = ((int64_t)(blockIdx.x / (unsigned)A.stem_count) * WPB + wave) * run;
    if (f0 >= RCOLD(total_frames)) return;                    // no block barrier below
    const unsigned skip_bit = (RCOLD(frame_skip) && !HARM) ? (unsigned)stem : 0u;   // bit 0: unvoiced, bit 1: breath transform skipped
    uint64_t skip_mask = 0;
    int64_t skip_base = -(int64_t)WAVE;
    auto skipped = [&](int64_t f) {
        if (skip_bit == 0u) return false;
        if (f >= skip_base + WAVE || f < skip_base) {
            skip_base = f;
            const int64_t g = f + lane;
            skip_mask = __ballot(g < RCOLD(total_frames) && (RCOLD(frame_skip)[g] & skip_bit) != 0);
        }
        return ((skip_mask >> (int)(f - skip_base)) & 1ull) != 0;
    };
    const int64_t f1 = f0 + run < RCOLD(total_frames) ? f0 + run : RCOLD(total_frames);
    const float inv_m = 0.5f / (float)M;                     // 1/M of the transform and the 1/2 of the input stage (irfft_pre)
    float *out = A.out[stem];
    const float *pulse = A.pulse, *env = HARM ? A.env_h : A.env_n, *phi = A.phi;

    int64_t fs = f0;
    {
        const int nt = RCOLD(frame_note)[f0];
        const int64_t t0 = f0 - RCOLD(frame_off)[nt];
        fs = f0 - (t0 < halo ? t0 : halo);
    }
    frame_block fb;
    auto load_block = [&](int64_t first) {
        fb.load(first, RCOLD(total_frames), RCOLD(frame_note), RCOLD(frame_off), RCOLD(sample_off), RCOLD(row_src), RCOLD(picks), lane);
    };
    load_block(fs);

    int note = -1;
    int64_t base = 0;
    int n = 0, T = 0, ns = 0, out_len = 0;
    float gain = 0.f, kps = 0.f;
    double step_n = 0.0, step_s = 0.0;
    const double *ss = nullptr;
    // note scalars of the spectrum stage
    int apply_bright = 0;
    harm_note hn;
    uint64_t key = 0;
    const int max_back = (NF - 1) / hop;
    auto wc_of = [&](int k) { return (k <= M / 2) ? cconj(twh[k]) : make_float2(-twh[M - k].x, -twh[M - k].y); };
    auto win_of = [&](int k) { return make_float2(win[2 * k] * inv_m, -(win[2 * k + 1] * inv_m)); };

    // summed squared window of this lane's hop samples j = lane + 64 u for interior hops (k_irfft_ola1)
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
    constexpr int KPL = (512 / MASK_DS + KNOT_MARGIN + WAVE - 1) / WAVE;
    double kn_r[KPL];
    int kn_lo = 0;
    const bool slots_ok = hop <= 512 && !HARM;            // the harmonic stem has no mask gain
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
            if constexpr (!HARM) {
                const float ms = slots_ok ? smooth_mask_at32(knot_l, ns, i, n, step_n, step_s, kps)
                                          : smooth_mask_at32(knot_g, ns, i, n, step_n, step_s, kps);
                x = (x * (stem == 1 ? 1.0f - ms : ms)) * gain;
            }
            out[base + i] = x;
        }
    };

    float *ra = reinterpret_cast<float *>(buf);              // the fp32 envelope row staged in the exchange buffer
    for (int64_t f = fs; f < f1; ++f) {
        // The lane index is made opaque once per frame.  Every LDS address of the body (twiddles, window, exchange slots: a few
        // hundred lane-dependent, frame-independent values at 16 points per lane) is otherwise hoisted out of the loop, where
        // there are no registers for it: the compiler spilled 200 of them to scratch and re-loaded them per frame.  Recomputed
        // per frame they are one or two integer instructions each.
        asm volatile("" : "+v"(lane));
        if (!fb.holds(f)) load_block(f);
        const int idx = (int)(f - fb.blk0);
        const int nt = FB_GET(fb, note, idx);
        if (nt != note) {
            note = nt;
            n = FB_GET(fb, n, idx);
            T = FB_GET(fb, T, idx);
            base = FB_BASE(fb, idx);
            ns = (n + MASK_DS - 1) / MASK_DS;
            out_len = hop * (T - 1);
            const goofer_note_params &p = RCOLD(params)[note];
            const double *steps = RCOLD(steps);
            gain = stem == 1 ? p.uv_strength : p.breath_strength;
            step_n = steps[2 * note];
            step_s = steps[2 * note + 1];
            kps = n > 1 ? (float)(ns - 1) / (float)(n - 1) : 0.f;
            ss = RCOLD(short_s) + (base / MASK_DS + note);    // short_base()
            apply_bright = p.apply_brightness;
            key = RCOLD(seed) ^ ((uint64_t)p.seed[0] | ((uint64_t)p.seed[1] << 32));
            if constexpr (HARM) hn.load(p, RCOLD(formants) != nullptr, RCOLD(row_src) != nullptr);
        }
        const int t = FB_GET(fb, t, idx);
        const int shift = (t * hop) & (NF - 1);
        const bool skip_this = skipped(f);
        if (f >= f0 && slots_ok && !skip_this) knots_fetch(t);
        if (skip_this) {
            // no transform: the slots this frame would have started from zero are zeroed, the others keep their sums
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                if (t == 0 || 2 * m >= NF - hop) *reinterpret_cast<float2 *>(ring + ((2 * m + shift) & (NF - 1))) = make_float2(0.f, 0.f);
            }
        } else {
            const float f0f = FB_GETF(fb, f0, idx);
            const bool voiced = apply_bright && FB_GETF(fb, mk, idx) > 0.f;
            const int src = FB_GET(fb, src, idx);
            const float *er = env + (int64_t)src * ld;
            float2 X[PER];                                    // the frame's spectrum: bins lane + 64 i, X[R] = the Nyquist bin (lane 0)
            if constexpr (HARM) {
                // ---- harmonic: k_rfft_frames + k_harm_shape ----
                float ev[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int k = lane + WAVE * i;
                    ev[i] = er[k < B ? k : B - 1];
                }
                {
                    float2 v[R];
                    fetch_frame_pairs<M>(v, pulse + base, t * hop - M, n, buf, lane);
                    warp_grid wg;
                    wg.nyq = RCOLD(grid_nyq); wg.step = RCOLD(grid_step); wg.inv_step = RCOLD(grid_inv);
                    harm_spectrum<M, ring_cold>(X, v, ev, buf, tw, twh, win, lane, hn, voiced, f0f, src, wg, seg, note);
                }
            } else {
                // ---- noise: k_noise_spectra ----
                float e[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int k = lane + WAVE * i;
                    e[i] = er[k < B ? k : B - 1];
                }
                const double *taps175 = RCOLD(taps175);
                if (RCOLD(row_src) && taps175) {
                    // sigma-1.75 blur of the envelope row (GOOFER.py:993): fp32 FMAs in tap order
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int k = lane + WAVE * i;
                        if (k < B) ra[k] = e[i];
                    }
                    wave_lds_sync();
                    float t175[15];
#pragma unroll
                    for (int j = 0; j < 15; ++j) t175[j] = (float)taps175[j];
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int k = lane + WAVE * i;
                        if (k >= B) continue;
                        float acc;
                        // bins 64 .. 64 * (R - 1) - 1 are at least 7 bins inside the row for every lane: no reflect code for them
                        if ((i >= 1 && i <= R - 2) || (k >= 7 && k + 7 < B)) {
                            acc = t175[0] * ra[k - 7];
#pragma unroll
                            for (int j = 1; j < 15; ++j) acc = fmaf(t175[j], ra[k + j - 7], acc);
                        } else {
                            auto refl = [&](int q) { return q < 0 ? -q : (q >= B ? 2 * (B - 1) - q : q); };
                            acc = t175[0] * ra[refl(k - 7)];
#pragma unroll
                            for (int j = 1; j < 15; ++j) acc = fmaf(t175[j], ra[refl(k + j - 7)], acc);
                        }
                        e[i] = acc;
                    }
                    wave_lds_sync();                          // the row is read before the blur row / the exchange overwrite it
                }
                const float *freqs = RCOLD(freqs), *bright = RCOLD(bright_b);
                uint4 rnd = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int k = lane + WAVE * i;
                    X[i] = make_float2(0.f, 0.f);
                    if (k >= B) continue;
                    float c, s;
                    if constexpr (PHI) {
                        const float ph = phi[f * (int64_t)ld + k];
                        c = cosf(ph);
                        s = sinf(ph);
                    } else {
                        // one Philox block feeds eight bins of this lane (bins lane + 64 i, i = 8q..8q+7), 16 bits each
                        if ((i & 7) == 0) rnd = philox_4x32(key, (uint64_t)t, (uint32_t)(lane + WAVE * (i >> 3)));
                        const float rev = (float)philox_half(rnd, i & 7) * (1.0f / 65536.0f);      // phase / 2 pi, uniform in [0, 1)
                        c = __builtin_amdgcn_cosf(rev);
                        s = __builtin_amdgcn_sinf(rev);
                    }
                    float2 u2 = make_float2(c * e[i], s * e[i]);
                    if (stem == 2) {
                        const float h = hp_mask(freqs[k], f0f);
                        u2 = make_float2(u2.x * h, u2.y * h);
                        if (voiced) {
                            const float br = bright[k];
                            u2.x *= br; u2.y *= br;
                            buf[k] = u2;
                        }
                    }
                    X[i] = u2;
                }
                if (stem == 2 && voiced) {
                    const double *t5 = RCOLD(taps5);
                    wave_lds_sync();
                    blur5_row<R>(X, buf, lane, B, t5, std::make_integer_sequence<int, PER>{});
                    wave_lds_sync();
                }
            }

            // ---- k_irfft_ola1: inverse transform into the ring ----
            asm volatile("" : "+v"(lane));
            float2 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) buf[lane + WAVE * r] = X[r];
            if (lane == 0) buf[M] = X[R];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + WAVE * r;
                float2 xk = X[r], xm = buf[M - k];
                if (k == 0) { xk.y = 0.f; xm.y = 0.f; }       // irfft ignores Im of DC and Nyquist
                v[r] = irfft_pre(xk, xm, wc_of(k));
            }
            wave_lds_sync();                                  // the row is read before the transform reuses buf
            float2 z[R];
            wave_fft_keep<M>(v, buf, tw, lane, z);            // the lane's output points stay in registers
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

bool ring_walkers_supported(const goofer_plan_t &p) { return p.n_fft == 2048 && (p.hop & 1) == 0 && p.bl_L == 0; }

// stems: bit 0 harmonic, bit 1 the two noise stems.  The frame maps (frame_note, row_src, picks), the mask steps and — for the
// noise stems — the smoothed mask are in place on `st`; note_mag is zeroed.
int launch_stem_ring(goofer_ctx *ctx, int stems, const float *pulse, const float *env_h, const float *env_n, bool preblurred,
                     const float *phi, int ld, const int64_t *row_src, const double *formants, int64_t total_frames,
                     const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, const float2 *picks,
                     const goofer_note_params *params, uint64_t seed, const double *short_s, const double *steps, float *note_mag,
                     float *harm, float *uv, float *bre, const unsigned char *frame_skip, hipStream_t st)
{
    if (total_frames <= 0 || !(stems & 3)) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    if (!ring_walkers_supported(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the ring walkers are built for n_fft 2048 and an even hop");
    if (!picks) return goofer_fail(ctx, GOOFER_EINVAL, "the ring walkers need the per-frame picks");
    constexpr int M = 1024, WPB = 8;
    const int halo = (p.n_fft + p.hop - 1) / p.hop - 1;
    const size_t lds = ring_lds_bytes<M, WPB>(p.hop);
    if (lds > 160 * 1024) return goofer_fail(ctx, GOOFER_EINVAL, "ring walkers: %zu bytes of LDS", lds);
    int cus = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const int64_t slots = (int64_t)(cus > 0 ? cus : 256) * WPB;          // one workgroup per CU
    ring_args A;
    A.pulse = pulse; A.env_h = env_h; A.env_n = env_n; A.phi = phi;
    A.out[0] = harm; A.out[1] = uv; A.out[2] = bre;
    A.ld = ld; A.hop = p.hop; A.halo = halo;
    A.total_frames = total_frames; A.seed = seed; A.frame_note = frame_note; A.frame_off = frame_off; A.sample_off = sample_off;
    A.row_src = row_src; A.picks = picks; A.short_s = short_s; A.steps = steps; A.formants = formants; A.taps5 = p.blur5;
    A.taps175 = preblurred ? nullptr : p.blur175; A.params = params; A.freqs = p.freqs; A.boost = p.boost; A.bright_h = p.bright_h;
    A.bright_b = p.bright_b; A.note_mag = note_mag; A.g_tw = p.tw_full; A.g_twh = p.tw_half; A.g_win = p.window;
    A.frame_skip = frame_skip;
    {
        const warp_grid wg = make_warp_grid(p.sr, p.n_bins);
        A.grid_nyq = wg.nyq; A.grid_step = wg.step; A.grid_inv = wg.inv_step;
    }
    // one launch per kind of workgroup (the harmonic stem's frames cost two transforms, the noise stems' one or none): jobs =
    // stems x runs; runs sized so that the jobs fill the device a whole number of times, a run >= 8 halos
    for (int kind = 0; kind < 2; ++kind) {
        if (!(stems & (1 << kind))) continue;
        const int count = kind == 0 ? 1 : 2;
        const void *fn = kind == 0 ? (const void *)k_stem_ring<M, WPB, true, false>
                                   : (phi ? (const void *)k_stem_ring<M, WPB, false, true> : (const void *)k_stem_ring<M, WPB, false, false>);
        int rc = kernel_allow_max_lds(ctx, fn);
        if (rc) return rc;
        const int min_run = halo <= 4 ? 32 : 8 * halo;
        const int64_t rounds = (count * total_frames + slots * 256 - 1) / (slots * 256);
        const int64_t fit = (count * total_frames + rounds * slots - 1) / (rounds * slots);
        const int run = (int)(fit > min_run ? fit : min_run);
        const int64_t runs = (total_frames + run - 1) / run;
        A.run = run; A.stem_first = kind == 0 ? 0 : 1; A.stem_count = count;
        const dim3 grid((unsigned)(count * ((runs + WPB - 1) / WPB)));
        if (kind == 0) hipLaunchKernelGGL((k_stem_ring<M, WPB, true, false>), grid, dim3(64 * WPB), lds, st, A);
        else if (phi) hipLaunchKernelGGL((k_stem_ring<M, WPB, false, true>), grid, dim3(64 * WPB), lds, st, A);
        else hipLaunchKernelGGL((k_stem_ring<M, WPB, false, false>), grid, dim3(64 * WPB), lds, st, A);
        LAUNCH_CHECK(ctx);
    }
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// k_rfft_frames and k_harm_shape as one kernel: the frame's spectrum is shaped in the registers the transform leaves it in and
// written once — the unshaped spectrum (8 KB per frame at n_fft 2048) is neither written nor read back.  For the spectra-in-HBM
// pipeline of n_fft 2048 (the inverse transforms follow in k_irfft_ola1); one wave per frame, FRAMES_PER_BLOCK contiguous
// frames per workgroup.
struct rshape_args {
    const float *x, *env;
    float2 *S;
    int ld, ldc, hop;
    int64_t total_frames;
    const int *frame_note;
    const int64_t *frame_off, *sample_off, *row_src;
    const float2 *picks;
    const float *f0, *mask;
    const double *formants, *taps5;
    const goofer_note_params *params;
    const float *freqs, *boost, *bright_h;
    float *note_mag;
    const float2 *g_tw, *g_twh;
    const float *g_win;
    double grid_nyq, grid_step, grid_inv;
};
#define SCOLD(field) COLD(rshape_args, field)
struct rshape_cold {
    static __device__ __forceinline__ const double *formants() { return SCOLD(formants); }
    static __device__ __forceinline__ const double *taps5() { return SCOLD(taps5); }
    static __device__ __forceinline__ const float *freqs() { return SCOLD(freqs); }
    static __device__ __forceinline__ const float *boost() { return SCOLD(boost); }
    static __device__ __forceinline__ const float *bright_h() { return SCOLD(bright_h); }
    static __device__ __forceinline__ float *note_mag() { return SCOLD(note_mag); }
};

template <int M, bool NT>
__global__ __launch_bounds__(256, 2) void k_rfft_shape(const rshape_args A)
{
    constexpr int R = fft_cfg<M>::R, BUF = fft_cfg<M>::BUF, B = M + 1, PER = R + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);
    float2 *twh = tw + M;
    float2 *bufs = twh + (M / 2 + 1);
    float *win = reinterpret_cast<float *>(bufs + WAVES_PER_BLOCK * BUF);
    double *segs = reinterpret_cast<double *>(win + 2 * M + 2);
    load_tables<M>(tw, twh, win, SCOLD(g_tw), SCOLD(g_twh), SCOLD(g_win));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    float2 *buf = bufs + wave * BUF;
    double *seg = segs + wave * WARP_SEG_DOUBLES;
    const int64_t f_begin = (int64_t)blockIdx.x * FRAMES_PER_BLOCK;
    const int ld = A.ld, ldc = A.ldc, hop = A.hop;
    const float *x = A.x, *env = A.env;
    float2 *S = A.S;
    int cur_note = -1;
    harm_note hn;
    int apply_bright = 0;
    for (int q = wave; q < FRAMES_PER_BLOCK; q += WAVES_PER_BLOCK) {
        const int64_t f = f_begin + q;
        if (f >= SCOLD(total_frames)) break;                  // wave-uniform
        asm volatile("" : "+v"(lane));                        // (see k_stem_ring)
        const int note = SCOLD(frame_note)[f];
        const int64_t *sample_off = SCOLD(sample_off);
        const int64_t base = sample_off[note];
        const int n = (int)(sample_off[note + 1] - base);
        const int t = (int)(f - SCOLD(frame_off)[note]);
        if (note != cur_note) {
            cur_note = note;
            const goofer_note_params &p = SCOLD(params)[note];
            apply_bright = p.apply_brightness;
            hn.load(p, SCOLD(formants) != nullptr, SCOLD(row_src) != nullptr);
        }
        float f0f;
        bool voiced;
        if (const float2 *picks = SCOLD(picks)) {             // (f0, mask) record of the frame, written by the map kernel
            const float2 pv = picks[f];
            f0f = pv.x;
            voiced = apply_bright && pv.y > 0.f;
        } else {
            const int64_t pk = pick_index(t, n, hop);
            f0f = n > 0 ? SCOLD(f0)[base + pk] : 0.f;
            voiced = apply_bright && n > 0 && SCOLD(mask)[base + pk] > 0.f;
        }
        const int64_t *row_src = SCOLD(row_src);
        const int64_t src = row_src ? row_src[f] : f;
        const float *er = env + src * (int64_t)ld;
        float ev[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            ev[i] = er[k < B ? k : B - 1];
        }
        float2 X[PER];
        {
            float2 v[R];
            fetch_frame_pairs<M>(v, x + base, t * hop - M, n, buf, lane);
            warp_grid wg;
            wg.nyq = SCOLD(grid_nyq); wg.step = SCOLD(grid_step); wg.inv_step = SCOLD(grid_inv);
            harm_spectrum<M, rshape_cold>(X, v, ev, buf, tw, twh, win, lane, hn, voiced, f0f, (int)src, wg, seg, note);
        }
        float2 *row = S + f * (int64_t)ldc;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            if (k < B) store_f2(row + k, X[i], NT);
        }
    }
}

bool rfft_shape_supported(const goofer_plan_t &p) { return p.n_fft == 2048 && p.bl_L == 0; }

int launch_rfft_shape(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off, const int *frame_note,
                      int64_t total_frames, float2 *S, int ldc, const float *f0, const float *mask, const float *env, int ld,
                      const goofer_note_params *params, float *note_mag, const int64_t *row_src, const double *formants, hipStream_t st)
{
    if (total_frames <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    if (!rfft_shape_supported(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the fused rFFT + shaping kernel is built for n_fft 2048");
    constexpr int M = 1024;
    rshape_args A;
    A.x = x; A.env = env; A.S = S; A.ld = ld; A.ldc = ldc; A.hop = p.hop; A.total_frames = total_frames; A.frame_note = frame_note;
    A.frame_off = frame_off; A.sample_off = sample_off; A.row_src = row_src; A.picks = ctx->frame_picks; A.f0 = f0; A.mask = mask;
    A.formants = formants; A.taps5 = p.blur5; A.params = params; A.freqs = p.freqs; A.boost = p.boost; A.bright_h = p.bright_h;
    A.note_mag = note_mag; A.g_tw = p.tw_full; A.g_twh = p.tw_half; A.g_win = p.window;
    {
        const warp_grid wg = make_warp_grid(p.sr, p.n_bins);
        A.grid_nyq = wg.nyq; A.grid_step = wg.step; A.grid_inv = wg.inv_step;
    }
    const size_t lds = fft_lds_bytes<M>() + sizeof(double) * WAVES_PER_BLOCK * WARP_SEG_DOUBLES + 16;
    const unsigned blocks = (unsigned)((total_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK);
    const void *fn = ctx->nt_spectra ? (const void *)k_rfft_shape<M, true> : (const void *)k_rfft_shape<M, false>;
    int rc = kernel_allow_max_lds(ctx, fn);
    if (rc) return rc;
    if (ctx->nt_spectra) hipLaunchKernelGGL((k_rfft_shape<M, true>), dim3(blocks), dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_rfft_shape<M, false>), dim3(blocks), dim3(256), lds, st, A);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
