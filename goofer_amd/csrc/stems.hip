// Stem-split frame walkers of goofer_synth_batch for the reference's own geometry, hop == n_fft / 4 (gfx950).
//
// The one-kernel-per-reference-step pipeline keeps three [frames x bins] complex spectra in HBM between the framewise
// rFFT, the shaping kernels and the inverse transforms (33 KB per frame, 7 GB per 1024-note step).  Here a wave walks a
// run of consecutive frames and carries each frame from its inputs to finished samples:
//
//   k_noise_stems  sigma-1.75 blur of the envelope row -> random-phase spectra (unvoiced, breath) -> high-pass /
//                  brightness / 5-tap blur -> 2 x irfft -> overlap-add -> stem gains      GOOFER.py:993, 1148-1183
//   k_harm_stem    stft(pulse) -> high-pass -> per-note max -> envelope warps, * env * boost, brightness + blur
//                  -> irfft -> overlap-add                                   GOOFER.py:1099-1146 (+ :840-875, 618-627)
//   k_note_finish  harm / max|S|, per-note peak, gain, reconstruct, V/B/U mix    GOOFER.py:1121, 1208-1218,
//                                                                                  SillySampler.py:1142-1151
//
// No spectrum and no windowed frame ever reaches HBM: per frame the walkers read one envelope row each and the pulse
// samples, and write the three stems (12 KB per frame instead of ~62 KB).  The noise walker needs nothing from the
// pulse chain, so it runs beside the latency-bound phase walk.
//
// Overlap-add in registers: with hop = n_fft / 4 a frame is R = n_fft / 128 groups of 64 sample pairs, lane l holding
// pair l of every group, and frame t's group r lands on absolute group r + (R/4) t — the same lane for every frame.
// The running sums of the R - R/4 groups still open therefore live in that lane's registers (no LDS ring); after frame t
// its first R/4 groups are complete and leave as 8-byte stores.  Every output sample receives its covering frames in
// ascending frame order, the reference's fp32 order (GOOFER.py:379-385), and every product / sum / quotient is the one
// the separate kernels compute, so the results are bit-identical to that path (tested).
#include "binops_core.h"
#include "fft_core.h"
#include "samples_core.h"
#include "stems_core.h"


// The two sigmas are constants of the reference (gaussian_filter1d(env, 1.75), GOOFER.py:993; sigma = 0.5, :1143 / :1171), so
// the walkers carry the taps as literals: twenty scalar registers fewer in a frame loop that was spilling them (the noise
// walker: 40 -> 16 SGPR spills, 0.506 -> 0.48 ms), and the products become v_fmamk / v_fmaak with the tap in the instruction.
// The launchers compare them bit for bit with the taps goofer_plan computes (exp, normalised in fp64, rounded to fp32) and
// refuse to run on a mismatch.
#define STEM_T5 {0x1.14aebe0000000p-12f, 0x1.b405ba0000000p-4f, 0x1.92b9660000000p-1f, 0x1.b405ba0000000p-4f, 0x1.14aebe0000000p-12f}
#define STEM_T175 {0x1.40c2ee0000000p-14f, 0x1.4edb880000000p-11f, 0x1.f8612e0000000p-9f, 0x1.120a540000000p-6f, 0x1.ada7fc0000000p-5f, 0x1.e5fa880000000p-4f, 0x1.8c8df20000000p-3f, 0x1.d2e20e0000000p-3f, 0x1.8c8df20000000p-3f, 0x1.e5fa880000000p-4f, 0x1.ada7fc0000000p-5f, 0x1.120a540000000p-6f, 0x1.f8612e0000000p-9f, 0x1.4edb880000000p-11f, 0x1.40c2ee0000000p-14f}
static bool stem_taps_match(const goofer_plan_t &p)
{
    const float t5[5] = STEM_T5, t175[15] = STEM_T175;
    return memcmp(t5, p.taps5_f, sizeof(t5)) == 0 && memcmp(t175, p.taps175_f, sizeof(t175)) == 0;
}


template <int M> struct stem_cfg {
    static constexpr int R = M / 64;           // sample pairs (and FFT points) per lane
    static constexpr int G = R / 4;            // 64-pair groups per hop
    static constexpr int NF = 2 * M;
    static constexpr int HOP = M / 2;
    static constexpr int B = M + 1;
    static constexpr int PER = R + 1;          // bins per lane: k = lane + 64 i; i == R is the Nyquist bin (lane 0 only)
    static constexpr int ROWF = (B + 3) & ~3;  // floats of a staged fp32 row / bin table
    static constexpr int KN = HOP / MASK_DS + KNOT_MARGIN;
    static constexpr int KPL = (KN + WAVE - 1) / WAVE;
    // Workgroup tables: FFT twiddles, conj-trick twiddles per bin, scaled synthesis window per sample pair, NTAB per-bin
    // curves, and (WIN) the plain window.  Per wave: the FFT exchange buffer — which also stages the envelope row, the
    // complex row of the 5-tap blur and the mirrored-bin exchange, one after the other — and the mask knots of a hop.
    // Sized so that four workgroups share a CU (4 waves per SIMD): a walker is a latency-bound chain of LDS exchanges and
    // needs the waves more than it needs registers.
    static constexpr size_t wave_bytes = sizeof(float2) * fft_cfg<M>::BUF + sizeof(double) * KN;
    template <int NTAB, bool WIN> static constexpr size_t table_bytes()
    {
        return sizeof(float2) * (4 * M + fft_tw_tabs<M>::TOTAL) + sizeof(float) * (NTAB * ROWF + (WIN ? NF : 0) + 4 * G * WAVE);
    }
    template <int NTAB, bool WIN> static constexpr size_t lds_bytes() { return table_bytes<NTAB, WIN>() + WAVES_PER_BLOCK * wave_bytes; }
    static_assert(wave_bytes % 16 == 0, "16-byte aligned LDS carving");
};

// cmul(conj(a), b) in two packed instructions (see cmul in fft_core.h): the conjugation is a neg modifier
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)
{
    v2f_t va = {a.x, a.y}, vb = {b.x, b.y}, t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return make_float2(r.x, r.y);
}

// blur5 (binops_core.h) with the taps already rounded to fp32: the same five FMAs
__device__ __forceinline__ float2 blur5f(const float2 *r, int k, int n_bins, const float (&t)[5])
{
    float2 v0, v1, v2, v3, v4;
    if (k >= 2 && k + 2 < n_bins) {
        v0 = r[k - 2]; v1 = r[k - 1]; v2 = r[k]; v3 = r[k + 1]; v4 = r[k + 2];
    } else {
        auto at = [&](int q) { return r[q < 0 ? -q : (q >= n_bins ? 2 * (n_bins - 1) - q : q)]; };
        v0 = at(k - 2); v1 = at(k - 1); v2 = at(k); v3 = at(k + 1); v4 = at(k + 2);
    }
    float re = t[0] * v0.x, im = t[0] * v0.y;
    re = fmaf(t[1], v1.x, re); im = fmaf(t[1], v1.y, im);
    re = fmaf(t[2], v2.x, re); im = fmaf(t[2], v2.y, im);
    re = fmaf(t[3], v3.x, re); im = fmaf(t[3], v3.y, im);
    re = fmaf(t[4], v4.x, re); im = fmaf(t[4], v4.y, im);
    return make_float2(re, im);
}


// Shared state of a walker wave: tables, per-lane constants, the note it is in.
template <int M, int NTAB, bool WIN, bool WS_LDS = false> struct walker {
    using C = stem_cfg<M>;
    static constexpr int R = C::R, G = C::G, NF = C::NF, HOP = C::HOP, B = C::B;
    float2 *tw, *wct, *wsc, *wsv, *buf, *tw1, *tw2;
    float *win, *tab, *wst;                    // wst[(2 g + c) * 64 + lane]: summed squared window of the lane's hop sample (g, c) on interior
                                               // hops, [+ 2 G * 64]: its reciprocal — WS_LDS: read from this table where they are used
                                               // (the harmonic walker: eight registers towards its third wave per SIMD), else copied
                                               // into registers once (the noise walker, which has no third wave to gain)
    float ws_r[G][2], rws_r[G][2];
    double *kbuf;
    int lane;
    // note state (wave-uniform)
    int note = -1, n = 0, T = 0, out_len = 0;
    int64_t base = 0;

    __device__ __forceinline__ void init(unsigned char *smem, const float2 *g_tw, const float2 *g_twh, const float *g_win,
                                         const float *g_winb, const float *t0, const float *t1, const float *t2)
    {
        tw = reinterpret_cast<float2 *>(smem);
        wct = tw + M;
        wsc = wct + M;
        wsv = wsc + M;
        tw1 = wsv + M;                                         // the two passes' twiddles, contiguous per pass (fft_core.h)
        tw2 = tw1 + fft_tw_tabs<M>::N1;
        tab = reinterpret_cast<float *>(tw2 + fft_tw_tabs<M>::N2);
        win = tab + NTAB * C::ROWF;
        wst = win + (WIN ? NF : 0);
        const float *tsrc[3] = {t0, t1, t2};
        for (int i = threadIdx.x; i < C::ROWF; i += blockDim.x) {
            const int k = i < B ? i : B - 1;
#pragma unroll
            for (int q = 0; q < NTAB; ++q) tab[q * C::ROWF + i] = tsrc[q][k];
        }
        const float inv_m = 0.5f / (float)M;   // 1/M of the transform and the 1/2 of the input stage (irfft_pre)
        for (int k = threadIdx.x; k < M; k += blockDim.x) {
            tw[k] = g_tw[k];
            // conj-trick twiddle of bin k: the conjugate of the half-bin twiddle exp(-i pi k / M)
            wct[k] = (k <= M / 2) ? cconj(g_twh[k]) : make_float2(-g_twh[M - k].x, -g_twh[M - k].y);
            // synthesis window of sample pair k with 1/M and 1/2 folded in: (z / M) * w == z * (w / M) exactly (M is a
            // power of two), and the conjugate's sign rides along
            wsc[k] = make_float2(g_win[2 * k] * inv_m, -(g_win[2 * k + 1] * inv_m));
            // the same for frames whose spectrum the reference blurs along bins with the sigma-0.5 taps (brightened voiced
            // frames, GOOFER.py:1143 / 1171): blurring a spectrum circularly is multiplying the samples by the taps' transform,
            // so those frames are windowed with window x W and the 5-tap pass over the bins is not run at all
            wsv[k] = make_float2(g_winb[2 * k] * inv_m, -(g_winb[2 * k + 1] * inv_m));
        }
        if (WIN)
            for (int i = threadIdx.x; i < NF; i += blockDim.x) win[i] = g_win[i];
        if (threadIdx.x < WAVE) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int j = 2 * ((int)threadIdx.x + WAVE * g) + c;
                    float ws = 0.f;
                    for (int q = (NF - 1 - j) / HOP; q >= 0; --q) {       // ascending frame order = descending offset
                        const float wv = g_win[j + q * HOP];
                        ws += wv * wv;
                    }
                    wst[(2 * g + c) * WAVE + threadIdx.x] = ws;
                    wst[(2 * G + 2 * g + c) * WAVE + threadIdx.x] = 1.0f / ws;
                }
        }
        __syncthreads();
        fill_tw_tabs<M>(tw1, tw2, tw);
        __syncthreads();
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        unsigned char *per = smem + C::template table_bytes<NTAB, WIN>() + (size_t)wave * C::wave_bytes;
        buf = reinterpret_cast<float2 *>(per);
        kbuf = reinterpret_cast<double *>(buf + fft_cfg<M>::BUF);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                ws_r[g][c] = WS_LDS ? 0.f : wst[(2 * g + c) * WAVE + lane];
                rws_r[g][c] = WS_LDS ? 0.f : wst[(2 * G + 2 * g + c) * WAVE + lane];
            }
    }

    __device__ __forceinline__ float ws_of(int g, int c) const { return WS_LDS ? wst[(2 * g + c) * WAVE + lane] : ws_r[g][c]; }
    __device__ __forceinline__ float rws_of(int g, int c) const { return WS_LDS ? wst[(2 * G + 2 * g + c) * WAVE + lane] : rws_r[g][c]; }

    // run of frames [f0, f1) of this wave; fs = first frame to transform (the halo in front of a run that starts inside a note)
    __device__ __forceinline__ bool range(int64_t total_frames, int run, const int *frame_note, const int64_t *frame_off, int64_t &fs,
                                          int64_t &f0, int64_t &f1)
    {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        f0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * run;
        if (f0 >= total_frames) return false;
        f1 = f0 + run < total_frames ? f0 + run : total_frames;
        const int64_t t0 = f0 - frame_off[frame_note[f0]];
        constexpr int halo = NF / HOP - 1;
        fs = f0 - (t0 < halo ? t0 : halo);
        return true;
    }

    __device__ __forceinline__ void enter_note(const frame_block &fb, int idx)
    {
        note = FB_GET(fb, note, idx);
        n = FB_GET(fb, n, idx);
        T = FB_GET(fb, T, idx);
        out_len = HOP * (T - 1);
        base = FB_BASE(fb, idx);
    }

    // irFFT of the spectrum whose bins k = lane + 64 i this lane holds in x[] (x[R]: the Nyquist bin, lane 0): the complex row
    // goes through `buf` for the mirrored bins, the complex M-point transform runs (twiddles from LDS), the frame is windowed
    // and overlap-added into `carry`; out[g] = the finished groups of hop t.
    __device__ __forceinline__ void inverse_ola(const float2 (&x)[C::PER], int t, float2 (&carry)[R - G], float2 (&out)[G],
                                                const float2 *wtab)
    {
#pragma unroll
        for (int i = 0; i < R; ++i) buf[lane + WAVE * i] = x[i];
        if (lane == 0) buf[M] = x[R];
        wave_lds_sync();
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = lane + WAVE * r;
            float2 xk = x[r], xq = buf[M - k];
            if (k == 0) { xk.y = 0.f; xq.y = 0.f; }                  // irfft ignores Im of DC and Nyquist
            v[r] = irfft_pre(xk, xq, wct[k]);
        }
        wave_lds_sync();                                             // the row is read before the transform reuses buf
        float2 z[R];
        wave_fft_keep_tab<M>(v, buf, tw1, tw2, lane, z);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float2 wn = wtab[lane + WAVE * r];
            const float a = z[r].x * wn.x, b = z[r].y * wn.y;
            float2 s;
            if (r < R - G && t != 0) s = make_float2(carry[r].x + a, carry[r].y + b);
            else s = make_float2(a, b);                              // first contribution: y starts from zero
            if (r < G) out[r] = s;
            else carry[r - G] = s;                                   // slot r - G was consumed G steps ago
        }
    }

    // The reference blurs a voiced frame's spectrum over the bins with numpy-'reflect' edges; the blurred window (wsv) blurs it
    // over its Hermitian continuation.  This adds, to the six bins next to either edge, the purely imaginary D with
    // blur(D) = the difference (see goofer_plan: blur_edge), so that the two agree to fp32 rounding.  ec[]: this lane's four
    // coefficients; t0, t1: the outer taps.
    __device__ __forceinline__ void blur_edges(float2 (&x)[C::PER], const float (&ec)[4], float t0, float t1) const
    {
        const float im0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[0].y), 0));
        const float im1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[0].y), 1));
        const float imM = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[R].y), 0));
        const float imN = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[R - 1].y), WAVE - 1));
        const float a = fmaf(2.0f * t0, im1, t1 * im0), b = t0 * im0;
        const float at = fmaf(2.0f * t0, imN, t1 * imM), bt = t0 * imM;
        x[0].y = fmaf(ec[0], a, fmaf(ec[1], b, x[0].y));
        x[R - 1].y = fmaf(ec[2], at, fmaf(ec[3], bt, x[R - 1].y));
    }

    // A frame whose transform is skipped (its spectrum cannot reach a non-zero output sample): the ring still moves on
    __device__ __forceinline__ void skip_ola(int t, float2 (&carry)[R - G], float2 (&out)[G])
    {
        const float2 zero = make_float2(0.f, 0.f);
#pragma unroll
        for (int g = 0; g < G; ++g) out[g] = t != 0 ? carry[g] : zero;
#pragma unroll
        for (int r = G; r < R; ++r) carry[r - G] = (r < R - G && t != 0) ? carry[r] : zero;
    }

    // Overlap-add divisor of this lane's sample (g, c) of hop h (GOOFER.py:385-389): the summed squared window over the
    // frames that exist.  Interior hops (wave-uniform test) have the per-lane constant and its reciprocal at hand.
    __device__ __forceinline__ bool interior(int h) const { return h >= (NF - 1) / HOP && h <= T - 1; }
    __device__ __forceinline__ float partial_ws(int h, int g, int c, const float *g_win) const
    {
        const int j = 2 * (lane + WAVE * g) + c;
        const int back = (NF - 1 - j) / HOP;
        const int flo = h - back < 0 ? 0 : h - back, fhi = h > T - 1 ? T - 1 : h;
        float ws = 0.f;
        for (int fr = flo; fr <= fhi; ++fr) {
            const float w = g_win[j + (h - fr) * HOP];
            ws += w * w;
        }
        return ws;
    }
};

// ---------------------------------------------------------------------------------------------
// PHI: injected phases (parity runs; accurate libm sin / cos) instead of Philox + hardware sin / cos.  A template parameter,
// not a branch: the libm code is nine inlined range reductions and would triple the loop body of the production kernel.
struct noise_args {
    // used every frame
    const float *env, *phi;
    float *uv, *bre;
    unsigned char *hopz;           // per output hop of a note (slot: first frame of the note + 3 * note + hop): bit 0 the unvoiced stem's
                                   // samples of the hop are exactly zero and were NOT stored, bit 1 the same for the breath stem
    const double *short_s;
    int ld, mode, run;             // mode bit 0: blur the rows here; bit 1: never skip a transform (A/B); bit 2: the 5-tap bin blur
                                   // of voiced frames as a window on the samples
    // set-up, frame records (every 64 frames), note entry: read where they are used (cold_arg)
    int64_t total_frames;
    uint64_t seed;
    const int64_t *row_src;
    const int *frame_note;
    const int64_t *frame_off, *sample_off;
    const float2 *picks;
    const goofer_note_params *params;
    const double *steps;
    const float *freqs, *bright;
    const float2 *g_tw, *g_twh;
    const float *g_win, *g_winb, *g_edge;
};
#define NCOLD(field) COLD(noise_args, field)

template <int M, bool PHI>
__global__ __launch_bounds__(256, 2) void k_noise_stems(const noise_args A)
{
    using C = stem_cfg<M>;
    constexpr int R = C::R, G = C::G, HOP = C::HOP, B = C::B, PER = C::PER, KN = C::KN, KPL = C::KPL, ROWF = C::ROWF;
    static_assert(R == 8, "the row staging below moves two 16-byte pieces per lane");
    extern __shared__ __align__(16) unsigned char smem[];
    walker<M, 2, false, true> w;
    w.init(smem, NCOLD(g_tw), NCOLD(g_twh), NCOLD(g_win), NCOLD(g_winb), NCOLD(freqs), NCOLD(bright), nullptr);
    const int lane = w.lane;
    const float *t_fq = w.tab, *t_br = w.tab + ROWF;
    const float *__restrict__ env = A.env;
    const float *__restrict__ phi = A.phi;
    float *__restrict__ uv = A.uv, *__restrict__ bre = A.bre;
    const int ld = A.ld, mode = A.mode;
    // frame indices as 32-bit integers in the loop (a batch of 2^31 frames would be 4 TB of envelope rows): four scalar
    // registers fewer in a loop that spills them
    int fs, f0, f1;
    {
        const int64_t total_frames = NCOLD(total_frames);
        int64_t fs64, f064, f164;
        if (!w.range(total_frames, A.run, NCOLD(frame_note), NCOLD(frame_off), fs64, f064, f164)) return;   // no block barrier below
        fs = (int)fs64; f0 = (int)f064; f1 = (int)f164;
    }
    frame_block fb;
    auto load_block = [&](int64_t first) {
        fb.load(first, NCOLD(total_frames), NCOLD(frame_note), NCOLD(frame_off), NCOLD(sample_off), NCOLD(row_src), NCOLD(picks), lane);
        if constexpr (!PHI) fb.load_nyquist(NCOLD(params), NCOLD(seed), M);
    };
    load_block(fs);
    // bins 64 .. M sit a whole 64-bin stride above the lowest bins: when f0 + 100 Hz is still below bin 64 their high-pass
    // factor is exactly 1.0f (1 + exp(-z) rounds to 1 for z > 18, i.e. 90 Hz above f0; rcp(1) = 1) and is not evaluated
    const float fq64 = w.tab[WAVE];
    float ec[4];
    {
        const float *g_edge = NCOLD(g_edge);
#pragma unroll
        for (int q = 0; q < 4; ++q) ec[q] = g_edge[q * WAVE + lane];
    }
    constexpr float t5[5] = STEM_T5, t175[15] = STEM_T175;

    // the next frame's envelope row is in flight while the current one is transformed: bins 8 lane .. 8 lane + 7 as two
    // 16-byte loads, the Nyquist bin beside them
    float4 ea, eb4;
    float e_ny;
    auto fetch = [&](int src) {
        const float *er = env + (int64_t)src * ld;
        ea = *reinterpret_cast<const float4 *>(er + 8 * lane);
        eb4 = *reinterpret_cast<const float4 *>(er + 8 * lane + 4);
        e_ny = er[B - 1];
    };
    fetch(FB_GET(fb, src, 0));

    float2 carry_u[R - G], carry_b[R - G];
#pragma unroll
    for (int r = 0; r < R - G; ++r) carry_u[r] = carry_b[r] = make_float2(0.f, 0.f);

    // note scalars of the output stage
    int ns = 0;
    float g_b = 0.f, g_u = 0.f, kps = 0.f;
    const double *ss = nullptr;
    uint64_t key = 0;
    int apply_bright = 0;

    // The smoothed-mask knots a hop needs are fetched lane-parallel a frame ahead and parked in LDS, so the output stage
    // reads them with LDS latency instead of issuing dependent global loads per sample.
    double kn_r[KPL];
    int kn_lo = 0;
    auto knots_fetch = [&](int h) {
        int i0 = h * HOP - M;
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

    // Exact sparsity of the stem gains (GOOFER.py:1179-1183): where the smoothed mask is flat at 1 the unvoiced stem is
    // multiplied by exactly 0, where it is flat at 0 the breath stem is.  Frame t reaches hops t .. t + 3 only; when all the
    // mask knots those hops can read are one constant c with 1 - float(c) == 0 (float(c) == 0), the frame's unvoiced (breath)
    // spectrum cannot reach a non-zero sample and its transform is skipped — every sample that is not exactly zero still
    // receives all its frames, in order.  Bit j of the masks: hop t + j is flat at one / at zero; unknown hops (the first
    // three frames of a run or a note) count as not flat.
    unsigned one_bits = 0, zero_bits = 0;
    double hk[2] = {0.0, 0.0}, hk_c = 0.0;
    int hk_lo = 0, hk_hi = -1;
    bool hk_none = false;
    auto hop_check_issue = [&](int h) {
        int i_lo = (h - 2) * HOP, i_hi = (h - 1) * HOP - 1;
        hk_none = i_hi < 0 || i_lo >= w.n || ns <= 0;            // the hop has no output sample
        i_lo = i_lo < 0 ? 0 : i_lo;
        i_hi = i_hi > w.n - 1 ? w.n - 1 : i_hi;
        int lo = (int)((float)i_lo * kps) - 4, hi = (int)((float)i_hi * kps) + 6;
        lo = lo < 0 ? 0 : lo;
        hi = hi > ns - 1 ? ns - 1 : hi;
        hk_lo = lo;
        hk_hi = hk_none ? lo - 1 : hi;
        if (!hk_none) {
            hk_c = ss[lo];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = lo + lane + WAVE * c;
                hk[c] = ss[k <= hi ? k : hi];
            }
        }
    };
    auto hop_check_done = [&]() {
        bool one = !(mode & 2), zero = one;
        if (!hk_none) {
            const bool same = hk[0] == hk_c && hk[1] == hk_c && hk_hi - hk_lo < 2 * WAVE;
            const bool flat = __all(same);
            const float cf = (float)hk_c;
            one = one && flat && (1.0f - cf) == 0.0f;
            zero = zero && flat && cf == 0.0f;
        }
        one_bits = (one_bits >> 1) | (one ? 8u : 0u);
        zero_bits = (zero_bits >> 1) | (zero ? 8u : 0u);
    };

    float *rp = reinterpret_cast<float *>(w.buf);             // staged fp32 rows (dead before the spectra use buf)
    for (int f = fs; f < f1; ++f) {
        const int idx = f - (int)fb.blk0;
        const int t = FB_GET(fb, t, idx);
        if (FB_GET(fb, note, idx) != w.note) {
            one_bits = zero_bits = 0;
            w.enter_note(fb, idx);
            const int nt = w.note;
            const goofer_note_params &p = NCOLD(params)[nt];
            ns = (w.n + MASK_DS - 1) / MASK_DS;
            g_b = p.breath_strength;
            g_u = p.uv_strength;
            kps = w.n > 1 ? (float)(ns - 1) / (float)(w.n - 1) : 0.f;
            ss = A.short_s + (w.base / MASK_DS + nt);         // short_base()
            key = NCOLD(seed) ^ ((uint64_t)p.seed[0] | ((uint64_t)p.seed[1] << 32));
            apply_bright = p.apply_brightness;
        }
        const float f0f = FB_GETF(fb, f0, idx);
        const bool voiced = apply_bright && FB_GETF(fb, mk, idx) > 0.f;
        // hop t's flatness was settled three frames ago (bit 1 until this frame's check shifts the masks): a flat hop's mask
        // gain is one constant and needs no knots
        const bool flat_t = (((one_bits | zero_bits) >> 1) & 1u) != 0;
        if (f >= f0 && !flat_t) knots_fetch(t);               // lands during the two transforms below
        hop_check_issue(t + 3);
        const uint32_t ny_u = PHI ? 0u : (uint32_t)FB_GET(fb, ny_u, idx);
        const bool hp_low_only = fq64 - f0f > 100.0f;

        // 1. noise envelope: sigma-1.75 blur of the un-warped row (GOOFER.py:993), fp32 FMAs in tap order.  A lane blurs its
        //    eight consecutive bins from a 24-value window (its own eight, eight on either side from the neighbouring lanes), lane 63 also the
        //    Nyquist bin; the blurred row then goes through LDS once more into the transform's layout k = lane + 64 i.
        float en[PER];
        {
            float o9[9];
            if (mode & 1) {
                // the eight bins of the lane on either side through DPP wave shifts (lane i reads lane i -+ 1): sixteen moves on
                // the vector pipe instead of an LDS round trip of the row (two 16-byte stores, a wait, four 16-byte loads)
                float x[24];
                {
                    const float own[8] = {ea.x, ea.y, ea.z, ea.w, eb4.x, eb4.y, eb4.z, eb4.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        x[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(own[j]), 0x138, 0xf, 0xf, false));        // wave_shr:1 <- lane - 1
                        x[16 + j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(own[j]), 0x130, 0xf, 0xf, false));   // wave_shl:1 <- lane + 1
                    }
                }
                const float ec[8] = {ea.x, ea.y, ea.z, ea.w, eb4.x, eb4.y, eb4.z, eb4.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) x[8 + j] = ec[j];
                // numpy 'reflect' at the two ends of the row: bins -1..-7 are bins 1..7, bins 513..519 are bins 511..505
#pragma unroll
                for (int j = 1; j < 8; ++j) {
                    x[8 - j] = lane == 0 ? ec[j] : x[8 - j];
                    x[16 + j] = lane == 63 ? ec[8 - j] : x[16 + j];
                }
                x[16] = lane == 63 ? e_ny : x[16];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float acc = t175[0] * x[j + 1];
#pragma unroll
                    for (int q = 1; q < 15; ++q) acc = fmaf(t175[q], x[j + 1 + q], acc);
                    o9[j] = acc;
                }
            } else {
                o9[0] = ea.x; o9[1] = ea.y; o9[2] = ea.z; o9[3] = ea.w; o9[4] = eb4.x; o9[5] = eb4.y; o9[6] = eb4.z; o9[7] = eb4.w;
                o9[8] = e_ny;
            }
            *reinterpret_cast<float4 *>(rp + 8 * lane) = make_float4(o9[0], o9[1], o9[2], o9[3]);
            *reinterpret_cast<float4 *>(rp + 8 * lane + 4) = make_float4(o9[4], o9[5], o9[6], o9[7]);
            if (lane == 63) rp[B - 1] = o9[8];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < PER; ++i) en[i] = rp[lane + WAVE * i < B ? lane + WAVE * i : B - 1];
            wave_lds_sync();                                  // row dead: buf is free
        }
        if (f + 1 < f1) {                                     // the row registers are consumed: start the next frame's row
            if (!fb.holds(f + 1)) load_block(f + 1);
            fetch(FB_GET(fb, src, f + 1 - (int)fb.blk0));
        }

        // 2. U * env_n (unvoiced spectrum; waits in registers) and U * env_n * HP (* brightness) (breath spectrum)  GOOFER.py:1148-1173
        float2 su[PER], sb[PER];
        uint4 rnd = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            su[i] = sb[i] = make_float2(0.f, 0.f);
            if (k >= B) continue;
            float c, s;
            if constexpr (PHI) {
                const float ph = phi[f * (int64_t)ld + k];
                c = cosf(ph);
                s = sinf(ph);
            } else {
                // one Philox block feeds the lane's eight bins (16-bit phases); the Nyquist bin's was drawn with the frame records
                if (i == 0) rnd = philox_4x32(key, (uint64_t)t, (uint32_t)lane);
                const uint32_t u = i == R ? ny_u : philox_half(rnd, i);
                const float rev = (float)u * (1.0f / 65536.0f);                 // phase / 2 pi, uniform in [0, 1)
                c = __builtin_amdgcn_cosf(rev);
                s = __builtin_amdgcn_sinf(rev);
            }
            su[i] = make_float2(c * en[i], s * en[i]);
            const float h = (i == 0 || !hp_low_only) ? hp_mask(t_fq[k], f0f) : 1.0f;
            sb[i] = make_float2(su[i].x * h, su[i].y * h);
            if (voiced) {
                const float b = t_br[k];
                sb[i].x *= b; sb[i].y *= b;
            }
        }
        const bool td_blur = (mode & 4) != 0;
        if (voiced && !td_blur) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = lane + WAVE * i;
                if (k < B) w.buf[k] = sb[i];
            }
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = lane + WAVE * i;
                if (k < B) sb[i] = blur5f(w.buf, k, B, t5);
            }
            wave_lds_sync();
        }
        // 3. inverse transforms + overlap-add (skipped where the stem's gain is exactly zero over everything the frame reaches);
        //    the breath frame of a voiced frame leaves through the blurred window (see walker::init) unless the pass above ran
        hop_check_done();
        float2 ob[G], ou[G];
        if (zero_bits == 15u) {
            w.skip_ola(t, carry_b, ob);
        } else {
            if (voiced && td_blur) w.blur_edges(sb, ec, t5[0], t5[1]);
            w.inverse_ola(sb, t, carry_b, ob, (voiced && td_blur) ? w.wsv : w.wsc);
        }
        if (one_bits == 15u) w.skip_ola(t, carry_u, ou);
        else w.inverse_ola(su, t, carry_u, ou, w.wsc);

        // 4. hop t -> window-sum quotient -> mask upsample -> stem gains -> out (GOOFER.py:385-389, 1179-1183).  First the common
        //    case without a test per sample: the hop lies inside the note, every frame over it exists, the note goes on, and the
        //    hop's mask gain is one constant (see hop_check_done) — the same quotients and products as the general loop below
        const int p0t = t * HOP - M;
        const bool flat1_t = (one_bits & 1u) != 0, flat0_t = (zero_bits & 1u) != 0;
        const int hop_slot = f - t + 3 * w.note;               // (f - t: the note's first frame)
        if (f >= f0 && (flat1_t || flat0_t) && w.interior(t) && t != w.T - 1 && p0t >= 0 && p0t + HOP <= w.n && p0t + HOP <= w.out_len) {
            // The hop's mask gain is one constant: the stem it multiplies by exactly zero is exactly zero over the whole hop (a
            // product with 0.0f: +-0).  That stem's 1 KB is not stored; the hop's byte says so and k_note_finish takes zeros.
            if (lane == 0) NCOLD(hopz)[hop_slot + t] = (unsigned char)((flat1_t ? 1 : 0) | (flat0_t ? 2 : 0));
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i0 = p0t + 2 * (lane + WAVE * g);
                float x[2] = {flat1_t ? ob[g].x : ou[g].x, flat1_t ? ob[g].y : ou[g].y};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float ws = w.ws_of(g, c), rw = w.rws_of(g, c);
                    if (ws > 1e-9f) x[c] = div_by(x[c], ws, rw);
                    x[c] = (x[c] * 1.0f) * (flat1_t ? g_b : g_u);          // (x * ms) * g_b with ms = 1 / (x * (1 - ms)) * g_u with ms = 0
                }
                *reinterpret_cast<float2_u *>((flat1_t ? bre : uv) + w.base + i0) = make_float2(x[0], x[1]);
            }
        } else if (f >= f0) {
            //    ... and the general case: behind a note's last frame also the hop still open in the registers and the zero tail
            for (int h = t;;) {
                // a hop known flat (see hop_check_done; bit h - t of the masks) has ONE mask gain: 1 where the smoothed mask is
                // flat at the tap sum, 0 where it is flat at zero — smooth_mask_at32 would return exactly that for every sample
                const int sh = h - t;
                const bool flat1 = sh < 4 && ((one_bits >> sh) & 1u), flat0 = sh < 4 && ((zero_bits >> sh) & 1u);
                const bool flat = flat1 || flat0;
                const float ms_flat = flat1 ? 1.0f : 0.0f;
                if (!flat) {
#pragma unroll
                    for (int c = 0; c < KPL; ++c) {
                        const int e = lane + WAVE * c;
                        if (e < KN) w.kbuf[e] = kn_r[c];
                    }
                    wave_lds_sync();
                }
                if (lane == 0) NCOLD(hopz)[hop_slot + h] = 0;  // both stems stored
                const int p0 = h * HOP - M;
                const int e_hi = KN - 1, lo = kn_lo;
                auto knot = [&](int k) {
                    const int e = k - lo;
                    return w.kbuf[e < e_hi ? e : e_hi];
                };
                const bool inner = w.interior(h);
                // (the two linspace steps of the note: read here, where a hop that is not flat interpolates its gains — two scalar
                // loads per such hop instead of four scalar registers held across every frame)
                double step_n = 0.0, step_s = 0.0;
                if (!flat) {
                    const double *steps = NCOLD(steps);
                    step_n = steps[2 * w.note];
                    step_s = steps[2 * w.note + 1];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int i0 = p0 + 2 * (lane + WAVE * g);
                    if (i0 < 0 || i0 >= w.n) continue;
                    float xu[2] = {ou[g].x, ou[g].y}, xb[2] = {ob[g].x, ob[g].y};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int i = i0 + c;
                        if (i < w.out_len) {
                            if (inner) {
                                const float ws = w.ws_of(g, c), rw = w.rws_of(g, c);
                                if (ws > 1e-9f) { xu[c] = div_by(xu[c], ws, rw); xb[c] = div_by(xb[c], ws, rw); }
                            } else {
                                const float ws = w.partial_ws(h, g, c, NCOLD(g_win));
                                if (ws > 1e-9f) { xu[c] /= ws; xb[c] /= ws; }
                            }
                        } else {
                            xu[c] = 0.f;                             // zero tail of istft (GOOFER.py:409-412)
                            xb[c] = 0.f;
                        }
                        const float ms = flat ? ms_flat : smooth_mask_at32(knot, ns, i < w.n ? i : w.n - 1, w.n, step_n, step_s, kps);
                        xb[c] = (xb[c] * ms) * g_b;
                        xu[c] = (xu[c] * (1.0f - ms)) * g_u;
                    }
                    if (i0 + 1 < w.n) {
                        *reinterpret_cast<float2_u *>(uv + w.base + i0) = make_float2(xu[0], xu[1]);
                        *reinterpret_cast<float2_u *>(bre + w.base + i0) = make_float2(xb[0], xb[1]);
                    } else {
                        uv[w.base + i0] = xu[0];
                        bre[w.base + i0] = xb[0];
                    }
                }
                ++h;
                if (t != w.T - 1 || h * HOP - M >= w.n) break;
                // flush: hop T is the sums still in the registers; beyond it only zeros are written (i >= out_len)
#pragma unroll
                for (int g = 0; g < G; ++g) { ou[g] = carry_u[g]; ob[g] = carry_b[g]; }
                if (!(h - t < 4 && (((one_bits | zero_bits) >> (h - t)) & 1u))) knots_fetch(h);
            }
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// `env` holds the harmonic envelope rows as the shaping step needs them: already warped (k_warp_bins, one row per frame)
// or, when no note of the batch warps, the source rows addressed through row_src.  env_plain (goofer_render_batch): the
// assembled rows; `env` then holds warped copies of the warping notes' rows only (frame_block::mark_plain).
template <int M>
__global__ __launch_bounds__(256, 3) void k_harm_stem(const float *__restrict__ pulse, const float *__restrict__ env,
                                                      const float *__restrict__ env_plain, int have_formants, int ld,
                                                      const int64_t *__restrict__ row_src, int64_t total_frames,
                                                      const int *__restrict__ frame_note, const int64_t *__restrict__ frame_off,
                                                      const int64_t *__restrict__ sample_off, const float2 *__restrict__ picks,
                                                      const goofer_note_params *__restrict__ params, const float *__restrict__ freqs,
                                                      const float *__restrict__ boost, const float *__restrict__ bright,
                                                      float *__restrict__ harm, float *__restrict__ note_mag,
                                                      int run, const float2 *__restrict__ g_tw, const float2 *__restrict__ g_twh,
                                                      const float *__restrict__ g_win, const float *__restrict__ g_winb, const float *__restrict__ g_edge, int td_blur)
{
    using C = stem_cfg<M>;
    constexpr int R = C::R, G = C::G, HOP = C::HOP, B = C::B, PER = C::PER, ROWF = C::ROWF;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr float t5[5] = STEM_T5;
    walker<M, 3, true, true> w;
    w.init(smem, g_tw, g_twh, g_win, g_winb, freqs, boost, bright);
    const int lane = w.lane;
    const float *t_fq = w.tab, *t_bo = w.tab + ROWF, *t_br = w.tab + 2 * ROWF;
    int64_t fs, f0, f1;
    if (!w.range(total_frames, run, frame_note, frame_off, fs, f0, f1)) return;   // no block barrier below
    frame_block fb;
    fb.load(fs, total_frames, frame_note, frame_off, sample_off, row_src, picks, lane);
    if (env_plain) fb.mark_plain(params, have_formants != 0);

    // Three waves per SIMD (168 registers): the frame's inputs — raw sample pairs (reflect-padded at the note ends,
    // GOOFER.py:358-360) and the envelope row — are fetched at the head of the frame that uses them.  Round 3 held the NEXT
    // frame's 25 values in registers across the whole frame (two waves per SIMD: 221 registers); at that occupancy the prefetch
    // bought nothing (0.525 ms with and without), and without it, with the window sums in an LDS table, the kernel fits a third
    // wave (0.525 -> 0.47 ms: a lone wave spends three quarters of its time waiting on its own dependency chain).
    float2 raw[R];
    float ev[PER];
    auto fetch = [&](int idx) {
        const int nn = FB_GET(fb, n, idx);
        const int start = FB_GET(fb, t, idx) * HOP - M;            // first sample of the frame, un-padded coordinates
        const float *xs = pulse + FB_BASE(fb, idx);
        if (start >= 0 && start + 2 * M <= nn) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                raw[r] = make_float2(xs[start + 2 * m], xs[start + 2 * m + 1]);
            }
        } else {
            // numpy 'reflect' as a periodic map (n == 1: 'edge'); 32-bit: notes are far shorter than 2^31 samples
            const int period = nn > 1 ? 2 * (nn - 1) : 1;
            auto refl = [&](int i) {
                if (nn <= 1) return 0;
                if (i < 0 || i >= period) {
                    i %= period;
                    if (i < 0) i += period;
                }
                return i < nn ? i : period - i;
            };
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int m = lane + WAVE * r;
                const float a = nn > 0 ? xs[refl(start + 2 * m)] : 0.f;
                const float b = nn > 0 ? xs[refl(start + 2 * m + 1)] : 0.f;
                raw[r] = make_float2(a, b);
            }
        }
        const int srow = FB_GET(fb, src, idx);                      // (bit 31: the note does not warp — read the assembled row)
        const float *er = (srow < 0 ? env_plain : env) + (int64_t)(srow & 0x7fffffff) * ld;
#pragma unroll
        for (int i = 0; i < PER; ++i) ev[i] = er[lane + WAVE * i < B ? lane + WAVE * i : B - 1];
    };

    float2 carry[R - G];
#pragma unroll
    for (int r = 0; r < R - G; ++r) carry[r] = make_float2(0.f, 0.f);
    int apply_bright = 0, cut_below = 0;
    float ec[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ec[q] = g_edge[q * WAVE + lane];

    for (int64_t f = fs; f < f1; ++f) {
        float2 X[PER];
        float evc[PER];
        if (!fb.holds(f)) {
            fb.load(f, total_frames, frame_note, frame_off, sample_off, row_src, picks, lane);
            if (env_plain) fb.mark_plain(params, have_formants != 0);
        }
        const int idx = (int)(f - fb.blk0);
        fetch(idx);
        const int t = FB_GET(fb, t, idx);
        if (FB_GET(fb, note, idx) != w.note) {
            w.enter_note(fb, idx);
            const goofer_note_params &p = params[w.note];
            apply_bright = p.apply_brightness;
            cut_below = p.cut_below_f0;
        }
        const float f0f = FB_GETF(fb, f0, idx);
        const bool voiced = apply_bright && FB_GETF(fb, mk, idx) > 0.f;
        {
            // 1. windowed frame -> complex FFT; the lane's points Z[lane + 64 t] stay in registers
            float2 z[R];
            {
                float2 v[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int m = lane + WAVE * r;
                    const float2 wn = *reinterpret_cast<const float2 *>(w.win + 2 * m);
                    v[r] = make_float2(raw[r].x * wn.x, raw[r].y * wn.y);
                }
                wave_fft_keep_tab<M>(v, w.buf, w.tw1, w.tw2, lane, z);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) evc[i] = ev[i];

            // 2. even/odd split: X[k] = (Z[k] + conj Z[M-k])/2 - i/2 e^{-i pi k/M} (Z[k] - conj Z[M-k]); mirrored points through LDS
#pragma unroll
            for (int r = 0; r < R; ++r) w.buf[lds_pad(lane + WAVE * r)] = z[r];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = lane + WAVE * r;
                X[r] = rfft_post(z[r], w.buf[lds_pad((M - k) & (M - 1))], w.wct[k]);   // the split twiddle is the conjugate of the inverse one
            }
            X[R] = make_float2(z[0].x - z[0].y, 0.f);                // Nyquist bin from Z[0] (lane 0)
            wave_lds_sync();
        }

        // 3. shaping (GOOFER.py:1102-1144); 1 / max(|S| + 1e-8) commutes with the linear chain and is applied by k_note_finish
        const bool hp_low_only = t_fq[WAVE] - f0f > 100.0f;
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = lane + WAVE * i;
            if (k >= B) { X[i] = make_float2(0.f, 0.f); continue; }
            float2 s = X[i];
            if (cut_below && (i == 0 || !hp_low_only)) {          // bins 64 and up: the factor is exactly 1.0f (see k_noise_stems)
                const float h = hp_mask(t_fq[k], f0f);
                s.x *= h; s.y *= h;
            }
            mx = fmaxf(mx, s.x * s.x + s.y * s.y);                 // |s|^2: the square root is taken once, of the maximum
            const float bo = t_bo[k];
            s.x = (s.x * evc[i]) * bo;
            s.y = (s.y * evc[i]) * bo;
            if (voiced) {
                const float b = t_br[k];
                s.x *= b; s.y *= b;
            }
            X[i] = s;
        }
        mx = __builtin_amdgcn_sqrtf(wave_max(mx)) + 1e-8f;          // max(|s| + 1e-8) = sqrt(max |s|^2) + 1e-8: sqrt is monotone
        if (lane == 0) atomic_max_pos(note_mag + w.note, mx);
        if (voiced && !td_blur) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = lane + WAVE * i;
                if (k < B) w.buf[k] = X[i];
            }
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = lane + WAVE * i;
                if (k < B) X[i] = blur5f(w.buf, k, B, t5);
            }
            wave_lds_sync();
        }

        // 4. inverse transform + overlap-add; hop t leaves un-normalised by the note's spectrum maximum
        float2 e[G];
        if (voiced && td_blur) w.blur_edges(X, ec, t5[0], t5[1]);
        w.inverse_ola(X, t, carry, e, (voiced && td_blur) ? w.wsv : w.wsc);   // voiced: the bin blur rides on the window
        // hop t lies inside the note, every frame over it exists and the note goes on: no bounds test per sample, no flush — the
        // same quotients by the same per-lane window sums (most hops of a note; the general loop below has the rest)
        const int p0t = t * HOP - M;
        if (f >= f0 && w.interior(t) && t != w.T - 1 && p0t >= 0 && p0t + HOP <= w.n && p0t + HOP <= w.out_len) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i0 = p0t + 2 * (lane + WAVE * g);
                float x[2] = {e[g].x, e[g].y};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float ws = w.ws_of(g, c);
                    if (ws > 1e-9f) x[c] = div_by(x[c], ws, w.rws_of(g, c));
                }
                *reinterpret_cast<float2_u *>(harm + w.base + i0) = make_float2(x[0], x[1]);
            }
        } else if (f >= f0) {
            for (int h = t;;) {
                const int p0 = h * HOP - M;
                const bool inner = w.interior(h);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int i0 = p0 + 2 * (lane + WAVE * g);
                    if (i0 < 0 || i0 >= w.n) continue;
                    float x[2] = {e[g].x, e[g].y};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (i0 + c < w.out_len) {
                            if (inner) {
                                const float ws = w.ws_of(g, c);
                                if (ws > 1e-9f) x[c] = div_by(x[c], ws, w.rws_of(g, c));
                            } else {
                                const float ws = w.partial_ws(h, g, c, g_win);
                                if (ws > 1e-9f) x[c] /= ws;
                            }
                        } else {
                            x[c] = 0.f;                              // zero tail of istft (GOOFER.py:409-412)
                        }
                    }
                    if (i0 + 1 < w.n) *reinterpret_cast<float2_u *>(harm + w.base + i0) = make_float2(x[0], x[1]);
                    else harm[w.base + i0] = x[0];
                }
                ++h;
                if (t != w.T - 1 || h * HOP - M >= w.n) break;
#pragma unroll
                for (int g = 0; g < G; ++g) e[g] = carry[g];        // flush: the sums still in the registers, then zeros
            }
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// One workgroup per note, two passes over its three stems: harm / max|S|, peak of harm + uv + breath, then
// gain = (1 / peak)^normalize on everything (GOOFER.py:1121, 1208-1218) and the V/B/U mix (SillySampler.py:1142-1151).  Same
// operations in the same order as k_irfft_ola3's output stage + k_apply_gain.
// For the first FIN_KEEP * 4096 samples of a note (1.1 s at 44.1 kHz) the second pass takes harm / max|S| and the unvoiced stem from
// the REGISTERS the first pass left them in (2 x 48 per thread at 1024 threads) and re-reads only the breath stem (later samples: all three): with 256
// notes in flight nothing of the first pass is still in L2 (the pass ran at 1.4 GB per 1024 notes; now 1.0).
constexpr int FIN_THREADS = 1024;
constexpr int FIN_KEEP = 12;              // float4 per stem and thread kept across the peak reduction

__global__ __launch_bounds__(FIN_THREADS) void k_note_finish(float *__restrict__ harm, float *__restrict__ uv, float *__restrict__ bre,
                                                             float *__restrict__ rec, float *__restrict__ mix,
                                                             const int64_t *__restrict__ sample_off,
                                                             const goofer_note_params *__restrict__ params,
                                                             const float *__restrict__ note_mag, float *__restrict__ note_peak,
                                                             int write_stems, int lds_rows, const unsigned char *__restrict__ hopz,
                                                             const int64_t *__restrict__ frame_off, int hz_m, int hz_shift)
{
    __shared__ float s_red[FIN_THREADS / WAVE];
    extern __shared__ __align__(16) unsigned char fin_smem[];
    float4 *sb4 = reinterpret_cast<float4 *>(fin_smem);     // [lds_rows][FIN_THREADS]: the breath stem's first rows, parked between the passes
    const int note = blockIdx.x;
    const int64_t base = sample_off[note];
    const int n = (int)(sample_off[note + 1] - base);
    const float mag = note_mag[note], rmag = 1.0f / mag;
    float *h_ = harm + base, *u_ = uv + base, *b_ = bre + base;
    // 16-byte accesses on the aligned body [a0, a1) of the note; the ragged head and tail go sample by sample
    int a0 = (int)((4 - (base & 3)) & 3);
    if (a0 > n) a0 = n;
    const int a1 = a0 + ((n - a0) & ~3);
    const bool vec = ((((uintptr_t)harm | (uintptr_t)uv | (uintptr_t)bre | (uintptr_t)rec | (uintptr_t)mix) & 15) == 0);
    // The noise walker does not store a stem over a hop on which its gain is exactly zero (hopz, see k_noise_stems): bit 0 of a
    // hop's byte: the unvoiced stem is zero there and ABSENT (the array holds anything), bit 1: the breath stem.  Sample i lies in
    // hop (i + n_fft/2) / hop.  A 16-byte group may straddle two hops: zpair(i) = the bytes of the hops of samples i and i + 3
    // (bits 0-1 and 2-3), stem4 loads a group and puts zeros where its hop says so.
    const unsigned char *hz = hopz ? hopz + (frame_off[note] + 3 * (int64_t)note) : nullptr;
    auto zbits = [&](int i) -> unsigned { return hz ? (unsigned)hz[(i + hz_m) >> hz_shift] : 0u; };
    auto zpair = [&](int i) -> unsigned {
        if (!hz) return 0u;
        const int h0 = (i + hz_m) >> hz_shift, h1 = (i + 3 + hz_m) >> hz_shift;
        const unsigned z0 = hz[h0] & 3u;
        return z0 | ((h1 != h0 ? (hz[h1] & 3u) : z0) << 2);
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stem4 = [&](const float *p, int i, unsigned zp, unsigned bit) -> float4 {
        const bool s0 = (zp & bit) != 0, s1 = ((zp >> 2) & bit) != 0;
        if (s0 && s1) return zero4;
        float4 v = *reinterpret_cast<const float4 *>(p + i);
        if (s0 || s1) {
            const int split = ((((i + hz_m) >> hz_shift) + 1) << hz_shift) - hz_m - i;   // samples of the group in its first hop (1..3)
            v.x = s0 ? 0.f : v.x;
            v.y = (1 < split ? s0 : s1) ? 0.f : v.y;
            v.z = (2 < split ? s0 : s1) ? 0.f : v.z;
            v.w = s1 ? 0.f : v.w;
        }
        return v;
    };

    float pk = 0.f;
    const bool keep = vec;                                              // workgroup-uniform; rows past FIN_KEEP take the re-reading loops
    const int a_keep = a0 + 4 * FIN_THREADS * FIN_KEEP;                 // first sample past the kept rows
    float4 kh[FIN_KEEP], ku[FIN_KEEP];
    uint64_t zall = 0;                                                  // four bits per kept row: fetched up front, off the loads' chain
    if (keep && hz) {
#pragma unroll
        for (int q = 0; q < FIN_KEEP; ++q) {
            const int i = a0 + 4 * ((int)threadIdx.x + FIN_THREADS * q);
            if (i < a1) zall |= (uint64_t)zpair(i) << (4 * q);
        }
    }
    if (keep) {
#pragma unroll
        for (int q = 0; q < FIN_KEEP; ++q) {
            const int i = a0 + 4 * ((int)threadIdx.x + FIN_THREADS * q);
            kh[q] = ku[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < a1) {
                const unsigned z = (unsigned)(zall >> (4 * q)) & 15u;
                const float4 h = *reinterpret_cast<const float4 *>(h_ + i);
                const float4 u = stem4(u_, i, z, 1u);
                const float4 b = stem4(b_, i, z, 2u);
                kh[q] = make_float4(div_by(h.x, mag, rmag), div_by(h.y, mag, rmag), div_by(h.z, mag, rmag), div_by(h.w, mag, rmag));
                ku[q] = u;
                if (q < lds_rows) sb4[q * FIN_THREADS + (int)threadIdx.x] = b;     // (read back by this thread only)
                pk = fmaxf(pk, fabsf((kh[q].x + u.x) + b.x));
                pk = fmaxf(pk, fabsf((kh[q].y + u.y) + b.y));
                pk = fmaxf(pk, fabsf((kh[q].z + u.z) + b.z));
                pk = fmaxf(pk, fabsf((kh[q].w + u.w) + b.w));
            }
            // (one group of three loads in flight per thread — 48 KB per CU at sixteen waves — not all twelve: the breath values
            // are transient, and hoisted together they push the kept ones out to scratch)
            if (q % 2 == 1) asm volatile("" ::: "memory");
        }
        for (int i = a_keep + 4 * (int)threadIdx.x; i < a1; i += 4 * FIN_THREADS) {
            const unsigned z = zpair(i);
            const float4 h = *reinterpret_cast<const float4 *>(h_ + i);
            const float4 u = stem4(u_, i, z, 1u);
            const float4 b = stem4(b_, i, z, 2u);
            pk = fmaxf(pk, fabsf((div_by(h.x, mag, rmag) + u.x) + b.x));
            pk = fmaxf(pk, fabsf((div_by(h.y, mag, rmag) + u.y) + b.y));
            pk = fmaxf(pk, fabsf((div_by(h.z, mag, rmag) + u.z) + b.z));
            pk = fmaxf(pk, fabsf((div_by(h.w, mag, rmag) + u.w) + b.w));
        }
    }
    for (int i = (int)threadIdx.x; i < n; i += FIN_THREADS) {
        if (vec && i >= a0 && i < a1) continue;
        const unsigned z = zbits(i);
        pk = fmaxf(pk, fabsf((div_by(h_[i], mag, rmag) + ((z & 1u) ? 0.f : u_[i])) + ((z & 2u) ? 0.f : b_[i])));
    }
    pk = wave_max(pk);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = pk;
    __syncthreads();
    float peak = 0.f;
#pragma unroll
    for (int q = 0; q < FIN_THREADS / WAVE; ++q) peak = fmaxf(peak, s_red[q]);
    if (threadIdx.x == 0) note_peak[note] = peak;

    const goofer_note_params &p = params[note];
    const float pk12 = peak + 1e-12f;                                 // fp32 add, like np.float32 + 1e-12
    const float gain = peak_gain(pk12, p.normalize);
    const float m_h = p.mix_harm, m_b = p.mix_breath, m_u = p.mix_unvoiced, vol = p.volume;
    auto one = [&](float h, float u, float b, float &ho, float &uo, float &bo, float &ro, float &mo) {
        h = div_by(h, mag, rmag);
        const float comb = (h + u) + b;
        ho = h * gain; uo = u * gain; bo = b * gain;
        ro = comb * gain;
        mo = ((ho * m_h + bo * m_b) + uo * m_u) * vol;
    };
    // (the quotient harm / max|S| of the kept samples is already taken)
    auto one_kept = [&](float h, float u, float b, float &ho, float &uo, float &bo, float &ro, float &mo) {
        const float comb = (h + u) + b;
        ho = h * gain; uo = u * gain; bo = b * gain;
        ro = comb * gain;
        mo = ((ho * m_h + bo * m_b) + uo * m_u) * vol;
    };
    if (keep) {
#pragma unroll
        for (int q = 0; q < FIN_KEEP; ++q) {
            const int i = a0 + 4 * ((int)threadIdx.x + FIN_THREADS * q);
            if (i < a1) {
                const float4 h = kh[q], u = ku[q];
                const float4 b = q < lds_rows ? sb4[q * FIN_THREADS + (int)threadIdx.x] : stem4(b_, i, (unsigned)(zall >> (4 * q)) & 15u, 2u);
                float4 ho, uo, bo, ro, mo;
                one_kept(h.x, u.x, b.x, ho.x, uo.x, bo.x, ro.x, mo.x);
                one_kept(h.y, u.y, b.y, ho.y, uo.y, bo.y, ro.y, mo.y);
                one_kept(h.z, u.z, b.z, ho.z, uo.z, bo.z, ro.z, mo.z);
                one_kept(h.w, u.w, b.w, ho.w, uo.w, bo.w, ro.w, mo.w);
                if (write_stems & 1) {
                    *reinterpret_cast<float4 *>(h_ + i) = ho;
                    *reinterpret_cast<float4 *>(u_ + i) = uo;
                    *reinterpret_cast<float4 *>(b_ + i) = bo;
                }
                if (rec) store_f4(rec + base + i, ro, (write_stems & 2) != 0);
                if (mix) store_f4(mix + base + i, mo, (write_stems & 2) != 0);
            }
            if (q % 2 == 1) asm volatile("" ::: "memory");
        }
        for (int i = a_keep + 4 * (int)threadIdx.x; i < a1; i += 4 * FIN_THREADS) {
            const unsigned z = zpair(i);
            const float4 h = *reinterpret_cast<const float4 *>(h_ + i);
            const float4 u = stem4(u_, i, z, 1u);
            const float4 b = stem4(b_, i, z, 2u);
            float4 ho, uo, bo, ro, mo;
            one(h.x, u.x, b.x, ho.x, uo.x, bo.x, ro.x, mo.x);
            one(h.y, u.y, b.y, ho.y, uo.y, bo.y, ro.y, mo.y);
            one(h.z, u.z, b.z, ho.z, uo.z, bo.z, ro.z, mo.z);
            one(h.w, u.w, b.w, ho.w, uo.w, bo.w, ro.w, mo.w);
            if (write_stems & 1) {
                *reinterpret_cast<float4 *>(h_ + i) = ho;
                *reinterpret_cast<float4 *>(u_ + i) = uo;
                *reinterpret_cast<float4 *>(b_ + i) = bo;
            }
            if (rec) store_f4(rec + base + i, ro, (write_stems & 2) != 0);
            if (mix) store_f4(mix + base + i, mo, (write_stems & 2) != 0);
        }
    }
    for (int i = (int)threadIdx.x; i < n; i += FIN_THREADS) {
        if (vec && i >= a0 && i < a1) continue;
        const unsigned z = zbits(i);
        float ho, uo, bo, ro, mo;
        one(h_[i], (z & 1u) ? 0.f : u_[i], (z & 2u) ? 0.f : b_[i], ho, uo, bo, ro, mo);
        if (write_stems & 1) { h_[i] = ho; u_[i] = uo; b_[i] = bo; }
        if (rec) rec[base + i] = ro;
        if (mix) mix[base + i] = mo;
    }
}

// per-frame picks of the per-sample arrays: x[::hop] edge-padded to the frame count (GOOFER.py:1104-1106)
__global__ void k_frame_picks(const int64_t *__restrict__ frame_off, const int *__restrict__ frame_note, int64_t total_frames,
                              const int64_t *__restrict__ sample_off, const float *__restrict__ f0, const float *__restrict__ mask,
                              int hop, float2 *__restrict__ picks)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total_frames) return;
    const int note = frame_note[f];
    const int64_t t = f - frame_off[note];
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
    float2 pv = make_float2(0.f, 0.f);
    if (n > 0) {
        const int64_t at = pick_index(t, n, hop);
        pv = make_float2(f0[base + at], mask[base + at]);
    }
    picks[f] = pv;
}

// ---------------------------------------------------------------------------------------------
// Frames per wave.  A run replays `halo` frames it does not emit, so longer runs waste less; and every wave does the same
// work, so the grid should fill the device a whole number of times: with k rounds of `slots` waves,
// run = ceil(frames / (k slots)), k the smallest that keeps a run at or below 128 frames.
static int run_length(int64_t total_frames, int slots)
{
    const int64_t rounds = (total_frames + (int64_t)slots * 128 - 1) / ((int64_t)slots * 128);
    const int64_t fit = (total_frames + rounds * slots - 1) / (rounds * slots);
    return (int)(fit > 32 ? fit : 32);
}

bool stems_supported(const goofer_plan_t &p) { return p.n_fft == 1024 && p.hop * 4 == p.n_fft; }

int launch_frame_picks(goofer_ctx *ctx, const int64_t *frame_off, const int *frame_note, int64_t F, const int64_t *sample_off,
                       const float *f0, const float *mask, float2 *picks, hipStream_t st)
{
    if (F <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_frame_picks, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, frame_off, frame_note, F, sample_off, f0, mask,
                       ctx->plan.hop, picks);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_noise_stems(goofer_ctx *ctx, const float *env, int ld, const int64_t *row_src, const float *phi, int64_t F,
                       const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, const float2 *picks,
                       const goofer_note_params *params, uint64_t seed, bool preblurred, const double *short_s, const double *steps,
                       float *uv, float *bre, unsigned char *hopz, hipStream_t st)
{
    if (F <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    if (!stems_supported(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the stem walkers need n_fft 1024 and hop == n_fft / 4");
    if ((ld & 3) || ((uintptr_t)env & 15)) return goofer_fail(ctx, GOOFER_EINVAL, "envelope rows must be 16-byte aligned");
    if (!stem_taps_match(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the plan's blur taps differ from the walkers' literals");
    constexpr int M = 512;
    const void *fn = phi ? (const void *)k_noise_stems<M, true> : (const void *)k_noise_stems<M, false>;
    size_t lds = stem_cfg<M>::lds_bytes<2, false>();
    int rc, slots = 0;
    if ((rc = kernel_allow_max_lds(ctx, fn))) return rc;
    if ((rc = kernel_resident_waves(ctx, fn, lds, &slots))) return rc;
    const int run = run_length(F, slots);
    const int64_t runs = (F + run - 1) / run;
    const dim3 grid((unsigned)((runs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
    noise_args A;
    A.env = env; A.phi = phi; A.uv = uv; A.bre = bre; A.hopz = hopz; A.short_s = short_s;
    A.ld = ld; A.mode = (preblurred ? 0 : 1) | (ctx->skip_zero ? 0 : 2) | (ctx->td_blur ? 4 : 0); A.run = run;
    A.total_frames = F; A.seed = seed; A.row_src = row_src; A.frame_note = frame_note; A.frame_off = frame_off;
    A.sample_off = sample_off; A.picks = picks; A.params = params; A.steps = steps; A.freqs = p.freqs; A.bright = p.bright_b;
    A.g_tw = p.tw_full; A.g_twh = p.tw_half; A.g_win = p.window; A.g_winb = p.window_blur; A.g_edge = p.blur_edge;
    if (phi) hipLaunchKernelGGL((k_noise_stems<M, true>), grid, dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_noise_stems<M, false>), grid, dim3(256), lds, st, A);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// env: warped rows, one per frame (row_src == nullptr), or source rows through row_src when nothing warps; env_plain != nullptr:
// `env` holds the warped copies of the warping notes only, the other notes' rows are read from env_plain (same row index)
int launch_harm_stem(goofer_ctx *ctx, const float *pulse, const float *env, const float *env_plain, bool have_formants, int ld,
                     const int64_t *row_src, int64_t F,
                     const int *frame_note, const int64_t *frame_off, const int64_t *sample_off, const float2 *picks,
                     const goofer_note_params *params, float *harm, float *note_mag, hipStream_t st)
{
    if (F <= 0) return GOOFER_OK;
    const goofer_plan_t &p = ctx->plan;
    if (!stems_supported(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the stem walkers need n_fft 1024 and hop == n_fft / 4");
    if (!stem_taps_match(p)) return goofer_fail(ctx, GOOFER_EINVAL, "the plan's blur taps differ from the walkers' literals");
    constexpr int M = 512;
    const void *fn = (const void *)k_harm_stem<M>;
    size_t lds = stem_cfg<M>::lds_bytes<3, true>();
    int rc, slots = 0;
    if ((rc = kernel_allow_max_lds(ctx, fn))) return rc;
    if ((rc = kernel_resident_waves(ctx, fn, lds, &slots))) return rc;
    const int run = run_length(F, slots);
    const int64_t runs = (F + run - 1) / run;
    hipLaunchKernelGGL(k_harm_stem<M>, dim3((unsigned)((runs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), dim3(256), lds, st, pulse, env,
                       env_plain, have_formants ? 1 : 0, ld, row_src, F, frame_note, frame_off, sample_off, picks, params, p.freqs, p.boost, p.bright_h, harm,
                       note_mag, run, p.tw_full, p.tw_half, p.window, p.window_blur, p.blur_edge, ctx->td_blur ? 1 : 0);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// hopz / frame_off: the noise walker's per-hop bytes (stems it did not store because they are exactly zero), or nullptr: every
// sample of every stem is there
int launch_note_finish(goofer_ctx *ctx, float *harm, float *uv, float *bre, float *rec, float *mix, const int64_t *sample_off,
                       int n_notes, const goofer_note_params *params, const float *note_mag, float *note_peak, bool write_stems,
                       const unsigned char *hopz, const int64_t *frame_off, hipStream_t st)
{
    if (n_notes <= 0) return GOOFER_OK;
    // One 1024-thread workgroup fills a CU's register file (16 waves x 128 VGPRs), so its LDS is the workgroup's own: the first
    // rows of the breath stem wait there between the passes (9 rows x 16 KB)
    const int rows = 9;
    const size_t lds = (size_t)rows * FIN_THREADS * sizeof(float4);
    if (lds > 48 * 1024)
        if (int arc = kernel_allow_max_lds(ctx, (const void *)k_note_finish, FIN_KEEP * FIN_THREADS * (int)sizeof(float4) > 159 * 1024 ? 159 * 1024 : FIN_KEEP * FIN_THREADS * (int)sizeof(float4))) return arc;   // (beside 64 B of static LDS)
    hipLaunchKernelGGL(k_note_finish, dim3((unsigned)n_notes), dim3(FIN_THREADS), lds, st, harm, uv, bre, rec, mix, sample_off, params,
                       note_mag, note_peak, (write_stems ? 1 : 0) | 2 /* mix / rec: non-temporal stores */, rows, hopz, frame_off,
                       ctx->plan.n_fft / 2, 31 - __builtin_clz((unsigned)ctx->plan.hop));
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
