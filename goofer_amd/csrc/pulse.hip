// LF glottal pulse train for gfx950 — replaces gf.pulse_train_numba (GOOFER.py:473-554).
//
// The onset positions are a discontinuous function of a strictly sequential fp64 sum
// (`total_phase += f0[i]/sr`, GOOFER.py:491-493): re-associating it moves onsets by a sample
// (SURVEY.md §7.3-1).  So the work is split three ways:
//   (k_phase_inc    inc[i] = (double)f0[i] / sr as a pass of its own: the sub-harmonic layer's tracker still uses one)
//   k_pulse_onsets_scan  one WAVE per note walks its increments in order (nothing but the dependent fp64 adds) and
//                   extracts the onset samples from the partial sums chunk by chunk, in parallel; k_onset_finish
//                   completes the onset list (sample, T0, period, running max of sample+T0)
//   k_pulse_place   fully parallel gather: every output sample sums, in ascending onset order, the
//                   LF shapes that cover it — no atomics, same fp32 accumulation order as the
//                   reference's `pulse[j] += cache[k]`
// Shapes are evaluated on the fly in fp64 (numba's typing) and normalised by a per-T0 peak table.
#include "common.h"

#define PT_PI 3.141592653589793

// un-normalised LF shape sample k of a T0-sample pulse with period T (GOOFER.py:509-519), rounded
// to fp32 like the reference's `buf[j] = ...` store.  lf: the model's Ra, Rg, Rk (gf.pulse_train_numba's keyword arguments;
// 0.02 / 1.7 / 0.8 is what gf.synthesize passes, GOOFER.py:1074)
__device__ __forceinline__ float lf_raw(int k, int T0, double T, const lf_model &lf)
{
    double ti = ((double)k * T) / (double)T0;
    double Tp = lf.ra * T;
    double Tc = Tp + lf.rk * (T - Tp);
    double v;
    if (ti < Tp) {
        double s = sin(PT_PI * ti / (2.0 * Tp + 1e-12));
        v = s * s;
    } else if (ti < Tc) {
        double tau = (ti - Tp) / (Tc - Tp + 1e-12);
        v = exp(-lf.rg * tau) * cos(PT_PI * tau / 2.0);
    } else {
        v = 0.0;
    }
    return (float)v;
}

// peak[T0] = max_k |buf[k]| for the nominal period T = T0/sr; one block per T0
__global__ __launch_bounds__(256) void k_pulse_peak(float *__restrict__ peak, double sr, const lf_model lf)
{
    __shared__ float red[4];
    int T0 = blockIdx.x;
    float m = 0.f;
    if (T0 >= 3) {
        double T = (double)T0 / sr;
        for (int k = threadIdx.x; k < T0; k += blockDim.x) m = fmaxf(m, fabsf(lf_raw(k, T0, T, lf)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) peak[T0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (T0 == 0 && threadIdx.x == 0) {
        double *tail = reinterpret_cast<double *>(peak + 8194);   // (the model behind the table: goofer_debug_table and the tests read it; no kernel does)
        tail[0] = lf.ra; tail[1] = lf.rg; tail[2] = lf.rk;
    }
}

// The normalised pulse of length T0, tabulated at the nominal period T = T0/sr.  The LF shape depends on the period
// only through ratios (ti/Tp = k / (Ra T0), tau = (k/T0 - Ra) / (Rk (1 - Ra))), so an onset's true period
// 1/f0 moves the fp64 value by ~1e-11 relative (the 1e-12 guards) — below fp32 resolution except on rare ties.
static_assert((int64_t)PULSE_TAB_MAX * (PULSE_TAB_MAX + 1) / 2 < (1ll << 31), "32-bit row offsets");
__device__ __forceinline__ int pulse_tab_row(int T0) { return T0 * (T0 - 1) / 2 - 3; }

__global__ __launch_bounds__(256) void k_pulse_shape_table(float *__restrict__ tab, const float *__restrict__ peak, double sr, const lf_model lf)
{
    const int T0 = blockIdx.x + 3;
    const double T = (double)T0 / sr;
    const double m = (double)peak[T0];
    float *row = tab + pulse_tab_row(T0);
    for (int k = threadIdx.x; k < T0; k += blockDim.x) {
        const float raw = lf_raw(k, T0, T, lf);
        row[k] = m > 0.0 ? (float)((double)raw / m) : raw;
    }
}

int launch_pulse_peak(goofer_ctx *ctx, float *peak, double sr, hipStream_t st)
{
    hipLaunchKernelGGL(k_pulse_peak, dim3(8193), dim3(256), 0, st, peak, sr, ctx->plan.lf);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

size_t pulse_shape_table_floats() { return (size_t)((int64_t)(PULSE_TAB_MAX + 1) * PULSE_TAB_MAX / 2 - 3); }

int launch_pulse_shape_table(goofer_ctx *ctx, float *tab, const float *peak, double sr, hipStream_t st)
{
    hipLaunchKernelGGL(k_pulse_shape_table, dim3(PULSE_TAB_MAX - 2), dim3(256), 0, st, tab, peak, sr, ctx->plan.lf);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

__global__ __launch_bounds__(256) void k_phase_inc(const float *__restrict__ f0, int64_t n, double sr, float scale,
                                                   double *__restrict__ inc)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) inc[g] = (double)(f0[g] * scale) / sr;
}

struct onset_t {
    int32_t i;        // sample index inside the note
    int32_t T0;       // pulse length in samples
    int32_t end_max;  // max over onsets <= this one of (i + T0): monotone, bounds the look-back
    int32_t pad;
    double T;         // period used for the shape
};

// One wave per note.  The walk itself is wave-uniform (every lane carries the same phase), which lets
// the memory side be fully parallel: the wave fetches 512-sample chunks of increments with coalesced
// vector loads (next chunk prefetched into registers while the current one is walked), parks them in
// LDS, and reads them back as uniform 16-byte broadcasts.  A lone wave issues roughly one instruction
// per 4 cycles whatever its kind, so the hot loop is kept minimal.  The 16 partial sums of a block ARE
// the reference's sequential phases (same additions, same order: GOOFER.py:491), so events are
// bit-exact.  Zero padding of the last chunk adds +0.0, which leaves the phase unchanged.
#define OC 512   // samples per chunk (64 lanes x 8)
#define OB 16    // samples per walk block

// Placement: a workgroup is 4 waves = 4 notes (one per SIMD of a CU), and the launcher pads the
// dynamic LDS request so that only ceil(blocks/256) workgroups fit on a CU — otherwise the dispatcher
// packs many of these latency-bound waves onto a few CUs and they time-slice one SIMD.
//
// k_pulse_onsets_wrap is the sub-harmonic layer's tracker (GOOFER.py:693-696): an event fires when the phase reaches 1
// and the phase then drops by 1.0, so the events feed back into the chain.  Increments are >= 0 in practice, so a
// block of 16 can only hold an event if its last partial sum reaches 1; only then (or when the chunk holds a negative
// increment, found in parallel at fetch time) is the block re-walked sample by sample.
__global__ __launch_bounds__(256) void k_pulse_onsets_wrap(const double *__restrict__ inc, const int64_t *__restrict__ sample_off,
                                                      int n_notes, int32_t *__restrict__ onset_idx,
                                                      int32_t *__restrict__ onset_cnt, int32_t *__restrict__ overflow,
                                                      const unsigned char *__restrict__ note_on)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // this wave is one long dependent chain: when another kernel shares the SIMD (the side stream's noise spectra),
    // let the arbiter issue it first
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int note = blockIdx.x * 4 + wv;
    if (note >= n_notes) return;                              // whole wave; no block barrier below
    if (note_on && !note_on[note]) {
        if (lane == 0) onset_cnt[note] = 0;
        return;
    }
    double (*tile)[OC] = reinterpret_cast<double (*)[OC]>(smem) + 2 * wv;
    const int64_t base = sample_off[note];
    const int64_t n = sample_off[note + 1] - base;
    // a slot per sample (+ 16): the tracker fires at most once per sample — and on every sample once the increment passes 1
    const int64_t obase = base + 16 * (int64_t)note;
    const int32_t cap = (int32_t)(n + 16);
    const double *__restrict__ a = inc + base;
    int32_t *__restrict__ out = onset_idx + obase;
    double phase = 0.0;
    int32_t cnt = 0;
    // Event samples are parked in a 64-entry LDS queue and written out once per chunk as one coalesced store, issued
    // right after the next prefetch: a global store from inside the walk would sit in vmcnt and stall the wave for
    // its acknowledgement at the next chunk boundary.
    int32_t *queue = reinterpret_cast<int32_t *>(smem + 4 * 2 * OC * sizeof(double)) + WAVE * wv;
    int pend = 0;
    auto flush = [&]() {
        const int32_t at = cnt - pend + lane;
        if (lane < pend && at < cap) out[at] = queue[lane];
        pend = 0;
    };
    auto push = [&](int32_t idx) {
        if (lane == 0) queue[pend] = idx;
        ++pend;
        ++cnt;
        if (pend == WAVE) {
            wave_lds_sync();
            flush();
        }
    };

    double r[8];
    // NOTE the branch is wave-uniform on purpose: a per-lane if/else writing the same registers makes
    // the compiler drain vmcnt(0) between the two arms, i.e. right after issuing the prefetch.
    auto fetch = [&](int64_t c0) {
        const int64_t s = c0 + (int64_t)lane * 8;
        if (c0 + OC <= n) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = a[s + k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t i = s + k;
                const double v = a[i < n ? i : n - 1];
                r[k] = i < n ? v : 0.0;
            }
        }
    };
    fetch(0);
    int buf = 0;
    for (int64_t c0 = 0; c0 < n; c0 += OC, buf ^= 1) {
        double *t = tile[buf];
        bool r_neg = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t[lane * 8 + k] = r[k];
            r_neg |= r[k] < 0.0;
        }
        const bool chunk_neg = __any(r_neg);
        wave_lds_sync();
        if (c0 + OC < n) fetch(c0 + OC);                    // in flight during the walk below
        flush();                                            // the previous chunk's onsets
        const int64_t left = n - c0;
        const int blocks = left >= OC ? OC / OB : (int)((left + OB - 1) / OB);
        // one walk block from registers: chain, group test, rare per-sample checks
        auto walk = [&](const double (&x)[OB], int g) {
            double ps[OB];
            ps[0] = phase + x[0];
#pragma unroll
            for (int k = 1; k < OB; ++k) ps[k] = ps[k - 1] + x[k];
            if (__any((ps[OB - 1] >= 1.0) || chunk_neg)) {
                const int32_t i0 = (int32_t)c0 + g * OB;
                double ph = phase;
#pragma unroll
                for (int j = 0; j < OB; ++j) {
                    ph += x[j];
                    if (__any(ph >= 1.0)) {
                        push(i0 + j);
                        ph -= 1.0;
                    }
                }
                phase = ph;
            } else {
                phase = ps[OB - 1];
            }
        };
        auto load = [&](double (&x)[OB], int g) {
            const double *q = t + (g < blocks ? g : blocks - 1) * OB;
#pragma unroll
            for (int k = 0; k < OB; ++k) x[k] = q[k];
        };
        // ping-pong two register sets: the LDS broadcasts of block g+1 are in flight during block g's adds
        double xa[OB], xb[OB];
        load(xa, 0);
#pragma unroll 1
        for (int g = 0; g < blocks; g += 2) {
            load(xb, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            walk(xa, g);
            load(xa, g + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < blocks) walk(xb, g + 1);
        }
    }
    wave_lds_sync();
    flush();
    if (lane == 0) {
        onset_cnt[note] = cnt < cap ? cnt : cap;
        if (cnt > cap) atomicMax(overflow, note + 1);          // reported at the next synchronising call (goofer_check)
    }
}

// Inclusive wave scans of non-negative int32 values on the DPP path (row shifts inside each 16-lane row, then the two
// row broadcasts): ~8 VALU instructions instead of six LDS round trips.
template <typename Op>
__device__ __forceinline__ int32_t wave_scan_incl(int32_t x, Op op)
{
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));   // row_shr:1
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));   // row_shr:2
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));   // row_shr:4
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));   // row_shr:8
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return x;
}

// The same scan inside segments of LPN consecutive lanes (16: one DPP row, 32: two rows, 64: the wave): the row-broadcast
// steps that would carry a segment's total into the next one are left out.
template <int LPN, typename Op>
__device__ __forceinline__ int32_t note_scan_incl(int32_t x, Op op)
{
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));   // row_shr:1
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));   // row_shr:2
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));   // row_shr:4
    x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));   // row_shr:8
    if (LPN >= 32) x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    if (LPN >= 64) x = op(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return x;
}

// The plain (non-wrapping) accumulator of the pulse train, GOOFER.py:487-493, with the onset test taken off the
// sequential chain.  After sample i the reference has recorded R_i = max(R_{i-1}, floor(phase_i)) onsets (its
// `while phase >= next_k` loop, next_k = R + 1), so the onsets are a function of the partial sums alone.  The
// walk therefore runs nothing but the dependent fp64 adds; on the way a lane keeps the phase in front of its own
// samples of the chunk (one v_cndmask per walk block, in the shadow of the adds).  The onsets of a finished chunk are
// then extracted in parallel while the next chunk is already being walked: every lane replays its additions from the
// phase it kept — the same additions in the same order, so the same partial sums — takes floor, and a max-scan and a
// sum-scan across the note's lanes give each lane R in front of its samples and the slot of its first onset.  Negative
// increments and several onsets at one sample need no special case.
//
// NPW notes share a wave, 64 / NPW lanes each.  A walk instruction is one dependent add whatever the lanes hold, and with
// one note per wave all 64 lanes held the same chain: a quarter of the wave per note runs FOUR chains on the same
// instructions (the chunk of a note is spread over its 16 lanes, 32 samples each, and the scans stop at the 16-lane DPP
// rows).  The chain latency — one dependent v_add_f64 per sample, 16.8 cycles — is what the kernel takes either way; what
// shrinks is the vector-issue and LDS bandwidth it takes from the kernels running beside it (94 M -> 30 M wave
// instructions per 1024-note batch).  The notes of a wave are walked to the longest one's length (padding adds +0.0).
template <int NPW>
__device__ __forceinline__ void onset_walk(double *tiles, int group, const float *__restrict__ f0, double sr, const int64_t *__restrict__ sample_off,
                                           int n_notes, int32_t *__restrict__ onset_idx, int32_t *__restrict__ onset_cnt,
                                           int32_t *__restrict__ overflow)
{
    constexpr int LPN = WAVE / NPW;                           // lanes per note
    constexpr int SPL = OC / LPN;                             // samples of a chunk per lane
    constexpr int BPL = SPL / OB > 0 ? SPL / OB : 1;          // walk blocks per lane (NPW = 4: 2), or lanes per block (NPW = 1: 2)
    static_assert(NPW == 1 || NPW == 2 || NPW == 4, "a note's lanes are whole DPP rows");
    __builtin_amdgcn_s_setprio(3);                            // see k_pulse_onsets_wrap
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPN, ln = lane % LPN;              // which note of the wave, lane inside the note
    const int note_raw = group * NPW + sub;
    if (__builtin_amdgcn_readfirstlane(group * NPW) >= n_notes) return;   // whole wave; no block barrier below
    const bool live = note_raw < n_notes;
    const int note = live ? note_raw : n_notes - 1;
    // tiles: [buffer][note of the wave][OC] increments of this wave
    const int64_t base = sample_off[note];
    const int64_t n = live ? sample_off[note + 1] - base : 0;
    const int64_t obase = base / 2 + 16 * (int64_t)note;
    const int32_t cap = (int32_t)((sample_off[note + 1] / 2 + 16 * (int64_t)(note + 1)) - obase);
    // longest note of the wave: the walk's trip count
    int64_t n_max = n;
#pragma unroll
    for (int o = LPN; o < WAVE; o <<= 1) {
        const int64_t other = __shfl_xor(n_max, o, WAVE);
        n_max = other > n_max ? other : n_max;
    }
    n_max = __builtin_amdgcn_readfirstlane((int)(n_max >> 32)) * (int64_t(1) << 32) + (uint32_t)__builtin_amdgcn_readfirstlane((int)n_max);
    // The increments f0[i] / sr (GOOFER.py:491) are formed here, in the parallel fetch stage, instead of by a pass of their
    // own that writes 8 bytes per sample and reads them back.  With r = RN(1 / sr) the quotient correction
    // q + fma(-q, sr, x) r is the correctly rounded x / sr (Markstein), i.e. the increment the reference divides out.
    const float *__restrict__ a = f0 + base;
    const double rsr = 1.0 / sr;
    int32_t *__restrict__ out = onset_idx + obase;
    double phase = 0.0;                                       // per lane: the chain of this lane's note
    int32_t cnt = 0;                                          // onsets recorded so far == R (the same in all lanes of a note)

    float r[SPL];
    auto fetch = [&](int64_t c0) {
        const int64_t s = c0 + (int64_t)ln * SPL;
        if (__all(c0 + OC <= n)) {                            // wave-uniform branch on purpose, see k_pulse_onsets_wrap
#pragma unroll
            for (int k = 0; k < SPL; ++k) r[k] = a[s + k];
        } else {
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int64_t i = s + k;
                const float v = n > 0 ? a[i < n ? i : n - 1] : 0.f;
                r[k] = i < n ? v : 0.f;                       // 0 / sr = +0.0: the padding leaves the phase alone
            }
        }
    };
    // onsets of a walked chunk: t = the note's increments, p0 = this lane's phase in front of its SPL samples
    auto emit = [&](const double *t, double p0, int32_t c0, int64_t n_left) {
        const int valid = n_left >= OC ? OC : (n_left > 0 ? (int)n_left : 0);
        int32_t m[SPL];
        int32_t run = 0;
        double acc = p0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int i = ln * SPL + k;
            acc += t[i];
            const int32_t f = i < valid ? (int32_t)acc : 0;   // trunc == floor wherever it can raise the running max
            run = max(run, f);
            m[k] = run;
        }
        const int32_t upto = note_scan_incl<LPN>(run, [](int32_t x, int32_t y) { return max(x, y); });
        int32_t before = __shfl_up(upto, 1);
        if (ln == 0) before = 0;
        const int32_t start = max(cnt, before);               // R in front of this lane's first sample
        const int32_t mine = max(run, start) - start;
        const int32_t s = note_scan_incl<LPN>(mine, [](int32_t x, int32_t y) { return x + y; });
        const int32_t tot = __shfl(s, (lane | (LPN - 1)), WAVE);              // the note's last lane
        if (mine > 0) {
            int32_t at = cnt + s - mine, prev = start;
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int32_t c = max(prev, m[k]);
                for (; prev < c; ++prev, ++at)
                    if (at < cap) out[at] = c0 + ln * SPL + k;
            }
        }
        cnt += tot;
    };

    fetch(0);
    int buf = 0;
    double kept = 0.0, kept_prev = 0.0;
    for (int64_t c0 = 0; c0 < n_max; c0 += OC, buf ^= 1) {
        double *t = tiles + ((size_t)buf * NPW + sub) * OC;                  // this lane's note
        double *t_prev = tiles + ((size_t)(buf ^ 1) * NPW + sub) * OC;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const double x = (double)r[k], q = x * rsr;
            t[ln * SPL + k] = fma(fma(-q, sr, x), rsr, q);
        }
        wave_lds_sync();
        if (c0 + OC < n_max) fetch(c0 + OC);                // in flight during the walk below
        if (c0 > 0) emit(t_prev, kept_prev, (int32_t)(c0 - OC), n - (c0 - OC));   // its stores complete during the walk as well
        const int64_t left = n_max - c0;
        // blocks are walked in pairs; a padded block adds +0.0 sixteen times and leaves the phase where it was
        const int blocks = left >= OC ? OC / OB : (int)((left + 2 * OB - 1) / (2 * OB)) * 2;
        auto walk = [&](const double (&x)[OB], int g) {
            double ps[OB];
            if (NPW == 1) {                                   // two lanes per block: lane 2g in front of it, lane 2g + 1 in its middle
                kept = ln == 2 * g ? phase : kept;
            } else {                                          // BPL blocks per lane: lane g / BPL in front of its first one
                kept = (g % BPL == 0 && ln == g / BPL) ? phase : kept;
            }
            ps[0] = phase + x[0];
#pragma unroll
            for (int k = 1; k < OB; ++k) ps[k] = ps[k - 1] + x[k];
            if (NPW == 1) kept = ln == 2 * g + 1 ? ps[OB / 2 - 1] : kept;
            phase = ps[OB - 1];
        };
        auto load = [&](double (&x)[OB], int g) {
            const double *q = t + (g < blocks ? g : blocks - 1) * OB;
#pragma unroll
            for (int k = 0; k < OB; ++k) x[k] = q[k];
        };
        double xa[OB], xb[OB];
        // lgkmcnt(0) here, once per chunk: a scalar load left pending on some path above makes the compiler treat the
        // counter as out of order inside the loop and wait for *all* LDS reads in front of every block
        __builtin_amdgcn_s_waitcnt(0xc07f);
        load(xa, 0);
#pragma unroll 1
        for (int g = 0; g < blocks; g += 2) {
            load(xb, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            walk(xa, g);
            load(xa, g + 2);
            __builtin_amdgcn_sched_barrier(0);
            walk(xb, g + 1);
        }
        kept_prev = kept;
    }
    if (n_max > 0) {
        const int64_t c_last = (n_max - 1) / OC * OC;
        double *t_last = tiles + ((size_t)(buf ^ 1) * NPW + sub) * OC;
        emit(t_last, kept_prev, (int32_t)c_last, n - c_last);
    }
    if (ln == 0 && live) {
        onset_cnt[note] = cnt < cap ? cnt : cap;
        if (cnt > cap) atomicMax(overflow, note + 1);          // reported at the next synchronising call (goofer_check)
    }
}

// a workgroup is 4 / NPW waves = four notes either way: the same 32 KiB of tiles per workgroup (and, with the launcher's
// padded LDS request, one workgroup per CU)
template <int NPW>
__global__ __launch_bounds__(256) void k_pulse_onsets_scan(const float *__restrict__ f0, double sr, const int64_t *__restrict__ sample_off,
                                                           int n_notes, int32_t *__restrict__ onset_idx,
                                                           int32_t *__restrict__ onset_cnt, int32_t *__restrict__ overflow)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wv = threadIdx.x >> 6;
    onset_walk<NPW>(reinterpret_cast<double *>(smem) + (size_t)wv * 2 * NPW * OC, blockIdx.x * (4 / NPW) + wv, f0, sr, sample_off, n_notes,
                    onset_idx, onset_cnt, overflow);
}

// Inclusive fp64 sum scan across the wave on the DPP path (see wave_scan_incl): lanes without a source add +0.0.  Any
// association is as good as another here: see k_pulse_onsets_par.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_add_f64(double x)
{
    x = x + dpp_f64<0x111, 0xf>(x);   // row_shr:1
    x = x + dpp_f64<0x112, 0xf>(x);   // row_shr:2
    x = x + dpp_f64<0x114, 0xf>(x);   // row_shr:4
    x = x + dpp_f64<0x118, 0xf>(x);   // row_shr:8
    x = x + dpp_f64<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
    x = x + dpp_f64<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
    return x;
}

// The onsets WITHOUT the sequential walk, wherever that is provably the same thing.
//
// The reference's phase after sample i is the fp64 running sum p_i = fl(p_{i-1} + x_i), x_i = f0[i] / sr (GOOFER.py:491), and
// the onsets depend on it only through floor(p_i) (R_i = max(R_{i-1}, floor(p_i)), see onset_walk).  A blocked parallel scan
// S_i adds the same x_1 .. x_i in another order.  For ANY order of fp64 additions of i terms the computed sum differs from
// the exact one by at most gamma_{i-1} sum|x_j|, gamma_k = k u / (1 - k u), u = 2^-53 (Higham, Accuracy and Stability of
// Numerical Algorithms, 4.2), so |p_i - S_i| <= 2 gamma_{i-1} A_i with A_i = sum_{j<=i} |x_j|.  With every x_j >= 0
// (checked), A_i is the exact sum itself, A_i <= S_i / (1 - gamma), and
//     tol_i = 2.3e-16 (c0 + 2048) S_i   >=   2 gamma_{i-1} A_i       (2 u = 2.2205e-16; c0 + 2048 > i; i < 2^31)
// with 3.5 % to spare for the second-order terms and the rounding of tol_i itself.  If no integer lies within tol_i of S_i
// then floor(p_i) = floor(S_i).  A note for which that holds at every sample gets its onsets from the scan: the same
// integers the walk would have produced, hence the same onset list.  A note with a sample inside the band (f0 an exact
// divisor of sr from phase 0 on, as in the 441 Hz vector; about one ordinary note in 10^4), a negative, non-finite or huge
// increment is walked sequentially by the first wave of its workgroup right here (onset_walk), as before.  S_i == 0 means
// every term so far was +0: exact, never in the band.
//
// One workgroup of four waves per note, 2048 samples per round: wave w takes samples 512 w .. 512 w + 511 of the round, lane l
// of it samples 8 l .. 8 l + 7.  A sample's S is (carry of the rounds before + totals of the waves before) + (scan of the lane
// totals before + the lane's own running sum): one summation tree over x_1 .. x_i.  Two workgroup barriers per round: the
// wave totals, then the waves' largest floor (the onset count in front of a wave) and the band flags.
#define PAR_TOL 2.3e-16
#define PAR_ROUND (4 * OC)
__global__ __launch_bounds__(256) void k_pulse_onsets_par(const float *__restrict__ f0, double sr, const int64_t *__restrict__ sample_off,
                                                          int n_notes, int32_t *__restrict__ onset_idx,
                                                          int32_t *__restrict__ onset_cnt, int32_t *__restrict__ overflow,
                                                          int32_t *__restrict__ stats, int force)
{
    constexpr int SPL = OC / WAVE;
    extern __shared__ __align__(16) unsigned char smem[];     // the walk's tiles (one wave: 2 x OC doubles)
    __shared__ double s_tot[2][4];
    __shared__ int32_t s_top[2][4], s_bad[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int note = blockIdx.x;
    const int64_t base = sample_off[note];
    const int64_t n = sample_off[note + 1] - base;
    const int64_t obase = base / 2 + 16 * (int64_t)note;
    const int32_t cap = (int32_t)((sample_off[note + 1] / 2 + 16 * (int64_t)(note + 1)) - obase);
    const float *__restrict__ a = f0 + base;
    const double rsr = 1.0 / sr;
    int32_t *__restrict__ out = onset_idx + obase;
    double carry = 0.0;                                       // the scan's phase in front of the round (uniform)
    int32_t cnt = 0;                                          // onsets in front of the round (uniform)
    bool unsure = force != 0;

    float r[SPL];
    auto fetch = [&](int64_t w0) {                            // w0: first sample of this wave's part of a round
        const int64_t s = w0 + (int64_t)lane * SPL;
        if (w0 + OC <= n) {
#pragma unroll
            for (int k = 0; k < SPL; ++k) r[k] = a[s + k];
        } else {
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int64_t i = s + k;
                const float v = n > 0 ? a[i < n ? i : n - 1] : 0.f;
                r[k] = i < n ? v : 0.f;                       // 0 / sr = +0.0: the padding leaves the phase alone
            }
        }
    };
    if (!unsure) fetch((int64_t)wv * OC);
    int par = 0;
    for (int64_t c0 = 0; c0 < n && !unsure; c0 += PAR_ROUND, par ^= 1) {
        const int64_t w0 = c0 + (int64_t)wv * OC;
        double l[SPL];
        bool nb = false;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const double x = (double)r[k], q = x * rsr;
            const double inc = fma(fma(-q, sr, x), rsr, q);   // RN(x / sr), as in onset_walk
            nb |= !(inc >= 0.0);                              // negative or NaN
            l[k] = k ? l[k - 1] + inc : inc;
        }
        if (c0 + PAR_ROUND < n) fetch(w0 + PAR_ROUND);
        const double incl = wave_scan_add_f64(l[SPL - 1]);
        double excl = __shfl_up(incl, 1, WAVE);
        excl = lane == 0 ? 0.0 : excl;
        if (lane == WAVE - 1) s_tot[par][wv] = incl;
        __syncthreads();
        double before_w = carry, total = carry;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double t = s_tot[par][q];
            total = total + t;
            before_w = q < wv ? total : before_w;
        }
        const double p0 = before_w + excl;
        const double tolf = PAR_TOL * (double)(c0 + PAR_ROUND);
        const int64_t left = n - w0;
        const int valid = left >= OC ? OC : (left > 0 ? (int)left : 0);
        int32_t m[SPL];
        int32_t run = 0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int i = lane * SPL + k;
            const double S = p0 + l[k];
            const double d = S - rint(S), tol = S * tolf;
            nb |= (fabs(d) <= tol) && (tol > 0.0);
            nb |= !(S < 1073741824.0);                        // (also inf / NaN)
            const int32_t f = i < valid ? (int32_t)S : 0;     // S >= 0: trunc == floor
            run = max(run, f);
            m[k] = run;
        }
        const int32_t upto = wave_scan_incl(run, [](int32_t x, int32_t y) { return max(x, y); });
        const bool wave_bad = __any(nb);
        if (lane == WAVE - 1) { s_top[par][wv] = upto; s_bad[par][wv] = wave_bad ? 1 : 0; }
        __syncthreads();
        int32_t cnt_w = cnt, cnt_all = cnt;                   // onsets in front of this wave's samples / behind the round
        int bad = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cnt_all = max(cnt_all, s_top[par][q]);
            cnt_w = q < wv ? cnt_all : cnt_w;
            bad |= s_bad[par][q];
        }
        if (bad) { unsure = true; break; }
        // onsets of the wave's samples: as onset_walk's emit
        int32_t before = __shfl_up(upto, 1);
        if (lane == 0) before = 0;
        const int32_t start = max(cnt_w, before);             // R in front of this lane's first sample
        const int32_t mine = max(run, start) - start;
        if (mine > 0) {
            int32_t at = start, prev = start;                 // the slot of onset number k + 1 is k
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const int32_t c = max(prev, m[k]);
                for (; prev < c; ++prev, ++at)
                    if (at < cap) out[at] = (int32_t)w0 + lane * SPL + k;
            }
        }
        cnt = cnt_all;
        carry = total;
    }
    if (threadIdx.x == 0 && stats) {
        atomicAdd(stats + 2, 1);
        if (unsure) atomicAdd(stats + 1, 1);
    }
    if (unsure) {
        if (wv == 0) onset_walk<1>(reinterpret_cast<double *>(smem), note, f0, sr, sample_off, n_notes, onset_idx, onset_cnt, overflow);
        return;
    }
    if (threadIdx.x == 0) {
        onset_cnt[note] = cnt < cap ? cnt : cap;
        if (cnt > cap) atomicMax(overflow, note + 1);          // reported at the next synchronising call (goofer_check)
    }
}

// One wave per note, lanes over onsets: T = 1/max(last_valid_f0, 1e-6) with last_valid_f0 the most
// recent f0 > 1e-6 at or before the onset sample (160 Hz before any), T0 = clip(round_half_even(sr*T),
// 3, 8192) (GOOFER.py:488-499), and end_max = running max of (sample + T0) in onset order (the bound
// k_pulse_place uses to stop its look-back).
__global__ __launch_bounds__(64) void k_onset_finish(const float *__restrict__ f0, const int64_t *__restrict__ sample_off,
                                                     int n_notes, double sr, const int32_t *__restrict__ onset_idx,
                                                     const int32_t *__restrict__ onset_cnt, onset_t *__restrict__ onsets)
{
    const int note = blockIdx.x, lane = threadIdx.x;
    const int64_t base = sample_off[note];
    const int64_t obase = base / 2 + 16 * (int64_t)note;
    const float *__restrict__ f = f0 + base;
    const int cnt = onset_cnt[note];
    int32_t carry = 0;
    for (int k0 = 0; k0 < cnt; k0 += WAVE) {
        const int k = k0 + lane;
        int32_t i = 0, T0 = 0, e = 0;
        double T = 0.0;
        if (k < cnt) {
            i = onset_idx[obase + k];
            int64_t b = i;
            while (b >= 0 && !(f[b] > 1e-6f)) --b;
            const double last = b >= 0 ? (double)f[b] : 160.0;
            T = 1.0 / fmax(last, 1e-6);
            long t0 = (long)rint(sr * T);                    // round-half-even, like Python round()
            T0 = (int32_t)(t0 < 3 ? 3 : (t0 > 8192 ? 8192 : t0));
            e = i + T0;
        }
        // inclusive prefix max across the wave, then across chunks
        int32_t m = e;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            int32_t o = __shfl_up(m, off, WAVE);
            if (lane >= off) m = o > m ? o : m;
        }
        m = m > carry ? m : carry;
        if (k < cnt) {
            onset_t o;
            o.i = i; o.T0 = T0; o.end_max = m; o.pad = 0; o.T = T;
            onsets[obase + k] = o;
        }
        carry = __shfl(m, WAVE - 1, WAVE);
    }
}

// One workgroup = 256 * PP_SPT consecutive samples.  When the tile lies inside one note (almost
// always) the onsets that can touch it — first onset whose running end_max passes the tile start, up to
// the last onset starting inside the tile — are staged once in LDS; every sample then scans that short
// list in ascending onset order (the reference's accumulation order).  Other tiles (note boundaries, or
// more onsets than the LDS list holds) take the per-sample search of the compact list in global memory.
// PP_SPT (common.h): eight — 48 registers, eight waves per SIMD, 0.135 -> 0.113 ms against sixteen (87 registers, five waves)
// once the on-the-fly pulses had left the kernel (with them it held 115 registers either way and sixteen was the faster)
#define PP_MAXON 512

__device__ __forceinline__ float pulse_value(const onset_t &o, int j, const float *__restrict__ peak, const float *__restrict__ tab)
{
    const int d = j - o.i;
    if (d < 0 || d >= o.T0) return 0.f;
    return tab[pulse_tab_row(o.T0) + d];                      // (T0 <= PULSE_TAB_MAX = the cap of the onset kernels)
}

// Which onsets can touch which tile, once per tile instead of once per workgroup through two rounds of note search, a count over
// the note's whole onset list, two atomics and two barriers (k_pulse_place's set-up chain was most of its time): a thread per
// tile finds the tile's note and, when the tile lies inside one note, the first onset whose running end_max passes the tile's
// first sample and the last onset starting at or before its last sample — binary searches on the two monotone columns.
// tiles[t] = {note or -1 (the tile crosses a note boundary), k0, k1, 0}.
__global__ void k_pulse_tiles(const onset_t *__restrict__ onsets, const int32_t *__restrict__ onset_cnt, const int64_t *__restrict__ sample_off,
                              int n_notes, int64_t total_samples, int n_tiles, int4 *__restrict__ tiles)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t g0 = (int64_t)t * (256 * PP_SPT);
    int64_t gl = g0 + 256 * PP_SPT - 1;
    if (gl > total_samples - 1) gl = total_samples - 1;
    const int lo_n = csr_find(sample_off, n_notes, g0), hi_n = csr_find(sample_off, n_notes, gl);
    if (lo_n != hi_n) {
        tiles[t] = make_int4(-1, lo_n, hi_n, 0);
        return;
    }
    const int64_t base = sample_off[lo_n];
    const onset_t *ol = onsets + (base / 2 + 16 * (int64_t)lo_n);
    const int cnt = onset_cnt[lo_n];
    const int32_t j_lo = (int32_t)(g0 - base), j_hi = (int32_t)(gl - base);
    int a = 0, b = cnt;                                       // onsets with end_max <= j_lo: a prefix
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (ol[mid].end_max <= j_lo) a = mid + 1; else b = mid;
    }
    const int k0 = a;
    a = 0; b = cnt;                                           // onsets with i <= j_hi: a prefix
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (ol[mid].i <= j_hi) a = mid + 1; else b = mid;
    }
    tiles[t] = make_int4(lo_n, k0, a - 1, 0);
}

__global__ __launch_bounds__(256) void k_pulse_place(const onset_t *__restrict__ onsets, const int32_t *__restrict__ onset_cnt,
                                                     const float *__restrict__ peak, const float *__restrict__ tab,
                                                     const int64_t *__restrict__ sample_off, int n_notes, int64_t total_samples,
                                                     float *__restrict__ pulse, const int4 *__restrict__ tiles)
{
    __shared__ int s_pair[2];
    __shared__ int s_rng[2];
    __shared__ onset_t s_on[PP_MAXON];
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x * PP_SPT);
    int64_t gl = g0 + (int64_t)blockDim.x * PP_SPT - 1;
    if (gl > total_samples - 1) gl = total_samples - 1;
    int lo_n, hi_n;
    int k0 = 0, k1 = -1;
    if (tiles) {
        const int4 tl = tiles[blockIdx.x];                    // (workgroup-uniform: a scalar load)
        lo_n = tl.x >= 0 ? tl.x : tl.y;
        hi_n = tl.x >= 0 ? tl.x : tl.z;
        k0 = tl.y;
        k1 = tl.z;
    } else {
    if (threadIdx.x == 0) {
        s_rng[0] = 0;
        s_rng[1] = 0;
    }
    block_note_range_last(sample_off, n_notes, g0, gl, s_pair, lo_n, hi_n);
    if (lo_n == hi_n) {
        // onsets that can touch the tile: [first with end_max > j_lo (end_max is monotone), last with i <= j_hi].  Both
        // are counts over the sorted list, taken by the whole workgroup at once instead of two serial binary searches.
        const int64_t base = sample_off[lo_n];
        const onset_t *ol = onsets + (base / 2 + 16 * (int64_t)lo_n);
        const int cnt = onset_cnt[lo_n];
        const int32_t j_lo = (int32_t)(g0 - base), j_hi = (int32_t)(gl - base);
        int c_first = 0, c_last = 0;
        for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
            c_first += ol[k].end_max <= j_lo;
            c_last += ol[k].i <= j_hi;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c_first += __shfl_xor(c_first, o, 64);
            c_last += __shfl_xor(c_last, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_rng[0], c_first);
            atomicAdd(&s_rng[1], c_last);
        }
        __syncthreads();
        k0 = __builtin_amdgcn_readfirstlane(s_rng[0]);
        k1 = __builtin_amdgcn_readfirstlane(s_rng[1]) - 1;
    }
    }
    const int nk = k1 - k0 + 1;
    if (lo_n == hi_n && nk <= PP_MAXON) {
        const int64_t base = sample_off[lo_n];
        const onset_t *ol = onsets + (base / 2 + 16 * (int64_t)lo_n);
        for (int k = threadIdx.x; k < nk; k += blockDim.x) s_on[k] = ol[k0 + k];
        __syncthreads();
        // a thread owns PP_SPT consecutive samples (16-byte stores): they share their covering onsets, so the short
        // search — last onset starting at or before the fourth sample, back to the first whose running end_max passes
        // the first sample — is paid once per PP_SPT outputs, and the sums run over those one to three onsets only
        const int64_t g = g0 + (int64_t)threadIdx.x * PP_SPT;
        if (g > gl) return;
        const int32_t j = (int32_t)(g - base);
        const int live = gl - g + 1 < PP_SPT ? (int)(gl - g + 1) : PP_SPT;
        int lo = -1, hi = nk;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_on[mid].i <= j + live - 1) lo = mid; else hi = mid;
        }
        float acc[PP_SPT];
#pragma unroll
        for (int e = 0; e < PP_SPT; ++e) acc[e] = 0.f;
        if (lo >= 0) {
            int first = lo;
            while (first > 0 && s_on[first - 1].end_max > j) --first;
            for (int k = first; k <= lo; ++k) {                                  // ascending onsets: the reference's order
                const onset_t o = s_on[k];
                {
                    // the eight table values first (index clamped into the pulse), then the range test as a select: the loads of
                    // a thread's samples are in flight together instead of one round trip per sample behind its own branch
                    const float *__restrict__ row = tab + pulse_tab_row(o.T0);
                    const int d0 = j - o.i;
                    float tv[PP_SPT];
#pragma unroll
                    for (int e = 0; e < PP_SPT; ++e) {
                        const int d = d0 + e;
                        tv[e] = row[d < 0 ? 0 : (d >= o.T0 ? o.T0 - 1 : d)];
                    }
#pragma unroll
                    for (int e = 0; e < PP_SPT; ++e) {
                        const int d = d0 + e;
                        acc[e] += (d < 0 || d >= o.T0) ? 0.f : tv[e];
                    }
                }
            }
        }
        if (live == PP_SPT) {
#pragma unroll
            for (int e = 0; e < PP_SPT; e += 4)
                *reinterpret_cast<float4 *>(pulse + g + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
        } else {
            for (int e = 0; e < live; ++e) pulse[g + e] = acc[e];
        }
        return;
    }
    for (int u = 0; u < PP_SPT; ++u) {
        const int64_t g = g0 + threadIdx.x + (int64_t)u * blockDim.x;
        if (g >= total_samples) break;
        int note = lo_n;
        while (sample_off[note + 1] <= g) ++note;
        const int64_t base = sample_off[note];
        const int32_t j = (int32_t)(g - base);
        const onset_t *ol = onsets + (base / 2 + 16 * (int64_t)note);
        const int cnt = onset_cnt[note];
        float acc = 0.f;
        int lo = -1, hi = cnt;   // last onset with i <= j
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (ol[mid].i <= j) lo = mid; else hi = mid;
        }
        if (lo >= 0) {
            int first = lo;
            while (first > 0 && ol[first - 1].end_max > j) --first;
            for (int k = first; k <= lo; ++k) acc += pulse_value(ol[k], j, peak, tab);
        }
        pulse[g] = acc;
    }
}

int launch_phase_inc(goofer_ctx *ctx, const float *f0, float f0_scale, int64_t total_samples, double *inc, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_phase_inc, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, st, f0, total_samples,
                       (double)ctx->plan.sr, f0_scale, inc);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// tiles != nullptr (16-byte aligned, 4 ints per tile of 256 * PP_SPT samples of total_samples): also the pulse placement's tile table
// (k_pulse_tiles) — launch_pulse_place then is a single launch
int launch_pulse_onsets(goofer_ctx *ctx, const float *f0, float f0_scale, const int64_t *sample_off,
                        int n_notes, onset_t *onsets, int32_t *onset_idx, int32_t *onset_cnt, int32_t *overflow, int64_t total_samples,
                        int32_t *tiles, hipStream_t st)
{
    if (n_notes <= 0) return GOOFER_OK;
    if (f0_scale != 1.0f) return goofer_fail(ctx, GOOFER_EINVAL, "pulse onsets expect pre-scaled f0");
    if (ctx->pulse_scan != 0) {
        // parallel phase scan, the sequential walk inside it for the notes it cannot settle; `overflow` is the handle's block of
        // sticky words: [0] overflow, [1] notes walked, [2] notes scanned
        hipLaunchKernelGGL(k_pulse_onsets_par, dim3(n_notes), dim3(256), 2 * OC * sizeof(double), st, f0, (double)ctx->plan.sr, sample_off,
                           n_notes, onset_idx, onset_cnt, overflow, overflow, ctx->pulse_scan == 2 ? 1 : 0);
        LAUNCH_CHECK(ctx);
    } else {
        // the sequential walk for every note (A/B reference of the scan): one note per wave, four notes per workgroup
        const int blocks = (n_notes + 3) / 4;
        const int per_cu = (blocks + 255) / 256;              // MI355X: 256 CUs, 160 KiB LDS each
        size_t lds = (size_t)(160 * 1024) / per_cu;
        if (lds > (size_t)81 * 1024) lds = (size_t)81 * 1024;   // 81 KiB: two of these cannot share a CU, and 79 KiB stay free for
                                                                // the kernel running beside the walk
        lds = lds / 1024 * 1024;
        const size_t need = 4 * 2 * OC * sizeof(double);      // 32 KiB actually used
        if (lds < need) lds = need;
        if (int arc = kernel_allow_max_lds(ctx, (const void *)k_pulse_onsets_scan<1>)) return arc;
        hipLaunchKernelGGL(k_pulse_onsets_scan<1>, dim3(blocks), dim3(256), lds, st, f0, (double)ctx->plan.sr, sample_off, n_notes, onset_idx, onset_cnt, overflow);
        LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_onset_finish, dim3(n_notes), dim3(64), 0, st, f0, sample_off, n_notes, (double)ctx->plan.sr, onset_idx,
                       onset_cnt, onsets);
    LAUNCH_CHECK(ctx);
    if (total_samples > 0 && tiles && ((uintptr_t)tiles & 15) == 0) {
        const unsigned n_tiles = (unsigned)((total_samples + 256 * PP_SPT - 1) / (256 * PP_SPT));
        hipLaunchKernelGGL(k_pulse_tiles, dim3((n_tiles + 63) / 64), dim3(64), 0, st, onsets, onset_cnt, sample_off, n_notes, total_samples,
                           (int)n_tiles, reinterpret_cast<int4 *>(tiles));
        LAUNCH_CHECK(ctx);
    }
    return GOOFER_OK;
}

// tiles: the table launch_pulse_onsets made (4 ints per tile of 256 * PP_SPT samples), or nullptr: every
// workgroup searches for itself (goofer_pulse_train with an unaligned scratch pointer)
int launch_pulse_place(goofer_ctx *ctx, const onset_t *onsets, const int32_t *onset_cnt, const int64_t *sample_off, int n_notes,
                       int64_t total_samples, float *pulse, const int32_t *tiles, hipStream_t st)
{
    if (total_samples <= 0) return GOOFER_OK;
    const unsigned n_tiles = (unsigned)((total_samples + 256 * PP_SPT - 1) / (256 * PP_SPT));
    const int4 *tl = (tiles && ((uintptr_t)tiles & 15) == 0) ? reinterpret_cast<const int4 *>(tiles) : nullptr;
    hipLaunchKernelGGL(k_pulse_place, dim3(n_tiles), dim3(256), 0, st, onsets, onset_cnt, ctx->plan.pulse_peak, ctx->plan.pulse_shape,
                       sample_off, n_notes, total_samples, pulse, tl);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_pulse_train(goofer_ctx *ctx, const float *f0, float f0_scale, const int64_t *sample_off, int n_notes,
                       int64_t total_samples, float *pulse, double *inc, onset_t *onsets, int32_t *onset_idx, int32_t *onset_cnt,
                       int32_t *overflow, hipStream_t st)
{
    if (total_samples <= 0 || n_notes <= 0) return GOOFER_OK;
    int rc;
    // the walk divides f0 by sr itself: the increments array (8 bytes per sample) is free and holds the placement's tile table
    int32_t *tiles = reinterpret_cast<int32_t *>(inc);
    if ((rc = launch_pulse_onsets(ctx, f0, f0_scale, sample_off, n_notes, onsets, onset_idx, onset_cnt, overflow, total_samples, tiles, st))) return rc;
    return launch_pulse_place(ctx, onsets, onset_cnt, sample_off, n_notes, total_samples, pulse, tiles, st);
}

// ---------------------------------------------------------------------------------------------
// Sub-harmonic pulse layer ('sg' flag) — gf.add_subharms + apply_subharm_vibrato (GOOFER.py:672-766).
//
//   k_subharm_inc     vibrato'd f0 (in the precision of the reference's array: float32 — or float64 behind gf.synthesize's time
//                     stretch, goofer_batch.f0_64) and the tracker increments sub_f0/sr
//   k_pulse_onsets_wrap  the wrapped phase tracker, exact sequential order
//   k_subharm_finish  per event: T = 1/sub_f0, n = max(3, round_half_even(sr T)), fp32 peak of its LF pulse, end_max
//   k_subharm_place   gather of the covering LF pulses (ascending events), * voicing mask, per-note max
//   k_subharm_add     pulse += sub / max * weight
// LF pulse here is lf_model_pulse (GOOFER.py:437-471) with Ra .02, Rg 1.7, Rk 1 on an fp32 time grid.
struct sub_cfg {
    double ratio;        // 2^(semitones/12)
    double vib_rate, vib_depth;
    int vib_on, vib_fade;   // fade = int(delay * sr)
    double sr;
};

__device__ __forceinline__ float sub_lf_raw(int k, int n, double T)
{
    const double step = T / (double)n;
    const float tk = (float)((double)k * step);               // np.linspace(..., endpoint=False, dtype=float32)
    const double Tp = 0.02 * T;
    const double Tc = Tp + 1.0 * (T - Tp);
    if ((double)tk < Tp) {
        const float a = 3.141592653589793f * tk;               // np.pi * t[mask1] stays fp32 (weak python scalar)
        const double s = sin((double)a / (2.0 * Tp));
        return (float)(s * s);
    }
    if ((double)tk < Tc) {
        const double tau = ((double)tk - Tp) / (Tc - Tp);
        return (float)(exp(-1.7 * tau) * cos(3.141592653589793 * tau / 2.0));
    }
    return 0.f;
}

__global__ __launch_bounds__(256) void k_subharm_inc(const float *__restrict__ f0, const double *__restrict__ f0_64,
                                                     const float *__restrict__ mask,
                                                     const int64_t *__restrict__ sample_off, int n_notes, int64_t total,
                                                     const goofer_note_params *__restrict__ params, sub_cfg c,
                                                     double *__restrict__ fm, double *__restrict__ inc)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    if (!(params[note].subharm_weight > 0.f)) { inc[g] = 0.0; fm[g] = 0.0; return; }
    const int64_t base = sample_off[note], n = sample_off[note + 1] - base, i = g - base;
    // the f0 array the layer tracks is float32 in the reference (modulated_f0 = f0_interp.copy() keeps the type, :763-764) —
    // and float64 when the time stretch made it one (:1053): f0_64
    double f = f0_64 ? f0_64[g] : (double)f0[g];
    if (c.vib_on && f > 0.0) {
        double v = sin(((2.0 * 3.141592653589793) * c.vib_rate) * ((double)i / c.sr) + 0.0);
        if (c.vib_fade < n && i < c.vib_fade) {
            const double fade = c.vib_fade > 1 ? (i >= c.vib_fade - 1 ? 1.0 : (double)i * (1.0 / (double)(c.vib_fade - 1))) : 0.0;
            v *= fade;
        }
        f = f * (1.0 + v * c.vib_depth);
        if (!f0_64) f = (double)(float)f;
    }
    fm[g] = f;
    double a = 0.0;
    if (mask[g] > 0.f && f > 0.0) {
        const double sub = f * c.ratio;
        if (!(sub < 1e-2)) a = sub / c.sr;
    }
    inc[g] = a;
}

// The reference keeps one LF pulse per '{sub_f0:.2f}' key (GOOFER.py:716-724): an event reuses the pulse of the FIRST
// event whose sub_f0 prints to the same two decimals.  keys = the note's slice of the increment buffer (free once
// the tracker has run), holding each event's own sub_f0.
__global__ __launch_bounds__(64) void k_subharm_finish(const double *__restrict__ fm, const int64_t *__restrict__ sample_off, int n_notes,
                                                       sub_cfg c, const int32_t *__restrict__ onset_idx,
                                                       const int32_t *__restrict__ onset_cnt, double *__restrict__ keys_all,
                                                       onset_t *__restrict__ onsets)
{
    const int note = blockIdx.x, lane = threadIdx.x;
    const int64_t base = sample_off[note];
    const int64_t obase = base + 16 * (int64_t)note;         // (the sub-harmonic layer's slots: one per sample, k_pulse_onsets_wrap)
    const int cnt = onset_cnt[note];
    double *keys = keys_all + base;                              // (cnt <= n: at most one event per sample)
    const int64_t n_note = sample_off[note + 1] - base;
    const int kcap = (int)(cnt < n_note ? cnt : n_note);
    for (int k = lane; k < kcap; k += WAVE) keys[k] = fm[base + onset_idx[obase + k]] * c.ratio;
    __threadfence_block();
    wave_lds_sync();
    int32_t carry = 0;
    for (int k0 = 0; k0 < cnt; k0 += WAVE) {
        const int k = k0 + lane;
        int32_t i = 0, n = 0, e = 0;
        float peak = 0.f;
        double T = 0.0;
        if (k < cnt) {
            i = onset_idx[obase + k];
            double sub = fm[base + i] * c.ratio;
            if (k < kcap) {
                const double key = rint(sub * 100.0);
                for (int j = 0; j < k; ++j) {
                    const double sj = keys[j];
                    if (rint(sj * 100.0) == key) { sub = sj; break; }
                }
            }
            T = 1.0 / sub;
            long t0 = (long)rint(c.sr * T);
            n = (int32_t)(t0 <= 3 ? 3 : t0);
            for (int q = 0; q < n; ++q) peak = fmaxf(peak, fabsf(sub_lf_raw(q, n, T)));
            e = i + n;
        }
        int32_t m = e;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            int32_t o = __shfl_up(m, off, WAVE);
            if (lane >= off) m = o > m ? o : m;
        }
        m = m > carry ? m : carry;
        if (k < cnt) {
            onset_t o;
            o.i = i; o.T0 = n; o.end_max = m; o.pad = __float_as_int(peak); o.T = T;
            onsets[obase + k] = o;
        }
        carry = __shfl(m, WAVE - 1, WAVE);
    }
}

__global__ __launch_bounds__(256) void k_subharm_place(const onset_t *__restrict__ onsets, const int32_t *__restrict__ onset_cnt,
                                                       const float *__restrict__ mask, const int64_t *__restrict__ sample_off,
                                                       int n_notes, int64_t total, const goofer_note_params *__restrict__ params,
                                                       double *__restrict__ sub, unsigned long long *__restrict__ max_bits,
                                                       int accumulate, int last)
{
    // several ratios (a list of subharm_semitones): each has its own event list; the pulses of all of them are summed
    // (accumulate), and the voicing mask and the joint maximum are applied once, after the last one
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo_n, hi_n;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo_n, hi_n);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total) return;
    int note = lo_n;
    while (sample_off[note + 1] <= g) ++note;
    if (!(params[note].subharm_weight > 0.f)) return;
    const int64_t base = sample_off[note];
    const int32_t j = (int32_t)(g - base);
    const onset_t *ol = onsets + (base + 16 * (int64_t)note);
    const int cnt = onset_cnt[note];
    double acc = 0.0;
    int lo = -1, hi = cnt;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ol[mid].i <= j) lo = mid; else hi = mid;
    }
    if (lo >= 0) {
        int first = lo;
        while (first > 0 && ol[first - 1].end_max > j) --first;
        for (int k = first; k <= lo; ++k) {
            const onset_t o = ol[k];
            const int d = j - o.i;
            if (d < o.T0) {
                const float pk = __int_as_float(o.pad);
                float v = sub_lf_raw(d, o.T0, o.T);
                if (pk > 0.f) v = v / pk;
                acc += (double)v;
            }
        }
    }
    if (accumulate) acc += sub[g];
    if (last) acc *= (double)mask[g];
    sub[g] = acc;
    if (last) {
        const double m = fabs(acc);
        if (m > 0.0) atomicMax(max_bits + note, (unsigned long long)__double_as_longlong(m));
    }
}

__global__ __launch_bounds__(256) void k_subharm_add(float *__restrict__ pulse, const double *__restrict__ sub,
                                                     const unsigned long long *__restrict__ max_bits,
                                                     const int64_t *__restrict__ sample_off, int n_notes, int64_t total,
                                                     const goofer_note_params *__restrict__ params)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    int lo, hi;
    block_note_range(sample_off, n_notes, g0, total, s_pair, lo, hi);
    const int64_t g = g0 + threadIdx.x;
    if (g >= total) return;
    int note = lo;
    while (sample_off[note + 1] <= g) ++note;
    const float w = params[note].subharm_weight;
    if (!(w > 0.f)) return;
    double v = sub[g];
    const double mx = __longlong_as_double((long long)max_bits[note]);
    if (mx > 1e-6) v /= mx;
    v *= (double)w;
    pulse[g] = (float)((double)pulse[g] + v);
}

int launch_subharm(goofer_ctx *ctx, const float *f0s, const double *f0_64, const float *mask, const int64_t *sample_off, int n_notes,
                   int64_t total, const goofer_note_params *params, const double *ratios, int n_ratios, int vib_on, double vib_rate,
                   double vib_depth, double vib_delay, double *fm, double *inc, onset_t *onsets, int32_t *onset_idx, int32_t *onset_cnt,
                   int32_t *overflow, const unsigned char *note_on, double *sub, unsigned long long *max_bits, float *pulse,
                   hipStream_t st)
{
    if (total <= 0 || n_notes <= 0 || n_ratios <= 0) return GOOFER_OK;
    const unsigned nb = (unsigned)((total + 255) / 256);
    for (int ri = 0; ri < n_ratios; ++ri) {
        sub_cfg c;
        c.ratio = ratios[ri]; c.vib_rate = vib_rate; c.vib_depth = vib_depth; c.vib_on = vib_on;
        c.sr = (double)ctx->plan.sr;
        c.vib_fade = (int)(vib_delay * c.sr);
        hipLaunchKernelGGL(k_subharm_inc, dim3(nb), dim3(256), 0, st, f0s, f0_64, mask, sample_off, n_notes, total, params, c, fm, inc);
        LAUNCH_CHECK(ctx);
        {
            const int blocks = (n_notes + 3) / 4;
            const int per_cu = (blocks + 255) / 256;
            size_t lds = (size_t)(160 * 1024) / per_cu;
            lds = lds / 1024 * 1024;
            const size_t need = 4 * 2 * OC * sizeof(double) + 4 * WAVE * sizeof(int32_t);
            if (lds < need) lds = need;
            if (int arc = kernel_allow_max_lds(ctx, (const void *)k_pulse_onsets_wrap)) return arc;
            hipLaunchKernelGGL(k_pulse_onsets_wrap, dim3(blocks), dim3(256), lds, st, inc, sample_off, n_notes, onset_idx, onset_cnt,
                               overflow, note_on);
            LAUNCH_CHECK(ctx);
        }
        hipLaunchKernelGGL(k_subharm_finish, dim3(n_notes), dim3(64), 0, st, fm, sample_off, n_notes, c, onset_idx, onset_cnt, inc, onsets);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(k_subharm_place, dim3(nb), dim3(256), 0, st, onsets, onset_cnt, mask, sample_off, n_notes, total, params, sub,
                           max_bits, ri > 0 ? 1 : 0, ri == n_ratios - 1 ? 1 : 0);
        LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_subharm_add, dim3(nb), dim3(256), 0, st, pulse, sub, max_bits, sample_off, n_notes, total, params);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
