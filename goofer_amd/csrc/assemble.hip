// Note assembly on the device — executes the plans of goofer_amd/sampler.py (SillySampler.resample).
//
//   k_env_edit         knot decode + br tilt + es smooth/sharpen + fw width warp on the source rows a note
//                      uses                                 GOOFER.py:149-168, SillySampler.py:502-574
//   k_row_recs + k_env_rows  4-tap frame gather (slicing, L0 cross-fades / L1 mirror mean / L2 stretch, velocity
//                      prefix stretch) + formant-strength gain bells
//                                                           SillySampler.py:625-696, 765-773, 791-833
//   k_sample_assemble  per-sample voicing mask (slice, tile, reverse, force-voiced, velocity stretch) and
//                      pitch curve -> f0                    SillySampler.py:698-712, 787-788, 835-855
// One wave per row for the matrix kernels (row staged in LDS), one thread per sample for the last.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "binops_core.h"

constexpr int A_ROWS = 4;   // rows (waves) per workgroup
constexpr int ES_HALO = 64; // k_env_edit: halo floats either side of the staged row (the 'es' blur has radius <= 28)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// e^x for the log-domain knot lerp (|x| < 80): x log2(e) split into the rounded product and its exact remainder (one FMA for
// the product's rounding error, one for the constant's), the hardware exp2 of the first, a first-order correction for the
// second — six full-rate instructions, within 2 ulp of fp32 of libm's expf (~20 instructions).
__device__ __forceinline__ float exp_split(float x)
{
    const float L2E = 0x1.715476p+0f, L2E_LO = 0x1.4ae0c0p-26f, LN2 = 0x1.62e430p-1f;
    const float hi = x * L2E;
    float lo = __builtin_fmaf(x, L2E, -hi);
    lo = __builtin_fmaf(x, L2E_LO, lo);
    const float r = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(r, lo * LN2, r);
}

// ---------------------------------------------------------------------------------------------
// V64: round 4's arithmetic (fp64 'es' blur, mean match and 'fw' interpolation, libm expf) — option "value_f64", A/B and the
// error-budget tests.  Default: the same steps in fp32 (no index or comparison depends on these values).
// CH: 64-bin chunks of a row when known at compile time (9: n_fft 1024, 17: n_fft 2048; 0 = any width).  With CH the table loads of
// a row's bins (lerp plan, tilt; the fw plan) are issued together before the first use — the rolled loops waited out one round
// trip per chunk and table, ~20 per row.
template <bool V64, int CH>
__global__ __launch_bounds__(256) void k_env_edit(const goofer_assembly a, int64_t total_edit_rows, const int *__restrict__ row_note)
{
    using acc_t = std::conditional_t<V64, double, float>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = a.n_bins;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // uniform row: the note's plan comes in through scalar loads
    const int64_t er = (int64_t)blockIdx.x * A_ROWS + wave;
    if (er >= total_edit_rows) return;                       // no block barrier below
    const int stride = (3 * B + 2 * ES_HALO + a.max_K + 3) & ~3;   // floats per wave, 16-byte multiple
    acc_t *tmp = reinterpret_cast<acc_t *>(reinterpret_cast<float *>(smem) + (size_t)wave * stride);   // [B] scratch (fp64 slots either way)
    float *row = reinterpret_cast<float *>(reinterpret_cast<double *>(tmp) + B) + ES_HALO;   // [B] current row, ES_HALO floats of reflected halo either side
    float *kv = row + B + ES_HALO;                            // [max_K] decoded knot values
    const int note = row_note[er];
    const goofer_note_plan &p = a.notes[note];
    const int r = (int)(er - p.edit_off);                     // index inside the note's edited window
    const int logical = p.row_lo + r;
    const int phys = p.reverse ? p.n_src_rows - 1 - logical : logical;

    // 1. knot decode (2-tap lerp in the log domain + exp)     GOOFER.py:164-165
    const __half *kn = reinterpret_cast<const __half *>(a.knots) + p.knot_off + (int64_t)phys * p.K;
    for (int k = lane; k < p.K; k += WAVE) kv[k] = __half2float(kn[k]);
    wave_lds_sync();
    const bool dense = p.lerp_plan < 0;                       // 'full' mode source: the rows are the fp16 envelope itself
    const int *li = dense ? nullptr : a.lerp_idx + (int64_t)p.lerp_plan * B;
    const float *l0 = dense ? nullptr : a.lerp_w0 + (int64_t)p.lerp_plan * B, *l1 = dense ? nullptr : a.lerp_w1 + (int64_t)p.lerp_plan * B;
    const float *tilt = p.tilt >= 0 ? a.tilts + (int64_t)p.tilt * B : nullptr;
    auto idx = [&](int c) {
        const int b = c * WAVE + lane;
        return (CH == 0 || c < CH - 1 || b < B) ? b : B - 1;
    };
    if constexpr (CH > 0) {
        float v[CH], tv[CH];
        if (tilt) {
#pragma unroll
            for (int c = 0; c < CH; ++c) tv[c] = tilt[idx(c)];
        }
        if (dense) {
#pragma unroll
            for (int c = 0; c < CH; ++c) v[c] = kv[idx(c)];
        } else {
            int iv[CH];
            float a0[CH], a1[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                iv[c] = li[idx(c)];
                a0[c] = l0[idx(c)];
                a1[c] = l1[idx(c)];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float x = a0[c] * kv[iv[c]] + a1[c] * kv[iv[c] + 1];
                v[c] = V64 ? expf(x) : exp_split(x);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int b = c * WAVE + lane;
            if (tilt) v[c] *= tv[c];                           // 2. br: env *= tilt (fp32)   :513-515
            if (c < CH - 1 || b < B) row[b] = v[c];
        }
    } else {
    for (int b = lane; b < B; b += WAVE) {
        float v;
        if (dense) {
            v = kv[b];
        } else {
            int i = li[b];
            const float x = l0[b] * kv[i] + l1[b] * kv[i + 1];
            v = V64 ? expf(x) : exp_split(x);
        }
        if (tilt) v *= tilt[b];                               // 2. br: env *= tilt (fp32)   :513-515
        row[b] = v;
    }
    }
    wave_lds_sync();

    // 3. es: smooth (blur, rematch frame mean, clamp) or sharpen (unsharp, clamp, rematch)   :517-551
    if (p.es_mode) {
        const double *taps = a.es_taps + p.es_taps_off;
        const int rad = p.es_radius;
        acc_t s_src = 0, s_mod = 0;
        const acc_t amount = (acc_t)p.es_amount;
        const bool halo = rad <= ES_HALO && rad < B;          // numpy 'reflect' halo parked beside the row: no index map per tap
        if (halo) {
            for (int h = lane; h < 2 * rad; h += WAVE) {
                const int i = h < rad ? -1 - h : B + (h - rad);
                row[i] = row[(int)reflect_index(i, B)];
            }
            wave_lds_sync();
        }
        // A lane blurs NB CONSECUTIVE bins.  The NB + 2 rad values they need are a window that slides by one per tap: it lives
        // in NB registers as a ring (slot = window index mod NB: with the tap loop unrolled NB times every slot index is a
        // constant), so a tap costs one LDS read, one conversion and one scalar load for the whole lane instead of one of each
        // per bin (the lane-strided layout re-read and re-converted every value for every tap: three instructions per
        // product).  Every bin still adds its products in ascending tap order (fp64: the same sums, bit for bit; fp32: FMAs).
        auto blur_rows = [&](auto nb_tag) {
            constexpr int NB = decltype(nb_tag)::value;
            const int b0 = NB * lane;
            const bool live = b0 < B;
            const float *x0 = row + (live ? b0 : 0) - rad;      // window element e = x0[e], e < NB + 2 rad (inside the halo)
            acc_t acc[NB], xw[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                acc[i] = 0;
                xw[i] = (acc_t)x0[i];
            }
            // the taps (at most 2 * 28 + 1) one per lane, read back with v_readlane: a scalar load per tap would share its
            // counter with the window's LDS reads and put a memory round trip into every step of the loop
            const double tapv = lane <= 2 * rad ? taps[lane] : 0.0;
            const int tap_lo = (int)(uint32_t)__double_as_longlong(tapv), tap_hi = (int)(__double_as_longlong(tapv) >> 32);
            const int tap_f = __float_as_int((float)tapv);
            for (int jb = 0; jb <= 2 * rad; jb += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int j = jb + u;
                    if (j <= 2 * rad) {                         // (wave-uniform)
                        if constexpr (V64) {
                            const double tj = __longlong_as_double(((long long)__builtin_amdgcn_readlane(tap_hi, j) << 32) |
                                                                   (uint32_t)__builtin_amdgcn_readlane(tap_lo, j));
#pragma unroll
                            for (int i = 0; i < NB; ++i) acc[i] += tj * xw[(i + u) % NB];
                        } else {
                            const float tj = __int_as_float(__builtin_amdgcn_readlane(tap_f, j));
#pragma unroll
                            for (int i = 0; i < NB; ++i) acc[i] = __builtin_fmaf(tj, xw[(i + u) % NB], acc[i]);
                        }
                        xw[u] = (acc_t)x0[j + NB];               // element j is done with; its slot takes element j + NB
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int b = b0 + i;
                if (live && b < B) {
                    const acc_t src = (acc_t)row[b];
                    tmp[b] = p.es_mode == 1 ? acc[i] : (V64 ? (acc_t)fmax(0.0, (double)(src + amount * (src - acc[i]))) : (acc_t)fmaxf(0.0f, (float)__builtin_fmaf((float)amount, (float)(src - acc[i]), (float)src)));
                }
            }
            wave_lds_sync();
            // the two row sums in the order the strided layout took them (bin lane + 64 i per lane, then across the wave)
            for (int b = lane; b < B; b += WAVE) {
                s_src += (acc_t)row[b];
                s_mod += tmp[b];
            }
        };
        auto finish_bin = [&](int b, acc_t acc) {
            acc_t src = (acc_t)row[b];
            acc_t mod = p.es_mode == 1 ? acc : (V64 ? (acc_t)fmax(0.0, (double)(src + amount * (src - acc))) : (acc_t)fmaxf(0.0f, (float)(src + amount * (src - acc))));
            tmp[b] = mod;
            s_src += src;
            s_mod += mod;
        };
        const int chunks = (B + WAVE - 1) / WAVE;
        // (a live lane starts below bin B, so its window and the one prefetch past it end below B + NB + rad: inside the halo)
        if (halo && chunks == 17 && 17 + rad <= ES_HALO) blur_rows(std::integral_constant<int, 17>{});
        else if (halo && chunks == 9 && 9 + rad <= ES_HALO) blur_rows(std::integral_constant<int, 9>{});
        else if (halo && chunks == 5 && 5 + rad <= ES_HALO) blur_rows(std::integral_constant<int, 5>{});
        else
            for (int b = lane; b < B; b += WAVE) {
                acc_t acc = 0;
                if (halo) {
                    const float *x = row + (b - rad);
                    for (int j = 0; j <= 2 * rad; ++j) acc += (acc_t)taps[j] * (acc_t)x[j];
                } else {
                    for (int j = 0; j <= 2 * rad; ++j) acc += (acc_t)taps[j] * (acc_t)row[reflect_index(b + j - rad, B)];
                }
                finish_bin(b, acc);
            }
        s_src = wave_sum(s_src);
        s_mod = wave_sum(s_mod);
        acc_t scale;
        if constexpr (V64) {
            const float m0 = (float)(s_src / (double)B);          // np.mean of the fp32 block row -> fp32
            scale = (double)m0 / (s_mod / (double)B + 1e-12);
        } else {
            const float rB = 1.0f / (float)B;
            scale = (s_src * rB) / (s_mod * rB + 1e-12f);          // (one division per row)
        }
        wave_lds_sync();
        for (int b = lane; b < B; b += WAVE) {
            float v = (float)(tmp[b] * scale);
            row[b] = p.es_mode == 1 ? fmaxf(0.0f, v) : v;
        }
        wave_lds_sync();
    }

    // 4. fw: affine stretch of the bin axis about its centre, linear interpolation   :553-574
    float *__restrict__ out = a.edit_rows + er * (int64_t)a.ld;
    if (p.fw_plan >= 0) {
        const int *__restrict__ lo = a.fw_lo + (int64_t)p.fw_plan * B, *__restrict__ hi = a.fw_hi + (int64_t)p.fw_plan * B;
        const double *__restrict__ fr = a.fw_frac + (int64_t)p.fw_plan * B;
        if constexpr (CH > 0) {
            int lv[CH], hv[CH];
            double fv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                lv[c] = lo[idx(c)];
                hv[c] = hi[idx(c)];
                fv[c] = fr[idx(c)];
            }
            float o[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (V64) {
                    o[c] = (float)((1.0 - fv[c]) * (double)row[lv[c]] + fv[c] * (double)row[hv[c]]);
                } else {
                    const float r0 = row[lv[c]];
                    o[c] = __builtin_fmaf((float)fv[c], row[hv[c]] - r0, r0);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int b = c * WAVE + lane;
                if (c < CH - 1 || b < B) out[b] = o[c];
            }
        } else {
        for (int b = lane; b < B; b += WAVE) {
            if constexpr (V64) {
                out[b] = (float)((1.0 - fr[b]) * (double)row[lo[b]] + fr[b] * (double)row[hi[b]]);
            } else {
                const float r0 = row[lo[b]];
                out[b] = __builtin_fmaf((float)fr[b], row[hi[b]] - r0, r0);
            }
        }
        }
    } else if constexpr (CH > 0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int b = c * WAVE + lane;
            if (c < CH - 1 || b < B) out[b] = row[b];
        }
    } else {
        for (int b = lane; b < B; b += WAVE) out[b] = row[b];
    }
}

// ---------------------------------------------------------------------------------------------
// WARP: also write the row as gf.synthesize's harmonic branch wants it — formant-anchored + uniform warp (GOOFER.py:1004-1017,
// warp_row of binops_core.h) with the synthesis batch's per-note shifts and per-row formants — while the row is at hand
// (goofer_render_batch: one read of the edited rows instead of a second pass over the assembled envelope).  Rows of a note
// that does not warp (every f_shift == 1 and formant_shift == 1: note_warps()) are NOT written a second time — the
// harmonic walker reads the assembled row itself for them (frame_block::mark_plain, stems_core.h).
// CH: 64-bin chunks of a row when known at compile time (the loop over them is unrolled: constant offsets, no per-chunk
// address arithmetic), 0 = any width.
// V64: the round-4 arithmetic — fp64 tap blend and fp64 warp interpolation (option "value_f64", A/B and error-budget tests).
// The default blends and interpolates in fp32: no index, threshold or comparison depends on those values, and the
// results stay within 2 ulp of fp32 of the fp64 ones.
//
// Everything that depends on the row only — the four bells' centres, widths and reach, the warp's anchors — is computed
// once per row ACROSS lanes (lane k owns formant k) and handed to the per-bin code through v_readlane, instead of once per
// lane: the per-row set-up used to be half of this kernel's vector instructions and most of its scalar ones.
struct env_loop_grid {
    double fstep;            // np.linspace(0, sr/2, B) spacing of the bell frequencies
    float inv_fstep, nyq_f;
    warp_grid warp;
    int nt;                  // rows leave as non-temporal stores
    const float *freqs_f;    // [B] np.linspace(0, sr/2, B) as fp32 (the plan's table), or null: computed per bin
};

// ---------------------------------------------------------------------------------------------
// k_row_recs + k_env_rows — the frame gather with the per-row set-up taken off the rows' critical path.
//
// Round 4's single kernel (k_env_loop, removed in round 6) gave every output row a wave, and that wave walked a chain of
// dependent loads before it touched a bin:
// row -> note -> plan -> taps -> source rows -> (formants, parameters) -> stores, with the bells' reach and the warp's anchor
// table (fp64) computed across its lanes in between.  Measured (round 5, AMD_SERIALIZE_KERNEL trace): 0.43 ms alone for
// 194 560 rows whether the per-bin arithmetic is fp64 or fp32 — three quarters of its wave-cycles are waits, the rest ~1 200
// instructions per row of which the bins need a third.  (A wave walking a run of rows with the records built lane-parallel in
// LDS was measured too: 0.50-0.62 ms — three or four such waves per SIMD hide less latency than eight short ones.)
//
// Here the set-up is a kernel of its own with one THREAD per row (k_row_recs): absolute source rows and fp32 weights of the
// 4 taps, the four bells (centre, gain - 1, reach as a chunk mask), and the warp — anchors compacted, sortedness, per-segment
// (D, s - 1, c) records and integer thresholds: exactly the fp64 set-up of warp_row's sorted path, in the same order of
// operations.  It writes a 224-byte record per row (43 MB per 1024-note batch against 1.2 GB of rows).  k_env_rows then is
// what is left: the record arrives through a few scalar loads (straight into SGPRs), the source row(s), bells, store, and
// for warping notes the fp32 lerp(s) straight into the warped copy.  Rows with crossing anchors (np.interp's guess chain,
// rare) take warp_row's cross-lane path unchanged.
struct alignas(16) env_row_rec {
    uint32_t src[4];         // rows of the edited-row scratch
    float w[4];              // tap weights (fp32)
    float F[4], g[4];        // bells: centre (Hz), gain - 1
    uint32_t cm[4];          // ... and the 64-bin chunks each can move (0: bell off)
    int32_t thr[5];          // sorted-anchor warp: first bin of segments 1..5 (n_bins: absent)
    int32_t flags;           // ER_* bits
    float irm1;              // uniform warp: displacement per bin, 1 / ratio - 1
    int32_t note;
    int32_t pad[2];
    warp_seg_f32 seg[6];
};
static_assert(sizeof(env_row_rec) == 224, "record layout");
enum { ER_COPY = 1, ER_NZ0 = 2, ER_WARP = 32, ER_STAGE1 = 64, ER_SORTED = 128, ER_STAGE2 = 256 };
size_t env_row_rec_bytes() { return sizeof(env_row_rec); }

template <bool WARP>
__global__ __launch_bounds__(256) void k_row_recs(const goofer_assembly a, int64_t total_out_rows, const int *__restrict__ row_note,
                                                  const double *__restrict__ w_formants, const goofer_note_params *__restrict__ w_params,
                                                  env_row_rec *__restrict__ recs, const env_loop_grid eg)
{
    const int64_t orow = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (orow >= total_out_rows) return;
    const int B = a.n_bins;
    const double nyq = eg.warp.nyq, step = eg.warp.step, inv_step = eg.warp.inv_step;
    const int note = row_note[orow];
    const goofer_note_plan *pp = a.notes + note;
    const int64_t t = orow - pp->env_off;
    const int32_t *ti = a.tap_idx + (pp->tap_off + t) * 4;
    const double *tw = a.tap_w + (pp->tap_off + t) * 4;
    const int64_t e0 = pp->edit_off - pp->row_lo;
    env_row_rec rc;
    int flags = 0;
    double w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = tw[k];
        rc.src[k] = (uint32_t)(e0 + ti[k]);
        rc.w[k] = (float)w[k];
        if (w[k] != 0.0) flags |= ER_NZ0 << k;
    }
    if (w[0] == 1.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 0.0) flags |= ER_COPY;
    // formant-strength bells                                        SillySampler.py:817-830
    // 1/sigma and -0.5*log2(e): the bell exp(-0.5 ((f - F)/sigma)^2) goes through the hardware exp2 in k_env_rows.  Bins a bell
    // cannot move: the factor 1.0f + g*wt rounds to exactly 1.0f once |g| wt < 2^-25, i.e. beyond z^2 > (25 + log2|g|) /
    // (0.5 log2 e).  Two more bits and a bin on either side cover the hardware exp2 / log2 and the rounding of the bin
    // frequency; a 64-bin chunk wholly outside the reach skips the bell (a wave-uniform branch), which is most of them: sigma
    // is 100-500 Hz against 22 kHz of bins.  The product is unchanged bit for bit.  The reach is a bit per chunk it touches
    // (rounded outwards: an extra chunk only multiplies by exactly 1.0f).
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sv = pp->fst[k];
        const float Fl = a.fst_tracks[orow * 4 + k];
        const bool on = !(fabs(sv) < 1e-6) && isfinite(Fl) && !(Fl <= 50.0f) && !(Fl >= eg.nyq_f);
        const float gl = (float)((1.0 + sv) - 1.0);              // python-float (gain - 1.0), weak-cast to fp32
        const float sgl = k == 0 ? 100.0f : (k == 1 ? 200.0f : (k == 2 ? 350.0f : 500.0f));
        unsigned cm = 0;
        if (on) {
            const float lg = __builtin_amdgcn_logf(fabsf(gl));                    // log2
            const float z2 = (27.0f + fmaxf(lg, 0.0f)) * 1.3862943611198906f;     // / (0.5 log2 e)
            const float R = __builtin_amdgcn_sqrtf(z2) * sgl * 1.0001f;           // (x sigma for / (1/sigma): rounded outwards)
            const float blo = (Fl - R) * eg.inv_fstep - 1.0f, bhi = (Fl + R) * eg.inv_fstep + 1.0f;
            const int c_lo = (int)floorf(fminf(fmaxf((blo - 63.0f) * (1.0f / 64.0f), 0.0f), 31.0f));
            const int c_hi = (int)ceilf(fminf(fmaxf(bhi * (1.0f / 64.0f), -1.0f), 31.0f));
            if (c_hi >= c_lo) cm = (0xffffffffu >> (31 - c_hi)) & (0xffffffffu << c_lo);
        }
        rc.F[k] = Fl;
        rc.g[k] = gl;
        rc.cm[k] = cm;
    }
    rc.irm1 = 0.f;
    rc.note = note;
    rc.pad[0] = rc.pad[1] = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) rc.thr[q] = B;
#pragma unroll
    for (int q = 0; q < 6; ++q) rc.seg[q] = warp_seg_f32{0.f, 0.f, 0.f, 0.f};
    if (WARP) {
        const goofer_note_params *qp = w_params + note;
        double fs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fs[k] = qp->f_shift[k];
        const double ratio = (double)qp->formant_shift;
        const bool stage1 = w_formants != nullptr && (fs[0] != 1.0 || fs[1] != 1.0 || fs[2] != 1.0 || fs[3] != 1.0);
        const bool stage2 = ratio != 1.0;
        if (stage1 || stage2) flags |= ER_WARP;
        if (stage2) {
            flags |= ER_STAGE2;
            rc.irm1 = (float)(fast_rcp(ratio) - 1.0);
        }
        if (stage1) {
            flags |= ER_STAGE1;
            // anchors: (0,0), valid (shifted -> orig) in formant order, (nyq, nyq)      GOOFER.py:850-865
            double x[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            int len = 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double fo = w_formants[orow * 4 + k];
                const double fsft = fo * fs[k];
                const bool valid = fo > 50.0 && fo < nyq && fsft > 50.0;
#pragma unroll
                for (int q = 1; q <= 4; ++q)
                    if (valid && q == len) { x[q] = fsft; y[q] = fo; }
                len += valid ? 1 : 0;
            }
#pragma unroll
            for (int q = 1; q <= 5; ++q)
                if (q == len) { x[q] = nyq; y[q] = nyq; }
            ++len;
            bool sorted = true;
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q < len - 1 && !(x[q] <= x[q + 1])) sorted = false;
            if (sorted) {
                flags |= ER_SORTED;
                auto xb = [&](int c) { return c >= B - 1 ? nyq : (double)c * step; };
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (q < len) {
                        int thr = B;
                        if (q >= 1) {
                            const double dk = x[q];
                            const double est = ceil(dk * inv_step);
                            int c = est < 0.0 ? 0 : (est > (double)(B - 1) ? B - 1 : (int)est);
                            if (c > 0 && dk <= xb(c - 1)) --c;
                            else if (c < B && !(dk <= xb(c))) ++c;
                            thr = c;
                            rc.thr[q - 1] = thr;
                        }
                        const double xn = x[q < 5 ? q + 1 : 5], yn = y[q < 5 ? q + 1 : 5];
                        const double sl = q < len - 1 ? (yn - y[q]) * fast_rcp(xn - x[q]) : 0.0;
                        const double ck = q >= 1 ? (double)thr : 0.0;
                        const double A = (y[q] - sl * x[q]) * inv_step;
                        rc.seg[q] = warp_seg_f32{(float)(A + (sl - 1.0) * ck), (float)(sl - 1.0), (float)ck, 0.f};
                    }
                }
            }
        }
    }
    rc.flags = flags;
    recs[orow] = rc;
}

template <bool WARP, int CH>
__global__ __launch_bounds__(256) void k_env_rows(const goofer_assembly a, int64_t total_out_rows, const env_row_rec *__restrict__ recs,
                                                  const double *__restrict__ w_formants, const goofer_note_params *__restrict__ w_params,
                                                  float *__restrict__ w_out, const env_loop_grid eg)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_seg[A_ROWS][WARP_SEG_DOUBLES];
    const int B = a.n_bins;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // uniform row: the record comes in through scalar loads
    const int64_t orow = (int64_t)blockIdx.x * A_ROWS + wave;
    if (orow >= total_out_rows) return;                      // no block barrier below
    const env_row_rec &rc = recs[orow];
    float *ra = reinterpret_cast<float *>(smem) + (size_t)(2 * wave) * (B + 1), *rb = ra + B + 1;
    const float isig[4] = {1.0f / 100.0f, 1.0f / 200.0f, 1.0f / 350.0f, 1.0f / 500.0f};
    const float topf = (float)(B - 1);
    const bool nt = eg.nt != 0;
    const int flags = rc.flags;
    // (the matrices never overlap: without `restrict` every load of a later chunk has to stay behind the stores of the earlier
    // ones, and a row became nine dependent round trips to memory instead of one — which, not the arithmetic, was k_env_loop's time)
    const float *__restrict__ edit = a.edit_rows;
    float *__restrict__ out = a.env_out + orow * (int64_t)a.ld;
    const bool copy = (flags & ER_COPY) != 0;
    auto row_ptr = [&](int k) { return edit + (uint64_t)rc.src[k] * (uint32_t)a.ld; };
    unsigned cmk[4];
    float Fk[4], gk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cmk[k] = rc.cm[k];
        Fk[k] = rc.F[k];
        gk[k] = rc.g[k];
    }
    const unsigned cm_any = cmk[0] | cmk[1] | cmk[2] | cmk[3];
    const bool warp_on = WARP && (flags & ER_WARP) != 0;
    auto bells = [&](int c, float fb) {
        float gain = 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((cmk[k] >> c) & 1u) {
                const float z = (fb - Fk[k]) * isig[k];
                const float wt = __builtin_amdgcn_exp2f((z * z) * -0.72134752044448170368f);
                gain *= 1.0f + gk[k] * wt;
            }
        }
        return gain;
    };
    // np.linspace(0, sr/2, B) as fp32: the plan's table where the plan has this many bins
    auto bin_freq = [&](int b) { return eg.freqs_f ? eg.freqs_f[b] : (float)(b >= B - 1 ? (double)a.sr / 2.0 : (double)b * eg.fstep); };
    const bool stage1 = (flags & ER_STAGE1) != 0, stage2 = (flags & ER_STAGE2) != 0;
    const bool fast1 = warp_on && stage1 && (flags & ER_SORTED) != 0;
    constexpr int CHS = (CH > 0 && CH <= 18) ? CH : 1;
    constexpr int NWS = (CHS + 8) / 9;
    uint32_t jws[NWS] = {0};                                 // packed segment index of the lane's bin in every chunk (sorted-anchor warp)
    __shared__ __align__(16) float s_rec[A_ROWS][24];                      // ... and the row's six segment records, staged once per row
    if constexpr (CH > 0 && CH <= 18) {
        // the records come in with the row's other loads (behind the row's stores they would wait for those to be acknowledged: one
        // counter, in order): one dword per lane into LDS, read back per bin where the lerp wants them — per-lane copies of every
        // chunk's record were 27 registers held across the row (96 VGPRs, five waves per SIMD)
        if (fast1) {
            const int tk[5] = {rc.thr[0], rc.thr[1], rc.thr[2], rc.thr[3], rc.thr[4]};
            warp_seg_words<NWS>(tk, lane, jws);
            if (lane < 24) s_rec[wave][lane] = reinterpret_cast<const float *>(rc.seg)[lane];
        }
    }
    if constexpr (CH > 0) {
        // every load of the row first — the source row(s), the bin frequencies of the chunks a bell reaches — then the arithmetic
        float v[CH], fb[CH];
        auto idx = [&](int c) {
            const int b = c * WAVE + lane;
            return (c < CH - 1 || b < B) ? b : B - 1;
        };
        const float *s0 = row_ptr(0);
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = s0[idx(c)];
        if (!copy) {
            // zero-weight taps are skipped (their rows may hold anything); the others are one product and FMAs in tap order —
            // an L1 mirror mean (0.5, 0.5) is still the exact fp32 (a + b) / 2
            float t1[CH], t2[CH], t3[CH];
            const float w0 = rc.w[0], w1 = rc.w[1], w2 = rc.w[2], w3 = rc.w[3];
            if (flags & (ER_NZ0 << 1)) {
                const float *s1 = row_ptr(1);
#pragma unroll
                for (int c = 0; c < CH; ++c) t1[c] = s1[idx(c)];
            }
            if (flags & (ER_NZ0 << 2)) {
                const float *s2 = row_ptr(2);
#pragma unroll
                for (int c = 0; c < CH; ++c) t2[c] = s2[idx(c)];
            }
            if (flags & (ER_NZ0 << 3)) {
                const float *s3 = row_ptr(3);
#pragma unroll
                for (int c = 0; c < CH; ++c) t3[c] = s3[idx(c)];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float vf = 0.f;
                if (flags & (ER_NZ0 << 0)) vf = w0 * v[c];
                if (flags & (ER_NZ0 << 1)) vf = __builtin_fmaf(w1, t1[c], vf);
                if (flags & (ER_NZ0 << 2)) vf = __builtin_fmaf(w2, t2[c], vf);
                if (flags & (ER_NZ0 << 3)) vf = __builtin_fmaf(w3, t3[c], vf);
                v[c] = vf;
            }
        }
        if (cm_any) {
            // (all nine table loads at once, needed or not: a load behind a per-chunk test would wait out its own round trip)
            if (eg.freqs_f) {
#pragma unroll
                for (int c = 0; c < CH; ++c) fb[c] = eg.freqs_f[idx(c)];
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) fb[c] = bin_freq(idx(c));
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if ((cm_any >> c) & 1u) v[c] *= bells(c, fb[c]);
        }
        if (!warp_on) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int b = c * WAVE + lane;
                if (c < CH - 1 || b < B) store_f1(out + b, v[c], nt);
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int b = c * WAVE + lane;
            if (c < CH - 1 || b < B) {
                ra[b] = v[c];
                if (c == CH - 1 && b == B - 1) ra[B] = v[c];            // pad element of the fp32 lerp
            }
        }
        if (fast1 || !stage1) {
            // the usual warps, straight from the LDS row into registers; both rows' stores leave together at the end
            wave_lds_sync();
            float *__restrict__ wo = w_out + orow * (int64_t)a.ld;
            float u[CH];
            if (stage1) {
                // sorted anchors: delta(b) = D_k + (s_k - 1)(b - c_k) on segment k, 2-tap lerp (warp_row's fp32 path)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int b = lane + WAVE * c;
                    const float bf = (float)b;
                    const warp_seg_f32 as = *reinterpret_cast<const warp_seg_f32 *>(&s_rec[wave][4 * warp_seg_at<NWS>(jws, c < CHS ? c : 0)]);
                    u[c] = (c < CH - 1 || b < B) ? warp_lerp_f32(ra, b, bf, __builtin_fmaf(as.s, bf - as.c, as.d), topf) : 0.f;
                }
                if (stage2) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int b = lane + WAVE * c;
                        if (c < CH - 1 || b < B) {
                            rb[b] = u[c];
                            if (c == CH - 1 && b == B - 1) rb[B] = u[c];
                        }
                    }
                    wave_lds_sync();
                }
            }
            if (stage2) {
                const float *cur = stage1 ? rb : ra;
                const float irm1 = rc.irm1;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int b = lane + WAVE * c;
                    const float bf = (float)b;
                    u[c] = (c < CH - 1 || b < B) ? warp_lerp_f32(cur, b, bf, irm1 * bf, topf) : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int b = c * WAVE + lane;
                if (c < CH - 1 || b < B) store_f1(out + b, v[c], nt);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int b = c * WAVE + lane;
                if (c < CH - 1 || b < B) store_f1(wo + b, u[c], nt);
            }
            return;
        }
        // (crossing anchors: below)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int b = c * WAVE + lane;
            if (c < CH - 1 || b < B) store_f1(out + b, v[c], nt);
        }
    } else {
        const float *src0 = row_ptr(0), *src1 = row_ptr(copy ? 0 : 1), *src2 = row_ptr(copy ? 0 : 2), *src3 = row_ptr(copy ? 0 : 3);
        const float w0 = rc.w[0], w1 = rc.w[1], w2 = rc.w[2], w3 = rc.w[3];
        for (int c = 0, b = lane; b < B; ++c, b += WAVE) {
            float vf = 0.f;
            if (copy) {
                vf = src0[b];
            } else {
                if (flags & (ER_NZ0 << 0)) vf = w0 * src0[b];
                if (flags & (ER_NZ0 << 1)) vf = __builtin_fmaf(w1, src1[b], vf);
                if (flags & (ER_NZ0 << 2)) vf = __builtin_fmaf(w2, src2[b], vf);
                if (flags & (ER_NZ0 << 3)) vf = __builtin_fmaf(w3, src3[b], vf);
            }
            const int cc = c < 31 ? c : 31;
            if ((cm_any >> cc) & 1u) vf *= bells(cc, bin_freq(b));
            store_f1(out + b, vf, nt);
            if (warp_on) {
                ra[b] = vf;
                if (b == B - 1) ra[B] = vf;
            }
        }
    }
    if (!warp_on) return;
    float *__restrict__ wo = w_out + orow * (int64_t)a.ld;
    wave_lds_sync();
    if (stage1 && !(flags & ER_SORTED)) {
        // crossing anchors: np.interp's guess chain, resolved across the lanes (warp_row) — and the uniform stage behind it
        const goofer_note_params &q = w_params[rc.note];
        double fs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) fs[k] = q.f_shift[k];
        const float *cur = warp_row<CH, true>(ra, rb, B, eg.warp, w_formants + orow * 4, fs, true, (double)q.formant_shift, lane, s_seg[wave]);
        for (int b = lane; b < B; b += WAVE) store_f1(wo + b, cur[b], nt);
        return;
    }
    const float *cur = ra;
    if (stage1) {
        // sorted anchors: delta(b) = D_k + (s_k - 1)(b - c_k) on segment k, 2-tap lerp (warp_row's fp32 path); the segment
        // records are read per lane from the row's record (16-byte loads that hit the lines the scalar loads brought in), all
        // of a lane's bins before the first lerp
        const int tk[5] = {rc.thr[0], rc.thr[1], rc.thr[2], rc.thr[3], rc.thr[4]};
        const warp_seg_f32 *segs = rc.seg;
        auto lerp_bin = [&](int b, const warp_seg_f32 &as, bool to_lds) {
            const float bf = (float)b;
            const float v = warp_lerp_f32(ra, b, bf, __builtin_fmaf(as.s, bf - as.c, as.d), topf);
            if (to_lds) {
                rb[b] = v;
                if (b == B - 1) rb[B] = v;
            } else {
                store_f1(wo + b, v, nt);
            }
        };
        for (int b = lane; b < B; b += WAVE)
            lerp_bin(b, segs[(b >= tk[0]) + (b >= tk[1]) + (b >= tk[2]) + (b >= tk[3]) + (b >= tk[4])], stage2);
        if (stage2) wave_lds_sync();
        cur = rb;
    }
    if (stage2) {
        const float irm1 = rc.irm1;
        if (CH > 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int b = lane + WAVE * c;
                const float bf = (float)b;
                if (c < CH - 1 || b < B) store_f1(wo + b, warp_lerp_f32(cur, b, bf, irm1 * bf, topf), nt);
            }
        } else {
            for (int b = lane; b < B; b += WAVE) {
                const float bf = (float)b;
                store_f1(wo + b, warp_lerp_f32(cur, b, bf, irm1 * bf, topf), nt);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// source mask value at index q of the (possibly reversed) source, before tiling.  Indices inside a note are 32-bit: the
// plan's lengths are int32, and 64-bit integer arithmetic and int64 -> double conversions are several instructions each.
__device__ __forceinline__ double mask_src(const float *__restrict__ m, const goofer_note_plan &p, int idx)
{
    if (p.force_voiced) return 1.0;
    return (double)m[p.reverse ? (int)p.ylen - 1 - idx : idx];
}

// k mod d for a wave-uniform divisor (the tail length) with rd ~ 1 / d (hardware reciprocal): for k < 2^21 the float quotient
// estimate is within one of the true quotient, and one conditional correction finishes it — seven full-rate instructions instead of the ~30 (two of them quarter-rate multiplies) of the
// generic 32-bit remainder.
__device__ __forceinline__ uint32_t mod_small(uint32_t k, uint32_t d, float rd)
{
    if (d >= (1u << 24) || k >= (1u << 21)) return k % d;    // (47 s of loop at 44.1 kHz: never, in practice)
    const uint32_t q = (uint32_t)((float)k * rd);            // within one of floor(k / d): the estimate is good to 2e-7 * 2^21 < 1/2
    uint32_t qd;                                             // (the compiler lowers __umul24 to the quarter-rate v_mul_lo_u32 here)
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(qd) : "v"(q), "v"(d));
    int32_t r = (int32_t)(k - qd);
    if (r < 0) r += (int32_t)d;
    else if (r >= (int32_t)d) r -= (int32_t)d;
    return (uint32_t)r;
}

// ... for callers that have checked d < 2^24 and k < 2^21 themselves
__device__ __forceinline__ uint32_t mod_small_nb(uint32_t k, uint32_t d, float rd)
{
    const uint32_t q = (uint32_t)((float)k * rd);
    uint32_t qd;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(qd) : "v"(q), "v"(d));
    int32_t r = (int32_t)(k - qd);
    r = r < 0 ? r + (int32_t)d : (r >= (int32_t)d ? r - (int32_t)d : r);
    return (uint32_t)r;
}

// mask of the assembled note BEFORE the velocity stretch, at index q in [0, n_before_vel); rd ~ 1 / tail_len
__device__ __forceinline__ double mask_stage1(const float *__restrict__ m, const goofer_note_plan &p, int q, float rd)
{
    if (q < p.n_pre) return mask_src(m, p, p.s_pre + q);
    uint32_t k = (uint32_t)(q - p.n_pre);
    if (p.tail_len < p.want_samples) k = mod_small(k, (uint32_t)p.tail_len, rd);
    return mask_src(m, p, p.s_tail + (int)k);
}

// x / d with r = RN(1 / d) at hand: q = RN(x r), the FMA residual x - q d is exact, and RN(q + residual r) is the
// correctly rounded quotient (Markstein 1990) for operands and quotients in the normal range — three FMAs instead of
// the dozen instructions of the IEEE division sequence.
__device__ __forceinline__ double div_by(double x, double d, double r)
{
    const double q = x * r;
    return fma(fma(-q, d, x), r, q);
}

// 2^x in fp64 for |x| < 1000: round-to-nearest split x = n + f, |f| <= 1/2, and the degree-12 Taylor polynomial of
// e^(f ln 2) with the powers of ln 2 folded into the coefficients (truncation 1.7e-16, Horner rounding ~2e-16 relative).
// libm's exp2 costs 80 vector instructions per sample here (half of them moves of its table constants); the curve only
// has to agree with the reference's pow(2, x) well below the fp32 rounding of f0 (6e-8) — either is a 1e-16 approximation.
// ln2^k / k!, k = 12 .. 1: read through the scalar cache into SGPRs once per wave (VOP3 has no 64-bit literals: as
// immediates every coefficient costs two v_mov per use).  Not `static`: a symbol the compiler must load, not fold.
__constant__ double EXP2_C[12] = {2.5678435993488206e-11, 4.4455382718708116e-10, 7.054911620801123e-09, 1.01780860092397e-07,
                                         1.321548679014431e-06,  1.5252733804059841e-05, 0.0001540353039338161,  0.0013333558146428443,
                                         0.009618129107628477,   0.05550410866482158,    0.24022650695910072,    0.6931471805599453};
__device__ __forceinline__ double exp2_poly(double x)
{
    const double n = rint(x), f = x - n;
    double q = EXP2_C[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) q = fma(q, f, EXP2_C[k]);
    q = fma(q, f, 1.0);
    return ldexp(q, (int)n);
}

__device__ __forceinline__ void sample_assemble_one(const goofer_assembly &a, const goofer_note_plan &p, int64_t g)
{
    const int i = (int)(g - p.out_sample_off);
    const float *m = a.mask_src + p.src_sample_off;
    const float rd_tail = __builtin_amdgcn_rcpf((float)p.tail_len);   // per note

    // voicing mask: direct, or np.interp over the pre-velocity sequence at old_pos   :176-187, 787-788
    double mk;
    if (p.vel_active) {
        const double pos = i < p.pre_new ? (double)i / p.vel_factor : (double)(i - p.pre_new) + (double)p.n_pre;
        const int n1 = p.n_before_vel;
        int j = pos >= 2147483647.0 ? n1 - 1 : (int)floor(pos);
        if (j > n1 - 1) j = n1 - 1;
        if (j < 0) j = 0;
        if (j >= n1 - 1) {
            mk = mask_stage1(m, p, n1 - 1, rd_tail);
        } else {
            const double y0 = mask_stage1(m, p, j, rd_tail), y1 = mask_stage1(m, p, j + 1, rd_tail);
            mk = pos == (double)j ? y0 : (y1 - y0) * (pos - (double)j) + y0;
        }
    } else {
        mk = mask_stage1(m, p, i, rd_tail);
    }

    // pitch curve: bend cents/100 + MIDI (+t), ticks of 60/(tempo*96) s, clamped linear interpolation
    const double *bend = a.bend + p.bend_off;                // MIDI semitones per tick, built on the host like the reference
    const double sr = (double)a.sr;
    double tsec = div_by((double)i, sr, 1.0 / sr);           // np.arange(n) / sr
    const double dt = p.tick_dt, t_last = (double)(p.n_bend - 1) * dt;
    tsec = tsec < 0.0 ? 0.0 : (tsec > t_last ? t_last : tsec);
    double midi;
    if (p.n_bend == 1) {
        midi = bend[0];
    } else {
        // tick index: the estimate tsec * RN(1 / dt) is within one of the answer on the true tick grid k * dt (relative
        // error 3e-16 on a quotient below 2^31), so one conditional step either way settles it
        const double rdt = fast_rcp(dt);                     // per note: hoisted out of the per-sample code by the compiler
        int j = (int)(tsec * rdt);
        if (j > p.n_bend - 1) j = p.n_bend - 1;
        double x0 = (double)j * dt, x1 = (double)(j + 1) * dt;
        if (j + 1 <= p.n_bend - 1 && x1 <= tsec) {
            ++j;
            x0 = x1;
            x1 = (double)(j + 1) * dt;
        } else if (j > 0 && x0 > tsec) {
            --j;
            x1 = x0;
            x0 = (double)j * dt;
        }
        if (j >= p.n_bend - 1) {
            midi = bend[p.n_bend - 1];
        } else {
            const double y0 = bend[j], y1 = bend[j + 1];
            // 1 / (x1 - x0): the spacing is dt up to the rounding of the two products (4e-14 relative), so one Newton step
            // from RN(1 / dt) is the 1e-16 reciprocal fast_rcp would build from scratch
            const double den = x1 - x0;
            const double rden = fma(fma(-den, rdt, 1.0), rdt, rdt);
            midi = tsec == x0 ? y0 : ((y1 - y0) * rden) * (tsec - x0) + y0;
        }
    }
    const double hz = 440.0 * exp2_poly(div_by(midi - 69.0, 12.0, 0.083333333333333329));   // RN(1/12)
    a.mask_out[g] = (float)mk;
    double f0 = mk * hz;
    if (p.pd_on && a.bend_out) a.bend_out[g] = (float)(midi - p.pd_base);                    // 'pd' bend in semitones   :861-863
    // vocal fry part 1: f0 pulled to fry_hz over a constant stretch plus a linear glide       :883-934
    if (p.fry_dir != 0) {
        const double base_hz = p.fry_hz * (mk > 0.0 ? 1.0 : 0.0);
        if (i >= p.fry_const_lo && i < p.fry_const_hi) {
            f0 = base_hz;
        } else if (i >= p.fry_glide_lo && i < p.fry_glide_hi) {
            const int m = p.fry_glide_hi - p.fry_glide_lo, k = i - p.fry_glide_lo;
            double w;                                         // np.linspace(0, 1, m) or np.linspace(1, 0, m), endpoint pinned
            if (p.fry_dir > 0) w = m > 1 ? (k == m - 1 ? 1.0 : (double)k * (1.0 / (double)(m - 1))) : 0.0;
            else w = m > 1 ? (k == m - 1 ? 0.0 : (double)k * (-1.0 / (double)(m - 1)) + 1.0) : 1.0;
            f0 = (1.0 - w) * base_hz + w * f0;
        }
    }
    a.f0_out[g] = (float)f0;
    if (a.f0_mul_out) a.f0_mul_out[g] = (float)(f0 * a.f0_mul[g]);                           // 'sj' layer f0, one rounding   :1065
}


// SA_SPT samples per thread, strided by the workgroup so loads and stores stay coalesced.  A workgroup starts with two rounds
// of dependent loads (which note?) and the scalar loads of the note's 300-byte plan before its first sample: more samples
// per workgroup amortise that latency.
// The usual note — no velocity stretch, no fry, no 'pd' / 'sj' side outputs — for SA_SPT samples of a thread at once, without a
// branch: first the index arithmetic of all of them (mask source index: slice / tile; tick index of the pitch curve with its
// one-step correction as selects), then ALL their loads (one mask value, two curve knots each) in flight together, then the
// fp64 curve, 2^x and the stores.  sample_assemble_one does the same per sample with a branch at every decision, so the
// thread's samples wait out their round trips to memory one after the other (k_sample_assemble was 0.31 ms alone for the same
// arithmetic).  Same operations on the same values: bit-identical to sample_assemble_one (tested through option "sa_fast" 0).
template <int SA_SPT>
__device__ __forceinline__ void sample_assemble_fast(const goofer_assembly &a, const goofer_note_plan &p, int64_t g0, int64_t total_samples)
{
    const float *__restrict__ m = a.mask_src + p.src_sample_off;
    const double *__restrict__ bend = a.bend + p.bend_off;
    float *__restrict__ mask_out = a.mask_out;
    float *__restrict__ f0_out = a.f0_out;
    const int n_pre = p.n_pre, s_pre = p.s_pre, s_tail = p.s_tail, tail_len = p.tail_len, ylen1 = (int)p.ylen - 1, nb1 = p.n_bend - 1;
    const bool tile = tail_len < p.want_samples, rev = p.reverse != 0, fv = p.force_voiced != 0;
    const float rd_tail = __builtin_amdgcn_rcpf((float)tail_len);
    const double sr = (double)a.sr, rsr = 1.0 / sr;
    const double dt = p.tick_dt, t_last = (double)nb1 * dt, rdt = fast_rcp(dt);
    const int64_t nbase = p.out_sample_off;
    int midx[SA_SPT], ja[SA_SPT], jb[SA_SPT];
    double tsec[SA_SPT], x0[SA_SPT], x1[SA_SPT];
    bool live[SA_SPT];
#pragma unroll
    for (int u = 0; u < SA_SPT; ++u) {
        int64_t g = g0 + threadIdx.x + (int64_t)u * blockDim.x;
        live[u] = g < total_samples;
        if (!live[u]) g = total_samples - 1;                  // (the block's note owns it: n_lo == n_hi covers the block's last sample)
        const int i = (int)(g - nbase);
        // voicing mask: slice of the source, the tail tiled                   SillySampler.py:698-712
        uint32_t k = (uint32_t)(i - n_pre);
        if (tile) k = mod_small_nb(k, (uint32_t)tail_len, rd_tail);
        int idx = i < n_pre ? s_pre + i : s_tail + (int)k;
        midx[u] = rev ? ylen1 - idx : idx;
        // pitch curve: tick index of the sample, estimate and one conditional step either way (see sample_assemble_one)
        double ts = div_by((double)i, sr, rsr);
        ts = ts < 0.0 ? 0.0 : (ts > t_last ? t_last : ts);
        int j = (int)(ts * rdt);
        j = j > nb1 ? nb1 : j;
        const double xa = (double)j * dt, xb = (double)(j + 1) * dt;
        const bool up = j + 1 <= nb1 && xb <= ts;
        const bool dn = !up && j > 0 && xa > ts;
        j += (up ? 1 : 0) - (dn ? 1 : 0);
        tsec[u] = ts;
        x0[u] = (double)j * dt;
        x1[u] = (double)(j + 1) * dt;
        const bool last = j >= nb1;
        ja[u] = last ? nb1 : j;
        jb[u] = last ? nb1 : j + 1;
    }
    float mv[SA_SPT];
    double y0[SA_SPT], y1[SA_SPT];
#pragma unroll
    for (int u = 0; u < SA_SPT; ++u) {
        mv[u] = fv ? 1.0f : m[midx[u]];
        y0[u] = bend[ja[u]];
        y1[u] = bend[jb[u]];
    }
    float fo[SA_SPT];
#pragma unroll
    for (int u = 0; u < SA_SPT; ++u) {
        // 1 / (x1 - x0): the spacing is dt up to the rounding of the two products (4e-14 relative), so one Newton step
        // from RN(1 / dt) is the 1e-16 reciprocal fast_rcp would build from scratch
        const double den = x1[u] - x0[u];
        const double rden = fma(fma(-den, rdt, 1.0), rdt, rdt);
        const double midi = (ja[u] == jb[u] || tsec[u] == x0[u]) ? y0[u] : ((y1[u] - y0[u]) * rden) * (tsec[u] - x0[u]) + y0[u];
        const double hz = 440.0 * exp2_poly(div_by(midi - 69.0, 12.0, 0.083333333333333329));   // RN(1/12)
        fo[u] = (float)((double)mv[u] * hz);
    }
#pragma unroll
    for (int u = 0; u < SA_SPT; ++u) {
        const int64_t g = g0 + threadIdx.x + (int64_t)u * blockDim.x;
        if (live[u]) {
            mask_out[g] = mv[u];
            f0_out[g] = fo[u];
        }
    }
}

template <int SA_SPT>
__global__ __launch_bounds__(256) void k_sample_assemble(const goofer_assembly a, int64_t total_samples, int fast)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x * SA_SPT);
    if (threadIdx.x < WAVE) {
        // notes own [out_sample_off, out_sample_off + n_out): first wave searches the plan array cooperatively
        auto key = [&](int k) { return a.notes[k].out_sample_off; };
        int64_t gl = g0 + (int64_t)blockDim.x * SA_SPT - 1;
        if (gl > total_samples - 1) gl = total_samples - 1;
        const int lo = wave_find(a.n_notes, g0, (int)threadIdx.x, key);
        const int hi = wave_find(a.n_notes, gl, (int)threadIdx.x, key);
        if (threadIdx.x == 0) { s_pair[0] = lo; s_pair[1] = hi; }
    }
    __syncthreads();
    const int n_lo = __builtin_amdgcn_readfirstlane(s_pair[0]), n_hi = __builtin_amdgcn_readfirstlane(s_pair[1]);
    if (n_lo == n_hi) {
        const goofer_note_plan &p = a.notes[n_lo];           // uniform note: the 300-byte plan comes in through scalar loads
        if (fast && !p.vel_active && p.fry_dir == 0 && !(p.pd_on && a.bend_out) && !a.f0_mul_out && p.n_out < (1 << 21) && p.tail_len < (1 << 24) &&
            p.tail_len > 0 && p.n_bend >= 1) {
            sample_assemble_fast<SA_SPT>(a, p, g0, total_samples);
            return;
        }
#pragma unroll
        for (int u = 0; u < SA_SPT; ++u) {
            const int64_t g = g0 + threadIdx.x + (int64_t)u * blockDim.x;
            if (g < total_samples) sample_assemble_one(a, p, g);
        }
    } else {
        for (int u = 0; u < SA_SPT; ++u) {
            const int64_t g = g0 + threadIdx.x + (int64_t)u * blockDim.x;
            if (g >= total_samples) break;
            int note = n_lo;
            while (note + 1 < a.n_notes && a.notes[note + 1].out_sample_off <= g) ++note;
            sample_assemble_one(a, a.notes[note], g);
        }
    }
}

// vocal fry part 1b: frames under the fry mask get their bin axis squeezed by 1 - 0.08 w   SillySampler.py:966-994
__device__ __forceinline__ float plan_fry_mask(const goofer_note_plan &p, int64_t i)
{
    const int a = p.fry_a, b = p.fry_b, fade = p.fry_fade;
    if (i < a || i >= b) return 0.f;
    float v = 1.0f;
    if (fade > 0) {
        const int a1 = b < a + fade ? b : a + fade;
        if (i < a1) {
            const int m = a1 - a, k = (int)(i - a);
            const double w = m > 1 ? (k == m - 1 ? 1.0 : (double)k * (1.0 / (double)(m - 1))) : 0.0;
            v = (float)((double)v * w);
        }
        const int b0 = a > b - fade ? a : b - fade;
        if (i >= b0) {
            const int m = b - b0, k = (int)(i - b0);
            const double w = m > 1 ? (k == m - 1 ? 0.0 : (double)k * (-1.0 / (double)(m - 1)) + 1.0) : 1.0;
            v = (float)((double)v * w);
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void k_env_fry(const goofer_assembly a, int64_t total_out_rows, const int *__restrict__ row_note, int hop)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = a.n_bins;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t orow = (int64_t)blockIdx.x * A_ROWS + wave;
    if (orow >= total_out_rows) return;
    const int note = row_note[orow];
    const goofer_note_plan p = a.notes[note];
    if (p.fry_a >= p.fry_b || p.n_out <= 0) return;
    const int64_t j = orow - p.env_off;
    int64_t c = j * hop + hop / 2;
    if (c > p.n_out - 1) c = p.n_out - 1;
    const float w = plan_fry_mask(p, c);
    if (!(w > 1e-6f)) return;
    const double sc = 1.0 - (double)w * (1.0 - 0.92);
    if (fabs(sc - 1.0) < 1e-6) return;
    float *row = reinterpret_cast<float *>(smem) + (size_t)wave * ((B + 3) & ~3);
    float *g = a.env_out + orow * (int64_t)a.ld;
    for (int b = lane; b < B; b += WAVE) row[b] = g[b];
    wave_lds_sync();
    for (int b = lane; b < B; b += WAVE) {
        double src = (double)b / sc;
        src = src < 0.0 ? 0.0 : (src > (double)(B - 1) ? (double)(B - 1) : src);
        const int lo = (int)floor(src);
        const int hi = lo + 1 < B - 1 ? lo + 1 : B - 1;
        const double fr = src - (double)lo;
        g[b] = (float)((1.0 - fr) * (double)row[lo] + fr * (double)row[hi]);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void k_row_notes(const goofer_note_plan *__restrict__ notes, int n_notes, int64_t total_rows, int which,
                            int *__restrict__ row_note)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total_rows) return;
    int lo = 0, hi = n_notes;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        int64_t off = which == 0 ? notes[mid].edit_off : notes[mid].env_off;
        if (off <= r) lo = mid; else hi = mid;
    }
    row_note[r] = lo;
}

// both row -> note maps of an assembly (edited source rows, output rows) in one launch
__global__ void k_row_notes2(const goofer_note_plan *__restrict__ notes, int n_notes, int64_t edit_rows, int64_t out_rows,
                             int *__restrict__ row_note_edit, int *__restrict__ row_note_out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= edit_rows && r >= out_rows) return;
    int lo = 0, hi = n_notes, lo2 = 0, hi2 = n_notes;
    while (hi - lo > 1 || hi2 - lo2 > 1) {
        if (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (notes[mid].edit_off <= r) lo = mid; else hi = mid;
        }
        if (hi2 - lo2 > 1) {
            const int mid = (lo2 + hi2) >> 1;
            if (notes[mid].env_off <= r) lo2 = mid; else hi2 = mid;
        }
    }
    if (r < edit_rows) row_note_edit[r] = lo;
    if (r < out_rows) row_note_out[r] = lo2;
}

int launch_assemble(goofer_ctx *ctx, const goofer_assembly *a, int *row_note_edit, int *row_note_out, void *row_recs, hipStream_t st)
{
    const int B = a->n_bins;
    // per-kernel HIP events of a profiled run (goofer_profile_begin): stage PROF_ASM0 + which, on the stream the kernel runs on
    hipEvent_t *pq = nullptr;
    if (ctx->prof_on && ctx->prof_asm && ctx->prof_asm_steps < ctx->prof_cap && ctx->prof_asm_steps < (int)sizeof(ctx->prof_asm_mask)) {
        pq = ctx->prof_asm + (size_t)ctx->prof_asm_steps * 6;
        ctx->prof_asm_mask[ctx->prof_asm_steps] = 0;
    }
    auto mark = [&](int which, int edge, hipStream_t s) -> hipError_t {
        if (!pq || !(ctx->prof_only < 0 || ctx->prof_only == PROF_ASM0 + which)) return hipSuccess;
        if (edge) ctx->prof_asm_mask[ctx->prof_asm_steps] |= (unsigned char)(1u << which);
        return hipEventRecord(pq[2 * which + edge], s);
    };
    // f0 and voicing mask first: the pulse chain of the synthesis (a long sequential walk) depends on nothing else, and
    // goofer_render_batch starts it on the side stream while the envelope kernels below are still running
    ctx->early_f0 = nullptr;
    ctx->f0_on_side = false;
    if (a->total_samples > 0) {
        // goofer_render_batch: the kernel goes to the side stream itself — the only consumers of f0 / mask before the frame maps
        // are the pulse walk and placement queued behind it there, and the envelope kernels below then run BESIDE it on the
        // caller's stream instead of behind it (it is 0.3 ms at the head of a 2.6 ms step).  The caller's stream waits for
        // ev_f0 where it first reads f0 / mask (goofer_synth_batch).  Not with the fry edit, which reads them here.
        const bool on_side = ctx->early_req && ctx->ev_f0 && ctx->side && !a->any_fry;
        hipStream_t fst = on_side ? ctx->side : st;
        if (on_side) HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_entry, 0));
        constexpr int spt = 4;                                  // (8 / 16 samples per thread measured no faster)
        const dim3 sgrid((unsigned)((a->total_samples + 256 * spt - 1) / (256 * spt)));
        const int sa_fast = ctx->sa_fast ? 1 : 0;
        HIP_TRY(ctx, mark(2, 0, fst));
        hipLaunchKernelGGL(k_sample_assemble<spt>, sgrid, dim3(256), 0, fst, *a, a->total_samples, sa_fast);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, mark(2, 1, fst));
        if (ctx->early_req && ctx->ev_f0) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_f0, fst));
            ctx->early_f0 = a->f0_out;
            ctx->f0_on_side = on_side;
        }
    }
    {
        const int64_t rows = std::max(a->total_edit_rows, a->total_out_rows);
        if (rows > 0) {
            hipLaunchKernelGGL(k_row_notes2, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, a->notes, a->n_notes, a->total_edit_rows,
                               a->total_out_rows, row_note_edit, row_note_out);
            LAUNCH_CHECK(ctx);
        }
    }
    if (a->total_edit_rows > 0) {
        size_t lds = (size_t)A_ROWS * ((3 * B + 2 * ES_HALO + a->max_K + 3) & ~3) * sizeof(float);
        if (lds > 64 * 1024) {
            if (lds > 160 * 1024) return goofer_fail(ctx, GOOFER_EINVAL, "envelope rows too wide for the edit kernel's LDS staging");
            const void *fns[] = {(const void *)k_env_edit<true, 0>, (const void *)k_env_edit<true, 9>, (const void *)k_env_edit<true, 17>,
                                 (const void *)k_env_edit<false, 0>, (const void *)k_env_edit<false, 9>, (const void *)k_env_edit<false, 17>};
            for (const void *fn : fns)
                if (int arc = kernel_allow_max_lds(ctx, fn)) return arc;
        }
        const dim3 egrid((unsigned)((a->total_edit_rows + A_ROWS - 1) / A_ROWS));
        const int echunks = (B + WAVE - 1) / WAVE;
        HIP_TRY(ctx, mark(0, 0, st));
#define ENV_EDIT(V, C) hipLaunchKernelGGL((k_env_edit<V, C>), egrid, dim3(256), lds, st, *a, a->total_edit_rows, row_note_edit)
        if (ctx->value_f64) {
            if (echunks == 9) ENV_EDIT(true, 9);
            else if (echunks == 17) ENV_EDIT(true, 17);
            else ENV_EDIT(true, 0);
        } else {
            if (echunks == 9) ENV_EDIT(false, 9);
            else if (echunks == 17) ENV_EDIT(false, 17);
            else ENV_EDIT(false, 0);
        }
#undef ENV_EDIT
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, mark(0, 1, st));
    }
    if (a->total_out_rows > 0) {
        const dim3 lgrid((unsigned)((a->total_out_rows + A_ROWS - 1) / A_ROWS));
        ctx->warp_done = false;
        const bool fused_warp = ctx->warp_out && !a->any_fry;   // (the fry edit rewrites rows afterwards: the warp then stays a pass of its own)
        const size_t lds_w = fused_warp ? sizeof(float) * 2 * A_ROWS * (B + 1) : 0;
        const double *wf = fused_warp ? ctx->warp_formants : nullptr;
        const goofer_note_params *wp = fused_warp ? ctx->warp_params : nullptr;
        float *wo = fused_warp ? ctx->warp_out : nullptr;
        env_loop_grid eg;
        eg.nt = 0;
        eg.fstep = ((double)a->sr / 2.0) / (double)(B - 1);
        eg.inv_fstep = (float)(1.0 / eg.fstep);
        eg.nyq_f = (float)((double)a->sr * 0.5);
        eg.warp = make_warp_grid(ctx->plan.sr, B);
        eg.freqs_f = (ctx->plan.n_bins == B && ctx->plan.sr == a->sr) ? ctx->plan.lin_freqs : nullptr;
        const int chunks = (B + WAVE - 1) / WAVE;
        env_row_rec *recs = reinterpret_cast<env_row_rec *>(row_recs);
        const dim3 tgrid((unsigned)((a->total_out_rows + 255) / 256));
        if (fused_warp) hipLaunchKernelGGL(k_row_recs<true>, tgrid, dim3(256), 0, st, *a, a->total_out_rows, row_note_out, wf, wp, recs, eg);
        else hipLaunchKernelGGL(k_row_recs<false>, tgrid, dim3(256), 0, st, *a, a->total_out_rows, row_note_out, wf, wp, recs, eg);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, mark(1, 0, st));
#define ENV_ROWS(W, C) hipLaunchKernelGGL((k_env_rows<W, C>), lgrid, dim3(256), lds_w, st, *a, a->total_out_rows, recs, wf, wp, wo, eg)
        if (fused_warp) {
            if (chunks == 9) ENV_ROWS(true, 9);
            else if (chunks == 17) ENV_ROWS(true, 17);
            else ENV_ROWS(true, 0);
            ctx->warp_done = true;
        } else {
            if (chunks == 9) ENV_ROWS(false, 9);
            else if (chunks == 17) ENV_ROWS(false, 17);
            else ENV_ROWS(false, 0);
        }
#undef ENV_ROWS
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, mark(1, 1, st));
        LAUNCH_CHECK(ctx);
        if (a->any_fry) {
            size_t lds = (size_t)A_ROWS * ((B + 3) & ~3) * sizeof(float);
            hipLaunchKernelGGL(k_env_fry, dim3((unsigned)((a->total_out_rows + A_ROWS - 1) / A_ROWS)), dim3(256), lds, st, *a,
                               a->total_out_rows, row_note_out, ctx->plan.hop);
            LAUNCH_CHECK(ctx);
        }
    }
    if (pq) ctx->prof_asm_steps++;
    return GOOFER_OK;
}
