// Note assembly on the device — executes the plans of goofer_amd/sampler.py (SillySampler.resample).
//
//   k_env_edit         knot decode + br tilt + es smooth/sharpen + fw width warp on the source rows a note
//                      uses                                 GOOFER.py:149-168, SillySampler.py:502-574
//   k_env_loop         4-tap frame gather (slicing, L0 cross-fades / L1 mirror mean / L2 stretch, velocity
//                      prefix stretch) + formant-strength gain bells
//                                                           SillySampler.py:625-696, 765-773, 791-833
//   k_sample_assemble  per-sample voicing mask (slice, tile, reverse, force-voiced, velocity stretch) and
//                      pitch curve -> f0                    SillySampler.py:698-712, 787-788, 835-855
// One wave per row for the matrix kernels (row staged in LDS), one thread per sample for the last.
#include <hip/hip_fp16.h>

#include "binops_core.h"

constexpr int A_ROWS = 4;   // rows (waves) per workgroup
constexpr int ES_HALO = 64; // k_env_edit: halo floats either side of the staged row (the 'es' blur has radius <= 28)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_env_edit(const goofer_assembly a, int64_t total_edit_rows, const int *__restrict__ row_note)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = a.n_bins;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t er = (int64_t)blockIdx.x * A_ROWS + wave;
    if (er >= total_edit_rows) return;                       // no block barrier below
    const int stride = (3 * B + 2 * ES_HALO + a.max_K + 3) & ~3;   // floats per wave, 16-byte multiple
    double *tmp = reinterpret_cast<double *>(reinterpret_cast<float *>(smem) + (size_t)wave * stride);   // [B] fp64 scratch
    float *row = reinterpret_cast<float *>(tmp + B) + ES_HALO;   // [B] current row, ES_HALO floats of reflected halo either side
    float *kv = row + B + ES_HALO;                            // [max_K] decoded knot values
    const int note = row_note[er];
    const goofer_note_plan p = a.notes[note];
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
    for (int b = lane; b < B; b += WAVE) {
        float v;
        if (dense) {
            v = kv[b];
        } else {
            int i = li[b];
            v = expf(l0[b] * kv[i] + l1[b] * kv[i + 1]);
        }
        if (tilt) v *= tilt[b];                               // 2. br: env *= tilt (fp32)   :513-515
        row[b] = v;
    }
    wave_lds_sync();

    // 3. es: smooth (blur, rematch frame mean, clamp) or sharpen (unsharp, clamp, rematch)   :517-551
    if (p.es_mode) {
        const double *taps = a.es_taps + p.es_taps_off;
        const int rad = p.es_radius;
        double s_src = 0.0, s_mod = 0.0;
        const bool halo = rad <= ES_HALO && rad < B;          // numpy 'reflect' halo parked beside the row: no index map per tap
        if (halo) {
            for (int h = lane; h < 2 * rad; h += WAVE) {
                const int i = h < rad ? -1 - h : B + (h - rad);
                row[i] = row[(int)reflect_index(i, B)];
            }
            wave_lds_sync();
        }
        for (int b = lane; b < B; b += WAVE) {
            double acc = 0.0;
            if (halo) {
                const float *x = row + (b - rad);
                for (int j = 0; j <= 2 * rad; ++j) acc += taps[j] * (double)x[j];
            } else {
                for (int j = 0; j <= 2 * rad; ++j) acc += taps[j] * (double)row[reflect_index(b + j - rad, B)];
            }
            double src = (double)row[b];
            double mod = p.es_mode == 1 ? acc : fmax(0.0, src + p.es_amount * (src - acc));
            tmp[b] = mod;
            s_src += src;
            s_mod += mod;
        }
        s_src = wave_sum(s_src);
        s_mod = wave_sum(s_mod);
        const float m0 = (float)(s_src / (double)B);          // np.mean of the fp32 block row -> fp32
        const double scale = (double)m0 / (s_mod / (double)B + 1e-12);
        wave_lds_sync();
        for (int b = lane; b < B; b += WAVE) {
            float v = (float)(tmp[b] * scale);
            row[b] = p.es_mode == 1 ? fmaxf(0.0f, v) : v;
        }
        wave_lds_sync();
    }

    // 4. fw: affine stretch of the bin axis about its centre, linear interpolation   :553-574
    float *out = a.edit_rows + er * (int64_t)a.ld;
    if (p.fw_plan >= 0) {
        const int *lo = a.fw_lo + (int64_t)p.fw_plan * B, *hi = a.fw_hi + (int64_t)p.fw_plan * B;
        const double *fr = a.fw_frac + (int64_t)p.fw_plan * B;
        for (int b = lane; b < B; b += WAVE) out[b] = (float)((1.0 - fr[b]) * (double)row[lo[b]] + fr[b] * (double)row[hi[b]]);
    } else {
        for (int b = lane; b < B; b += WAVE) out[b] = row[b];
    }
}

// ---------------------------------------------------------------------------------------------
// WARP: also write the row as gf.synthesize's harmonic branch wants it — formant-anchored + uniform warp (GOOFER.py:1004-1017,
// warp_row of binops_core.h) with the synthesis batch's per-note shifts and per-row formants — while the row is at hand
// (goofer_render_batch: one read of the edited rows instead of a second pass over the assembled envelope).
template <bool WARP>
__global__ __launch_bounds__(256) void k_env_loop(const goofer_assembly a, int64_t total_out_rows, const int *__restrict__ row_note,
                                                  const double *__restrict__ w_formants, const goofer_note_params *__restrict__ w_params,
                                                  float *__restrict__ w_out, double nyq_d)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double s_seg[A_ROWS][WARP_SEG_DOUBLES];
    const int B = a.n_bins;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // uniform row: scalar plan / tap loads
    const int64_t orow = (int64_t)blockIdx.x * A_ROWS + wave;
    if (orow >= total_out_rows) return;
    const int note = row_note[orow];
    const goofer_note_plan p = a.notes[note];
    const int64_t t = orow - p.env_off;
    const int32_t *ti = a.tap_idx + (p.tap_off + t) * 4;
    const double *tw = a.tap_w + (p.tap_off + t) * 4;
    const float *src[4];
    double w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = tw[k];
        src[k] = a.edit_rows + (p.edit_off + (ti[k] - p.row_lo)) * (int64_t)a.ld;
    }
    // formant-strength bells for this frame                    SillySampler.py:817-830
    float Fk[4];
    double sv[4];
    bool on[4];
    const float nyq = (float)((double)a.sr * 0.5);
    // 1/sigma and -0.5*log2(e): the bell exp(-0.5 ((f - F)/sigma)^2) through the hardware exp2 (<= 1e-6 relative on
    // the gain where the bell is not negligible) instead of an IEEE division and libm expf per bin and formant
    const float isig[4] = {1.0f / 100.0f, 1.0f / 200.0f, 1.0f / 350.0f, 1.0f / 500.0f};
    float gk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sv[k] = p.fst[k];
        Fk[k] = a.fst_tracks[(p.env_off + t) * 4 + k];
        on[k] = !(fabs(sv[k]) < 1e-6) && isfinite(Fk[k]) && !(Fk[k] <= 50.0f) && !(Fk[k] >= nyq);
        gk[k] = (float)((1.0 + sv[k]) - 1.0);                // python-float (gain - 1.0), weak-cast to fp32
    }
    const double fstep = ((double)a.sr / 2.0) / (double)(B - 1);
    // Bins a bell cannot move: the factor 1.0f + gk*wt rounds to exactly 1.0f once |gk| wt < 2^-25, i.e. beyond
    // z^2 > (25 + log2|gk|) / (0.5 log2 e).  Two more bits and a bin on either side cover the hardware exp2 / log2 and the
    // rounding of fb; a 64-bin chunk wholly outside [blo, bhi] skips the bell (a wave-uniform branch), which is most of
    // them: sigma is 100-500 Hz against 22 kHz of bins.  The product is unchanged bit for bit.
    float blo[4], bhi[4];
    const float inv_fstep = (float)(1.0 / fstep);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float lg = __builtin_amdgcn_logf(fabsf(gk[k]));                 // log2
        const float z2 = (27.0f + fmaxf(lg, 0.0f)) * 1.3862943611198906f;     // / (0.5 log2 e)
        const float R = __builtin_amdgcn_sqrtf(z2) / isig[k];
        blo[k] = on[k] ? (Fk[k] - R) * inv_fstep - 1.0f : 3.0e38f;
        bhi[k] = on[k] ? (Fk[k] + R) * inv_fstep + 1.0f : -3.0e38f;
    }
    float *out = a.env_out + orow * (int64_t)a.ld;
    // a plain copy of one source row (most rows: slices and loop repeats outside the cross-fades): 0.0 + 1.0 x is x, and
    // both roundings of the product below — fp32, or fp64 rounded to fp32 — are the fp32 product, so the row stays in fp32
    const bool copy = w[0] == 1.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 0.0;
    for (int c0 = 0; c0 < B; c0 += WAVE) {
        const int b = c0 + lane;
        if (b >= B) break;
        double v = 0.0;
        float vf = 0.f;
        if (copy) {
            vf = src[0][b];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w[k] != 0.0) v += w[k] * (double)src[k][b];
        }
        float gain = 1.0f;
        const float fb = (float)(b >= B - 1 ? (double)a.sr / 2.0 : (double)b * fstep);    // np.linspace(0, sr/2, B) as fp32
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((float)(c0 + WAVE - 1) >= blo[k] && (float)c0 <= bhi[k]) {
                const float z = (fb - Fk[k]) * isig[k];
                const float wt = __builtin_amdgcn_exp2f((z * z) * -0.72134752044448170368f);
                gain *= 1.0f + gk[k] * wt;
            }
        }
        const float o = copy ? vf * gain : (p.env_f64 ? (float)(v * (double)gain) : ((float)v) * gain);
        out[b] = o;
        if (WARP) reinterpret_cast<float *>(smem)[(size_t)(2 * wave) * B + b] = o;
    }
    if (WARP) {
        float *ra = reinterpret_cast<float *>(smem) + (size_t)(2 * wave) * B, *rb = ra + B;
        wave_lds_sync();
        const goofer_note_params &q = w_params[note];
        double fs[4];
        bool warp = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fs[k] = q.f_shift[k];
            warp |= fs[k] != 1.0;
        }
        const float *cur = warp_row(ra, rb, B, nyq_d, w_formants ? w_formants + orow * 4 : nullptr, fs, warp, (double)q.formant_shift, lane,
                                    s_seg[wave]);
        float *wo = w_out + orow * (int64_t)a.ld;
        for (int b = lane; b < B; b += WAVE) wo[b] = cur[b];
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

// mask of the assembled note BEFORE the velocity stretch, at index q in [0, n_before_vel)
__device__ __forceinline__ double mask_stage1(const float *__restrict__ m, const goofer_note_plan &p, int q)
{
    if (q < p.n_pre) return mask_src(m, p, p.s_pre + q);
    uint32_t k = (uint32_t)(q - p.n_pre);
    if (p.tail_len < p.want_samples) k = k % (uint32_t)p.tail_len;
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

__device__ __forceinline__ void sample_assemble_one(const goofer_assembly &a, const goofer_note_plan &p, int64_t g)
{
    const int i = (int)(g - p.out_sample_off);
    const float *m = a.mask_src + p.src_sample_off;

    // voicing mask: direct, or np.interp over the pre-velocity sequence at old_pos   :176-187, 787-788
    double mk;
    if (p.vel_active) {
        const double pos = i < p.pre_new ? (double)i / p.vel_factor : (double)(i - p.pre_new) + (double)p.n_pre;
        const int n1 = p.n_before_vel;
        int j = pos >= 2147483647.0 ? n1 - 1 : (int)floor(pos);
        if (j > n1 - 1) j = n1 - 1;
        if (j < 0) j = 0;
        if (j >= n1 - 1) {
            mk = mask_stage1(m, p, n1 - 1);
        } else {
            const double y0 = mask_stage1(m, p, j), y1 = mask_stage1(m, p, j + 1);
            mk = pos == (double)j ? y0 : (y1 - y0) * (pos - (double)j) + y0;
        }
    } else {
        mk = mask_stage1(m, p, i);
    }

    // pitch curve: bend cents/100 + MIDI (+t), ticks of 60/(tempo*96) s, clamped linear interpolation
    const double *bend = a.bend + p.bend_off;                // MIDI semitones per tick, built on the host like the reference
    const double sr = (double)a.sr;
    double tsec = div_by((double)i, sr, 1.0 / sr);           // np.arange(n) / sr
    const double t_last = (double)(p.n_bend - 1) * p.tick_dt;
    tsec = tsec < 0.0 ? 0.0 : (tsec > t_last ? t_last : tsec);
    double midi;
    if (p.n_bend == 1) {
        midi = bend[0];
    } else {
        int j = (int)(tsec * fast_rcp(p.tick_dt));           // estimate; the two loops below settle it on the true tick grid
        if (j > p.n_bend - 1) j = p.n_bend - 1;
        while (j + 1 <= p.n_bend - 1 && (double)(j + 1) * p.tick_dt <= tsec) ++j;
        while (j > 0 && (double)j * p.tick_dt > tsec) --j;
        if (j >= p.n_bend - 1) {
            midi = bend[p.n_bend - 1];
        } else {
            const double y0 = bend[j], y1 = bend[j + 1];
            const double x0 = (double)j * p.tick_dt, x1 = (double)(j + 1) * p.tick_dt;
            midi = tsec == x0 ? y0 : ((y1 - y0) * fast_rcp(x1 - x0)) * (tsec - x0) + y0;
        }
    }
    const double hz = 440.0 * exp2(div_by(midi - 69.0, 12.0, 0.083333333333333329));   // RN(1/12)
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


#define SA_SPT 4   // samples per thread, strided by the workgroup so loads and stores stay coalesced

__global__ __launch_bounds__(256) void k_sample_assemble(const goofer_assembly a, int64_t total_samples)
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

int launch_assemble(goofer_ctx *ctx, const goofer_assembly *a, int *row_note_edit, int *row_note_out, hipStream_t st)
{
    const int B = a->n_bins;
    // f0 and voicing mask first: the pulse chain of the synthesis (a long sequential walk) depends on nothing else, and
    // goofer_render_batch starts it on the side stream while the envelope kernels below are still running
    ctx->early_f0 = nullptr;
    if (a->total_samples > 0) {
        hipLaunchKernelGGL(k_sample_assemble, dim3((unsigned)((a->total_samples + 256 * SA_SPT - 1) / (256 * SA_SPT))), dim3(256), 0, st, *a,
                           a->total_samples);
        LAUNCH_CHECK(ctx);
        if (ctx->early_req && ctx->ev_f0) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_f0, st));
            ctx->early_f0 = a->f0_out;
        }
    }
    if (a->total_edit_rows > 0) {
        hipLaunchKernelGGL(k_row_notes, dim3((unsigned)((a->total_edit_rows + 255) / 256)), dim3(256), 0, st, a->notes, a->n_notes,
                           a->total_edit_rows, 0, row_note_edit);
        LAUNCH_CHECK(ctx);
        size_t lds = (size_t)A_ROWS * ((3 * B + 2 * ES_HALO + a->max_K + 3) & ~3) * sizeof(float);
        if (lds > 64 * 1024) {
            if (lds > 160 * 1024) return goofer_fail(ctx, GOOFER_EINVAL, "envelope rows too wide for the edit kernel's LDS staging");
            if (int arc = kernel_allow_max_lds(ctx, (const void *)k_env_edit)) return arc;
        }
        hipLaunchKernelGGL(k_env_edit, dim3((unsigned)((a->total_edit_rows + A_ROWS - 1) / A_ROWS)), dim3(256), lds, st, *a,
                           a->total_edit_rows, row_note_edit);
        LAUNCH_CHECK(ctx);
    }
    if (a->total_out_rows > 0) {
        hipLaunchKernelGGL(k_row_notes, dim3((unsigned)((a->total_out_rows + 255) / 256)), dim3(256), 0, st, a->notes, a->n_notes,
                           a->total_out_rows, 1, row_note_out);
        LAUNCH_CHECK(ctx);
        const dim3 lgrid((unsigned)((a->total_out_rows + A_ROWS - 1) / A_ROWS));
        ctx->warp_done = false;
        if (ctx->warp_out && !a->any_fry) {                   // (the fry edit rewrites rows afterwards: the warp then stays a pass of its own)
            hipLaunchKernelGGL(k_env_loop<true>, lgrid, dim3(256), sizeof(float) * 2 * A_ROWS * B, st, *a, a->total_out_rows, row_note_out,
                               ctx->warp_formants, ctx->warp_params, ctx->warp_out, (double)ctx->plan.sr / 2.0);
            ctx->warp_done = true;
        } else {
            hipLaunchKernelGGL(k_env_loop<false>, lgrid, dim3(256), 0, st, *a, a->total_out_rows, row_note_out, (const double *)nullptr,
                               (const goofer_note_params *)nullptr, (float *)nullptr, 0.0);
        }
        LAUNCH_CHECK(ctx);
        if (a->any_fry) {
            size_t lds = (size_t)A_ROWS * ((B + 3) & ~3) * sizeof(float);
            hipLaunchKernelGGL(k_env_fry, dim3((unsigned)((a->total_out_rows + A_ROWS - 1) / A_ROWS)), dim3(256), lds, st, *a,
                               a->total_out_rows, row_note_out, ctx->plan.hop);
            LAUNCH_CHECK(ctx);
        }
    }
    return GOOFER_OK;
}
