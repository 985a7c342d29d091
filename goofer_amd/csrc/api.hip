// C ABI of libgoofer_hip.so: handle lifetime, per-(sr, n_fft, hop) tables, scratch arena and the
// batch driver that strings the kernels into gf.synthesize (GOOFER.py:971-1220).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "samples_core.h"

// launchers living in the other translation units
int launch_frame_note(goofer_ctx *, const int64_t *, int, int64_t, int *, hipStream_t);
int launch_rfft_frames_mapped(goofer_ctx *, const float *, const int64_t *, const int64_t *, const int *, int64_t, float2 *, int,
                              hipStream_t);
int launch_pulse_peak(goofer_ctx *, float *, double, hipStream_t);
size_t pulse_shape_table_floats();
int launch_pulse_shape_table(goofer_ctx *, float *, const float *, double, hipStream_t);
struct onset_t;
int launch_pulse_train(goofer_ctx *, const float *, float, const int64_t *, int, int64_t, float *, double *, onset_t *, int32_t *,
                       int32_t *, int32_t *, hipStream_t);
int launch_pulse_onsets(goofer_ctx *, const float *, float, const int64_t *, int, onset_t *, int32_t *, int32_t *,
                        int32_t *, int64_t, int32_t *, hipStream_t);
int launch_pulse_place(goofer_ctx *, const onset_t *, const int32_t *, const int64_t *, int, int64_t, float *, const int32_t *, hipStream_t);
int launch_subharm(goofer_ctx *, const float *, const double *, const float *, const int64_t *, int, int64_t, const goofer_note_params *, const double *, int,
                   int, double, double, double, double *, double *, onset_t *, int32_t *, int32_t *, int32_t *, const unsigned char *,
                   double *, unsigned long long *, float *, hipStream_t);
int launch_gauss_bins(goofer_ctx *, const float *, float *, int64_t, int, int, const double *, int, const int64_t *, hipStream_t);
int launch_warp_bins(goofer_ctx *, const float *, float *, int64_t, int, int, const double *, const double *,
                     const goofer_note_params *, const int *, const int64_t *, double, hipStream_t);
int launch_knot_decode(goofer_ctx *, const uint16_t *, int, int64_t, const int *, const float *, const float *, float *, int, int,
                       hipStream_t);
int launch_harm_shape(goofer_ctx *, float2 *, int, int64_t, const int *, const int64_t *, const int64_t *, const float *,
                      const float *, const float *, int, const goofer_note_params *, float *, const int64_t *, const double *,
                      bool, hipStream_t);
int launch_noise_spectra(goofer_ctx *, float2 *, float2 *, int, int64_t, const int *, const int64_t *, const int64_t *,
                         const float *, const float *, const float *, const float *, int, const goofer_note_params *, uint64_t,
                         const int64_t *, bool, const unsigned char *, hipStream_t);
int launch_frame_skip(goofer_ctx *, const double *, int64_t, const int64_t *, const int64_t *, const int *, int, int64_t, unsigned char *,
                      unsigned char *, unsigned char *, hipStream_t);
int launch_mask_short(goofer_ctx *, const float *, const int64_t *, int, int64_t, const double *, int, double, double *, hipStream_t);
int launch_assemble(goofer_ctx *, const goofer_assembly *, int *, int *, void *, hipStream_t);
size_t env_row_rec_bytes();
int launch_mag_rows(goofer_ctx *, const float2 *, int, int64_t, int, float *, int, hipStream_t);
int launch_gauss_rows64(goofer_ctx *, const float *, int, double *, int, int64_t, int, const double *, int, hipStream_t);
int launch_knot_error(goofer_ctx *, const double *, int, const int64_t *, int, int, const int *, int, const int *, const float *,
                      const float *, unsigned long long *, hipStream_t);
int launch_knot_gather(goofer_ctx *, const double *, int, int64_t, const int *, int, uint16_t *, hipStream_t);
int launch_ola3_gains(goofer_ctx *, const float *, const float *, const float *, const float *, const double *, const int64_t *,
                      const int64_t *, int, int64_t, const goofer_note_params *, double *, float *, float *, float *, float *, hipStream_t);
template <typename Tin>
int launch_gauss_samples(goofer_ctx *, const Tin *, const int64_t *, int, int64_t, const double *, int, const unsigned char *, double *,
                         hipStream_t);
int launch_note_absmax(goofer_ctx *, const double *, const int64_t *, int, int64_t, const unsigned char *, unsigned long long *,
                       hipStream_t);
int launch_f0_jitter(goofer_ctx *, float *, double *, const float *, const double *, const unsigned long long *, const int64_t *, int, int64_t,
                     const goofer_note_params *, int, hipStream_t);
int launch_volume_jitter(goofer_ctx *, float *, float *, const double *, const double *, const double *, const unsigned long long *,
                         const unsigned long long *, const int64_t *, int, int64_t, const goofer_note_params *, int, double, hipStream_t);
int launch_onepole(goofer_ctx *, const float *, float *, const float *, const goofer_onepole_job *, int, hipStream_t);
int launch_post_layers(goofer_ctx *, float *, const float *, const float *, const goofer_post_note *, const int64_t *, int, int64_t,
                       hipStream_t);
int launch_post_fry(goofer_ctx *, float *, float *, const float *, const float *, const goofer_post_note *, const int64_t *, int, int64_t,
                    hipStream_t);
int launch_post_sd(goofer_ctx *, float *, const double *, const goofer_post_note *, const int64_t *, int, int64_t, hipStream_t);
int launch_note_sumsq(goofer_ctx *, const float *, const float *, const goofer_post_note *, const int64_t *, int, int64_t, double *,
                      hipStream_t);
int launch_post_tension(goofer_ctx *, float *, float *, const float *, const goofer_post_note *, const int64_t *, int, int64_t,
                        hipStream_t);
int launch_post_scale(goofer_ctx *, float *, float *, const double *, const double *, const goofer_post_note *, const int64_t *, int,
                      int64_t, hipStream_t);
int launch_post_mix(goofer_ctx *, const float *, const float *, const float *, const float *, const float *, const double *,
                    const goofer_post_note *, const unsigned char *, const goofer_note_params *, const int64_t *, int, int64_t, float *,
                    hipStream_t);
int launch_dyn_gain(goofer_ctx *, const double *, const double *, const unsigned char *, double *, const goofer_post_note *,
                    const int64_t *, int, int64_t, double *, hipStream_t);
int launch_irfft_ola3(goofer_ctx *, const float2 *, const float2 *, const float2 *, int, int64_t, const int *, const int64_t *,
                      const int64_t *, int, const float *, const double *, double *, const goofer_note_params *, float *, float *, float *,
                      float *, hipStream_t);
int launch_vocal_roughness(goofer_ctx *, const float *, const float *, const float *, const double *, int, const double *, const double *,
                           double, double, const float *, const int64_t *, int, int64_t, float *, hipStream_t);
int launch_lerp_axis0(goofer_ctx *, const float *, int64_t, int64_t, float *, int64_t, int64_t, int, hipStream_t);
int launch_lerp_1d(goofer_ctx *, const float *, int64_t, float *, int64_t, hipStream_t);
int launch_stem_peak(goofer_ctx *, const float *, const float *, const float *, const int64_t *, int, int64_t, float *, hipStream_t);
int launch_stem_gains(goofer_ctx *, float *, float *, float *, const double *, const int64_t *, int, int64_t,
                      const goofer_note_params *, float *, double *, hipStream_t);
int launch_apply_gain(goofer_ctx *, float *, float *, float *, float *, float *, const int64_t *, int, int64_t,
                      const goofer_note_params *, const float *, bool, hipStream_t);

int launch_note_steps(goofer_ctx *, const int64_t *, int, double *, hipStream_t);
int launch_mask_upsample(goofer_ctx *, const double *, const int64_t *, int, int64_t, double *, bool, float *, hipStream_t);
bool stems_supported(const goofer_plan_t &);
bool ola_split_supported(const goofer_plan_t &);
int launch_irfft_ola1(goofer_ctx *, const float2 *, const float2 *, const float2 *, int, int64_t, const int *, const int64_t *,
                      const int64_t *, int, const double *, double *, const goofer_note_params *, float *, float *, float *,
                      const unsigned char *, hipStream_t);
int launch_frame_picks(goofer_ctx *, const int64_t *, const int *, int64_t, const int64_t *, const float *, const float *, float2 *,
                       hipStream_t);
int launch_noise_stems(goofer_ctx *, const float *, int, const int64_t *, const float *, int64_t, const int *, const int64_t *,
                       const int64_t *, const float2 *, const goofer_note_params *, uint64_t, bool, const double *, const double *, float *,
                       float *, unsigned char *, hipStream_t);
int launch_harm_stem(goofer_ctx *, const float *, const float *, const float *, bool, int, const int64_t *, int64_t, const int *, const int64_t *,
                     const int64_t *, const float2 *, const goofer_note_params *, float *, float *, hipStream_t);
int launch_note_finish(goofer_ctx *, float *, float *, float *, float *, float *, const int64_t *, int, const goofer_note_params *,
                       const float *, float *, bool, const unsigned char *, const int64_t *, hipStream_t);

static const size_t ONSET_BYTES = 24;

int goofer_fail(goofer_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

static goofer_ctx::kernel_state *kstate_of(goofer_ctx *ctx, const void *fn)
{
    for (int i = 0; i < ctx->n_kstate; ++i)
        if (ctx->kstate[i].fn == fn) return &ctx->kstate[i];
    if (ctx->n_kstate >= 32) return nullptr;
    goofer_ctx::kernel_state *k = &ctx->kstate[ctx->n_kstate++];
    *k = {fn, 0, 0, false};
    return k;
}

int kernel_allow_max_lds(goofer_ctx *ctx, const void *fn, int bytes)
{
    goofer_ctx::kernel_state *k = kstate_of(ctx, fn);
    if (k && k->max_lds_set) return GOOFER_OK;
    HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (k) k->max_lds_set = true;
    return GOOFER_OK;
}

int kernel_resident_waves(goofer_ctx *ctx, const void *fn, size_t lds, int *waves)
{
    goofer_ctx::kernel_state *k = kstate_of(ctx, fn);
    if (k && k->waves > 0 && k->lds == lds) {
        *waves = k->waves;
        return GOOFER_OK;
    }
    int cus = 0, per_cu = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
    *waves = (cus > 0 ? cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;
    if (k) { k->waves = *waves; k->lds = lds; }
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// small helper kernels of the batch driver
// picks != nullptr: also the frame's picks of the per-sample arrays, x[::hop] edge-padded to the frame count
// (GOOFER.py:1104-1106), as one (f0, mask) record per frame.  The shaping kernels then find them one dependent load
// earlier (frame -> record) instead of three (frame -> note -> offsets -> sample).
__global__ void k_row_src(const int64_t *__restrict__ frame_off, const int64_t *__restrict__ env_off,
                          const int *__restrict__ frame_note, int64_t total_frames, int64_t *__restrict__ row_src,
                          const int64_t *__restrict__ sample_off, const float *__restrict__ f0, const float *__restrict__ mask,
                          int hop, float2 *__restrict__ picks)
{
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total_frames) return;
    int note = frame_note[f];
    int64_t t = f - frame_off[note];
    if (picks) {
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        float2 pv = make_float2(0.f, 0.f);
        if (n > 0) {
            int64_t at = t * hop;
            if (at >= n) at = ((n - 1) / hop) * hop;          // edge-padded: the last pick
            pv = make_float2(f0[base + at], mask[base + at]);
        }
        picks[f] = pv;
    }
    int64_t rows = env_off[note + 1] - env_off[note];
    if (t > rows - 1) t = rows - 1;     // edge-repeat (np.pad mode='edge'); truncation is implicit
    if (t < 0) t = 0;
    row_src[f] = env_off[note] + t;
}

// The frame maps of the stem path in one launch (they were a memset and three small kernels in a row on the critical path,
// ~10 us of dispatch each): frame -> note, frame -> envelope row, the frame's (f0, mask) picks; per note the two reciprocal
// steps of the mask upsampler and the zeroed maxima the walkers reduce into.
__global__ void k_frame_maps(const int64_t *__restrict__ frame_off, const int64_t *__restrict__ env_off, int n_notes, int64_t total_frames,
                             int *__restrict__ frame_note, int64_t *__restrict__ row_src, const int64_t *__restrict__ sample_off,
                             const float *__restrict__ f0, const float *__restrict__ mask, int hop, float2 *__restrict__ picks,
                             double *__restrict__ steps, float *__restrict__ note_mag)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f < 2 * (int64_t)n_notes) note_mag[f] = 0.f;
    if (f < n_notes) {
        const int64_t n = sample_off[f + 1] - sample_off[f];
        const int64_t ns = (n + MASK_DS - 1) / MASK_DS;
        steps[2 * f] = n > 1 ? 1.0 / (double)(n - 1) : 0.0;
        steps[2 * f + 1] = ns > 1 ? 1.0 / (double)(ns - 1) : 0.0;
    }
    if (f >= total_frames) return;
    const int note = csr_find(frame_off, n_notes, f);
    frame_note[f] = note;
    int64_t t = f - frame_off[note];
    if (picks) {
        const int64_t base = sample_off[note], n = sample_off[note + 1] - base;
        float2 pv = make_float2(0.f, 0.f);
        if (n > 0) {
            int64_t at = t * hop;
            if (at >= n) at = ((n - 1) / hop) * hop;          // edge-padded: the last pick
            pv = make_float2(f0[base + at], mask[base + at]);
        }
        picks[f] = pv;
    }
    const int64_t rows = env_off[note + 1] - env_off[note];
    if (t > rows - 1) t = rows - 1;     // edge-repeat (np.pad mode='edge'); truncation is implicit
    if (t < 0) t = 0;
    row_src[f] = env_off[note] + t;
}

// f0 *= pitch_shift (GOOFER.py:995), fp32.  1024 samples per workgroup, 16-byte accesses when the tile sits in one note.
__global__ __launch_bounds__(256) void k_scale_f0(const float *__restrict__ f0, const int64_t *__restrict__ sample_off, int n_notes,
                                                  int64_t total, const goofer_note_params *__restrict__ params, float *__restrict__ out)
{
    __shared__ int s_pair[2];
    const int64_t g0 = (int64_t)blockIdx.x * 1024;
    int64_t gl = g0 + 1023;
    if (gl > total - 1) gl = total - 1;
    int lo, hi;
    block_note_range_last(sample_off, n_notes, g0, gl, s_pair, lo, hi);
    const int64_t g = g0 + (int64_t)threadIdx.x * 4;
    if (g >= total) return;
    const bool vec = (((uintptr_t)f0 | (uintptr_t)out) & 15) == 0;
    if (lo == hi && g + 4 <= total && vec) {
        const float ps = params[lo].pitch_shift;
        float4 v = *reinterpret_cast<const float4 *>(f0 + g);
        v.x *= ps; v.y *= ps; v.z *= ps; v.w *= ps;
        *reinterpret_cast<float4 *>(out + g) = v;
        return;
    }
    int note = lo;
    for (int k = 0; k < 4 && g + k < total; ++k) {
        while (sample_off[note + 1] <= g + k) ++note;
        const float v = f0[g + k] * params[note].pitch_shift;
        out[g + k] = v;
    }
}

static void gauss_taps_host(double sigma, std::vector<double> &taps, int &radius);
static int ensure_small(goofer_ctx *ctx, size_t bytes);

__global__ void k_note_sub_flags(const goofer_note_params *__restrict__ params, int n_notes, unsigned char *__restrict__ on_sub,
                                 unsigned char *__restrict__ on_subj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_notes) {
        on_sub[i] = params[i].subharm_weight > 0.f;
        on_subj[i] = params[i].subharm_weight > 0.f && params[i].subharm_f0_jitter > 0.0;
    }
}

__global__ void k_note_flags(const goofer_note_params *__restrict__ params, int n_notes, unsigned char *__restrict__ on_f0,
                             unsigned char *__restrict__ on_vol)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_notes) return;
    on_f0[i] = params[i].f0_jitter > 0.0;
    on_vol[i] = params[i].vol_jitter_harm > 0.f || params[i].vol_jitter_breath > 0.f;
}

// taps of a sample-axis Gaussian, uploaded into the handle's small buffer at `slot` (3 slots of 16 KiB after 64 KiB)
// three tap slots behind the 64 KiB of tables in ctx->small: radius <= 8000, i.e. a jitter speed down to sr / 12 000 Hz (3.7 Hz at
// 44.1 kHz; the reference's defaults are 100 and 150 Hz.  Until round 6: radius <= 1000 = 29.4 Hz, which a random keyword set hit)
constexpr size_t JIT_SLOT_BYTES = 131072;
static int upload_jitter_taps(goofer_ctx *ctx, double sigma, int slot, const double **d_taps, int *radius, hipStream_t st)
{
    std::vector<double> taps;
    int r;
    gauss_taps_host(sigma, taps, r);
    if (r > 8000) return goofer_fail(ctx, GOOFER_EINVAL, "jitter sigma %g too large", sigma);
    int rc = ensure_small(ctx, 65536 + 3 * JIT_SLOT_BYTES);
    if (rc) return rc;
    double *dst = (double *)((char *)ctx->small + 65536 + (size_t)slot * JIT_SLOT_BYTES);
    HIP_TRY(ctx, hipMemcpyAsync(dst, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *d_taps = dst;
    *radius = r;
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// scratch arena
struct arena {
    char *base;
    size_t size, used;
    template <typename T> T *take(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > size) return nullptr;
        T *p = reinterpret_cast<T *>(base + used);
        used += bytes;
        return p;
    }
};

// spectra: 1 = the one-kernel-per-step pipeline will run (complex spectra + windowed frames of the three stems in HBM: 24 KB per
// frame, the noise envelope); 0 = the stem walkers, which need none of it.  subharm: the 'sg' trackers' fp64 phase increments.
static size_t scratch_need(const goofer_plan_t &p, int64_t frames, int64_t samples, int64_t notes, int spectra = 1, bool subharm = true)
{
    size_t ldc = spec_stride(p.n_bins), ld = (p.n_bins + 3) & ~3;
    size_t b = 0;
    auto add = [&](size_t bytes) { b += (bytes + 255) & ~(size_t)255; };
    add(frames * sizeof(int));                    // frame_note
    add(frames * sizeof(int64_t));                // row_src
    add(frames * sizeof(float2));                 // per-frame (f0, mask) picks
    add(samples * sizeof(float));                 // f0 scaled
    if (subharm) add(samples * sizeof(double));   // phase increments
    // onset slots: n / 2 + 16 per note for the pulse train (an f0 above sr / 2 is refused) — n + 16 with the sub-harmonic layer,
    // whose tracker fires at most once per sample and does so on every sample once its increment passes 1 (the resampler's
    // vibrato depth of 3 takes the layer to 8 x f0: above sr / 2 from F7 on)
    const size_t slots = (size_t)(subharm ? samples : samples / 2) + 16 * (size_t)notes + 16;
    add(slots * ONSET_BYTES);
    add(slots * sizeof(int32_t));                 // raw onset sample indices
    add(notes * sizeof(int32_t) + 64);            // onset counts
    add(64);                                      // overflow flag
    add(samples * sizeof(float));                 // pulse
    add((size_t)PULSE_TILE_INTS(samples) * sizeof(int32_t)); // pulse placement: 4 ints per tile
    if (spectra == 1) {
        add(3 * frames * ldc * sizeof(float2));       // S_h, S_uv, S_br
        add(3 * frames * (size_t)p.n_fft * sizeof(float));  // windowed time frames (three stems in the fused path)
    }
    add((spectra == 1 ? 2 : 1) * frames * ld * sizeof(float));   // env_h (, env_n)
    add((samples / 4 + notes + 16) * sizeof(double));  // smoothed decimated mask
    add(2 * notes * sizeof(float) + 64);          // note_mag, note_peak
    add(2 * notes * sizeof(double) + 64);         // per-note linspace steps
    add((size_t)frames + 3 * (size_t)notes + 64); // per-hop stem sparsity bytes of the walkers
    if (spectra) {                                // per-hop flatness + per-frame skip bits of the LDS-ring pipeline
        const int64_t reach = (p.n_fft + p.hop - 1) / p.hop;
        add((size_t)(frames + reach * notes) + 64);
        add((size_t)frames + 64);
        add((size_t)(samples / 4 + notes + 16) + 64);
    }
    return b + 4096;
}

static int ensure_scratch(goofer_ctx *ctx, size_t bytes)
{
    if (ctx->scratch_bytes >= bytes) return GOOFER_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->scratch) HIP_TRY(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->scratch, bytes);
    if (e != hipSuccess) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->scratch_bytes = bytes;
    return GOOFER_OK;
}

static int ensure_small(goofer_ctx *ctx, size_t bytes)
{
    if (ctx->small_bytes >= bytes) return GOOFER_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ctx->small) HIP_TRY(ctx, hipFree(ctx->small));
    ctx->small = nullptr;
    ctx->small_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->small, bytes));
    ctx->small_bytes = bytes;
    return GOOFER_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side table construction (fp64 then rounded exactly where numpy rounds)
static void gauss_taps_host(double sigma, std::vector<double> &taps, int &radius)
{
    radius = (int)(4.0 * sigma + 0.5);          // GOOFER.py:247
    taps.assign(2 * radius + 1, 0.0);
    double sum = 0.0;
    for (int t = -radius; t <= radius; ++t) {
        double q = (double)t / sigma;
        taps[t + radius] = exp(-0.5 * (q * q));
        sum += taps[t + radius];
    }
    for (auto &v : taps) v /= sum;
}

static void ramp_gain_host(int n_bins, double sr, double lo, double hi, double db, std::vector<float> &out)
{
    std::vector<double> f(n_bins), g(n_bins, 1.0);
    double step = (sr / 2.0) / (double)(n_bins - 1);
    for (int i = 0; i < n_bins; ++i) f[i] = (double)i * step;
    f[n_bins - 1] = sr / 2.0;
    int a = (int)(std::lower_bound(f.begin(), f.end(), lo) - f.begin());
    int b = (int)(std::lower_bound(f.begin(), f.end(), hi) - f.begin());
    double top = pow(10.0, db / 20.0);
    int m = b - a;
    for (int i = 0; i < m; ++i) {
        double rise = m > 1 ? (i == m - 1 ? 1.0 : (double)i * (1.0 / (double)(m - 1))) : 0.0;
        g[a + i] = 1.0 + rise * (top - 1.0);
    }
    for (int i = b; i < n_bins; ++i) g[i] = top;
    out.resize(n_bins);
    for (int i = 0; i < n_bins; ++i) out[i] = (float)g[i];
}

template <typename T> static int upload(goofer_ctx *ctx, T **dst, const std::vector<T> &src)
{
    if (*dst) HIP_TRY(ctx, hipFree(*dst));
    *dst = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GOOFER_OK;
}

static void free_plan(goofer_plan_t &p)
{
    void *ptrs[] = {p.window, p.window_blur, p.blur_edge, p.win_sq, p.freqs, p.lin_freqs, p.boost, p.bright_h, p.bright_b, p.tw_full, p.tw_half, p.pulse_peak, p.pulse_shape, p.blur5, p.blur175,
                    p.bl_chirp, p.bl_bhat, p.bl_tw, p.bl_twh};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    p = goofer_plan_t();
}

// ---------------------------------------------------------------------------------------------
extern "C" {

const char *goofer_version(void) { return "goofer_hip 0.1 (gfx950)"; }

int goofer_create(int device_id, goofer_ctx **out)
{
    if (!out) return GOOFER_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return GOOFER_EHIP;
    if (hipSetDevice(device_id) != hipSuccess) return GOOFER_EHIP;
    goofer_ctx *c = new goofer_ctx();
    c->device = device_id;
    // the onset-overflow word lives as long as the handle (never inside the re-carved, re-allocated scratch arena)
    if (hipMalloc((void **)&c->ovf_flag, 64) != hipSuccess || hipMemset(c->ovf_flag, 0, 64) != hipSuccess) {
        delete c;
        return GOOFER_EHIP;
    }
    *out = c;
    return GOOFER_OK;
}

void goofer_destroy(goofer_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_plan(ctx->plan);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->small) (void)hipFree(ctx->small);
    if (ctx->asm_scratch) (void)hipFree(ctx->asm_scratch);
    if (ctx->mask_taps) (void)hipFree(ctx->mask_taps);
    if (ctx->warp_rows) (void)hipFree(ctx->warp_rows);
    if (ctx->ovf_flag) (void)hipFree(ctx->ovf_flag);
    for (int i = 0; i < ctx->prof_cap * (PROF_STAGES + 1); ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
    free(ctx->prof_ev);
    for (int i = 0; ctx->prof_asm && i < ctx->prof_cap * 6; ++i) (void)hipEventDestroy(ctx->prof_asm[i]);
    free(ctx->prof_asm);
    for (int i = 0; ctx->prof_side && i < ctx->prof_cap * 4; ++i) (void)hipEventDestroy(ctx->prof_side[i]);
    free(ctx->prof_side);
    for (int i = 0; ctx->prof_main2 && i < ctx->prof_cap * 2; ++i) (void)hipEventDestroy(ctx->prof_main2[i]);
    free(ctx->prof_main2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_maps) (void)hipEventDestroy(ctx->ev_maps);
    if (ctx->ev_entry) (void)hipEventDestroy(ctx->ev_entry);
    if (ctx->ev_f0) (void)hipEventDestroy(ctx->ev_f0);
    if (ctx->ev_f0s) (void)hipEventDestroy(ctx->ev_f0s);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    delete ctx;
}

const char *goofer_last_error(const goofer_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int goofer_plan(goofer_ctx *ctx, int sr, int n_fft, int hop)
{
    if (!ctx) return GOOFER_EINVAL;
    const bool native = n_fft == 512 || n_fft == 768 || n_fft == 1024 || n_fft == 1536 || n_fft == 2048;
    if (!native && (n_fft < 64 || n_fft > 2048 || (n_fft & 1)))
        return goofer_fail(ctx, GOOFER_EINVAL, "n_fft must be an even number in [64, 2048] (got %d)", n_fft);
    if (hop <= 0 || hop > n_fft || sr <= 0) return goofer_fail(ctx, GOOFER_EINVAL, "bad sr/hop (%d, %d)", sr, hop);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    free_plan(ctx->plan);
    goofer_plan_t &p = ctx->plan;
    const int B = n_fft / 2 + 1, M = n_fft / 2;
    const double PI = 3.14159265358979323846;

    std::vector<float> win(n_fft), wsq(n_fft), freqs(B), boost(B), bh, bb;
    for (int i = 0; i < n_fft; ++i) {
        // np.hanning: 0.5 + 0.5 cos(pi n/(M-1)), n = 1-M, 3-M, ...   -> fp32 -> sqrt (numpy's ** 0.5)
        double n = (double)(1 - n_fft + 2 * i);
        float h = (float)(0.5 + 0.5 * cos(PI * n / (double)(n_fft - 1)));
        win[i] = sqrtf(h);
        wsq[i] = win[i] * win[i];
    }
    {
        double val = 1.0 / ((double)n_fft * (1.0 / (double)sr));   // np.fft.rfftfreq
        for (int k = 0; k < B; ++k) freqs[k] = (float)((double)k * val);
        double step = 99.0 / (double)(B - 1);                      // np.linspace(1, 100, B)
        for (int k = 0; k < B; ++k) boost[k] = (float)((double)k * step + 1.0);
        boost[B - 1] = 100.0f;
    }
    ramp_gain_host(B, (double)sr, 2000.0, 3500.0, 3.0, bh);
    ramp_gain_host(B, (double)sr, 3500.0, 5000.0, 20.0, bb);
    std::vector<float2> twf(M), twh(M / 2 + 1);
    for (int k = 0; k < M; ++k) twf[k] = make_float2((float)cos(-2.0 * PI * k / M), (float)sin(-2.0 * PI * k / M));
    for (int k = 0; k <= M / 2; ++k) twh[k] = make_float2((float)cos(-PI * k / M), (float)sin(-PI * k / M));
    std::vector<double> t5, t175;
    int r5, r175;
    gauss_taps_host(0.5, t5, r5);
    gauss_taps_host(1.75, t175, r175);

    // Time-domain image of the sigma-0.5 bin blur (GOOFER.py:1143, 1171).  A circular convolution of a frame's spectrum with
    // the symmetric taps t is a multiplication of its samples by W[n] = t2 + 2 t1 cos(2 pi n / N) + 2 t0 cos(4 pi n / N); the
    // stem walkers fold W into the synthesis window of the frames that get the blur (stems.hip).
    std::vector<float> winb(n_fft);
    for (int i = 0; i < n_fft; ++i) {
        const double a = 2.0 * PI * (double)i / (double)n_fft;
        winb[i] = (float)((double)win[i] * (t5[2] + 2.0 * t5[1] * cos(a) + 2.0 * t5[0] * cos(2.0 * a)));
    }
    // ... circularly, i.e. over the spectrum's own Hermitian continuation past DC and Nyquist, where the reference's
    // gaussian_filter1d reflects the array.  The two differ by a purely imaginary E on four bins (E_1 = i (2 t0 Im X_1 + t1 Im X_0),
    // E_2 = i t0 Im X_0, and the mirror image below Nyquist); adding D with blur(D) = E to the spectrum in front of the transform
    // makes the product form the reflected blur.  On imaginary, odd-continued sequences the blur is the matrix T below; its
    // inverse decays like 0.135^k, six bins carry it to 6e-6 of E.  c1 / c2 = the first two columns of T^-1, stored per lane of
    // the walkers' bin layout (bin k = lane + 64 i): rows 0, 1 for bins 1..6 (slot 0), rows 2, 3 for bins M-6..M-1 (last slot).
    std::vector<float> edge(4 * 64, 0.f);
    {
        const int K = 6;
        double T[6][12] = {{0}};
        for (int k = 1; k <= K; ++k) {
            const int dd[5] = {-2, -1, 0, 1, 2};
            for (int q = 0; q < 5; ++q) {
                int j = k + dd[q];
                double sg = 1.0;
                if (j == 0) continue;                          // Im D_0 = 0
                if (j < 0) { j = -j; sg = -1.0; }              // odd continuation
                if (j <= K) T[k - 1][j - 1] += sg * t5[q];
            }
            T[k - 1][K + k - 1] = 1.0;                          // [T | I] -> Gauss-Jordan
        }
        for (int c = 0; c < K; ++c) {
            int piv = c;
            for (int r = c + 1; r < K; ++r)
                if (fabs(T[r][c]) > fabs(T[piv][c])) piv = r;
            for (int j = 0; j < 2 * K; ++j) std::swap(T[c][j], T[piv][j]);
            const double d = T[c][c];
            for (int j = 0; j < 2 * K; ++j) T[c][j] /= d;
            for (int r = 0; r < K; ++r) {
                if (r == c) continue;
                const double f = T[r][c];
                for (int j = 0; j < 2 * K; ++j) T[r][j] -= f * T[c][j];
            }
        }
        for (int j = 1; j <= K; ++j) {
            edge[0 * 64 + j] = (float)T[j - 1][K + 0];           // c1_j at lane j (bin j)
            edge[1 * 64 + j] = (float)T[j - 1][K + 1];           // c2_j
            edge[2 * 64 + (64 - j)] = (float)T[j - 1][K + 0];    // bin M - j sits in lane 64 - j of the last slot
            edge[3 * 64 + (64 - j)] = (float)T[j - 1][K + 1];
        }
    }
    int rc;
    if (!native) {
        // Bluestein: X[k] = conj(c_k) sum_n (x_n conj(c_n)) c_{k-n}, c_n = exp(i pi n^2 / M): a circular convolution of length
        // L >= 2 M - 1 with the wrapped chirp, whose transform is made here in fp64 (n^2 mod 2 M keeps the phases exact)
        int L = 256;
        while (L < 2 * M - 1) L *= 2;
        std::vector<float2> chirp(M), bhat(L), twl(L), twhf(M + 1);
        std::vector<double> br(L, 0.0), bi(L, 0.0);
        for (int k = 0; k < M; ++k) {
            const long long q = ((long long)k * k) % (2LL * M);
            const double ang = PI * (double)q / (double)M;
            chirp[k] = make_float2((float)cos(ang), (float)sin(ang));
            br[k] = cos(ang); bi[k] = sin(ang);
            if (k) { br[L - k] = cos(ang); bi[L - k] = sin(ang); }
        }
        std::vector<double> cr(L), ci(L);
        for (int k = 0; k < L; ++k) { cr[k] = cos(-2.0 * PI * k / L); ci[k] = sin(-2.0 * PI * k / L); twl[k] = make_float2((float)cr[k], (float)ci[k]); }
        for (int k = 0; k < L; ++k) {
            double sr_ = 0.0, si_ = 0.0;
            for (int j = 0; j < L; ++j) {
                if (br[j] == 0.0 && bi[j] == 0.0) continue;
                const int t = (int)(((long long)k * j) & (L - 1));
                sr_ += br[j] * cr[t] - bi[j] * ci[t];
                si_ += br[j] * ci[t] + bi[j] * cr[t];
            }
            bhat[k] = make_float2((float)sr_, (float)si_);
        }
        for (int k = 0; k <= M; ++k) twhf[k] = make_float2((float)cos(-PI * k / M), (float)sin(-PI * k / M));
        if ((rc = upload(ctx, &p.bl_chirp, chirp))) return rc;
        if ((rc = upload(ctx, &p.bl_bhat, bhat))) return rc;
        if ((rc = upload(ctx, &p.bl_tw, twl))) return rc;
        if ((rc = upload(ctx, &p.bl_twh, twhf))) return rc;
        p.bl_L = L;
    }
    if ((rc = upload(ctx, &p.blur_edge, edge))) return rc;
    if ((rc = upload(ctx, &p.window_blur, winb))) return rc;
    if ((rc = upload(ctx, &p.window, win))) return rc;
    if ((rc = upload(ctx, &p.win_sq, wsq))) return rc;
    if ((rc = upload(ctx, &p.freqs, freqs))) return rc;
    {
        std::vector<float> lin(B);
        const double fstep = ((double)sr / 2.0) / (double)(B - 1);
        for (int k = 0; k < B; ++k) lin[k] = (float)(k >= B - 1 ? (double)sr / 2.0 : (double)k * fstep);
        if ((rc = upload(ctx, &p.lin_freqs, lin))) return rc;
    }
    if ((rc = upload(ctx, &p.boost, boost))) return rc;
    if ((rc = upload(ctx, &p.bright_h, bh))) return rc;
    if ((rc = upload(ctx, &p.bright_b, bb))) return rc;
    if ((rc = upload(ctx, &p.tw_full, twf))) return rc;
    if ((rc = upload(ctx, &p.tw_half, twh))) return rc;
    if ((rc = upload(ctx, &p.blur5, t5))) return rc;
    if ((rc = upload(ctx, &p.blur175, t175))) return rc;
    for (int j = 0; j < 5; ++j) p.taps5_f[j] = (float)t5[j];
    for (int j = 0; j < 15; ++j) p.taps175_f[j] = (float)t175[j];
    HIP_TRY(ctx, hipMalloc((void **)&p.pulse_peak, PULSE_PEAK_FLOATS * sizeof(float)));
    p.sr = sr; p.n_fft = n_fft; p.hop = hop; p.n_bins = B;
    if ((rc = launch_pulse_peak(ctx, p.pulse_peak, (double)sr, 0))) return rc;
    HIP_TRY(ctx, hipMalloc((void **)&p.pulse_shape, pulse_shape_table_floats() * sizeof(float)));
    if ((rc = launch_pulse_shape_table(ctx, p.pulse_shape, p.pulse_peak, (double)sr, 0))) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GOOFER_OK;
}

int goofer_pulse_model(goofer_ctx *ctx, double Ra, double Rg, double Rk)
{
    if (!ctx) return GOOFER_EINVAL;
    goofer_plan_t &p = ctx->plan;
    if (!p.pulse_peak || !p.pulse_shape) return goofer_fail(ctx, GOOFER_EINVAL, "goofer_pulse_model needs a plan (goofer_plan)");
    if (!std::isfinite(Ra) || !std::isfinite(Rg) || !std::isfinite(Rk))
        return goofer_fail(ctx, GOOFER_EINVAL, "Ra, Rg, Rk must be finite (got %g, %g, %g)", Ra, Rg, Rk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());                     // nothing in flight still reads the tables
    p.lf.ra = Ra; p.lf.rg = Rg; p.lf.rk = Rk;
    int rc;
    if ((rc = launch_pulse_peak(ctx, p.pulse_peak, (double)p.sr, 0))) return rc;
    if ((rc = launch_pulse_shape_table(ctx, p.pulse_shape, p.pulse_peak, (double)p.sr, 0))) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GOOFER_OK;
}

int goofer_reserve(goofer_ctx *ctx, int64_t max_frames, int64_t max_samples, int64_t max_notes)
{
    if (!ctx) return GOOFER_EINVAL;
    if (!ctx->plan.n_fft) return goofer_fail(ctx, GOOFER_ENOPLAN, "goofer_plan first");
    const bool walkers = ctx->stems && ctx->ola_fused && stems_supported(ctx->plan);   // else: room for the spectra of the one-kernel-per-step path
    return ensure_scratch(ctx, scratch_need(ctx->plan, max_frames, max_samples, max_notes, walkers ? 0 : 1, !walkers));
}

// copy one plan table to host memory (tests / debugging); which: 0 window 1 freqs 2 boost 3 bright_h
// 4 bright_b 5 pulse_peak; returns the element count or a negative error
int goofer_debug_table(goofer_ctx *ctx, int which, float *host_out, int capacity)
{
    if (!ctx || !ctx->plan.n_fft) return GOOFER_ENOPLAN;
    const goofer_plan_t &p = ctx->plan;
    const float *src[] = {p.window, p.freqs, p.boost, p.bright_h, p.bright_b, p.pulse_peak};
    const int cnt[] = {p.n_fft, p.n_bins, p.n_bins, p.n_bins, p.n_bins, 8193};
    if (which < 0 || which > 5) return GOOFER_EINVAL;
    int n = std::min(cnt[which], capacity);
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(host_out, src[which], n * sizeof(float), hipMemcpyDeviceToHost));
    return cnt[which];
}

// copy intermediate `which` of the last goofer_synth_batch to host memory (tests / debugging):
// 0 frame_note 1 row_src 2 f0_scaled 3 pulse 4 S_harm 5 S_uv 6 S_breath 7 frames(last stem) 8 env_harm
// 9 env_noise 10 mask_short 11 note_mag 12 note_peak 13 onset_cnt 14 onset_idx (13, 14: also of the last goofer_pulse_train)
// 15 frame_skip (the spectra-in-HBM pipeline with per-frame skipping: one byte per frame).
// Returns the byte size.
/* Host helper of the note planner (goofer_amd/sampler.py, SillySampler.py:264-283): Gaussian FIR along the rows of a small
 * fp64 matrix with numpy 'reflect' padding, accumulated tap by tap in ascending order (product rounded, then added: the
 * arithmetic of the planner's numpy loop, so the tracks are the same bits either way).  Pure CPU code: no device is touched. */
int goofer_host_gauss_rows(const double *x, int64_t rows, int T, const double *taps, int radius, double *out)
{
    if (!x || !taps || !out || rows < 0 || T <= 0 || radius < 0) return GOOFER_EINVAL;
    const int nt = 2 * radius + 1;
    std::vector<int> idx((size_t)T + 2 * radius);
    const int period = T > 1 ? 2 * (T - 1) : 1;
    for (int q = -radius; q < T + radius; ++q) {              // numpy 'reflect' as a periodic map (T == 1: 'edge')
        int m = T > 1 ? ((q % period) + period) % period : 0;
        idx[q + radius] = m < T ? m : period - m;
    }
    std::vector<double> pad((size_t)T + 2 * radius);
    for (int64_t r = 0; r < rows; ++r) {
        const double *xr = x + r * T;
        double *o = out + r * T;
        for (int q = 0; q < T + 2 * radius; ++q) pad[q] = xr[idx[q]];
        const double *pp = pad.data();
        for (int t = 0; t < T; ++t) o[t] = taps[0] * pp[t];
        for (int j = 1; j < nt; ++j) {                       // tap-outer: every o[t] still sums its taps in ascending order
            const double kj = taps[j];
            for (int t = 0; t < T; ++t) {
                const double prod = kj * pp[t + j];
                o[t] = o[t] + prod;
            }
        }
    }
    return GOOFER_OK;
}

/* Synchronise the device and report what the asynchronous batch calls could not: a note whose pulse onsets did not fit its
 * onset slots (n / 2 + 16 per note — more than one pulse per two samples; the onsets beyond were dropped).  The flag is a
 * handle-owned word every pulse-chain launch (goofer_pulse_train, goofer_synth_batch / goofer_render_batch incl. the extra
 * synthesis calls of the post chain) raises with atomicMax; it stays up until this call reads and clears it. */
int goofer_check(goofer_ctx *ctx)
{
    if (!ctx) return GOOFER_EINVAL;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (!ctx->ovf_flag) return GOOFER_OK;
    int32_t v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->ovf_flag, sizeof(v), hipMemcpyDeviceToHost));
    if (v != 0) {
        HIP_TRY(ctx, hipMemset(ctx->ovf_flag, 0, sizeof(v)));        // reported once
        return goofer_fail(ctx, GOOFER_EINVAL, "note %d of a batch since the last check has more pulse onsets than n / 2 + 16 (f0 above "
                           "sr / 2?): the pulses beyond its onset slots were dropped", v - 1);
    }
    return GOOFER_OK;
}

/* Cumulative counters of the handle (device words beside the overflow flag; the call synchronises the device):
 *   "pulse_scanned_notes"   notes whose onsets went through the parallel phase scan (k_pulse_onsets_par)
 *   "pulse_fallback_notes"  ... of which were walked sequentially afterwards (a sample within the error band of an integer
 *                           phase, a negative / non-finite increment, or option pulse_scan = 2) */
int goofer_counter(goofer_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value) return GOOFER_EINVAL;
    int which = !strcmp(name, "pulse_fallback_notes") ? 1 : (!strcmp(name, "pulse_scanned_notes") ? 2 : -1);
    if (which < 0) return goofer_fail(ctx, GOOFER_EINVAL, "unknown counter %s", name);
    HIP_TRY(ctx, hipDeviceSynchronize());
    int32_t v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->ovf_flag + which, sizeof(v), hipMemcpyDeviceToHost));
    *value = (int64_t)(uint32_t)v;
    return GOOFER_OK;
}

int64_t goofer_debug_fetch(goofer_ctx *ctx, int which, void *host_out, int64_t capacity_bytes)
{
    if (!ctx || which < 0 || which >= 16 || !ctx->dbg_ptr[which]) return GOOFER_EINVAL;
    if (int rc = goofer_check(ctx)) return rc;
    size_t nb = ctx->dbg_bytes[which] < (size_t)capacity_bytes ? ctx->dbg_bytes[which] : (size_t)capacity_bytes;
    if (hipMemcpy(host_out, ctx->dbg_ptr[which], nb, hipMemcpyDeviceToHost) != hipSuccess) return GOOFER_EHIP;
    return (int64_t)ctx->dbg_bytes[which];
}

static const char *const PROF_NAMES[PROF_STAGES] = {
    "setup_maps", "", "", "phase_inc", "pulse_onsets", "pulse_place", "rfft_frames", "harm_shape",
    "irfft_harm", "noise_spectra", "irfft_breath", "irfft_unvoiced", "mask_short", "ola3_gains", "apply_gain", "env_edit", "env_rows", "sample_assemble"};

// Per-stage timing of goofer_synth_batch with HIP events recorded on the caller's stream (so the
// numbers are what that stream really executed).  begin(max_steps) arms it; every synth batch then
// records PROF_STAGES+1 events; end() synchronises and returns the summed milliseconds per stage.
int goofer_profile_begin(goofer_ctx *ctx, int max_steps)
{
    if (!ctx || max_steps <= 0) return GOOFER_EINVAL;
    if (ctx->prof_cap < max_steps) {
        for (int i = 0; i < ctx->prof_cap * (PROF_STAGES + 1); ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
        free(ctx->prof_ev);
        for (int i = 0; ctx->prof_side && i < ctx->prof_cap * 4; ++i) (void)hipEventDestroy(ctx->prof_side[i]);
        free(ctx->prof_side);
        for (int i = 0; ctx->prof_main2 && i < ctx->prof_cap * 2; ++i) (void)hipEventDestroy(ctx->prof_main2[i]);
        free(ctx->prof_main2);
        for (int i = 0; ctx->prof_asm && i < ctx->prof_cap * 6; ++i) (void)hipEventDestroy(ctx->prof_asm[i]);
        free(ctx->prof_asm);
        ctx->prof_asm = (hipEvent_t *)calloc((size_t)max_steps * 6, sizeof(hipEvent_t));
        if (!ctx->prof_asm) return goofer_fail(ctx, GOOFER_ENOMEM, "event pool");
        for (int i = 0; i < max_steps * 6; ++i) HIP_TRY(ctx, hipEventCreate(&ctx->prof_asm[i]));
        ctx->prof_ev = (hipEvent_t *)calloc((size_t)max_steps * (PROF_STAGES + 1), sizeof(hipEvent_t));
        ctx->prof_side = (hipEvent_t *)calloc((size_t)max_steps * 4, sizeof(hipEvent_t));
        ctx->prof_main2 = (hipEvent_t *)calloc((size_t)max_steps * 2, sizeof(hipEvent_t));
        if (!ctx->prof_ev || !ctx->prof_side || !ctx->prof_main2) return goofer_fail(ctx, GOOFER_ENOMEM, "event pool");
        for (int i = 0; i < max_steps * (PROF_STAGES + 1); ++i) HIP_TRY(ctx, hipEventCreate(&ctx->prof_ev[i]));
        for (int i = 0; i < max_steps * 4; ++i) HIP_TRY(ctx, hipEventCreate(&ctx->prof_side[i]));
        for (int i = 0; i < max_steps * 2; ++i) HIP_TRY(ctx, hipEventCreate(&ctx->prof_main2[i]));
        ctx->prof_cap = max_steps;
    }
    ctx->prof_steps = 0;
    ctx->prof_asm_steps = 0;
    ctx->prof_on = true;
    return GOOFER_OK;
}

int goofer_profile_end(goofer_ctx *ctx, double *ms_per_stage, int n_stages)
{
    if (!ctx) return GOOFER_EINVAL;
    ctx->prof_on = false;
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (int s = 0; s < n_stages && s < PROF_STAGES; ++s) {
        double acc = 0.0;
        if (ctx->prof_only >= 0 && s != ctx->prof_only) {                // (its events were not recorded)
            ms_per_stage[s] = 0.0;
            continue;
        }
        if (s >= PROF_ASM0) {
            // the assembly's three large kernels (goofer_assemble_batch / goofer_render_batch), each bracketed on the stream it ran on
            for (int k = 0; k < ctx->prof_asm_steps; ++k) {
                hipEvent_t *q = ctx->prof_asm + (size_t)k * 6 + 2 * (s - PROF_ASM0);
                float ms = 0.f;
                if (ctx->prof_asm_mask[k] & (1u << (s - PROF_ASM0))) HIP_TRY(ctx, hipEventElapsedTime(&ms, q[0], q[1]));
                acc += ms;
            }
            ms_per_stage[s] = acc;
            continue;
        }
        for (int k = 0; k < ctx->prof_steps; ++k) {
            hipEvent_t *e = ctx->prof_ev + (size_t)k * (PROF_STAGES + 1);
            float ms = 0.f;
            if (ctx->prof_side_used && s >= 3 && s <= 5) {              // the pulse chain ran on the side stream
                hipEvent_t *q = ctx->prof_side + (size_t)k * 4;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, q[s - 3], q[s - 2]));
            } else if (ctx->prof_side_used && (s == (ctx->prof_stems ? 6 : 9) || s == (ctx->prof_stems ? 7 : 12))) {
                // the two kernels launched beside it on the caller's stream, in launch order
                hipEvent_t *q = ctx->prof_main2 + (size_t)k * 2;
                const bool first = s == (ctx->prof_stems ? 6 : 9);
                HIP_TRY(ctx, hipEventElapsedTime(&ms, first ? e[5] : q[0], first ? q[0] : q[1]));
            } else {
                HIP_TRY(ctx, hipEventElapsedTime(&ms, e[s], e[s + 1]));
            }
            acc += ms;
        }
        ms_per_stage[s] = acc;
    }
    return ctx->prof_steps;
}

const char *goofer_profile_stage_name(int stage) { return stage >= 0 && stage < PROF_STAGES ? PROF_NAMES[stage] : ""; }

static const char *const PROF_NAMES_STEMS[PROF_STAGES] = {
    "setup_maps", "", "", "phase_inc", "pulse_onsets", "pulse_place", "mask_short", "noise_stems",
    "", "harm_stem", "", "", "", "note_finish", "", "env_edit", "env_rows", "sample_assemble"};

static const char *const PROF_NAMES_OLA[PROF_STAGES] = {
    "setup_maps", "", "", "phase_inc", "pulse_onsets", "pulse_place", "rfft_frames", "harm_shape",
    "", "noise_spectra", "", "", "mask_short", "irfft_ola3", "apply_gain", "env_edit", "env_rows", "sample_assemble"};

const char *goofer_profile_stage_name_ex(const goofer_ctx *ctx, int stage)
{
    if (stage < 0 || stage >= PROF_STAGES) return "";
    if (ctx && ctx->prof_stems) return PROF_NAMES_STEMS[stage];
    return (ctx && ctx->ola_fused) ? PROF_NAMES_OLA[stage] : PROF_NAMES[stage];
}

/* options: see include/goofer_hip.h */
int goofer_set_option(goofer_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return GOOFER_EINVAL;
    if (!strcmp(name, "fused_ola")) { ctx->ola_fused = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "overlap")) { ctx->overlap = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "stems")) { ctx->stems = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "prof_only")) { ctx->prof_only = value < 0 || value >= PROF_STAGES ? -1 : value; return GOOFER_OK; }   // stage index, -1: all
    if (!strcmp(name, "skip_zero")) { ctx->skip_zero = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "td_blur")) { ctx->td_blur = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "pulse_scan")) { ctx->pulse_scan = value < 0 ? 0 : (value > 2 ? 2 : value); return GOOFER_OK; }
    if (!strcmp(name, "sa_fast")) { ctx->sa_fast = value != 0; return GOOFER_OK; }
    if (!strcmp(name, "value_f64")) { ctx->value_f64 = value != 0; return GOOFER_OK; }
    return goofer_fail(ctx, GOOFER_EINVAL, "unknown option %s", name);
}

int goofer_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(goofer_note_params);
    case 1: return (int)sizeof(goofer_batch);
    case 2: return (int)sizeof(goofer_note_plan);
    case 3: return (int)sizeof(goofer_assembly);
    case 4: return (int)sizeof(goofer_onepole_job);
    case 5: return (int)sizeof(goofer_post_note);
    case 6: return (int)sizeof(goofer_post);
    case 7: return (int)sizeof(goofer_plan_request);
    case 8: return (int)sizeof(goofer_plan_geometry);
    }
    return -1;
}

#define NEED_PLAN(ctx)                                                                    \
    if (!(ctx)) return GOOFER_EINVAL;                                                     \
    if (!(ctx)->plan.n_fft) return goofer_fail((ctx), GOOFER_ENOPLAN, "goofer_plan first")

int goofer_rfft_frames(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off, int n_notes,
                       int64_t total_frames, float *S, int ldc, void *stream)
{
    NEED_PLAN(ctx);
    if (ldc < ctx->plan.n_bins) return goofer_fail(ctx, GOOFER_EINVAL, "ldc %d < n_bins %d", ldc, ctx->plan.n_bins);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_scratch(ctx, total_frames * sizeof(int) + 4096);
    if (rc) return rc;
    int *frame_note = (int *)ctx->scratch;
    if ((rc = launch_frame_note(ctx, frame_off, n_notes, total_frames, frame_note, st))) return rc;
    return launch_rfft_frames_mapped(ctx, x, sample_off, frame_off, frame_note, total_frames, (float2 *)S, ldc, st);
}

int goofer_irfft_ola(goofer_ctx *ctx, const float *S, int ldc, const int64_t *sample_off, const int64_t *frame_off, int n_notes,
                     int64_t total_frames, int64_t total_samples, float *y, void *stream)
{
    NEED_PLAN(ctx);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_scratch(ctx, (size_t)total_frames * ctx->plan.n_fft * sizeof(float) + 4096);
    if (rc) return rc;
    float *frames = (float *)ctx->scratch;
    if ((rc = launch_irfft_frames(ctx, (const float2 *)S, ldc, total_frames, frames, st))) return rc;
    return launch_ola_gather(ctx, frames, sample_off, frame_off, n_notes, total_samples, y, nullptr, st);
}

int goofer_pulse_train(goofer_ctx *ctx, const float *f0, const int64_t *sample_off, int n_notes, int64_t total_samples,
                       float *pulse, void *stream)
{
    NEED_PLAN(ctx);
    hipStream_t st = (hipStream_t)stream;
    size_t need = total_samples * sizeof(double) + (total_samples / 2 + 16 * (size_t)n_notes + 16) * (ONSET_BYTES + 4) +
                  n_notes * sizeof(int32_t) + 8192;
    int rc = ensure_scratch(ctx, need);
    if (rc) return rc;
    arena a{(char *)ctx->scratch, ctx->scratch_bytes, 0};
    double *inc = a.take<double>(total_samples + 16);          // (also holds the placement's tile table: 16 bytes per 256 * PP_SPT samples)
    char *onsets = a.take<char>((total_samples / 2 + 16 * (size_t)n_notes + 16) * ONSET_BYTES);
    int32_t *oidx = a.take<int32_t>(total_samples / 2 + 16 * (size_t)n_notes + 16);
    int32_t *cnt = a.take<int32_t>(n_notes + 16);
    int32_t *ovf = ctx->ovf_flag;
    if (!inc || !onsets || !oidx || !cnt) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    for (int i = 0; i < 16; ++i) ctx->dbg_ptr[i] = nullptr;
    ctx->dbg_ptr[13] = cnt; ctx->dbg_bytes[13] = n_notes * sizeof(int32_t);
    ctx->dbg_ptr[14] = oidx; ctx->dbg_bytes[14] = (total_samples / 2 + 16 * (size_t)n_notes) * sizeof(int32_t);
    return launch_pulse_train(ctx, f0, 1.0f, sample_off, n_notes, total_samples, pulse, inc, (onset_t *)onsets, oidx, cnt, ovf, st);
}

int goofer_gauss_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld, const double *taps,
                      int radius, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (radius < 0 || radius > 4096 || n_bins <= 0 || ld < n_bins) return goofer_fail(ctx, GOOFER_EINVAL, "bad gauss geometry");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_small(ctx, std::max<size_t>(65536, (size_t)(2 * radius + 1) * sizeof(double) + 64));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->small, taps, (2 * radius + 1) * sizeof(double), hipMemcpyHostToDevice, st));
    return launch_gauss_bins(ctx, in, out, rows, n_bins, ld, (const double *)ctx->small, radius, nullptr, st);
}

int goofer_warp_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld, const double *formants,
                     const double *f_shift, double ratio, void *stream)
{
    NEED_PLAN(ctx);
    hipStream_t st = (hipStream_t)stream;
    const double *d_shift = nullptr;
    if (f_shift) {
        int rc = ensure_small(ctx, 65536);
        if (rc) return rc;
        d_shift = (const double *)((char *)ctx->small + 32768);
        HIP_TRY(ctx, hipMemcpyAsync((void *)d_shift, f_shift, 4 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    return launch_warp_bins(ctx, in, out, rows, n_bins, ld, formants, d_shift, nullptr, nullptr, nullptr, ratio, st);
}

int goofer_knot_decode(goofer_ctx *ctx, const uint16_t *knots_f16, int K, const float *hz_knots, int64_t rows, float *env,
                       int n_bins, int ld, void *stream)
{
    NEED_PLAN(ctx);
    if (K < 2 || K > 4096) return goofer_fail(ctx, GOOFER_EINVAL, "bad knot count %d", K);
    hipStream_t st = (hipStream_t)stream;
    const goofer_plan_t &p = ctx->plan;
    // 2-tap lerp plan, fp32 arithmetic like precompute_interp_matrix (GOOFER.py:84-90)
    std::vector<int> idx(n_bins);
    std::vector<float> w0(n_bins), w1(n_bins);
    double val = 1.0 / ((double)p.n_fft * (1.0 / (double)p.sr));
    for (int b = 0; b < n_bins; ++b) {
        float f = (float)((double)b * val);
        int i = (int)(std::upper_bound(hz_knots, hz_knots + K, f) - hz_knots) - 1;   // searchsorted(side='right') - 1
        i = std::min(std::max(i, 0), K - 2);
        float x0 = hz_knots[i], x1 = hz_knots[i + 1];
        float den = std::max(x1 - x0, 1e-12f);
        float b1 = (f - x0) / den;
        idx[b] = i; w1[b] = b1; w0[b] = 1.0f - b1;
    }
    int rc = ensure_small(ctx, 65536);
    if (rc) return rc;
    if ((size_t)n_bins * 12 > 32768) return goofer_fail(ctx, GOOFER_EINVAL, "n_bins too large");
    char *d = (char *)ctx->small;
    int *d_idx = (int *)d;
    float *d_w0 = (float *)(d + 4 * (size_t)n_bins), *d_w1 = (float *)(d + 8 * (size_t)n_bins);
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, idx.data(), n_bins * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_w0, w0.data(), n_bins * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_w1, w1.data(), n_bins * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));   // the host vectors die at return
    return launch_knot_decode(ctx, knots_f16, K, rows, d_idx, d_w0, d_w1, env, n_bins, ld, st);
}

static void knot_lerp_plan(const goofer_plan_t &p, const float *hz_knots, int K, int n_bins, std::vector<int> &idx,
                           std::vector<float> &w0, std::vector<float> &w1)
{
    // 2-tap lerp plan, fp32 arithmetic like precompute_interp_matrix (GOOFER.py:84-90)
    idx.resize(n_bins); w0.resize(n_bins); w1.resize(n_bins);
    double val = 1.0 / ((double)p.n_fft * (1.0 / (double)p.sr));
    for (int b = 0; b < n_bins; ++b) {
        float f = (float)((double)b * val);
        int i = (int)(std::upper_bound(hz_knots, hz_knots + K, f) - hz_knots) - 1;   // searchsorted(side='right') - 1
        i = std::min(std::max(i, 0), K - 2);
        float x0 = hz_knots[i], x1 = hz_knots[i + 1];
        float den = std::max(x1 - x0, 1e-12f);
        float b1 = (f - x0) / den;
        idx[b] = i; w1[b] = b1; w0[b] = 1.0f - b1;
    }
}

int goofer_mag_rows(goofer_ctx *ctx, const float *S, int ldc, int64_t rows, int n_bins, float *mag, int ld, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (ldc < n_bins || ld < n_bins) return goofer_fail(ctx, GOOFER_EINVAL, "bad strides");
    return launch_mag_rows(ctx, (const float2 *)S, ldc, rows, n_bins, mag, ld, (hipStream_t)stream);
}

int goofer_gauss_bins_f64(goofer_ctx *ctx, const float *in, int ld, double *out, int ld64, int64_t rows, int n_bins,
                          const double *taps, int radius, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (radius < 0 || radius > 4096 || ld < n_bins || ld64 < n_bins) return goofer_fail(ctx, GOOFER_EINVAL, "bad gauss geometry");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_small(ctx, std::max<size_t>(65536, (size_t)(2 * radius + 1) * sizeof(double) + 64));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->small, taps, (2 * radius + 1) * sizeof(double), hipMemcpyHostToDevice, st));
    return launch_gauss_rows64(ctx, in, ld, out, ld64, rows, n_bins, (const double *)ctx->small, radius, st);
}

int goofer_knot_fit_error(goofer_ctx *ctx, const double *env, int ld64, const int64_t *probe_rows, int n_probe, int n_bins,
                          const int32_t *knot_bin, int K, const float *hz_knots, double *max_rel_err, void *stream)
{
    NEED_PLAN(ctx);
    if (K < 2 || K > 4096 || !max_rel_err) return goofer_fail(ctx, GOOFER_EINVAL, "bad knot count %d", K);
    hipStream_t st = (hipStream_t)stream;
    std::vector<int> idx;
    std::vector<float> w0, w1;
    knot_lerp_plan(ctx->plan, hz_knots, K, n_bins, idx, w0, w1);
    int rc = ensure_small(ctx, 65536);
    if (rc) return rc;
    if ((size_t)n_bins * 12 + 64 > 32768) return goofer_fail(ctx, GOOFER_EINVAL, "n_bins too large");
    char *d = (char *)ctx->small;
    int *d_idx = (int *)d;
    float *d_w0 = (float *)(d + 4 * (size_t)n_bins), *d_w1 = (float *)(d + 8 * (size_t)n_bins);
    unsigned long long *d_err = (unsigned long long *)(d + 32768);
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, idx.data(), n_bins * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_w0, w0.data(), n_bins * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_w1, w1.data(), n_bins * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(d_err, 0, sizeof(unsigned long long), st));
    if ((rc = launch_knot_error(ctx, env, ld64, probe_rows, n_probe, n_bins, knot_bin, K, d_idx, d_w0, d_w1, d_err, st))) return rc;
    unsigned long long bits = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bits, d_err, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(max_rel_err, &bits, sizeof(double));
    return GOOFER_OK;
}

int goofer_knot_gather(goofer_ctx *ctx, const double *env, int ld64, int64_t rows, const int32_t *knot_bin, int K,
                       uint16_t *knots_f16, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    return launch_knot_gather(ctx, env, ld64, rows, knot_bin, K, knots_f16, (hipStream_t)stream);
}

/* gf.smooth_mask_ds (GOOFER.py:556-569) for a ragged batch of masks: decimate by 4, Gaussian sigma/4 (fp64), linear
 * upsample on float32 linspace grids.  `fast_interp` selects the interpolant form the stem walkers use. */
int goofer_smooth_mask_ds(goofer_ctx *ctx, const float *mask, const int64_t *sample_off, int n_notes, int64_t total_samples,
                          float sigma, int fast_interp, float *out, void *stream)
{
    NEED_PLAN(ctx);
    if (!mask || !sample_off || !out || n_notes <= 0) return goofer_fail(ctx, GOOFER_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> taps;
    int radius;
    gauss_taps_host(std::max(1.0, (double)sigma / 4.0), taps, radius);                // GOOFER.py:561
    if (radius > 2048) return goofer_fail(ctx, GOOFER_EINVAL, "transition sigma too large");
    const size_t n_short = (size_t)(total_samples / 4 + n_notes + 16);
    const size_t need = (taps.size() + 16 + n_short + 2 * (size_t)n_notes + 16) * sizeof(double);
    int rc = ensure_small(ctx, need);
    if (rc) return rc;
    double *d_taps = (double *)ctx->small, *short_s = d_taps + taps.size() + 16, *steps = short_s + n_short;
    HIP_TRY(ctx, hipMemcpyAsync(d_taps, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                                            // the host vector goes out of scope
    double acc = 0.0;
    for (double tv : taps) acc += tv * 1.0;
    if ((rc = launch_mask_short(ctx, mask, sample_off, n_notes, total_samples, d_taps, radius, acc, short_s, st))) return rc;
    return launch_mask_upsample(ctx, short_s, sample_off, n_notes, total_samples, steps, fast_interp != 0, out, st);
}

int goofer_assemble_batch(goofer_ctx *ctx, const goofer_assembly *asmb, void *stream)
{
    if (!ctx || !asmb) return GOOFER_EINVAL;
    if (asmb->n_notes <= 0) return GOOFER_OK;
    if (asmb->ld < asmb->n_bins || asmb->max_K < 2 || asmb->max_K > 4096) return goofer_fail(ctx, GOOFER_EINVAL, "bad assembly geometry");
    if (asmb->any_fry && ctx->plan.hop <= 0) return goofer_fail(ctx, GOOFER_EINVAL, "the fry envelope warp needs goofer_plan first");
    hipStream_t st = (hipStream_t)stream;
    goofer_assembly a = *asmb;
    size_t map_bytes = ((size_t)(a.total_edit_rows + a.total_out_rows) * sizeof(int) + 511) & ~(size_t)255;
    size_t rows_bytes = a.edit_rows ? 0 : (((size_t)a.total_edit_rows * a.ld * sizeof(float) + 255) & ~(size_t)255);
    size_t rec_bytes = ctx->value_f64 ? 0 : (size_t)a.total_out_rows * env_row_rec_bytes();   // per-row records of k_row_recs / k_env_rows
    size_t need = map_bytes + rows_bytes + rec_bytes + 4096;
    if (ctx->asm_bytes < need) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (ctx->asm_scratch) HIP_TRY(ctx, hipFree(ctx->asm_scratch));
        ctx->asm_scratch = nullptr;
        ctx->asm_bytes = 0;
        hipError_t e = hipMalloc(&ctx->asm_scratch, need);
        if (e != hipSuccess) return goofer_fail(ctx, GOOFER_ENOMEM, "assembly scratch hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        ctx->asm_bytes = need;
    }
    int *map_edit = (int *)ctx->asm_scratch;
    int *map_out = map_edit + a.total_edit_rows;
    if (!a.edit_rows) a.edit_rows = (float *)((char *)ctx->asm_scratch + map_bytes);
    void *recs = (char *)ctx->asm_scratch + map_bytes + rows_bytes;
    return launch_assemble(ctx, &a, map_edit, map_out, recs, st);
}

int goofer_stretch_rows(goofer_ctx *ctx, const float *in, int64_t ld_in, int64_t rows_in, float *out, int64_t ld_out, int64_t rows_out,
                        int n_cols, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (!in || !out) return goofer_fail(ctx, GOOFER_EINVAL, "null pointer");
    if (n_cols == 1 && ld_in == 1 && ld_out == 1) return launch_lerp_1d(ctx, in, rows_in, out, rows_out, (hipStream_t)stream);
    return launch_lerp_axis0(ctx, in, ld_in, rows_in, out, ld_out, rows_out, n_cols, (hipStream_t)stream);
}

int goofer_gauss_rows_f64(goofer_ctx *ctx, const double *in, const int64_t *row_off, int n_rows, int64_t total, const double *taps,
                          int radius, double *out, void *stream)
{
    if (!ctx) return GOOFER_EINVAL;
    if (!in || !out || !row_off || !taps) return goofer_fail(ctx, GOOFER_EINVAL, "null pointer");
    if (radius < 0 || radius > (1 << 22)) return goofer_fail(ctx, GOOFER_EINVAL, "gaussian radius %d outside [0, 2^22]", radius);
    if (n_rows <= 0 || total <= 0) return GOOFER_OK;
    hipStream_t st = (hipStream_t)stream;
    // taps behind the fixed small areas (tables at 0, the three jitter tap slots at 64 KiB)
    const size_t tap_off = 65536 + 3 * JIT_SLOT_BYTES, tap_bytes = (size_t)(2 * radius + 1) * sizeof(double);
    int rc = ensure_small(ctx, tap_off + tap_bytes);
    if (rc) return rc;
    double *d_taps = (double *)((char *)ctx->small + tap_off);
    HIP_TRY(ctx, hipMemcpyAsync(d_taps, taps, tap_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                   // the caller's taps buffer may be transient
    return launch_gauss_samples<double>(ctx, in, row_off, n_rows, total, d_taps, radius, nullptr, out, st);
}

int goofer_vocal_roughness(goofer_ctx *ctx, const float *y, const float *f0, const float *mask, const double *noise_s, int n_k,
                           const double *k_list, const double *h_list, double noise_amp, double hp_fc, const float *alpha_slewed,
                           const int64_t *sample_off, int n_notes, int64_t total_samples, float *out, void *stream)
{
    NEED_PLAN(ctx);
    if (!y || !f0 || !mask || !alpha_slewed || !sample_off || !out || (n_k > 0 && (!noise_s || !k_list || !h_list)))
        return goofer_fail(ctx, GOOFER_EINVAL, "null pointer");
    return launch_vocal_roughness(ctx, y, f0, mask, noise_s, n_k, k_list, h_list, noise_amp, hp_fc, alpha_slewed, sample_off, n_notes,
                                  total_samples, out, (hipStream_t)stream);
}

int goofer_onepole_cascade(goofer_ctx *ctx, const float *src, float *dst, const float *f0, const goofer_onepole_job *jobs, int n_jobs,
                           void *stream)
{
    NEED_PLAN(ctx);
    if (!src || !dst || !f0 || !jobs) return goofer_fail(ctx, GOOFER_EINVAL, "null pointer");
    return launch_onepole(ctx, src, dst, f0, jobs, n_jobs, (hipStream_t)stream);
}

// The post chain of one batch.  Host work: turn the per-note flags into job lists for the cascade kernel and upload
// them with the note table; everything per sample runs on the device, in the reference's order.
int goofer_post_batch(goofer_ctx *ctx, const goofer_post *p, void *stream)
{
    NEED_PLAN(ctx);
    if (!p || !p->notes || !p->sample_off || !p->sample_off_host || !p->harm || !p->uv || !p->bre || !p->mix || !p->f0 || !p->mask ||
        !p->params)
        return goofer_fail(ctx, GOOFER_EINVAL, "null pointer in goofer_post");
    const int n = p->n_notes;
    const int64_t N = p->total_samples;
    if (n <= 0 || N <= 0) return GOOFER_OK;
    hipStream_t st = (hipStream_t)stream;
    const double sr = (double)ctx->plan.sr;

    enum { J_SU, J_SJ, J_FRY_H, J_FRY_B, J_ST_H, J_ST_B, J_ST_HP, J_LISTS };
    std::vector<goofer_onepole_job> jobs[J_LISTS];
    std::vector<unsigned char> on_any(n, 0), on_sd(n, 0), on_pd(n, 0);
    bool any = false, any_layers = false, any_fry = false, any_sd = false, any_st = false, any_pd = false;
    for (int i = 0; i < n; ++i) {
        const goofer_post_note &q = p->notes[i];
        const int64_t off = p->sample_off_host[i];
        const int64_t len = p->sample_off_host[i + 1] - off;
        if (len <= 0) continue;
        if (len > INT32_MAX) return goofer_fail(ctx, GOOFER_EINVAL, "note too long");
        auto job = [&](int64_t so, int64_t d_o, int order, int hp, int mode, double cf) {
            goofer_onepole_job j;
            j.src_off = so; j.dst_off = d_o; j.f0_off = off; j.n = (int32_t)len; j.order = order; j.highpass = hp; j.f0_mode = mode;
            j.cutoff_factor = (float)cf; j.reserved = 0;
            return j;
        };
        bool on = false;
        if (q.su_off >= 0) {
            if (!p->su_harm) return goofer_fail(ctx, GOOFER_EINVAL, "su_off set but su_harm is null");
            jobs[J_SU].push_back(job(q.su_off, q.su_off, 12, 1, 1, 1.0));                  // two chained order-6 calls :1051-1058
            on = any_layers = true;
        }
        if (q.sj_off >= 0) {
            if (!p->sj_harm) return goofer_fail(ctx, GOOFER_EINVAL, "sj_off set but sj_harm is null");
            jobs[J_SJ].push_back(job(q.sj_off, q.sj_off, 12, 1, 1, 1.0));                  // :1078-1080
            on = any_layers = true;
        }
        if (q.fry_a < q.fry_b) {
            jobs[J_FRY_H].push_back(job(off, off, 6, 1, 2, 200.0));                        // :1090-1095
            jobs[J_FRY_B].push_back(job(off, off, 6, 1, 2, 200.0));
            on = any_fry = true;
        }
        if (q.sd_strength > 0.f) { on_sd[i] = 1; on = any_sd = true; }
        if (q.tension != 0.f) {
            const double t = fabs((double)q.tension);
            if (q.tension < 0.f) {
                long o = lrint(1.0 + t * 4.0);                                              // np.round: half to even   :1120-1121
                o = o < 1 ? 1 : (o > 6 ? 6 : o);
                jobs[J_ST_H].push_back(job(off, off, (int)o, 0, 0, 2.0 - t * 0.75));
                jobs[J_ST_B].push_back(job(off, off, 4, 1, 0, t));
            } else {
                jobs[J_ST_HP].push_back(job(off, off, 4, 1, 0, t * 4.0));                  // :1129
                jobs[J_ST_B].push_back(job(off, off, 6, 0, 0, (2.0 - t) / 0.5));           // :1133-1134
            }
            on = any_st = true;
        }
        if (q.sa_off >= 0) {
            if (!p->sa_uv || !p->sa_bre) return goofer_fail(ctx, GOOFER_EINVAL, "sa_off set but sa_uv / sa_bre is null");
            on = true;
        }
        if (q.pitch_dyn != 0.f) {
            if (!p->bend) return goofer_fail(ctx, GOOFER_EINVAL, "pitch_dyn set but bend is null");
            on_pd[i] = 1;
            on = any_pd = true;
        }
        on_any[i] = on;
        any |= on;
    }
    if (!any) return GOOFER_OK;

    // staging blob: note table | job lists | flags | taps
    std::vector<double> taps20, taps_pd;
    int r20 = 0, r_pd = 0;
    if (any_sd) gauss_taps_host(20.0, taps20, r20);                                         // :1109
    if (any_pd) gauss_taps_host((double)std::max(1, (int)(0.010 * sr)), taps_pd, r_pd);     // :865-866, 880
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t o_notes = 0, o_jobs[J_LISTS], o_any, o_sd, o_pd, o_t20, o_tpd, blob = al((size_t)n * sizeof(goofer_post_note));
    for (int k = 0; k < J_LISTS; ++k) { o_jobs[k] = blob; blob += al(jobs[k].size() * sizeof(goofer_onepole_job)); }
    o_any = blob; blob += al(n);
    o_sd = blob; blob += al(n);
    o_pd = blob; blob += al(n);
    o_t20 = blob; blob += al(taps20.size() * sizeof(double));
    o_tpd = blob; blob += al(taps_pd.size() * sizeof(double));
    std::vector<unsigned char> host(blob, 0);
    memcpy(host.data() + o_notes, p->notes, (size_t)n * sizeof(goofer_post_note));
    for (int k = 0; k < J_LISTS; ++k)
        if (!jobs[k].empty()) memcpy(host.data() + o_jobs[k], jobs[k].data(), jobs[k].size() * sizeof(goofer_onepole_job));
    memcpy(host.data() + o_any, on_any.data(), n);
    memcpy(host.data() + o_sd, on_sd.data(), n);
    memcpy(host.data() + o_pd, on_pd.data(), n);
    if (!taps20.empty()) memcpy(host.data() + o_t20, taps20.data(), taps20.size() * sizeof(double));
    if (!taps_pd.empty()) memcpy(host.data() + o_tpd, taps_pd.data(), taps_pd.size() * sizeof(double));

    const bool need_tmp = any_fry || any_st;
    const bool need_d = any_sd || any_pd;
    size_t need = blob + 4096 + (need_tmp ? 2 * al((size_t)N * sizeof(float)) : 0) +
                  (need_d ? 2 * al((size_t)N * sizeof(double)) : 0) + (any_pd ? al((size_t)N * sizeof(double)) : 0) +
                  4 * al((size_t)n * sizeof(double));
    int rc = ensure_scratch(ctx, need);
    if (rc) return rc;
    arena a{(char *)ctx->scratch, ctx->scratch_bytes, 0};
    unsigned char *d_blob = a.take<unsigned char>(blob);
    float *tmpA = need_tmp ? a.take<float>(N) : nullptr, *tmpB = need_tmp ? a.take<float>(N) : nullptr;
    double *tmpD1 = need_d ? a.take<double>(N) : nullptr, *tmpD2 = need_d ? a.take<double>(N) : nullptr;
    double *dyn = any_pd ? a.take<double>(N) : nullptr;
    double *sums = a.take<double>(2 * (size_t)n), *ref = a.take<double>(n);
    if (!d_blob || !sums || !ref || (need_tmp && (!tmpA || !tmpB)) || (need_d && (!tmpD1 || !tmpD2)) || (any_pd && !dyn))
        return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    HIP_TRY(ctx, hipMemcpyAsync(d_blob, host.data(), blob, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                   // the staging vector dies with this call
    const goofer_post_note *d_notes = (const goofer_post_note *)(d_blob + o_notes);
    auto d_jobs = [&](int k) { return (const goofer_onepole_job *)(d_blob + o_jobs[k]); };
    const unsigned char *d_any = d_blob + o_any, *d_sd = d_blob + o_sd, *d_pd = d_blob + o_pd;
    const double *d_t20 = (const double *)(d_blob + o_t20), *d_tpd = (const double *)(d_blob + o_tpd);

    // su / sj layers
    if (any_layers) {
        if ((rc = launch_onepole(ctx, p->su_harm, p->su_harm, p->f0, d_jobs(J_SU), (int)jobs[J_SU].size(), st))) return rc;
        if ((rc = launch_onepole(ctx, p->sj_harm, p->sj_harm, p->f0, d_jobs(J_SJ), (int)jobs[J_SJ].size(), st))) return rc;
        if ((rc = launch_post_layers(ctx, p->harm, p->su_harm, p->sj_harm, d_notes, p->sample_off, n, N, st))) return rc;
    }
    // fry part 2
    if (any_fry) {
        if ((rc = launch_onepole(ctx, p->harm, tmpA, p->f0, d_jobs(J_FRY_H), (int)jobs[J_FRY_H].size(), st))) return rc;
        if ((rc = launch_onepole(ctx, p->bre, tmpB, p->f0, d_jobs(J_FRY_B), (int)jobs[J_FRY_B].size(), st))) return rc;
        if ((rc = launch_post_fry(ctx, p->harm, p->bre, tmpA, tmpB, d_notes, p->sample_off, n, N, st))) return rc;
    }
    // sd dryness
    if (any_sd) {
        if ((rc = launch_gauss_samples<float>(ctx, p->mask, p->sample_off, n, N, d_t20, r20, d_sd, tmpD1, st))) return rc;
        if ((rc = launch_post_sd(ctx, p->bre, tmpD1, d_notes, p->sample_off, n, N, st))) return rc;
    }
    // st tension
    if (any_st) {
        HIP_TRY(ctx, hipMemsetAsync(sums, 0, 2 * (size_t)n * sizeof(double), st));
        if ((rc = launch_note_sumsq(ctx, p->harm, p->bre, d_notes, p->sample_off, n, N, sums, st))) return rc;
        if ((rc = launch_onepole(ctx, p->harm, tmpA, p->f0, d_jobs(J_ST_HP), (int)jobs[J_ST_HP].size(), st))) return rc;
        if ((rc = launch_onepole(ctx, p->harm, p->harm, p->f0, d_jobs(J_ST_H), (int)jobs[J_ST_H].size(), st))) return rc;
        if ((rc = launch_onepole(ctx, p->bre, p->bre, p->f0, d_jobs(J_ST_B), (int)jobs[J_ST_B].size(), st))) return rc;
        if ((rc = launch_post_tension(ctx, p->harm, p->bre, tmpA, d_notes, p->sample_off, n, N, st))) return rc;
        if ((rc = launch_note_sumsq(ctx, p->harm, p->bre, d_notes, p->sample_off, n, N, sums + n, st))) return rc;
        if ((rc = launch_post_scale(ctx, p->harm, p->bre, sums, sums + n, d_notes, p->sample_off, n, N, st))) return rc;
    }
    // pd gain curve
    if (any_pd) {
        if ((rc = launch_gauss_samples<float>(ctx, p->bend, p->sample_off, n, N, d_tpd, r_pd, d_pd, tmpD1, st))) return rc;
        if ((rc = launch_gauss_samples<float>(ctx, p->mask, p->sample_off, n, N, d_tpd, r_pd, d_pd, tmpD2, st))) return rc;
        if ((rc = launch_dyn_gain(ctx, tmpD1, tmpD2, d_pd, ref, d_notes, p->sample_off, n, N, dyn, st))) return rc;
    }
    return launch_post_mix(ctx, p->harm, p->uv, p->bre, p->sa_uv, p->sa_bre, dyn, d_notes, d_any, p->params, p->sample_off, n, N,
                           p->mix, st);
}

static int ensure_side_stream(goofer_ctx *ctx)
{
    if (ctx->side) return GOOFER_OK;
    {
        // the pulse chain (f0 kernel -> onsets -> placement) is the longest dependency chain of a step and shares the chip with
        // the envelope kernels and the noise walker of the caller's stream: its workgroups go first (highest stream priority)
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, greatest));
    }
    hipEvent_t *evs[] = {&ctx->ev_fork, &ctx->ev_join, &ctx->ev_maps, &ctx->ev_entry, &ctx->ev_f0, &ctx->ev_f0s};
    for (hipEvent_t *e : evs)
        if (!*e) HIP_TRY(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return GOOFER_OK;
}

int goofer_synth_batch(goofer_ctx *ctx, const goofer_batch *b, void *stream)
{
    NEED_PLAN(ctx);
    if (!b) return goofer_fail(ctx, GOOFER_EINVAL, "null batch");
    const goofer_plan_t &p = ctx->plan;
    if (b->n_bins != p.n_bins || b->ld < p.n_bins) return goofer_fail(ctx, GOOFER_EINVAL, "batch geometry does not match the plan");
    if (b->n_notes <= 0 || b->total_samples <= 0) return GOOFER_OK;
    hipStream_t st = (hipStream_t)stream;
    ctx->frame_picks = nullptr;
    const int64_t F = b->total_frames, N = b->total_samples;
    const int n = b->n_notes, ld = b->ld, ldc = spec_stride(p.n_bins);

    const bool jit_f0 = b->noise_f0 != nullptr, vol_vib = b->volume_vibrato != 0,
               jit_vol = vol_vib || (b->noise_vol_h != nullptr && b->noise_vol_b != nullptr);
    const bool sub_on = b->subharm_ratio > 0.0;
    const bool sub_jit = sub_on && b->noise_subharm != nullptr;
    // gf.synthesize behind its time stretch: f0_interp is a float64 array from there on (GOOFER.py:1053) — the f0 jitter multiplies
    // it (the pulse train sees the float32 cast of the product, :1071-1074) and the sub-harmonic trackers accumulate it (:1077-1097);
    // a private copy, since the jitters work in place
    const bool f64_on = b->f0_64 != nullptr && (jit_f0 || sub_on);
    const size_t jit_bytes = ((jit_f0 || jit_vol || sub_jit) ? (3 * (size_t)N * sizeof(double) + 3 * 256 * (size_t)n + 8192) : 0) +
                             (sub_on ? ((size_t)N * (sizeof(double) + sizeof(double)) + 3 * 256 * (size_t)n + 8192) : 0) +
                             (f64_on ? (size_t)N * sizeof(double) + 1024 : 0);
    // which pipeline will run decides what the arena holds (the same predicate as `stem_path` below)
    const bool walkers = ctx->stems && ctx->ola_fused && (p.hop % 2 == 0) && stems_supported(p) && !sub_on && !jit_vol;
    // n_fft 2048: one stem per wave (two waves per SIMD instead of one), then the per-note finish of the stem-split path
    const bool ola_split = !walkers && ctx->ola_fused && (p.hop % 2 == 0) && ctx->stems && ola_split_supported(p) && !jit_vol;
    int rc = ensure_scratch(ctx, scratch_need(p, F, N, n, walkers ? 0 : 1, sub_on) + (size_t)F * (ld - ((p.n_bins + 3) & ~3)) * 2 * sizeof(float) + jit_bytes);
    if (rc) return rc;
    arena a{(char *)ctx->scratch, ctx->scratch_bytes, 0};
    int *frame_note = a.take<int>(F);
    int64_t *row_src = a.take<int64_t>(F);
    float2 *picks = a.take<float2>(F);
    float *f0s = a.take<float>(N);
    double *inc = sub_on ? a.take<double>(N) : nullptr;       // increments of the sub-harmonic trackers ('sg') only
    const size_t onset_slots = (size_t)(sub_on ? N : N / 2) + 16 * (size_t)n + 16;   // (see scratch_need)
    char *onsets = a.take<char>(onset_slots * ONSET_BYTES);
    int32_t *onset_idx = a.take<int32_t>(onset_slots);
    int32_t *onset_cnt = a.take<int32_t>(n + 16);
    int32_t *ovf = ctx->ovf_flag;
    float *pulse = a.take<float>(N);
    int32_t *pulse_tiles = a.take<int32_t>((size_t)PULSE_TILE_INTS(N));
    const size_t spec_n = walkers ? 0 : (size_t)F * ldc, frame_n = walkers ? 0 : (size_t)F * p.n_fft;
    float2 *S_h = a.take<float2>(spec_n);
    float2 *S_uv = a.take<float2>(spec_n);
    float2 *S_br = a.take<float2>(spec_n);
    float *frames = a.take<float>(frame_n);
    float *frames_u = a.take<float>(frame_n);
    float *frames_b = a.take<float>(frame_n);
    float *env_h = a.take<float>((size_t)F * ld);
    float *env_n = a.take<float>(walkers ? 0 : (size_t)F * ld);
    double *short_s = a.take<double>(N / 4 + n + 16);
    float *note_mag = a.take<float>(2 * (size_t)n + 16);
    double *note_steps = a.take<double>(2 * (size_t)n + 16);
    // stem walkers: a byte per output hop of a note (T + 3 of them) — which stems the noise walker left unstored because they are
    // exactly zero there (k_noise_stems -> k_note_finish)
    unsigned char *hopz = a.take<unsigned char>(walkers ? (size_t)F + 3 * (size_t)n + 64 : 0);
    // ... with the exact sparsity of the noise stems decided per frame up front (k_frame_skip)
    const bool skip_frames = ola_split && ctx->skip_zero && ctx->overlap && !sub_on && p.hop <= 512;
    unsigned char *hop_flat = skip_frames ? a.take<unsigned char>((size_t)F + (size_t)((p.n_fft + p.hop - 1) / p.hop) * n + 16) : nullptr;
    unsigned char *frame_skip = skip_frames ? a.take<unsigned char>((size_t)F + 16) : nullptr;
    unsigned char *knot_eq = skip_frames ? a.take<unsigned char>((size_t)(N / 4 + n + 16)) : nullptr;
    if (skip_frames && (!hop_flat || !frame_skip || !knot_eq)) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    if (!picks || !frames_u || !frames_b || !frame_note || !row_src || !f0s || (sub_on && !inc) || !onsets || !onset_idx || !onset_cnt || !ovf || !pulse || !pulse_tiles || !S_h || !S_uv || !S_br || !frames ||
        !env_h || !env_n || !short_s || !note_mag || !note_steps || !hopz)
        return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    float *note_peak = note_mag + n;
    double *jit_a = nullptr, *jit_b = nullptr, *jit_c = nullptr;
    unsigned long long *jit_max = nullptr;
    unsigned char *on_f0 = nullptr, *on_vol = nullptr;
    if (jit_f0 || jit_vol || sub_jit) {
        jit_a = a.take<double>(N); jit_b = a.take<double>(N); jit_c = a.take<double>(N);
        jit_max = a.take<unsigned long long>(3 * (size_t)n + 16);
        on_f0 = a.take<unsigned char>(n + 16); on_vol = a.take<unsigned char>(n + 16);
        if (!jit_a || !jit_b || !jit_c || !jit_max || !on_f0 || !on_vol) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    }
    double *sub_buf = nullptr;
    double *sub_fm = nullptr;
    unsigned long long *sub_max = nullptr;
    unsigned char *on_sub = nullptr, *on_subj = nullptr;
    if (sub_on) {
        sub_buf = a.take<double>(N); sub_fm = a.take<double>(N);
        sub_max = a.take<unsigned long long>(n + 16); on_sub = a.take<unsigned char>(n + 16);
        on_subj = a.take<unsigned char>(n + 16);
        if (!sub_buf || !sub_fm || !sub_max || !on_sub || !on_subj) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
    }
    double *f0d = nullptr;
    if (f64_on) {
        f0d = a.take<double>(N);
        if (!f0d) return goofer_fail(ctx, GOOFER_ENOMEM, "scratch arena too small");
        HIP_TRY(ctx, hipMemcpyAsync(f0d, b->f0_64, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    {
        const void *ptrs[] = {frame_note, row_src, f0s, pulse, S_h, S_uv, S_br, frames, env_h, env_n, short_s, note_mag, note_peak, onset_cnt};
        size_t bytes[] = {F * sizeof(int), F * sizeof(int64_t), N * sizeof(float), N * sizeof(float), spec_n * sizeof(float2),
                          spec_n * sizeof(float2), spec_n * sizeof(float2), frame_n * sizeof(float),
                          (size_t)F * ld * sizeof(float), walkers ? 0 : (size_t)F * ld * sizeof(float), (N / 4 + n) * sizeof(double),
                          n * sizeof(float), n * sizeof(float), n * sizeof(int32_t)};
        for (int i = 0; i < 14; ++i) { ctx->dbg_ptr[i] = ptrs[i]; ctx->dbg_bytes[i] = bytes[i]; }
        ctx->dbg_ptr[14] = onset_idx; ctx->dbg_bytes[14] = (N / 2 + 16 * (size_t)n) * sizeof(int32_t);
        ctx->dbg_ptr[15] = frame_skip; ctx->dbg_bytes[15] = frame_skip ? (size_t)F : 0;   // per frame: bit 0 unvoiced, bit 1 breath transform skipped
    }

    // mask-smoothing taps for this call's sigma; device copy cached on the handle (steady state:
    // no host work, no synchronisation)
    if (ctx->mask_taps_sigma != b->transition_sigma || !ctx->mask_taps) {
        std::vector<double> mtaps;
        int mrad;
        gauss_taps_host(std::max(1.0, (double)b->transition_sigma / 4.0), mtaps, mrad);   // GOOFER.py:561
        if (mrad > 2048) return goofer_fail(ctx, GOOFER_EINVAL, "transition sigma too large");
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (!ctx->mask_taps) HIP_TRY(ctx, hipMalloc((void **)&ctx->mask_taps, 4097 * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(ctx->mask_taps, mtaps.data(), mtaps.size() * sizeof(double), hipMemcpyHostToDevice));
        ctx->mask_taps_sigma = b->transition_sigma;
        ctx->mask_taps_radius = mrad;
        double acc = 0.0;
        for (double tv : mtaps) acc += tv * 1.0;
        ctx->mask_taps_sum = acc;
    }
    const double *d_mtaps = ctx->mask_taps;
    const int mrad = ctx->mask_taps_radius;

    hipEvent_t *pev = nullptr;
    if (ctx->prof_on && ctx->prof_steps < ctx->prof_cap) pev = ctx->prof_ev + (size_t)ctx->prof_steps * (PROF_STAGES + 1);
    int stage = 0;
    // option "prof_only" = s: only the events stage s needs are recorded (bench.py's timed steps carry the dominant kernel's two
    // events; the twenty records of the full breakdown cost 0.06 ms of a 2.3 ms step).  The two kernels launched beside the pulse
    // chain on the caller's stream (stages 6 / 7 of the stem path, 9 / 12 of the other) are bracketed by the prof_main2 pair.
    const int only = ctx->prof_only;
    const bool want_q = only < 0 || only == 6 || only == 7 || only == 9 || only == 12;
#define MARK_Q(q)                                                    \
    do {                                                             \
        if (pev && want_q) HIP_TRY(ctx, hipEventRecord(ctx->prof_main2[(size_t)ctx->prof_steps * 2 + (q)], st)); \
    } while (0)
#define MARK()                                                       \
    do {                                                             \
        if (pev && (only < 0 || stage == only || stage == only + 1 || (stage == 5 && (only == 6 || only == 9)))) \
            HIP_TRY(ctx, hipEventRecord(pev[stage], st));           \
        ++stage;                                                     \
    } while (0)

    // the fused overlap-add rings index by position mod n_fft with a mask: power-of-two transforms only (768 / 1536 take the
    // separate irFFT + gather kernels)
    const bool ola_one = ctx->ola_fused && (p.hop % 2 == 0) && (p.n_fft & (p.n_fft - 1)) == 0 && p.bl_L == 0;   // (64 .. 256: Bluestein plans)
    unsigned fb = (unsigned)((F + 255) / 256);
    // goofer_render_batch: the assembly recorded ev_f0 right after the f0 / mask kernel.  The pulse chain (f0 scaling,
    // sequential walk, placement) then runs on the side stream from that point on, beside the envelope assembly and the
    // map kernels, instead of starting when this call's first kernel is reached in stream order.
    // Stem-split walkers (stems.hip): no spectra in HBM.  The legacy kernels stay for the other geometries, for the
    // volume-jitter / sub-harmonic layers (which edit the stems or the pulse train between the steps) and as the A/B path.
    const bool stem_path = walkers;                           // (== ctx->stems && ola_one && stems_supported(p) && !sub_on && !jit_vol)
    if (pev) ctx->prof_stems = stem_path;
    const bool side_on = ctx->overlap && ola_one && !sub_on;
    const bool early = side_on && !jit_f0 && ctx->early_req && ctx->early_f0 == b->f0 && ctx->side != nullptr;
    // f0 * pitch_shift (GOOFER.py:995).  When the caller vouches that every pitch_shift is 1 (the resampler path: the pitch
    // lives in the curve) and nothing jitters f0 in place, the input array IS the scaled f0 and the pass is skipped.
    const bool f0_alias = b->unit_pitch_shift && !jit_f0 && !sub_jit;
    if (f0_alias) {
        f0s = const_cast<float *>(b->f0);
        ctx->dbg_ptr[2] = f0s;
    }
    // goofer_render_batch ran the f0 / mask kernel on the side stream: the caller's stream reads them from here on
    if (ctx->f0_on_side) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_f0, 0));
        ctx->f0_on_side = false;
    }
    // stem path: one launch for all the frame maps
    const bool maps_fused = stem_path;
    if (!maps_fused) HIP_TRY(ctx, hipMemsetAsync(note_mag, 0, 2 * (size_t)n * sizeof(float), st));
    MARK();   // 0: setup
    if (early) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_entry, 0));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_f0, 0));
        if (!f0_alias) {
            hipLaunchKernelGGL(k_scale_f0, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, ctx->side, b->f0, b->sample_off, n, N,
                               b->params, f0s);
            LAUNCH_CHECK(ctx);
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_f0s, ctx->side));
    }
    if (!maps_fused && (rc = launch_frame_note(ctx, b->frame_off, n, F, frame_note, st))) return rc;
    if (!early && !f0_alias) {
        hipLaunchKernelGGL(k_scale_f0, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, b->f0, b->sample_off, n, N, b->params,
                           f0s);                                     // (the pulse walk divides by sr itself)
        LAUNCH_CHECK(ctx);
    }
    // per-frame (f0, mask) picks ride on the map kernel when the scaled f0 is final at this point of the caller's stream:
    // nothing jitters it in place later, and it is not being produced on the side stream
    const bool picks_on = !jit_f0 && !sub_jit && !(early && !f0_alias);
    ctx->frame_picks = picks_on ? picks : nullptr;
    if (maps_fused) {
        const int64_t threads = std::max<int64_t>(F, 2 * (int64_t)n);
        hipLaunchKernelGGL(k_frame_maps, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, b->frame_off, b->env_off, n, F, frame_note,
                           row_src, b->sample_off, (const float *)f0s, b->mask, p.hop, picks_on ? picks : (float2 *)nullptr, note_steps, note_mag);
    } else {
        hipLaunchKernelGGL(k_row_src, dim3(fb), dim3(256), 0, st, b->frame_off, b->env_off, frame_note, F, row_src, b->sample_off,
                           (const float *)f0s, b->mask, p.hop, picks_on ? picks : (float2 *)nullptr);
    }
    LAUNCH_CHECK(ctx);
    if (jit_f0 || jit_vol) {
        hipLaunchKernelGGL(k_note_flags, dim3((n + 255) / 256), dim3(256), 0, st, b->params, n, on_f0, on_vol);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, hipMemsetAsync(jit_max, 0, 3 * (size_t)n * sizeof(unsigned long long), st));
    }
    if (jit_f0) {   // 'sh': f0 *= 1 + (jitter - 1) * mask, after pitch_shift and before the pulse train (GOOFER.py:1069-1071)
        const double *d_t; int r;
        if ((rc = upload_jitter_taps(ctx, (double)b->f0_jitter_sigma, 0, &d_t, &r, st))) return rc;
        if ((rc = launch_gauss_samples<double>(ctx, b->noise_f0, b->sample_off, n, N, d_t, r, on_f0, jit_a, st))) return rc;
        if ((rc = launch_note_absmax(ctx, jit_a, b->sample_off, n, N, on_f0, jit_max, st))) return rc;
        if ((rc = launch_f0_jitter(ctx, f0s, f0d, b->mask, jit_a, jit_max, b->sample_off, n, N, b->params, 0, st))) return rc;
    }
    // aperiodic half of the stem-split path: smoothed mask knots, then the two noise stems straight to samples.  Needs the
    // final scaled f0 (frame picks) and nothing of the pulse chain.
    auto stems_aperiodic = [&]() -> int {
        int r2;
        if (!picks_on && (r2 = launch_frame_picks(ctx, b->frame_off, frame_note, F, b->sample_off, f0s, b->mask, picks, st))) return r2;
        if ((r2 = launch_mask_short(ctx, b->mask, b->sample_off, n, N, d_mtaps, mrad, ctx->mask_taps_sum, short_s, st))) return r2;
        if (side_on) MARK_Q(0);
        if (!maps_fused && (r2 = launch_note_steps(ctx, b->sample_off, n, note_steps, st))) return r2;
        if ((r2 = launch_noise_stems(ctx, b->env_noise ? b->env_noise : b->env, ld, row_src, b->phi, F, frame_note, b->frame_off,
                                     b->sample_off, picks, b->params, b->seed, b->env_noise != nullptr, short_s, note_steps, b->uv,
                                     b->bre, hopz, st)))
            return r2;
        if (side_on) MARK_Q(1);
        return GOOFER_OK;
    };
    // The pulse walk is one latency-bound wave per SIMD: it goes to a side stream FIRST (so its workgroups are resident
    // from the start), and the aperiodic branch — noise spectra, mask smoothing, which depend only on the maps and
    // the scaled f0 — fills the rest of the machine from the caller's stream meanwhile.
    hipEvent_t *sev = nullptr;
    hipStream_t pst = st;                                     // stream of the pulse chain
    if (side_on) {
        int rc2 = ensure_side_stream(ctx);
        if (rc2) return rc2;
        if (pev && (only < 0 || (only >= 3 && only <= 5))) sev = ctx->prof_side + (size_t)ctx->prof_steps * 4;
        if (stem_path && !ctx->warp_done) HIP_TRY(ctx, hipEventRecord(ctx->ev_maps, st));   // the frame maps and everything before them on this stream (for k_warp_bins)
        if (!early) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        }
        pst = ctx->side;
        ctx->prof_side_used = true;
    } else if (pev) {
        ctx->prof_side_used = false;
    }
    MARK();   // 1: noise envelope = sigma-1.75 blur of the un-warped rows (GOOFER.py:993)
    // (folded into k_noise_spectra / k_harm_shape: the standalone envelope kernels remain as C-ABI entry points)
    MARK();   // 2: harmonic envelope = formant-anchored + uniform warp

    MARK();   // 3..5: pulse train
    if (sev) HIP_TRY(ctx, hipEventRecord(sev[0], pst));
    MARK();
    if (sev) HIP_TRY(ctx, hipEventRecord(sev[1], pst));
    if ((rc = launch_pulse_onsets(ctx, f0s, 1.0f, b->sample_off, n, (onset_t *)onsets, onset_idx, onset_cnt, ovf, N, pulse_tiles, pst))) return rc;
    MARK();
    if (sev) HIP_TRY(ctx, hipEventRecord(sev[2], pst));
    if ((rc = launch_pulse_place(ctx, (onset_t *)onsets, onset_cnt, b->sample_off, n, N, pulse, pulse_tiles, pst))) return rc;
    // Harmonic envelope rows for the harmonic walker: formant-anchored + uniform warp, one wave per row (GOOFER.py:1004-1017),
    // behind the pulse placement on its stream (the caller's stream carries the mask smoothing and the noise walker meanwhile).
    // Not inside the walker: the crossing-anchor path is several times slower than the sorted one, and a walker wave holds
    // ~95 frames of ONE note, so the slow notes would set the kernel's time.
    const bool warp_ready = stem_path && ctx->warp_done;               // goofer_render_batch: the assembly already wrote the warped rows
    if (stem_path && side_on && !warp_ready) {
        HIP_TRY(ctx, hipStreamWaitEvent(pst, ctx->ev_maps, 0));   // frame_note / row_src come from the caller's stream
        if ((rc = launch_warp_bins(ctx, b->env, env_h, F, p.n_bins, ld, b->formants, nullptr, b->params, frame_note, row_src, 1.0, pst)))
            return rc;
    }
    if (side_on) {
        if (sev) HIP_TRY(ctx, hipEventRecord(sev[3], pst));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_join, pst));
        // meanwhile, on the caller's stream
        if (early && !f0_alias) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_f0s, 0));   // the scaled f0 comes from the side stream
        if (stem_path) {
            if ((rc = stems_aperiodic())) return rc;
        } else {
            // (the skip bits need the smoothed mask: it goes first then)
            if (skip_frames) {
                if ((rc = launch_mask_short(ctx, b->mask, b->sample_off, n, N, d_mtaps, mrad, ctx->mask_taps_sum, short_s, st))) return rc;
                if ((rc = launch_frame_skip(ctx, short_s, N / 4 + n, b->sample_off, b->frame_off, frame_note, n, F, knot_eq, hop_flat, frame_skip, st))) return rc;
            }
            if ((rc = launch_noise_spectra(ctx, S_uv, S_br, ldc, F, frame_note, b->frame_off, b->sample_off, f0s, b->mask,
                                           b->env_noise ? b->env_noise : b->env, b->phi, ld, b->params, b->seed, row_src,
                                           b->env_noise != nullptr, frame_skip, st)))
                return rc;
            MARK_Q(0);
            if (!skip_frames && (rc = launch_mask_short(ctx, b->mask, b->sample_off, n, N, d_mtaps, mrad, ctx->mask_taps_sum, short_s, st))) return rc;
            MARK_Q(1);
        }
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    }
    if (sub_on) {   // 'sg': extra LF pulse layer at f0 * ratio with vibrato, added to the pulse train (GOOFER.py:1076-1097)
        hipLaunchKernelGGL(k_note_sub_flags, dim3((n + 255) / 256), dim3(256), 0, st, b->params, n, on_sub, on_subj);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, hipMemsetAsync(sub_max, 0, (size_t)n * sizeof(unsigned long long), st));
        if (sub_jit) {   // subharm_f0_jitter: f0 (the array itself, as in the reference) *= 1 + (jitter - 1) * mask   :1078-1080
            const double *d_t; int r;
            if ((rc = upload_jitter_taps(ctx, (double)b->f0_jitter_sigma, 0, &d_t, &r, st))) return rc;
            HIP_TRY(ctx, hipMemsetAsync(jit_max, 0, (size_t)n * sizeof(unsigned long long), st));
            if ((rc = launch_gauss_samples<double>(ctx, b->noise_subharm, b->sample_off, n, N, d_t, r, on_subj, jit_a, st))) return rc;
            if ((rc = launch_note_absmax(ctx, jit_a, b->sample_off, n, N, on_subj, jit_max, st))) return rc;
            if ((rc = launch_f0_jitter(ctx, f0s, f0d, b->mask, jit_a, jit_max, b->sample_off, n, N, b->params, 1, st))) return rc;
        }
        double ratios[16];
        ratios[0] = b->subharm_ratio;
        for (int q = 0; q < 15; ++q) ratios[q + 1] = b->subharm_more[q];
        int n_ratios = 1;
        while (n_ratios < 16 && ratios[n_ratios] > 0.0) ++n_ratios;
        if ((rc = launch_subharm(ctx, f0s, f0d, b->mask, b->sample_off, n, N, b->params, ratios, n_ratios, b->subharm_vibrato,
                                 b->subharm_vib_rate, b->subharm_vib_depth, b->subharm_vib_delay, sub_fm, inc, (onset_t *)onsets,
                                 onset_idx, onset_cnt, ovf, on_sub, sub_buf, sub_max, pulse, st)))
            return rc;
    }
    if (stem_path) {
        MARK();   // 6: mask_short, 7: noise_stems (here when nothing runs beside the pulse chain)
        if (!side_on && (rc = launch_mask_short(ctx, b->mask, b->sample_off, n, N, d_mtaps, mrad, ctx->mask_taps_sum, short_s, st))) return rc;
        MARK();
        if (!side_on) {
            if (!picks_on && (rc = launch_frame_picks(ctx, b->frame_off, frame_note, F, b->sample_off, f0s, b->mask, picks, st))) return rc;
            if (!maps_fused && (rc = launch_note_steps(ctx, b->sample_off, n, note_steps, st))) return rc;
            if ((rc = launch_noise_stems(ctx, b->env_noise ? b->env_noise : b->env, ld, row_src, b->phi, F, frame_note, b->frame_off,
                                         b->sample_off, picks, b->params, b->seed, b->env_noise != nullptr, short_s, note_steps, b->uv,
                                         b->bre, hopz, st)))
                return rc;
            if (!ctx->warp_done &&
                (rc = launch_warp_bins(ctx, b->env, env_h, F, p.n_bins, ld, b->formants, nullptr, b->params, frame_note, row_src, 1.0, st)))
                return rc;
        }
        MARK();   // 8
        MARK();   // 9: harm_stem = rFFT + shaping + irFFT + overlap-add of the harmonic stem
        if ((rc = launch_harm_stem(ctx, pulse, ctx->warp_done ? ctx->warp_rows : env_h, ctx->warp_done ? b->env : nullptr, b->formants != nullptr, ld,
                                   ctx->warp_done ? row_src : nullptr, F, frame_note,
                                   b->frame_off, b->sample_off, picks, b->params, b->harm, note_mag, st)))
            return rc;
        MARK();   // 10..12
        MARK();
        MARK();
        MARK();   // 13: harm / max|S|, peak, gain, reconstruct, mix
        if ((rc = launch_note_finish(ctx, b->harm, b->uv, b->bre, b->rec, b->mix, b->sample_off, n, b->params, note_mag, note_peak,
                                     !(b->mix_only && (b->mix || b->rec)), hopz, b->frame_off, st)))
            return rc;
        MARK();   // 14..17 unused
        MARK();
        MARK();
        MARK();
        MARK();   // end
        if (pev) ctx->prof_steps++;
        ctx->frame_picks = nullptr;
        return GOOFER_OK;
    }
    // spectra -> windowed time frames of the three stems
    {
        MARK();   // 6: framewise rFFT of the pulse train
        if ((rc = launch_rfft_frames_mapped(ctx, pulse, b->sample_off, b->frame_off, frame_note, F, S_h, ldc, st))) return rc;
        MARK();   // 7
        if ((rc = launch_harm_shape(ctx, S_h, ldc, F, frame_note, b->frame_off, b->sample_off, f0s, b->mask, b->env, ld, b->params,
                                    note_mag, row_src, b->formants, b->no_warp != 0, st)))
            return rc;
        MARK();   // 8
        if (!ola_one && (rc = launch_irfft_frames(ctx, S_h, ldc, F, frames, st))) return rc;
        MARK();   // 9: aperiodic spectra
        if (!side_on && (rc = launch_noise_spectra(ctx, S_uv, S_br, ldc, F, frame_note, b->frame_off, b->sample_off, f0s, b->mask,
                                                   b->env_noise ? b->env_noise : b->env, b->phi, ld, b->params, b->seed, row_src,
                                                   b->env_noise != nullptr, nullptr, st)))
            return rc;
        MARK();   // 10, 11
        if (!ola_one && (rc = launch_irfft_frames(ctx, S_br, ldc, F, frames_b, st))) return rc;
        MARK();
        if (!ola_one && (rc = launch_irfft_frames(ctx, S_uv, ldc, F, frames_u, st))) return rc;
    }
    MARK();   // 12: decimated + smoothed voicing mask
    if (!side_on && (rc = launch_mask_short(ctx, b->mask, b->sample_off, n, N, d_mtaps, mrad, ctx->mask_taps_sum, short_s, st))) return rc;
    MARK();   // 13: (irFFT of the three stems +) overlap-add + gains + per-note peak, one pass
    if (ola_split) {
        if ((rc = launch_irfft_ola1(ctx, S_h, S_uv, S_br, ldc, F, frame_note, b->frame_off, b->sample_off, n, short_s, note_steps,
                                    b->params, b->harm, b->uv, b->bre, frame_skip, st)))
            return rc;
        MARK();   // 14
        if ((rc = launch_note_finish(ctx, b->harm, b->uv, b->bre, b->rec, b->mix, b->sample_off, n, b->params, note_mag, note_peak,
                                     !(b->mix_only && (b->mix || b->rec)), nullptr, nullptr, st)))
            return rc;
        MARK();   // 15..17 unused
        MARK();
        MARK();
        MARK();   // end
        if (pev) ctx->prof_steps++;
        ctx->frame_picks = nullptr;
        return GOOFER_OK;
    }
    if (ola_one) {
        if ((rc = launch_irfft_ola3(ctx, S_h, S_uv, S_br, ldc, F, frame_note, b->frame_off, b->sample_off, n, note_mag, short_s,
                                    note_steps, b->params, b->harm, b->uv, b->bre, note_peak, st)))
            return rc;
    } else if ((rc = launch_ola3_gains(ctx, frames, frames_u, frames_b, note_mag, short_s, b->sample_off, b->frame_off, n, N,
                                       b->params, note_steps, b->harm, b->uv, b->bre, note_peak, st)))
        return rc;
    if (jit_vol) {  // 'sr': volume jitter on harm / breath, then the peak is taken again (GOOFER.py:1185-1193)
        const double *d_t = nullptr, *d_t20; int r = 0, r20;
        if (!vol_vib && (rc = upload_jitter_taps(ctx, (double)b->vol_jitter_sigma, 1, &d_t, &r, st))) return rc;
        if ((rc = upload_jitter_taps(ctx, 20.0, 2, &d_t20, &r20, st))) return rc;
        if (!vol_vib) {
            if ((rc = launch_gauss_samples<double>(ctx, b->noise_vol_h, b->sample_off, n, N, d_t, r, on_vol, jit_a, st))) return rc;
            if ((rc = launch_gauss_samples<double>(ctx, b->noise_vol_b, b->sample_off, n, N, d_t, r, on_vol, jit_b, st))) return rc;
            if ((rc = launch_note_absmax(ctx, jit_a, b->sample_off, n, N, on_vol, jit_max + n, st))) return rc;
            if ((rc = launch_note_absmax(ctx, jit_b, b->sample_off, n, N, on_vol, jit_max + 2 * (size_t)n, st))) return rc;
        }
        if ((rc = launch_gauss_samples<float>(ctx, b->mask, b->sample_off, n, N, d_t20, r20, on_vol, jit_c, st))) return rc;
        if ((rc = launch_volume_jitter(ctx, b->harm, b->bre, jit_a, jit_b, jit_c, jit_max + n, jit_max + 2 * (size_t)n, b->sample_off, n,
                                       N, b->params, vol_vib ? 1 : 0, (double)b->vol_jitter_speed, st)))
            return rc;
        HIP_TRY(ctx, hipMemsetAsync(note_peak, 0, (size_t)n * sizeof(float), st));
        if ((rc = launch_stem_peak(ctx, b->harm, b->uv, b->bre, b->sample_off, n, N, note_peak, st))) return rc;
    }
    MARK();   // 14: gain, reconstruct, mix
    if ((rc = launch_apply_gain(ctx, b->harm, b->uv, b->bre, b->rec, b->mix, b->sample_off, n, N, b->params, note_peak,
                                !(b->mix_only && (b->mix || b->rec)), st)))
        return rc;
    MARK();   // 15..17 unused
    MARK();
    MARK();
    MARK();   // end
#undef MARK
    if (pev) ctx->prof_steps++;
    ctx->frame_picks = nullptr;
    return GOOFER_OK;
}

// SillySampler.resample end to end for one batch (SillySampler.py:698-1151 up to the post chain): assembly and synthesis as
// one call.  Same kernels and results as goofer_assemble_batch followed by goofer_synth_batch; the difference is scheduling.
// Because both descriptors are in hand at once, everything they point to is known to be enqueued before this call, so
// the synthesis' pulse chain may start on the side stream as soon as the assembled f0 exists.
int goofer_render_batch(goofer_ctx *ctx, const goofer_assembly *asmb, const goofer_batch *b, void *stream)
{
    NEED_PLAN(ctx);
    if (!asmb || !b) return goofer_fail(ctx, GOOFER_EINVAL, "null descriptor");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (ctx->overlap && asmb->f0_out == b->f0 && asmb->n_notes > 0) {
        if ((rc = ensure_side_stream(ctx))) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_entry, st));      // every input of either descriptor precedes this point
        ctx->early_req = true;
    }
    // Stem-split path: the harmonic walker wants warped envelope rows.  The assembly's frame-gather kernel has every row in
    // hand, so it writes the warped copy too (k_env_rows<true>) — one pass instead of a separate read + write of the matrix.
    ctx->warp_out = nullptr;
    ctx->warp_done = false;
    if (ctx->stems && ctx->ola_fused && ctx->overlap && stems_supported(ctx->plan) && asmb->env_out == b->env &&
        asmb->n_notes == b->n_notes && asmb->total_out_rows == b->total_env_rows && asmb->ld == b->ld && !asmb->any_fry &&
        // ... and the synthesis will take the stem path (the 'sg' layer and the volume jitter run the legacy kernels, which warp
        // in k_harm_shape: a warped copy written here would never be read)
        !(b->subharm_ratio > 0.0) && !(b->volume_vibrato != 0 || (b->noise_vol_h != nullptr && b->noise_vol_b != nullptr))) {
        const size_t need = (size_t)b->total_env_rows * b->ld * sizeof(float);
        if (need > ctx->warp_rows_bytes) {
            HIP_TRY(ctx, hipDeviceSynchronize());
            if (ctx->warp_rows) HIP_TRY(ctx, hipFree(ctx->warp_rows));
            ctx->warp_rows = nullptr;
            ctx->warp_rows_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->warp_rows, need + need / 4));
            ctx->warp_rows_bytes = need + need / 4;
        }
        ctx->warp_formants = b->formants;
        ctx->warp_params = b->params;
        ctx->warp_out = ctx->warp_rows;
    }
    rc = goofer_assemble_batch(ctx, asmb, stream);
    ctx->warp_out = nullptr;
    if (!rc) rc = goofer_synth_batch(ctx, b, stream);
    ctx->warp_done = false;
    if (ctx->f0_on_side) {                                    // (the synthesis returned before it placed the wait)
        (void)hipStreamWaitEvent(st, ctx->ev_f0, 0);
        ctx->f0_on_side = false;
    }
    ctx->early_req = false;
    ctx->early_f0 = nullptr;
    return rc;
}

}  // extern "C"
