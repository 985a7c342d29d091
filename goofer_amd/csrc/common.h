// Internal declarations shared by the HIP translation units of libgoofer_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/goofer_hip.h"

#define WAVE 64
#define PROF_STAGES 18
#define PROF_ASM0 15           // profile stages 15..17: the assembly's k_env_edit, k_env_rows, k_sample_assemble
#define PP_SPT 8                // k_pulse_place: consecutive samples per thread; a tile = one workgroup = 256 * PP_SPT samples
#define PULSE_TILE_INTS(samples) (4 * (((samples) + 256 * PP_SPT - 1) / (256 * PP_SPT)) + 64)   // k_pulse_tiles' table: 4 ints per tile
#define PULSE_TAB_MAX 8192     // pulse lengths served from the shape table: all of them (the reference caps T0 at 8192, GOOFER.py:497-498) — 134 MB of
                               // a 288 GB device; until late in round 5 the table ended at 2048 and k_pulse_place evaluated longer pulses on the fly, whose
                               // fp64 sin / exp / cos set the kernel's registers (115, four waves per SIMD) though no note of the workloads reached them

// the LF glottal-pulse model of gf.pulse_train_numba (GOOFER.py:474, 508): its keyword arguments, defaults = what gf.synthesize passes
struct lf_model {
    double ra = 0.02, rg = 1.7, rk = 0.8;
};
#define PULSE_PEAK_FLOATS (8193 + 1 + 6)   // k_pulse_peak's table + the model's three doubles behind it (8-byte aligned; goofer_debug_table, tests)

struct goofer_plan_t {
    int sr = 0, n_fft = 0, hop = 0, n_bins = 0;
    float *window = nullptr;      // [n_fft] sqrt-Hann, fp32                      GOOFER.py:12-18
    float *window_blur = nullptr; // [n_fft] window x the time-domain image of the sigma-0.5 bin blur (see stems.hip)
    float *blur_edge = nullptr;   // [4][64] per-lane coefficients of the blur's edge correction (bins 1..6 and M-6..M-1)
    float *win_sq = nullptr;      // [n_fft] window*window in fp32 (OLA weights)  GOOFER.py:385
    float *freqs = nullptr;       // [n_bins] rfftfreq fp32                       GOOFER.py:20-26
    float *lin_freqs = nullptr;   // [n_bins] np.linspace(0, sr/2, n_bins) rounded to fp32 (the bells' bin frequencies, SillySampler.py:812)
    float *boost = nullptr;       // [n_bins] linspace(1,100)                     GOOFER.py:28-35
    float *bright_h = nullptr;    // [n_bins] harmonic brightness                 GOOFER.py:42
    float *bright_b = nullptr;    // [n_bins] breath brightness                   GOOFER.py:43
    float2 *tw_full = nullptr;    // [n_fft/2]   exp(-2 pi i k / (n_fft/2))
    float2 *tw_half = nullptr;    // [n_fft/4+1] exp(-2 pi i k / n_fft)
    // transform sizes without a native radix plan (any other even n_fft in [64, 2048]): Bluestein's chirp-z
    // transform of the n_fft/2-point complex DFT through power-of-two transforms of length bl_L >= n_fft - 1 (fft.hip)
    int bl_L = 0;                 // 0: native
    float2 *bl_chirp = nullptr;   // [M]     exp(+i pi n^2 / M)
    float2 *bl_bhat = nullptr;    // [bl_L]  FFT of the wrapped chirp
    float2 *bl_tw = nullptr;      // [bl_L]  exp(-2 pi i k / bl_L)
    float2 *bl_twh = nullptr;     // [M + 1] exp(-i pi k / M)
    lf_model lf;                  // goofer_pulse_model
    float *pulse_peak = nullptr;  // [PULSE_PEAK_FLOATS] peak of the un-normalised LF shape per T0 (fp64 math), then the model's Ra, Rg, Rk
    float *pulse_shape = nullptr; // normalised LF pulses for T0 = 3..PULSE_TAB_MAX back to back (row T0 at T0(T0-1)/2 - 3)
    double *blur5 = nullptr;      // [5] sigma=0.5 taps (brightness blur)         GOOFER.py:1143
    double *blur175 = nullptr;    // [15] sigma=1.75 taps                         GOOFER.py:993
    float taps5_f[5] = {0}, taps175_f[15] = {0};   // the same taps rounded to fp32, host side (passed to kernels by value)
};

struct goofer_ctx {
    int device = 0;
    char err[512] = {0};
    goofer_plan_t plan;
    // scratch (grown by ensure_scratch)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *asm_scratch = nullptr;  // assembly scratch: edited rows + row->note maps
    size_t asm_bytes = 0;
    void *small = nullptr;        // small staging buffer for taps etc.
    size_t small_bytes = 0;
    // device pointers of the last synth batch's intermediates (goofer_debug_fetch; tests only)
    const void *dbg_ptr[16] = {nullptr};
    size_t dbg_bytes[16] = {0};
    bool overlap = true;          // noise spectra + mask smoothing on a side stream, beside the latency-bound pulse walk
    hipStream_t side = nullptr;   // created on first use
                                  // ... with the highest stream priority (measured: 2.41 ms per step against 2.50 at the default, 2.54 at the lowest)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_maps = nullptr;
    // goofer_render_batch: the pulse chain forks as soon as the assembled f0 exists, not when the synthesis call starts
    hipEvent_t ev_entry = nullptr, ev_f0 = nullptr, ev_f0s = nullptr;
    int32_t *ovf_flag = nullptr;           // handle-owned device words ([1], [2]: cumulative counters, goofer_counter); [0] sticky between goofer_check calls: 1 + index (inside its batch) of a
                                           // note whose pulse onsets overflowed their slots, written with atomicMax by every pulse-chain launch
    // goofer_render_batch, stem-split path: the assembly's frame-gather kernel also writes the rows the harmonic walker needs
    // (formant-anchored + uniform warp), into a buffer the handle owns
    const double *warp_formants = nullptr;
    const goofer_note_params *warp_params = nullptr;
    float *warp_out = nullptr;            // non-null for the duration of one goofer_render_batch that wants the fused warp
    bool warp_done = false;               // the assembly wrote warp_rows for the batch being synthesised
    float *warp_rows = nullptr;
    size_t warp_rows_bytes = 0;
    const float2 *frame_picks = nullptr;   // per-frame (f0, mask) picks of the running goofer_synth_batch, or null
    bool early_req = false;           // set for the duration of one goofer_render_batch
    const float *early_f0 = nullptr;  // f0 array ev_f0 stands for (null: no event recorded)
    bool f0_on_side = false;          // goofer_render_batch ran the f0 / mask kernel on the side stream, in front of the pulse chain it feeds: the
                                      // caller's stream waits for ev_f0 before it reads f0 / mask
    hipEvent_t *prof_side = nullptr;    // [prof_cap][4]: boundaries of the pulse chain on the side stream
    hipEvent_t *prof_main2 = nullptr;   // [prof_cap][2]: ends of noise_spectra / mask_short when they run beside it
    hipEvent_t *prof_asm = nullptr;     // [prof_cap][3][2]: the assembly's three large kernels, each on its own stream
    unsigned char prof_asm_mask[4096] = {0};   // which of the three pairs assembly k of the profiled run recorded
    int prof_asm_steps = 0;
    bool prof_side_used = false;
    bool ola_fused = true;        // irFFT x3 + overlap-add + gains in one kernel (k_irfft_ola3); false: separate irFFT launches + k_ola3_gains
    bool stems = true;            // stem-split frame walkers (stems.hip) where the geometry allows (hop == n_fft / 4); false: the
                                  // one-kernel-per-reference-step pipeline with the spectra in HBM (A/B parity path)
    bool skip_zero = true;        // noise walker: skip transforms whose stem gain is exactly zero over everything they reach (option "skip_zero")
    bool td_blur = true;          // stem walkers: the 5-tap bin blur of voiced frames as a window on the frame's samples (option "td_blur")
    bool prof_stems = false;      // the last profiled batch ran the stem-split path (stage order differs)
    bool sa_fast = true;          // k_sample_assemble: the branch-free path with all of a thread's loads in flight together (option "sa_fast"; 0: A/B)
    bool value_f64 = false;       // k_env_edit: round 4's fp64 value arithmetic (fw interpolation, es blur) instead of fp32 — A/B and the
                                  // error-budget tests (option "value_f64")
    int pulse_scan = 1;           // 1: onsets from the parallel phase scan, the sequential walk only for the notes it cannot settle;
                                  // 0: the sequential walk kernel for every note; 2: the scan kernel walks every note (tests)
    // per-context kernel state: hipFuncSetAttribute is per device, and a handle belongs to one device, so what was set /
    // queried is remembered here and never in process-wide statics
    struct kernel_state {
        const void *fn;
        size_t lds;               // dynamic LDS the occupancy below was queried for
        int waves;                // waves of this kernel the device holds at once (0: not queried)
        bool max_lds_set;
    } kstate[32] = {};
    int n_kstate = 0;
    // per-stage HIP-event timing of goofer_synth_batch (goofer_profile_begin/end)
    bool prof_on = false;
    int prof_only = -1;           // >= 0: goofer_profile_begin .. end bracket this stage only (option "prof_only")
    int prof_steps = 0, prof_cap = 0;
    hipEvent_t *prof_ev = nullptr;      // [prof_cap][PROF_STAGES + 1]
    double *mask_taps = nullptr;  // device taps of the voicing-mask smoother, cached per sigma
    float mask_taps_sigma = -1.f;
    int mask_taps_radius = 0;
    double mask_taps_sum = 0.0;   // running fp64 sum of the taps in tap order (the FIR's answer on a window of ones)
};

int goofer_fail(goofer_ctx *ctx, int code, const char *fmt, ...);
// float2 slots per row of a [frames x bins] complex spectrum matrix: n_bins rounded up to 16 (128-byte aligned rows, so that
// the framewise rFFT's 16-byte stores and every row-wise reader start on a cache-line boundary)
static inline int spec_stride(int n_bins) { return (n_bins + 15) & ~15; }
// opt a kernel in to the full 160 KiB of dynamic LDS on this handle's device (once per handle)
int kernel_allow_max_lds(goofer_ctx *ctx, const void *fn, int bytes = 160 * 1024);
// waves of `fn` (256-thread workgroups, `lds` bytes of dynamic LDS) resident on this handle's device at once
int kernel_resident_waves(goofer_ctx *ctx, const void *fn, size_t lds, int *waves);

#define HIP_TRY(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return goofer_fail((ctx), GOOFER_EHIP, "%s failed: %s (%s:%d)", #call,               \
                               hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

#define LAUNCH_CHECK(ctx)                                                                        \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess)                                                                    \
            return goofer_fail((ctx), GOOFER_EHIP, "kernel launch failed: %s (%s:%d)",           \
                               hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

// ---- device helpers ------------------------------------------------------------------------

// numpy 'reflect' (no edge repeat) as a periodic map; n == 1 degenerates to 'edge'.
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n)
{
    if (n <= 1) return 0;
    if (i >= 0 && i < n) return i;                           // in range: the usual case
    const int64_t period = 2 * (n - 1);
    if (i > -n && i < period) return i < 0 ? -i : period - i;   // one reflection, no 64-bit modulo (it costs ~100 instructions)
    int64_t m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - m;
}

// note owning global frame/sample index g given CSR offsets off[0..n]: largest k with off[k] <= g.
__device__ __forceinline__ int csr_find(const int64_t *__restrict__ off, int n, int64_t g)
{
    int lo = 0, hi = n;  // invariant off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// Wave-cooperative version of csr_find for any non-decreasing key(k), k in [0, n]: all 64 lanes probe 64 evenly
// spaced positions per round (one round of dependent loads narrows the range 64x: 2 rounds for 4096 notes instead
// of 12 serial steps).  Every lane of the calling wave must be active; every lane gets the result.
template <typename Key>
__device__ __forceinline__ int wave_find(int n, int64_t g, int lane, Key key)
{
    int lo = 0, hi = n;                                      // invariant key(lo) <= g < key(hi)
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int p = lo + (lane + 1) * step;
        const bool le = p < hi && key(p) <= g;
        const int c = __popcll(__ballot(le));                // probes <= g form a prefix (keys are sorted)
        const int nhi = lo + (c + 1) * step;
        lo += c * step;
        if (nhi < hi) hi = nhi;
    }
    return lo;
}

// Kernels that tile the concatenated sample axis call this first: it finds the notes of the block's
// first and last sample.  When they coincide (almost always: a note is ~190 blocks long) the caller
// runs its body with that index held in an SGPR, so every per-note load behind it (offsets, params,
// constants) is a scalar load instead of a chain of dependent per-lane vector loads.  The search itself is
// done by the first wave cooperatively (the serial binary search used to cost ~20 dependent loads per block,
// which bounded the short elementwise kernels).  Workgroups must be at least one full wave.
__device__ __forceinline__ void block_note_range_last(const int64_t *__restrict__ off, int n_notes, int64_t g0, int64_t gl,
                                                      int *s_pair, int &lo, int &hi)
{
    if (threadIdx.x < WAVE) {
        const int lane = threadIdx.x;
        auto key = [&](int k) { return off[k]; };
        const int a = wave_find(n_notes, g0, lane, key);
        const int b = wave_find(n_notes, gl, lane, key);
        if (lane == 0) { s_pair[0] = a; s_pair[1] = b; }
    }
    __syncthreads();
    lo = __builtin_amdgcn_readfirstlane(s_pair[0]);
    hi = __builtin_amdgcn_readfirstlane(s_pair[1]);
}

__device__ __forceinline__ void block_note_range(const int64_t *__restrict__ off, int n_notes, int64_t g0, int64_t total,
                                                 int *s_pair, int &lo, int &hi)
{
    int64_t gl = g0 + blockDim.x - 1;
    if (gl > total - 1) gl = total - 1;
    block_note_range_last(off, n_notes, g0, gl, s_pair, lo, hi);
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS ops of one wave complete in issue order; this only stops the compiler reordering
    // across the exchange and waits for outstanding LDS traffic.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// write-once data: a store with the non-temporal hint (global_store ... nt) when `nt`
typedef float v4f_nt __attribute__((ext_vector_type(4)));
typedef float v2f_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_f4(float *p, float4 v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v4f_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_nt *>(p));
    else *reinterpret_cast<float4 *>(p) = v;
}
__device__ __forceinline__ void store_f2(float2 *p, float2 v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v2f_nt{v.x, v.y}, reinterpret_cast<v2f_nt *>(p));
    else *p = v;
}
__device__ __forceinline__ void store_f1(float *p, float v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ~1e-16-accurate reciprocal (v_rcp_f64 + two Newton steps), used where the reference divides but
// the quotient only positions or scales a continuous interpolant (the fp32 result is unaffected
// except in knife-edge roundings)
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// atomic max on a non-negative float through its bit pattern
__device__ __forceinline__ void atomic_max_pos(float *addr, float v)
{
    atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// ---- kernel launchers implemented in the .hip files ---------------------------------------
int launch_rfft_frames(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off,
                       int n_notes, int64_t total_frames, float2 *S, int ldc, hipStream_t st);
int launch_irfft_frames(goofer_ctx *ctx, const float2 *S, int ldc, int64_t total_frames, float *frames, hipStream_t st);
int launch_ola_gather(goofer_ctx *ctx, const float *frames, const int64_t *sample_off, const int64_t *frame_off,
                      int n_notes, int64_t total_samples, float *y, const float *scale_per_note, hipStream_t st);
