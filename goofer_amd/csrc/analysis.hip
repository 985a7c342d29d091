// Analysis half that does not need Praat (gfx950): spectral envelope of a wav and its mel-knot encoding.
//
//   k_mag_rows     |S| + 1e-8 of the STFT rows (fp32, like np.abs(complex64) + 1e-8)     GOOFER.py:945
//   k_gauss_rows64 Gaussian FIR along bins with fp64 output (sigma = 0.5 pre-blur)       GOOFER.py:100
//   k_knot_error   for one knot count K: max over probe frames and bins of
//                  |exp(lerp(log knots)) - env| / (env + 1e-8)                            GOOFER.py:112-121
//   k_knot_gather  log-envelope sampled at the knots' nearest bins -> fp16 [rows x K]     GOOFER.py:114-115, 126
// The sigma = 2 blur of GOOFER.py:946 is goofer_gauss_bins; the K search loop (9 candidates) is host logic.
#include <hip/hip_fp16.h>

#include "common.h"

constexpr int AN_ROWS = 4;

__global__ __launch_bounds__(256) void k_mag_rows(const float2 *__restrict__ S, int ldc, int64_t rows, int n_bins,
                                                  float *__restrict__ mag, int ld)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * AN_ROWS + wave;
    if (r >= rows) return;
    for (int b = lane; b < n_bins; b += WAVE) {
        float2 s = S[r * ldc + b];
        mag[r * ld + b] = hypotf(s.x, s.y) + 1e-8f;
    }
}

__global__ __launch_bounds__(256) void k_gauss_rows64(const float *__restrict__ in, int ld, double *__restrict__ out, int ld64,
                                                      int64_t rows, int n_bins, const double *__restrict__ taps, int radius)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_rows = reinterpret_cast<float *>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * AN_ROWS + wave;
    if (r >= rows) return;
    float *row = s_rows + wave * n_bins;
    for (int b = lane; b < n_bins; b += WAVE) row[b] = in[r * ld + b];
    wave_lds_sync();
    for (int b = lane; b < n_bins; b += WAVE) {
        double acc = 0.0;
        for (int j = 0; j <= 2 * radius; ++j) acc += taps[j] * (double)row[reflect_index(b + j - radius, n_bins)];
        out[r * ld64 + b] = acc;
    }
}

// err_bits: max relative error as the bit pattern of a non-negative double (order-preserving for atomicMax)
__global__ __launch_bounds__(256) void k_knot_error(const double *__restrict__ env2, int ld64, const int64_t *__restrict__ probe,
                                                    int n_probe, int n_bins, const int *__restrict__ knot_bin, int K,
                                                    const int *__restrict__ lerp_idx, const float *__restrict__ w0,
                                                    const float *__restrict__ w1, unsigned long long *__restrict__ err_bits)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_kv = reinterpret_cast<float *>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pi = blockIdx.x * AN_ROWS + wave;
    if (pi >= n_probe) return;
    const double *row = env2 + probe[pi] * (int64_t)ld64;
    float *kv = s_kv + wave * K;
    for (int k = lane; k < K; k += WAVE) kv[k] = (float)log(fmax(row[knot_bin[k]], 1e-8));
    wave_lds_sync();
    double worst = 0.0;
    for (int b = lane; b < n_bins; b += WAVE) {
        int i = lerp_idx[b];
        float rec = w0[b] * kv[i] + w1[b] * kv[i + 1];
        double e = row[b];
        double err = fabs((double)expf(rec) - e) / (e + 1e-8);
        worst = fmax(worst, err);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
    if (lane == 0) atomicMax(err_bits, (unsigned long long)__double_as_longlong(worst));
}

__global__ __launch_bounds__(256) void k_knot_gather(const double *__restrict__ env2, int ld64, int64_t rows,
                                                     const int *__restrict__ knot_bin, int K, __half *__restrict__ knots)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * K) return;
    int64_t r = g / K;
    int k = (int)(g - r * K);
    float v = (float)log(fmax(env2[r * ld64 + knot_bin[k]], 1e-8));     // log in fp64, cast to fp32 (DCOMPUTE)
    knots[g] = __float2half(v);                                         // then to fp16 (DSTORAGE)
}

int launch_mag_rows(goofer_ctx *ctx, const float2 *S, int ldc, int64_t rows, int n_bins, float *mag, int ld, hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_mag_rows, dim3((unsigned)((rows + AN_ROWS - 1) / AN_ROWS)), dim3(256), 0, st, S, ldc, rows, n_bins, mag, ld);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_gauss_rows64(goofer_ctx *ctx, const float *in, int ld, double *out, int ld64, int64_t rows, int n_bins,
                        const double *d_taps, int radius, hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_gauss_rows64, dim3((unsigned)((rows + AN_ROWS - 1) / AN_ROWS)), dim3(256), sizeof(float) * AN_ROWS * n_bins,
                       st, in, ld, out, ld64, rows, n_bins, d_taps, radius);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_knot_error(goofer_ctx *ctx, const double *env2, int ld64, const int64_t *probe, int n_probe, int n_bins,
                      const int *knot_bin, int K, const int *lerp_idx, const float *w0, const float *w1,
                      unsigned long long *err_bits, hipStream_t st)
{
    if (n_probe <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_knot_error, dim3((n_probe + AN_ROWS - 1) / AN_ROWS), dim3(256), sizeof(float) * AN_ROWS * K, st, env2, ld64,
                       probe, n_probe, n_bins, knot_bin, K, lerp_idx, w0, w1, err_bits);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

int launch_knot_gather(goofer_ctx *ctx, const double *env2, int ld64, int64_t rows, const int *knot_bin, int K, uint16_t *knots,
                       hipStream_t st)
{
    if (rows <= 0) return GOOFER_OK;
    hipLaunchKernelGGL(k_knot_gather, dim3((unsigned)((rows * K + 255) / 256)), dim3(256), 0, st, env2, ld64, rows, knot_bin, K,
                       reinterpret_cast<__half *>(knots));
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
