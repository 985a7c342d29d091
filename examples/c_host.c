/* Minimal C host for libgoofer_hip.so: no Python, no torch — the ABI takes raw device pointers and a hipStream_t.
 * Builds with:  gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/c_host.c -Iinclude -I/opt/rocm/include -Lgoofer_amd -lgoofer_hip \
 *                  -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/goofer_amd -Wl,-rpath,/opt/rocm/lib -o c_host
 * (tests/test_abi.py compiles and links it on every CPU run; it only executes on a box with an MI355X.)
 *
 * It does what gf.stft + gf.istft do for one note (GOOFER.py:355-413): framewise rFFT, inverse FFT with weighted
 * overlap-add, and checks the sqrt-Hann / 75 % overlap reconstruction. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "goofer_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GF(x) do { int r_ = (x); if (r_ != GOOFER_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, goofer_last_error(h)); return 3; } } while (0)

int main(void)
{
    const int sr = 44100, n_fft = 1024, hop = 256, n_bins = n_fft / 2 + 1, ldc = n_bins + 1;
    const int64_t n = 44100, T = 1 + n / hop;
    goofer_ctx *h = NULL;
    if (goofer_create(0, &h) != GOOFER_OK) { fprintf(stderr, "goofer_create failed (no MI355X visible?)\n"); return 1; }
    CHECK_GF(goofer_plan(h, sr, n_fft, hop));

    float *x = (float *)malloc(n * sizeof(float)), *y = (float *)malloc(n * sizeof(float));
    for (int64_t i = 0; i < n; ++i) x[i] = sinf(0.01f * (float)i) * 0.5f + 0.25f * sinf(0.37f * (float)i);
    const int64_t s_off[2] = {0, n}, f_off[2] = {0, T};

    float *d_x, *d_y, *d_S;
    int64_t *d_s, *d_f;
    CHECK_HIP(hipMalloc((void **)&d_x, n * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_y, n * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_S, (size_t)T * ldc * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_s, sizeof s_off));
    CHECK_HIP(hipMalloc((void **)&d_f, sizeof f_off));
    CHECK_HIP(hipMemcpy(d_x, x, n * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_s, s_off, sizeof s_off, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_f, f_off, sizeof f_off, hipMemcpyHostToDevice));

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_GF(goofer_rfft_frames(h, d_x, d_s, d_f, 1, T, d_S, ldc, (void *)st));          /* gf.stft  */
    CHECK_GF(goofer_irfft_ola(h, d_S, ldc, d_s, d_f, 1, T, n, d_y, (void *)st));           /* gf.istft */
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(y, d_y, n * sizeof(float), hipMemcpyDeviceToHost));

    double worst = 0.0;
    for (int64_t i = 0; i < (int64_t)hop * (T - 1); ++i) worst = fmax(worst, fabs((double)y[i] - (double)x[i]));
    printf("%s: %lld samples, %lld frames, max |istft(stft(x)) - x| = %.3g\n", goofer_version(), (long long)n, (long long)T, worst);

    hipFree(d_x); hipFree(d_y); hipFree(d_S); hipFree(d_s); hipFree(d_f);
    hipStreamDestroy(st);
    goofer_destroy(h);
    free(x); free(y);
    return worst < 1e-4 ? 0 : 4;
}
