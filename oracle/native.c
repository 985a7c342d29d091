/*
 * oracle/native.c — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Scalar C restatement of the three sequential loops the reference JIT-compiles with numba:
 *   ola_f32          <- GOOFER.py:372-390   (_overlap_add)
 *   pulse_train_f32  <- GOOFER.py:473-554   (pulse_train_numba; numba types the scalars as fp64)
 *   onepole_cascade  <- SillySampler.py:142-174 (_dynamic_butter_filter_core recurrences)
 * Built by oracle/build.py with plain `gcc -O2` (no -ffast-math: every operation keeps the
 * source order, true division, round-half-even), loaded through ctypes by oracle/goofer_ref.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* frames is [n_fft][n_frames] C-order like the reference; y/wsum have n_fft + hop*(n_frames-1). */
void ola_f32(const float *frames, const float *window, int64_t n_fft, int64_t n_frames,
             int64_t hop, float *y)
{
    int64_t len = n_fft + hop * (n_frames - 1);
    float *wsum = (float *)calloc((size_t)len, sizeof(float));
    memset(y, 0, (size_t)len * sizeof(float));
    for (int64_t i = 0; i < n_frames; ++i) {
        int64_t start = i * hop;
        for (int64_t j = 0; j < n_fft; ++j) {
            float v = frames[j * n_frames + i] * window[j];
            y[start + j] += v;
            wsum[start + j] += window[j] * window[j];
        }
    }
    for (int64_t i = 0; i < len; ++i)
        if (wsum[i] > 1e-9f) y[i] /= wsum[i];
    free(wsum);
}

/* Same arithmetic, frames given as [n_frames][n_fft] (the build's device layout). */
void ola_f32_rows(const float *frames, const float *window, int64_t n_fft, int64_t n_frames,
                  int64_t hop, float *y)
{
    int64_t len = n_fft + hop * (n_frames - 1);
    float *wsum = (float *)calloc((size_t)len, sizeof(float));
    memset(y, 0, (size_t)len * sizeof(float));
    for (int64_t i = 0; i < n_frames; ++i) {
        int64_t start = i * hop;
        for (int64_t j = 0; j < n_fft; ++j) {
            float v = frames[i * n_fft + j] * window[j];
            y[start + j] += v;
            wsum[start + j] += window[j] * window[j];
        }
    }
    for (int64_t i = 0; i < len; ++i)
        if (wsum[i] > 1e-9f) y[i] /= wsum[i];
    free(wsum);
}

#define PT_CACHE 5
#define PT_MAXLEN 8192

/* Optionally reports onsets (sample index, T0) so tests can check the integer path bit-exactly. */
int64_t pulse_train_f32(const float *f0, int64_t n, double sr, double Ra, double Rg, double Rk,
                        float *pulse, int64_t *onset_idx, int64_t *onset_T0, int64_t onset_cap)
{
    static const double PI = 3.141592653589793;
    double total_phase = 0.0, next_k = 1.0, last_valid = 160.0;
    int64_t cache_T0[PT_CACHE] = {0, 0, 0, 0, 0};
    int cache_len = 0;
    float *bank = (float *)calloc((size_t)PT_CACHE * PT_MAXLEN, sizeof(float));
    int64_t n_on = 0;
    memset(pulse, 0, (size_t)n * sizeof(float));

    for (int64_t i = 0; i < n; ++i) {
        float f0i = f0[i];
        if (f0i > 1e-6f) last_valid = (double)f0i;
        total_phase += (double)f0i / sr;
        while (total_phase >= next_k) {
            double T = 1.0 / (last_valid > 1e-6 ? last_valid : 1e-6);
            int64_t T0 = (int64_t)nearbyint(sr * T);
            if (T0 < 3) T0 = 3;
            if (T0 > PT_MAXLEN) T0 = PT_MAXLEN;
            int found = -1;
            for (int c = 0; c < cache_len; ++c)
                if (cache_T0[c] == T0) { found = c; break; }
            if (found < 0) {
                int slot = cache_len < PT_CACHE ? cache_len : 0;
                float *buf = bank + (size_t)slot * PT_MAXLEN;
                double Tp = Ra * T, Tc = Tp + Rk * (T - Tp);
                for (int64_t j = 0; j < T0; ++j) {
                    double ti = ((double)j * T) / (double)T0;
                    double v;
                    if (ti < Tp) {
                        double s = sin(PI * ti / (2.0 * Tp + 1e-12));
                        v = s * s;
                    } else if (ti < Tc) {
                        double tau = (ti - Tp) / (Tc - Tp + 1e-12);
                        v = exp(-Rg * tau) * cos(PI * tau / 2.0);
                    } else {
                        v = 0.0;
                    }
                    buf[j] = (float)v;
                }
                double m = 0.0;
                for (int64_t j = 0; j < T0; ++j) {
                    double a = fabs((double)buf[j]);
                    if (a > m) m = a;
                }
                if (m > 0.0)
                    for (int64_t j = 0; j < T0; ++j) buf[j] = (float)((double)buf[j] / m);
                cache_T0[slot] = T0;
                if (cache_len < PT_CACHE) cache_len++;
                found = slot;
            }
            int64_t end = i + cache_T0[found];
            if (end > n) end = n;
            const float *src = bank + (size_t)found * PT_MAXLEN;
            for (int64_t j = i, k = 0; j < end; ++j, ++k) pulse[j] += src[k];
            if (onset_idx && n_on < onset_cap) { onset_idx[n_on] = i; onset_T0[n_on] = T0; }
            n_on++;
            next_k += 1.0;
        }
    }
    free(bank);
    return n_on;
}

/* alpha[] per sample, `order` passes in place; highpass != 0 selects the HP recurrence. */
void onepole_cascade(float *y, const float *alpha, int64_t n, int order, int highpass)
{
    if (order < 1) order = 1;
    for (int p = 0; p < order; ++p) {
        float yp = 0.0f;
        if (!highpass) {
            for (int64_t i = 0; i < n; ++i) {
                float xp = y[i];
                yp = yp + alpha[i] * (xp - yp);
                y[i] = yp;
            }
        } else {
            float prev = n > 0 ? y[0] : 0.0f;
            for (int64_t i = 0; i < n; ++i) {
                float xp = y[i];
                yp = alpha[i] * (yp + xp - prev);
                y[i] = yp;
                prev = xp;
            }
        }
    }
}
