// gf.stretch_feature (GOOFER.py:597-616) for gfx950: resample axis 0 of a [rows x cols] fp32 matrix (or a 1-D array,
// cols = 1) to R_out rows on normalised coordinates — np.interp(linspace(0, 1, R_out), linspace(0, 1, R_in), column),
// evaluated in fp64 like numpy and rounded to fp32.  Used by gf.synthesize's optional time stretch (GOOFER.py:1019-1067).
#include "common.h"

// np.linspace(0, 1, n)[i]
__device__ __forceinline__ double lin01(int64_t i, int64_t n, double step)
{
    if (n <= 1) return 0.0;
    if (i >= n - 1) return 1.0;
    return (double)i * step;
}

__global__ __launch_bounds__(256) void k_lerp_axis0(const float *__restrict__ in, int64_t ld_in, int64_t R_in, float *__restrict__ out,
                                                    int64_t ld_out, int64_t R_out, int n_cols, double step_in, double step_out)
{
    const int64_t r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R_out || c >= n_cols) return;
    if (R_in == 1) {                                          // one knot: constant                GOOFER.py:183-191
        out[r * ld_out + c] = in[c];
        return;
    }
    const double x = lin01(r, R_out, step_out);
    int64_t j = (int64_t)(x * (double)(R_in - 1));
    if (j > R_in - 1) j = R_in - 1;
    if (j < 0) j = 0;
    while (j + 1 <= R_in - 1 && lin01(j + 1, R_in, step_in) <= x) ++j;
    while (j > 0 && lin01(j, R_in, step_in) > x) --j;
    double v;
    if (j >= R_in - 1) {
        v = (double)in[(R_in - 1) * ld_in + c];
    } else {
        const double x0 = lin01(j, R_in, step_in), x1 = lin01(j + 1, R_in, step_in);
        const double y0 = (double)in[j * ld_in + c], y1 = (double)in[(j + 1) * ld_in + c];
        v = x == x0 ? y0 : ((y1 - y0) / (x1 - x0)) * (x - x0) + y0;      // np.interp's slope form
    }
    out[r * ld_out + c] = (float)v;
}

int launch_lerp_axis0(goofer_ctx *ctx, const float *in, int64_t ld_in, int64_t R_in, float *out, int64_t ld_out, int64_t R_out,
                      int n_cols, hipStream_t st)
{
    if (R_out <= 0 || n_cols <= 0) return GOOFER_OK;
    if (R_in <= 0) return goofer_fail(ctx, GOOFER_EINVAL, "cannot stretch an empty feature");
    const double step_in = R_in > 1 ? 1.0 / (double)(R_in - 1) : 0.0, step_out = R_out > 1 ? 1.0 / (double)(R_out - 1) : 0.0;
    if (R_out > 65535) return goofer_fail(ctx, GOOFER_EINVAL, "more than 65535 output rows: resample in pieces (1-D arrays have their own entry)");
    dim3 grid((unsigned)((n_cols + 255) / 256), (unsigned)R_out);
    hipLaunchKernelGGL(k_lerp_axis0, grid, dim3(256), 0, st, in, ld_in, R_in, out, ld_out, R_out, n_cols, step_in, step_out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}

// 1-D flavour: thread per output element
__global__ __launch_bounds__(256) void k_lerp_1d(const float *__restrict__ in, int64_t R_in, float *__restrict__ out, int64_t R_out,
                                                 double step_in, double step_out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R_out) return;
    if (R_in == 1) {
        out[r] = in[0];
        return;
    }
    const double x = lin01(r, R_out, step_out);
    int64_t j = (int64_t)(x * (double)(R_in - 1));
    if (j > R_in - 1) j = R_in - 1;
    if (j < 0) j = 0;
    while (j + 1 <= R_in - 1 && lin01(j + 1, R_in, step_in) <= x) ++j;
    while (j > 0 && lin01(j, R_in, step_in) > x) --j;
    double v;
    if (j >= R_in - 1) {
        v = (double)in[R_in - 1];
    } else {
        const double x0 = lin01(j, R_in, step_in), x1 = lin01(j + 1, R_in, step_in);
        const double y0 = (double)in[j], y1 = (double)in[j + 1];
        v = x == x0 ? y0 : ((y1 - y0) / (x1 - x0)) * (x - x0) + y0;
    }
    out[r] = (float)v;
}

int launch_lerp_1d(goofer_ctx *ctx, const float *in, int64_t R_in, float *out, int64_t R_out, hipStream_t st)
{
    if (R_out <= 0) return GOOFER_OK;
    if (R_in <= 0) return goofer_fail(ctx, GOOFER_EINVAL, "cannot stretch an empty feature");
    const double step_in = R_in > 1 ? 1.0 / (double)(R_in - 1) : 0.0, step_out = R_out > 1 ? 1.0 / (double)(R_out - 1) : 0.0;
    hipLaunchKernelGGL(k_lerp_1d, dim3((unsigned)((R_out + 255) / 256)), dim3(256), 0, st, in, R_in, out, R_out, step_in, step_out);
    LAUNCH_CHECK(ctx);
    return GOOFER_OK;
}
