// Micro-benchmark: what a row-structured kernel (one [frames x bins] fp32 matrix in, one or two out, 2 KB rows) can reach on
// gfx950, by access shape.  hipcc -O3 --offload-arch=gfx950 scripts/micro/rowcopy.hip -o /tmp/rowcopy && /tmp/rowcopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int B = 513, LD = 516;

// A/B: wave per row, dword per lane, NOUT outputs
template <int NOUT, int RPW>
__global__ __launch_bounds__(256) void k_dword(const float *__restrict__ in, float *__restrict__ o0, float *__restrict__ o1, int64_t rows)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t r = r0 + q;
        if (r >= rows) return;
        const float *s = in + r * LD;
        float v[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = s[(c < 8 || lane == 0) ? c * 64 + lane : 512];
#pragma unroll
        for (int c = 0; c < 9; ++c)
            if (c < 8 || lane == 0) {
                o0[r * LD + c * 64 + lane] = v[c];
                if (NOUT > 1) o1[r * LD + c * 64 + lane] = v[c] * 1.5f;
            }
    }
}

// C: wave per row, 16 bytes per lane (two instructions cover the row: lanes x float4 x 2 = 512 floats, + 1 tail float)
template <int NOUT, int RPW>
__global__ __launch_bounds__(256) void k_vec4(const float *__restrict__ in, float *__restrict__ o0, float *__restrict__ o1, int64_t rows)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int64_t r = r0 + q;
        if (r >= rows) return;
        const float *s = in + r * LD;
        const float4 a = *reinterpret_cast<const float4 *>(s + 4 * lane);
        const float4 b = *reinterpret_cast<const float4 *>(s + 256 + 4 * lane);
        const float t = s[512];
        *reinterpret_cast<float4 *>(o0 + r * LD + 4 * lane) = a;
        *reinterpret_cast<float4 *>(o0 + r * LD + 256 + 4 * lane) = b;
        if (lane == 0) o0[r * LD + 512] = t;
        if (NOUT > 1) {
            *reinterpret_cast<float4 *>(o1 + r * LD + 4 * lane) = make_float4(a.x * 1.5f, a.y * 1.5f, a.z * 1.5f, a.w * 1.5f);
            *reinterpret_cast<float4 *>(o1 + r * LD + 256 + 4 * lane) = make_float4(b.x * 1.5f, b.y * 1.5f, b.z * 1.5f, b.w * 1.5f);
            if (lane == 0) o1[r * LD + 512] = t * 1.5f;
        }
    }
}

// E: flat streaming copy, float4, grid-stride
template <int NOUT>
__global__ __launch_bounds__(256) void k_flat(const float4 *__restrict__ in, float4 *__restrict__ o0, float4 *__restrict__ o1, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = in[i];
        o0[i] = a;
        if (NOUT > 1) o1[i] = make_float4(a.x * 1.5f, a.y * 1.5f, a.z * 1.5f, a.w * 1.5f);
    }
}

// F: write only (fill), float4
__global__ __launch_bounds__(256) void k_fill(float4 *__restrict__ o0, int64_t n4, float v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) o0[i] = make_float4(v, v, v, v);
}
// G: read only (sum), float4
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ in, float *__restrict__ out, int64_t n4)
{
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) { const float4 a = in[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static float time_ms(F f, int reps = 10)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const int64_t rows = 194560;
    const size_t bytes = (size_t)rows * LD * 4;
    float *in, *o0, *o1;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&o0, bytes)); CK(hipMalloc(&o1, bytes));
    CK(hipMemset(in, 0, bytes));
    const double gb = bytes / 1e9;
    const int64_t n4 = bytes / 16;
    auto run = [&](const char *name, int nbuf, auto f) {
        const float ms = time_ms(f);
        printf("%-44s %.3f ms  %.2f TB/s (%d x %.2f GB)\n", name, ms, nbuf * gb / ms, nbuf, gb);
    };
    const dim3 g1((unsigned)((rows + 3) / 4)), g4((unsigned)((rows + 15) / 16)), g16((unsigned)((rows + 63) / 64)), blk(256);
    run("dword, wave per row, 1 out", 2, [&] { k_dword<1, 1><<<g1, blk>>>(in, o0, o1, rows); });
    run("dword, wave per row, 2 out", 3, [&] { k_dword<2, 1><<<g1, blk>>>(in, o0, o1, rows); });
    run("dword, 4 rows per wave, 2 out", 3, [&] { k_dword<2, 4><<<g4, blk>>>(in, o0, o1, rows); });
    run("dword, 16 rows per wave, 2 out", 3, [&] { k_dword<2, 16><<<g16, blk>>>(in, o0, o1, rows); });
    run("float4, wave per row, 1 out", 2, [&] { k_vec4<1, 1><<<g1, blk>>>(in, o0, o1, rows); });
    run("float4, wave per row, 2 out", 3, [&] { k_vec4<2, 1><<<g1, blk>>>(in, o0, o1, rows); });
    run("float4, 4 rows per wave, 2 out", 3, [&] { k_vec4<2, 4><<<g4, blk>>>(in, o0, o1, rows); });
    run("float4, 16 rows per wave, 2 out", 3, [&] { k_vec4<2, 16><<<g16, blk>>>(in, o0, o1, rows); });
    for (int g : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, sizeof nm, "flat float4 grid %d, 1 out", g);
        run(nm, 2, [&] { k_flat<1><<<dim3(g), blk>>>((const float4 *)in, (float4 *)o0, (float4 *)o1, n4); });
        snprintf(nm, sizeof nm, "flat float4 grid %d, 2 out", g);
        run(nm, 3, [&] { k_flat<2><<<dim3(g), blk>>>((const float4 *)in, (float4 *)o0, (float4 *)o1, n4); });
    }
    run("fill float4 grid 8192", 1, [&] { k_fill<<<dim3(8192), blk>>>((float4 *)o0, n4, 1.0f); });
    run("read float4 grid 8192", 1, [&] { k_read<<<dim3(8192), blk>>>((const float4 *)in, o1, n4); });
    return 0;
}
