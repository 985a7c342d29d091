// The 8x8 transpose between a lane's register index and bits 3..5 of its lane id, without LDS: v_permlane32_swap (lane bit 5),
// v_permlane16_swap (bit 4), DPP row_ror:8 under bank masks (bit 3) — 32 vector instructions for eight float2 per lane.
// Checks the data movement the walkers' first FFT exchange wants (fft_core.h: transpose_reg_lanehi) on the device.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_transpose scripts/micro/lane_transpose.hip && /tmp/lane_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void swap_hi32(float &a, float &b)   // a @ lanes 32..63 <-> b @ lanes 0..31
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_hi16(float &a, float &b)   // a @ odd rows of 16 <-> b @ even rows
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_hi8(float &a, float &b)    // a @ lanes with bit 3 set <-> b @ lanes with bit 3 clear
{
    const int ai = __float_as_int(a), bi = __float_as_int(b);
    const int na = __builtin_amdgcn_update_dpp(ai, bi, 0x128, 0xf, 0xc, false);   // row_ror:8 into lanes 8..15 of each row
    const int nb = __builtin_amdgcn_update_dpp(bi, ai, 0x128, 0xf, 0x3, false);   // ... into lanes 0..7
    a = __int_as_float(na);
    b = __int_as_float(nb);
}

__global__ void k(const float *in, float *out)
{
    const int lane = threadIdx.x;
    float x[8];
    for (int t = 0; t < 8; ++t) x[t] = in[lane * 8 + t];
    for (int t = 0; t < 4; ++t) swap_hi32(x[t], x[t + 4]);
    for (int t : {0, 1, 4, 5}) swap_hi16(x[t], x[t + 2]);
    for (int t : {0, 2, 4, 6}) swap_hi8(x[t], x[t + 1]);
    for (int t = 0; t < 8; ++t) out[lane * 8 + t] = x[t];
}

int main()
{
    std::vector<float> h(512), o(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    float *d_in, *d_out;
    hipMalloc(&d_in, 2048);
    hipMalloc(&d_out, 2048);
    hipMemcpy(d_in, h.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(o.data(), d_out, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < 8; ++t) {
            const int src_lane = (t << 3) | (lane & 7), src_reg = lane >> 3;   // x'[t] of lane (h, lo) = x[h] of lane (t, lo)
            if (o[lane * 8 + t] != (float)(src_lane * 8 + src_reg)) {
                if (bad < 8) printf("lane %d reg %d: got %g want %d\n", lane, t, o[lane * 8 + t], src_lane * 8 + src_reg);
                ++bad;
            }
        }
    printf(bad ? "FAILED: %d wrong\n" : "transpose ok (%d wrong)\n", bad);
    return bad != 0;
}
