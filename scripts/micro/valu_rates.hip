// Micro-benchmark: SIMD cycles per wave-instruction of the vector instructions the walkers are made of (gfx950).
// Each kernel runs ITER iterations of 16 independent instances of one instruction (independent destinations, so the
// dependent-issue latency does not show), with W waves resident per SIMD; s_memtime stamps give shader cycles.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define ITER 256

#define BODY16(INS)                                                                                                     \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

template <int OP>
__global__ __launch_bounds__(512) void k_rate(uint64_t *out, float seedf)
{
    float a[16], b[16];
    double d[8], e[8];
    uint32_t u[16];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 16; ++i) { a[i] = seedf + i + lane; b[i] = seedf * 0.5f + i; u[i] = (uint32_t)(lane * 77 + i * 13 + 1); }
    for (int i = 0; i < 8; ++i) { d[i] = seedf + i; e[i] = seedf * 0.25 + i + lane; }
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[8], q[8];
    for (int i = 0; i < 8; ++i) { p[i] = v2{a[2 * i], a[2 * i + 1]}; q[i] = v2{b[2 * i], b[2 * i + 1]}; }
    __shared__ float lds[512 * 4];
    lds[threadIdx.x] = seedf;
    __syncthreads();
    uint32_t ldsaddr = (uint32_t)(threadIdx.x * 8) & 2047u;
    int sidx = lane & 31;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {
#define I(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[n]) : "v"(b[n]), "v"(b[(n + 1) & 15]));
            BODY16(I)
#undef I
        } else if (OP == 1) {
#define I(n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[n & 7]) : "v"(q[n & 7]), "v"(q[(n + 1) & 7]));
            BODY16(I)
#undef I
        } else if (OP == 2) {
#define I(n) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[n & 7]) : "v"(e[n & 7]), "v"(e[(n + 1) & 7]));
            BODY16(I)
#undef I
        } else if (OP == 3) {
#define I(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[n & 7]) : "v"(a[n]));
            BODY16(I)
#undef I
        } else if (OP == 4) {
#define I(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&d[n & 7]) : "v"(u[n]), "v"(u[(n + 1) & 15]) : "vcc");
            BODY16(I)
#undef I
        } else if (OP == 5) {
#define I(n) asm volatile("v_sin_f32 %0, %1" : "=v"(a[n]) : "v"(b[n]));
            BODY16(I)
#undef I
        } else if (OP == 6) {
#define I(n) asm volatile("v_exp_f32 %0, %1" : "=v"(a[n]) : "v"(b[n]));
            BODY16(I)
#undef I
        } else if (OP == 7) {
#define I(n) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[n]) : "v"(b[n]), "v"(b[(n + 1) & 15]) : "vcc");
            BODY16(I)
#undef I
        } else if (OP == 8) {
#define I(n) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[n]) : "s20");
            BODY16(I)
#undef I
        } else if (OP == 9) {
#define I(n) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u[n]) : "v"(u[(n + 1) & 15]), "v"(u[(n + 2) & 15]));
            BODY16(I)
#undef I
        } else if (OP == 10) {
#define I(n) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[n & 7]) : "v"(e[n & 7]), "v"(e[(n + 1) & 7]));
            BODY16(I)
#undef I
        } else if (OP == 11) {
#define I(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[n]) : "v"(e[n & 7]));
            BODY16(I)
#undef I
        } else if (OP == 12) {
#define I(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[n]) : "v"(e[n & 7]));
            BODY16(I)
#undef I
        } else if (OP == 13) {
#define I(n) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[n & 7]) : "v"(q[n & 7]), "v"(q[(n + 1) & 7]));
            BODY16(I)
#undef I
        } else if (OP == 14) {
#define I(n) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[n & 7]) : "v"(q[n & 7]), "v"(q[(n + 1) & 7]));
            BODY16(I)
#undef I
        } else if (OP == 15) {
#define I(n) asm volatile("ds_read_b64 %0, %1" : "=v"(d[n & 7]) : "v"(ldsaddr) : "memory");
            BODY16(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 16) {
#define I(n) asm volatile("ds_write_b64 %0, %1" : : "v"(ldsaddr), "v"(d[n & 7]) : "memory");
            BODY16(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 17) {
#define I(n) asm volatile("v_mov_b32 %0, %1" : "=v"(a[n]) : "v"(b[n]));
            BODY16(I)
#undef I
        } else if (OP == 18) {
#define I(n) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
            BODY16(I)
#undef I
        } else if (OP == 19) {
#define I(n) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[n]) : "v"(b[n]));
            BODY16(I)
#undef I
        } else if (OP == 20) {
#define I(n) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u[n]) : "v"(u[(n + 1) & 15]), "v"(u[(n + 2) & 15]));
            BODY16(I)
#undef I
        } else if (OP == 21) {
#define I(n) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3" : "+v"(a[n]), "+v"(b[n]) : "v"(seedf), "v"(seedf));
            BODY16(I)
#undef I
        } else if (OP == 22) {
#define I(n) asm volatile("ds_read_b32 %0, %1" : "=v"(a[n]) : "v"(ldsaddr) : "memory");
            BODY16(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 23) {
#define I(n) asm volatile("ds_read_b128 %0, %1" : "=v"(*(float4 *)&d[(n & 3) * 2]) : "v"(ldsaddr * 2) : "memory");
            BODY16(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += a[i] + b[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) acc += (float)d[i] + (float)e[i] + p[i].x + p[i].y + q[i].x;
    if (acc == 12345.678f) out[1000000] = 1;          // keep everything alive
    if (lane == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static void run(const char *name, uint64_t *d_out, int waves_per_simd)
{
    const int threads = 256 * (waves_per_simd >= 2 ? 2 : 1);          // 4 or 8 waves per workgroup
    const int blocks_per_cu = waves_per_simd >= 2 ? waves_per_simd / 2 : 1;
    const int blocks = 256 * blocks_per_cu;
    const int nw = blocks * threads / 64;
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nw);
    hipMemcpy(h.data(), d_out, nw * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[nw / 2];
    // per SIMD: waves_per_simd waves each issued ITER*16 instructions in `med` cycles
    printf("%-16s W=%d  %7.2f cycles per wave-instruction (wave view)   %6.2f SIMD cycles per instruction\n", name, waves_per_simd,
           med / (ITER * 16.0), med / (ITER * 16.0 * waves_per_simd));
}

#define RUN(OP, NAME) run<OP>(NAME, d_out, 1); run<OP>(NAME, d_out, 2); run<OP>(NAME, d_out, 4);

int main()
{
    uint64_t *d_out;
    hipMalloc(&d_out, sizeof(uint64_t) * 1100000);
    RUN(0, "v_fma_f32") RUN(21, "2x v_fma_f32") RUN(1, "v_pk_fma_f32") RUN(13, "v_pk_mul_f32") RUN(14, "v_pk_add_f32") RUN(2, "v_fma_f64") RUN(10, "v_add_f64")
    RUN(3, "v_cvt_f64_f32") RUN(11, "v_cvt_f32_f64") RUN(12, "v_cvt_i32_f64") RUN(4, "v_mad_u64_u32") RUN(9, "v_mul_lo_u32") RUN(20, "v_mul_u32_u24")
    RUN(5, "v_sin_f32") RUN(6, "v_exp_f32") RUN(19, "v_rcp_f32") RUN(7, "v_cndmask_b32") RUN(8, "v_readlane_b32") RUN(17, "v_mov_b32") RUN(18, "s_add_u32")
    RUN(22, "ds_read_b32") RUN(15, "ds_read_b64") RUN(23, "ds_read_b128") RUN(16, "ds_write_b64")
    return 0;
}
