#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(double *out, const double *in, int iters)
{
    double a = in[0], b = in[1], p = in[2], q = in[3];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) p = p + a;
            if (MODE == 1) p = __builtin_fma(p, 1.0, a);
            if (MODE == 2) { p = p + a; q = q + b; }
            if (MODE == 3) { float pf = (float)p; pf = pf + (float)a; p = pf; }
        }
        asm volatile("" : "+v"(p), "+v"(q));
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = p + q;
}
template <int MODE> void run(const char *name, int blocks, int threads)
{
    double *in, *out; hipMalloc(&in, 64); hipMalloc(&out, 8 * blocks * threads);
    double h[4] = {1e-3, 2e-3, 0.5, 0.25}; hipMemcpy(in, h, 32, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, in, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, in, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d x %3d: %.3f ms -> %.1f ns per step (16/iter) => %.1f cycles@2.4GHz per op-step\n", name, blocks, threads, ms,
           ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.4);
}
int main()
{
    run<0>("add_f64 chain", 1, 64); run<1>("fma_f64 chain", 1, 64); run<2>("2 indep add chains", 1, 64); run<3>("f32 add chain (cvt)", 1, 64);
    run<0>("add_f64 chain", 256, 256); run<0>("add_f64 chain", 1024, 64); run<0>("add_f64 chain", 1024, 256); run<0>("add_f64 chain", 2048, 256);
    return 0;
}
