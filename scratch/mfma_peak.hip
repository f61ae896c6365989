// micro-benchmark: fp32 MFMA issue rate with dependent / independent accumulator chains and 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[0] = s;
}
template <int NACC> void run(int wgs_per_cu, int threads) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * threads / 64, mfma = waves * iters * 16;
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("NACC=%d wgs/cu=%d threads=%d waves/SIMD=%.2f : %.3f ms  %.1f TF/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", NACC, wgs_per_cu, threads,
           waves / 1024, ms, tf, ms * 1e-3 * 2.4e9 / (mfma / 1024));
    hipFree(d);
}
int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256);
    run<1>(2, 256); run<2>(2, 256); run<1>(3, 256); run<1>(4, 256); run<4>(2, 256);
    run<1>(1, 64); run<1>(1, 128);
    return 0;
}
