// how long does a GEMM-shaped launch of pure MFMA work take as a function of the grid size?  (tile quantisation / dispatch / clock)
// every wave: `iters` x 16 dependent v_mfma_f32_32x32x2_f32 on one accumulator (what a 64x64-tile conv workgroup does per K chunk),
// 256-thread workgroups with 33 KB of static LDS like k_conv_gemm<2,2,1,1,4,*,32>.  operands: constants or per-lane pseudo-random values
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0, int rnd) {
    __shared__ float lds[33 * 256];
    lds[threadIdx.x] = a0;
    __syncthreads();
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0, b = b0;
    if (rnd) { unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u; a = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f; b = (float)((h * 40503u) >> 8) * (1.f / 16777216.f) - 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); if (rnd) { a = -a; b = b * 1.0001f; } }
    }
    float s = lds[(threadIdx.x + 1) & 255];
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256, 512, 600, 768, 1024, 1200, 1536, 2048};
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int iters : {18, 72}) {
            for (int g : grids) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, iters, 1.f, 2.f, rnd);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                const int rep = 20;
                for (int w = 0; w < rep; ++w) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, iters, 1.f, 2.f, rnd);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms / rep * 1e3, mfma = (double)g * 4 * iters * 16;
                const double rounds = (double)g * 4 / 1024;      // waves per SIMD
                printf("%s iters=%2d grid=%4d (%.2f waves/SIMD): %7.1f us  %6.1f TF/s  ideal@2.4GHz(ceil rounds) %.1f us\n", rnd ? "random  " : "constant", iters, g, rounds, us,
                       mfma * 4096 / (us * 1e-6) / 1e12, (double)((g * 4 + 1023) / 1024) * iters * 16 * 64 / 2.4e3);
            }
        }
    return 0;
}
