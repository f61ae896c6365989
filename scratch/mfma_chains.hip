// Sustained fp32 MFMA rate of this part as a function of (a) independent accumulator chains per wave, (b) waves per SIMD, (c) MFMA shape.
// Round 2's mfma_grid.hip issued ONE dependent chain per wave; the microarchitecture guide's 155 TF/s figure is for independent
// accumulators.  Every wave: `iters` x 16 MFMAs spread round-robin over NCH accumulators (same MFMA count for every NCH), no memory
// traffic; grid = waves_per_simd x 256 workgroups of 256 threads.  Also stamps the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chains mfma_chains.hip && ./mfma_chains
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NCH, int SHAPE>   // SHAPE 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters) {
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float a = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f, b = (float)((h * 40503u) >> 8) * (1.f / 16777216.f) - 0.5f;
    f16v acc[NCH];
    f4v acc4[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[c][r] = 0.f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NCH; ++u) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (SHAPE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
                else acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[c], 0, 0, 0);
            }
            a = -a;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        for (int r = 0; r < 16; ++r) s += acc[c][r];
        for (int r = 0; r < 4; ++r) s += acc4[c][r];
    }
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NCH, int SHAPE>
static void run(float* d, unsigned long long* st, int wps, int iters) {
    const int g = 256 * wps;
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k<NCH, SHAPE>), dim3(g), dim3(256), 0, 0, d, st, iters);      // ~0.1 s of load first
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int rep = 50;
    for (int w = 0; w < rep; ++w) hipLaunchKernelGGL((k<NCH, SHAPE>), dim3(g), dim3(256), 0, 0, d, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hs[512];
    hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
    double clk = 0; int n = 0;
    for (int i = 0; i < 256; ++i) if (hs[2 * i + 1]) { clk += (double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0; ++n; }   // MHz: memrealtime ticks at 100 MHz
    const double us = ms / rep * 1e3, nm = (double)g * 4 * iters * 16, flop = nm * (SHAPE == 0 ? 4096.0 : 2048.0);
    const double cyc = (double)hs[0] / (iters * 16.0);
    printf("%s chains=%d waves/SIMD=%d iters=%4d: %8.1f us/launch %6.1f TF/s | in-kernel clock %.0f MHz, %.1f cycles per MFMA per wave (block 0)\n",
           SHAPE == 0 ? "32x32x2" : "16x16x4", NCH, wps, iters, us, flop / (us * 1e-6) / 1e12, clk / (n ? n : 1), cyc);
}

int main() {
    float* d; hipMalloc(&d, 4);
    unsigned long long* st; hipMalloc(&st, 512 * 8); hipMemset(st, 0, 512 * 8);
    for (int wps : {1, 2, 4}) {
        const int it = 512 / wps;
        run<1, 0>(d, st, wps, it); run<2, 0>(d, st, wps, it); run<4, 0>(d, st, wps, it);
        run<1, 1>(d, st, wps, 2 * it); run<2, 1>(d, st, wps, 2 * it); run<4, 1>(d, st, wps, 2 * it);
    }
    return 0;
}
