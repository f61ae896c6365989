// microbenchmark of the backward-weight slab reduction (conv.hip k_wgrad_reduce) and variants.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ZL, int MODE>   // MODE 0: as shipped; 1: contiguous write; 2: no LDS stage (ZL lanes only, skip write of others)
__global__ __launch_bounds__(64 * ZL) void k_red(const float* __restrict__ slabs, int S, int Mpad, int Npad, int Cout, int Cin, int KH, int KW,
                                                 int has_bias, float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float part[ZL][64];
    const int Kconv = KH * KW * Cin, Ng = Kconv + (has_bias ? 1 : 0);
    const int64_t total = (int64_t)Cout * Ng;
    const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t slab_elems = (int64_t)Mpad * Npad;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + e;
        const bool on = i < total;
        const int m = on ? (int)(i / Ng) : 0, n = on ? (int)(i - (int64_t)m * Ng) : 0;
        const float* src = slabs + (int64_t)m * Npad + n;
        float sacc = 0.f;
        int z = w;
        for (; z + 7 * ZL < S; z += 8 * ZL) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = on ? src[(int64_t)(z + ZL * j) * slab_elems] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sacc += v[j];
        }
        for (; z < S; z += ZL) sacc += on ? src[(int64_t)z * slab_elems] : 0.f;
        part[w][e] = sacc;
        __syncthreads();
        if (w == 0 && on) {
            float t = part[0][e];
#pragma unroll
            for (int j = 1; j < ZL; ++j) t += part[j][e];
            if (MODE == 1) dw[i] = t;
            else if (n < Kconv) {
                const int tap = n / Cin, ci = n - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
                dw[(((int64_t)m * Cin + ci) * KH + kh) * KW + kw] = t;
            } else if (dbias) dbias[m] = t;
        }
        __syncthreads();
    }
}

// variant B: thread owns 4 consecutive n (float4 loads), wave = 64 quads = 1 KB contiguous per slab; WG = ZL waves over slabs
template <int ZL>
__global__ __launch_bounds__(64 * ZL) void k_red4(const float* __restrict__ slabs, int S, int Mpad, int Npad, int Cout, int Cin, int KH, int KW,
                                                  int has_bias, float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float4 part[ZL][64];
    const int Kconv = KH * KW * Cin, Ng = Kconv + (has_bias ? 1 : 0);
    const int nq = Npad / 4;
    const int64_t totalq = (int64_t)Cout * nq;
    const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t slab_elems = (int64_t)Mpad * Npad;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < totalq; base += (int64_t)gridDim.x * 64) {
        const int64_t q = base + e;
        const bool on = q < totalq;
        const int m = on ? (int)(q / nq) : 0, n = on ? (int)(q - (int64_t)m * nq) * 4 : 0;
        const float4* src = (const float4*)(slabs + (int64_t)m * Npad + n);
        float4 s = {0, 0, 0, 0};
        int z = w;
        for (; z + 3 * ZL < S; z += 4 * ZL) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = on ? src[(int64_t)(z + ZL * j) * (slab_elems / 4)] : float4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        for (; z < S; z += ZL) { float4 v = on ? src[(int64_t)z * (slab_elems / 4)] : float4{0, 0, 0, 0}; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        part[w][e] = s;
        __syncthreads();
        if (w == 0 && on) {
            float4 t = part[0][e];
#pragma unroll
            for (int j = 1; j < ZL; ++j) { float4 u = part[j][e]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nn = n + k;
                if (nn < Kconv) {
                    const int tap = nn / Cin, ci = nn - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
                    dw[(((int64_t)m * Cin + ci) * KH + kh) * KW + kw] = tv[k];
                } else if (nn == Kconv && dbias) dbias[m] = tv[k];
            }
        }
        __syncthreads();
    }
}

int main() {
    struct Case { const char* name; int S, Cout, Cin, k, tm, tn; } cases[] = {
        {"layer1", 113, 64, 64, 3, 64, 64}, {"layer2", 28, 128, 128, 3, 64, 64}, {"layer3", 7, 256, 256, 3, 64, 64}, {"layer4", 2, 512, 512, 3, 64, 64},
        {"up(4,1)", 4, 256, 512, 3, 64, 64}, {"up(2,1)", 57, 64, 128, 3, 64, 64}, {"up(0,1)", 512, 16, 16, 3, 32, 128}, {"up(1,1)", 147, 32, 96, 3, 32, 128}};
    for (auto& c : cases) {
        const int Ng = c.k * c.k * c.Cin + 1;
        const int Mpad = (c.Cout + c.tm - 1) / c.tm * c.tm, Npad = (Ng + c.tn - 1) / c.tn * c.tn;
        const int64_t slab = (int64_t)Mpad * Npad, tot = slab * c.S;
        float *slabs, *dw, *db;
        CK(hipMalloc(&slabs, tot * 4)); CK(hipMalloc(&dw, (int64_t)c.Cout * Ng * 4 + 64)); CK(hipMalloc(&db, c.Cout * 4));
        CK(hipMemset(slabs, 0, tot * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto grid = [&](int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); };
        auto run = [&](int which) {
            const int g = grid((int64_t)c.Cout * Ng * 4), g4 = grid((int64_t)c.Cout * Npad);
            switch (which) {
            case 0: if (c.S >= 16) hipLaunchKernelGGL((k_red<16, 0>), dim3(g), dim3(1024), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db);
                    else hipLaunchKernelGGL((k_red<4, 0>), dim3(g), dim3(256), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db); break;
            case 1: if (c.S >= 16) hipLaunchKernelGGL((k_red<16, 1>), dim3(g), dim3(1024), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db);
                    else hipLaunchKernelGGL((k_red<4, 1>), dim3(g), dim3(256), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db); break;
            case 2: if (c.S >= 16) hipLaunchKernelGGL((k_red4<16>), dim3(g4), dim3(1024), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db);
                    else hipLaunchKernelGGL((k_red4<4>), dim3(g4), dim3(256), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db); break;
            case 3: if (c.S >= 8) hipLaunchKernelGGL((k_red4<8>), dim3(g4), dim3(512), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db);
                    else hipLaunchKernelGGL((k_red4<2>), dim3(g4), dim3(128), 0, 0, slabs, c.S, Mpad, Npad, c.Cout, c.Cin, c.k, c.k, 1, dw, db); break;
            }
        };
        printf("%-8s S=%3d slabs %.1f MB out %.2f MB:", c.name, c.S, tot * 4 / 1e6, (double)c.Cout * Ng * 4 / 1e6);
        for (int which = 0; which < 4; ++which) {
            for (int i = 0; i < 3; ++i) run(which);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) run(which);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  v%d %.1f us", which, ms / 20 * 1e3);
        }
        printf("\n");
        CK(hipFree(slabs)); CK(hipFree(dw)); CK(hipFree(db));
    }
    return 0;
}
