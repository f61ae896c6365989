// e2e_common.h -- shared host/device helpers for libe2eslam_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/e2eslam.h"

// ---- host side: error plumbing ---------------------------------------------------------------
void e2e_set_error(const char* fmt, ...);

#define E2E_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            e2e_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

#define E2E_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            e2e_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return E2E_ERR_LAUNCH;                                                     \
        }                                                                              \
    } while (0)

static inline int e2e_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device side ----------------------------------------------------------------------------
#define E2E_WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Deterministic block reduction: lane 0 of each wave writes to LDS, thread 0 adds them in wave order.
// `scratch` must hold blockDim/64 floats.  Result valid in thread 0 only.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    v = wave_sum(v);
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (tid == 0)
        for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// ReflectionPad2d(1) index map: -1 -> 1, n -> n-2 (valid for n >= 2).
__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// Per-batch projection geometry: cam = Ki * [x,y,1];  c = P[:, :3] * (d * cam) + P[:, 3]
// reference: depth_estimation/view_synthesis.py:36-39 (Kinv3 @ pix, * depth) and :57-59 (P = (K@T)[:3]).
struct Geom {
    float Ki[9];
    float P[12];
};

__device__ __forceinline__ Geom load_geom(const float* __restrict__ K, const float* __restrict__ invK,
                                          const float* __restrict__ T) {
    Geom g;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g.Ki[i * 3 + j] = invK[i * 4 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s = fmaf(K[i * 4 + k], T[k * 4 + j], s);
            g.P[i * 4 + j] = s;
        }
    return g;
}

enum { E2E_PAD_ZEROS = 0, E2E_PAD_BORDER = 1 };

// Everything the sampler needs about one target pixel.
struct Proj {
    float cam[3];  // Ki * [x,y,1]
    float r[3];    // P3 * cam            (dc/dd)
    float z, u, v; // z = c2 + eps
    float gx, gy;  // normalised grid coordinate (align_corners=True style normalisation: /(W-1))
    float mask;    // max(|gx|,|gy|) <= 1
};

__device__ __forceinline__ Proj project_pixel(const Geom& g, float x, float y, float d, int W, int H) {
    Proj p;
#pragma unroll
    for (int i = 0; i < 3; ++i) p.cam[i] = fmaf(g.Ki[i * 3 + 0], x, fmaf(g.Ki[i * 3 + 1], y, g.Ki[i * 3 + 2]));
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p.r[i] = fmaf(g.P[i * 4 + 0], p.cam[0], fmaf(g.P[i * 4 + 1], p.cam[1], g.P[i * 4 + 2] * p.cam[2]));
        c[i] = fmaf(d, p.r[i], g.P[i * 4 + 3]);
    }
    p.z = c[2] + 1e-7f;                      // view_synthesis.py:60 (eps added before the division)
    p.u = c[0] / p.z;
    p.v = c[1] / p.z;
    p.gx = (p.u / (float)(W - 1) - 0.5f) * 2.f;   // view_synthesis.py:66-68
    p.gy = (p.v / (float)(H - 1) - 0.5f) * 2.f;
    p.mask = (fmaxf(fabsf(p.gx), fabsf(p.gy)) <= 1.f) ? 1.f : 0.f;  // view_synthesis.py:70-71
    return p;
}

// grid_sample source index (ATen grid_sampler_compute_source_index_set_grad semantics).
template <int PAD, bool ALIGN>
__device__ __forceinline__ float source_index(float g, int size, float& mult) {
    float ix;
    if (ALIGN) {
        ix = ((g + 1.f) / 2.f) * (float)(size - 1);
        mult = (float)(size - 1) / 2.f;
    } else {
        ix = ((g + 1.f) * (float)size - 1.f) / 2.f;
        mult = (float)size / 2.f;
    }
    if (PAD == E2E_PAD_BORDER) {
        if (ix <= 0.f) {
            ix = 0.f;
            mult = 0.f;
        } else if (ix >= (float)(size - 1)) {
            ix = (float)(size - 1);
            mult = 0.f;
        }
    }
    return ix;
}

struct Bilin {
    int x0, y0;            // north-west corner
    float wnw, wne, wsw, wse;
    float tx, ty;          // ix - x0, iy - y0
    bool in_x0, in_x1, in_y0, in_y1;
};

__device__ __forceinline__ Bilin bilinear_setup(float ix, float iy, int W, int H) {
    Bilin b;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    // clamp before the int conversion so that wild coordinates (|ix| > 2^31) stay defined; such
    // corners are out of bounds either way.
    b.x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f);
    b.y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
    b.tx = ix - fx0;
    b.ty = iy - fy0;
    const float x1w = (fx0 + 1.f) - ix, y1w = (fy0 + 1.f) - iy;
    b.wnw = x1w * y1w;
    b.wne = b.tx * y1w;
    b.wsw = x1w * b.ty;
    b.wse = b.tx * b.ty;
    b.in_x0 = (b.x0 >= 0) & (b.x0 < W);
    b.in_x1 = (b.x0 + 1 >= 0) & (b.x0 + 1 < W);
    b.in_y0 = (b.y0 >= 0) & (b.y0 < H);
    b.in_y1 = (b.y0 + 1 >= 0) & (b.y0 + 1 < H);
    return b;
}
