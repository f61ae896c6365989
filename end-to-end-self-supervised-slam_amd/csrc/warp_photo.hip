// warp_photo.hip -- inverse-warp view synthesis + SSIM/L1 photometric loss for gfx950 (CDNA4).
//
// Reference semantics (cited per kernel): depth_estimation/view_synthesis.py, loss/losses.py,
// online_adaption.py:412-455,473-511,544-564,612-623.  All kernels are HBM/latency bound: one
// thread per pixel, 32x8 pixel tiles (4 wave64 per workgroup, 1200 workgroups at 480x640 ->
// every CU holds 4-5 resident workgroups), image tiles staged through LDS with their halo,
// wave-shuffle + LDS block reductions, and a fixed-order second-stage reduction (no float
// atomics => bitwise reproducible losses).
#include "e2e_common.h"

#define TW 32
#define TH 8
#define NTHREADS (TW * TH)

// ---------------------------------------------------------------------------------------------
// small element-wise / per-pixel kernels behind the reference's module-level API
// ---------------------------------------------------------------------------------------------

// BackprojectDepth.forward -- view_synthesis.py:34-40
__global__ void k_backproject_fwd(const float* __restrict__ depth, const float* __restrict__ invK,
                                  float* __restrict__ cam, int H, int W) {
    const int N = H * W;
    const int b = blockIdx.y;
    const float* Ki = invK + b * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float x = (float)(i % W), y = (float)(i / W);
        const float d = depth[(int64_t)b * N + i];
        float* o = cam + (int64_t)b * 4 * N + i;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[(int64_t)r * N] = d * fmaf(Ki[r * 4 + 0], x, fmaf(Ki[r * 4 + 1], y, Ki[r * 4 + 2]));
        o[(int64_t)3 * N] = 1.f;
    }
}

__global__ void k_backproject_bwd(const float* __restrict__ gcam, const float* __restrict__ invK,
                                  float* __restrict__ gdepth, int H, int W) {
    const int N = H * W;
    const int b = blockIdx.y;
    const float* Ki = invK + b * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float x = (float)(i % W), y = (float)(i / W);
        const float* g = gcam + (int64_t)b * 4 * N + i;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            s = fmaf(g[(int64_t)r * N], fmaf(Ki[r * 4 + 0], x, fmaf(Ki[r * 4 + 1], y, Ki[r * 4 + 2])), s);
        gdepth[(int64_t)b * N + i] = s;
    }
}

__device__ __forceinline__ void load_P(const float* K, const float* T, float* P) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s = fmaf(K[i * 4 + k], T[k * 4 + j], s);
            P[i * 4 + j] = s;
        }
}

// Project3D.forward -- view_synthesis.py:54-78
__global__ void k_project3d_fwd(const float* __restrict__ pts, const float* __restrict__ K,
                                const float* __restrict__ T, float* __restrict__ grid,
                                float* __restrict__ valid, float* __restrict__ zout, int H, int W) {
    const int N = H * W;
    const int b = blockIdx.y;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float* p = pts + (int64_t)b * 4 * N + i;
        const float p0 = p[0], p1 = p[N], p2 = p[(int64_t)2 * N], p3 = p[(int64_t)3 * N];
        float c[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            c[r] = fmaf(P[r * 4 + 0], p0, fmaf(P[r * 4 + 1], p1, fmaf(P[r * 4 + 2], p2, P[r * 4 + 3] * p3)));
        const float z = c[2] + 1e-7f;
        const float gx = ((c[0] / z) / (float)(W - 1) - 0.5f) * 2.f;
        const float gy = ((c[1] / z) / (float)(H - 1) - 0.5f) * 2.f;
        const int64_t o = (int64_t)b * N + i;
        grid[o * 2 + 0] = gx;
        grid[o * 2 + 1] = gy;
        valid[o] = (fmaxf(fabsf(gx), fabsf(gy)) <= 1.f) ? 1.f : 0.f;
        if (zout) zout[o] = fmaxf(c[2], 1e-3f);
    }
}

__global__ void k_project3d_bwd(const float* __restrict__ pts, const float* __restrict__ K,
                                const float* __restrict__ T, const float* __restrict__ ggrid,
                                const float* __restrict__ gz, float* __restrict__ gpts, int H, int W) {
    const int N = H * W;
    const int b = blockIdx.y;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float* p = pts + (int64_t)b * 4 * N + i;
        const float p0 = p[0], p1 = p[N], p2 = p[(int64_t)2 * N], p3 = p[(int64_t)3 * N];
        float c[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            c[r] = fmaf(P[r * 4 + 0], p0, fmaf(P[r * 4 + 1], p1, fmaf(P[r * 4 + 2], p2, P[r * 4 + 3] * p3)));
        const float z = c[2] + 1e-7f;
        const float u = c[0] / z, v = c[1] / z;
        const int64_t o = (int64_t)b * N + i;
        const float gu = ggrid[o * 2 + 0] * 2.f / (float)(W - 1);
        const float gv = ggrid[o * 2 + 1] * 2.f / (float)(H - 1);
        float gc[3];
        gc[0] = gu / z;
        gc[1] = gv / z;
        gc[2] = -(gu * u + gv * v) / z;
        if (gz && c[2] > 1e-3f) gc[2] += gz[o];     // clamp(min=1e-3) passes gradient above the bound
        float* g = gpts + (int64_t)b * 4 * N + i;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            g[(int64_t)j * N] = fmaf(P[0 * 4 + j], gc[0], fmaf(P[1 * 4 + j], gc[1], P[2 * 4 + j] * gc[2]));
    }
}

// F.grid_sample bilinear forward -- ATen grid_sampler_2d semantics (online_adaption.py:450-453)
template <int PAD, bool ALIGN>
__global__ void k_grid_sample_fwd(const float* __restrict__ in, e2e_strides s, const float* __restrict__ grid,
                                  float* __restrict__ out, int C, int Hi, int Wi, int Ho, int Wo) {
    const int No = Ho * Wo;
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < No; i += gridDim.x * blockDim.x) {
        const int64_t o = (int64_t)b * No + i;
        float mx, my;
        const float ix = source_index<PAD, ALIGN>(grid[o * 2 + 0], Wi, mx);
        const float iy = source_index<PAD, ALIGN>(grid[o * 2 + 1], Hi, my);
        const Bilin bl = bilinear_setup(ix, iy, Wi, Hi);
        const float* base = in + b * s.sb;
        const int64_t o00 = bl.y0 * s.sh + bl.x0 * s.sw;
        for (int c = 0; c < C; ++c) {
            const float* pc = base + c * s.sc + o00;
            float acc = 0.f;
            if (bl.in_y0 & bl.in_x0) acc += pc[0] * bl.wnw;
            if (bl.in_y0 & bl.in_x1) acc += pc[s.sw] * bl.wne;
            if (bl.in_y1 & bl.in_x0) acc += pc[s.sh] * bl.wsw;
            if (bl.in_y1 & bl.in_x1) acc += pc[s.sh + s.sw] * bl.wse;
            out[((int64_t)b * C + c) * No + i] = acc;
        }
    }
}

template <int PAD, bool ALIGN>
__global__ void k_grid_sample_bwd(const float* __restrict__ in, e2e_strides s, const float* __restrict__ grid,
                                  const float* __restrict__ gout, float* __restrict__ ggrid,
                                  float* __restrict__ gin, int C, int Hi, int Wi, int Ho, int Wo, unsigned long long* __restrict__ gin_fx) {
    const int No = Ho * Wo;
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < No; i += gridDim.x * blockDim.x) {
        const int64_t o = (int64_t)b * No + i;
        float mx, my;
        const float ix = source_index<PAD, ALIGN>(grid[o * 2 + 0], Wi, mx);
        const float iy = source_index<PAD, ALIGN>(grid[o * 2 + 1], Hi, my);
        const Bilin bl = bilinear_setup(ix, iy, Wi, Hi);
        const float* base = in + b * s.sb;
        const int64_t o00 = bl.y0 * s.sh + bl.x0 * s.sw;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* pc = base + c * s.sc + o00;
            const float g = gout[((int64_t)b * C + c) * No + i];
            const float nw = (bl.in_y0 & bl.in_x0) ? pc[0] : 0.f;
            const float ne = (bl.in_y0 & bl.in_x1) ? pc[s.sw] : 0.f;
            const float sw = (bl.in_y1 & bl.in_x0) ? pc[s.sh] : 0.f;
            const float se = (bl.in_y1 & bl.in_x1) ? pc[s.sh + s.sw] : 0.f;
            gix += g * ((ne - nw) * (1.f - bl.ty) + (se - sw) * bl.ty);
            giy += g * ((sw - nw) * (1.f - bl.tx) + (se - ne) * bl.tx);
            if (gin_fx) {  // scatter-add into a (B,C,Hi,Wi) image of 2^-48 fixed-point sums: integer adds commute, so the result does
                // not depend on the arrival order (bitwise reproducible; |sum| < 32768, resolution 3.6e-15)
                // A contribution that is not finite, or too large for the format (|v| >= 4096: sums are specified below 32768), is NOT added:
                // it raises the poison word behind the image (gin_fx[B C Hi Wi]) and the conversion pass then returns NaN everywhere --
                // torch would put NaN / Inf at the affected pixels; a silently wrapped or saturated finite value is the one wrong answer
                unsigned long long* gi = gin_fx + (((int64_t)b * C + c) * Hi + bl.y0) * Wi + bl.x0;
                const float v4[4] = {g * bl.wnw, g * bl.wne, g * bl.wsw, g * bl.wse};
                const bool on4[4] = {(bool)(bl.in_y0 & bl.in_x0), (bool)(bl.in_y0 & bl.in_x1), (bool)(bl.in_y1 & bl.in_x0), (bool)(bl.in_y1 & bl.in_x1)};
                const int64_t of4[4] = {0, 1, Wi, (int64_t)Wi + 1};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!on4[q]) continue;
                    if (fabsf(v4[q]) < 4096.f) atomicAdd(gi + of4[q], (unsigned long long)__double2ll_rn((double)v4[q] * 281474976710656.0));
                    else atomicOr(gin_fx + (int64_t)gridDim.y * C * Hi * Wi, 1ull);
                }
            } else if (gin) {  // floating-point scatter-add: run-to-run differences in the last bits (kept for callers without scratch)
                float* gi = gin + (((int64_t)b * C + c) * Hi + bl.y0) * Wi + bl.x0;
                if (bl.in_y0 & bl.in_x0) atomicAdd(gi, g * bl.wnw);
                if (bl.in_y0 & bl.in_x1) atomicAdd(gi + 1, g * bl.wne);
                if (bl.in_y1 & bl.in_x0) atomicAdd(gi + Wi, g * bl.wsw);
                if (bl.in_y1 & bl.in_x1) atomicAdd(gi + Wi + 1, g * bl.wse);
            }
        }
        ggrid[o * 2 + 0] = mx * gix;
        ggrid[o * 2 + 1] = my * giy;
    }
}

// ---------------------------------------------------------------------------------------------
// SSIM statistics on an LDS plane.  xs/ys point at the window's top-left element, `ld` = row pitch.
// Summation order = avg_pool2d's (rows outer, columns inner), then /9 -- losses.py:27-32.
// ---------------------------------------------------------------------------------------------
struct Stats {
    float mux, muy, sxx, syy, sxy;
};
__device__ __forceinline__ Stats window_stats(const float* xs, const float* ys, int ld) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float a = xs[dy * ld + dx], b = ys[dy * ld + dx];
            sx += a;
            sy += b;
            sxx += a * a;
            syy += b * b;
            sxy += a * b;
        }
    Stats s;
    s.mux = sx / 9.f;
    s.muy = sy / 9.f;
    s.sxx = sxx / 9.f;
    s.syy = syy / 9.f;
    s.sxy = sxy / 9.f;
    return s;
}

#define SSIM_C1 1e-4f   // 0.01**2  losses.py:20
#define SSIM_C2 9e-4f   // 0.03**2  losses.py:21

__device__ __forceinline__ float ssim_value(const Stats& s) {
    const float sigx = s.sxx - s.mux * s.mux;
    const float sigy = s.syy - s.muy * s.muy;
    const float sigxy = s.sxy - s.mux * s.muy;
    const float n = (2.f * s.mux * s.muy + SSIM_C1) * (2.f * sigxy + SSIM_C2);
    const float d = (s.mux * s.mux + s.muy * s.muy + SSIM_C1) * (sigx + sigy + SSIM_C2);
    return fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
}

// d(ssim_c)/d(mux, sxx, sxy) scaled by the upstream gradient g of ssim_c  
__device__ __forceinline__ void ssim_partials(const Stats& s, float g, float& G1, float& G2, float& G3) {
    const float sigx = s.sxx - s.mux * s.mux;
    const float sigy = s.syy - s.muy * s.muy;
    const float sigxy = s.sxy - s.mux * s.muy;
    const float A1 = 2.f * s.mux * s.muy + SSIM_C1, A2 = 2.f * sigxy + SSIM_C2;
    const float B1 = s.mux * s.mux + s.muy * s.muy + SSIM_C1, B2 = sigx + sigy + SSIM_C2;
    const float S = (A1 * A2) / (B1 * B2);
    const float t = (1.f - S) / 2.f;
    const float gS = (t >= 0.f && t <= 1.f) ? -0.5f * g : 0.f;   // clamp passes grad on the closed interval
    const float inv = 1.f / (B1 * B2);
    G1 = gS * (2.f * s.muy * (A2 - A1) * inv - S * 2.f * s.mux * (1.f / B1 - 1.f / B2));
    G2 = gS * (-S / B2);
    G3 = gS * (2.f * A1 * inv);
}

// multiplicity with which output pixel q's reflect-padded 3-window sees input pixel p (|p-q| <= 1)
__device__ __forceinline__ float refl_mult(int p, int q, int n) {
    return ((q == 0 && p == 1) || (q == n - 1 && p == n - 2)) ? 2.f : 1.f;
}

// ---------------------------------------------------------------------------------------------
// modular photometric forward / backward: C channels, one LDS plane pair per channel pass
// SSIM.forward losses.py:23-37, photometric_loss losses.py:97-117
// ---------------------------------------------------------------------------------------------
#define FP_LD (TW + 2)
__global__ __launch_bounds__(NTHREADS) void k_photometric_fwd(
    const float* __restrict__ x, e2e_strides xs, const float* __restrict__ y, e2e_strides ys,
    float* __restrict__ ssim_out, float* __restrict__ pmap_out, int C, int H, int W) {
    __shared__ float lx[(TH + 2) * FP_LD], ly[(TH + 2) * FP_LD];
    const int b = blockIdx.z, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const int px = tx0 + threadIdx.x, py = ty0 + threadIdx.y;
    const bool live = px < W && py < H;
    float ssim_acc = 0.f, l1_acc = 0.f;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
        for (int i = tid; i < (TH + 2) * FP_LD; i += NTHREADS) {
            const int l_y = i / FP_LD, l_x = i % FP_LD;
            const int gy = ty0 + l_y - 1, gx = tx0 + l_x - 1;
            float a = 0.f, bb = 0.f;
            if (gx >= -1 && gx <= W && gy >= -1 && gy <= H) {
                const int qx = reflect1(gx, W), qy = reflect1(gy, H);
                a = x[b * xs.sb + c * xs.sc + qy * xs.sh + qx * xs.sw];
                bb = y[b * ys.sb + c * ys.sc + qy * ys.sh + qx * ys.sw];
            }
            lx[i] = a;
            ly[i] = bb;
        }
        __syncthreads();
        if (live) {
            const Stats s = window_stats(lx + threadIdx.y * FP_LD + threadIdx.x, ly + threadIdx.y * FP_LD + threadIdx.x, FP_LD);
            const float v = ssim_value(s);
            if (ssim_out) ssim_out[(((int64_t)b * C + c) * H + py) * W + px] = v;
            ssim_acc += v;
            const int ctr = (threadIdx.y + 1) * FP_LD + threadIdx.x + 1;
            l1_acc += fabsf(ly[ctr] - lx[ctr]);
        }
    }
    if (live && pmap_out) pmap_out[((int64_t)b * H + py) * W + px] = 0.85f * (ssim_acc / (float)C) + 0.15f * (l1_acc / (float)C);
}

#define BP_LD (TW + 4)
#define BG_LD (TW + 2)
__global__ __launch_bounds__(NTHREADS) void k_photometric_bwd(
    const float* __restrict__ x, e2e_strides xs, const float* __restrict__ y, e2e_strides ys,
    const float* __restrict__ g_pmap, const float* __restrict__ g_ssim, float* __restrict__ g_x,
    int C, int H, int W) {
    __shared__ float lx[(TH + 4) * BP_LD], ly[(TH + 4) * BP_LD];
    __shared__ float lg[3][(TH + 2) * BG_LD];
    const int b = blockIdx.z, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const int px = tx0 + threadIdx.x, py = ty0 + threadIdx.y;
    const bool live = px < W && py < H;
    const float gp_c = (live && g_pmap) ? g_pmap[((int64_t)b * H + py) * W + px] : 0.f;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
        for (int i = tid; i < (TH + 4) * BP_LD; i += NTHREADS) {
            const int l_y = i / BP_LD, l_x = i % BP_LD;
            const int gy = ty0 + l_y - 2, gx = tx0 + l_x - 2;
            float a = 0.f, bb = 0.f;
            if (gx >= -1 && gx <= W && gy >= -1 && gy <= H) {
                const int qx = reflect1(gx, W), qy = reflect1(gy, H);
                a = x[b * xs.sb + c * xs.sc + qy * xs.sh + qx * xs.sw];
                bb = y[b * ys.sb + c * ys.sc + qy * ys.sh + qx * ys.sw];
            }
            lx[i] = a;
            ly[i] = bb;
        }
        __syncthreads();
        for (int i = tid; i < (TH + 2) * BG_LD; i += NTHREADS) {
            const int l_y = i / BG_LD, l_x = i % BG_LD;
            const int qy = ty0 + l_y - 1, qx = tx0 + l_x - 1;
            float G1 = 0.f, G2 = 0.f, G3 = 0.f;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                float g = 0.f;
                if (g_pmap) g += g_pmap[((int64_t)b * H + qy) * W + qx] * (0.85f / (float)C);
                if (g_ssim) g += g_ssim[(((int64_t)b * C + c) * H + qy) * W + qx];
                const Stats s = window_stats(lx + l_y * BP_LD + l_x, ly + l_y * BP_LD + l_x, BP_LD);
                ssim_partials(s, g, G1, G2, G3);
            }
            lg[0][i] = G1;
            lg[1][i] = G2;
            lg[2][i] = G3;
        }
        __syncthreads();
        if (live) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int qx = px + dx, qy = py + dy;
                    if (qx < 0 || qx >= W || qy < 0 || qy >= H) continue;
                    const float m = refl_mult(px, qx, W) * refl_mult(py, qy, H);
                    const int gi = (threadIdx.y + 1 + dy) * BG_LD + threadIdx.x + 1 + dx;
                    s1 += m * lg[0][gi];
                    s2 += m * lg[1][gi];
                    s3 += m * lg[2][gi];
                }
            const int ctr = (threadIdx.y + 2) * BP_LD + threadIdx.x + 2;
            const float xv = lx[ctr], yv = ly[ctr];
            float g = (s1 + 2.f * xv * s2 + yv * s3) / 9.f;
            const float df = yv - xv;
            const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
            g -= gp_c * (0.15f / (float)C) * sg;
            g_x[(((int64_t)b * C + c) * H + py) * W + px] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused forward: warp + mask + photometric (+ depth regulariser) -> synth, valid, pmap, partial sums
// online_adaption.py:412-455 (novel_view_synthesis), :544-564, :482-511, :612-623
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample3(const float* __restrict__ base, const e2e_strides& s, const Bilin& bl, float* out) {
    const float* p = base + bl.y0 * s.sh + bl.x0 * s.sw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pc = p + c * s.sc;
        float acc = 0.f;
        if (bl.in_y0 & bl.in_x0) acc += pc[0] * bl.wnw;
        if (bl.in_y0 & bl.in_x1) acc += pc[s.sw] * bl.wne;
        if (bl.in_y1 & bl.in_x0) acc += pc[s.sh] * bl.wsw;
        if (bl.in_y1 & bl.in_x1) acc += pc[s.sh + s.sw] * bl.wse;
        out[c] = acc;
    }
}

template <int PAD>
__global__ __launch_bounds__(NTHREADS) void k_warp_photo_fwd(
    const float* __restrict__ depth, const float* __restrict__ src, e2e_strides ss,
    const float* __restrict__ tgt, e2e_strides ts, const float* __restrict__ K,
    const float* __restrict__ invK, const float* __restrict__ T, float* __restrict__ synth,
    float* __restrict__ valid, float* __restrict__ pmap, int use_mask, int reg_kind,
    const float* __restrict__ ri_t, const float* __restrict__ ri_s, const float* __restrict__ d_s,
    float* __restrict__ partials, int H, int W) {
    __shared__ float lx[3][(TH + 2) * FP_LD], ly[3][(TH + 2) * FP_LD];
    __shared__ float red[NTHREADS / 64];
    const int b = blockIdx.z, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const int64_t N = (int64_t)H * W;
    const Geom g = load_geom(K + b * 16, invK + b * 16, T + b * 16);
    const float* dep = depth + b * N;
    const float* sb = src + b * ss.sb;
    const float* tb = tgt + b * ts.sb;

    for (int i = tid; i < (TH + 2) * FP_LD; i += NTHREADS) {
        const int l_y = i / FP_LD, l_x = i % FP_LD;
        const int gy = ty0 + l_y - 1, gx = tx0 + l_x - 1;
        float xv[3] = {0.f, 0.f, 0.f}, yv[3] = {0.f, 0.f, 0.f};
        if (gx >= -1 && gx <= W && gy >= -1 && gy <= H) {
            const int qx = reflect1(gx, W), qy = reflect1(gy, H);
            const Proj p = project_pixel(g, (float)qx, (float)qy, dep[qy * W + qx], W, H);
            float mx, my;
            const float ix = source_index<PAD, false>(p.gx, W, mx);
            const float iy = source_index<PAD, false>(p.gy, H, my);
            const Bilin bl = bilinear_setup(ix, iy, W, H);
            float s[3];
            sample3(sb, ss, bl, s);
            const float m = use_mask ? p.mask : 1.f;
            const float* tp = tb + qy * ts.sh + qx * ts.sw;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xv[c] = s[c] * m;
                yv[c] = tp[c * ts.sc] * m;
            }
            if (qx == gx && qy == gy && l_x >= 1 && l_x <= TW && l_y >= 1 && l_y <= TH) {  // tile interior, real pixel
                const int64_t o = b * N + (int64_t)qy * W + qx;
                valid[o] = p.mask;
#pragma unroll
                for (int c = 0; c < 3; ++c) synth[(b * 3 + c) * N + (int64_t)qy * W + qx] = s[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lx[c][i] = xv[c];
            ly[c][i] = yv[c];
        }
    }
    __syncthreads();

    const int px = tx0 + threadIdx.x, py = ty0 + threadIdx.y;
    float val = 0.f, reg = 0.f;
    if (px < W && py < H) {
        float sacc = 0.f, lacc = 0.f;
        const int w0 = threadIdx.y * FP_LD + threadIdx.x, ctr = w0 + FP_LD + 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Stats s = window_stats(lx[c] + w0, ly[c] + w0, FP_LD);
            sacc += ssim_value(s);
            lacc += fabsf(ly[c][ctr] - lx[c][ctr]);
        }
        val = 0.85f * (sacc / 3.f) + 0.15f * (lacc / 3.f);
        const int64_t o = b * N + (int64_t)py * W + px;
        if (pmap) pmap[o] = val;
        if (reg_kind) {
            const float e0 = ri_t[o] - depth[o], e1 = ri_s[o] - d_s[o];
            reg = (reg_kind == 2) ? (e0 * e0 + e1 * e1) : (fabsf(e0) + fabsf(e1));
        }
    }
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const float s0 = block_sum(val, red);
    if (tid == 0) partials[blk] = s0;
    if (reg_kind) {                      // wave-uniform: reg_kind is a kernel argument
        __syncthreads();
        const float s1 = block_sum(reg, red);
        if (tid == 0) partials[nblk + blk] = s1;
    }
}

// second stage: fixed-order double-precision sum of the per-block partials (1 block)
__global__ void k_reduce_partials(const float* __restrict__ partials, int nblk, int nsets, float scale,
                                  float* __restrict__ out) {
    __shared__ double sh[4];
    for (int s = 0; s < nsets; ++s) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += (double)partials[s * nblk + i];
        acc = wave_sum_d(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
            out[s] = (float)(t * (double)scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused backward: photometric adjoint (3x3 box + reflect pad fold-back) -> warp adjoint -> d/d depth
// ---------------------------------------------------------------------------------------------
template <int PAD>
__global__ __launch_bounds__(NTHREADS) void k_warp_photo_bwd(
    const float* __restrict__ depth, const float* __restrict__ src, e2e_strides ss,
    const float* __restrict__ tgt, e2e_strides ts, const float* __restrict__ K,
    const float* __restrict__ invK, const float* __restrict__ T, const float* __restrict__ synth,
    const float* __restrict__ valid, int use_mask, int reg_kind, const float* __restrict__ ri_t,
    const float* __restrict__ ri_s, const float* __restrict__ d_s, const float* __restrict__ g_loss,
    float* __restrict__ g_dt, float* __restrict__ g_ds, int B, int H, int W) {
    __shared__ float lx[3][(TH + 4) * BP_LD], ly[3][(TH + 4) * BP_LD];
    __shared__ float lg[9][(TH + 2) * BG_LD];
    const int b = blockIdx.z, tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const int64_t N = (int64_t)H * W;
    const float kmean = 1.f / ((float)B * (float)H * (float)W);
    const float gl0 = g_loss[0] * kmean;          // upstream on every pmap element
    const float* tb = tgt + b * ts.sb;

    for (int i = tid; i < (TH + 4) * BP_LD; i += NTHREADS) {
        const int l_y = i / BP_LD, l_x = i % BP_LD;
        const int gy = ty0 + l_y - 2, gx = tx0 + l_x - 2;
        float xv[3] = {0.f, 0.f, 0.f}, yv[3] = {0.f, 0.f, 0.f};
        if (gx >= -1 && gx <= W && gy >= -1 && gy <= H) {
            const int qx = reflect1(gx, W), qy = reflect1(gy, H);
            const int64_t o = (int64_t)qy * W + qx;
            const float m = use_mask ? valid[b * N + o] : 1.f;
            const float* tp = tb + qy * ts.sh + qx * ts.sw;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xv[c] = synth[(b * 3 + c) * N + o] * m;
                yv[c] = tp[c * ts.sc] * m;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lx[c][i] = xv[c];
            ly[c][i] = yv[c];
        }
    }
    __syncthreads();
    for (int i = tid; i < (TH + 2) * BG_LD; i += NTHREADS) {
        const int l_y = i / BG_LD, l_x = i % BG_LD;
        const int qy = ty0 + l_y - 1, qx = tx0 + l_x - 1;
        const bool in = qx >= 0 && qx < W && qy >= 0 && qy < H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float G1 = 0.f, G2 = 0.f, G3 = 0.f;
            if (in) {
                const Stats s = window_stats(lx[c] + l_y * BP_LD + l_x, ly[c] + l_y * BP_LD + l_x, BP_LD);
                ssim_partials(s, gl0 * (0.85f / 3.f), G1, G2, G3);
            }
            lg[c * 3 + 0][i] = G1;
            lg[c * 3 + 1][i] = G2;
            lg[c * 3 + 2][i] = G3;
        }
    }
    __syncthreads();

    const int px = tx0 + threadIdx.x, py = ty0 + threadIdx.y;
    if (px >= W || py >= H) return;
    const int64_t o = b * N + (int64_t)py * W + px;
    float G[3];
    {
        float mult[9];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int qx = px + dx, qy = py + dy;
                mult[(dy + 1) * 3 + dx + 1] = (qx < 0 || qx >= W || qy < 0 || qy >= H) ? 0.f : refl_mult(px, qx, W) * refl_mult(py, qy, H);
            }
        const int ctr = (threadIdx.y + 2) * BP_LD + threadIdx.x + 2;
        const float m = use_mask ? valid[o] : 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int gi = (threadIdx.y + dy) * BG_LD + threadIdx.x + dx;
                    const float mm = mult[dy * 3 + dx];
                    s1 += mm * lg[c * 3 + 0][gi];
                    s2 += mm * lg[c * 3 + 1][gi];
                    s3 += mm * lg[c * 3 + 2][gi];
                }
            const float xv = lx[c][ctr], yv = ly[c][ctr];
            float gx_ = (s1 + 2.f * xv * s2 + yv * s3) / 9.f;
            const float df = yv - xv;
            const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
            gx_ -= gl0 * (0.15f / 3.f) * sg;
            G[c] = gx_ * m;                                // d/d synth = d/dx * mask
        }
    }
    // warp adjoint: d synth / d grid (bilinear), grid -> (u,v) -> c -> depth
    const Geom g = load_geom(K + b * 16, invK + b * 16, T + b * 16);
    const float d = depth[o];
    const Proj p = project_pixel(g, (float)px, (float)py, d, W, H);
    float mx, my;
    const float ix = source_index<PAD, false>(p.gx, W, mx);
    const float iy = source_index<PAD, false>(p.gy, H, my);
    const Bilin bl = bilinear_setup(ix, iy, W, H);
    const float* sp = src + b * ss.sb + bl.y0 * ss.sh + bl.x0 * ss.sw;
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pc = sp + c * ss.sc;
        const float nw = (bl.in_y0 & bl.in_x0) ? pc[0] : 0.f;
        const float ne = (bl.in_y0 & bl.in_x1) ? pc[ss.sw] : 0.f;
        const float sw = (bl.in_y1 & bl.in_x0) ? pc[ss.sh] : 0.f;
        const float se = (bl.in_y1 & bl.in_x1) ? pc[ss.sh + ss.sw] : 0.f;
        gix += G[c] * ((ne - nw) * (1.f - bl.ty) + (se - sw) * bl.ty);
        giy += G[c] * ((sw - nw) * (1.f - bl.tx) + (se - ne) * bl.tx);
    }
    const float gu = (mx * gix) * 2.f / (float)(W - 1);
    const float gv = (my * giy) * 2.f / (float)(H - 1);
    const float gc0 = gu / p.z, gc1 = gv / p.z, gc2 = -(gu * p.u + gv * p.v) / p.z;
    float gd = gc0 * p.r[0] + gc1 * p.r[1] + gc2 * p.r[2];
    if (reg_kind) {
        const float gl1 = g_loss[1] * kmean;
        const float e0 = ri_t[o] - d, e1 = ri_s[o] - d_s[o];
        if (reg_kind == 2) {
            gd += gl1 * (-2.f * e0);
            g_ds[o] = gl1 * (-2.f * e1);
        } else {
            gd += gl1 * ((e0 > 0.f) ? -1.f : ((e0 < 0.f) ? 1.f : 0.f));
            g_ds[o] = gl1 * ((e1 > 0.f) ? -1.f : ((e1 < 0.f) ? 1.f : 0.f));
        }
    }
    g_dt[o] = gd;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline dim3 tile_grid(int B, int H, int W) { return dim3(e2e_ceil_div(W, TW), e2e_ceil_div(H, TH), B); }
static inline dim3 flat_grid(int B, int64_t N) {
    int g = e2e_ceil_div(N, 256);
    if (g > 2048) g = 2048;
    return dim3(g, B, 1);
}
#define CHECK_DIMS(name)                                                                      \
    E2E_REQUIRE(B > 0 && H > 1 && W > 1 && (int64_t)B * H * W < (1ll << 31), E2E_ERR_ARG,     \
                name ": bad dims B=%d H=%d W=%d", B, H, W)

extern "C" {

int e2e_backproject_fwd(const float* depth, const float* inv_K, float* cam, int B, int H, int W, void* stream) {
    CHECK_DIMS("e2e_backproject_fwd");
    E2E_REQUIRE(depth && inv_K && cam, E2E_ERR_ARG, "e2e_backproject_fwd: null pointer");
    hipLaunchKernelGGL(k_backproject_fwd, flat_grid(B, (int64_t)H * W), dim3(256), 0, (hipStream_t)stream, depth, inv_K, cam, H, W);
    E2E_LAUNCH_CHECK("e2e_backproject_fwd");
    return E2E_OK;
}

int e2e_backproject_bwd(const float* g_cam, const float* inv_K, float* g_depth, int B, int H, int W, void* stream) {
    CHECK_DIMS("e2e_backproject_bwd");
    E2E_REQUIRE(g_cam && inv_K && g_depth, E2E_ERR_ARG, "e2e_backproject_bwd: null pointer");
    hipLaunchKernelGGL(k_backproject_bwd, flat_grid(B, (int64_t)H * W), dim3(256), 0, (hipStream_t)stream, g_cam, inv_K, g_depth, H, W);
    E2E_LAUNCH_CHECK("e2e_backproject_bwd");
    return E2E_OK;
}

int e2e_project3d_fwd(const float* points, const float* K, const float* T, float* grid, float* valid, float* z_out,
                      int B, int H, int W, void* stream) {
    CHECK_DIMS("e2e_project3d_fwd");
    E2E_REQUIRE(points && K && T && grid && valid, E2E_ERR_ARG, "e2e_project3d_fwd: null pointer");
    hipLaunchKernelGGL(k_project3d_fwd, flat_grid(B, (int64_t)H * W), dim3(256), 0, (hipStream_t)stream, points, K, T, grid, valid, z_out, H, W);
    E2E_LAUNCH_CHECK("e2e_project3d_fwd");
    return E2E_OK;
}

int e2e_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid, const float* g_z,
                      float* g_points, int B, int H, int W, void* stream) {
    CHECK_DIMS("e2e_project3d_bwd");
    E2E_REQUIRE(points && K && T && g_grid && g_points, E2E_ERR_ARG, "e2e_project3d_bwd: null pointer");
    hipLaunchKernelGGL(k_project3d_bwd, flat_grid(B, (int64_t)H * W), dim3(256), 0, (hipStream_t)stream, points, K, T, g_grid, g_z, g_points, H, W);
    E2E_LAUNCH_CHECK("e2e_project3d_bwd");
    return E2E_OK;
}

#define DISPATCH_PAD_ALIGN(KERN, ...)                                                             \
    do {                                                                                          \
        if (padding_mode == E2E_PADDING_BORDER && align_corners)                                  \
            hipLaunchKernelGGL((KERN<E2E_PAD_BORDER, true>), __VA_ARGS__);                        \
        else if (padding_mode == E2E_PADDING_BORDER)                                              \
            hipLaunchKernelGGL((KERN<E2E_PAD_BORDER, false>), __VA_ARGS__);                       \
        else if (align_corners)                                                                   \
            hipLaunchKernelGGL((KERN<E2E_PAD_ZEROS, true>), __VA_ARGS__);                         \
        else                                                                                      \
            hipLaunchKernelGGL((KERN<E2E_PAD_ZEROS, false>), __VA_ARGS__);                        \
    } while (0)

int e2e_grid_sample_fwd(const float* input, e2e_strides in_strides, const float* grid, float* out, int B, int C,
                        int Hi, int Wi, int Ho, int Wo, int padding_mode, int align_corners, void* stream) {
    E2E_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, E2E_ERR_ARG, "e2e_grid_sample_fwd: bad dims");
    E2E_REQUIRE(input && grid && out, E2E_ERR_ARG, "e2e_grid_sample_fwd: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_grid_sample_fwd: padding_mode %d not supported (zeros|border)", padding_mode);
    DISPATCH_PAD_ALIGN(k_grid_sample_fwd, flat_grid(B, (int64_t)Ho * Wo), dim3(256), 0, (hipStream_t)stream, input,
                       in_strides, grid, out, C, Hi, Wi, Ho, Wo);
    E2E_LAUNCH_CHECK("e2e_grid_sample_fwd");
    return E2E_OK;
}

int e2e_grid_sample_bwd(const float* input, e2e_strides in_strides, const float* grid, const float* g_out,
                        float* g_grid, float* g_input, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        int padding_mode, int align_corners, void* stream) {
    E2E_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, E2E_ERR_ARG, "e2e_grid_sample_bwd: bad dims");
    E2E_REQUIRE(input && grid && g_out && g_grid, E2E_ERR_ARG, "e2e_grid_sample_bwd: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_grid_sample_bwd: padding_mode %d not supported (zeros|border)", padding_mode);
    DISPATCH_PAD_ALIGN(k_grid_sample_bwd, flat_grid(B, (int64_t)Ho * Wo), dim3(256), 0, (hipStream_t)stream, input,
                       in_strides, grid, g_out, g_grid, g_input, C, Hi, Wi, Ho, Wo, (unsigned long long*)nullptr);
    E2E_LAUNCH_CHECK("e2e_grid_sample_bwd");
    return E2E_OK;
}

__global__ void k_fixed48_to_float(const long long* __restrict__ fx, float* __restrict__ out, int64_t n) {
    const bool poisoned = fx[n] != 0;                     // a non-finite / out-of-range contribution was refused: the result is not a number
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = poisoned ? __uint_as_float(0x7FC00000u) : (float)((double)fx[i] * (1.0 / 281474976710656.0));
}

int e2e_grid_sample_bwd_exact(const float* input, e2e_strides in_strides, const float* grid, const float* g_out, float* g_grid,
                              long long* g_input_fixed, float* g_input, int B, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode,
                              int align_corners, void* stream) {
    E2E_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, E2E_ERR_ARG, "e2e_grid_sample_bwd_exact: bad dims");
    E2E_REQUIRE(input && grid && g_out && g_grid && g_input_fixed && g_input, E2E_ERR_ARG, "e2e_grid_sample_bwd_exact: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_grid_sample_bwd_exact: padding_mode %d not supported (zeros|border)", padding_mode);
    const int64_t n = (int64_t)B * C * Hi * Wi;
    E2E_REQUIRE(hipMemsetAsync(g_input_fixed, 0, (size_t)(n + 1) * sizeof(long long), (hipStream_t)stream) == hipSuccess, E2E_ERR_LAUNCH,
                "e2e_grid_sample_bwd_exact: hipMemsetAsync of the fixed-point scratch failed");
    DISPATCH_PAD_ALIGN(k_grid_sample_bwd, flat_grid(B, (int64_t)Ho * Wo), dim3(256), 0, (hipStream_t)stream, input,
                       in_strides, grid, g_out, g_grid, g_input, C, Hi, Wi, Ho, Wo, (unsigned long long*)g_input_fixed);
    hipLaunchKernelGGL(k_fixed48_to_float, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       g_input_fixed, g_input, n);
    E2E_LAUNCH_CHECK("e2e_grid_sample_bwd_exact");
    return E2E_OK;
}

int e2e_photometric_fwd(const float* x, e2e_strides xs, const float* y, e2e_strides ys, float* ssim_out,
                        float* pmap_out, int B, int C, int H, int W, void* stream) {
    CHECK_DIMS("e2e_photometric_fwd");
    E2E_REQUIRE(C > 0 && x && y && (ssim_out || pmap_out), E2E_ERR_ARG, "e2e_photometric_fwd: null pointer / C<=0");
    hipLaunchKernelGGL(k_photometric_fwd, tile_grid(B, H, W), dim3(TW, TH), 0, (hipStream_t)stream, x, xs, y, ys, ssim_out, pmap_out, C, H, W);
    E2E_LAUNCH_CHECK("e2e_photometric_fwd");
    return E2E_OK;
}

int e2e_photometric_bwd(const float* x, e2e_strides xs, const float* y, e2e_strides ys, const float* g_pmap,
                        const float* g_ssim, float* g_x, int B, int C, int H, int W, void* stream) {
    CHECK_DIMS("e2e_photometric_bwd");
    E2E_REQUIRE(C > 0 && x && y && g_x && (g_pmap || g_ssim), E2E_ERR_ARG, "e2e_photometric_bwd: null pointer / C<=0");
    hipLaunchKernelGGL(k_photometric_bwd, tile_grid(B, H, W), dim3(TW, TH), 0, (hipStream_t)stream, x, xs, y, ys, g_pmap, g_ssim, g_x, C, H, W);
    E2E_LAUNCH_CHECK("e2e_photometric_bwd");
    return E2E_OK;
}

int64_t e2e_warp_photo_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const dim3 g = tile_grid(B, H, W);
    return 2ll * g.x * g.y * g.z;
}

int e2e_warp_photo_fwd(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                       const float* K, const float* inv_K, const float* T, float* synth, float* valid, float* pmap,
                       int use_mask, int padding_mode, int reg_kind, const float* reg_init_tgt,
                       const float* reg_init_src, const float* depth_src, float* loss_out, float* workspace, int B,
                       int H, int W, void* stream) {
    CHECK_DIMS("e2e_warp_photo_fwd");
    E2E_REQUIRE(depth_tgt && src && tgt && K && inv_K && T && synth && valid && loss_out && workspace, E2E_ERR_ARG,
                "e2e_warp_photo_fwd: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_warp_photo_fwd: padding_mode %d not supported (zeros|border)", padding_mode);
    E2E_REQUIRE(reg_kind >= 0 && reg_kind <= 2, E2E_ERR_ARG, "e2e_warp_photo_fwd: reg_kind %d (0 none, 1 l1, 2 l2)", reg_kind);
    E2E_REQUIRE(!reg_kind || (reg_init_tgt && reg_init_src && depth_src), E2E_ERR_ARG, "e2e_warp_photo_fwd: regulariser buffers missing");
    const dim3 g = tile_grid(B, H, W);
    const int nblk = g.x * g.y * g.z;
    if (padding_mode == E2E_PADDING_BORDER)
        hipLaunchKernelGGL(k_warp_photo_fwd<E2E_PAD_BORDER>, g, dim3(TW, TH), 0, (hipStream_t)stream, depth_tgt, src, ss,
                           tgt, ts, K, inv_K, T, synth, valid, pmap, use_mask, reg_kind, reg_init_tgt, reg_init_src,
                           depth_src, workspace, H, W);
    else
        hipLaunchKernelGGL(k_warp_photo_fwd<E2E_PAD_ZEROS>, g, dim3(TW, TH), 0, (hipStream_t)stream, depth_tgt, src, ss,
                           tgt, ts, K, inv_K, T, synth, valid, pmap, use_mask, reg_kind, reg_init_tgt, reg_init_src,
                           depth_src, workspace, H, W);
    E2E_LAUNCH_CHECK("e2e_warp_photo_fwd");
    const float scale = (float)(1.0 / ((double)B * H * W));
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nblk, reg_kind ? 2 : 1, scale, loss_out);
    E2E_LAUNCH_CHECK("e2e_warp_photo_fwd(reduce)");
    return E2E_OK;
}

int e2e_warp_photo_bwd(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                       const float* K, const float* inv_K, const float* T, const float* synth, const float* valid,
                       int use_mask, int padding_mode, int reg_kind, const float* reg_init_tgt,
                       const float* reg_init_src, const float* depth_src, const float* g_loss, float* g_depth_tgt,
                       float* g_depth_src, int B, int H, int W, void* stream) {
    CHECK_DIMS("e2e_warp_photo_bwd");
    E2E_REQUIRE(depth_tgt && src && tgt && K && inv_K && T && synth && valid && g_loss && g_depth_tgt, E2E_ERR_ARG,
                "e2e_warp_photo_bwd: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_warp_photo_bwd: padding_mode %d not supported (zeros|border)", padding_mode);
    E2E_REQUIRE(reg_kind >= 0 && reg_kind <= 2, E2E_ERR_ARG, "e2e_warp_photo_bwd: reg_kind %d (0 none, 1 l1, 2 l2)", reg_kind);
    E2E_REQUIRE(!reg_kind || (reg_init_tgt && reg_init_src && depth_src && g_depth_src), E2E_ERR_ARG, "e2e_warp_photo_bwd: regulariser buffers missing");
    const dim3 g = tile_grid(B, H, W);
    if (padding_mode == E2E_PADDING_BORDER)
        hipLaunchKernelGGL(k_warp_photo_bwd<E2E_PAD_BORDER>, g, dim3(TW, TH), 0, (hipStream_t)stream, depth_tgt, src, ss,
                           tgt, ts, K, inv_K, T, synth, valid, use_mask, reg_kind, reg_init_tgt, reg_init_src, depth_src,
                           g_loss, g_depth_tgt, g_depth_src, B, H, W);
    else
        hipLaunchKernelGGL(k_warp_photo_bwd<E2E_PAD_ZEROS>, g, dim3(TW, TH), 0, (hipStream_t)stream, depth_tgt, src, ss,
                           tgt, ts, K, inv_K, T, synth, valid, use_mask, reg_kind, reg_init_tgt, reg_init_src, depth_src,
                           g_loss, g_depth_tgt, g_depth_src, B, H, W);
    E2E_LAUNCH_CHECK("e2e_warp_photo_bwd");
    return E2E_OK;
}

}  // extern "C"
