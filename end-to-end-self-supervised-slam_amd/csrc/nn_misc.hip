// nn_misc.hip -- the non-GEMM layers of the depth network for gfx950 (all NHWC, HBM-bound streaming kernels):
//   * 3x3 / stride 2 / pad 1 max-pool of the ResNet stem, forward and INDEX-FREE backward (every input element
//     re-derives the arg-max of the <= 4 windows that contain it and gathers their gradients: no index tensor, no
//     atomics)                                                          -- networks.py:53 (encoder.maxpool)
//   * eval-mode BatchNorm whose affine parameters still train (the reference freezes parameters by the substring
//     "bn", so `downsample.1` stays trainable: online_adaption.py:182-184): fold (gamma, beta, running stats) ->
//     (scale, shift); y = z * scale + shift; d gamma / d beta by a two-stage fixed-order reduction
//   * the same affine with C = 1 is Conv1x1(1, 1, bias) / ScaleLayer of the scale-learning experiments
//     (networks.py:191-215, absolute_scale.py:207-240)
//   * nearest x2 upsample + channel concat (the stand-alone `upsample()` helper, networks.py:218-221)
#include "e2e_common.h"

#include <math.h>

typedef float f4v __attribute__((ext_vector_type(4)));

static inline int mgrid(int64_t n, int cap = 4096) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------------------------------
// max-pool 3x3 / 2 / pad 1.  torch scans the window row-major and keeps the FIRST maximum (`val > maxval || isnan(val)`),
// so ties -- frequent behind a ReLU -- send the gradient to the first maximal element; the backward below re-derives
// exactly that choice.  NHWC with C % 4 == 0 (the stem has 64 channels).
// ---------------------------------------------------------------------------------------------------------------------
// forward: one thread = one output pixel x 4 channels (float4 loads along the NHWC channel axis)
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int c4 = (int)(t % C4); t /= C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho); t /= Ho;
        const int b = (int)t;
        f4v best = (f4v){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = 2 * oh - 1 + kh;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = 2 * ow - 1 + kw;
                if (xx < 0 || xx >= W) continue;
                const f4v v = *(const f4v*)(x + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) best[e] = v[e];
            }
        }
        *(f4v*)(y + i * 4) = best;
    }
}

// backward: one thread = one 2x2 block of INPUT pixels x 4 channels.  The block (by, bx) = pixels (2by..2by+1, 2bx..2bx+1) is
// touched by exactly the four windows (by..by+1, bx..bx+1), which together cover the 5x5 patch of rows 2by-1..2by+3: the
// patch is loaded once (25 float4), each window's FIRST maximum is re-derived in ATen's scan order, and every pixel of the
// block gathers the gradients of the windows whose maximum it is (no index tensor, no atomics, fixed order of the <= 4
// addends).  mul_relu: the result is multiplied by [x > 0] (gradient through the stem's ReLU, taken from its output);
// accumulate: added to dx.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B,
                                                     int H, int W, int C, int Ho, int Wo, int accumulate, int mul_relu) {
    const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const int64_t total = (int64_t)B * Hb * Wb * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int c4 = (int)(t % C4); t /= C4;
        const int bx = (int)(t % Wb); t /= Wb;
        const int by = (int)(t % Hb); t /= Hb;
        const int b = (int)t;
        const float* xb = x + (int64_t)b * H * W * C + c4 * 4;
        f4v p[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int yy = 2 * by - 1 + r, xx = 2 * bx - 1 + c;
                p[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? *(const f4v*)(xb + ((int64_t)yy * W + xx) * C)
                                                                    : (f4v){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        f4v g[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) g[r][c] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wa = 0; wa < 2; ++wa)
#pragma unroll
            for (int wb = 0; wb < 2; ++wb) {
                const int oh = by + wa, ow = bx + wb;
                if (oh >= Ho || ow >= Wo) continue;
                const f4v gy = *(const f4v*)(dy + (((int64_t)b * Ho + oh) * Wo + ow) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // first maximum of window rows 2wa..2wa+2, cols 2wb..2wb+2 of the patch; out-of-image cells hold -inf and
                    // are skipped exactly like ATen skips them (the scan starts at the first valid cell)
                    float best = -INFINITY;
                    int arg = -1;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int yy = 2 * by - 1 + 2 * wa + kh, xx = 2 * bx - 1 + 2 * wb + kw;
                            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                            const float v = p[2 * wa + kh][2 * wb + kw][e];
                            if (arg < 0) arg = (2 * wa + kh) * 5 + 2 * wb + kw;
                            if (v > best || v != v) { best = v; arg = (2 * wa + kh) * 5 + 2 * wb + kw; }
                        }
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (arg == (1 + r) * 5 + 1 + c) g[r][c][e] += gy[e];
                }
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int yy = 2 * by + r, xx = 2 * bx + c;
                if (yy >= H || xx >= W) continue;
                f4v v = g[r][c];
                if (mul_relu)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p[1 + r][1 + c][e] > 0.f ? v[e] : 0.f;
                f4v* d = (f4v*)(dx + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
                if (accumulate) { const f4v o = *d; v = o + v; }
                *d = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-channel affine of an eval-mode BatchNorm with trainable gamma / beta
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                                 const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                 float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float r = 1.f / sqrtf(var[c] + eps);              // torch: weight / sqrt(running_var + eps)
    const float s = gamma[c] * r;
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
    if (rstd) rstd[c] = r;
}

__global__ __launch_bounds__(256) void k_affine_fwd(const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ res, int relu, float* __restrict__ y, int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = fmaf(z[i], scale[c], shift ? shift[c] : 0.f);
        if (res) v += res[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

// stage 1: block (cx, s) sums channels [64 cx, 64 cx + 64) over pixel slice s; thread = (channel lane, 1 of 4 pixel lanes)
// d gamma[c] = sum_p dy[p,c] * (z[p,c] - mean[c]) * rstd[c],  d beta[c] = sum_p dy[p,c]     (mean / rstd NULL: 0 / 1)
__global__ __launch_bounds__(256) void k_affine_bwd_partial(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int64_t P, int C, int64_t pix_per_slice,
                                                            float* __restrict__ partial /*[S][2][C]*/) {
    __shared__ float sg[4][64], sb[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t p0 = (int64_t)blockIdx.y * pix_per_slice, p1 = (p0 + pix_per_slice < P) ? p0 + pix_per_slice : P;
    float ag = 0.f, ab = 0.f;
    if (c < C) {
        const float m = mean ? mean[c] : 0.f, r = rstd ? rstd[c] : 1.f;
        for (int64_t p = p0 + pl; p < p1; p += 4) {
            const float g = dy[p * C + c];
            ag = fmaf(g, (z[p * C + c] - m) * r, ag);
            ab += g;
        }
    }
    sg[pl][cl] = ag;
    sb[pl][cl] = ab;
    __syncthreads();
    if (pl == 0 && c < C) {
        float* o = partial + (int64_t)blockIdx.y * 2 * C;
        o[c] = ((sg[0][cl] + sg[1][cl]) + sg[2][cl]) + sg[3][cl];
        o[C + c] = ((sb[0][cl] + sb[1][cl]) + sb[2][cl]) + sb[3][cl];
    }
}

// C == 1 (scale layer): a flat reduction, one partial pair per workgroup
__global__ __launch_bounds__(256) void k_affine_bwd_partial1(const float* __restrict__ dy, const float* __restrict__ z, int64_t P,
                                                             float* __restrict__ partial /*[S][2]*/) {
    __shared__ float red[4];
    float ag = 0.f, ab = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const float g = dy[p];
        ag = fmaf(g, z[p], ag);
        ab += g;
    }
    const float a = block_sum(ag, red);
    __syncthreads();
    const float b = block_sum(ab, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = a; partial[blockIdx.x * 2 + 1] = b; }
}

// stage 2: fixed-order sum over the slices, 64 channels x 4 slice lanes per workgroup (lane w sums s = w, w + 4, ...; the four
// partial sums are combined in a fixed order); accumulate != 0 adds to the gradient buffers (flat gradient bucket)
__global__ __launch_bounds__(256) void k_affine_bwd_final(const float* __restrict__ partial, int S, int C, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int accumulate) {
    __shared__ float sg[4][64], sb[4][64];
    const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float g = 0.f, b = 0.f;
    if (c < C)
        for (int s = w; s < S; s += 4) {
            g += partial[(int64_t)s * 2 * C + c];
            b += partial[(int64_t)s * 2 * C + C + c];
        }
    sg[w][cl] = g;
    sb[w][cl] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        g = ((sg[0][cl] + sg[1][cl]) + sg[2][cl]) + sg[3][cl];
        b = ((sb[0][cl] + sb[1][cl]) + sb[2][cl]) + sb[3][cl];
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + g : g;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + b : b;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// nearest x2 upsample of x (B,h,w,C1) [+ concat skip (B,2h,2w,C2)] -> (B,2h,2w,C1+C2)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample2_concat(const float* __restrict__ x, const float* __restrict__ skip, float* __restrict__ y, int B,
                                                          int h, int w, int C1, int C2) {
    const int H = 2 * h, W = 2 * w, C = C1 + C2;
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H); t /= H;
        const int b = (int)t;
        y[i] = (c < C1) ? x[(((int64_t)b * h + (yy >> 1)) * w + (xx >> 1)) * C1 + c]
                        : skip[(((int64_t)b * H + yy) * W + xx) * C2 + (c - C1)];
    }
}

#define AFF_SLICES 64

// Max-pool with the position of the maximum kept: one byte per output element (kh * 3 + kw of the FIRST maximum in ATen's scan
// order).  The backward then reads, per 2 x 2 input block and channel quad, the <= 4 windows' position bytes and gradients
// (~150 bytes) instead of re-scanning a 5 x 5 patch of the input (400 bytes: 59 us for the stem's 39 MB tensor).
__global__ __launch_bounds__(256) void k_maxpool_fwd_idx(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                         int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int c4 = (int)(t % C4); t /= C4;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho); t /= Ho;
        const int b = (int)t;
        f4v best = (f4v){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = 2 * oh - 1 + kh;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = 2 * ow - 1 + kw;
                if (xx < 0 || xx >= W) continue;
                const f4v v = *(const f4v*)(x + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (arg[e] < 0) arg[e] = kh * 3 + kw;           // the scan starts at the first valid cell
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; arg[e] = kh * 3 + kw; }
                }
            }
        }
        *(f4v*)(y + i * 4) = best;
        *(unsigned int*)(idx + i * 4) = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
    }
}

__global__ __launch_bounds__(256) void k_maxpool_bwd_idx(const float* __restrict__ x, const unsigned char* __restrict__ idx, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, int accumulate, int mul_relu) {
    const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const int64_t total = (int64_t)B * Hb * Wb * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int c4 = (int)(t % C4); t /= C4;
        const int bx = (int)(t % Wb); t /= Wb;
        const int by = (int)(t % Hb); t /= Hb;
        const int b = (int)t;
        f4v g[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) g[r][c] = (f4v){0.f, 0.f, 0.f, 0.f};
        // input block rows 2 by, 2 by + 1 sit at window rows (wa = 0: kh 1, 2), (wa = 1: kh 0 for the second row only); same for columns
#pragma unroll
        for (int wa = 0; wa < 2; ++wa)
#pragma unroll
            for (int wb = 0; wb < 2; ++wb) {
                const int oh = by + wa, ow = bx + wb;
                if (oh >= Ho || ow >= Wo) continue;
                const int64_t o = (((int64_t)b * Ho + oh) * Wo + ow) * C + c4 * 4;
                const f4v gy = *(const f4v*)(dy + o);
                const unsigned pk = *(const unsigned int*)(idx + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int a = (int)((pk >> (8 * e)) & 0xFFu);
                    const int kh = a / 3, kw = a - kh * 3;
                    const int r = 2 * wa + kh - 1, c = 2 * wb + kw - 1;     // position inside the 2 x 2 block (or outside it)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
                            if (r == rr && c == cc) g[rr][cc][e] += gy[e];
                }
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int yy = 2 * by + r, xx = 2 * bx + c;
                if (yy >= H || xx >= W) continue;
                const int64_t o = (((int64_t)b * H + yy) * W + xx) * C + c4 * 4;
                f4v v = g[r][c];
                if (mul_relu) {
                    const f4v xv = *(const f4v*)(x + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xv[e] > 0.f ? v[e] : 0.f;
                }
                f4v* d = (f4v*)(dx + o);
                if (accumulate) { const f4v old = *d; v = old + v; }
                *d = v;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Many device-to-device copies in one launch (e2e_copy_batched).  The forward-reuse of the refinement loop moves every layer's activations of
// one image from batch slot 1 to slot 0 at the start of a keyframe (NetPlan.move_slot): ~70 copies of 0.04 - 10 MB, 5 us each as separate
// copy-engine / blit launches.  A work item is 16 KB of one copy; a descriptor table in device memory says where each copy's items start.
// ---------------------------------------------------------------------------------------------------------------------
#define COPY_ITEM_BYTES 16384
__global__ __launch_bounds__(256) void k_copy_batched(const e2e_copy_desc* __restrict__ d, int n, long long total_items) {
    for (long long item = blockIdx.x; item < total_items; item += gridDim.x) {
        int lo = 0, hi = n - 1;                                   // last descriptor whose first_item <= item (workgroup-uniform)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (d[mid].first_item <= item) lo = mid; else hi = mid - 1;
        }
        const long long off = (item - d[lo].first_item) * COPY_ITEM_BYTES, left = d[lo].bytes - off;
        const uint4* s = (const uint4*)((const char*)d[lo].src + off);
        uint4* t = (uint4*)((char*)d[lo].dst + off);
        const int nq = (int)((left < COPY_ITEM_BYTES ? left : COPY_ITEM_BYTES) / 16);
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((int)threadIdx.x + j * 256 < nq) v[j] = s[threadIdx.x + j * 256];
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((int)threadIdx.x + j * 256 < nq) t[threadIdx.x + j * 256] = v[j];
    }
}

extern "C" {

int e2e_maxpool3x3s2_fwd_idx(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C, void* stream) {
    E2E_REQUIRE(x && y && argmax && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, E2E_ERR_ARG, "e2e_maxpool3x3s2_fwd_idx: bad argument (C %% 4 == 0)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_fwd_idx, dim3(mgrid((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, argmax, B, H, W, C, Ho, Wo);
    E2E_LAUNCH_CHECK("e2e_maxpool3x3s2_fwd_idx");
    return E2E_OK;
}

int e2e_maxpool3x3s2_bwd_idx(const float* x, const unsigned char* argmax, const float* dy, float* dx, int B, int H, int W, int C, int accumulate,
                             int mul_relu, void* stream) {
    E2E_REQUIRE(argmax && dy && dx && (x || !mul_relu) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, E2E_ERR_ARG,
                "e2e_maxpool3x3s2_bwd_idx: bad argument (C %% 4 == 0)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_bwd_idx, dim3(mgrid((int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       argmax, dy, dx, B, H, W, C, Ho, Wo, accumulate, mul_relu);
    E2E_LAUNCH_CHECK("e2e_maxpool3x3s2_bwd_idx");
    return E2E_OK;
}

int e2e_maxpool3x3s2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    E2E_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, E2E_ERR_ARG, "e2e_maxpool3x3s2_fwd: bad argument (C %% 4 == 0)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(mgrid((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo);
    E2E_LAUNCH_CHECK("e2e_maxpool3x3s2_fwd");
    return E2E_OK;
}

int e2e_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int accumulate, int mul_relu,
                         void* stream) {
    E2E_REQUIRE(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, E2E_ERR_ARG, "e2e_maxpool3x3s2_bwd: bad argument (C %% 4 == 0)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(mgrid((int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       dy, dx, B, H, W, C, Ho, Wo, accumulate, mul_relu);
    E2E_LAUNCH_CHECK("e2e_maxpool3x3s2_bwd");
    return E2E_OK;
}

int e2e_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale,
                float* shift, float* rstd, int C, void* stream) {
    E2E_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, E2E_ERR_ARG, "e2e_bn_fold: bad argument");
    hipLaunchKernelGGL(k_bn_fold, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale,
                       shift, rstd, C);
    E2E_LAUNCH_CHECK("e2e_bn_fold");
    return E2E_OK;
}

int e2e_affine_fwd(const float* z, const float* scale, const float* shift, const float* residual, int relu, float* y, int64_t n, int C,
                   void* stream) {
    E2E_REQUIRE(z && scale && y && n > 0 && C > 0 && n % C == 0, E2E_ERR_ARG, "e2e_affine_fwd: bad argument");
    hipLaunchKernelGGL(k_affine_fwd, dim3(mgrid(n)), dim3(256), 0, (hipStream_t)stream, z, scale, shift, residual, relu, y, n, C);
    E2E_LAUNCH_CHECK("e2e_affine_fwd");
    return E2E_OK;
}

int64_t e2e_affine_bwd_workspace_floats(int C) { return (int64_t)AFF_SLICES * 2 * (C > 0 ? C : 1); }

int e2e_affine_bwd(const float* dy, const float* z, const float* mean, const float* rstd, int64_t P, int C, float* dgamma, float* dbeta,
                   int accumulate, float* workspace, void* stream) {
    E2E_REQUIRE(dy && z && workspace && P > 0 && C > 0 && (dgamma || dbeta), E2E_ERR_ARG, "e2e_affine_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int S = AFF_SLICES;
    if (C == 1) {
        E2E_REQUIRE(!mean && !rstd, E2E_ERR_ARG, "e2e_affine_bwd: the single-channel form takes no statistics");
        if ((P + 255) / 256 < S) S = (int)((P + 255) / 256);
        hipLaunchKernelGGL(k_affine_bwd_partial1, dim3(S), dim3(256), 0, st, dy, z, P, workspace);
    } else {
        int64_t pps = (P + S - 1) / S;
        if (pps < 4) pps = 4;
        S = (int)((P + pps - 1) / pps);
        hipLaunchKernelGGL(k_affine_bwd_partial, dim3((C + 63) / 64, S), dim3(256), 0, st, dy, z, mean, rstd, P, C, pps, workspace);
    }
    hipLaunchKernelGGL(k_affine_bwd_final, dim3((C + 63) / 64), dim3(256), 0, st, workspace, S, C, dgamma, dbeta, accumulate);
    E2E_LAUNCH_CHECK("e2e_affine_bwd");
    return E2E_OK;
}

int e2e_upsample2_concat(const float* x, const float* skip, float* y, int B, int h, int w, int C1, int C2, void* stream) {
    E2E_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && C1 > 0 && C2 >= 0 && (C2 == 0 || skip), E2E_ERR_ARG, "e2e_upsample2_concat: bad argument");
    hipLaunchKernelGGL(k_upsample2_concat, dim3(mgrid((int64_t)B * 4 * h * w * (C1 + C2))), dim3(256), 0, (hipStream_t)stream, x, skip, y, B, h, w,
                       C1, C2);
    E2E_LAUNCH_CHECK("e2e_upsample2_concat");
    return E2E_OK;
}

/* fills first_item of n descriptors in HOST memory; returns the total number of work items, or -1 on a malformed descriptor
 * (NULL pointer, size not a positive multiple of 16, pointer not 16-byte aligned) */
long long e2e_copy_batch_prepare(e2e_copy_desc* descs_host, int n) {
    if (!descs_host || n <= 0) return -1;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        e2e_copy_desc& d = descs_host[i];
        if (!d.src || !d.dst || d.bytes <= 0 || d.bytes % 16 || ((uintptr_t)d.src & 15) || ((uintptr_t)d.dst & 15)) return -1;
        d.first_item = total;
        total += (d.bytes + COPY_ITEM_BYTES - 1) / COPY_ITEM_BYTES;
    }
    return total;
}

int e2e_copy_batched(const e2e_copy_desc* descs_dev, int n, long long total_items, void* stream) {
    E2E_REQUIRE(descs_dev && n > 0 && total_items > 0, E2E_ERR_ARG, "e2e_copy_batched: bad argument");
    hipLaunchKernelGGL(k_copy_batched, dim3((unsigned)(total_items < 8192 ? total_items : 8192)), dim3(256), 0, (hipStream_t)stream, descs_dev, n, total_items);
    E2E_LAUNCH_CHECK("e2e_copy_batched");
    return E2E_OK;
}

}  // extern "C"
