// aux_losses.hip -- the off-by-default losses of loss/losses.py and train_depth.py for gfx950.
// Every loss is ONE pass that produces the loss's partial sums AND the gradient map scaled for a unit upstream
// gradient (the gradient of a mean does not depend on its value), plus a fixed-order second stage for the scalar(s):
//   * edge-aware disparity smoothness               losses.py:119-132
//   * geometric consistency of two depth maps        losses.py:84-95 (the `mask.sum() > 10000` gate stays on the device)
//   * sparse ground-truth depth loss                 losses.py:151-160
//   * minimum reprojection over stacked error maps   train_depth.py:657-661 (torch.min over dim 1, then mean)
//   * dual-disparity blend                           train_depth.py:224-237
// All are HBM-bound streaming kernels; algorithmic bytes are listed per kernel.  Sums are per-workgroup partials added
// in a fixed order in double (bitwise reproducible, no floating-point atomics).
#include "e2e_common.h"

#define AT 256
#define APARTS 512

static inline int agrid(int64_t n, int cap = APARTS) {
    int64_t g = (n + AT - 1) / AT;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// out[s] = scale[s] * sum_i partials[s*nparts + i]   (one wave per set, fixed order)
__global__ __launch_bounds__(64) void k_aux_final(const float* __restrict__ partials, int nparts, int nsets, double s0, double s1,
                                                  float* __restrict__ out) {
    for (int s = 0; s < nsets; ++s) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) acc += (double)partials[s * nparts + i];
        acc = wave_sum_d(acc);
        if (threadIdx.x == 0) out[s] = (float)(acc * (s == 0 ? s0 : s1));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// smoothness: mean_x(|d(y,x) - d(y,x+1)| * exp(-mean_c |I(y,x) - I(y,x+1)|)) + the same along y.
// in: disp 4N + img 4CN (each read ~3x through L2); out: g_disp 4N.  One thread per pixel computes the four edges it
// touches (right/down for the loss sum, plus left/up for the gradient) -- no atomics, no second pass for the gradient.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_smoothness(const float* __restrict__ disp, const float* __restrict__ img, e2e_strides is, int B, int C,
                                                   int H, int W, float gx_scale, float gy_scale, float* __restrict__ g_disp,
                                                   float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    const int64_t N = (int64_t)B * H * W;
    float sx = 0.f, sy = 0.f;
    const float invC = 1.f / (float)C;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < N; i += (int64_t)gridDim.x * AT) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        const float* ib = img + b * is.sb + y * is.sh + x * is.sw;
        const float d = disp[i];
        float g = 0.f;
        // weight of the edge between this pixel and its neighbour at offset `o` (elements of the image)
        auto edge_w = [&](int64_t o) {
            float a = 0.f;
            for (int c = 0; c < C; ++c) a += fabsf(ib[c * is.sc] - ib[c * is.sc + o]);
            return expf(-(a * invC));
        };
        auto sgn = [](float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); };
        if (x + 1 < W) {
            const float w = edge_w(is.sw), e = d - disp[i + 1];
            sx += fabsf(e) * w;
            g += gx_scale * sgn(e) * w;
        }
        if (x > 0) {
            const float w = edge_w(-is.sw), e = disp[i - 1] - d;
            g -= gx_scale * sgn(e) * w;
        }
        if (y + 1 < H) {
            const float w = edge_w(is.sh), e = d - disp[i + W];
            sy += fabsf(e) * w;
            g += gy_scale * sgn(e) * w;
        }
        if (y > 0) {
            const float w = edge_w(-is.sh), e = disp[i - W] - d;
            g -= gy_scale * sgn(e) * w;
        }
        if (g_disp) g_disp[i] = g;
    }
    const float a = block_sum(sx, red);
    __syncthreads();
    const float c2 = block_sum(sy, red);
    if (threadIdx.x == 0) { partials[blockIdx.x] = a; partials[gridDim.x + blockIdx.x] = c2; }
}

// ---------------------------------------------------------------------------------------------------------------------
// geometric consistency: f = clamp(|a-b| / (a+b), 0, 1); loss = sum(f m) / sum(m) if sum(m) > 10000 else 0.
// Pass 1: partial sums of (f m, m) and the per-element derivatives (da, db) for unit upstream / unit normaliser;
// pass 2 (k_geom_scale) multiplies by 1/sum(m) (or 0 when the gate is closed) -- the normaliser is only known then.
// in: 12n, out: 8n (+ 16n in the scale pass).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_geom(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m, int64_t n,
                                             float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    float s = 0.f, sm = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const float x = a[i], y = b[i], w = m[i];
        const float den = x + y, ad = fabsf(x - y);
        const float f = ad / den;
        const float fc = fminf(fmaxf(f, 0.f), 1.f);
        s += fc * w;
        sm += w;
        if (ga) {
            const bool pass = f >= 0.f && f <= 1.f;             // torch.clamp backward: gradient inside [min, max]
            const float sg = (x > y) ? 1.f : ((x < y) ? -1.f : 0.f);
            const float inv2 = 1.f / (den * den);
            ga[i] = pass ? w * (sg * den - ad) * inv2 : 0.f;
            gb[i] = pass ? w * (-sg * den - ad) * inv2 : 0.f;
        }
    }
    const float t0 = block_sum(s, red);
    __syncthreads();
    const float t1 = block_sum(sm, red);
    if (threadIdx.x == 0) { partials[blockIdx.x] = t0; partials[gridDim.x + blockIdx.x] = t1; }
}

__global__ __launch_bounds__(64) void k_geom_final(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
    double acc[2];
    for (int s = 0; s < 2; ++s) {
        double v = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) v += (double)partials[s * nparts + i];
        acc[s] = wave_sum_d(v);
    }
    if (threadIdx.x == 0) {
        const bool open = acc[1] > 10000.0;                       // losses.py:90
        out[0] = open ? (float)(acc[0] / acc[1]) : 0.f;
        out[1] = (float)acc[1];
        out[2] = open ? (float)(1.0 / acc[1]) : 0.f;              // gradient normaliser
    }
}

__global__ __launch_bounds__(AT) void k_scale2(float* __restrict__ ga, float* __restrict__ gb, int64_t n, const float* __restrict__ stats) {
    const float k = stats[2];
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        ga[i] *= k;
        gb[i] *= k;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// sparse ground truth: mean |p m - g|; d/dp = sign(p m - g) m / n.   in 12n, out 4n
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_masked_l1(const float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ m, int64_t n,
                                                  float gscale, float* __restrict__ gp, float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const float w = m[i], e = p[i] * w - g[i];
        s += fabsf(e);
        if (gp) gp[i] = gscale * w * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// minimum reprojection: mean over (b,y,x) of min_c e[b,c,y,x]; the gradient goes to the FIRST minimal channel.
// in 4Cn, out 4Cn
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_min_reproj(const float* __restrict__ e, int B, int C, int64_t HW, float gscale, float* __restrict__ ge,
                                                   float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    const int64_t n = (int64_t)B * HW;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const int64_t b = i / HW, r = i - b * HW;
        const float* eb = e + b * C * HW + r;
        float best = eb[0];
        int bc = 0;
        for (int c = 1; c < C; ++c) {
            const float v = eb[c * HW];
            if (v < best || (v != v && best == best)) { best = v; bc = c; }     // NaN propagates like torch.min
        }
        s += best;
        if (ge) {
            float* gb = ge + b * C * HW + r;
            for (int c = 0; c < C; ++c) gb[c * HW] = (c == bc) ? gscale : 0.f;
        }
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// dual-disparity blend: left = d[0], right = flip_w(d[1]); l = 1 - clip(20 (y/(H-1) - 0.05), 0, 1) (the reference's mesh
// varies along ROWS, so flipping it along the width changes nothing: r = l); out = r left + l right + (1 - l - r) middle.
// The expression is evaluated term by term in the reference's order (algebraically it collapses to the mean).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_disp_blend(const float* __restrict__ d, int H, int W, float* __restrict__ out) {
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float left = d[i], right = d[n + (int64_t)y * W + (W - 1 - x)];
        const float middle = 0.5f * (left + right);
        const float t = (H > 1) ? (float)y / (float)(H - 1) : 0.f;
        const float l = 1.f - fminf(fmaxf(20.f * (t - 0.05f), 0.f), 1.f);
        out[i] = l * left + l * right + (1.f - l - l) * middle;
    }
}

// adjoint: g_d[0][y,x] = g (l + 0.5 (1 - 2l)) ; g_d[1][y, W-1-x] = the same
__global__ __launch_bounds__(AT) void k_disp_blend_bwd(const float* __restrict__ g, int H, int W, float* __restrict__ gd) {
    const int64_t n = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float t = (H > 1) ? (float)y / (float)(H - 1) : 0.f;
        const float l = 1.f - fminf(fmaxf(20.f * (t - 0.05f), 0.f), 1.f);
        const float k = l + 0.5f * (1.f - l - l);
        gd[i] = g[i] * k;
        gd[n + (int64_t)y * W + (W - 1 - x)] = g[i] * k;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// masked mean: loss = sum_i [gate_i != 0] v_i / #{gate_i != 0};  d loss / d v_i = weight [gate_i != 0] / count.
// The 3-D point loss (online_adaption.py:638-645, losses.py:57-63) averages the nearest-neighbour distances of the pixels
// with valid depth; the reference boolean-indexes those rows out (a dynamic shape and a host sync), here every pixel keeps
// its row and the gate (the depth map itself) masks it.  in 8n (+ 4n for the gradient pass), out 4n
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_masked_mean_partial(const float* __restrict__ v, const float* __restrict__ gate, int64_t n,
                                                            float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    float s = 0.f, c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const bool on = gate[i] != 0.f;
        s += on ? v[i] : 0.f;
        c += on ? 1.f : 0.f;
    }
    const float t0 = block_sum(s, red);
    __syncthreads();
    const float t1 = block_sum(c, red);
    if (threadIdx.x == 0) { partials[blockIdx.x] = t0; partials[gridDim.x + blockIdx.x] = t1; }
}

__global__ __launch_bounds__(64) void k_masked_mean_final(const float* __restrict__ partials, int nparts, float weight, float* __restrict__ out3) {
    double acc[2];
    for (int s = 0; s < 2; ++s) {
        double v = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) v += (double)partials[s * nparts + i];
        acc[s] = wave_sum_d(v);
    }
    if (threadIdx.x == 0) {
        out3[0] = acc[1] > 0.0 ? (float)(acc[0] / acc[1]) : (float)(0.0 / 0.0);     // mean of an empty selection is NaN, as in torch
        out3[1] = (float)acc[1];
        out3[2] = acc[1] > 0.0 ? (float)((double)weight / acc[1]) : 0.f;
    }
}

__global__ __launch_bounds__(AT) void k_masked_mean_grad(const float* __restrict__ gate, int64_t n, const float* __restrict__ out3,
                                                         float* __restrict__ g) {
    const float k = out3[2];
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) g[i] = gate[i] != 0.f ? k : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// small image-space helpers of train_depth.py's operator-by-operator loss assembly (off the fused path):
//   * prediction * valid_mask with the mask broadcast over channels (train_depth.py:713-714), strided input read in place
//   * mean over the channel axis of stacked photometric maps (`photmetric.mean(1, keepdim=True)`, :630) and its adjoint
//   * disp / (mean_hw(disp) + 1e-7) (:768-770) and its adjoint (two fixed-order reductions per image)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AT) void k_mask_mul(const float* __restrict__ x, e2e_strides xs, const float* __restrict__ mask, int B, int C, int H, int W,
                                                 float* __restrict__ out) {
    const int64_t n = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        int64_t t = i;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H); t /= H;
        const int c = (int)(t % C); t /= C;
        const int b = (int)t;
        out[i] = x[b * xs.sb + c * xs.sc + h * xs.sh + w * xs.sw] * mask[((int64_t)b * H + h) * W + w];
    }
}

// fwd: out[b,0,p] = mean_c x[b,c,p] ; bwd (adjoint != 0): out[b,c,p] = x[b,0,p] / C
__global__ __launch_bounds__(AT) void k_channel_mean(const float* __restrict__ x, int B, int C, int64_t HW, int adjoint, float* __restrict__ out) {
    const int64_t n = adjoint ? (int64_t)B * C * HW : (int64_t)B * HW;
    const float inv = 1.f / (float)C;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        if (adjoint) {
            const int64_t b = i / ((int64_t)C * HW), p = i % HW;
            out[i] = x[b * HW + p] * inv;
        } else {
            const int64_t b = i / HW, p = i - b * HW;
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += x[(b * C + c) * HW + p];
            out[i] = s * inv;
        }
    }
}

// per image b: partial sums of a[i] (and of a[i] * b2[i] when b2 != NULL) over its HW elements; grid = (parts, B)
__global__ __launch_bounds__(AT) void k_image_sums(const float* __restrict__ a, const float* __restrict__ b2, int64_t HW, float* __restrict__ partials) {
    __shared__ float red[AT / 64];
    const float* ab = a + (int64_t)blockIdx.y * HW;
    const float* bb = b2 ? b2 + (int64_t)blockIdx.y * HW : nullptr;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < HW; i += (int64_t)gridDim.x * AT) s += bb ? ab[i] * bb[i] : ab[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = t;
}
__global__ __launch_bounds__(64) void k_image_sums_final(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) v += (double)partials[blockIdx.x * nparts + i];
    v = wave_sum_d(v);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)v;
}
// fwd: out = d / (mean + 1e-7), mean = sum[b] / HW ; bwd: out = g / (m + eps) - sum(g d)[b] / (HW (m + eps)^2)
__global__ __launch_bounds__(AT) void k_mean_normalize(const float* __restrict__ d, const float* __restrict__ g, const float* __restrict__ sum_d,
                                                       const float* __restrict__ sum_gd, int B, int64_t HW, float* __restrict__ out) {
    const int64_t n = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * AT + threadIdx.x; i < n; i += (int64_t)gridDim.x * AT) {
        const int b = (int)(i / HW);
        const float den = sum_d[b] / (float)HW + 1e-7f;
        out[i] = g ? g[i] / den - sum_gd[b] / ((float)HW * den * den) : d[i] / den;
    }
}

extern "C" {

int e2e_mask_mul(const float* x, e2e_strides x_strides, const float* mask, int B, int C, int H, int W, float* out, void* stream) {
    E2E_REQUIRE(x && mask && out && B > 0 && C > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_mask_mul: bad argument");
    hipLaunchKernelGGL(k_mask_mul, dim3(agrid((int64_t)B * C * H * W, 2048)), dim3(AT), 0, (hipStream_t)stream, x, x_strides, mask, B, C, H, W, out);
    E2E_LAUNCH_CHECK("e2e_mask_mul");
    return E2E_OK;
}

int e2e_channel_mean(const float* x, int B, int C, int H, int W, int adjoint, float* out, void* stream) {
    E2E_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_channel_mean: bad argument");
    const int64_t n = (int64_t)B * (adjoint ? C : 1) * H * W;
    hipLaunchKernelGGL(k_channel_mean, dim3(agrid(n, 2048)), dim3(AT), 0, (hipStream_t)stream, x, B, C, (int64_t)H * W, adjoint, out);
    E2E_LAUNCH_CHECK("e2e_channel_mean");
    return E2E_OK;
}

/* workspace: B * (2 * 64 + 2) floats.  g == NULL: forward (out = d / (mean_hw d + 1e-7)); else the adjoint for upstream gradient g. */
int e2e_mean_normalize(const float* d, const float* g, int B, int H, int W, float* out, float* workspace, void* stream) {
    E2E_REQUIRE(d && out && workspace && B > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_mean_normalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    const int parts = 64;
    float* p0 = workspace;                    // [B][parts]
    float* p1 = workspace + (int64_t)B * parts;
    float* s0 = p1 + (int64_t)B * parts;      // [B]
    float* s1 = s0 + B;
    hipLaunchKernelGGL(k_image_sums, dim3(parts, B), dim3(AT), 0, st, d, (const float*)nullptr, HW, p0);
    hipLaunchKernelGGL(k_image_sums_final, dim3(B), dim3(64), 0, st, p0, parts, s0);
    if (g) {
        hipLaunchKernelGGL(k_image_sums, dim3(parts, B), dim3(AT), 0, st, g, d, HW, p1);
        hipLaunchKernelGGL(k_image_sums_final, dim3(B), dim3(64), 0, st, p1, parts, s1);
    }
    hipLaunchKernelGGL(k_mean_normalize, dim3(agrid((int64_t)B * HW, 2048)), dim3(AT), 0, st, d, g, s0, s1, B, HW, out);
    E2E_LAUNCH_CHECK("e2e_mean_normalize");
    return E2E_OK;
}

int e2e_masked_mean_lossgrad(const float* values, const float* gate, int64_t n, float weight, float* out3, float* g_values, float* workspace,
                             void* stream) {
    E2E_REQUIRE(values && gate && out3 && workspace && n > 0, E2E_ERR_ARG, "e2e_masked_mean_lossgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = agrid(n);
    hipLaunchKernelGGL(k_masked_mean_partial, dim3(g), dim3(AT), 0, st, values, gate, n, workspace);
    hipLaunchKernelGGL(k_masked_mean_final, dim3(1), dim3(64), 0, st, workspace, g, weight, out3);
    if (g_values) hipLaunchKernelGGL(k_masked_mean_grad, dim3(agrid(n, 2048)), dim3(AT), 0, st, gate, n, out3, g_values);
    E2E_LAUNCH_CHECK("e2e_masked_mean_lossgrad");
    return E2E_OK;
}

int64_t e2e_aux_workspace_floats(void) { return 2 * APARTS; }

int e2e_smoothness_lossgrad(const float* disp, const float* img, e2e_strides img_strides, int B, int C, int H, int W, float* loss_out,
                            float* g_disp, float* workspace, void* stream) {
    E2E_REQUIRE(disp && img && loss_out && workspace && B > 0 && C > 0 && H > 1 && W > 1, E2E_ERR_ARG, "e2e_smoothness_lossgrad: bad argument (H, W >= 2)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    const int g = agrid(n);
    const double nx = (double)B * H * (W - 1), ny = (double)B * (H - 1) * W;
    hipLaunchKernelGGL(k_smoothness, dim3(g), dim3(AT), 0, st, disp, img, img_strides, B, C, H, W, (float)(1.0 / nx), (float)(1.0 / ny), g_disp, workspace);
    hipLaunchKernelGGL(k_aux_final, dim3(1), dim3(64), 0, st, workspace, g, 2, 1.0 / nx, 1.0 / ny, loss_out);   // {x term, y term}
    E2E_LAUNCH_CHECK("e2e_smoothness_lossgrad");
    return E2E_OK;
}

int e2e_geometric_consistency_lossgrad(const float* warped_depth, const float* interpolated_depth, const float* mask, int64_t n,
                                       float* stats_out3, float* g_warped, float* g_interpolated, float* workspace, void* stream) {
    E2E_REQUIRE(warped_depth && interpolated_depth && mask && stats_out3 && workspace && n > 0 && ((g_warped == nullptr) == (g_interpolated == nullptr)),
                E2E_ERR_ARG, "e2e_geometric_consistency_lossgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = agrid(n);
    hipLaunchKernelGGL(k_geom, dim3(g), dim3(AT), 0, st, warped_depth, interpolated_depth, mask, n, g_warped, g_interpolated, workspace);
    hipLaunchKernelGGL(k_geom_final, dim3(1), dim3(64), 0, st, workspace, g, stats_out3);
    if (g_warped) hipLaunchKernelGGL(k_scale2, dim3(agrid(n, 2048)), dim3(AT), 0, st, g_warped, g_interpolated, n, stats_out3);
    E2E_LAUNCH_CHECK("e2e_geometric_consistency_lossgrad");
    return E2E_OK;
}

int e2e_masked_l1_lossgrad(const float* prediction, const float* sparse_gt, const float* sparse_mask, int64_t n, float* loss_out,
                           float* g_prediction, float* workspace, void* stream) {
    E2E_REQUIRE(prediction && sparse_gt && sparse_mask && loss_out && workspace && n > 0, E2E_ERR_ARG, "e2e_masked_l1_lossgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = agrid(n);
    hipLaunchKernelGGL(k_masked_l1, dim3(g), dim3(AT), 0, st, prediction, sparse_gt, sparse_mask, n, (float)(1.0 / (double)n), g_prediction, workspace);
    hipLaunchKernelGGL(k_aux_final, dim3(1), dim3(64), 0, st, workspace, g, 1, 1.0 / (double)n, 0.0, loss_out);
    E2E_LAUNCH_CHECK("e2e_masked_l1_lossgrad");
    return E2E_OK;
}

int e2e_min_reprojection_lossgrad(const float* errors, int B, int C, int H, int W, float* loss_out, float* g_errors, float* workspace, void* stream) {
    E2E_REQUIRE(errors && loss_out && workspace && B > 0 && C > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_min_reprojection_lossgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * W;
    const int g = agrid(n);
    hipLaunchKernelGGL(k_min_reproj, dim3(g), dim3(AT), 0, st, errors, B, C, (int64_t)H * W, (float)(1.0 / (double)n), g_errors, workspace);
    hipLaunchKernelGGL(k_aux_final, dim3(1), dim3(64), 0, st, workspace, g, 1, 1.0 / (double)n, 0.0, loss_out);
    E2E_LAUNCH_CHECK("e2e_min_reprojection_lossgrad");
    return E2E_OK;
}

int e2e_disp_blend_fwd(const float* disp_pair, int H, int W, float* out, void* stream) {
    E2E_REQUIRE(disp_pair && out && H > 0 && W > 0, E2E_ERR_ARG, "e2e_disp_blend_fwd: bad argument");
    hipLaunchKernelGGL(k_disp_blend, dim3(agrid((int64_t)H * W, 2048)), dim3(AT), 0, (hipStream_t)stream, disp_pair, H, W, out);
    E2E_LAUNCH_CHECK("e2e_disp_blend_fwd");
    return E2E_OK;
}

int e2e_disp_blend_bwd(const float* g_out, int H, int W, float* g_disp_pair, void* stream) {
    E2E_REQUIRE(g_out && g_disp_pair && H > 0 && W > 0, E2E_ERR_ARG, "e2e_disp_blend_bwd: bad argument");
    hipLaunchKernelGGL(k_disp_blend_bwd, dim3(agrid((int64_t)H * W, 2048)), dim3(AT), 0, (hipStream_t)stream, g_out, H, W, g_disp_pair);
    E2E_LAUNCH_CHECK("e2e_disp_blend_bwd");
    return E2E_OK;
}

}  // extern "C"
