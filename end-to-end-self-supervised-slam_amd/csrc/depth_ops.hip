// depth_ops.hip -- the element-wise / reduction operators of a refinement step that sit between the depth
// network and the image-space losses, for gfx950:
//   * lower-median selection (torch.median) by 11/11/10-bit radix select     online_adaption.py:295,343
//   * disp -> depth, median scaling and its exact autograd chain            online_adaption.py:282,292-298
//   * depth regulariser as a stand-alone op                                 loss/losses.py:134-148
//   * depth metrics                                                          loss/losses.py:162-201
//   * fused Adam over a flat parameter buffer                               utils/training_utils.py:23-25
// All are HBM-bound streaming kernels (4-16 B per element per pass); reductions use integer atomics or
// fixed-order partial sums only, so results are bitwise reproducible.
#include "e2e_common.h"

#define DT 256

__device__ __forceinline__ unsigned int f2key(float f) {          // order-preserving float -> uint
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---------------------------------------------------------------------------------------------
// radix select, three digit passes: bits [31:21] (2048 bins), [20:10] (2048), [9:0] (1024).
// state (device, 8 uint32): [0]=prefix key so far, [1]=remaining rank, [2]=result key, [3]=result index
// Each histogram pass first derives (digit, remaining rank) of the PREVIOUS pass from its histogram, so
// there is no separate "select" launch between passes.
// ---------------------------------------------------------------------------------------------
#define MED_BINS 2048
struct MedState {
    unsigned int prefix, rank, key, index, count;      // index: smallest element holding the median value; count: how many hold it
};

// find the bin that holds rank `r` in hist[0..nb) (nb <= 2048): returns the bin, writes the rank inside the bin.
// Workgroup-cooperative: thread t owns bins [8t, 8t+8); block-wide exclusive scan of the per-thread sums; the one
// thread whose range contains r resolves the bin.  All threads must call it.  sh: 8 words.
__device__ unsigned int select_bin(const unsigned int* __restrict__ hist, int nb, unsigned int r, unsigned int* r_in, unsigned int* sh) {
    unsigned int c[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x * 8 + j;
        c[j] = (i < nb) ? hist[i] : 0u;
        s += c[j];
    }
    unsigned int x = s;                               // inclusive scan across the workgroup (4 waves)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int y = __shfl_up(x, o, 64);
        if ((threadIdx.x & 63) >= o) x += y;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh[2 + (threadIdx.x >> 6)] = x;
    __syncthreads();
    unsigned int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += sh[2 + w];
    const unsigned int excl = woff + x - s;
    if (r >= excl && r < excl + s) {                  // exactly one thread (bins are disjoint, total > r)
        unsigned int acc = excl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (r < acc + c[j]) { sh[0] = threadIdx.x * 8 + j; sh[1] = r - acc; break; }
            acc += c[j];
        }
    }
    __syncthreads();
    *r_in = sh[1];
    return sh[0];
}

// pass p (0,1,2): histogram of digit p over the elements whose higher digits equal the prefix so far
// RECIP (pass 0 of the scale chain): x is the DISPARITY; the pass computes delta = 1 / disp, stores it and histograms it -- one launch and
// one read of the tensor less than a separate reciprocal kernel (the same expression: the same bits)
template <int PASS, bool RECIP = false>
__global__ __launch_bounds__(DT) void k_median_hist(const float* __restrict__ x, int64_t n, unsigned int rank0,
                                                    unsigned int* __restrict__ hist /* 3 x 2048 */, MedState* __restrict__ st,
                                                    float* __restrict__ recip_out = nullptr) {
    __shared__ unsigned int lh[MED_BINS];
    __shared__ unsigned int sh[8];
    for (int i = threadIdx.x; i < MED_BINS; i += DT) lh[i] = 0;
    unsigned int prefix = 0;
    if (PASS == 1) {
        unsigned int rin;
        const unsigned int b = select_bin(hist, MED_BINS, rank0, &rin, sh);
        prefix = b << 21;
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->prefix = prefix; st->rank = rin; }
    } else if (PASS == 2) {
        unsigned int rin0, rin1;
        const unsigned int b0 = select_bin(hist, MED_BINS, rank0, &rin0, sh);
        __syncthreads();
        const unsigned int b1 = select_bin(hist + MED_BINS, MED_BINS, rin0, &rin1, sh);
        prefix = (b0 << 21) | (b1 << 10);
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->prefix = prefix; st->rank = rin1; }
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        float v = x[i];
        if (RECIP) { v = 1.0f / v; recip_out[i] = v; }
        const unsigned int k = f2key(v);
        if (PASS == 0) atomicAdd(&lh[k >> 21], 1u);
        else if (PASS == 1) { if ((k >> 21) == (prefix >> 21)) atomicAdd(&lh[(k >> 10) & 2047u], 1u); }
        else { if ((k >> 10) == (prefix >> 10)) atomicAdd(&lh[k & 1023u], 1u); }
    }
    __syncthreads();
    unsigned int* gh = hist + PASS * MED_BINS;
    for (int i = threadIdx.x; i < MED_BINS; i += DT)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// final: resolve the last digit, then the smallest index holding that key (torch returns one index of the median)
__global__ __launch_bounds__(DT) void k_median_final(const float* __restrict__ x, int64_t n, const unsigned int* __restrict__ hist,
                                                     MedState* __restrict__ st, float* __restrict__ value_out) {
    __shared__ unsigned int sh[8];
    unsigned int rin;
    const unsigned int b2 = select_bin(hist + 2 * MED_BINS, 1024, st->rank, &rin, sh);
    const unsigned int key = st->prefix | b2;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->key = key; *value_out = key2f(key); }
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT)
        if (f2key(x[i]) == key) { atomicMin(&st->index, (unsigned int)i); atomicAdd(&st->count, 1u); }
}

// the scale chain's last launch: the final digit of the median (every workgroup resolves it for itself from the third histogram: 1024
// bins), rho = m_gt / median, depth = rho * delta, and -- in the same pass over delta -- which elements hold the median (k_median_final's
// second half).  One launch and one read of delta less than k_median_final + k_scale_by_ratio.
__global__ __launch_bounds__(DT) void k_median_final_scale(const float* __restrict__ delta, int64_t n, const unsigned int* __restrict__ hist,
                                                           MedState* __restrict__ st, const float* __restrict__ m_gt, float* __restrict__ median_out,
                                                           float* __restrict__ depth, float* __restrict__ ratio_out) {
    __shared__ unsigned int sh[8];
    unsigned int rin;
    const unsigned int b2 = select_bin(hist + 2 * MED_BINS, 1024, st->rank, &rin, sh);
    const unsigned int key = st->prefix | b2;
    const float med = key2f(key);
    const float rho = m_gt[0] / med;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->key = key;
        *median_out = med;
        if (ratio_out) ratio_out[0] = rho;
    }
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float d = delta[i];
        depth[i] = d * rho;
        if (f2key(d) == key) { atomicMin(&st->index, (unsigned int)i); atomicAdd(&st->count, 1u); }
    }
}

__global__ void k_median_init(unsigned int* hist, MedState* st) {
    for (int i = threadIdx.x; i < 3 * MED_BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { st->prefix = 0; st->rank = 0; st->key = 0; st->index = 0xFFFFFFFFu; st->count = 0; }
}

// ---------------------------------------------------------------------------------------------
// disp -> depth and the median-scale chain
//   delta = 1/disp ; rho = m_gt / median(delta) ; depth = rho * delta           (forward: folded into the median's first and last pass)
//   g_delta = rho*g + [delta_i == median] * (-(rho/median) * sum(g*delta)) / #{j: delta_j == median} ; g_disp = -delta^2 * g_delta   (backward)
//   torch.median(x) without a dim differentiates as evenly_distribute_backward: the gradient of the median VALUE is shared equally by all
//   elements that hold it (one element unless values tie exactly).  `elems` (n_elems > 0): the caller names the elements instead.
// ---------------------------------------------------------------------------------------------
// S = sum(g * delta): per-workgroup partials (fixed assignment of elements to workgroups)
__global__ __launch_bounds__(DT) void k_dot_partials(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                     float* __restrict__ partials) {
    __shared__ float red[DT / 64];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) acc = fmaf(a[i], b[i], acc);
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(DT) void k_scale_chain_bwd(const float* __restrict__ g, const float* __restrict__ delta,
                                                        const float* __restrict__ m_gt, const float* __restrict__ m_delta,
                                                        const MedState* __restrict__ st, const float* __restrict__ partials,
                                                        int nparts, float* __restrict__ g_disp, int64_t n, const int* __restrict__ elems, int n_elems) {
    __shared__ float sS;
    if (threadIdx.x == 0) {                 // every workgroup re-adds the (few hundred) partials in the same order
        double acc = 0.0;
        for (int i = 0; i < nparts; ++i) acc += (double)partials[i];
        sS = (float)acc;
    }
    __syncthreads();
    const float md = m_delta[0], rho = m_gt[0] / md;
    const float corr = -(rho / md) * sS;
    const float share = corr / (float)(n_elems > 0 ? (unsigned int)n_elems : st->count);
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float d = delta[i];
        float gd = rho * g[i];
        if (n_elems > 0) {
            for (int e = 0; e < n_elems; ++e) if ((int)i == elems[e]) gd += share;
        } else if (d == md) gd += share;
        g_disp[i] = -(d * d) * gd;
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone mean |a-b| / mean (a-b)^2 (depth_reguralizer) : forward partials + backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DT) void k_diff_partials(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int kind,
                                                      float* __restrict__ partials) {
    __shared__ float red[DT / 64];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float e = a[i] - b[i];
        acc += (kind == 2) ? e * e : fabsf(e);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void k_sum_partials(const float* __restrict__ partials, int nparts, double scale, float* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += (double)partials[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((((sh[0] + sh[1]) + sh[2]) + sh[3]) * scale);
}

// d/db of mean-diff(a, b) * g   (a = initial depth: no gradient)
__global__ __launch_bounds__(DT) void k_diff_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                                 int64_t n, int kind, float* __restrict__ gb) {
    const float s = g[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float e = a[i] - b[i];
        gb[i] = (kind == 2) ? s * (-2.f * e) : s * ((e > 0.f) ? -1.f : ((e < 0.f) ? 1.f : 0.f));
    }
}

// ---------------------------------------------------------------------------------------------
// depth metrics: abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 over the kept pixels (TUM: gt != 0)
// ---------------------------------------------------------------------------------------------
#define NMET 8
__global__ __launch_bounds__(DT) void k_metrics_partials(const float* __restrict__ gt, const float* __restrict__ pred, int64_t n,
                                                         int mask_zero_gt, float* __restrict__ partials) {
    __shared__ float red[DT / 64];
    float acc[NMET] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float g = gt[i], p = pred[i];
        if (mask_zero_gt && g == 0.f) continue;
        const float th = fmaxf(g / p, p / g);
        const float e = g - p, le = logf(g) - logf(p);
        acc[0] += fabsf(e) / g;
        acc[1] += (e * e) / g;
        acc[2] += e * e;
        acc[3] += le * le;
        acc[4] += (th < 1.25f) ? 1.f : 0.f;
        acc[5] += (th < 1.5625f) ? 1.f : 0.f;
        acc[6] += (th < 1.953125f) ? 1.f : 0.f;
        acc[7] += 1.f;
    }
    for (int m = 0; m < NMET; ++m) {
        const float s = block_sum(acc[m], red);
        if (threadIdx.x == 0) partials[m * gridDim.x + blockIdx.x] = s;
        __syncthreads();
    }
}

__global__ void k_metrics_final(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
    __shared__ double tot[NMET];
    if (threadIdx.x < NMET) {
        double acc = 0.0;
        for (int i = 0; i < nparts; ++i) acc += (double)partials[threadIdx.x * nparts + i];
        tot[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double v = tot[threadIdx.x] / tot[7];
        if (threadIdx.x == 2 || threadIdx.x == 3) v = sqrt(v);
        out[threadIdx.x] = (float)v;
    }
}

// ---------------------------------------------------------------------------------------------
// fused Adam on a flat fp32 buffer (torch.optim.Adam, amsgrad off, weight_decay 0):
//   m = m + (1-b1)(g - m) ; v = b2 v + (1-b2) g g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------
// `participants` (device scalar or NULL): the gradient bucket holds the SUM over the data-parallel ranks and this is how
// many of them contributed (it travels as an extra element of the same all-reduce); g / max(participants, 1) is the mean
// -- the division rides in this kernel instead of a separate pass over the 57 MB bucket.  NULL or 1: g unchanged.
// `schedule` (device, or NULL): the per-step scalars {lr / (1 - beta1^t), sqrt(1 - beta2^t)} for t = 1, 2, ... as the HOST
// computes them (python doubles rounded to fp32, like torch.optim.Adam), indexed by the device-resident step counter: a
// launch that is replayed from a captured hipGraph still advances through the bias corrections (k_adam_advance bumps the
// counter behind it).
__global__ __launch_bounds__(DT) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                             int64_t n, float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                             float bc2_sqrt, float eps, const float* __restrict__ participants,
                                             const float2* __restrict__ schedule, const int* __restrict__ step_counter, int schedule_len) {
    const int64_t n4 = n >> 2;
    const float cnt = participants ? fmaxf(participants[0], 1.f) : 1.f;
    if (schedule) {
        int t = step_counter[0];
        t = t < 1 ? 1 : (t > schedule_len ? schedule_len : t);
        const float2 sc = schedule[t - 1];
        step_size = sc.x;
        bc2_sqrt = sc.y;
    }
    float4* p4 = (float4*)p; const float4* g4 = (const float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * DT) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ga[k] = ga[k] / cnt;
            ma[k] = ma[k] + one_minus_b1 * (ga[k] - ma[k]);
            va[k] = va[k] * b2 + one_minus_b2 * (ga[k] * ga[k]);
            const float den = sqrtf(va[k]) / bc2_sqrt + eps;
            pa[k] = pa[k] - step_size * (ma[k] / den);
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float gi = g[i] / cnt;
        const float mi = m[i] + one_minus_b1 * (gi - m[i]);
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

__global__ void k_adam_advance(int* __restrict__ step_counter) { step_counter[0] += 1; }

// ---------------------------------------------------------------------------------------------
static inline int sgrid(int64_t n, int cap = 1024) {
    int64_t g = (n + DT - 1) / DT;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
#define RED_PARTS 512

extern "C" {

int64_t e2e_median_workspace_bytes(void) { return (3 * MED_BINS + 8) * 4; }

int e2e_median_lower(const float* x, int64_t n, float* value_out, void* workspace, void* stream) {
    E2E_REQUIRE(n > 0 && n < 0xFFFFFFFFll && x && value_out && workspace, E2E_ERR_ARG, "e2e_median_lower: bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned int* hist = (unsigned int*)workspace;
    MedState* ms = (MedState*)(hist + 3 * MED_BINS);
    const unsigned int rank = (unsigned int)((n - 1) / 2);        // torch.median: the LOWER median
    // few workgroups: the histograms of depth data are concentrated in a few dozen bins, and every workgroup's flush
    // is a same-address device atomic (~0.18 us each, serialised) -- 512 workgroups made these passes 78-111 us
    const int g = sgrid(n, 96);
    hipLaunchKernelGGL(k_median_init, dim3(1), dim3(256), 0, st, hist, ms);
    hipLaunchKernelGGL(k_median_hist<0>, dim3(g), dim3(DT), 0, st, x, n, rank, hist, ms, (float*)nullptr);
    hipLaunchKernelGGL(k_median_hist<1>, dim3(g), dim3(DT), 0, st, x, n, rank, hist, ms, (float*)nullptr);
    hipLaunchKernelGGL(k_median_hist<2>, dim3(g), dim3(DT), 0, st, x, n, rank, hist, ms, (float*)nullptr);
    hipLaunchKernelGGL(k_median_final, dim3(g), dim3(DT), 0, st, x, n, hist, ms, value_out);
    E2E_LAUNCH_CHECK("e2e_median_lower");
    return E2E_OK;
}

/* workspace of the scale chain = median workspace + RED_PARTS floats */
int64_t e2e_depth_scale_workspace_bytes(void) { return e2e_median_workspace_bytes() + RED_PARTS * 4 + 64; }

int e2e_depth_scale_fwd(const float* disp, const float* median_gt, float* delta, float* depth, float* median_delta,
                        float* ratio_out, void* workspace, int64_t n, void* stream) {
    E2E_REQUIRE(n > 0 && disp && median_gt && delta && depth && median_delta && workspace, E2E_ERR_ARG, "e2e_depth_scale_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    E2E_REQUIRE(n < 0xFFFFFFFFll, E2E_ERR_ARG, "e2e_depth_scale_fwd: too many elements");
    // five launches (round 3: seven): the reciprocal rides in the first histogram pass, the median's last digit and the elements that hold
    // it in the scaling pass -- e2e_median_lower's kernels otherwise, and its results bit for bit
    unsigned int* hist = (unsigned int*)workspace;
    MedState* ms = (MedState*)(hist + 3 * MED_BINS);
    const unsigned int rank = (unsigned int)((n - 1) / 2);
    const int g = sgrid(n, 96);
    hipLaunchKernelGGL(k_median_init, dim3(1), dim3(256), 0, st, hist, ms);
    hipLaunchKernelGGL((k_median_hist<0, true>), dim3(g), dim3(DT), 0, st, disp, n, rank, hist, ms, delta);
    hipLaunchKernelGGL(k_median_hist<1>, dim3(g), dim3(DT), 0, st, (const float*)delta, n, rank, hist, ms, (float*)nullptr);
    hipLaunchKernelGGL(k_median_hist<2>, dim3(g), dim3(DT), 0, st, (const float*)delta, n, rank, hist, ms, (float*)nullptr);
    hipLaunchKernelGGL(k_median_final_scale, dim3(sgrid(n)), dim3(DT), 0, st, (const float*)delta, n, (const unsigned int*)hist, ms, median_gt, median_delta, depth, ratio_out);
    E2E_LAUNCH_CHECK("e2e_depth_scale_fwd");
    return E2E_OK;
}

// fixed scale (train_depth.py:343-345: depth = 1 / disp, then `*= ABLATION.scaling_depth`): delta = 1 / disp, depth = delta * scale, in the
// reference's operation order; backward g_disp = -(g_depth * scale) * delta * delta
__global__ __launch_bounds__(DT) void k_fixed_scale_fwd(const float* __restrict__ disp, float scale, float* __restrict__ delta, float* __restrict__ depth,
                                                        int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float d = 1.f / disp[i];
        if (delta) delta[i] = d;
        depth[i] = d * scale;
    }
}
__global__ __launch_bounds__(DT) void k_fixed_scale_bwd(const float* __restrict__ g_depth, const float* __restrict__ disp, float scale,
                                                        float* __restrict__ g_disp, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT) {
        const float d = 1.f / disp[i];
        g_disp[i] = -(g_depth[i] * scale) * d * d;
    }
}

int e2e_depth_fixed_scale_fwd(const float* disp, float scale, float* delta, float* depth, int64_t n, void* stream) {
    E2E_REQUIRE(n > 0 && disp && depth, E2E_ERR_ARG, "e2e_depth_fixed_scale_fwd: bad argument");
    hipLaunchKernelGGL(k_fixed_scale_fwd, dim3(sgrid(n)), dim3(DT), 0, (hipStream_t)stream, disp, scale, delta, depth, n);
    E2E_LAUNCH_CHECK("e2e_depth_fixed_scale_fwd");
    return E2E_OK;
}

int e2e_depth_fixed_scale_bwd(const float* g_depth, const float* disp, float scale, float* g_disp, int64_t n, void* stream) {
    E2E_REQUIRE(n > 0 && g_depth && disp && g_disp, E2E_ERR_ARG, "e2e_depth_fixed_scale_bwd: bad argument");
    hipLaunchKernelGGL(k_fixed_scale_bwd, dim3(sgrid(n)), dim3(DT), 0, (hipStream_t)stream, g_depth, disp, scale, g_disp, n);
    E2E_LAUNCH_CHECK("e2e_depth_fixed_scale_bwd");
    return E2E_OK;
}

int e2e_depth_scale_bwd(const float* g_depth, const float* delta, const float* median_gt, const float* median_delta,
                        float* g_disp, void* workspace, int64_t n, void* stream) {
    return e2e_depth_scale_bwd_at(g_depth, delta, median_gt, median_delta, nullptr, 0, g_disp, workspace, n, stream);
}

int e2e_depth_scale_bwd_at(const float* g_depth, const float* delta, const float* median_gt, const float* median_delta, const int* elements,
                           int n_elements, float* g_disp, void* workspace, int64_t n, void* stream) {
    E2E_REQUIRE(n > 0 && g_depth && delta && median_gt && median_delta && g_disp && workspace, E2E_ERR_ARG, "e2e_depth_scale_bwd: bad argument");
    E2E_REQUIRE(n_elements >= 0 && n_elements <= 64 && (n_elements == 0 || elements), E2E_ERR_ARG, "e2e_depth_scale_bwd_at: 0..64 named elements");
    hipStream_t st = (hipStream_t)stream;
    const MedState* ms = (const MedState*)((unsigned int*)workspace + 3 * MED_BINS);
    float* parts = (float*)((char*)workspace + e2e_median_workspace_bytes() + 32);
    const int g = sgrid(n, RED_PARTS);
    hipLaunchKernelGGL(k_dot_partials, dim3(g), dim3(DT), 0, st, g_depth, delta, n, parts);
    hipLaunchKernelGGL(k_scale_chain_bwd, dim3(sgrid(n)), dim3(DT), 0, st, g_depth, delta, median_gt, median_delta, ms, parts, g, g_disp, n, elements, n_elements);
    E2E_LAUNCH_CHECK("e2e_depth_scale_bwd");
    return E2E_OK;
}

int e2e_mean_diff_fwd(const float* a, const float* b, int64_t n, int kind, float* out, float* workspace, void* stream) {
    E2E_REQUIRE(n > 0 && a && b && out && workspace && (kind == 1 || kind == 2), E2E_ERR_ARG, "e2e_mean_diff_fwd: bad argument (kind 1 = l1, 2 = l2)");
    hipStream_t st = (hipStream_t)stream;
    const int g = sgrid(n, RED_PARTS);
    hipLaunchKernelGGL(k_diff_partials, dim3(g), dim3(DT), 0, st, a, b, n, kind, workspace);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, workspace, g, 1.0 / (double)n, out);
    E2E_LAUNCH_CHECK("e2e_mean_diff_fwd");
    return E2E_OK;
}

int e2e_mean_diff_bwd(const float* a, const float* b, const float* g_out, int64_t n, int kind, float* g_b, void* stream) {
    E2E_REQUIRE(n > 0 && a && b && g_out && g_b && (kind == 1 || kind == 2), E2E_ERR_ARG, "e2e_mean_diff_bwd: bad argument");
    hipLaunchKernelGGL(k_diff_bwd, dim3(sgrid(n)), dim3(DT), 0, (hipStream_t)stream, a, b, g_out, n, kind, g_b);
    E2E_LAUNCH_CHECK("e2e_mean_diff_bwd");
    return E2E_OK;
}

int e2e_depth_metrics(const float* gt, const float* pred, int64_t n, int mask_zero_gt, float* out7, float* workspace, void* stream) {
    E2E_REQUIRE(n > 0 && gt && pred && out7 && workspace, E2E_ERR_ARG, "e2e_depth_metrics: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = sgrid(n, 256);
    hipLaunchKernelGGL(k_metrics_partials, dim3(g), dim3(DT), 0, st, gt, pred, n, mask_zero_gt, workspace);
    hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(64), 0, st, workspace, g, out7);
    E2E_LAUNCH_CHECK("e2e_depth_metrics");
    return E2E_OK;
}
int64_t e2e_reduce_workspace_floats(void) { return NMET * 256 > RED_PARTS ? NMET * 256 : RED_PARTS; }

int e2e_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream) {
    E2E_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && step >= 1, E2E_ERR_ARG, "e2e_adam_step: bad argument");
    E2E_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, E2E_ERR_ARG,
                "e2e_adam_step: buffers must be 16-byte aligned");
    // bias corrections as torch computes them (python doubles, then fp32 scalars)
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(k_adam, dim3(sgrid(n >> 2 > 0 ? n >> 2 : 1, 2048)), dim3(DT), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, 1.0f - beta1, beta2, 1.0f - beta2, step_size, bc2_sqrt, eps, (const float*)nullptr,
                       (const float2*)nullptr, (const int*)nullptr, 0);
    E2E_LAUNCH_CHECK("e2e_adam_step");
    return E2E_OK;
}

int e2e_adam_step_mean(float* params, const float* grad_sums, const float* participants, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, int step, void* stream) {
    E2E_REQUIRE(n > 0 && params && grad_sums && participants && exp_avg && exp_avg_sq && step >= 1, E2E_ERR_ARG, "e2e_adam_step_mean: bad argument");
    E2E_REQUIRE(((uintptr_t)params | (uintptr_t)grad_sums | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, E2E_ERR_ARG,
                "e2e_adam_step_mean: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(k_adam, dim3(sgrid(n >> 2 > 0 ? n >> 2 : 1, 2048)), dim3(DT), 0, (hipStream_t)stream, params, grad_sums, exp_avg,
                       exp_avg_sq, n, 1.0f - beta1, beta2, 1.0f - beta2, step_size, bc2_sqrt, eps, participants,
                       (const float2*)nullptr, (const int*)nullptr, 0);
    E2E_LAUNCH_CHECK("e2e_adam_step_mean");
    return E2E_OK;
}

int e2e_adam_step_resident(float* params, const float* grad_sums, const float* participants, float* exp_avg, float* exp_avg_sq, int64_t n,
                           float beta1, float beta2, float eps, const float* schedule, int schedule_len, int* step_counter, void* stream) {
    E2E_REQUIRE(n > 0 && params && grad_sums && exp_avg && exp_avg_sq && schedule && schedule_len > 0 && step_counter, E2E_ERR_ARG,
                "e2e_adam_step_resident: bad argument");
    E2E_REQUIRE(((uintptr_t)params | (uintptr_t)grad_sums | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (uintptr_t)schedule % 8 == 0,
                E2E_ERR_ARG, "e2e_adam_step_resident: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adam, dim3(sgrid(n >> 2 > 0 ? n >> 2 : 1, 2048)), dim3(DT), 0, st, params, grad_sums, exp_avg, exp_avg_sq, n,
                       1.0f - beta1, beta2, 1.0f - beta2, 0.f, 1.f, eps, participants, (const float2*)schedule, (const int*)step_counter, schedule_len);
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, st, step_counter);
    E2E_LAUNCH_CHECK("e2e_adam_step_resident");
    return E2E_OK;
}

}  // extern "C"
