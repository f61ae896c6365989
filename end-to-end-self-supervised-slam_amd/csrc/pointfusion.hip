// pointfusion.hip -- RGB-D unprojection (vertex / normal maps) and the PointFusion map step for
// gfx950.  Semantics: gradslam RGBDImages / fusionutils as restated in oracle/pointfusion.py (SURVEY.md
// Appendix A; reference call sites online_adaption.py:347-363, :461-469).
//
// BIT-EXACTNESS: the index tables / masks must equal the CPU oracle's.  Every fp32 expression that
// feeds a comparison is written in the oracle's operation order and this file is compiled with
// -ffp-contract=off (no implicit FMA); divisions and sqrt are IEEE (hipcc default).  Ordering comes from
// prefix sums and integer atomicMin only -- never from "first atomic wins".
//
// Data layout: a frame's maps are (H,W,3) channels-last fp32 (as gradslam); the global map is four
// capacity-sized arrays points/normals/colors (cap,3) + ccounts (cap) that stay resident in HBM
// (60 frames x 307 200 px x 40 B = 737 MB << 288 GB), appended in place.
#include "e2e_common.h"

#define PF_T 256
#define PF_NONE 0xFFFFFFFFu
#define PF_KEY_NONE 0xFFFFFFFFFFFFFFFFull

struct Pose {          // row-major 4x4 pieces
    float R[9], t[3];
};
__device__ __forceinline__ Pose load_pose(const float* __restrict__ T) {
    Pose p;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) p.R[i * 3 + j] = T[i * 4 + j];
        p.t[i] = T[i * 4 + 3];
    }
    return p;
}
// rigid inverse [R^T | -R^T t], oracle order: -((Rt_i0*t0 + Rt_i1*t1) + Rt_i2*t2)
__device__ __forceinline__ Pose inverse_pose(const Pose& p) {
    Pose q;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) q.R[i * 3 + j] = p.R[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) q.t[i] = -((q.R[i * 3 + 0] * p.t[0] + q.R[i * 3 + 1] * p.t[1]) + q.R[i * 3 + 2] * p.t[2]);
    return q;
}
// Correctly rounded fp32 square root.  hipcc lowers sqrtf() -- and (float)sqrt((double)x), which LLVM folds
// back to sqrtf -- to a bare v_sqrt_f32 (1 ulp) on gfx950, while the oracle's torch.sqrt is IEEE.  Fix-up of
// the 1-ulp estimate with exact FMA residuals against its two neighbours (the sequence LLVM itself uses
// for an IEEE f32 sqrt); inputs here are normal, non-negative numbers or zero.
__device__ __forceinline__ float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    if (!(x > 0.f) || !(x < 3.0e38f)) return s;           // 0, NaN, inf: the instruction's answer stands
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = fmaf(-sd, s, x);
    const float vs = fmaf(-su, s, x);
    if (vp <= 0.f) s = sd;
    if (vs > 0.f) s = su;
    return s;
}

__device__ __forceinline__ void rot(const float* R, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (R[i * 3 + 0] * v[0] + R[i * 3 + 1] * v[1]) + R[i * 3 + 2] * v[2];
}

// ---------------------------------------------------------------------------------------------
// K9: vertex / normal maps.   V = (Ki00*w + Ki02, Ki11*h + Ki12, 1) * d * valid ;
// n = normalize(cross(V[h,w+1]-V[h,w], V[h+1,w]-V[h,w])) * valid  (last column / row difference = 0);
// Vg = (R V + t) * valid ; ng = R n ; alpha = exp(-|V|^2 / den)           (gradslam get_alpha)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void local_vertex(float d, float kx, float kxc, float ky, float kyc, int w, int h, float* V) {
    const float vf = (d != 0.f) ? 1.f : 0.f;
    const float rx = kx * (float)w + kxc, ry = ky * (float)h + kyc;
    V[0] = (rx * d) * vf;
    V[1] = (ry * d) * vf;
    V[2] = d * vf;
}

__global__ __launch_bounds__(PF_T) void k_vertex_normal_maps(const float* __restrict__ depth, const float* __restrict__ K,
                                                             const float* __restrict__ pose, float alpha_den,
                                                             float* __restrict__ V, float* __restrict__ Nm,
                                                             float* __restrict__ Vg, float* __restrict__ Ng,
                                                             float* __restrict__ alpha, int H, int W) {
    const int b = blockIdx.y;
    const int64_t N = (int64_t)H * W;
    const float* Kb = K + b * 16;
    const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
    const float kx = 1.0f / fx, ky = 1.0f / fy, kxc = -cx / fx, kyc = -cy / fy;     // oracle intrinsics_inverse
    const Pose P = load_pose(pose + b * 16);
    const float* dep = depth + b * N;
    for (int i = blockIdx.x * PF_T + threadIdx.x; i < N; i += gridDim.x * PF_T) {
        const int h = i / W, w = i - h * W;
        const float d = dep[i];
        const float vf = (d != 0.f) ? 1.f : 0.f;
        float v0[3], vr[3], vd[3];
        local_vertex(d, kx, kxc, ky, kyc, w, h, v0);
        float dh[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f};
        if (w + 1 < W) {
            local_vertex(dep[i + 1], kx, kxc, ky, kyc, w + 1, h, vr);
#pragma unroll
            for (int c = 0; c < 3; ++c) dh[c] = vr[c] - v0[c];
        }
        if (h + 1 < H) {
            local_vertex(dep[i + W], kx, kxc, ky, kyc, w, h + 1, vd);
#pragma unroll
            for (int c = 0; c < 3; ++c) dv[c] = vd[c] - v0[c];
        }
        float n[3];
        n[0] = dh[1] * dv[2] - dh[2] * dv[1];
        n[1] = dh[2] * dv[0] - dh[0] * dv[2];
        n[2] = dh[0] * dv[1] - dh[1] * dv[0];
        const float nrm = sqrt_rn((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
        const float den = (nrm == 0.f) ? 1.f : nrm;
#pragma unroll
        for (int c = 0; c < 3; ++c) n[c] = (n[c] / den) * vf;
        float vg[3], ng[3];
        rot(P.R, v0, vg);
        rot(P.R, n, ng);
#pragma unroll
        for (int c = 0; c < 3; ++c) vg[c] = (vg[c] + P.t[c]) * vf;
        const int64_t o = (b * N + i) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (V) V[o + c] = v0[c];
            if (Nm) Nm[o + c] = n[c];
            Vg[o + c] = vg[c];
            if (Ng) Ng[o + c] = ng[c];
        }
        if (alpha) {
            const float s = (v0[0] * v0[0] + v0[1] * v0[1]) + v0[2] * v0[2];
            alpha[b * N + i] = expf(-s / alpha_den);
        }
    }
}

// gradient of the vertex maps wrt depth (the normal map is treated as non-differentiable):
// dV/dd = (rx, ry, 1) ; dVg/dd = R (rx, ry, 1)   for valid pixels
__global__ __launch_bounds__(PF_T) void k_vertex_maps_bwd(const float* __restrict__ depth, const float* __restrict__ K,
                                                          const float* __restrict__ pose, const float* __restrict__ gV,
                                                          const float* __restrict__ gVg, float* __restrict__ gdepth, int H, int W) {
    const int b = blockIdx.y;
    const int64_t N = (int64_t)H * W;
    const float* Kb = K + b * 16;
    const float fx = Kb[0], fy = Kb[5], cx = Kb[2], cy = Kb[6];
    const float kx = 1.0f / fx, ky = 1.0f / fy, kxc = -cx / fx, kyc = -cy / fy;
    const Pose P = load_pose(pose + b * 16);
    for (int i = blockIdx.x * PF_T + threadIdx.x; i < N; i += gridDim.x * PF_T) {
        const int h = i / W, w = i - h * W;
        float g = 0.f;
        if (depth[b * N + i] != 0.f) {
            const float ray[3] = {kx * (float)w + kxc, ky * (float)h + kyc, 1.f};
            const int64_t o = (b * N + i) * 3;
            if (gV) g += (gV[o] * ray[0] + gV[o + 1] * ray[1]) + gV[o + 2] * ray[2];
            if (gVg) {
                float rr[3];
                rot(P.R, ray, rr);
                g += (gVg[o] * rr[0] + gVg[o + 1] * rr[1]) + gVg[o + 2] * rr[2];
            }
        }
        gdepth[b * N + i] = g;
    }
}

// K15: transform_pointcloud (gradslam geometryutils): out = R p + t ; bwd: g_p = R^T g
__global__ __launch_bounds__(PF_T) void k_transform_points(const float* __restrict__ p, const float* __restrict__ T,
                                                           float* __restrict__ out, int64_t n, int transpose_only) {
    const Pose P = load_pose(T);
    for (int64_t i = (int64_t)blockIdx.x * PF_T + threadIdx.x; i < n; i += (int64_t)gridDim.x * PF_T) {
        const float v[3] = {p[i * 3], p[i * 3 + 1], p[i * 3 + 2]};
        float o[3];
        if (!transpose_only) {
            rot(P.R, v, o);
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] += P.t[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (P.R[0 * 3 + c] * v[0] + P.R[1 * 3 + c] * v[1]) + P.R[2 * 3 + c] * v[2];
        }
        out[i * 3] = o[0];
        out[i * 3 + 1] = o[1];
        out[i * 3 + 2] = o[2];
    }
}

// ---------------------------------------------------------------------------------------------
// association.  One thread per map point.
//   pass 1 (K10+K11): project into the live camera, in-frame test, round-half-even to (h,w), compare with the
//           frame's global vertex / normal -> per-point pix (or NONE), flags, and per-pixel 64-bit
//           atomicMin of key = (bits(1/(c+1e-20)) << 32) | bits(dist^2)     [both floats >= 0]
//   pass 2 (K12): points whose key equals their pixel's minimum race with atomicMin on the point index
//           => per pixel: max confidence, then min distance, then min index -- order independent.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PF_T) void k_pf_assoc1(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                    const float* __restrict__ cc, int64_t M, const float* __restrict__ K,
                                                    const float* __restrict__ pose, const float* __restrict__ Vg,
                                                    const float* __restrict__ Ng, float dist_th, float dot_th,
                                                    float u_hi, float v_hi, unsigned int* __restrict__ pix_out,
                                                    unsigned char* __restrict__ flags,
                                                    unsigned long long* __restrict__ pix_key, int H, int W, const long long* __restrict__ Mp) {
    if (Mp) M = (int64_t)*Mp;                         // resident map: the live size is device data (e2e_pf_associate_dev)
    const float fx = K[0], fy = K[5], cx = K[2], cy = K[6];
    const Pose Ti = inverse_pose(load_pose(pose));
    for (int64_t n = (int64_t)blockIdx.x * PF_T + threadIdx.x; n < M; n += (int64_t)gridDim.x * PF_T) {
        const float p[3] = {pts[n * 3], pts[n * 3 + 1], pts[n * 3 + 2]};
        float c[3];
        rot(Ti.R, p, c);
#pragma unroll
        for (int i = 0; i < 3; ++i) c[i] += Ti.t[i];
        const float u = (fx * c[0] + cx * c[2]) / c[2];
        const float v = (fy * c[1] + cy * c[2]) / c[2];
        const bool active = (u > -1e-3f) && (u < u_hi) && (v > -1e-3f) && (v < v_hi) && (c[2] > 0.f);
        unsigned int pix = PF_NONE;
        unsigned char fl = 0;
        if (active) {
            fl = 1;
            int w = (int)rintf(u), h = (int)rintf(v);          // torch.round = round-half-even
            w = min(max(w, 0), W - 1);
            h = min(max(h, 0), H - 1);
            const int q = h * W + w;
            const float* fv = Vg + (int64_t)q * 3;
            const float* fn = Ng + (int64_t)q * 3;
            const float d0 = fv[0] - p[0], d1 = fv[1] - p[1], d2 = fv[2] - p[2];
            const float dsq = (d0 * d0 + d1 * d1) + d2 * d2;
            const float dot = (fn[0] * nrm[n * 3] + fn[1] * nrm[n * 3 + 1]) + fn[2] * nrm[n * 3 + 2];
            pix = (unsigned int)q;
            if (sqrt_rn(dsq) < dist_th && dot > dot_th) {
                fl = 3;
                // oracle find_best_unique: d2 = |p - Vg|^2 (same value: squares of negated differences)
                const float inv_c = 1.0f / (cc[n] + 1e-20f);
                const unsigned long long key = ((unsigned long long)__float_as_uint(inv_c) << 32) | (unsigned long long)__float_as_uint(dsq);
                atomicMin(pix_key + q, key);
            }
        }
        pix_out[n] = pix;
        flags[n] = fl;
    }
}

__global__ __launch_bounds__(PF_T) void k_pf_assoc2(const float* __restrict__ pts, const float* __restrict__ cc, int64_t M,
                                                    const float* __restrict__ Vg, const unsigned int* __restrict__ pix_in,
                                                    const unsigned char* __restrict__ flags,
                                                    const unsigned long long* __restrict__ pix_key,
                                                    unsigned int* __restrict__ pix_best, unsigned int* __restrict__ any_match,
                                                    const long long* __restrict__ Mp) {
    if (Mp) M = (int64_t)*Mp;
    for (int64_t n = (int64_t)blockIdx.x * PF_T + threadIdx.x; n < M; n += (int64_t)gridDim.x * PF_T) {
        if (flags[n] != 3) continue;
        const unsigned int q = pix_in[n];
        const float* fv = Vg + (int64_t)q * 3;
        const float d0 = fv[0] - pts[n * 3], d1 = fv[1] - pts[n * 3 + 1], d2 = fv[2] - pts[n * 3 + 2];
        const float dsq = (d0 * d0 + d1 * d1) + d2 * d2;
        const float inv_c = 1.0f / (cc[n] + 1e-20f);
        const unsigned long long key = ((unsigned long long)__float_as_uint(inv_c) << 32) | (unsigned long long)__float_as_uint(dsq);
        if (key == pix_key[q]) {
            atomicMin(pix_best + q, (unsigned int)n);
            *any_match = 1u;            // every writer stores the same value
        }
    }
}

// K13: every map point goes through X' = (c X + a X_f) / where(c+a == 0, 1, c+a) (a = 0, X_f = 0 when
// it is not its pixel's winner), exactly as the padded-tensor formulation does.
__global__ __launch_bounds__(PF_T) void k_pf_fuse(float* __restrict__ pts, float* __restrict__ nrm, float* __restrict__ col,
                                                  float* __restrict__ cc, int64_t M, const unsigned int* __restrict__ pix_in,
                                                  const unsigned char* __restrict__ flags, const unsigned int* __restrict__ pix_best,
                                                  const float* __restrict__ Vg, const float* __restrict__ Ng,
                                                  const float* __restrict__ rgb, const float* __restrict__ alpha,
                                                  const unsigned int* __restrict__ any_match, const long long* __restrict__ Mp) {
    // gradslam fuses only `if has_points and correspondences exist`; otherwise the map is left untouched
    if (*any_match == 0u) return;
    if (Mp) M = (int64_t)*Mp;
    for (int64_t n = (int64_t)blockIdx.x * PF_T + threadIdx.x; n < M; n += (int64_t)gridDim.x * PF_T) {
        float a = 0.f, fp[3] = {0.f, 0.f, 0.f}, fn[3] = {0.f, 0.f, 0.f}, fc[3] = {0.f, 0.f, 0.f};
        if (flags[n] == 3) {
            const unsigned int q = pix_in[n];
            if (pix_best[q] == (unsigned int)n) {
                a = alpha[q];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    fp[c] = Vg[(int64_t)q * 3 + c];
                    fn[c] = Ng[(int64_t)q * 3 + c];
                    fc[c] = rgb[(int64_t)q * 3 + c];
                }
            }
        }
        const float c0 = cc[n];
        const float cn = c0 + a;
        const float den = (cn == 0.f) ? 1.f : cn;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[n * 3 + c] = (c0 * pts[n * 3 + c] + a * fp[c]) / den;
            nrm[n * 3 + c] = (c0 * nrm[n * 3 + c] + a * fn[c]) / den;
            col[n * 3 + c] = (c0 * col[n * 3 + c] + a * fc[c]) / den;
        }
        cc[n] = cn;
    }
}

// ---------------------------------------------------------------------------------------------
// ordered stream compaction over `total` items, 1024 items per workgroup:
//   count pass -> k_scan_counts (1 workgroup, exclusive scan + total) -> scatter pass.
// `pred(i)` is recomputed in both passes from the per-item arrays (cheap), so no flag array is stored.
// ---------------------------------------------------------------------------------------------
#define CP_ITEMS 4
#define CP_BLOCK (PF_T * CP_ITEMS)

enum { PRED_ACTIVE = 0, PRED_SIMILAR = 1, PRED_PIX_MATCHED = 2, PRED_PIX_NEW = 3 };

template <int PRED>
__device__ __forceinline__ bool cp_pred(int64_t i, const unsigned char* flags, const unsigned int* pix_best, const float* depth) {
    if (PRED == PRED_ACTIVE) return flags[i] != 0;
    if (PRED == PRED_SIMILAR) return flags[i] == 3;
    if (PRED == PRED_PIX_MATCHED) return pix_best[i] != PF_NONE;
    return pix_best[i] == PF_NONE && depth[i] != 0.f;     // new point: valid depth, no correspondence
}

template <int PRED>
__global__ __launch_bounds__(PF_T) void k_cp_count(int64_t total, const unsigned char* __restrict__ flags,
                                                   const unsigned int* __restrict__ pix_best, const float* __restrict__ depth,
                                                   unsigned int* __restrict__ counts) {
    __shared__ unsigned int sh[PF_T / 64];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (base + j < total && cp_pred<PRED>(base + j, flags, pix_best, depth)) ++c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive scan of nb block counts in place; total -> *total_out (int64).  One workgroup.
__global__ __launch_bounds__(1024) void k_scan_counts(unsigned int* __restrict__ counts, int nb, long long* __restrict__ total_out,
                                                      long long add_to_total, const long long* __restrict__ add_dev) {
    if (add_dev) add_to_total = *add_dev;             // (every thread reads it before thread 0 may overwrite *total_out == *add_dev: barriers below)
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned int v = (i < nb) ? counts[i] : 0u;
        unsigned int x = v;                       // inclusive scan within the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int y = __shfl_up(x, o, 64);
            if ((threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        unsigned int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const unsigned int carry = carry_s;
        if (i < nb) counts[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (long long)carry_s + add_to_total;
}

// intra-workgroup exclusive offset of this thread's CP_ITEMS items (item order preserved)
__device__ __forceinline__ unsigned int cp_block_offset(unsigned int mine, unsigned int* sh) {
    unsigned int x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int y = __shfl_up(x, o, 64);
        if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += sh[w];
    return woff + x - mine;
}

// rows [n, h, w] (int64) of the map points selected by PRED (ascending n)
template <int PRED>
__global__ __launch_bounds__(PF_T) void k_cp_point_rows(int64_t M, const unsigned char* __restrict__ flags,
                                                        const unsigned int* __restrict__ pix_in,
                                                        const unsigned int* __restrict__ offsets, int W,
                                                        long long* __restrict__ rows) {
    __shared__ unsigned int sh[PF_T / 64];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    bool pr[CP_ITEMS];
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j) {
        pr[j] = base + j < M && cp_pred<PRED>(base + j, flags, nullptr, nullptr);
        c += pr[j];
    }
    unsigned int o = offsets[blockIdx.x] + cp_block_offset(c, sh);
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (pr[j]) {
            const unsigned int q = pix_in[base + j];
            rows[(int64_t)o * 3 + 0] = base + j;
            rows[(int64_t)o * 3 + 1] = q / W;
            rows[(int64_t)o * 3 + 2] = q % W;
            ++o;
        }
}

// rows [n, h, w] of the per-pixel winners, ordered by pixel (h, w)
__global__ __launch_bounds__(PF_T) void k_cp_pixel_rows(int64_t N, const unsigned int* __restrict__ pix_best,
                                                        const unsigned int* __restrict__ offsets, int W,
                                                        long long* __restrict__ rows) {
    __shared__ unsigned int sh[PF_T / 64];
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    bool pr[CP_ITEMS];
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j) {
        pr[j] = base + j < N && pix_best[base + j] != PF_NONE;
        c += pr[j];
    }
    unsigned int o = offsets[blockIdx.x] + cp_block_offset(c, sh);
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (pr[j]) {
            rows[(int64_t)o * 3 + 0] = pix_best[base + j];
            rows[(int64_t)o * 3 + 1] = (base + j) / W;
            rows[(int64_t)o * 3 + 2] = (base + j) % W;
            ++o;
        }
}

// K14: append the frame's unmatched valid pixels (row-major order) behind the M existing map points
__global__ __launch_bounds__(PF_T) void k_pf_append(int64_t N, int64_t M, const unsigned int* __restrict__ pix_best,
                                                    const float* __restrict__ depth, const unsigned int* __restrict__ offsets,
                                                    const float* __restrict__ Vg, const float* __restrict__ Ng,
                                                    const float* __restrict__ rgb, const float* __restrict__ alpha,
                                                    float* __restrict__ pts, float* __restrict__ nrm, float* __restrict__ col,
                                                    float* __restrict__ cc, int64_t cap, const long long* __restrict__ Mp) {
    __shared__ unsigned int sh[PF_T / 64];
    if (Mp) M = (int64_t)*Mp;
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    bool pr[CP_ITEMS];
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j) {
        pr[j] = base + j < N && cp_pred<PRED_PIX_NEW>(base + j, nullptr, pix_best, depth);
        c += pr[j];
    }
    unsigned int o = offsets[blockIdx.x] + cp_block_offset(c, sh);
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (pr[j]) {
            const int64_t dst = M + o, q = base + j;
            if (dst < cap) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pts[dst * 3 + k] = Vg[q * 3 + k];
                    nrm[dst * 3 + k] = Ng[q * 3 + k];
                    col[dst * 3 + k] = rgb[q * 3 + k];
                }
                cc[dst] = alpha[q];
            }
            ++o;
        }
}

__global__ void k_fill_u32(unsigned int* p, int64_t n, unsigned int v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill_u64(unsigned long long* p, int64_t n, unsigned long long v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline int grid_for(int64_t n, int per_block = PF_T, int cap = 8192) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" {

int e2e_vertex_normal_maps(const float* depth, const float* K, const float* pose, float alpha_den, float* V,
                           float* Nm, float* Vg, float* Ng, float* alpha, int B, int H, int W, void* stream) {
    E2E_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)B * H * W * 3 < (1ll << 31), E2E_ERR_ARG, "e2e_vertex_normal_maps: bad dims");
    E2E_REQUIRE(depth && K && pose && Vg, E2E_ERR_ARG, "e2e_vertex_normal_maps: null pointer");
    hipLaunchKernelGGL(k_vertex_normal_maps, dim3(grid_for((int64_t)H * W, PF_T, 2048), B), dim3(PF_T), 0, (hipStream_t)stream,
                       depth, K, pose, alpha_den, V, Nm, Vg, Ng, alpha, H, W);
    E2E_LAUNCH_CHECK("e2e_vertex_normal_maps");
    return E2E_OK;
}

int e2e_vertex_maps_bwd(const float* depth, const float* K, const float* pose, const float* g_V, const float* g_Vg,
                        float* g_depth, int B, int H, int W, void* stream) {
    E2E_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)B * H * W * 3 < (1ll << 31), E2E_ERR_ARG, "e2e_vertex_maps_bwd: bad dims");
    E2E_REQUIRE(depth && K && pose && g_depth && (g_V || g_Vg), E2E_ERR_ARG, "e2e_vertex_maps_bwd: null pointer");
    hipLaunchKernelGGL(k_vertex_maps_bwd, dim3(grid_for((int64_t)H * W, PF_T, 2048), B), dim3(PF_T), 0, (hipStream_t)stream,
                       depth, K, pose, g_V, g_Vg, g_depth, H, W);
    E2E_LAUNCH_CHECK("e2e_vertex_maps_bwd");
    return E2E_OK;
}

int e2e_transform_points(const float* points, const float* T, float* out, int64_t n, int transpose_rotation_only, void* stream) {
    E2E_REQUIRE(n >= 0 && points && T && out, E2E_ERR_ARG, "e2e_transform_points: bad argument");
    if (n == 0) return E2E_OK;
    hipLaunchKernelGGL(k_transform_points, dim3(grid_for(n)), dim3(PF_T), 0, (hipStream_t)stream, points, T, out, n, transpose_rotation_only);
    E2E_LAUNCH_CHECK("e2e_transform_points");
    return E2E_OK;
}

int64_t e2e_pf_workspace_bytes(int64_t map_capacity, int H, int W) {
    if (map_capacity < 0 || H <= 0 || W <= 0) return 0;
    const int64_t N = (int64_t)H * W;
    const int64_t nbm = (map_capacity + CP_BLOCK - 1) / CP_BLOCK + 1, nbp = (N + CP_BLOCK - 1) / CP_BLOCK + 1;
    // pix_key u64[N] | pix_best u32[N] | pix_of_point u32[cap] | counts u32[max(nbm,nbp)] | flags u8[cap]
    int64_t b = 8 * N + 4 * N + 4 * map_capacity + 4 * ((nbm > nbp ? nbm : nbp) + 4) + map_capacity;
    return (b + 255) & ~255ll;
}

struct PfWs {
    unsigned long long* pix_key;
    unsigned int* pix_best;
    unsigned int* pix_of_point;
    unsigned int* counts;
    unsigned int* any_match;
    unsigned char* flags;
};
static PfWs pf_ws(void* ws, int64_t cap, int H, int W) {
    const int64_t N = (int64_t)H * W;
    const int64_t nbm = (cap + CP_BLOCK - 1) / CP_BLOCK + 1, nbp = (N + CP_BLOCK - 1) / CP_BLOCK + 1;
    PfWs w;
    char* p = (char*)ws;
    w.pix_key = (unsigned long long*)p; p += 8 * N;
    w.pix_best = (unsigned int*)p; p += 4 * N;
    w.pix_of_point = (unsigned int*)p; p += 4 * cap;
    w.counts = (unsigned int*)p; p += 4 * (nbm > nbp ? nbm : nbp);
    w.any_match = (unsigned int*)p; p += 16;
    w.flags = (unsigned char*)p;
    return w;
}

int e2e_pf_associate(const float* map_points, const float* map_normals, const float* map_ccounts, int64_t M,
                     const float* K, const float* pose, const float* Vg, const float* Ng, float dist_th, float dot_th,
                     void* workspace, int64_t map_capacity, int H, int W, void* stream) {
    E2E_REQUIRE(M >= 0 && M <= map_capacity && M < (1ll << 32) - 1 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), E2E_ERR_ARG,
                "e2e_pf_associate: bad sizes M=%lld cap=%lld", (long long)M, (long long)map_capacity);
    E2E_REQUIRE(K && pose && Vg && Ng && workspace && (M == 0 || (map_points && map_normals && map_ccounts)), E2E_ERR_ARG,
                "e2e_pf_associate: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(N, 256, 1024)), dim3(256), 0, st, w.pix_key, N, PF_KEY_NONE);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(N, 256, 1024)), dim3(256), 0, st, w.pix_best, N, PF_NONE);
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(64), 0, st, w.any_match, 4, 0u);
    if (M > 0) {
        // thresholds evaluated in double then rounded to fp32, as the oracle's tensor-vs-python-float comparisons do
        const float u_hi = (float)((double)W - 0.999), v_hi = (float)((double)H - 0.999);
        hipLaunchKernelGGL(k_pf_assoc1, dim3(grid_for(M)), dim3(PF_T), 0, st, map_points, map_normals, map_ccounts, M, K, pose, Vg, Ng,
                           dist_th, dot_th, u_hi, v_hi, w.pix_of_point, w.flags, w.pix_key, H, W, (const long long*)nullptr);
        hipLaunchKernelGGL(k_pf_assoc2, dim3(grid_for(M)), dim3(PF_T), 0, st, map_points, map_ccounts, M, Vg, w.pix_of_point, w.flags,
                           w.pix_key, w.pix_best, w.any_match, (const long long*)nullptr);
    }
    E2E_LAUNCH_CHECK("e2e_pf_associate");
    return E2E_OK;
}

/* e2e_pf_associate on a RESIDENT map whose live size is device data: map_count_dev[0] = M (int64, 0 <= M <= map_capacity).  The
 * host never reads it -- grids are sized by the capacity, every argument is constant from one keyframe to the next. */
int e2e_pf_associate_dev(const float* map_points, const float* map_normals, const float* map_ccounts, const long long* map_count_dev,
                         const float* K, const float* pose, const float* Vg, const float* Ng, float dist_th, float dot_th,
                         void* workspace, int64_t map_capacity, int H, int W, void* stream) {
    E2E_REQUIRE(map_capacity > 0 && map_capacity < (1ll << 32) - 1 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), E2E_ERR_ARG,
                "e2e_pf_associate_dev: bad sizes cap=%lld", (long long)map_capacity);
    E2E_REQUIRE(K && pose && Vg && Ng && workspace && map_points && map_normals && map_ccounts && map_count_dev, E2E_ERR_ARG,
                "e2e_pf_associate_dev: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(N, 256, 1024)), dim3(256), 0, st, w.pix_key, N, PF_KEY_NONE);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(N, 256, 1024)), dim3(256), 0, st, w.pix_best, N, PF_NONE);
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(64), 0, st, w.any_match, 4, 0u);
    const float u_hi = (float)((double)W - 0.999), v_hi = (float)((double)H - 0.999);
    hipLaunchKernelGGL(k_pf_assoc1, dim3(grid_for(map_capacity)), dim3(PF_T), 0, st, map_points, map_normals, map_ccounts, (int64_t)0, K, pose, Vg, Ng,
                       dist_th, dot_th, u_hi, v_hi, w.pix_of_point, w.flags, w.pix_key, H, W, map_count_dev);
    hipLaunchKernelGGL(k_pf_assoc2, dim3(grid_for(map_capacity)), dim3(PF_T), 0, st, map_points, map_ccounts, (int64_t)0, Vg, w.pix_of_point, w.flags,
                       w.pix_key, w.pix_best, w.any_match, map_count_dev);
    E2E_LAUNCH_CHECK("e2e_pf_associate_dev");
    return E2E_OK;
}

// ---- frame-to-model odometry: the ACTIVE map points (find_active_map_points) of the last association, every dsratio-th of them in
// ascending map order (gradslam: downsample_pointclouds(pc, pc2im_bnhw, dsratio) = pc2im_bnhw[::dsratio]), gathered with their normals
// into a dense target cloud.  The live map size and the number of targets are DEVICE data. ----
__global__ __launch_bounds__(PF_T) void k_cp_count_active_dev(int64_t cap, const long long* __restrict__ count_dev, const unsigned char* __restrict__ flags,
                                                              unsigned int* __restrict__ counts) {
    __shared__ unsigned int sh[PF_T / 64];
    const int64_t M = *count_dev < cap ? *count_dev : cap;
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (base + j < M && flags[base + j] != 0) ++c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(PF_T) void k_gather_active_sub(int64_t cap, const long long* __restrict__ count_dev, const unsigned char* __restrict__ flags,
                                                            const unsigned int* __restrict__ offsets, int ds, const float* __restrict__ points,
                                                            const float* __restrict__ normals, float* __restrict__ tgt, float* __restrict__ tgt_n,
                                                            int64_t tgt_capacity, long long* __restrict__ tgt_count /* {targets, active points, overflow} */) {
    __shared__ unsigned int sh[PF_T / 64];
    const int64_t M = *count_dev < cap ? *count_dev : cap;
    const int64_t base = (int64_t)blockIdx.x * CP_BLOCK + (int64_t)threadIdx.x * CP_ITEMS;
    bool pr[CP_ITEMS];
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j) {
        pr[j] = base + j < M && flags[base + j] != 0;
        c += pr[j];
    }
    unsigned int o = offsets[blockIdx.x] + cp_block_offset(c, sh);
#pragma unroll
    for (int j = 0; j < CP_ITEMS; ++j)
        if (pr[j]) {
            if (o % (unsigned int)ds == 0) {
                const int64_t r = o / (unsigned int)ds, n = base + j;
                if (r < tgt_capacity) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { tgt[r * 3 + k] = points[n * 3 + k]; tgt_n[r * 3 + k] = normals[n * 3 + k]; }
                } else {
                    tgt_count[2] = 1;
                }
            }
            ++o;
        }
}

__global__ void k_active_sub_count(long long* __restrict__ tgt_count, int ds, long long tgt_capacity) {
    const long long P = tgt_count[1];
    long long t = (P + ds - 1) / ds;
    tgt_count[0] = t > tgt_capacity ? tgt_capacity : t;
}

int e2e_pf_active_subsample_dev(const float* map_points, const float* map_normals, const long long* map_count_dev, int64_t map_capacity,
                                void* workspace, int H, int W, int dsratio, float* tgt, float* tgt_normals, long long* tgt_count_dev,
                                int64_t tgt_capacity, void* stream) {
    E2E_REQUIRE(map_points && map_normals && map_count_dev && workspace && tgt && tgt_normals && tgt_count_dev, E2E_ERR_ARG,
                "e2e_pf_active_subsample_dev: null pointer");
    E2E_REQUIRE(map_capacity > 0 && dsratio > 0 && tgt_capacity > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_pf_active_subsample_dev: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int nb = (int)((map_capacity + CP_BLOCK - 1) / CP_BLOCK);
    hipLaunchKernelGGL(k_cp_count_active_dev, dim3(nb), dim3(PF_T), 0, st, map_capacity, map_count_dev, w.flags, w.counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, tgt_count_dev + 1, 0ll, (const long long*)nullptr);
    hipLaunchKernelGGL(k_gather_active_sub, dim3(nb), dim3(PF_T), 0, st, map_capacity, map_count_dev, w.flags, w.counts, dsratio, map_points,
                       map_normals, tgt, tgt_normals, tgt_capacity, tgt_count_dev);
    hipLaunchKernelGGL(k_active_sub_count, dim3(1), dim3(1), 0, st, tgt_count_dev, dsratio, (long long)tgt_capacity);
    E2E_LAUNCH_CHECK("e2e_pf_active_subsample_dev");
    return E2E_OK;
}

int e2e_pf_table(int which, int64_t M, void* workspace, int64_t map_capacity, int H, int W, long long* rows,
                 long long* count_out, void* stream) {
    E2E_REQUIRE(which >= 0 && which <= 2 && M >= 0 && M <= map_capacity && workspace && rows && count_out, E2E_ERR_ARG,
                "e2e_pf_table: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int64_t N = (int64_t)H * W;
    if (which < 2) {
        const int nb = (int)((M + CP_BLOCK - 1) / CP_BLOCK);
        if (nb == 0) {
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, 0, count_out, 0ll, (const long long*)nullptr);
        } else if (which == 0) {
            hipLaunchKernelGGL(k_cp_count<PRED_ACTIVE>, dim3(nb), dim3(PF_T), 0, st, M, w.flags, nullptr, nullptr, w.counts);
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, count_out, 0ll, (const long long*)nullptr);
            hipLaunchKernelGGL(k_cp_point_rows<PRED_ACTIVE>, dim3(nb), dim3(PF_T), 0, st, M, w.flags, w.pix_of_point, w.counts, W, rows);
        } else {
            hipLaunchKernelGGL(k_cp_count<PRED_SIMILAR>, dim3(nb), dim3(PF_T), 0, st, M, w.flags, nullptr, nullptr, w.counts);
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, count_out, 0ll, (const long long*)nullptr);
            hipLaunchKernelGGL(k_cp_point_rows<PRED_SIMILAR>, dim3(nb), dim3(PF_T), 0, st, M, w.flags, w.pix_of_point, w.counts, W, rows);
        }
    } else {
        const int nb = (int)((N + CP_BLOCK - 1) / CP_BLOCK);
        hipLaunchKernelGGL(k_cp_count<PRED_PIX_MATCHED>, dim3(nb), dim3(PF_T), 0, st, N, nullptr, w.pix_best, nullptr, w.counts);
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, count_out, 0ll, (const long long*)nullptr);
        hipLaunchKernelGGL(k_cp_pixel_rows, dim3(nb), dim3(PF_T), 0, st, N, w.pix_best, w.counts, W, rows);
    }
    E2E_LAUNCH_CHECK("e2e_pf_table");
    return E2E_OK;
}

int e2e_pf_fuse_append(float* map_points, float* map_normals, float* map_colors, float* map_ccounts, int64_t M,
                       int64_t map_capacity, const float* depth, const float* Vg, const float* Ng, const float* rgb,
                       const float* alpha, void* workspace, int H, int W, long long* new_count_out, void* stream) {
    E2E_REQUIRE(M >= 0 && M <= map_capacity && H > 0 && W > 0, E2E_ERR_ARG, "e2e_pf_fuse_append: bad sizes");
    E2E_REQUIRE(map_points && map_normals && map_colors && map_ccounts && depth && Vg && Ng && rgb && alpha && workspace && new_count_out,
                E2E_ERR_ARG, "e2e_pf_fuse_append: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int64_t N = (int64_t)H * W;
    if (M > 0)
        hipLaunchKernelGGL(k_pf_fuse, dim3(grid_for(M)), dim3(PF_T), 0, st, map_points, map_normals, map_colors, map_ccounts, M,
                           w.pix_of_point, w.flags, w.pix_best, Vg, Ng, rgb, alpha, w.any_match, (const long long*)nullptr);
    const int nb = (int)((N + CP_BLOCK - 1) / CP_BLOCK);
    hipLaunchKernelGGL(k_cp_count<PRED_PIX_NEW>, dim3(nb), dim3(PF_T), 0, st, N, nullptr, w.pix_best, depth, w.counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, new_count_out, (long long)M, (const long long*)nullptr);
    hipLaunchKernelGGL(k_pf_append, dim3(nb), dim3(PF_T), 0, st, N, M, w.pix_best, depth, w.counts, Vg, Ng, rgb, alpha, map_points,
                       map_normals, map_colors, map_ccounts, map_capacity, (const long long*)nullptr);
    E2E_LAUNCH_CHECK("e2e_pf_fuse_append");
    return E2E_OK;
}

// commit of a resident map step: count[0] = min(count[1], cap); an overflowing append (its points beyond `cap` were dropped by
// k_pf_append) raises the sticky flag count[2], which the host reads once at the end of a pass
__global__ void k_pf_commit(long long* count, long long cap) {
    const long long m = count[1];
    if (m > cap) count[2] = m;
    count[0] = m > cap ? cap : m;
}

/* e2e_pf_fuse_append on a RESIDENT map: map_count_dev = int64[3] in device memory {M (in: live size, out: size after the append),
 * scratch, sticky overflow flag (0, or the size an append would have needed when it exceeded map_capacity)}.  No host read-back:
 * the caller checks map_count_dev[2] whenever it synchronises anyway. */
int e2e_pf_fuse_append_dev(float* map_points, float* map_normals, float* map_colors, float* map_ccounts, long long* map_count_dev,
                           int64_t map_capacity, const float* depth, const float* Vg, const float* Ng, const float* rgb,
                           const float* alpha, void* workspace, int H, int W, void* stream) {
    E2E_REQUIRE(map_capacity > 0 && H > 0 && W > 0, E2E_ERR_ARG, "e2e_pf_fuse_append_dev: bad sizes");
    E2E_REQUIRE(map_points && map_normals && map_colors && map_ccounts && depth && Vg && Ng && rgb && alpha && workspace && map_count_dev,
                E2E_ERR_ARG, "e2e_pf_fuse_append_dev: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const PfWs w = pf_ws(workspace, map_capacity, H, W);
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_pf_fuse, dim3(grid_for(map_capacity)), dim3(PF_T), 0, st, map_points, map_normals, map_colors, map_ccounts, (int64_t)0,
                       w.pix_of_point, w.flags, w.pix_best, Vg, Ng, rgb, alpha, w.any_match, (const long long*)map_count_dev);
    const int nb = (int)((N + CP_BLOCK - 1) / CP_BLOCK);
    hipLaunchKernelGGL(k_cp_count<PRED_PIX_NEW>, dim3(nb), dim3(PF_T), 0, st, N, nullptr, w.pix_best, depth, w.counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, w.counts, nb, map_count_dev + 1, 0ll, (const long long*)map_count_dev);
    hipLaunchKernelGGL(k_pf_append, dim3(nb), dim3(PF_T), 0, st, N, (int64_t)0, w.pix_best, depth, w.counts, Vg, Ng, rgb, alpha, map_points,
                       map_normals, map_colors, map_ccounts, map_capacity, (const long long*)map_count_dev);
    hipLaunchKernelGGL(k_pf_commit, dim3(1), dim3(1), 0, st, map_count_dev, (long long)map_capacity);
    E2E_LAUNCH_CHECK("e2e_pf_fuse_append_dev");
    return E2E_OK;
}

}  // extern "C"
