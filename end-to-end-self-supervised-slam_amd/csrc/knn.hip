// knn.hip -- K=1 nearest neighbour in 3-D (squared L2) for gfx950: chamferdist.knn_points as used by
// loss/losses.py:57-61 (knn_points_loss) through online_adaption.py:638-645.
//
// Exact brute force, bit-identical to oracle/knn_brute.c: d = ((dx*dx + dy*dy) + dz*dz) in fp32 with
// no FMA (-ffp-contract=off), first minimum wins on ties (= smallest reference index).
// Roofline: fp32 VALU (K-dimension 3 is too thin for MFMA): 8 flop + compare/select per pair.
// Mapping: one query per lane held in registers; reference points stream through LDS in 1024-point
// tiles as float4 (one broadcast ds_read_b128 per point per wave).  The reference set is split into
// `slices` ranges (grid.y) so that small query sets still fill 256 CUs; each (query, slice) result is
// merged with a 64-bit integer atomicMin on (bits(dist) << 32 | index) -- order independent, so the
// result is deterministic and ties resolve to the smallest index exactly as a sequential scan does.
#include "e2e_common.h"

#define KT 256
#define KTILE 1024

__global__ __launch_bounds__(KT) void k_knn1(const float* __restrict__ p1, int64_t n1, const float* __restrict__ p2, int64_t n2,
                                             int64_t slice_len, unsigned long long* __restrict__ best) {
    __shared__ float4 tile[KTILE];
    const int64_t i = (int64_t)blockIdx.x * KT + threadIdx.x;
    const bool live = i < n1;
    const float x = live ? p1[i * 3] : 0.f, y = live ? p1[i * 3 + 1] : 0.f, z = live ? p1[i * 3 + 2] : 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * slice_len;
    const int64_t j1 = (j0 + slice_len < n2) ? j0 + slice_len : n2;
    float bd = 3.402823466e38f;
    unsigned int bi = 0xFFFFFFFFu;
    for (int64_t t0 = j0; t0 < j1; t0 += KTILE) {
        const int cnt = (int)((j1 - t0 < KTILE) ? (j1 - t0) : KTILE);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += KT) {
            const float* q = p2 + (t0 + k) * 3;
            tile[k] = make_float4(q[0], q[1], q[2], 0.f);
        }
        __syncthreads();
        const unsigned int base = (unsigned int)t0;
#pragma unroll 8
        for (int k = 0; k < cnt; ++k) {
            const float4 q = tile[k];
            const float dx = x - q.x, dy = y - q.y, dz = z - q.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < bd) {          // strict: the first minimum wins
                bd = d;
                bi = base + (unsigned int)k;
            }
        }
    }
    if (live && bi != 0xFFFFFFFFu) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)bi;
        atomicMin(best + i, key);
    }
}

__global__ void k_knn1_init(unsigned long long* best, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) best[i] = 0xFFFFFFFFFFFFFFFFull;
}

// unpack + (optionally) block partial sums of the distances for the mean (fixed-order second stage on the host side op)
__global__ __launch_bounds__(KT) void k_knn1_unpack(const unsigned long long* __restrict__ best, int64_t n, float* __restrict__ dists,
                                                    long long* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * KT + threadIdx.x; i < n; i += (int64_t)gridDim.x * KT) {
        const unsigned long long k = best[i];
        dists[i] = __uint_as_float((unsigned int)(k >> 32));
        idx[i] = (long long)(k & 0xFFFFFFFFull);
    }
}

// backward of dists wrt p1:  g_p1 = 2 g (p1 - p2[idx])      (losses.py:57-61 through chamferdist autograd)
__global__ __launch_bounds__(KT) void k_knn1_bwd(const float* __restrict__ g, const float* __restrict__ p1, const float* __restrict__ p2,
                                                 const long long* __restrict__ idx, int64_t n1, float* __restrict__ gp1) {
    for (int64_t i = (int64_t)blockIdx.x * KT + threadIdx.x; i < n1; i += (int64_t)gridDim.x * KT) {
        const long long j = idx[i];
        const float gi = 2.f * g[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) gp1[i * 3 + c] = gi * (p1[i * 3 + c] - p2[j * 3 + c]);
    }
}

extern "C" {

int64_t e2e_knn1_workspace_bytes(int64_t n1) { return n1 > 0 ? 8 * n1 : 0; }

int e2e_knn1_fwd(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists, long long* idx, void* workspace,
                 void* stream) {
    E2E_REQUIRE(n1 > 0 && n2 > 0 && n2 < 0xFFFFFFFFll, E2E_ERR_ARG, "e2e_knn1_fwd: bad sizes n1=%lld n2=%lld", (long long)n1, (long long)n2);
    E2E_REQUIRE(p1 && p2 && dists && idx && workspace, E2E_ERR_ARG, "e2e_knn1_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* best = (unsigned long long*)workspace;
    const int qblocks = (int)((n1 + KT - 1) / KT);
    // enough (query-block, slice) workgroups for ~8 per CU; slices are whole tiles
    int64_t slices = (2048 + qblocks - 1) / qblocks;
    const int64_t tiles = (n2 + KTILE - 1) / KTILE;
    if (slices > tiles) slices = tiles;
    if (slices < 1) slices = 1;
    const int64_t slice_len = ((tiles + slices - 1) / slices) * KTILE;
    slices = (n2 + slice_len - 1) / slice_len;
    hipLaunchKernelGGL(k_knn1_init, dim3((unsigned)((n1 + 255) / 256 > 2048 ? 2048 : (n1 + 255) / 256)), dim3(256), 0, st, best, n1);
    hipLaunchKernelGGL(k_knn1, dim3(qblocks, (unsigned)slices), dim3(KT), 0, st, p1, n1, p2, n2, slice_len, best);
    hipLaunchKernelGGL(k_knn1_unpack, dim3((unsigned)(qblocks > 4096 ? 4096 : qblocks)), dim3(KT), 0, st, best, n1, dists, idx);
    E2E_LAUNCH_CHECK("e2e_knn1_fwd");
    return E2E_OK;
}

int e2e_knn1_bwd(const float* g_dists, const float* p1, const float* p2, const long long* idx, int64_t n1, float* g_p1, void* stream) {
    E2E_REQUIRE(n1 > 0 && g_dists && p1 && p2 && idx && g_p1, E2E_ERR_ARG, "e2e_knn1_bwd: bad argument");
    const int qblocks = (int)((n1 + KT - 1) / KT);
    hipLaunchKernelGGL(k_knn1_bwd, dim3((unsigned)(qblocks > 4096 ? 4096 : qblocks)), dim3(KT), 0, (hipStream_t)stream, g_dists, p1, p2, idx, n1, g_p1);
    E2E_LAUNCH_CHECK("e2e_knn1_bwd");
    return E2E_OK;
}

}  // extern "C"
