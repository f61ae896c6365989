// icp.hip -- the reduction step of frame-to-model point-to-plane ICP (gradslam odometry providers "icp" / "gradicp",
// MODEL.odom in configs/config.yaml:30).  Per Gauss-Newton iteration the GPU does: rigid
// transform of the source cloud (pointfusion.hip), exact nearest neighbours (knn.hip) and THIS kernel, which folds
// the N x 6 system  A_i = [n_i, s_i x n_i],  b_i = n_i . (t_i - s_i)  into the 21 + 6 numbers of A^T A and A^T b
// (+ inlier count and sum of squared residuals).  The 6x6 solve and the se(3) exponential run on the host in float64.
// Accumulation is in float64 with a fixed order (per-thread -> wave shuffle -> per-workgroup partials -> one
// workgroup), so the pose is reproducible run to run.
#include "e2e_common.h"

#define ICP_T 256
#define ICP_NACC 29          // 21 (upper triangle of AtA) + 6 (Atb) + count + sum r^2

__global__ __launch_bounds__(ICP_T) void k_icp_partials(const float* __restrict__ src, const float* __restrict__ tgt,
                                                        const float* __restrict__ tgt_n, const long long* __restrict__ idx,
                                                        const float* __restrict__ dists, float thresh_sq, int64_t n,
                                                        double* __restrict__ partials) {
    __shared__ double sh[ICP_T / 64];
    double acc[ICP_NACC];
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * ICP_T + threadIdx.x; i < n; i += (int64_t)gridDim.x * ICP_T) {
        if (thresh_sq >= 0.f && !(dists[i] < thresh_sq)) continue;
        const long long j = idx[i];
        const double s[3] = {src[i * 3], src[i * 3 + 1], src[i * 3 + 2]};
        const double t[3] = {tgt[j * 3], tgt[j * 3 + 1], tgt[j * 3 + 2]};
        const double nn[3] = {tgt_n[j * 3], tgt_n[j * 3 + 1], tgt_n[j * 3 + 2]};
        double a[6] = {nn[0], nn[1], nn[2], s[1] * nn[2] - s[2] * nn[1], s[2] * nn[0] - s[0] * nn[2], s[0] * nn[1] - s[1] * nn[0]};
        const double b = nn[0] * (t[0] - s[0]) + nn[1] * (t[1] - s[1]) + nn[2] * (t[2] - s[2]);
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) acc[k++] += a[r] * a[c];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] += a[r] * b;
        acc[27] += 1.0;
        acc[28] += b * b;
    }
    for (int k = 0; k < ICP_NACC; ++k) {
        const double v = wave_sum_d(acc[k]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) partials[(int64_t)blockIdx.x * ICP_NACC + k] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    }
}

__global__ void k_icp_final(const double* __restrict__ partials, int nparts, double* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= ICP_NACC) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += partials[(int64_t)i * ICP_NACC + k];
    out[k] = s;
}

extern "C" {

#define ICP_MAX_PARTS 256
int64_t e2e_icp_workspace_bytes(void) { return (int64_t)ICP_MAX_PARTS * ICP_NACC * 8; }

int e2e_icp_normal_equations(const float* src, const float* tgt, const float* tgt_normals, const long long* idx, const float* dists,
                             float dist_thresh, int64_t n, double* out29, void* workspace, void* stream) {
    E2E_REQUIRE(src && tgt && tgt_normals && idx && out29 && workspace && n > 0, E2E_ERR_ARG, "e2e_icp_normal_equations: bad argument");
    E2E_REQUIRE(dist_thresh < 0.f || dists, E2E_ERR_ARG, "e2e_icp_normal_equations: a distance threshold needs the distances");
    hipStream_t st = (hipStream_t)stream;
    int g = (int)((n + ICP_T - 1) / ICP_T);
    if (g > ICP_MAX_PARTS) g = ICP_MAX_PARTS;
    hipLaunchKernelGGL(k_icp_partials, dim3(g), dim3(ICP_T), 0, st, src, tgt, tgt_normals, idx, dists,
                       dist_thresh < 0.f ? -1.f : dist_thresh * dist_thresh, n, (double*)workspace);
    hipLaunchKernelGGL(k_icp_final, dim3(1), dim3(64), 0, st, (const double*)workspace, g, out29);
    E2E_LAUNCH_CHECK("e2e_icp_normal_equations");
    return E2E_OK;
}

}  // extern "C"
