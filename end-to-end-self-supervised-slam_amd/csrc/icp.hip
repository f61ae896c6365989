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
    __shared__ double sh[ICP_NACC][ICP_T / 64];
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
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) {
        const double v = wave_sum_d(acc[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();                                           // ONE barrier for the 29 sums (round 3: two per sum)
    if (threadIdx.x < ICP_NACC) partials[(int64_t)blockIdx.x * ICP_NACC + threadIdx.x] = ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}

// one wave per accumulator: lane l adds the partials l, l + 64, ... in order, then a fixed shuffle tree (reproducible; a single thread
// walking all partials took 18 us for 75 workgroups' worth)
__global__ __launch_bounds__(64) void k_icp_final(const double* __restrict__ partials, int nparts, double* __restrict__ out) {
    const int k = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partials[(int64_t)i * ICP_NACC + k];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[k] = s;
}

// ---------------------------------------------------------------------------------------------
// The Gauss-Newton / Levenberg-Marquardt update ON THE DEVICE (round 4): the 6x6 solve, the se(3) exponential, GradICP's smooth
// damping update and the pose composition in float64 by one thread, so that the numiters iterations of a keyframe's odometry are a
// fixed launch sequence with no host round trip (round 3: one 232-byte read-back + np.linalg.solve + upload per reduction, 40 per
// keyframe, 9.4 ms per refinement step).  Same arithmetic as oracle/icp.py: LU with partial pivoting (what numpy's solve calls),
// Rodrigues + the V matrix with the small-angle branch below 1e-8.
// state (float64): [0..15] T (row-major), [16..21] xi, [22] lambda, [23] err/cnt of the first reduction, [24] stopped, [25] iterations
// done, [26] damp, [32 + 2k], [33 + 2k]: (inlier count, sum r^2) of iteration k (trace, up to ICP_TRACE iterations)
// ---------------------------------------------------------------------------------------------
#define ICP_TRACE 64
#define ICP_STATE (32 + 2 * ICP_TRACE)

__device__ void icp_se3_exp(const double* xi, double* T) {
    const double v[3] = {xi[0], xi[1], xi[2]}, w[3] = {xi[3], xi[4], xi[5]};
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W2[r * 3 + c] = W[r * 3] * W[c] + W[r * 3 + 1] * W[3 + c] + W[r * 3 + 2] * W[6 + c];
    double a, b, c2;       // R = I + a W + b W^2 ; V = I + b W + c2 W^2
    if (th < 1e-8) { a = 1.0; b = 0.5; c2 = 1.0 / 6.0; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / (th * th); c2 = (th - sin(th)) / (th * th * th); }
    double V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        T[(i / 3) * 4 + (i % 3)] = I + a * W[i] + b * W2[i];
        V[i] = I + b * W[i] + c2 * W2[i];
    }
    for (int r = 0; r < 3; ++r) T[r * 4 + 3] = V[r * 3] * v[0] + V[r * 3 + 1] * v[1] + V[r * 3 + 2] * v[2];
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

// x = A^-1 b for a 6x6 system, LU with partial pivoting (row swaps on the largest |pivot| of the column)
__device__ void icp_solve6(double* A, double* b, double* x) {
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double big = fabs(A[k * 6 + k]);
        for (int r = k + 1; r < 6; ++r)
            if (fabs(A[r * 6 + k]) > big) { big = fabs(A[r * 6 + k]); p = r; }
        if (p != k) {
            for (int c = 0; c < 6; ++c) { const double t = A[k * 6 + c]; A[k * 6 + c] = A[p * 6 + c]; A[p * 6 + c] = t; }
            const double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        for (int r = k + 1; r < 6; ++r) {
            const double f = A[r * 6 + k] / A[k * 6 + k];
            for (int c = k; c < 6; ++c) A[r * 6 + c] -= f * A[k * 6 + c];
            b[r] -= f * b[k];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
        for (int c = r + 1; c < 6; ++c) s -= A[r * 6 + c] * x[c];
        x[r] = s / A[r * 6 + r];
    }
}

__device__ void icp_compose(const double* step, double* state, float* T32, const float* prev_pose, float* pose_out) {
    double Tn[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += step[r * 4 + k] * state[k * 4 + c];
            Tn[r * 4 + c] = s;
        }
    for (int i = 0; i < 16; ++i) { state[i] = Tn[i]; T32[i] = (float)Tn[i]; }
    if (pose_out)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Tn[r * 4 + k] * (double)prev_pose[k * 4 + c];
                pose_out[r * 4 + c] = (float)s;
            }
}

__global__ void k_icp_init(double* state, float* T32, float* step32, const float* prev_pose, float* pose_out, double damp) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < ICP_STATE; ++i) state[i] = 0.0;
    for (int i = 0; i < 16; ++i) { const double v = (i % 5 == 0) ? 1.0 : 0.0; state[i] = v; T32[i] = (float)v; step32[i] = (float)v; }
    state[22] = damp;
    state[26] = damp;
    if (pose_out) for (int i = 0; i < 16; ++i) pose_out[i] = prev_pose[i];
}

// mode 0 = icp (phase 0 only), 1 = gradicp (phase 0: solve + trial step; phase 1: damping update from the trial step's error, gated step)
// partials != NULL: the 29 sums are first folded from the workgroup partials of k_icp_partials (by the whole wave, as k_icp_final does)
// -- one launch less per reduction than k_icp_final + k_icp_update
__global__ __launch_bounds__(64) void k_icp_update(const double* __restrict__ out29_in, const double* __restrict__ partials, int nparts,
                             double* __restrict__ state, float* __restrict__ T32, float* __restrict__ step32,
                             const float* __restrict__ prev_pose, float* __restrict__ pose_out, int mode, int phase, double lambda_max,
                             double B, double B2, double nu) {
    __shared__ double sums[ICP_NACC];
    const double* out29 = out29_in;
    if (partials) {
        double v[ICP_NACC];                                 // lane l adds the rows l, l + 64, ... of all 29 accumulators (independent loads: one
#pragma unroll                                             // accumulator at a time was 29 dependent round trips, 20 us), then 29 fixed shuffle trees
        for (int k = 0; k < ICP_NACC; ++k) v[k] = 0.0;
        for (int i = threadIdx.x; i < nparts; i += 64) {
#pragma unroll
            for (int k = 0; k < ICP_NACC; ++k) v[k] += partials[(int64_t)i * ICP_NACC + k];
        }
#pragma unroll
        for (int k = 0; k < ICP_NACC; ++k) {
            const double t = wave_sum_d(v[k]);
            if (threadIdx.x == 0) sums[k] = t;
        }
        __syncthreads();
        out29 = sums;
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[24] != 0.0) return;                          // stopped (fewer than 6 inliers at some iteration): the pose stays
    const double cnt = out29[27], err = out29[28];
    double step[16];
    if (phase == 0) {
        if (cnt < 6.0) {
            state[24] = 1.0;
            for (int i = 0; i < 16; ++i) step32[i] = (i % 5 == 0) ? 1.f : 0.f;
            return;
        }
        double A[36], b[6], xi[6];
        int k = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) { A[r * 6 + c] = A[c * 6 + r] = out29[k]; ++k; }
        for (int r = 0; r < 6; ++r) { A[r * 6 + r] += state[22]; b[r] = out29[21 + r]; }
        icp_solve6(A, b, xi);
        for (int r = 0; r < 6; ++r) state[16 + r] = xi[r];
        const int it = (int)state[25];
        if (it < ICP_TRACE) { state[32 + 2 * it] = cnt; state[33 + 2 * it] = err; }
        icp_se3_exp(xi, step);
        if (mode == 0) {
            icp_compose(step, state, T32, prev_pose, pose_out);
            state[25] += 1.0;
        } else {
            state[23] = err / (cnt > 1.0 ? cnt : 1.0);
            for (int i = 0; i < 16; ++i) step32[i] = (float)step[i];
        }
    } else {
        const double delta = err / (cnt > 1.0 ? cnt : 1.0) - state[23];
        state[22] *= 1.0 / lambda_max + (lambda_max - 1.0 / lambda_max) / (1.0 + B * exp(-B2 * nu * delta));
        double z = nu * delta;
        z = z < -60.0 ? -60.0 : (z > 60.0 ? 60.0 : z);
        const double gate = 1.0 / (1.0 + exp(z));
        double xi[6];
        for (int r = 0; r < 6; ++r) xi[r] = gate * state[16 + r];
        icp_se3_exp(xi, step);
        icp_compose(step, state, T32, prev_pose, pose_out);
        state[25] += 1.0;
    }
}

// source cloud of the odometry: every dsratio-th pixel (both directions) of the live frame's world vertex map, row-major.
// gradslam drops pixels without depth; a resident launch sequence has a fixed number of queries, so a pixel without depth raises
// *status instead (network-predicted depths 1 / disp are positive everywhere: it never fires on the path this serves).
__global__ __launch_bounds__(256) void k_icp_source(const float* __restrict__ Vg, const float* __restrict__ depth, int H, int W, int ds,
                                                    float* __restrict__ src, int* __restrict__ status) {
    const int hs = (H + ds - 1) / ds, wsub = (W + ds - 1) / ds;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hs * wsub; i += gridDim.x * 256) {
        const int64_t q = (int64_t)(i / wsub) * ds * W + (int64_t)(i % wsub) * ds;
        if (depth[q] == 0.f) atomicOr(status, 1);
        src[i * 3 + 0] = Vg[q * 3 + 0];
        src[i * 3 + 1] = Vg[q * 3 + 1];
        src[i * 3 + 2] = Vg[q * 3 + 2];
    }
}

extern "C" {

int64_t e2e_icp_state_doubles(void) { return ICP_STATE; }

int e2e_icp_state_init(double* state, float* T32, float* step32, const float* prev_pose, float* pose_out, double damp, void* stream) {
    E2E_REQUIRE(state && T32 && step32 && (!pose_out || prev_pose), E2E_ERR_ARG, "e2e_icp_state_init: bad argument");
    hipLaunchKernelGGL(k_icp_init, dim3(1), dim3(64), 0, (hipStream_t)stream, state, T32, step32, prev_pose, pose_out, damp);
    E2E_LAUNCH_CHECK("e2e_icp_state_init");
    return E2E_OK;
}

int e2e_icp_update(const double* out29, double* state, float* T32, float* step32, const float* prev_pose, float* pose_out, int mode, int phase,
                   double lambda_max, double B, double B2, double nu, void* stream) {
    E2E_REQUIRE(out29 && state && T32 && step32 && (!pose_out || prev_pose), E2E_ERR_ARG, "e2e_icp_update: null pointer");
    E2E_REQUIRE((mode == 0 && phase == 0) || (mode == 1 && (phase == 0 || phase == 1)), E2E_ERR_ARG, "e2e_icp_update: mode 0 (icp, phase 0) or 1 (gradicp, phase 0 / 1)");
    hipLaunchKernelGGL(k_icp_update, dim3(1), dim3(64), 0, (hipStream_t)stream, out29, (const double*)nullptr, 0, state, T32, step32, prev_pose, pose_out, mode,
                       phase, lambda_max, B, B2, nu);
    E2E_LAUNCH_CHECK("e2e_icp_update");
    return E2E_OK;
}

int e2e_icp_source_subsample(const float* Vg, const float* depth, int H, int W, int dsratio, float* src, int* status, void* stream) {
    E2E_REQUIRE(Vg && depth && src && status && H > 0 && W > 0 && dsratio > 0, E2E_ERR_ARG, "e2e_icp_source_subsample: bad argument");
    const int n = ((H + dsratio - 1) / dsratio) * ((W + dsratio - 1) / dsratio);
    hipLaunchKernelGGL(k_icp_source, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, Vg, depth, H, W, dsratio, src, status);
    E2E_LAUNCH_CHECK("e2e_icp_source_subsample");
    return E2E_OK;
}

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
    hipLaunchKernelGGL(k_icp_final, dim3(ICP_NACC), dim3(64), 0, st, (const double*)workspace, g, out29);
    E2E_LAUNCH_CHECK("e2e_icp_normal_equations");
    return E2E_OK;
}

/* e2e_icp_normal_equations + e2e_icp_update in two launches instead of three: the update kernel folds the workgroup partials itself. */
int e2e_icp_reduce_update(const float* src, const float* tgt, const float* tgt_normals, const long long* idx, const float* dists, float dist_thresh,
                          int64_t n, void* workspace, double* state, float* T32, float* step32, const float* prev_pose, float* pose_out, int mode,
                          int phase, double lambda_max, double B, double B2, double nu, void* stream) {
    E2E_REQUIRE(src && tgt && tgt_normals && idx && workspace && state && T32 && step32 && n > 0 && (!pose_out || prev_pose), E2E_ERR_ARG,
                "e2e_icp_reduce_update: bad argument");
    E2E_REQUIRE(dist_thresh < 0.f || dists, E2E_ERR_ARG, "e2e_icp_reduce_update: a distance threshold needs the distances");
    E2E_REQUIRE((mode == 0 && phase == 0) || (mode == 1 && (phase == 0 || phase == 1)), E2E_ERR_ARG, "e2e_icp_reduce_update: mode 0 (icp, phase 0) or 1 (gradicp, phase 0 / 1)");
    hipStream_t st = (hipStream_t)stream;
    int g = (int)((n + ICP_T - 1) / ICP_T);
    if (g > ICP_MAX_PARTS) g = ICP_MAX_PARTS;
    hipLaunchKernelGGL(k_icp_partials, dim3(g), dim3(ICP_T), 0, st, src, tgt, tgt_normals, idx, dists, dist_thresh < 0.f ? -1.f : dist_thresh * dist_thresh, n,
                       (double*)workspace);
    hipLaunchKernelGGL(k_icp_update, dim3(1), dim3(64), 0, st, (const double*)nullptr, (const double*)workspace, g, state, T32, step32, prev_pose, pose_out, mode,
                       phase, lambda_max, B, B2, nu);
    E2E_LAUNCH_CHECK("e2e_icp_reduce_update");
    return E2E_OK;
}

}  // extern "C"
