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

// gradient wrt the REFERENCE cloud: g_p2[idx[i]] += -2 g[i] (p1[i] - p2[idx[i]]) -- a scatter with collisions (several queries share a
// neighbour), accumulated as 2^-48 fixed-point integers so that the sum does not depend on the arrival order (bitwise reproducible;
// |sum| < 32768).  A contribution that is not finite or >= 4096 is refused and poisons the result (NaN everywhere) instead of wrapping.
__global__ __launch_bounds__(KT) void k_knn1_bwd_ref(const float* __restrict__ g, const float* __restrict__ p1, const float* __restrict__ p2,
                                                     const long long* __restrict__ idx, int64_t n1, int64_t n2, unsigned long long* __restrict__ fx) {
    for (int64_t i = (int64_t)blockIdx.x * KT + threadIdx.x; i < n1; i += (int64_t)gridDim.x * KT) {
        const long long j = idx[i];
        const float gi = -2.f * g[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = gi * (p1[i * 3 + c] - p2[j * 3 + c]);
            if (fabsf(v) < 4096.f) atomicAdd(fx + j * 3 + c, (unsigned long long)__double2ll_rn((double)v * 281474976710656.0));
            else atomicOr(fx + n2 * 3, 1ull);
        }
    }
}
__global__ void k_knn1_fixed48_to_float(const long long* __restrict__ fx, float* __restrict__ out, int64_t n) {
    const bool poisoned = fx[n] != 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = poisoned ? __uint_as_float(0x7FC00000u) : (float)((double)fx[i] * (1.0 / 281474976710656.0));
}


// ---------------------------------------------------------------------------------------------
// Exact uniform-grid acceleration (results identical to the brute force, ties included).
//   build : bbox (integer atomics on order-preserving keys) -> cell size h = max(extent/(G-1), H_MIN), G = 128 or 256 ->
//           per-cell counts -> exclusive scan -> counting-sort scatter of (x,y,z,index) float4s
//   query : one query per lane; visit the cube of radius r cells around the query's cell, shell by shell, keeping
//           the lexicographic minimum of (distance, index); stop as soon as the best distance is strictly below
//           the (safety-shrunk) distance to the nearest unexplored face -- nothing outside can tie or win.
//           Rows / cells whose lower-bound distance exceeds the best distance so far are skipped (exact: <= keeps ties).
//           Queries still unresolved after radius GRID_RMAX are finished by k_knn1_rest, one wave per query.
// Intra-cell order after the scatter depends on atomic arrival order, but min over (distance, index) does not.
// ---------------------------------------------------------------------------------------------
#define GRID_MAX_SMALL 128    // cells per axis: reference sets below GRID_BIG_N2 points
#define GRID_MAX_BIG 256      // ... and above: a map is a SURFACE, so its points crowd the occupied cells (~300 per cell of a 128^3 grid
#define GRID_BIG_N2 250000    //     at 5 M points); halving h quarters that.  Measured on the benchmark sequence (0.3 - 1.7 M points,
                              //     307 200 queries, profiles/r02_knn_grid_tuning.txt): 256^3 from 250 k points on + 3 shells in the
                              //     per-lane pass = 180 + 52 us per query set, against 220 + 198 us with 128^3 up to 1.5 M points and 2
                              //     shells; the finer grid costs 25 us more per index build
static inline int grid_max_for(int64_t n2) { return n2 >= GRID_BIG_N2 ? GRID_MAX_BIG : GRID_MAX_SMALL; }
static inline int64_t grid_cells_cap(int64_t n2) { const int64_t g = grid_max_for(n2); return g * g * g; }
#define GRID_HMIN 0.01f
#define GRID_RMAX 3      // shells of the one-query-per-lane pass; whatever it cannot bound goes to the wave-per-query pass
                         // (2: 130 + 158 us, 3: 180 + 52 us, 4: 282 + 45 us on the 256^3 grid)
#define SCAN_BLOCK 1024

struct GridInfo {
    unsigned int bb[6];        // order-preserving keys of min xyz, max xyz
    float origin[3], h, inv_h;
    int dims[3];
    unsigned int n_unresolved;
    unsigned int n_points;     // reference points the index was built over (validates warm-start indices)
    unsigned int n_cells;      // dims[0] * dims[1] * dims[2]: the cells the build addresses (the workspace holds the cubic budget)
    float eps;                 // absolute slack of every geometric bound: fp32 rounding of (v - origin) * inv_h at the cloud's
};                             // largest coordinate can move a point across a cell face by a few ulps of that coordinate

__device__ __forceinline__ unsigned int fkey(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned int k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// counters of the cells the box at hand addresses (+ the scan's end slot): a map's box is rarely a cube, so this is typically half of the budget
__global__ void k_grid_zero(const GridInfo* __restrict__ gi, unsigned int* __restrict__ counts) {
    const int64_t n = (int64_t)gi->n_cells + 1;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) *(uint4*)(counts + i) = make_uint4(0u, 0u, 0u, 0u);
        else for (int64_t j = i; j < n; ++j) counts[j] = 0u;
    }
}

// NOTE: device-scope atomics that hit ONE address serialise at the memory side at ~0.18 us each (3072 of them made
// this kernel 555 us): the workgroups publish their partial boxes instead and k_grid_setup folds them.
#define BBOX_BLOCKS 256          // threads of k_grid_setup
#define BBOX_MAX_PARTS 2048      // workgroups of k_grid_bbox = partial boxes k_grid_setup folds
// `nptr` (every build kernel): when not NULL the number of reference points is read from DEVICE memory (a resident map whose size the
// host never learns: e2e_knn1_index_build_dev); grids are then sized by the capacity and the loops stop at *nptr
// vec != 0 (p is 16-byte aligned): a lane reads FOUR points as three float4s of the flat coordinate stream -- 48 contiguous bytes, three loads in
// flight -- instead of twelve dwords at stride 12 (the scalar loop left this kernel latency-bound at 1.6 TB/s on an 11 M point map)
__global__ __launch_bounds__(KT) void k_grid_bbox(const float* __restrict__ p, int64_t n, const long long* __restrict__ nptr, unsigned int* __restrict__ part, int vec) {
    __shared__ unsigned int slo[3][KT / 64], shi[3][KT / 64];
    if (nptr) n = (int64_t)*nptr;
    unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t g = (int64_t)blockIdx.x * KT + threadIdx.x; g < n4; g += (int64_t)gridDim.x * KT) {
        const float4* q = (const float4*)(p + g * 12);
        const float4 a = q[0], b = q[1], d = q[2];
        const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const unsigned int k = fkey(v[j]);
            lo[j % 3] = min(lo[j % 3], k);
            hi[j % 3] = max(hi[j % 3], k);
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * KT + threadIdx.x; i < n; i += (int64_t)gridDim.x * KT)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned int k = fkey(p[i * 3 + c]);
            lo[c] = min(lo[c], k);
            hi[c] = max(hi[c], k);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = min(lo[c], (unsigned int)__shfl_down((int)lo[c], o, 64));
            hi[c] = max(hi[c], (unsigned int)__shfl_down((int)hi[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { slo[c][threadIdx.x >> 6] = lo[c]; shi[c][threadIdx.x >> 6] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {                       // one pair of integer atomics per workgroup and axis
        const int c = threadIdx.x;
        unsigned int l = slo[c][0], h = shi[c][0];
        for (int w = 1; w < KT / 64; ++w) { l = min(l, slo[c][w]); h = max(h, shi[c][w]); }
        part[blockIdx.x * 6 + c] = l;
        part[blockIdx.x * 6 + 3 + c] = h;
    }
}

__global__ __launch_bounds__(BBOX_BLOCKS) void k_grid_setup(GridInfo* gi, const unsigned int* __restrict__ part, int nparts, int gmax,
                                                            const long long* __restrict__ nptr, long long n_host, int gmax_force) {
    __shared__ unsigned int sh[6][BBOX_BLOCKS / 64];
    if (nptr) gmax = (*nptr >= GRID_BIG_N2) ? GRID_MAX_BIG : GRID_MAX_SMALL;      // the host's rule (grid_max_for) on the device-resident count
    if (gmax_force > 0) gmax = gmax_force;
    if (threadIdx.x == 0) { gi->n_points = (unsigned int)(nptr ? *nptr : n_host); gi->n_unresolved = 0; }
    for (int c = 0; c < 6; ++c) {
        unsigned int v = c < 3 ? 0xFFFFFFFFu : 0u;
        for (int j = threadIdx.x; j < nparts; j += BBOX_BLOCKS) v = (c < 3) ? min(v, part[j * 6 + c]) : max(v, part[j * 6 + c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned int w = (unsigned int)__shfl_down((int)v, o, 64);
            v = (c < 3) ? min(v, w) : max(v, w);
        }
        if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int c = 0; c < 6; ++c) {
        unsigned int v = sh[c][0];
        for (int w = 1; w < BBOX_BLOCKS / 64; ++w) v = (c < 3) ? min(v, sh[c][w]) : max(v, sh[c][w]);
        gi->bb[c] = v;
    }
    float lo[3], ext = 0.f;
    for (int c = 0; c < 3; ++c) {
        lo[c] = fkey_inv(gi->bb[c]);
        ext = fmaxf(ext, fkey_inv(gi->bb[3 + c]) - lo[c]);
    }
    // Cell size: cubic budget, h = longest extent / (gmax - 1).  A map is a thin surface in a box that is rarely a cube, so many of the
    // gmax^3 cells are never addressed; round 3 tried cells as small as the cell budget allows for the box at hand (h = cbrt(volume /
    // budget), up to 1024 cells along an axis).  Exact all the same, but SLOWER on the refinement path: the queries are back-projected
    // PREDICTED depths, typically several cells away from the map surface, and the shell search visits ~(1/h)^3 cells to cover that
    // distance -- 0.71 -> 1.12 ms per keyframe at 1.66 M points, 177.3 -> 172.0 steps/s over a whole pass
    // (profiles/r03_knn_adaptive_cells_rejected.txt).  Reverted.  Round 4 tried the milder rule "refine only once the box holds more than
    // ppc points per cell on average" (the map ends a pass of the benchmark sequence at 9.9 per cell): ppc 1.5 / 2.5 / 4 / 6 gave 194.6 /
    // 196.5 / 197.1 / 197.2 steps/s over a whole pass against 197.1 without it (profiles/r04_knn_density_cells_rejected.txt).  Not kept.
    const float h = fmaxf(ext / (float)(gmax - 1), GRID_HMIN);
    float maxabs = 0.f;
    for (int c = 0; c < 3; ++c) maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[c]), fabsf(fkey_inv(gi->bb[3 + c]))));
    gi->eps = 1e-5f + 2e-6f * maxabs;                              // ~16 ulps of the largest coordinate
    gi->h = h;
    gi->inv_h = 1.0f / h;
    for (int c = 0; c < 3; ++c) {
        gi->origin[c] = lo[c];
        int d = (int)floorf((fkey_inv(gi->bb[3 + c]) - lo[c]) / h) + 1;
        gi->dims[c] = min(max(d, 1), gmax);
    }
    gi->n_cells = (unsigned int)(gi->dims[0] * gi->dims[1] * gi->dims[2]);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int dim) {
    return min(max((int)floorf((v - o) * inv_h), 0), dim - 1);
}

// Lanes of a wave that fall into the same cell share ONE atomic: the map is stored in frame order, so the 64 consecutive points of a wave are
// neighbouring pixels of one frame and land in a handful of cells.  One round per distinct cell (ballot on the first ungrouped lane's cell): at
// worst 64 rounds of a dozen scalar / lane instructions, which is nothing next to a device-scope atomic -- on this part those execute at the
// memory side, not in the XCD's L2, and their rate bounds both kernels.  group_by_cell returns, per lane, the leader lane of its group, its rank
// in the group and the group's size.
__device__ __forceinline__ void group_by_cell(unsigned int c, bool act, int lane, int& leader_of, unsigned int& rank, unsigned int& gsize) {
    leader_of = lane; rank = 0u; gsize = 1u;
    unsigned long long rem = __ballot(act);
    const unsigned long long below = (1ull << lane) - 1ull;
    while (rem) {
        const int leader = (int)__builtin_ctzll(rem);
        const unsigned int cl = (unsigned int)__builtin_amdgcn_readlane((int)c, leader);
        const unsigned long long m = __ballot(act && c == cl) & rem;
        if ((m >> lane) & 1ull) {
            leader_of = leader;
            rank = (unsigned int)__builtin_popcountll(m & below);
            gsize = (unsigned int)__builtin_popcountll(m);
        }
        rem &= ~m;
    }
}

__global__ __launch_bounds__(KT) void k_grid_count(const float* __restrict__ p, int64_t n, const long long* __restrict__ nptr, const GridInfo* __restrict__ gi,
                                                   unsigned int* __restrict__ counts) {
    if (nptr) n = (int64_t)*nptr;
    const float ox = gi->origin[0], oy = gi->origin[1], oz = gi->origin[2], ih = gi->inv_h;
    const int dx = gi->dims[0], dy = gi->dims[1], dz = gi->dims[2];
    const int lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * KT + (threadIdx.x & ~63); b < n; b += (int64_t)gridDim.x * KT) {      // wave-uniform trip count
        const int64_t i = b + lane;
        const bool act = i < n;
        unsigned int c = 0xFFFFFFFFu;
        if (act) {
            const int cx = cell_coord(p[i * 3], ox, ih, dx), cy = cell_coord(p[i * 3 + 1], oy, ih, dy), cz = cell_coord(p[i * 3 + 2], oz, ih, dz);
            c = (unsigned int)((cz * dy + cy) * dx + cx);
        }
        int leader_of;
        unsigned int rank, gsize;
        group_by_cell(c, act, lane, leader_of, rank, gsize);
        if (act && leader_of == lane) atomicAdd(&counts[c], gsize);
    }
}

// three-kernel exclusive scan over the n_cells + 1 counters the box addresses (launched for the cubic budget; workgroups beyond leave at once):
// block sums -> scan of block sums -> local scan + offset
__global__ __launch_bounds__(KT) void k_scan_blocksum(const unsigned int* __restrict__ v, const GridInfo* __restrict__ gi, unsigned int* __restrict__ bsum) {
    __shared__ unsigned int sh[KT / 64];
    const int64_t n = (int64_t)gi->n_cells + 1;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    if ((int64_t)blockIdx.x * SCAN_BLOCK >= n) return;
    unsigned int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (base + j < n) ? v[base + j] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(1024) void k_scan_bsums(unsigned int* __restrict__ bsum, const GridInfo* __restrict__ gi) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry_s;
    const int nb = (int)(((int64_t)gi->n_cells + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned int v = (i < nb) ? bsum[i] : 0u;
        unsigned int x = v;
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
        if (i < nb) bsum[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
}
__global__ __launch_bounds__(KT) void k_scan_apply(const unsigned int* __restrict__ v, const GridInfo* __restrict__ gi, const unsigned int* __restrict__ bsum,
                                                   unsigned int* __restrict__ out) {
    __shared__ unsigned int sh[KT / 64];
    const int64_t n = (int64_t)gi->n_cells + 1;
    if ((int64_t)blockIdx.x * SCAN_BLOCK >= n) return;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    unsigned int a[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = (base + j < n) ? v[base + j] : 0u; s += a[j]; }
    unsigned int x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int y = __shfl_up(x, o, 64);
        if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += sh[w];
    unsigned int run = bsum[blockIdx.x] + woff + x - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) out[base + j] = run;
        run += a[j];
    }
}

// counting-sort scatter.  The per-cell counters double as cursors (they are not needed once scanned): a group of `gsize` same-cell lanes takes the
// slots [old - gsize, old) of its cell with one atomicSub, so the lanes of a group write CONSECUTIVE float4s.  The cell is recomputed from the point
// (the same expression on the same inputs as k_grid_count) instead of being carried through memory.
__global__ __launch_bounds__(KT) void k_grid_scatter(const float* __restrict__ p, int64_t n, const long long* __restrict__ nptr, const GridInfo* __restrict__ gi,
                                                     const unsigned int* __restrict__ starts, unsigned int* __restrict__ counts,
                                                     float4* __restrict__ sorted) {
    if (nptr) n = (int64_t)*nptr;
    const float ox = gi->origin[0], oy = gi->origin[1], oz = gi->origin[2], ih = gi->inv_h;
    const int dx = gi->dims[0], dy = gi->dims[1], dz = gi->dims[2];
    const int lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * KT + (threadIdx.x & ~63); b < n; b += (int64_t)gridDim.x * KT) {
        const int64_t i = b + lane;
        const bool act = i < n;
        unsigned int c = 0xFFFFFFFFu;
        float x = 0.f, y = 0.f, z = 0.f;
        if (act) {
            x = p[i * 3]; y = p[i * 3 + 1]; z = p[i * 3 + 2];
            const int cx = cell_coord(x, ox, ih, dx), cy = cell_coord(y, oy, ih, dy), cz = cell_coord(z, oz, ih, dz);
            c = (unsigned int)((cz * dy + cy) * dx + cx);
        }
        int leader_of;
        unsigned int rank, gsize;
        group_by_cell(c, act, lane, leader_of, rank, gsize);
        unsigned int old = 0u;
        if (act && leader_of == lane) old = atomicSub(&counts[c], gsize);
        old = (unsigned int)__shfl((int)old, leader_of, 64);
        if (act) sorted[starts[c] + old - gsize + rank] = make_float4(x, y, z, __uint_as_float((unsigned int)i));
    }
}

// row_len > 0: the queries are the pixels of an image of that width in row-major order (n1 a whole number of rows; width and height
// multiples of 8, checked by the host): the 64 lanes of a wave then take an 8 x 8 pixel TILE instead of 64 consecutive pixels of a row.
// Back-projected pixels of a tile lie within a cell or two of each other, so the lanes of a wave walk the same cell rows -- the same
// `starts` words and the same ranges of `sorted` -- instead of a strip of the scene ~64 pixels wide.  Results are identical.
// warm != NULL (may alias idx): warm[i] is the answer of an EARLIER query for a point near p1[i] against the SAME reference set (the
// refinement steps of one keyframe move a pixel's point by a fraction of a cell).  Its distance to the new query is a valid upper bound
// -- it is a real reference point -- so the search starts with the ball already shrunk to it: cells and x-ranges beyond it are never
// read.  The result is still the exact nearest neighbour with the lowest index among ties (the candidate only enters the same comparison
// as every scanned point).  ref = the unsorted reference points the indices refer to.
__global__ __launch_bounds__(KT) void k_grid_query(const float* __restrict__ p1, int64_t n1, int row_len, GridInfo* __restrict__ gi,
                                                   const unsigned int* __restrict__ starts, const float4* __restrict__ sorted,
                                                   float* __restrict__ dists, long long* idx,
                                                   unsigned int* __restrict__ unresolved, const float* __restrict__ ref, const long long* warm) {
    int64_t i0 = (int64_t)blockIdx.x * KT + threadIdx.x;
    if (row_len > 0) {
        const int64_t tile = i0 >> 6;
        const int within = (int)(i0 & 63), tiles_x = row_len >> 3;
        const int64_t ty = tile / tiles_x;
        const int tx = (int)(tile - ty * tiles_x);
        i0 = (ty * 8 + (within >> 3)) * (int64_t)row_len + tx * 8 + (within & 7);
    }
    const bool live = i0 < n1;
    const int64_t i = live ? i0 : n1 - 1;                      // idle lanes shadow the last query and write nothing
    const float x = p1[i * 3], y = p1[i * 3 + 1], z = p1[i * 3 + 2];
    const float q[3] = {x, y, z};
    const float h = gi->h, ih = gi->inv_h;
    const float eps = gi->eps + 2e-6f * fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));      // + the query's own rounding scale
    const int dims[3] = {gi->dims[0], gi->dims[1], gi->dims[2]};
    const float org[3] = {gi->origin[0], gi->origin[1], gi->origin[2]};
    int cq[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cq[c] = cell_coord(q[c], org[c], ih, dims[c]);
    float bd = 3.402823466e38f;
    unsigned int bi = 0xFFFFFFFFu;
    if (warm && live) {
        const long long w = warm[i];
        if (w >= 0 && w < (long long)gi->n_points) {
            const float dx = x - ref[w * 3], dy = y - ref[w * 3 + 1], dz = z - ref[w * 3 + 2];
            bd = (dx * dx + dy * dy) + dz * dz;              // the same expression (and contraction setting) as for every scanned point
            bi = (unsigned int)w;
        }
    }
    bool done = !live;
    int pl[3] = {0, 0, 0}, ph[3] = {-1, -1, -1};                  // previously explored cube (empty)
    for (int r = 0; r <= GRID_RMAX && !done; ++r) {
        int lo[3], hi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = max(cq[c] - r, 0); hi[c] = min(cq[c] + r, dims[c] - 1); }
        for (int cz = lo[2]; cz <= hi[2]; ++cz) {
            // Pruning (exact): a cell row / cell can only matter if its lower-bound distance to the query does not exceed
            // the best distance so far (<=, a tie with a smaller index may hide there).  Bounds are shrunk by the same
            // safety margin as the stop test because points sit in their cells only up to fp32 rounding.
            const float gz = fmaxf(fmaxf(org[2] + (float)cz * h - z, z - (org[2] + (float)(cz + 1) * h)), 0.f);
            const float gzs = fmaxf(gz * 0.999f - eps, 0.f);
            if (gzs * gzs > bd) continue;
            for (int cy = lo[1]; cy <= hi[1]; ++cy) {
                const float gy = fmaxf(fmaxf(org[1] + (float)cy * h - y, y - (org[1] + (float)(cy + 1) * h)), 0.f);
                const float gys = fmaxf(gy * 0.999f - eps, 0.f);
                const float dyz = gzs * gzs + gys * gys;
                if (dyz > bd) continue;
                // cells of one (cz,cy) row are consecutive => their points form ONE contiguous range of `sorted`.
                // A row that was already inside the previous cube only contributes its two new end segments.
                const bool row_in_prev = cz >= pl[2] && cz <= ph[2] && cy >= pl[1] && cy <= ph[1];
                const unsigned int rowbase = (unsigned int)((cz * dims[1] + cy) * dims[0]);
                int seg_lo[2] = {lo[0], 0}, seg_hi[2] = {hi[0], -1};
                if (row_in_prev) { seg_hi[0] = pl[0] - 1; seg_lo[1] = ph[0] + 1; seg_hi[1] = hi[0]; }
                int xlo = lo[0], xhi = hi[0];
                if (bd < 3.0e38f) {                                 // x extent of the ball of radius sqrt(bd) at this row, inflated
                    const float rx = sqrtf(bd - dyz) * 1.001f + 2.f * eps;
                    xlo = max(xlo, (int)floorf(fmaxf((x - rx - org[0]) * ih, -1.f)));
                    xhi = min(xhi, (int)floorf(fminf((x + rx - org[0]) * ih, 1.0e6f)));
                }
#pragma unroll
                for (int sgm = 0; sgm < 2; ++sgm) {
                    const int s0 = max(seg_lo[sgm], xlo), s1 = min(seg_hi[sgm], xhi);
                    if (s0 > s1) continue;
                    unsigned int k = starts[rowbase + s0];
                    const unsigned int e = starts[rowbase + s1 + 1];
                    for (; k + 4 <= e; k += 4) {
                        float4 t[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) t[u] = sorted[k + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float dx = x - t[u].x, dy = y - t[u].y, dz = z - t[u].z;
                            const float d = (dx * dx + dy * dy) + dz * dz;
                            const unsigned int id = __float_as_uint(t[u].w);
                            if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
                        }
                    }
                    for (; k < e; ++k) {
                        const float4 t = sorted[k];
                        const float dx = x - t.x, dy = y - t.y, dz = z - t.z;
                        const float d = (dx * dx + dy * dy) + dz * dz;
                        const unsigned int id = __float_as_uint(t.w);
                        if (d < bd || (d == bd && id < bi)) { bd = d; bi = id; }
                    }
                }
            }
        }
        // lower bound on the distance to any point outside the explored cube
        float db = 3.402823466e38f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (lo[c] > 0) db = fminf(db, q[c] - (org[c] + (float)lo[c] * h));
            if (hi[c] < dims[c] - 1) db = fminf(db, (org[c] + (float)(hi[c] + 1) * h) - q[c]);
        }
        if (db >= 3.0e38f) done = true;                            // the whole grid has been visited
        else {
            const float dbe = db * 0.999f - eps;                 // points sit in their cell up to fp32 rounding
            if (dbe > 0.f && bd < dbe * dbe) done = true;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { pl[c] = lo[c]; ph[c] = hi[c]; }
    }
    if (live) {                                                 // unresolved queries leave their best-so-far (a real point or nothing):
        dists[i] = bd;                                          // k_knn1_rest starts from it instead of from scratch
        idx[i] = (bi == 0xFFFFFFFFu) ? -1ll : (long long)bi;
    }
    // unresolved queries: ONE atomic per wave (ballot + prefix popcount), not one per lane
    const unsigned long long m = __ballot(!done);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned int base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&gi->n_unresolved, (unsigned int)__builtin_popcountll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (!done) unresolved[base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned int)i;
    }
}

// Queries the per-lane pass could not bound within GRID_RMAX shells (they look at a part of the scene the map does not
// cover yet) are finished by ONE WAVE each: same shell search, (distance, index) keys combined with integer min.  While
// nothing has been found the radius doubles.  Worst case every point of the grid is visited once -- the brute force this
// replaces read all n2 points per query.
// Work distribution inside the wave: the (cz,cy) rows of a shell go to the lanes 64 at a time, each lane works out the
// range(s) of `sorted` its row contributes (two dependent loads of `starts`), and then the POINTS of all 64 ranges -- not the
// rows -- are dealt to the lanes: an inclusive scan of the range lengths over the wave, and lane l takes the points l, l + 64, ...
// of the concatenation (a 6-step binary search over the scan, by lane permutes, finds the range a point belongs to).  Rows
// crossing the map surface hold hundreds of points next to rows holding none; a lane per row left one lane scanning such a row
// point by point with the others idle, which made this pass grow from 38 to 423 us over a pass of the benchmark sequence
// (0.7 -> 11.7 M map points).  Consecutive lanes now read consecutive float4s, and 64 points are in flight per step.
// What is left is latency: a far query (a pixel looking at a part of the scene the map does not cover: nearest point a metre = 30 cells
// away) walks thousands of rows, two dependent loads each, so a lane takes REST_ROWS rows per batch with all their `starts` words in flight.
#define REST_ROWS 4
__global__ __launch_bounds__(KT) void k_knn1_rest(const float* __restrict__ p1, const GridInfo* __restrict__ gi,
                                                  const unsigned int* __restrict__ starts, const float4* __restrict__ sorted,
                                                  const unsigned int* __restrict__ unresolved, float* dists, long long* idx) {
    const unsigned int cnt = gi->n_unresolved;
    const int lane = threadIdx.x & 63;
    const unsigned int wave0 = blockIdx.x * (KT / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (KT / 64);
    const float h = gi->h, ih = gi->inv_h;
    const int dims[3] = {gi->dims[0], gi->dims[1], gi->dims[2]};
    const float org[3] = {gi->origin[0], gi->origin[1], gi->origin[2]};
    for (unsigned int u = wave0; u < cnt; u += nwaves) {
        const unsigned int i = unresolved[u];
        const float x = p1[(int64_t)i * 3], y = p1[(int64_t)i * 3 + 1], z = p1[(int64_t)i * 3 + 2];
        const float q[3] = {x, y, z};
        const float eps = gi->eps + 2e-6f * fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int cq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) cq[c] = cell_coord(q[c], org[c], ih, dims[c]);
        unsigned long long best = 0xFFFFFFFFFFFFFFFFull;
        {                                                       // what the per-lane pass (or a warm start) had found: an upper bound from a real point
            const long long i0 = idx[i];
            if (i0 >= 0) best = ((unsigned long long)__float_as_uint(dists[i]) << 32) | (unsigned long long)(unsigned int)i0;
        }
        int pl[3] = {0, 0, 0}, ph[3] = {-1, -1, -1};
        int r = 0;
        while (true) {
            int lo[3], hi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = max(cq[c] - r, 0); hi[c] = min(cq[c] + r, dims[c] - 1); }
            const int ny = hi[1] - lo[1] + 1, nrows = ny * (hi[2] - lo[2] + 1);
            for (int t0 = 0; t0 < nrows; t0 += 64 * REST_ROWS) {             // wave-uniform trip count: all lanes take part in the permutes below
                // the bound, refreshed per batch of rows: the smallest distance any lane holds (bit patterns of non-negative floats order like
                // unsigned integers; 0xFFFFFFFF = nothing yet).  A far query then stops reading rows outside the ball of the first points it meets
                // instead of scanning the whole doubled cube.
                unsigned int bdu = (unsigned int)(best >> 32);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) bdu = min(bdu, (unsigned int)__shfl_xor((int)bdu, o, 64));
                const bool bounded = bdu != 0xFFFFFFFFu;
                const float bd0 = __uint_as_float(bdu);
                // REST_ROWS rows per lane and batch, their `starts` words fetched together (a pruned row reads word 0 twice: length 0)
                unsigned int ia[REST_ROWS][4];
#pragma unroll
                for (int j = 0; j < REST_ROWS; ++j) {
                    const int t = t0 + j * 64 + lane;
                    ia[j][0] = ia[j][1] = ia[j][2] = ia[j][3] = 0u;
                    if (t < nrows) {
                        const int cz = lo[2] + t / ny, cy = lo[1] + t % ny;
                        const bool row_in_prev = cz >= pl[2] && cz <= ph[2] && cy >= pl[1] && cy <= ph[1];
                        const unsigned int rowbase = (unsigned int)((cz * dims[1] + cy) * dims[0]);
                        int seg_lo[2] = {lo[0], 0}, seg_hi[2] = {hi[0], -1};
                        if (row_in_prev) { seg_hi[0] = pl[0] - 1; seg_lo[1] = ph[0] + 1; seg_hi[1] = hi[0]; }
                        int xlo = lo[0], xhi = hi[0];
                        bool keep = true;
                        if (bounded) {                         // same exact ball pruning as the per-lane pass
                            const float gz = fmaxf(fmaxf(org[2] + (float)cz * h - z, z - (org[2] + (float)(cz + 1) * h)), 0.f);
                            const float gy = fmaxf(fmaxf(org[1] + (float)cy * h - y, y - (org[1] + (float)(cy + 1) * h)), 0.f);
                            const float gzs = fmaxf(gz * 0.999f - eps, 0.f), gys = fmaxf(gy * 0.999f - eps, 0.f);
                            const float dyz = gzs * gzs + gys * gys;
                            keep = !(dyz > bd0);
                            const float rx = sqrtf(fmaxf(bd0 - dyz, 0.f)) * 1.001f + 2.f * eps;
                            xlo = max(xlo, (int)floorf(fmaxf((x - rx - org[0]) * ih, -1.f)));
                            xhi = min(xhi, (int)floorf(fminf((x + rx - org[0]) * ih, 1.0e6f)));
                        }
                        if (keep) {
                            const int a0 = max(seg_lo[0], xlo), a1 = min(seg_hi[0], xhi), b0 = max(seg_lo[1], xlo), b1 = min(seg_hi[1], xhi);
                            if (a0 <= a1) { ia[j][0] = rowbase + a0; ia[j][1] = rowbase + a1 + 1; }
                            if (b0 <= b1) { ia[j][2] = rowbase + b0; ia[j][3] = rowbase + b1 + 1; }
                        }
                    }
                }
                unsigned int sv[REST_ROWS][4];
#pragma unroll
                for (int j = 0; j < REST_ROWS; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[j][e] = starts[ia[j][e]];
#pragma unroll
                for (int j = 0; j < REST_ROWS; ++j) {
                    if (t0 + j * 64 >= nrows) break;                          // wave-uniform
                    const unsigned int ka = sv[j][0], la = sv[j][1] - sv[j][0], kb = sv[j][2], lb = sv[j][3] - sv[j][2];
                    const unsigned int len = la + lb;
                    unsigned int incl = len;                                   // inclusive scan of the lengths over the wave
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned int v = (unsigned int)__shfl_up((int)incl, o, 64);
                        if (lane >= o) incl += v;
                    }
                    const unsigned int total = (unsigned int)__shfl((int)incl, 63, 64);
                    for (unsigned int g0 = 0; g0 < total; g0 += 64) {
                        const unsigned int g = g0 + lane;
                        const bool live = g < total;
                        const unsigned int gg = live ? g : total - 1u;
                        int lo_l = 0, hi_l = 63;                               // first lane whose inclusive sum exceeds gg (exists: gg < total)
#pragma unroll
                        for (int sidx = 0; sidx < 6; ++sidx) {
                            const int mid = (lo_l + hi_l) >> 1;
                            const unsigned int v = (unsigned int)__shfl((int)incl, mid, 64);
                            if (v <= gg) lo_l = mid + 1; else hi_l = mid;
                        }
                        const unsigned int r_incl = (unsigned int)__shfl((int)incl, lo_l, 64), r_len = (unsigned int)__shfl((int)len, lo_l, 64);
                        const unsigned int r_ka = (unsigned int)__shfl((int)ka, lo_l, 64), r_la = (unsigned int)__shfl((int)la, lo_l, 64);
                        const unsigned int r_kb = (unsigned int)__shfl((int)kb, lo_l, 64);
                        const unsigned int off = gg - (r_incl - r_len);
                        const unsigned int k = off < r_la ? r_ka + off : r_kb + (off - r_la);
                        if (live) {
                            const float4 tq = sorted[k];
                            const float dx = x - tq.x, dy = y - tq.y, dz = z - tq.z;
                            const float d = (dx * dx + dy * dy) + dz * dz;
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(tq.w);
                            best = (key < best) ? key : best;
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {                 // all lanes end up with the wave minimum
                const unsigned long long other = __shfl_xor(best, o, 64);
                best = (other < best) ? other : best;
            }
            float db = 3.402823466e38f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (lo[c] > 0) db = fminf(db, q[c] - (org[c] + (float)lo[c] * h));
                if (hi[c] < dims[c] - 1) db = fminf(db, (org[c] + (float)(hi[c] + 1) * h) - q[c]);
            }
            if (db >= 3.0e38f) break;                          // the whole grid has been visited
            const float bd = __uint_as_float((unsigned int)(best >> 32));
            const float dbe = db * 0.999f - eps;
            if (best != 0xFFFFFFFFFFFFFFFFull && dbe > 0.f && bd < dbe * dbe) break;
#pragma unroll
            for (int c = 0; c < 3; ++c) { pl[c] = lo[c]; ph[c] = hi[c]; }
            // nothing found yet: double the radius.  Something found at distance sqrt(bd): no closer point can lie outside
            // the cube of that half-width, so jump straight to the radius that bounds it.
            if (best == 0xFFFFFFFFFFFFFFFFull) r = max(r + 1, 2 * r);
            else r = max(r + 1, (int)fminf(ceilf(sqrtf(bd) * ih) + 1.f, 1.0e6f));
        }
        if (lane == 0) {
            dists[i] = __uint_as_float((unsigned int)(best >> 32));
            idx[i] = (long long)(best & 0xFFFFFFFFull);
        }
    }
}

// Small query sets (frame-to-model odometry: 19 200 queries): EVERY query goes to the wave-per-query pass.  One query per lane means 300
// waves on 1024 SIMDs, each lane walking its <= 49 cell rows through two dependent loads per row -- 70 us of pure latency; a wave per query
// spreads the rows of a shell over its lanes and finishes in a few dependent steps.  This kernel only seeds the search: the warm candidate's
// distance (or "nothing") and the list of all queries.
__global__ __launch_bounds__(256) void k_knn1_all_unresolved(const float* __restrict__ p1, int64_t n1, GridInfo* __restrict__ gi, unsigned int* __restrict__ unresolved,
                                                             const float* __restrict__ ref, const long long* __restrict__ warm, float* __restrict__ dists,
                                                             long long* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) gi->n_unresolved = (unsigned int)n1;
    if (i >= n1) return;
    long long w = -1;
    float d = 3.402823466e38f;
    if (warm) {
        const long long c = warm[i];
        if (c >= 0 && c < (long long)gi->n_points) {
            const float dx = p1[i * 3] - ref[c * 3], dy = p1[i * 3 + 1] - ref[c * 3 + 1], dz = p1[i * 3 + 2] - ref[c * 3 + 2];
            d = (dx * dx + dy * dy) + dz * dz;                  // the same expression as for every scanned point
            w = c;
        }
    }
    dists[i] = d;
    idx[i] = w;
    unresolved[i] = (unsigned int)i;
}

extern "C" {

static int knn1_brute(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists, long long* idx, void* workspace, hipStream_t st);
#define KNN_GRID_MIN_N2 8192      // below this the brute force is already cheap

static int64_t grid_ws_bytes(int64_t n1, int64_t n2, bool big_always = false, int cells = 0) {
    const int64_t nc = cells > 0 ? (int64_t)cells * cells * cells : (big_always ? (int64_t)GRID_MAX_BIG * GRID_MAX_BIG * GRID_MAX_BIG : grid_cells_cap(n2)),
                  nb = (nc + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
    // GridInfo | counts[nc+1 .. 16 B] | starts[nc+1 .. 16 B] | bsum[nb] | bbpart | unresolved[n1] | sorted float4[n2]
    return 256 + 4 * (nc + 4) * 2 + 4 * (nb + 1) + 4 * 6 * BBOX_MAX_PARTS + 4 * n1 + 64 + 16 * n2;
}

int64_t e2e_knn1_workspace_bytes(int64_t n1, int64_t n2) {
    if (n1 <= 0 || n2 <= 0) return 0;
    // sized for EITHER algorithm (the caller may force the grid on a small set)
    const int64_t brute = 8 * n1, grid = grid_ws_bytes(n1, n2);
    return brute > grid ? brute : grid;
}

struct GridWs {
    GridInfo* gi;
    unsigned int *counts, *starts, *bsum, *bbpart, *unresolved;
    float4* sorted;
    int64_t nc;
    int nb;
};

// carve the workspace: `nq` = number of query slots reserved for the unresolved list
// cells > 0: a caller-chosen resolution (cells per axis, the *_res entry points) instead of the map-tuned rule
static GridWs grid_ws(void* workspace, int64_t nq, int64_t n2, bool big_always = false, int cells = 0) {
    GridWs g;
    g.nc = cells > 0 ? (int64_t)cells * cells * cells : (big_always ? (int64_t)GRID_MAX_BIG * GRID_MAX_BIG * GRID_MAX_BIG : grid_cells_cap(n2));
    g.nb = (int)((g.nc + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK);
    char* w = (char*)workspace;
    g.gi = (GridInfo*)w; w += 256;
    g.counts = (unsigned int*)w; w += 4 * ((g.nc + 4) & ~(int64_t)3);          // 16-byte aligned (k_grid_zero stores uint4s)
    g.starts = (unsigned int*)w; w += 4 * ((g.nc + 4) & ~(int64_t)3);
    g.bsum = (unsigned int*)w; w += 4 * (g.nb + 1);
    g.bbpart = (unsigned int*)w; w += 4 * 6 * BBOX_MAX_PARTS;
    g.unresolved = (unsigned int*)w; w += 4 * nq;
    w = (char*)(((uintptr_t)w + 63) & ~(uintptr_t)63);
    g.sorted = (float4*)w;
    return g;
}

__global__ void k_grid_reset_unresolved(GridInfo* gi) { gi->n_unresolved = 0; }

// n2: number of reference points, or -- with n2_dev != NULL -- their CAPACITY (the live count is read on the device)
static void grid_build(const float* p2, int64_t n2, const long long* n2_dev, const GridWs& g, hipStream_t st, int cells = 0) {
    const int gp = (int)((n2 + KT - 1) / KT > 4096 ? 4096 : (n2 + KT - 1) / KT);
    const int bb_blocks = gp > BBOX_MAX_PARTS ? BBOX_MAX_PARTS : gp;
    hipLaunchKernelGGL(k_grid_bbox, dim3(bb_blocks), dim3(KT), 0, st, p2, n2, n2_dev, g.bbpart, (int)(((uintptr_t)p2 & 15) == 0));
    hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(BBOX_BLOCKS), 0, st, g.gi, g.bbpart, bb_blocks, grid_max_for(n2), n2_dev, (long long)n2, cells);
    hipLaunchKernelGGL(k_grid_zero, dim3(2048), dim3(256), 0, st, g.gi, g.counts);
    hipLaunchKernelGGL(k_grid_count, dim3(gp), dim3(KT), 0, st, p2, n2, n2_dev, g.gi, g.counts);
    hipLaunchKernelGGL(k_scan_blocksum, dim3(g.nb), dim3(KT), 0, st, g.counts, g.gi, g.bsum);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, st, g.bsum, g.gi);
    hipLaunchKernelGGL(k_scan_apply, dim3(g.nb), dim3(KT), 0, st, g.counts, g.gi, g.bsum, g.starts);
    hipLaunchKernelGGL(k_grid_scatter, dim3(gp), dim3(KT), 0, st, p2, n2, n2_dev, g.gi, g.starts, g.counts, g.sorted);
}

static void grid_query(const float* p1, int64_t n1, const GridWs& g, float* dists, long long* idx, hipStream_t st, int row_len = 0,
                       const float* ref = nullptr, const long long* warm = nullptr) {
    if (row_len <= 0 || (row_len & 7) || n1 % ((int64_t)row_len * 8)) row_len = 0;      // tile order needs whole 8 x 8 tiles
    if (!ref) warm = nullptr;
    hipLaunchKernelGGL(k_grid_query, dim3((unsigned)((n1 + KT - 1) / KT)), dim3(KT), 0, st, p1, n1, row_len, g.gi, g.starts, g.sorted, dists, idx, g.unresolved, ref,
                       warm);
    // one WAVE per unresolved query: 1024 workgroups (4096 waves) suit the refinement queries, of which a few per cent stay unresolved; a small
    // query set that is mostly unresolved (frame-to-model odometry while the pose is still wrong: 19 200 queries, decimetres off the targets)
    // gets a wave per query instead of 2-5 queries per wave in sequence -- idle workgroups leave at once
    const int64_t rest_blocks = n1 <= 65536 ? (n1 + 3) / 4 : 2048;
    hipLaunchKernelGGL(k_knn1_rest, dim3((unsigned)(rest_blocks < 1024 ? 1024 : (rest_blocks > 8192 ? 8192 : rest_blocks))), dim3(KT), 0, st, p1, g.gi, g.starts,
                       g.sorted, g.unresolved, dists, idx);
}

static int knn1_grid(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists, long long* idx, void* workspace, hipStream_t st) {
    const GridWs g = grid_ws(workspace, n1, n2);
    grid_build(p2, n2, nullptr, g, st);
    grid_query(p1, n1, g, dists, idx, st);
    return E2E_OK;
}

/* Persistent index: build once for a reference set, query many times (ICP iterations, the refinement steps of one
 * keyframe).  The index occupies e2e_knn1_workspace_bytes(max_queries, n2) bytes owned by the caller. */
int e2e_knn1_index_build(const float* p2, int64_t n2, int64_t max_queries, void* index, void* stream) {
    E2E_REQUIRE(p2 && index && n2 > 0 && n2 < 0xFFFFFFFFll && max_queries > 0 && max_queries < 0xFFFFFFFFll, E2E_ERR_ARG, "e2e_knn1_index_build: bad argument");
    grid_build(p2, n2, nullptr, grid_ws(index, max_queries, n2), (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_knn1_index_build");
    return E2E_OK;
}

/* The same index over a RESIDENT reference set whose live size lives in device memory (*n2_dev <= n2_capacity, > 0): the host never
 * reads it, every launch argument is constant from one build to the next (capturable into a hipGraph), grids are sized by the
 * capacity.  The index occupies e2e_knn1_index_capacity_bytes(max_queries, n2_capacity) bytes; query it with
 * e2e_knn1_index_query_dev.  Results are those of e2e_knn1_index_build / _query on the first *n2_dev points. */
int64_t e2e_knn1_index_capacity_bytes(int64_t max_queries, int64_t n2_capacity) {
    if (max_queries <= 0 || n2_capacity <= 0) return 0;
    return grid_ws_bytes(max_queries, n2_capacity, true);
}

int e2e_knn1_index_build_dev(const float* p2, const long long* n2_dev, int64_t n2_capacity, int64_t max_queries, void* index, void* stream) {
    E2E_REQUIRE(p2 && n2_dev && index && n2_capacity > 0 && n2_capacity < 0xFFFFFFFFll && max_queries > 0 && max_queries < 0xFFFFFFFFll, E2E_ERR_ARG,
                "e2e_knn1_index_build_dev: bad argument");
    grid_build(p2, n2_capacity, n2_dev, grid_ws(index, max_queries, n2_capacity, true), (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_knn1_index_build_dev");
    return E2E_OK;
}

int e2e_knn1_index_query_dev(const float* p1, int64_t n1, int64_t n2_capacity, int64_t max_queries, void* index, float* dists, long long* idx, void* stream) {
    E2E_REQUIRE(p1 && index && dists && idx && n1 > 0 && n1 <= max_queries && n2_capacity > 0, E2E_ERR_ARG,
                "e2e_knn1_index_query_dev: bad argument (n1=%lld must not exceed the max_queries=%lld the index was built for)", (long long)n1, (long long)max_queries);
    const GridWs g = grid_ws(index, max_queries, n2_capacity, true);
    hipLaunchKernelGGL(k_grid_reset_unresolved, dim3(1), dim3(1), 0, (hipStream_t)stream, g.gi);
    grid_query(p1, n1, g, dists, idx, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_knn1_index_query_dev");
    return E2E_OK;
}

int e2e_knn1_index_query_dev_image(const float* p1, int64_t n1, int row_len, int64_t n2_capacity, int64_t max_queries, void* index, float* dists, long long* idx,
                                   void* stream) {
    E2E_REQUIRE(p1 && index && dists && idx && n1 > 0 && n1 <= max_queries && n2_capacity > 0 && row_len >= 0, E2E_ERR_ARG,
                "e2e_knn1_index_query_dev_image: bad argument (n1=%lld must not exceed the max_queries=%lld the index was built for)", (long long)n1, (long long)max_queries);
    const GridWs g = grid_ws(index, max_queries, n2_capacity, true);
    hipLaunchKernelGGL(k_grid_reset_unresolved, dim3(1), dim3(1), 0, (hipStream_t)stream, g.gi);
    grid_query(p1, n1, g, dists, idx, (hipStream_t)stream, row_len);
    E2E_LAUNCH_CHECK("e2e_knn1_index_query_dev_image");
    return E2E_OK;
}

int e2e_knn1_index_query_dev_image_warm(const float* p1, int64_t n1, int row_len, const float* ref_points, const long long* warm_idx, int64_t n2_capacity,
                                        int64_t max_queries, void* index, float* dists, long long* idx, void* stream) {
    E2E_REQUIRE(p1 && index && dists && idx && ref_points && n1 > 0 && n1 <= max_queries && n2_capacity > 0 && row_len >= 0, E2E_ERR_ARG,
                "e2e_knn1_index_query_dev_image_warm: bad argument (n1=%lld must not exceed the max_queries=%lld the index was built for)", (long long)n1,
                (long long)max_queries);
    const GridWs g = grid_ws(index, max_queries, n2_capacity, true);
    hipLaunchKernelGGL(k_grid_reset_unresolved, dim3(1), dim3(1), 0, (hipStream_t)stream, g.gi);
    grid_query(p1, n1, g, dists, idx, (hipStream_t)stream, row_len, ref_points, warm_idx);
    E2E_LAUNCH_CHECK("e2e_knn1_index_query_dev_image_warm");
    return E2E_OK;
}

/* The resident index at a caller-chosen resolution (cells per axis, 4 .. 256): the rule above is tuned for 307 200 queries against a
 * map of 0.25 - 12 M points, a few centimetres off its surface.  Frame-to-model odometry asks 19 200 queries of a sparse target set
 * (every 4th active point), possibly decimetres away while the pose is still wrong: a lane then walks thousands of empty fine cells
 * with dependent loads and there are too few waves to hide them; a coarse grid (32 per axis) bounds the walk to a few cells of ~100
 * points each.  Results are identical at any resolution.  ref_points / warm_idx: both NULL (cold) or both set (warm start). */
int64_t e2e_knn1_index_capacity_bytes_res(int64_t max_queries, int64_t n2_capacity, int cells_per_axis) {
    if (max_queries <= 0 || n2_capacity <= 0 || cells_per_axis < 4 || cells_per_axis > GRID_MAX_BIG) return 0;
    return grid_ws_bytes(max_queries, n2_capacity, true, cells_per_axis);
}

int e2e_knn1_index_build_dev_res(const float* p2, const long long* n2_dev, int64_t n2_capacity, int64_t max_queries, void* index, int cells_per_axis,
                                 void* stream) {
    E2E_REQUIRE(p2 && n2_dev && index && n2_capacity > 0 && n2_capacity < 0xFFFFFFFFll && max_queries > 0 && max_queries < 0xFFFFFFFFll &&
                cells_per_axis >= 4 && cells_per_axis <= GRID_MAX_BIG, E2E_ERR_ARG, "e2e_knn1_index_build_dev_res: bad argument");
    grid_build(p2, n2_capacity, n2_dev, grid_ws(index, max_queries, n2_capacity, true, cells_per_axis), (hipStream_t)stream, cells_per_axis);
    E2E_LAUNCH_CHECK("e2e_knn1_index_build_dev_res");
    return E2E_OK;
}

int e2e_knn1_index_query_dev_res(const float* p1, int64_t n1, const float* ref_points, const long long* warm_idx, int64_t n2_capacity,
                                 int64_t max_queries, void* index, int cells_per_axis, float* dists, long long* idx, void* stream) {
    E2E_REQUIRE(p1 && index && dists && idx && n1 > 0 && n1 <= max_queries && n2_capacity > 0 && cells_per_axis >= 4 && cells_per_axis <= GRID_MAX_BIG &&
                ((ref_points == nullptr) == (warm_idx == nullptr)), E2E_ERR_ARG, "e2e_knn1_index_query_dev_res: bad argument");
    const GridWs g = grid_ws(index, max_queries, n2_capacity, true, cells_per_axis);
    hipStream_t st = (hipStream_t)stream;
    if (n1 <= 32768) {                                          // a wave per query from the start (see k_knn1_all_unresolved)
        hipLaunchKernelGGL(k_knn1_all_unresolved, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, p1, n1, g.gi, g.unresolved, ref_points, warm_idx, dists, idx);
        hipLaunchKernelGGL(k_knn1_rest, dim3((unsigned)((n1 + 3) / 4)), dim3(KT), 0, st, p1, g.gi, g.starts, g.sorted, g.unresolved, dists, idx);
    } else {
        hipLaunchKernelGGL(k_grid_reset_unresolved, dim3(1), dim3(1), 0, st, g.gi);
        grid_query(p1, n1, g, dists, idx, st, 0, ref_points, warm_idx);
    }
    E2E_LAUNCH_CHECK("e2e_knn1_index_query_dev_res");
    return E2E_OK;
}

int e2e_knn1_index_query(const float* p1, int64_t n1, int64_t n2, int64_t max_queries, void* index, float* dists, long long* idx, void* stream) {
    E2E_REQUIRE(p1 && index && dists && idx && n1 > 0 && n1 <= max_queries && n2 > 0, E2E_ERR_ARG,
                "e2e_knn1_index_query: bad argument (n1=%lld must not exceed the max_queries=%lld the index was built for)", (long long)n1, (long long)max_queries);
    const GridWs g = grid_ws(index, max_queries, n2);
    hipLaunchKernelGGL(k_grid_reset_unresolved, dim3(1), dim3(1), 0, (hipStream_t)stream, g.gi);
    grid_query(p1, n1, g, dists, idx, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_knn1_index_query");
    return E2E_OK;
}

int e2e_knn1_fwd(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists, long long* idx, void* workspace,
                  int algorithm, void* stream) {
    E2E_REQUIRE(n1 > 0 && n2 > 0 && n2 < 0xFFFFFFFFll && n1 < 0xFFFFFFFFll, E2E_ERR_ARG, "e2e_knn1_fwd: bad sizes n1=%lld n2=%lld", (long long)n1, (long long)n2);
    E2E_REQUIRE(p1 && p2 && dists && idx && workspace, E2E_ERR_ARG, "e2e_knn1_fwd: null pointer");
    E2E_REQUIRE(algorithm >= 0 && algorithm <= 2, E2E_ERR_ARG, "e2e_knn1_fwd: algorithm %d (0 auto, 1 brute force, 2 grid)", algorithm);
    hipStream_t st = (hipStream_t)stream;
    const bool grid = algorithm == 2 || (algorithm == 0 && n2 >= KNN_GRID_MIN_N2);
    const int rc = grid ? knn1_grid(p1, n1, p2, n2, dists, idx, workspace, st) : knn1_brute(p1, n1, p2, n2, dists, idx, workspace, st);
    if (rc) return rc;
    E2E_LAUNCH_CHECK("e2e_knn1_fwd");
    return E2E_OK;
}

static int knn1_brute(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists, long long* idx, void* workspace, hipStream_t st) {
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

int e2e_knn1_bwd_ref(const float* g_dists, const float* p1, const float* p2, const long long* idx, int64_t n1, int64_t n2, long long* scratch_fixed,
                     float* g_p2, void* stream) {
    E2E_REQUIRE(n1 > 0 && n2 > 0 && g_dists && p1 && p2 && idx && scratch_fixed && g_p2, E2E_ERR_ARG, "e2e_knn1_bwd_ref: bad argument");
    hipStream_t st = (hipStream_t)stream;
    E2E_REQUIRE(hipMemsetAsync(scratch_fixed, 0, (size_t)(3 * n2 + 1) * sizeof(long long), st) == hipSuccess, E2E_ERR_LAUNCH,
                "e2e_knn1_bwd_ref: hipMemsetAsync of the fixed-point scratch failed");
    const int64_t qb = (n1 + KT - 1) / KT, cb = (3 * n2 + 255) / 256;
    hipLaunchKernelGGL(k_knn1_bwd_ref, dim3((unsigned)(qb > 4096 ? 4096 : qb)), dim3(KT), 0, st, g_dists, p1, p2, idx, n1, n2, (unsigned long long*)scratch_fixed);
    hipLaunchKernelGGL(k_knn1_fixed48_to_float, dim3((unsigned)(cb > 4096 ? 4096 : cb)), dim3(256), 0, st, scratch_fixed, g_p2, 3 * n2);
    E2E_LAUNCH_CHECK("e2e_knn1_bwd_ref");
    return E2E_OK;
}

}  // extern "C"
