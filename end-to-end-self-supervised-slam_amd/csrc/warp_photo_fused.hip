// warp_photo_fused.hip -- ONE launch for the image-space part of a refinement step on gfx950:
//   depth, src, tgt  ->  photometric (+ regulariser) loss  AND  d(loss)/d(depth)
//
// The gradient of a mean does not depend on the value of the mean, so forward and backward of
//   backproject -> project -> grid_sample -> mask -> SSIM/L1 -> mean   (+ depth regulariser)
// (online_adaption.py:412-455, :544-564, :482-511, :612-623) collapse into a single pass over the
// frame: synth / valid / the loss map never touch HBM.  Algorithmic traffic per pixel: depth 4 B +
// src 12 B (gather) + tgt 12 B in, d/d(depth) 4 B out (+ regulariser: 12 B in, 4 B out).
//
// Workgroup = 256 threads = 32x16 pixel tile, two pixels per thread.
//   phase 1  warp every position of the tile + 2-px halo (36x20): {x,y} = {synth*m, tgt*m} -> LDS as
//            float2 (one ds_read_b64 serves both SSIM operands; window sums run on v_pk_*_f32);
//            the thread keeps its own two pixels' taps / projection in registers for the adjoint.
//   phase 2  SSIM statistics at every q of tile + 1-px halo (34x18), walked as 3 x 5 x 34 (channel, 4-row
//            strip, column) units, two per thread, two vertically adjacent q's per v_pk_* operation:
//            loss contribution of the tile's own pixels and (G1,G2,G3) = upstream *
//            dS/d(mu_x, E[x^2], E[xy]) -> nine LDS planes.
//   phase 3  per pixel: fold the 3x3 neighbourhood of G back (reflect-pad adjoint = multiplicities
//            at the image border), bilinear-tap adjoint from the kept registers, chain to depth.
// The loss sums leave the kernel either as per-workgroup partials for k_reduce_partials2 (classic)
// or, in the chained form, as order-independent fixed-point slot sums that the NEXT launch finalises
// (one kernel per step).
// fp contraction is ON here (FMA): tolerance for this path is 1e-4 relative (BASELINE.json); the
// bit-exact index kernels live in other files with contraction off.
#include <stdlib.h>

#include "e2e_common.h"
#pragma clang fp contract(fast)

#define LT_W 32
#define LX_W (LT_W + 4)
#define LG_W (LT_W + 2)
#define LNT 256

typedef float f2 __attribute__((ext_vector_type(2)));

struct HostGeo {       // c = d * (M [x,y,1]) + p4 : rows of M (9) then p4 (3); use = 1 when the caller computed it on the host
    float g[12];
    int use;
};

// Chained launches (e2e_warp_photo_lossgrad_chain): instead of leaving per-workgroup partial sums for a second-stage
// kernel, every workgroup adds its partial sums -- converted to 2^-36 fixed point, so the integer sum is independent of
// the arrival order and bitwise reproducible -- into one of 64 slots of slot set `cur`, and workgroup 0 turns the
// COMPLETE set `prev` of the previous launch into that launch's loss and clears it.  One kernel per step; the last
// step of a chain is finished by e2e_warp_photo_lossgrad_chain_flush.  (64 slots: same-address device atomics
// serialise at ~0.18 us each, 600 workgroups on one address would cost 100 us.)
#define CHAIN_SLOTS 64
#define CHAIN_SET_U64 (2 * CHAIN_SLOTS + 1)          // two sums x 64 slots + a "non-finite partial" flag
#define CHAIN_SETS 8
#define CHAIN_FIX 68719476736.0f                     // 2^36
struct Chain {
    unsigned long long* slots;                        // NULL: classic mode (partials + second-stage kernel)
    int cur, prev;                                    // prev < 0: nothing to finalise
    float* loss_prev;
    double scale;                                     // 1 / (B H W)
    int prev_sets;                                    // 2 when the regulariser sum exists, else 1
};

__device__ __forceinline__ void chain_finalise(unsigned long long* set, int nsums, double scale, float* out, int lane) {
    // one wave: lane l owns slot l of both sums
    unsigned long long v0 = set[lane], v1 = (nsums > 1) ? set[CHAIN_SLOTS + lane] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v0 += __shfl_xor(v0, o, 64);
        v1 += __shfl_xor(v1, o, 64);
    }
    const bool bad = set[2 * CHAIN_SLOTS] != 0ull;
    set[lane] = 0ull;
    set[CHAIN_SLOTS + lane] = 0ull;
    if (lane == 0) {
        set[2 * CHAIN_SLOTS] = 0ull;
        const float nanv = __uint_as_float(0x7FC00000u);
        out[0] = bad ? nanv : (float)((double)v0 * (1.0 / (double)CHAIN_FIX) * scale);
        out[1] = (nsums > 1) ? (bad ? nanv : (float)((double)v1 * (1.0 / (double)CHAIN_FIX) * scale)) : 0.f;
    }
}

struct Taps {          // what the adjoint needs about one of the thread's own pixels
    float nw[3], ne[3], sw[3], se[3];
    float tx, ty, mx, my;   // bilinear fractions, d(ix)/d(u) and d(iy)/d(v) (0 where the border clamp is active)
    float r0, r1, r2, rz, u, v;
    float m;                // validity mask (1 when use_mask == 0)
};

// PPT = image rows per thread (1 or 2).  Tile = 32 x (8*PPT) pixels, 256 threads.
template <int PAD, bool NHWC, int PPT>
__global__ __launch_bounds__(LNT) void k_warp_photo_lossgrad(
    const float* __restrict__ depth, const float* __restrict__ src, e2e_strides ss,
    const float* __restrict__ tgt, e2e_strides ts, const float* __restrict__ K,
    const float* __restrict__ invK, const float* __restrict__ T, int use_mask, int reg_kind,
    const float* __restrict__ ri_t, const float* __restrict__ ri_s, const float* __restrict__ d_s,
    float w_photo, float w_reg, float* __restrict__ g_dt, float* __restrict__ g_ds,
    float* __restrict__ partials, int B, int H, int W, HostGeo hg, Chain chain) {
    constexpr int LT_H = 8 * PPT, LX_H = LT_H + 4, LG_H = LT_H + 2;
    constexpr int N_HALO = LX_W * LX_H - LT_W * LT_H;      // 176 (PPT 1) / 208 (PPT 2)
    constexpr int NE = PPT + 1;                            // warp evaluations per thread
    __shared__ f2 sxy[3][LX_H * LX_W];
    __shared__ float sg[9][LG_H * LG_W];
    __shared__ float sgeo[12];
    __shared__ float red[LNT / 64];
    const int b = blockIdx.z, tx0 = blockIdx.x * LT_W, ty0 = blockIdx.y * LT_H;
    const int tid = threadIdx.y * LT_W + threadIdx.x;
    const int64_t N = (int64_t)H * W;
    const float* dep = depth + b * N;
    const float* sb = src + b * ss.sb;
    const float* tb = tgt + b * ts.sb;

    // ---- phase 0: per-batch geometry, once per workgroup:  c = d * (M [x,y,1]) + p4 ----------------
    // (skipped when the 12 numbers arrive as kernel arguments: saves a dependent global round trip + a barrier)
    if (!hg.use && tid < 12) {
        const float* Kb = K + b * 16; const float* Tb = T + b * 16; const float* Ib = invK + b * 16;
        const int r = (tid < 9) ? tid / 3 : tid - 9;
        float P[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) P[j] = fmaf(Kb[r * 4 + 0], Tb[0 * 4 + j], fmaf(Kb[r * 4 + 1], Tb[1 * 4 + j], fmaf(Kb[r * 4 + 2], Tb[2 * 4 + j], Kb[r * 4 + 3] * Tb[3 * 4 + j])));
        if (tid < 9) {
            const int cidx = tid % 3;
            sgeo[tid] = fmaf(P[0], Ib[0 * 4 + cidx], fmaf(P[1], Ib[1 * 4 + cidx], P[2] * Ib[2 * 4 + cidx]));
        } else {
            sgeo[tid] = P[3];
        }
    }

    // ---- phase 1: warp tile + halo into LDS; loads of all NE positions are issued together -----------
    int lpos[NE], qxs[NE], qys[NE];
    bool dom[NE], live[PPT];
    float dval[NE], tv[NE][3];
    float rg_it[PPT], rg_is[PPT], rg_ds[PPT];          // regulariser operands of the own pixels
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        int ly, lx;
        if (e < PPT) {
            ly = threadIdx.y * PPT + e + 2;
            lx = threadIdx.x + 2;
        } else {
            const int h = (tid < N_HALO) ? tid : 0;
            if (h < 2 * LX_W) { ly = h / LX_W; lx = h % LX_W; }
            else if (h < 4 * LX_W) { ly = LX_H - 2 + (h - 2 * LX_W) / LX_W; lx = (h - 2 * LX_W) % LX_W; }
            else { const int k4 = h - 4 * LX_W; ly = 2 + (k4 >> 2); const int k = k4 & 3; lx = (k < 2) ? k : LX_W - 4 + k; }
        }
        const int gy = ty0 + ly - 2, gx = tx0 + lx - 2;
        dom[e] = gx >= -1 && gx <= W && gy >= -1 && gy <= H;
        if (e < PPT) {
            live[e] = gx < W && gy < H;
            rg_it[e] = rg_is[e] = rg_ds[e] = 0.f;
            if (reg_kind && live[e]) {
                const int64_t o = b * N + (int64_t)gy * W + gx;
                rg_it[e] = ri_t[o]; rg_is[e] = ri_s[o]; rg_ds[e] = d_s[o];
            }
        }
        // reflect, then clamp so that even unused slots address valid memory (their result is zeroed)
        qxs[e] = min(max(reflect1(gx, W), 0), W - 1);
        qys[e] = min(max(reflect1(gy, H), 0), H - 1);
        lpos[e] = ly * LX_W + lx;
        dval[e] = dep[qys[e] * W + qxs[e]];
        if (NHWC) {
            const float* tp = tb + (qys[e] * W + qxs[e]) * 3;
            tv[e][0] = tp[0]; tv[e][1] = tp[1]; tv[e][2] = tp[2];
        } else {
            const float* tp = tb + qys[e] * ts.sh + qxs[e] * ts.sw;
            tv[e][0] = tp[0]; tv[e][1] = tp[ts.sc]; tv[e][2] = tp[2 * ts.sc];
        }
    }
    float geo[12];
    if (hg.use) {                                     // wave-uniform (kernel argument)
#pragma unroll
        for (int i = 0; i < 12; ++i) geo[i] = hg.g[i];
    } else {
        __syncthreads();                              // sgeo ready
#pragma unroll
        for (int i = 0; i < 12; ++i) geo[i] = sgeo[i];
    }
    const float sxw = (float)W / (float)(W - 1), syh = (float)H / (float)(H - 1);

    Taps kp[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        Taps& k = kp[e];
        const float x = (float)qxs[e], y = (float)qys[e];
        k.r0 = fmaf(geo[0], x, fmaf(geo[1], y, geo[2]));
        k.r1 = fmaf(geo[3], x, fmaf(geo[4], y, geo[5]));
        k.r2 = fmaf(geo[6], x, fmaf(geo[7], y, geo[8]));
        const float c0 = fmaf(dval[e], k.r0, geo[9]), c1 = fmaf(dval[e], k.r1, geo[10]), c2 = fmaf(dval[e], k.r2, geo[11]);
        k.rz = 1.f / (c2 + 1e-7f);     // IEEE divide: the tap coordinates decide which pixels are blended
        k.u = c0 * k.rz;
        k.v = c1 * k.rz;
        // mask: max(|gx|,|gy|) <= 1 with g = (u/(W-1) - 0.5)*2   <=>   0 <= u <= W-1 and 0 <= v <= H-1
        const float inb = (k.u >= 0.f && k.u <= (float)(W - 1) && k.v >= 0.f && k.v <= (float)(H - 1)) ? 1.f : 0.f;
        k.m = use_mask ? inb : 1.f;
        // ix = ((g+1)*W - 1)/2 = u*W/(W-1) - 0.5        (align_corners=False sampling of a /(W-1) grid)
        float ix = fmaf(k.u, sxw, -0.5f), iy = fmaf(k.v, syh, -0.5f);
        k.mx = sxw; k.my = syh;
        if (PAD == E2E_PAD_BORDER) {
            if (!(ix > 0.f)) { ix = 0.f; k.mx = 0.f; }
            if (ix >= (float)(W - 1)) { ix = (float)(W - 1); k.mx = 0.f; }
            if (!(iy > 0.f)) { iy = 0.f; k.my = 0.f; }
            if (iy >= (float)(H - 1)) { iy = (float)(H - 1); k.my = 0.f; }
        } else {   // non-finite / wild coordinates: every tap is out of bounds, keep the weights finite
            if (!(fabsf(ix) < 1e9f)) ix = -2.f;
            if (!(fabsf(iy) < 1e9f)) iy = -2.f;
        }
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        k.tx = ix - fx0; k.ty = iy - fy0;
        if (PAD == E2E_PAD_BORDER) {
            const int x0 = (int)fx0, y0 = (int)fy0;
            const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);   // weight is 0 whenever the clamp bites
            if (NHWC) {
                const float* p00 = sb + (y0 * W + x0) * 3;
                const float* p01 = sb + (y0 * W + x1) * 3;
                const float* p10 = sb + (y1 * W + x0) * 3;
                const float* p11 = sb + (y1 * W + x1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) { k.nw[c] = p00[c]; k.ne[c] = p01[c]; k.sw[c] = p10[c]; k.se[c] = p11[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* pc = sb + c * ss.sc;
                    k.nw[c] = pc[y0 * ss.sh + x0 * ss.sw]; k.ne[c] = pc[y0 * ss.sh + x1 * ss.sw];
                    k.sw[c] = pc[y1 * ss.sh + x0 * ss.sw]; k.se[c] = pc[y1 * ss.sh + x1 * ss.sw];
                }
            }
        } else {
            const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
            const bool ix0 = x0 >= 0 && x0 < W, ix1 = x0 + 1 >= 0 && x0 + 1 < W;
            const bool iy0 = y0 >= 0 && y0 < H, iy1 = y0 + 1 >= 0 && y0 + 1 < H;
            const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
            const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {              // unconditional loads from clamped addresses, then select
                const float* pc = sb + c * ss.sc;
                const float a = pc[cy0 * ss.sh + cx0 * ss.sw], bq = pc[cy0 * ss.sh + cx1 * ss.sw];
                const float cq = pc[cy1 * ss.sh + cx0 * ss.sw], dq = pc[cy1 * ss.sh + cx1 * ss.sw];
                k.nw[c] = (iy0 && ix0) ? a : 0.f;
                k.ne[c] = (iy0 && ix1) ? bq : 0.f;
                k.sw[c] = (iy1 && ix0) ? cq : 0.f;
                k.se[c] = (iy1 && ix1) ? dq : 0.f;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const Taps& k = kp[e];
        const float w00 = (1.f - k.tx) * (1.f - k.ty), w01 = k.tx * (1.f - k.ty), w10 = (1.f - k.tx) * k.ty, w11 = k.tx * k.ty;
        const float m = dom[e] ? k.m : 0.f;           // outside the reflect domain: zeros
        if (e < PPT || tid < N_HALO) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sv = fmaf(k.nw[c], w00, fmaf(k.ne[c], w01, fmaf(k.sw[c], w10, k.se[c] * w11)));
                sxy[c][lpos[e]] = (f2){sv * m, tv[e][c] * m};
            }
        }
    }
    __syncthreads();

    // ---- phase 2: SSIM statistics -> loss + (G1,G2,G3) ---------------------------------------------
    const float kmean = 1.f / ((float)B * (float)H * (float)W);
    const float gup = w_photo * kmean * (0.85f / 3.f);      // upstream gradient on every ssim_c(q)
    float lsum = 0.f;
    {
        // thread -> one column of the 34-wide statistics region and a strip of RPS consecutive rows.  The RPS+2 window
        // rows are read from LDS once per channel and their horizontal sums shared by the strip's q's; the q's are then
        // processed in vertical PAIRS so that the whole SSIM / gradient formula runs on v_pk_*_f32 (the kernel is
        // VALU-issue bound: profiles/r01_warp_photo_lossgrad_history.md).  Threads past the last strip skip the phase.
        constexpr int RPS = 2 * PPT;                               // 4 (PPT 2) / 2 (PPT 1)
        constexpr int NSTRIP = (LG_H + RPS - 1) / RPS;             // 5
        constexpr int NUNIT = 3 * NSTRIP * LG_W;                   // (channel, strip, column) units: 510 -> 2 per thread
        constexpr int UPT = (NUNIT + LNT - 1) / LNT;
        f2 lacc = (f2){0.f, 0.f};
#pragma unroll 1
        for (int uu = 0; uu < UPT; ++uu) {
            const int unit = tid + uu * LNT;
            if (unit >= NUNIT) break;
            const int c = unit / (NSTRIP * LG_W), rem = unit - c * (NSTRIP * LG_W);
            const int strip = rem / LG_W, lx = rem - strip * LG_W;
            const int ly0 = strip * RPS;
            const int qx = tx0 + lx - 1;
            const bool col_ok = qx >= 0 && qx < W;
            const bool own_col = lx >= 1 && lx <= LT_W;
            {
                f2 hs1[RPS + 2], hs2[RPS + 2];
                float hxy[RPS + 2], adf[RPS + 2];
#pragma unroll
                for (int r = 0; r < RPS + 2; ++r) {
                    const int row = min(ly0 + r, LX_H - 1);        // rows past the region are never combined into a valid q
                    const f2* w = &sxy[c][row * LX_W + lx];
                    const f2 e0 = w[0], e1 = w[1], e2 = w[2];
                    hs1[r] = e0 + e1 + e2;
                    hs2[r] = e0 * e0 + e1 * e1 + e2 * e2;
                    hxy[r] = fmaf(e0.x, e0.y, fmaf(e1.x, e1.y, e2.x * e2.y));
                    adf[r] = fabsf(e1.y - e1.x);
                }
#pragma unroll
                for (int jp = 0; jp < RPS; jp += 2) {              // q rows (ly0+jp, ly0+jp+1) in the two halves of every f2
                    const int lya = ly0 + jp, lyb = lya + 1;
                    const int qya = ty0 + lya - 1, qyb = qya + 1;
                    const bool in_a = col_ok && lya < LG_H && qya >= 0 && qya < H;
                    const bool in_b = col_ok && lyb < LG_H && qyb >= 0 && qyb < H;
                    const f2 m1 = hs1[jp + 1] + hs1[jp + 2], m2 = hs2[jp + 1] + hs2[jp + 2];
                    const float mxy = hxy[jp + 1] + hxy[jp + 2];
                    const f2 s1a = m1 + hs1[jp], s1b = m1 + hs1[jp + 3];
                    const f2 s2a = m2 + hs2[jp], s2b = m2 + hs2[jp + 3];
                    const f2 k9 = (f2){1.f / 9.f, 1.f / 9.f};
                    const f2 MX = (f2){s1a.x, s1b.x} * k9, MY = (f2){s1a.y, s1b.y} * k9;
                    const f2 EXX = (f2){s2a.x, s2b.x} * k9, EYY = (f2){s2a.y, s2b.y} * k9;
                    const f2 EXY = (f2){mxy + hxy[jp], mxy + hxy[jp + 3]} * k9;
                    const f2 MXX = MX * MX, MYY = MY * MY, MXY = MX * MY;
                    const f2 SIGX = EXX - MXX, SIGY = EYY - MYY, SIGXY = EXY - MXY;
                    const f2 c1 = (f2){1e-4f, 1e-4f}, c2 = (f2){9e-4f, 9e-4f}, two = (f2){2.f, 2.f};
                    const f2 A1 = two * MXY + c1, A2 = two * SIGXY + c2;
                    const f2 B1 = MXX + MYY + c1, B2 = SIGX + SIGY + c2;
                    const f2 D = B1 * B2;
                    const f2 INV = (f2){__builtin_amdgcn_rcpf(D.x), __builtin_amdgcn_rcpf(D.y)};
                    const f2 S = A1 * A2 * INV;
                    const f2 TT = (f2){0.5f, 0.5f} - (f2){0.5f, 0.5f} * S;          // (1 - S)/2
                    const bool act_a = TT.x >= 0.f && TT.x <= 1.f, act_b = TT.y >= 0.f && TT.y <= 1.f;
                    const bool own_a = in_a && own_col && lya >= 1 && lya <= LT_H;
                    const bool own_b = in_b && own_col && lyb >= 1 && lyb <= LT_H;
                    const f2 lv = (f2){0.85f / 3.f, 0.85f / 3.f} * (f2){fminf(fmaxf(TT.x, 0.f), 1.f), fminf(fmaxf(TT.y, 0.f), 1.f)} +
                                  (f2){0.15f / 3.f, 0.15f / 3.f} * (f2){adf[jp + 1], adf[jp + 2]};
                    lacc += (f2){own_a ? lv.x : 0.f, own_b ? lv.y : 0.f};
                    // dS/dmu_x = 2 mu_y (A2 - A1)/(B1 B2) - 2 S mu_x (1/B1 - 1/B2) ; 1/B1 = B2*inv, 1/B2 = B1*inv
                    const f2 GI = (f2){act_a ? -0.5f * gup : 0.f, act_b ? -0.5f * gup : 0.f} * INV;
                    const f2 G1 = two * GI * (MY * (A2 - A1) - S * MX * (B2 - B1));
                    const f2 G2 = -(GI * S) * B1;
                    const f2 G3 = two * GI * A1;
                    if (lya < LG_H) {
                        const int i = lya * LG_W + lx;
                        sg[c * 3 + 0][i] = in_a ? G1.x : 0.f;
                        sg[c * 3 + 1][i] = in_a ? G2.x : 0.f;
                        sg[c * 3 + 2][i] = in_a ? G3.x : 0.f;
                    }
                    if (lyb < LG_H) {
                        const int i = lyb * LG_W + lx;
                        sg[c * 3 + 0][i] = in_b ? G1.y : 0.f;
                        sg[c * 3 + 1][i] = in_b ? G2.y : 0.f;
                        sg[c * 3 + 2][i] = in_b ? G3.y : 0.f;
                    }
                }
            }
        }
        lsum = lacc.x + lacc.y;
    }
    __syncthreads();

    // ---- phase 3: adjoint per own pixel (PPT vertically adjacent pixels share window rows) ----------
    float rsum = 0.f;
    const float gl1 = w_reg * kmean;
    {
        const int lxc = threadIdx.x, ly0 = threadIdx.y * PPT;          // tile-local
        const int px = tx0 + lxc;
        float wx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int qx = px + k - 1;
            wx[k] = (qx < 0 || qx >= W) ? 0.f : (((qx == 0 && px == 1) || (qx == W - 1 && px == W - 2)) ? 2.f : 1.f);
        }
        float gsy[PPT][3];     // d loss / d synth per own pixel and channel
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float rs[3][PPT + 2];      // x-folded row sums of G1,G2,G3 for the PPT+2 rows the pixels touch
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < PPT + 2; ++r) {
                    const float* row = &sg[c * 3 + j][(ly0 + r) * LG_W + lxc];
                    rs[j][r] = fmaf(row[0], wx[0], fmaf(row[1], wx[1], row[2] * wx[2]));
                }
#pragma unroll
            for (int e = 0; e < PPT; ++e) {
                const int py = ty0 + ly0 + e;
                float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int qy = py + k - 1;
                    const float wy = (qy < 0 || qy >= H) ? 0.f : (((qy == 0 && py == 1) || (qy == H - 1 && py == H - 2)) ? 2.f : 1.f);
                    a1 = fmaf(rs[0][e + k], wy, a1);
                    a2 = fmaf(rs[1][e + k], wy, a2);
                    a3 = fmaf(rs[2][e + k], wy, a3);
                }
                const f2 ce = sxy[c][(ly0 + e + 2) * LX_W + lxc + 2];
                float g = (a1 + 2.f * ce.x * a2 + ce.y * a3) * (1.f / 9.f);
                const float df = ce.y - ce.x;
                const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                g = fmaf(-w_photo * kmean * (0.15f / 3.f), sgn, g);
                gsy[e][c] = g * kp[e].m;                             // d/d synth = d/dx * mask
            }
        }
#pragma unroll
        for (int e = 0; e < PPT; ++e) {
            if (!live[e]) continue;
            const int py = ty0 + ly0 + e;
            const int64_t o = b * N + (int64_t)py * W + px;
            const Taps& kq = kp[e];
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gix = fmaf(gsy[e][c], fmaf(kq.ne[c] - kq.nw[c], 1.f - kq.ty, (kq.se[c] - kq.sw[c]) * kq.ty), gix);
                giy = fmaf(gsy[e][c], fmaf(kq.sw[c] - kq.nw[c], 1.f - kq.tx, (kq.se[c] - kq.ne[c]) * kq.tx), giy);
            }
            const float gu = gix * kq.mx, gv = giy * kq.my;
            const float gc0 = gu * kq.rz, gc1 = gv * kq.rz, gc2 = -(gu * kq.u + gv * kq.v) * kq.rz;
            float gd = fmaf(gc0, kq.r0, fmaf(gc1, kq.r1, gc2 * kq.r2));
            if (reg_kind) {
                const float e0 = rg_it[e] - dval[e], e1 = rg_is[e] - rg_ds[e];
                if (reg_kind == 2) {
                    rsum += e0 * e0 + e1 * e1;
                    gd = fmaf(gl1, -2.f * e0, gd);
                    g_ds[o] = gl1 * (-2.f * e1);
                } else {
                    rsum += fabsf(e0) + fabsf(e1);
                    gd += gl1 * ((e0 > 0.f) ? -1.f : ((e0 < 0.f) ? 1.f : 0.f));
                    g_ds[o] = gl1 * ((e1 > 0.f) ? -1.f : ((e1 < 0.f) ? 1.f : 0.f));
                }
            }
            g_dt[o] = gd;
        }
    }

    // ---- loss: per-workgroup partial sums; a 2-wave second-stage kernel adds them in a fixed order
    // (bitwise reproducible).  A last-arriver reduction inside this kernel was measured and rejected:
    // the per-workgroup agent-scope release (buffer_wbl2) behind freshly written gradients took the
    // kernel from 13 us to 35 us (profiles/r01_notes.md).
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const float s0 = block_sum(lsum, red);
    float s1 = 0.f;
    if (reg_kind) {
        __syncthreads();
        s1 = block_sum(rsum, red);
    }
    if (chain.slots == nullptr) {
        if (tid == 0) {
            partials[blk] = s0;
            if (reg_kind) partials[nblk + blk] = s1;
        }
        return;
    }
    if (tid == 0) {
        unsigned long long* set = chain.slots + (size_t)chain.cur * CHAIN_SET_U64;
        const int slot = blk & (CHAIN_SLOTS - 1);
        // finite, non-negative and small enough that the 2^-36 fixed-point total of all workgroups stays below 2^64
        const float lim = 2.6e8f / (float)nblk;
        const bool ok = s0 >= 0.f && s0 < lim && (!reg_kind || (s1 >= 0.f && s1 < lim));
        if (ok) {
            atomicAdd(&set[slot], (unsigned long long)(s0 * CHAIN_FIX));
            if (reg_kind) atomicAdd(&set[CHAIN_SLOTS + slot], (unsigned long long)(s1 * CHAIN_FIX));
        } else {
            atomicOr(&set[2 * CHAIN_SLOTS], 1ull);
        }
    }
    if (blk == 0 && chain.prev >= 0 && tid < 64)
        chain_finalise(chain.slots + (size_t)chain.prev * CHAIN_SET_U64, chain.prev_sets, chain.scale, chain.loss_prev, tid);
}

__global__ __launch_bounds__(64) void k_chain_flush(unsigned long long* set, int nsums, double scale, float* out) {
    chain_finalise(set, nsums, scale, out, threadIdx.x);
}

// second stage: fixed-order sum of the per-workgroup partials (1 workgroup, loads issued up front)
#define RED_T 256
#define RED_MAXV 8          // up to 2048 partials per set in the unrolled path
__global__ __launch_bounds__(RED_T) void k_reduce_partials2(const float* __restrict__ partials, int nblk, int nsets, double scale,
                                                            float* __restrict__ out) {
    __shared__ double sh[2][RED_T / 64];
    const int tid = threadIdx.x;
    float v[2][RED_MAXV];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < RED_MAXV; ++j) {
            const int i = tid + j * RED_T;
            v[s][j] = (s < nsets && i < nblk) ? partials[(int64_t)s * nblk + i] : 0.f;
        }
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < RED_MAXV; ++j) acc[s] += (double)v[s][j];
        if (s < nsets)
            for (int i = tid + RED_MAXV * RED_T; i < nblk; i += RED_T) acc[s] += (double)partials[(int64_t)s * nblk + i];
        acc[s] = wave_sum_d(acc[s]);
        if ((tid & 63) == 0) sh[s][tid >> 6] = acc[s];
    }
    __syncthreads();
    if (tid < 2) out[tid] = (float)((((sh[tid][0] + sh[tid][1]) + sh[tid][2]) + sh[tid][3]) * scale);
}

extern "C" int64_t e2e_warp_photo_lossgrad_workspace_floats(int B, int H, int W);

extern "C" {

int64_t e2e_warp_photo_lossgrad_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    // 2 partial sums per workgroup of the finest tiling (32x8), then (8-byte aligned) the slot sets of the chained mode
    const int64_t parts = (2ll * e2e_ceil_div(W, LT_W) * e2e_ceil_div(H, 8) * B + 1) / 2 * 2;
    return parts + 2ll * CHAIN_SETS * CHAIN_SET_U64;
}

static unsigned long long* chain_area(float* workspace, int B, int H, int W) {
    const int64_t parts = (2ll * e2e_ceil_div(W, LT_W) * e2e_ceil_div(H, 8) * B + 1) / 2 * 2;
    return (unsigned long long*)(workspace + parts);
}

static int lossgrad_impl(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                         const float* K, const float* inv_K, const float* T, HostGeo hg, int use_mask, int padding_mode,
                         int reg_kind, const float* reg_init_tgt, const float* reg_init_src, const float* depth_src,
                         float w_photo, float w_reg, float* loss_out, float* g_depth_tgt, float* g_depth_src,
                         float* workspace, int B, int H, int W, void* stream, Chain chain = Chain{}) {
    E2E_REQUIRE(B > 0 && H > 1 && W > 1 && (int64_t)B * H * W * 3 < (1ll << 31), E2E_ERR_ARG,
                "e2e_warp_photo_lossgrad: bad dims B=%d H=%d W=%d", B, H, W);
    E2E_REQUIRE(depth_tgt && src && tgt && (hg.use || (K && inv_K && T)) && g_depth_tgt && workspace, E2E_ERR_ARG,
                "e2e_warp_photo_lossgrad: null pointer");
    E2E_REQUIRE(padding_mode == E2E_PADDING_BORDER || padding_mode == E2E_PADDING_ZEROS, E2E_ERR_ARG,
                "e2e_warp_photo_lossgrad: padding_mode %d not supported (zeros|border)", padding_mode);
    E2E_REQUIRE(reg_kind >= 0 && reg_kind <= 2, E2E_ERR_ARG, "e2e_warp_photo_lossgrad: reg_kind %d (0 none, 1 l1, 2 l2)", reg_kind);
    E2E_REQUIRE(!reg_kind || (reg_init_tgt && reg_init_src && depth_src && g_depth_src), E2E_ERR_ARG,
                "e2e_warp_photo_lossgrad: regulariser buffers missing");
    // two rows per thread (32x16 tiles) measured faster than one row (32x8) at both 1 and 8 pairs per launch
    // (the one-row variant stays compiled for the A/B in profiles/r01_warp_photo_lossgrad_history.md; no per-launch environment lookup)
    static const int ppt = [] { const char* ev = getenv("E2E_LOSSGRAD_PPT"); return (ev && ev[0] == '1') ? 1 : 2; }();
    const dim3 g(e2e_ceil_div(W, LT_W), e2e_ceil_div(H, 8 * ppt), B);
    const int nblk = g.x * g.y * g.z;
    // channels-last frames (the reference's memory layout) take the 12-byte-per-tap path
    const bool nhwc = ss.sc == 1 && ss.sw == 3 && ss.sh == 3ll * W && ts.sc == 1 && ts.sw == 3 && ts.sh == 3ll * W;
    hipStream_t st = (hipStream_t)stream;
#define LG_ARGS depth_tgt, src, ss, tgt, ts, K, inv_K, T, use_mask, reg_kind, reg_init_tgt, reg_init_src, depth_src, w_photo, w_reg, \
                g_depth_tgt, g_depth_src, workspace, B, H, W, hg, chain
#define LG_LAUNCH(PADV, NH)                                                                                              \
    do {                                                                                                                 \
        if (ppt == 2) hipLaunchKernelGGL((k_warp_photo_lossgrad<PADV, NH, 2>), g, dim3(LT_W, 8), 0, st, LG_ARGS);        \
        else hipLaunchKernelGGL((k_warp_photo_lossgrad<PADV, NH, 1>), g, dim3(LT_W, 8), 0, st, LG_ARGS);                 \
    } while (0)
    if (padding_mode == E2E_PADDING_BORDER) {
        if (nhwc) LG_LAUNCH(E2E_PAD_BORDER, true);
        else LG_LAUNCH(E2E_PAD_BORDER, false);
    } else {
        LG_LAUNCH(E2E_PAD_ZEROS, false);
    }
#undef LG_LAUNCH
#undef LG_ARGS
    E2E_LAUNCH_CHECK("e2e_warp_photo_lossgrad");
    if (loss_out && !chain.slots) {     // loss_out == NULL: gradients only (the per-workgroup partial sums stay in the workspace)
        hipLaunchKernelGGL(k_reduce_partials2, dim3(1), dim3(RED_T), 0, st, workspace, nblk, reg_kind ? 2 : 1,
                           1.0 / ((double)B * H * W), loss_out);
        E2E_LAUNCH_CHECK("e2e_warp_photo_lossgrad(reduce)");
    }
    return E2E_OK;
}

int e2e_warp_photo_lossgrad(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                            const float* K, const float* inv_K, const float* T, int use_mask, int padding_mode,
                            int reg_kind, const float* reg_init_tgt, const float* reg_init_src, const float* depth_src,
                            float w_photo, float w_reg, float* loss_out, float* g_depth_tgt, float* g_depth_src,
                            float* workspace, int B, int H, int W, void* stream) {
    HostGeo hg{};
    hg.use = 0;
    return lossgrad_impl(depth_tgt, src, ss, tgt, ts, K, inv_K, T, hg, use_mask, padding_mode, reg_kind, reg_init_tgt, reg_init_src,
                         depth_src, w_photo, w_reg, loss_out, g_depth_tgt, g_depth_src, workspace, B, H, W, stream);
}

int e2e_warp_photo_lossgrad_hostgeo(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                                    const float* geometry12_host, int use_mask, int padding_mode, int reg_kind,
                                    const float* reg_init_tgt, const float* reg_init_src, const float* depth_src, float w_photo,
                                    float w_reg, float* loss_out, float* g_depth_tgt, float* g_depth_src, float* workspace, int H,
                                    int W, void* stream) {
    E2E_REQUIRE(geometry12_host, E2E_ERR_ARG, "e2e_warp_photo_lossgrad_hostgeo: geometry is NULL");
    HostGeo hg{};
    for (int i = 0; i < 12; ++i) hg.g[i] = geometry12_host[i];
    hg.use = 1;
    return lossgrad_impl(depth_tgt, src, ss, tgt, ts, nullptr, nullptr, nullptr, hg, use_mask, padding_mode, reg_kind, reg_init_tgt,
                         reg_init_src, depth_src, w_photo, w_reg, loss_out, g_depth_tgt, g_depth_src, workspace, 1, H, W, stream);
}

int e2e_warp_photo_lossgrad_chain(const float* depth_tgt, const float* src, e2e_strides ss, const float* tgt, e2e_strides ts,
                                  const float* K, const float* inv_K, const float* T, const float* geometry12_host, int use_mask,
                                  int padding_mode, int reg_kind, const float* reg_init_tgt, const float* reg_init_src,
                                  const float* depth_src, float w_photo, float w_reg, int set_cur, int set_prev, float* loss_prev_out,
                                  float* g_depth_tgt, float* g_depth_src, float* workspace, int B, int H, int W, void* stream) {
    E2E_REQUIRE(set_cur >= 0 && set_cur < CHAIN_SETS && set_prev < CHAIN_SETS && set_prev != set_cur && (set_prev < 0 || loss_prev_out),
                E2E_ERR_ARG, "e2e_warp_photo_lossgrad_chain: slot sets must be distinct and in [0, %d); set_prev >= 0 needs loss_prev_out", CHAIN_SETS);
    E2E_REQUIRE(!geometry12_host || B == 1, E2E_ERR_ARG, "e2e_warp_photo_lossgrad_chain: host geometry describes one pair (B == 1)");
    E2E_REQUIRE(workspace && B > 0 && H > 1 && W > 1, E2E_ERR_ARG, "e2e_warp_photo_lossgrad_chain: bad argument");
    HostGeo hg{};
    hg.use = geometry12_host ? 1 : 0;
    if (geometry12_host)
        for (int i = 0; i < 12; ++i) hg.g[i] = geometry12_host[i];
    Chain c{};
    c.slots = chain_area(workspace, B, H, W);
    c.cur = set_cur; c.prev = set_prev; c.loss_prev = loss_prev_out;
    c.scale = 1.0 / ((double)B * H * W);
    c.prev_sets = reg_kind ? 2 : 1;
    return lossgrad_impl(depth_tgt, src, ss, tgt, ts, K, inv_K, T, hg, use_mask, padding_mode, reg_kind, reg_init_tgt, reg_init_src,
                         depth_src, w_photo, w_reg, nullptr, g_depth_tgt, g_depth_src, workspace, B, H, W, stream, c);
}

int e2e_warp_photo_lossgrad_chain_flush(float* workspace, int set, int reg_kind, float* loss_out, int B, int H, int W, void* stream) {
    E2E_REQUIRE(workspace && loss_out && set >= 0 && set < CHAIN_SETS && B > 0 && H > 1 && W > 1, E2E_ERR_ARG, "e2e_warp_photo_lossgrad_chain_flush: bad argument");
    hipLaunchKernelGGL(k_chain_flush, dim3(1), dim3(64), 0, (hipStream_t)stream, chain_area(workspace, B, H, W) + (size_t)set * CHAIN_SET_U64,
                       reg_kind ? 2 : 1, 1.0 / ((double)B * H * W), loss_out);
    E2E_LAUNCH_CHECK("e2e_warp_photo_lossgrad_chain_flush");
    return E2E_OK;
}

}  // extern "C"
