// conv.hip -- the depth network's convolutions for gfx950: fp32 implicit GEMM on v_mfma_f32_32x32x2_f32
// (exact fp32: a k-ordered fmaf chain, so the 1e-4 parity tolerance holds; bf16 MFMA would not).
// Replaces torch/MIOpen conv + separate BN / activation / reflection-pad / upsample / concat kernels of
// depth_estimation/networks.py:44-57 (encoder), :157-189 (ConvBlock / Conv3x3), :277-292 (decoder).
//
// Activations are NHWC.  One GEMM kernel serves
//   forward        out[n, co]  = act( scale[co] * sum_k A[n,k] W[k,co] + shift[co] (+ res[n,co]) )
//                  n = (b,oh,ow), k = (kh,kw,ci), A gathered on the fly: stride, zero / reflection padding, nearest x2
//                  upsample of a low-res source and channel concatenation with a skip tensor -- none materialised;
//   backward-data  dXp[n, ci] = sum_k A'[n,k] W'[k,ci],  n = (b,y,x) over the (padded) input domain, k = (kh,kw,co),
//                  A'[n,k] = dZ[b,(y-kh)/s,(x-kw)/s,co] where that is an integer in range  (TRANSPOSED gather)
// and a second kernel computes backward-weight as a split-K GEMM over pixels with fixed-order slab reduction.
//
// Tiling: workgroup = WM x WN waves, each wave owns 32 rows x (32*TN) columns (TN accumulators of 16 VGPRs);
// K is consumed in chunks of 32 (16 where Cin % 32 != 0) through double-buffered LDS tiles stored k-major so that both
// MFMA operands are bank-conflict-free ds_read_b32 (lane l reads [k = 2kk + (l>>5)][i = l & 31]).  fp32 MFMA issues one
// instruction per 64 cycles per SIMD, so one wave per SIMD with its accumulators in flight saturates the matrix pipe;
// global loads of the next chunk are issued before the current chunk's MFMAs.  Operands are gathered through buffer
// resources with 32-bit offsets: padding, stride holes and tile tails are out-of-range offsets that the hardware
// answers with zeros, so the K loop has no branches and no 64-bit address arithmetic.
#include <type_traits>

#include "e2e_common.h"

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CBK 16

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_DISP = 3 };

struct ConvArgs {
    // sources: channels [0,C1) come from src0 (at 1/up resolution), channels [C1,Cin) from src1 (full resolution)
    const float* src0;
    const float* src1;
    const float* w;        // GEMM B operand, [K][ldw] row-major (k-major), ldw >= Ncols
    const float* scale;    // per output column or NULL
    const float* shift;    // per output column or NULL
    const float* res;      // residual [n][Ncols] or NULL
    float* out;            // [n][Ncols]
    int B, Hs, Ws;         // spatial size of the gather SOURCE domain at full resolution (src1 / upsampled src0)
    int Cin, C1, up;       // gather channels, split point, upsample factor of src0 (1 or 2)
    int Hd, Wd;            // spatial size of the OUTPUT domain (rows n = (b, yd, xd))
    int Ncols, ldw;        // GEMM N (= Cout fwd, = Cin of the conv for backward-data) and weight leading dimension
    int KH, KW, stride, pad, pad_mode;   // pad_mode 0 zeros, 1 reflect (forward only)
    int off;               // backward-data: offset added to (yd,xd) to get padded-domain coordinates
    int act;
    float in_sub, in_mul;  // VEC==1 path only: gathered value -> (v - in_sub) * in_mul  (input normalisation, networks.py:50)
    // split-K (layers with few output tiles): grid.z slices of `cps` chunks write raw accumulators to slab[z][n][Ncols];
    // k_conv_splitk_epilogue adds the slices in order and applies the epilogue.  ksplit == 1: direct epilogue.
    int ksplit, cps;
    float* slab;
    // backward-data of a stride-2 convolution: the input-gradient pixels split into 4 parity classes (y & 1, x & 1); a pixel
    // of class (py, px) only meets the taps with kh = (py + off) mod 2 (+2), kw likewise -- 1, 2, 2 and 4 of the 9 taps of a
    // 3x3 kernel.  cls != 0: workgroup blockIdx.x handles tile blockIdx.x >> 2 of class blockIdx.x & 3 with that tap subset
    // (the plain transposed gather walks all 9 taps for every pixel and finds 3/4 of them to be stride holes).
    int cls;
    // backward-data only: the gradient this launch produces belongs to the tensor x_in = act_in(u) that the forward convolution READ
    // (same [row][Ncols] layout as the output); with dact != 0 the epilogue multiplies by act_in'(u), taken from x_in itself
    // (ReLU: [x > 0]; ELU: x > 0 ? 1 : x + 1), so the launch emits d loss / d u -- the gradient with respect to the PRE-activation
    // of the producing layer -- and that layer needs no separate activation-backward pass (e2ehip.netplan).
    const float* xin;
    int dact;
    // transposed form: optional addend in the output's layout, added BEFORE the act' factor: out = (acc + pre) * act'(x) [+ res].
    // A BasicBlock's input receives conv1's backward-data and the block's residual gradient -- both are gradients with respect to
    // the same tensor, so the residual one rides here instead of in a separate accumulate pass.
    const float* pre;
    int64_t cls_rows;                 // class form with per-tap slices: rows of one (slice, class) slab = B * ceil(Hd/2) * ceil(Wd/2)
    int64_t bytes0, bytes1, bytesw;   // extents of src0 / src1 / w for the buffer resources (< 2 GB each)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == ACT_DISP) return 10.f / (1.f + expf(-v)) + 0.01f;          // networks.py:290
    return v;
}

// act'(u) recovered from the activation's OUTPUT y = act(u)
__device__ __forceinline__ float act_deriv(float y, int act) {
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// source pixel for output-domain pixel (yd,xd) and tap (kh,kw); returns false when the tap reads a structural zero
template <bool TRANSPOSED>
__device__ __forceinline__ bool tap_coord(const ConvArgs& a, int yd, int xd, int kh, int kw, int& ys, int& xs) {
    if (!TRANSPOSED) {
        ys = yd * a.stride + kh - a.pad;
        xs = xd * a.stride + kw - a.pad;
        if (a.pad_mode == 1) {
            ys = reflect1(ys, a.Hs);
            xs = reflect1(xs, a.Ws);
            return true;
        }
        return ys >= 0 && ys < a.Hs && xs >= 0 && xs < a.Ws;
    } else {
        const int ty = yd + a.off - kh, tx = xd + a.off - kw;
        if (ty < 0 || tx < 0) return false;
        if (a.stride == 2) {
            if ((ty | tx) & 1) return false;
            ys = ty >> 1;
            xs = tx >> 1;
        } else {
            ys = ty;
            xs = tx;
        }
        return ys < a.Hs && xs < a.Ws;
    }
}

// ---- direct epilogue of one 32x32 accumulator block (shared by k_conv_gemm and k_conv_gemm_sk) ---------------------------------
// A lane holds column `col` and rows (r & 3) + 8 (r >> 2) of the block that starts at row `nb` (nb already includes the lane's
// 4 * (lane >> 5)): the 16 rows sit at FIXED distances from the lane's first row, so every store / residual load is one buffer
// instruction with the lane's byte offset in the VGPR and the row distance in the (wave-uniform) scalar offset -- no per-element
// address arithmetic; rows past the end of the tensor get an out-of-range VGPR offset (one compare + select per element in the last
// row tile; the hardware's range check does not include the scalar offset) and are dropped by the hardware.  The activation is
// selected once, outside the element loops.  (The first version spent ~145 instructions per element here -- 64-bit row arithmetic,
// bounds tests and a branch tree per element -- and every wave of a launch runs its epilogue at the same time, so none of it
// overlapped with MFMA work: ~30 % of the kernel's cycles on the 64x64 layers.)
struct EpiRsrc { __amdgpu_buffer_rsrc_t out, res, xin, pre; bool has_res; };

template <bool TRANSPOSED>
__device__ __forceinline__ EpiRsrc make_epi_rsrc(const ConvArgs& a) {
    const int64_t out_elems = (int64_t)a.B * a.Hd * a.Wd * a.Ncols;
    EpiRsrc e;
    e.out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(out_elems * 4), 0x00020000);
    e.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.out), 0, (int)(a.res ? out_elems * 4 : 0), 0x00020000);
    e.has_res = a.res != nullptr;
    e.xin = __builtin_amdgcn_make_buffer_rsrc((void*)((TRANSPOSED && a.xin) ? a.xin : a.out), 0, (int)((TRANSPOSED && a.xin) ? out_elems * 4 : 0), 0x00020000);
    e.pre = __builtin_amdgcn_make_buffer_rsrc((void*)((TRANSPOSED && a.pre) ? a.pre : a.out), 0, (int)((TRANSPOSED && a.pre) ? out_elems * 4 : 0), 0x00020000);
    return e;
}

template <bool TRANSPOSED>
__device__ __forceinline__ void epilogue_block(const ConvArgs& a, const EpiRsrc& er, const f16v& acc, int col, int64_t nb, int64_t Ntot) {
    constexpr unsigned OOB = 0x80000000u;
    const bool col_ok = col < a.Ncols;
    const float sc = (a.scale && col_ok) ? a.scale[col] : 1.f, sh = (a.shift && col_ok) ? a.shift[col] : 0.f;
    const unsigned voff0 = (col_ok && nb < Ntot) ? (unsigned)((nb * a.Ncols + col) * 4) : OOB;
    const int left = (int)((Ntot - nb < 32) ? Ntot - nb : 32);      // rows of this block that exist: the hardware's range check
    unsigned vo[16];                                                // ignores the scalar offset, so rows past the end are masked here
#pragma unroll
    for (int r = 0; r < 16; ++r) vo[r] = ((r & 3) + 8 * (r >> 2) < left) ? voff0 : OOB;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[r], sc, sh);
    if (TRANSPOSED && a.pre) {                      // a second gradient of the same tensor (residual branch)
        float pp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(er.pre, vo[r], ((r & 3) + 8 * (r >> 2)) * a.Ncols * 4, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += pp[r];
    }
    if (TRANSPOSED && a.dact) {                     // x (act') of the tensor this gradient belongs to
        float xx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xx[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(er.xin, vo[r], ((r & 3) + 8 * (r >> 2)) * a.Ncols * 4, 0));
        if (a.dact == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xx[r] > 0.f ? v[r] : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = xx[r] > 0.f ? v[r] : v[r] * (xx[r] + 1.f);
        }
    }
    if (er.has_res) {
        float rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(er.res, vo[r], ((r & 3) + 8 * (r >> 2)) * a.Ncols * 4, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rr[r];
    }
    if (a.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (a.act == ACT_ELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : expm1f(v[r]);
    } else if (a.act == ACT_DISP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 10.f / (1.f + expf(-v[r])) + 0.01f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), er.out, vo[r], ((r & 3) + 8 * (r >> 2)) * a.Ncols * 4, 0);
}

// Thread -> A element mapping.  VEC == 4 (every layer but conv1): consecutive lanes fetch consecutive 16-byte channel
// quads of ONE gathered pixel, so a wave-level load touches 64*16/(4*CB) rows x (4*CB contiguous bytes) -- full 128-byte
// lines at CB = 32.  (The first version gave each lane its own row: 64 different lines per instruction, and the kernel
// ran at the L1 request rate, not the MFMA rate: time did not react to chunk depth, read scheduling or warp
// specialisation -- profiles/r01_notes.md.)  The LDS tile stays k-major for conflict-free MFMA operand reads; its row
// stride is padded so that the transposing stores of 4*CB/16 lanes per row spread over banks.
// ---- branch-free gather offsets (VEC == 4 loaders of k_conv_gemm / k_conv_gemm_sk) ---------------------------------------------------
// A slot's byte offset of tap (kh, kw) is SEPARABLE: row part (image, source row) + column part (source column, channel quad).  Both
// parts are tabulated per slot for the <= 3 values of kh and kw when a tile starts -- stride, zero / reflection padding, the transposed
// (backward-data) index map with its stride holes, the x2 upsample and the two sources of a concat layer all resolved there, once --
// so that moving to the next tap inside the K loop is two selects and one add per source, with no branches.  A part that reads a
// structural zero is a large constant (row: 2^31, column: 2^30): any sum containing one lies beyond num_records (operands stay below
// 2^30 bytes, checked by the host) and the buffer load returns 0.
// (The first form re-derived the coordinates of every tap in the loop: ~90 VALU + ~115 SALU instructions per chunk in a dozen
// exec-masked basic blocks, none of which the scheduler could move under the 16 MFMAs of the chunk -- the loop spent as long outside
// its matrix instructions as inside them: round-3 disassembly and phase stamps.)
#define TAP_ROW_ZERO 0x80000000u
#define TAP_COL_ZERO 0x40000000u

template <bool TRANSPOSED>
__device__ __forceinline__ int axis_src(const ConvArgs& a, int d, int k, int n_src) {      // source coordinate along one axis, -1: structural zero
    // straight-line selects on the (wave-uniform) layer parameters: written with early returns this was a dozen exec-masked blocks per
    // slot and ~600 of the ~1270 instructions a workgroup executed before its first load (round-3 disassembly + prologue stamps)
    if (!TRANSPOSED) {
        const int s = d * a.stride + k - a.pad;
        const int r = reflect1(s, n_src);
        const bool inside = (unsigned)s < (unsigned)n_src;
        return (a.pad_mode == 1) ? r : (inside ? s : -1);
    } else {
        const int t = d + a.off - k;
        const int s2 = (a.stride == 2) ? 1 : 0;
        const int tt = t >> s2;
        const bool ok = t >= 0 && (t & s2) == 0 && tt < n_src;
        return ok ? tt : -1;
    }
}

// a * b + c on the full-rate 24-bit multiplier (a, b < 2^24; hipcc turns `__umul24(a, b) + c` into the quarter-rate v_mad_u64_u32)
__device__ __forceinline__ unsigned umad24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// GEMM row n -> (image, row, column) of an (Hc x Wc) lattice without integer division: reciprocal multiplication in fp32 + exact fix-up
// (n < 2^24, checked by the host).  A 64-bit division per slot was a third of the kernels' prologue.
__device__ __forceinline__ void decode_row(int n, int hw, int Wc, float inv_hw, float inv_w, int& b, int& y, int& x) {
    b = (int)((float)n * inv_hw);
    b -= (b * hw > n) ? 1 : 0;
    b += ((b + 1) * hw <= n) ? 1 : 0;
    const int r = n - b * hw;
    y = (int)((float)r * inv_w);
    y -= (y * Wc > r) ? 1 : 0;
    y += ((y + 1) * Wc <= r) ? 1 : 0;
    x = r - y * Wc;
}

template <bool TRANSPOSED, int A_PER>
struct TapTable {
    unsigned r0[A_PER][3], c0[A_PER][3], r1[A_PER][3], c1[A_PER][3];

    // slot j gathers output-domain pixel (b, yd, xd) (live == false: a row past the end of the GEMM), channel quad `quad`.
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): row indices < 2^24 and row pitches in bytes < 2^24 (operands < 1 GB)
    __device__ __forceinline__ void init(const ConvArgs& a, int j, bool live, int b, int yd, int xd, int quad) {
        const int sh = a.up >> 1, Hl = a.Hs >> sh, Wl = a.Ws >> sh, C2 = a.Cin - a.C1;
        const unsigned pitch0 = (unsigned)(Wl * a.C1 * 4), pitch1 = (unsigned)(a.Ws * C2 * 4), px0 = (unsigned)(a.C1 * 4), px1 = (unsigned)(C2 * 4);
        const unsigned row0 = (unsigned)(b * Hl), row1 = (unsigned)(b * a.Hs), q16 = (unsigned)quad * 16u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ys = (live && k < a.KH) ? axis_src<TRANSPOSED>(a, yd, k, a.Hs) : -1;
            const int xs = (live && k < a.KW) ? axis_src<TRANSPOSED>(a, xd, k, a.Ws) : -1;
            r0[j][k] = ys >= 0 ? __umul24(row0 + (unsigned)(ys >> sh), pitch0) : TAP_ROW_ZERO;
            r1[j][k] = ys >= 0 ? __umul24(row1 + (unsigned)ys, pitch1) : TAP_ROW_ZERO;
            c0[j][k] = xs >= 0 ? __umul24((unsigned)(xs >> sh), px0) + q16 : TAP_COL_ZERO;
            c1[j][k] = xs >= 0 ? __umul24((unsigned)xs, px1) + q16 : TAP_COL_ZERO;
        }
    }
    // (selection by bit masks of the wave-uniform tap index: written as `k == 0 ? t[0] : ...` hipcc turns the tables into a scratch array
    // indexed at run time -- eight scratch loads per chunk)
    __device__ __forceinline__ void tap(int kh, int kw, bool two, unsigned (&off0)[A_PER], unsigned (&off1)[A_PER]) const {
        const unsigned h0 = kh == 0 ? ~0u : 0u, h1 = kh == 1 ? ~0u : 0u, h2 = kh >= 2 ? ~0u : 0u;
        const unsigned w0 = kw == 0 ? ~0u : 0u, w1 = kw == 1 ? ~0u : 0u, w2 = kw >= 2 ? ~0u : 0u;
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            off0[j] = ((r0[j][0] & h0) | (r0[j][1] & h1) | (r0[j][2] & h2)) + ((c0[j][0] & w0) | (c0[j][1] & w1) | (c0[j][2] & w2));
        if (two) {                                              // (wave-uniform) only a concat layer has a second source
#pragma unroll
            for (int j = 0; j < A_PER; ++j)
                off1[j] = ((r1[j][0] & h0) | (r1[j][1] & h1) | (r1[j][2] & h2)) + ((c1[j][0] & w0) | (c1[j][1] & w1) | (c1[j][2] & w2));
        }
    }
};

// XCD-aware workgroup order (MI355X: 8 XCDs, each with its own 4 MB L2; workgroups are dealt to them round-robin in launch order, so
// consecutive ids land on DIFFERENT L2s).  Tiles that are neighbours in the image re-read each other's input rows (3x3 taps) and all
// column tiles / K slices of one row tile read the same rows: this remap hands every XCD a CONTIGUOUS range of the launch's linear ids
// (id -> start(id % 8) + id / 8), so those re-reads hit that XCD's L2 instead of the fabric (measured with in-kernel stamps, round 3:
// a K chunk took 1.5 us per workgroup wherever fewer than three workgroups shared a CU -- the latency of ONE prefetched chunk of
// fabric-served loads -- against 0.43 us of MFMA work).  Speed only: nothing depends on where a workgroup actually runs.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, k = id & 7u, j = id >> 3;
    return k * q + (k < r ? k : r) + j;
}

// Static wave priority by launch order.  The workgroups that share a CU run the same program and, with the matrix pipe arbitrated fairly
// between them, fall into LOCKSTEP: all of them multiply at the same time (n x 1024 cycles per chunk, interleaved) and then all of them
// stage / synchronise / read LDS at the same time with the pipe idle (~1570 cycles) -- measured per chunk: 0.66 us + 0.43 us x n
// (scratch/conv_stamps.py).  Workgroups id, id + 256, id + 512 ... are the ones that end up on one CU (256 CUs, round-robin dispatch):
// giving them DIFFERENT priorities lets the highest one run its MFMA chain unimpeded while the others fill its staging phase, and the
// phases stay interleaved.  Speed only.
__device__ __forceinline__ void set_wave_priority(unsigned linear_block_id) {
    const unsigned p = (linear_block_id >> 8) & 3u;
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
}

#ifdef E2E_CONV_STAMPS          // diagnostic build only (scratch/conv_stamps.py): s_memtime / s_memrealtime stamps of every workgroup's phases
__device__ unsigned long long g_stamps[8192 * 8];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_stamps2[8192 * 4];      // finer stamps inside the prologue of k_conv_gemm
#define STAMP2(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8192) g_stamps2[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// shader-clock time spent in the phases of the K loop (wave 0 of every workgroup, summed over its chunks).  The scheduling barriers pin
// the phases, so this build runs a slightly different (more serial) schedule than the product: it locates the stalls, it is not the product's timing
__device__ unsigned long long g_phases[8192 * 8];
#define PHASE_DECL unsigned long long ph_last = __builtin_amdgcn_s_memtime(), ph_acc[6] = {0, 0, 0, 0, 0, 0}
#define PHASE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_last; ph_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PHASE_WRITE do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8192) for (int i_ = 0; i_ < 6; ++i_) g_phases[blockIdx.x * 8 + i_] = ph_acc[i_]; } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP2(i) do { } while (0)
#define PHASE_DECL do { } while (0)
#define PHASE(i) do { } while (0)
#define PHASE_WRITE do { } while (0)
#endif

template <int WM, int WN, int TM, int TN, int VEC, bool TRANSPOSED, int CB>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_gemm(ConvArgs a) {
    constexpr int BM = 32 * WM * TM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int KQ = CB / VEC;                           // k-groups (of VEC) per chunk
    constexpr int A_CNT = BM * KQ;                         // (row, k-group) elements of the A tile
    constexpr int A_PER = (A_CNT + NT - 1) / NT;
    constexpr int B_CNT = CB * (BN / 4);                   // float4 loads of the B tile
    constexpr int B_PER = (B_CNT + NT - 1) / NT;
    constexpr int APAD = (VEC == 4) ? 1 : 0;
    static_assert(NT % KQ == 0 && NT % BM == 0, "tile / thread-count mismatch");
    __shared__ float As[2][CB][BM + APAD];
    __shared__ float Bs[2][CB][BN];
    STAMP(0);
    set_wave_priority(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    // parity-class form (transposed gather, stride 2): this workgroup's class, its pixel sub-lattice and its tap subset
    const bool CLS = TRANSPOSED && VEC == 4 && a.cls != 0;
    const int py = CLS ? (int)((blockIdx.x >> 1) & 1) : 0, px = CLS ? (int)(blockIdx.x & 1) : 0;
    const int Hc = CLS ? (a.Hd - py + 1) / 2 : a.Hd, Wc = CLS ? (a.Wd - px + 1) / 2 : a.Wd;
    const int kh0 = CLS ? ((py + a.off) & 1) : 0, kw0 = CLS ? ((px + a.off) & 1) : 0, kstep = CLS ? 2 : 1;
    const int nkh = CLS ? (a.KH - kh0 + 1) / 2 : a.KH, nkw = CLS ? (a.KW - kw0 + 1) / 2 : a.KW;
    const int64_t Ntot = (int64_t)a.B * Hc * Wc;
    if (Ntot > 0) STAMP2(0);                                 // (diagnostic) the kernel arguments have arrived
    // non-class form: the launch's linear id -> XCD-contiguous order, decoded with the column tile fastest, then the row tile, the K slice
    // slowest: an XCD's contiguous range then covers one or two K slices, i.e. that part of the WEIGHT matrix only -- with the row tile
    // slowest (the first form) every XCD's L2 fetched the whole matrix, eight times per launch in total (the bulk of FETCH_SIZE on the
    // deep layers, profiles/r03_gemm_k_order_and_fetch.txt); the column tiles of a row tile, which share its input rows, stay neighbours
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (!CLS) {
        const unsigned nxy = gridDim.x * gridDim.y;
        const unsigned lin = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nxy * gridDim.z);
        bz = lin / nxy;
        const unsigned rem = lin - bz * nxy;
        bx = rem / gridDim.y;
        by = rem - bx * gridDim.y;
    }
    const int64_t n0 = (int64_t)(CLS ? (bx >> 2) : bx) * BM;
    if (CLS && (n0 >= Ntot || nkh <= 0 || nkw <= 0)) return;        // classes are sized by the largest one; empty tap sets write nothing (output pre-zeroed by the host for KH < 2)
    const int c0 = by * BN;
    const int K = CLS ? nkh * nkw * a.Cin : a.KH * a.KW * a.Cin;
    const int nchunks_all = (K + CB - 1) / CB;
    const int cbeg = (a.ksplit > 1) ? bz * a.cps : 0;
    const int cend = (a.ksplit > 1) ? min(nchunks_all, cbeg + a.cps) : nchunks_all;
    if (CLS && a.ksplit > 1 && cbeg >= nchunks_all) return;         // class form split by TAP: this class has fewer taps than slices (the
                                                                    // class epilogue only adds the slices that exist)

    // ---- per-thread A slots: slot j is element tid + j*NT of the tile -----------------------------------------------
    //   VEC == 4: element e -> row e / KQ, k-quad e % KQ (the quad is the same for all slots since NT % KQ == 0)
    //   VEC == 1: element e -> row e % BM, k e / BM      (conv1: 3 channels, lanes walk along the image row)
    int arow[A_PER], ab[A_PER], ayd[A_PER], axd[A_PER];
    bool arow_ok[A_PER];
    const int akq = (VEC == 4) ? tid % KQ : tid / BM;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int e = tid + j * NT;
        arow[j] = (VEC == 4) ? e / KQ : e % BM;
        const int64_t an = n0 + arow[j];
        arow_ok[j] = e < A_CNT && an < Ntot;
        ab[j] = ayd[j] = axd[j] = 0;
        if (arow_ok[j]) {
            const int hw = Hc * Wc;
            if (Ntot < (1 << 24)) {
                decode_row((int)an, hw, Wc, 1.0f / (float)hw, 1.0f / (float)Wc, ab[j], ayd[j], axd[j]);
            } else {
                ab[j] = (int)(an / hw);
                const int r = (int)(an - (int64_t)ab[j] * hw);
                ayd[j] = r / Wc;
                axd[j] = r - ayd[j] * Wc;
            }
            if (CLS) { ayd[j] = 2 * ayd[j] + py; axd[j] = 2 * axd[j] + px; }
        }
    }
    const int sh = a.up >> 1;                                // up is 1 or 2: source coordinate = full-res coordinate >> sh
    const int Hl = a.Hs >> sh, Wl = a.Ws >> sh, C2 = a.Cin - a.C1;
    if (ab[0] >= 0) STAMP2(1);                               // (diagnostic) rows decoded

    // staging registers: TWO sets -- the global loads of chunk c + 2 are issued while chunk c is multiplied and chunk c + 1 waits in the
    // other set for its turn to be written to LDS.  (One set = loads one chunk ahead left every workgroup waiting ~1.5 us per chunk for
    // fabric-served loads against 0.43 us of MFMA work: round-3 phase stamps, scratch/conv_stamps.py.)
    f4v areg[2][A_PER];
    float areg1[2][A_PER];
    f4v breg[2][B_PER];

    // VEC == 4.  Everything the gather needs per chunk is ONE select and ONE buffer load per slot:
    //  * operands are addressed through buffer resources with 32-bit byte offsets (tensors < 2 GB, checked by the
    //    host): structural zeros (padding, stride holes, rows past the end) get an offset beyond num_records and the
    //    hardware returns 0 -- no branches, no 64-bit address arithmetic in the loop;
    //  * the K loop is walked tap-major; a slot's pixel offsets change only when the tap changes (every Cin/CB
    //    chunks) and cost ~25 VALU then; the channel offset of the chunk is wave-uniform and rides in soffset.
    // (The first version re-derived 64-bit pointers with integer divisions per tap: 1870 non-MFMA VALU instructions
    // per wave -- and VALU work does not hide under MFMAs: ablation showed T = T_valu + T_mfma, 23 + 27 us on layer1.)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, (int)(a.src1 ? a.bytes1 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.bytesw, 0x00020000);
    const int cpt = (VEC == 4) ? a.Cin / CB : 1;           // chunks per tap
    int ld_kh = 0, ld_kw = 0, ld_cc = 0;                     // state of the LOAD stream (runs two chunks ahead)
    unsigned off0[A_PER], off1[A_PER];                       // byte offsets of (tap pixel, this thread's quad) in src0 / src1
    TapTable<TRANSPOSED, A_PER> taps;                        // (unused by the scalar VEC == 1 loader)
    if (VEC == 4) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) taps.init(a, j, arow_ok[j], ab[j], ayd[j], axd[j], akq);
    }
    const bool two_src = a.src1 != nullptr;
    auto set_tap = [&]() { if (VEC == 4) taps.tap(ld_kh, ld_kw, two_src, off0, off1); };
    // (K order: tap-major -- all channel slices of tap 0, then tap 1, ...  Walking the taps fastest instead, so that the KH x KW re-reads
    // of a channel slice follow each other while it is still in L1 / L2, changed neither the time per layer nor the L2-fabric traffic
    // (43.6 -> 46.2 MB per 32x128 launch): what those launches fetch is the WEIGHT matrix, once per XCD L2 -- eight times per launch --
    // not the taps.  profiles/r03_gemm_k_order_and_fetch.txt.)
    // The vector pipe does not run under the matrix pipe of its SIMD for free: a round of three workgroups takes about its MFMA cycles PLUS
    // its VALU cycles (3 x (16 x 64 + ~100 x 4) cycles = 1.8 us against 1.9 us measured, round-3 stamps; round 1's ablation had said
    // T = T_valu + T_mfma).  The ~36 selects of a tap change are therefore executed only when the tap changes -- a wave-uniform branch,
    // every Cin / CB chunks -- and the second source's half only for concat layers.  (Taps past the last one of a slice are read but
    // never multiplied.)
    auto advance_tap = [&]() {
        const bool w1 = ++ld_cc == cpt;
        ld_cc = w1 ? 0 : ld_cc;
        ld_kw += w1 ? kstep : 0;
        const bool w2 = ld_kw >= a.KW;
        ld_kw = w2 ? kw0 : ld_kw;
        ld_kh += w2 ? kstep : 0;
        if (w1) set_tap();
    };
    unsigned boff[B_PER];                                    // B tile: fixed per-thread offset, the chunk rides in soffset
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int idx = tid + j * NT;                       // over CB x (BN/4)
        const int kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
        const int col = c0 + cq * 4;
        boff[j] = (idx < B_CNT && col < a.ldw) ? (unsigned)(kr * a.ldw + col) * 4u : OOB;     // ldw % 4 == 0, zero-padded columns
    }
    if (VEC == 4) {                                          // position the load stream on chunk `cbeg`
        const int t0 = cbeg / cpt;
        ld_kh = kh0 + kstep * (t0 / nkw);
        ld_kw = kw0 + kstep * (t0 - (t0 / nkw) * nkw);
        ld_cc = cbeg - t0 * cpt;
        set_tap();
    }
    if (off0[0] != 1u) STAMP2(2);                            // (diagnostic) tap tables built, first tap selected

    auto load_chunk = [&](int chunk, int set) {
        const int kbase = chunk * CB;
        if (VEC == 4) {
            const int cbase = ld_cc * CB;                    // wave-uniform: the whole chunk lies on one side of the concat split
            const int krow = (ld_kh * a.KW + ld_kw) * a.Cin + cbase;      // first weight row of this chunk (== kbase outside the class form)
            const bool use0 = cbase < a.C1;
            const __amdgpu_buffer_rsrc_t rs = use0 ? rs0 : rs1;
            const int soff = (use0 ? cbase : cbase - a.C1) * 4;
#pragma unroll
            for (int j = 0; j < A_PER; ++j)
                areg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs, use0 ? off0[j] : off1[j], soff, 0));
            advance_tap();
#pragma unroll
            for (int j = 0; j < B_PER; ++j)
                breg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsw, boff[j], krow * a.ldw * 4, 0));
        } else {
#pragma unroll
            for (int j = 0; j < A_PER; ++j) {
                const int kq = akq + j * (NT / BM);
                const int k = kbase + kq;
                float v = 0.f;
                if (arow_ok[j] && kq < KQ && k < K) {
                    const int tap = k / a.Cin, ci = k - tap * a.Cin;
                    const int kh = tap / a.KW, kw = tap - kh * a.KW;
                    int ys, xs;
                    if (tap_coord<TRANSPOSED>(a, ayd[j], axd[j], kh, kw, ys, xs))
                        v = (a.src0[(((int64_t)ab[j] * a.Hs + ys) * a.Ws + xs) * a.Cin + ci] - a.in_sub) * a.in_mul;
                }
                areg1[set][j] = v;
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                const int kr = (tid + j * NT) / (BN / 4);
                breg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsw, (kbase + kr < K) ? boff[j] : OOB, kbase * a.ldw * 4, 0));
            }
        }
    };
    auto store_chunk = [&](int buf, int set) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            if (VEC == 4) {
                if (A_CNT % NT != 0 && tid + j * NT >= A_CNT) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][akq * 4 + e][arow[j]] = areg[set][j][e];
            } else {
                const int kq = akq + j * (NT / BM);
                if (kq < KQ) As[buf][kq][arow[j]] = areg1[set][j];
            }
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + j * NT;
            if (B_CNT % NT != 0 && idx >= B_CNT) continue;
            const int kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
            *(f4v*)&Bs[buf][kr][cq * 4] = breg[set][j];
        }
    };

    f16v acc[TM][TN];       // TM x TN accumulators of 32x32 per wave: operand fragments are reused TN resp. TM times
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;

    if (cbeg < cend) {
        load_chunk(cbeg, 0);
        STAMP(1);
        if (VEC == 4 || cbeg + 1 < cend) load_chunk(cbeg + 1, 1);
        store_chunk(0, 0);
    }
    __syncthreads();
    STAMP(2);
    const int arow_l = wm * TM * 32 + (lane & 31), khalf = lane >> 5;
    PHASE_DECL;
    // one K chunk: chunk c (in LDS buffer `BUF`) is multiplied while the loads of chunk c + 2 go to register set BUF (free: its chunk c
    // is in LDS) and chunk c + 1 -- loaded one iteration earlier into set BUF ^ 1 -- is written to the other LDS buffer afterwards
    auto chunk_step = [&](int c, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        // VEC == 4: loads and stores are UNCONDITIONAL (chunks past the end of the slice are fetched -- out-of-range offsets return zeros --
        // and written to the LDS buffer nobody reads any more): the chunk is one basic block and the scheduler may place the staging
        // instructions between the MFMAs
        PHASE(5);                                            // (diagnostic) loop control between two chunks
        if (VEC == 4 || c + 2 < cend) load_chunk(c + 2, buf);
        PHASE(0);                                            // (diagnostic) loads of chunk c + 2 issued, next tap selected
        // operand fragments of the WHOLE chunk are read into registers first, the MFMAs follow with counted LDS waits: hipcc's own
        // schedule of the fused loop was read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs per k-pair on ONE register set, i.e. every pair of MFMAs
        // waited for a fresh LDS round trip (~190 cycles per 128 cycles of matrix work: the kernels ran at half the MFMA rate with
        // the pipe idle, round-3 disassembly)
        constexpr int KH2 = (TM * TN == 1) ? CB / 2 : CB / 4;          // k-steps per register batch (2 x 2 blocks: two batches per chunk)
#pragma unroll
        for (int k0 = 0; k0 < CB / 2; k0 += KH2) {
            float av[KH2][TM], bv[KH2][TN];
#pragma unroll
            for (int kk = 0; kk < KH2; ++kk) {
#pragma unroll
                for (int u = 0; u < TM; ++u) av[kk][u] = As[buf][(k0 + kk) * 2 + khalf][arow_l + u * 32];
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[kk][t] = Bs[buf][(k0 + kk) * 2 + khalf][(wn * TN + t) * 32 + (lane & 31)];
            }
#pragma unroll
            for (int kk = 0; kk < KH2; ++kk)
#pragma unroll
                for (int u = 0; u < TM; ++u)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][u], bv[kk][t], acc[u][t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, KH2 * (TM + TN) / 2, 0);     // all LDS reads of the batch (ds_read2: two values each) ...
            __builtin_amdgcn_sched_group_barrier(0x008, KH2 * TM * TN, 0);           // ... ahead of its MFMAs
        }
        PHASE(1);                                            // (diagnostic) fragments read, all MFMAs of the chunk ISSUED
        // nothing moves across this point: left to itself hipcc hoists the LDS stores of chunk c + 1 (and with them the s_waitcnt vmcnt
        // for its global loads, issued ONE step earlier) to the top of the step, ahead of 14 of the 16 MFMAs -- every step then waited
        // for memory with the matrix pipe idle (round-3 disassembly: 1.14 us per chunk for a workgroup alone on its SIMDs, 0.43 us of
        // it MFMA work).  Here the wait comes after the MFMAs have been issued: two steps after the loads.
        __builtin_amdgcn_sched_barrier(0);
        if (VEC == 4 || c + 1 < cend) store_chunk(buf ^ 1, buf ^ 1);
        PHASE(2);                                            // (diagnostic) chunk c + 1 has arrived from memory and is on its way to LDS
        __syncthreads();
        PHASE(3);                                            // (diagnostic) barrier passed
        if (c == cbeg) STAMP(4);
    };
    // ---- software-pipelined form (every VEC == 4 single-accumulator instantiation, i.e. every product launch) ----------------------
    // Step c of the form above is: read chunk c's fragments -> 16 dependent MFMAs -> store chunk c + 1 -> barrier; the matrix pipe idles
    // from the last MFMA of a step until the first fragments of the next chunk have come back from LDS (store, barrier, LDS round trip:
    // 330 - 500 of ~1500 cycles per step, round-3 phase clocks in scratch/conv_stamps.py).  Here the barrier sits in the MIDDLE of the step
    // and the fragments of chunk c + 1 are read under the second half of chunk c's MFMAs into a second fragment register set:
    //   store chunk c + 1 (loaded one step ago) | MFMAs 0-7 of chunk c | barrier | read fragments of chunk c + 1 | MFMAs 8-15 of chunk c
    // so the first MFMA of step c + 1 finds its operands in registers.  LDS stays double-buffered: buffer (c + 1) & 1 held chunk c - 1,
    // whose fragments every wave had read before it reached the barrier of step c - 1.
    constexpr bool PIPE = VEC == 4 && TM * TN == 1;
    float fa[2][CB / 2], fb[2][CB / 2];
    auto read_frags = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
#pragma unroll
        for (int kk = 0; kk < CB / 2; ++kk) {
            fa[buf][kk] = As[buf][kk * 2 + khalf][arow_l];
            fb[buf][kk] = Bs[buf][kk * 2 + khalf][wn * 32 + (lane & 31)];
        }
    };
    auto pipe_step = [&](int c, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        // (the scheduling barriers pin the three groups in this order; inside a group hipcc interleaves freely.  Unpinned it moved
        // fifteen of the sixteen MFMAs behind the barrier and re-read every fragment pair right before its use)
        load_chunk(c + 2, buf);                              // register set `buf` is free: its chunk c went to LDS a step ago
        store_chunk(buf ^ 1, buf ^ 1);                       // chunk c + 1, loaded one step ago
#pragma unroll
        for (int kk = 0; kk < CB / 4; ++kk) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][kk], fb[buf][kk], acc[0][0], 0, 0, 0);
        // the staging work of the step -- the scalar bookkeeping of the weight-tile loads, the loads, the LDS stores -- is dealt out BETWEEN
        // the MFMAs of the first half: hipcc's own order issued most of it ahead of the first MFMA (round-4 disassembly; +1.5 % steps/s,
        // profiles/r04_gemm_interleave_ab.txt).  Issuing the first two or four MFMAs ahead of load_chunk as well -- its tap bookkeeping ends in
        // a wave-uniform branch, i.e. in basic blocks of its own that no MFMA can be scheduled into -- did not pay (205.2 / 202.4 against 205.8).
#pragma unroll
        for (int g = 0; g < CB / 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(std::integral_constant<int, buf ^ 1>{});
#pragma unroll
        for (int kk = CB / 4; kk < CB / 2; ++kk) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][kk], fb[buf][kk], acc[0][0], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < CB / 4; ++g) {                   // one MFMA, then two LDS reads (ds_read2: four fragment values), eight times
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == cbeg) STAMP(4);
    };
    if (PIPE) {
        if (cbeg < cend) {
            read_frags(std::integral_constant<int, 0>{});
            int c = cbeg;
            for (; c + 1 < cend; c += 2) {
                pipe_step(c, std::integral_constant<int, 0>{});
                pipe_step(c + 1, std::integral_constant<int, 1>{});
            }
            if (c < cend) pipe_step(c, std::integral_constant<int, 0>{});
        }
    } else
    // pairs of steps, the odd last one OUTSIDE the loop: with `if (c + 1 < cend)` around the second step inside it, the control-flow graph
    // has an edge from the end of the first step back to the loop header, on which the four most recent loads are the ones the first step
    // itself issued into the registers it reads its fragments into next -- hipcc's s_waitcnt insertion must cover that (impossible) path and
    // put `s_waitcnt vmcnt(0)` at the top of EVERY first step: all prefetched loads drained before any MFMA (round-3 disassembly)
    {
        int c = cbeg;
        for (; c + 1 < cend; c += 2) {
            chunk_step(c, std::integral_constant<int, 0>{});
            chunk_step(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < cend) chunk_step(c, std::integral_constant<int, 0>{});
    }
    PHASE_WRITE;
    STAMP(3);

    // ---- epilogue (epilogue_block above; split-K slices store raw partial sums instead) --------------------------------------
    if (a.ksplit > 1) {                                      // raw partial sums; the scale / shift / activation run after the reduction
        // plain form: slab z = [Ntot][Ncols]; class form: slab (z, class) = [cls_rows][Ncols] rows of the class lattice
        const int64_t slab_id = CLS ? ((int64_t)bz * 4 + (py * 2 + px)) * a.cls_rows : (int64_t)bz * Ntot;
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.slab + slab_id * a.Ncols), 0,
                                                                             (int)(Ntot * a.Ncols * 4), 0x00020000);
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int col = c0 + (wn * TN + t) * 32 + (lane & 31);
                const int64_t nb = n0 + (wm * TM + u) * 32 + 4 * khalf;
                const unsigned voff = (col < a.Ncols && nb < Ntot) ? (unsigned)((nb * a.Ncols + col) * 4) : OOB;
                const int left = (int)((Ntot - nb < 32) ? Ntot - nb : 32);      // rows of this block that exist (the range check
                float v[16];                                                    // ignores the scalar offset: mask them in the VGPR)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = acc[u][t][r];
                    asm volatile("" : "+v"(v[r]));          // hipcc (ROCm 7.2) otherwise stores accumulator register 0 sixteen times
                }                                           // when the store data comes straight out of the AGPR tuple
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), rsl, ((r & 3) + 8 * (r >> 2) < left) ? voff : OOB,
                                                          ((r & 3) + 8 * (r >> 2)) * a.Ncols * 4, 0);
            }
        return;
    }
    if (!CLS) {
        const EpiRsrc er = make_epi_rsrc<TRANSPOSED>(a);
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t)
                epilogue_block<TRANSPOSED>(a, er, acc[u][t], c0 + (wn * TN + t) * 32 + (lane & 31), n0 + (wm * TM + u) * 32 + 4 * khalf, Ntot);
#ifdef E2E_CONV_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(5);
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8192) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_stamps[blockIdx.x * 8 + 6] = hw;
            g_stamps[blockIdx.x * 8 + 7] = xcc;
        }
#endif
        return;
    }
    // class form (3 small layers): GEMM row n = (b, yc, xc) of the class lattice -> output pixel (b, 2 yc + py, 2 xc + px), decoded like
    // the loader's rows (reciprocal multiplication + exact fix-up; the host keeps class lattices below 2^24 rows)
    const int hw_c = Hc * Wc;
    const float inv_hw_c = 1.0f / (float)hw_c, inv_w_c = 1.0f / (float)Wc;
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = c0 + (wn * TN + t) * 32 + (lane & 31);
            if (col >= a.Ncols) continue;
            const float sc = a.scale ? a.scale[col] : 1.f, sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t n = n0 + (wm * TM + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (n >= Ntot) continue;
                int b, yc, xc;
                decode_row((int)n, hw_c, Wc, inv_hw_c, inv_w_c, b, yc, xc);
                const int64_t orow = ((int64_t)b * a.Hd + 2 * yc + py) * a.Wd + 2 * xc + px;
                float v = fmaf(acc[u][t][r], sc, sh);
                if (a.pre) v += a.pre[orow * a.Ncols + col];
                if (a.dact) v *= act_deriv(a.xin[orow * a.Ncols + col], a.dact);
                if (a.res) v += a.res[orow * a.Ncols + col];
                a.out[orow * a.Ncols + col] = apply_act(v, a.act);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stream-K form of the same implicit GEMM (64x64 tiles, 2x2 waves, 16-byte channel-quad loader, forward and plain backward-data).
//
// Why: a conv layer of this network is ONE launch of 80 ... 2400 tiles x 18 ... 144 K-chunks on 256 CUs.  Tile-per-workgroup
// grids lose to quantisation (600 tiles = 2.34 per SIMD cost what 3 cost) and, on the deep layers, need split-K slabs plus a
// separate reduction launch (11 % of the GEMM family's time in round 2), and every workgroup pays its own prologue for one short
// tile.  Here G persistent workgroups (2-3 per CU, all resident) share the flattened iteration space (tile, chunk) in equal
// contiguous ranges: workgroup g owns iterations [I g / G, I (g + 1) / G).  A range that starts inside a tile begins with that
// tile's REMAINDER: the workgroup stores the partial accumulator to its slab (write-through stores) and raises its flag.  A range
// that ends inside a tile ends with that tile's HEAD (chunk 0 on): that workgroup is the tile's finisher -- it adds the slabs of
// the workgroups g + 1, g + 2, ... that hold the rest of the tile, in that order (= increasing K: a fixed summation order,
// bitwise reproducible), and runs the fused epilogue.  Everything else is whole tiles with the direct epilogue.
//   * no deadlock: a finisher waits only for pieces that their owners process FIRST in their own ranges, before any wait of their
//     own; all G workgroups are resident (G <= 4 per CU by LDS and registers), and even a late-scheduled owner is only waited for
//     by workgroups that do not block its scheduling (everyone else runs to completion).  Spins are bounded; a time-out raises the
//     error word flags[G] (read by the host through e2e_conv_streamk_error) instead of hanging the GPU;
//   * hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): producer = sc1 (write-through) slab stores -> every wave
//     s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane's relaxed agent-scope flag store; consumer = one lane polls the flag
//     (relaxed, agent scope) -> agent-scope acquire -> s_waitcnt vmcnt(0) -> workgroup barrier -> sc1 loads of the slab;
//   * flags are zero outside a launch: the finisher clears each flag it consumed (every flag has exactly one consumer), the
//     owner of the workspace zeroes the flag region ONCE when it allocates it.
// ---------------------------------------------------------------------------------------------------------------------
struct SkArgs {
    float* slabs;          // [G][64 * 64]
    unsigned* flags;       // [G] + error word
    int G, tiles_n, C;     // workgroups, column tiles, K chunks per tile
    long long I;           // tiles * C
};
#define SK_MAX_G 768        // most persistent workgroups of a stream-K launch (3 per CU)
#define SK_FLAG_FLOATS 1024 // head of every convolution workspace: stream-K flags [SK_MAX_G] + error word, zeroed ONCE by the owner
#define SK_ERR_INDEX SK_MAX_G
#define SK_SC1 16           // aux bit of the buffer instructions: sc1 (system-coherent level 1 = write-through / L1 bypass on gfx950)

template <bool TRANSPOSED, int CB>
__global__ __launch_bounds__(256, 2) void k_conv_gemm_sk(ConvArgs a, SkArgs s) {
    constexpr int BM = 64, BN = 64, NT = 256, KQ = CB / 4;
    constexpr int A_PER = BM * KQ / NT, B_PER = CB * (BN / 4) / NT;          // CB = 32: 2 + 2 loads of 16 B per thread and chunk; 16: 1 + 1
    static_assert(BM * KQ % NT == 0 && CB * (BN / 4) % NT == 0, "tile / thread-count mismatch");
    __shared__ float As[2][CB][BM + 1];
    __shared__ float Bs[2][CB][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    set_wave_priority(blockIdx.x);
    STAMP(0);
    int stamp_piece = 0;
    (void)stamp_piece;
    const int g = (int)xcd_contiguous(blockIdx.x, gridDim.x);          // consecutive iteration ranges (neighbouring tiles) share an XCD's L2
    const int64_t Ntot = (int64_t)a.B * a.Hd * a.Wd;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)a.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, (int)(a.src1 ? a.bytes1 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.bytesw, 0x00020000);
    const int sh = a.up >> 1, Hl = a.Hs >> sh, Wl = a.Ws >> sh, C2 = a.Cin - a.C1, cpt = a.Cin / CB, hw = a.Hd * a.Wd;
    const int akq = tid % KQ, arow_l = wm * 32 + (lane & 31), khalf = lane >> 5;
    const float inv_hw = 1.0f / (float)hw, inv_wd = 1.0f / (float)a.Wd;
    int64_t it = s.I * g / s.G;
    const int64_t it_end = s.I * (g + 1) / s.G;
    while (it < it_end) {
        const int tile = (int)(it / s.C), cb = (int)(it - (int64_t)tile * s.C);
        const int ce = (int)((it_end - it < s.C - cb) ? cb + (it_end - it) : s.C);
        const int tm = tile / s.tiles_n, tn = tile - tm * s.tiles_n;
        const int64_t n0 = (int64_t)tm * BM;
        const int c0 = tn * BN;
        // ---- this tile's gather slots (same mapping as k_conv_gemm, VEC == 4) --------------------------------------------------
        int arow[A_PER], ab[A_PER], ayd[A_PER], axd[A_PER];
        bool arow_ok[A_PER];
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            arow[j] = (tid + j * NT) / KQ;
            const int64_t an = n0 + arow[j];
            arow_ok[j] = an < Ntot;
            ab[j] = ayd[j] = axd[j] = 0;
            if (arow_ok[j]) {
                if (Ntot < (1 << 24)) {
                    decode_row((int)an, hw, a.Wd, inv_hw, inv_wd, ab[j], ayd[j], axd[j]);
                } else {
                    ab[j] = (int)(an / hw);
                    const int r = (int)(an - (int64_t)ab[j] * hw);
                    ayd[j] = r / a.Wd;
                    axd[j] = r - ayd[j] * a.Wd;
                }
            }
        }
        unsigned boff[B_PER];
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + j * NT, kr = idx / (BN / 4), cq = idx - kr * (BN / 4), col = c0 + cq * 4;
            boff[j] = (col < a.ldw) ? (unsigned)(kr * a.ldw + col) * 4u : OOB;
        }
        int ld_kh, ld_kw, ld_cc;
        unsigned off0[A_PER], off1[A_PER];
        TapTable<TRANSPOSED, A_PER> taps;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) taps.init(a, j, arow_ok[j], ab[j], ayd[j], axd[j], akq);
        {
            const int t0 = cb / cpt;
            ld_kh = t0 / a.KW;
            ld_kw = t0 - ld_kh * a.KW;
            ld_cc = cb - t0 * cpt;
            taps.tap(ld_kh, ld_kw, a.src1 != nullptr, off0, off1);
        }
        f4v areg[2][A_PER], breg[2][B_PER];                  // two staging sets: loads run two chunks ahead (see k_conv_gemm)
        auto load_chunk = [&](int set) {
            const int cbase = ld_cc * CB;
            const int krow = (ld_kh * a.KW + ld_kw) * a.Cin + cbase;
            const bool use0 = cbase < a.C1;
            const __amdgpu_buffer_rsrc_t rs = use0 ? rs0 : rs1;
            const int soff = (use0 ? cbase : cbase - a.C1) * 4;
#pragma unroll
            for (int j = 0; j < A_PER; ++j)
                areg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs, use0 ? off0[j] : off1[j], soff, 0));
            {                                                   // next chunk's tap: selects on wave-uniform values, no branches
                const bool w1 = ++ld_cc == cpt;
                ld_cc = w1 ? 0 : ld_cc;
                ld_kw += w1 ? 1 : 0;
                const bool w2 = ld_kw >= a.KW;
                ld_kw = w2 ? 0 : ld_kw;
                ld_kh += w2 ? 1 : 0;
                taps.tap(ld_kh, ld_kw, a.src1 != nullptr, off0, off1);
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j)
                breg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsw, boff[j], krow * a.ldw * 4, 0));
        };
        auto store_chunk = [&](int buf, int set) {
#pragma unroll
            for (int j = 0; j < A_PER; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][akq * 4 + e][arow[j]] = areg[set][j][e];
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                const int idx = tid + j * NT, kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
                *(f4v*)&Bs[buf][kr][cq * 4] = breg[set][j];
            }
        };
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        load_chunk(0);
        load_chunk(1);                                          // (unconditional, like every load / store of the loop: see k_conv_gemm)
        store_chunk(0, 0);
        __syncthreads();
        auto chunk_step = [&](int c, auto buf_c) {
            constexpr int buf = decltype(buf_c)::value;
            load_chunk(buf);
            float av[CB / 2], bv[CB / 2];                       // the chunk's fragments first, then the MFMAs (see k_conv_gemm)
#pragma unroll
            for (int kk = 0; kk < CB / 2; ++kk) {
                av[kk] = As[buf][kk * 2 + khalf][arow_l];
                bv[kk] = Bs[buf][kk * 2 + khalf][wn * 32 + (lane & 31)];
            }
#pragma unroll
            for (int kk = 0; kk < CB / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, CB / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CB / 2, 0);
            __builtin_amdgcn_sched_barrier(0);                  // the LDS stores (and their vmcnt wait) stay behind the MFMAs: see k_conv_gemm
            store_chunk(buf ^ 1, buf ^ 1);
            __syncthreads();
        };
        if (stamp_piece == 0) STAMP(1);
        {                                                       // (pairs + an odd tail outside the loop: see k_conv_gemm)
            int c = cb;
            for (; c + 1 < ce; c += 2) {
                chunk_step(c, std::integral_constant<int, 0>{});
                chunk_step(c + 1, std::integral_constant<int, 1>{});
            }
            if (c < ce) chunk_step(c, std::integral_constant<int, 0>{});
        }
        if (stamp_piece == 0) STAMP(2);
        // ---- what to do with the accumulator ------------------------------------------------------------------------------------
        const unsigned sl_off = (unsigned)((wave * 4 * 64 + lane) * 16);         // this lane's first 16-byte quad inside a slab
        if (cb != 0) {
            // the tile's remainder: partial sums to this workgroup's slab, then the flag
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(s.slabs + (int64_t)g * (BM * BN)), 0, BM * BN * 4, 0x00020000);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[r];
                asm volatile("" : "+v"(v[r]));              // (see k_conv_gemm's slab store)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, t), rsl, sl_off + q * 64 * 16, 0, SK_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&s.flags[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ce != s.C) {
                // the tile's head: this workgroup finishes the tile with the pieces of g + 1, g + 2, ... (increasing K)
                STAMP(4);
                const int64_t tile_end = (int64_t)(tile + 1) * s.C;
                int h_last = g;
                for (int h = g + 1; h < s.G && s.I * h / s.G < tile_end; ++h) h_last = h;
                if (tid == 0) {
                    for (int h = g + 1; h <= h_last; ++h) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(&s.flags[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1u << 24)) { __hip_atomic_store(&s.flags[SK_ERR_INDEX], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                STAMP(5);
                for (int h = g + 1; h <= h_last; ++h) {
                    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(s.slabs + (int64_t)h * (BM * BN)), 0, BM * BN * 4, 0x00020000);
                    f4v t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsl, sl_off + q * 64 * 16, 0, SK_SC1));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * q + e] += t[q][e];
                }
                if (tid == 0)
                    for (int h = g + 1; h <= h_last; ++h) __hip_atomic_store(&s.flags[h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const EpiRsrc er = make_epi_rsrc<TRANSPOSED>(a);
            epilogue_block<TRANSPOSED>(a, er, acc, c0 + wn * 32 + (lane & 31), n0 + wm * 32 + 4 * khalf, Ntot);
        }
        if (stamp_piece == 0) STAMP(3);
        ++stamp_piece;
        it += ce - cb;
    }
#ifdef E2E_CONV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    if (threadIdx.x == 0) g_stamps[blockIdx.x * 8 + 7] = (unsigned long long)stamp_piece;
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 convolutions with 16 or 32 input and 16 output channels at (almost) full resolution: the last decoder level
// (networks.py:262-266: upconv(0,0) 32 -> 16, upconv(0,1) 16 -> 16 behind the nearest x2 upsample) and upconv(0,1)'s backward-data.
// As implicit GEMMs they are 16 columns wide (half of every 32x32x2 MFMA is padding) and read each input pixel nine times through
// L2 (33-35 TF/s, 85 us for 2.8 GFLOP).  Here a workgroup stages the input PATCH of a TH x 64 output tile once in LDS, channel-major
// (upsample, reflection / zero padding resolved while staging), keeps all 9 x CIN x 16 weights in registers as MFMA B operands and
// runs v_mfma_f32_16x16x4_f32 with M = 16 consecutive output pixels of a row, N = the 16 output channels, K = 4 input channels of
// one tap: one conflict-free ds_read_b32 per MFMA, no im2col traffic.
//   domain position sampled by output pixel (y, x) and tap (kh', kw') = (y - org + kh', x - org + kw'); forward: org = 1 (pad);
//   backward-data on the padded grid: org = 2 and the taps flipped (dXp[q] = sum_t dZ[q - t] Wb[t]).
// ---------------------------------------------------------------------------------------------------------------------
struct ThinArgs {
    const float* src;      // (B, Hsrc, Wsrc, CIN) NHWC; the sampled domain is (Hsrc * UP) x (Wsrc * UP)
    const float* w;        // [(tap, k)][ldw]: k = input channel, columns = output channels
    const float* bias;     // per output channel or NULL
    float* out;            // (B, Hd, Wd, 16)
    int B, Hsrc, Wsrc, Hd, Wd, ldw, act, org, flip;
};

template <int CIN, int UP, bool REFLECT, int TH>
__global__ __launch_bounds__(256) void k_conv3x3_thin(ThinArgs a) {
    constexpr int TW = 64, PH = TH + 2, PW = TW + 2, PWS = 72, PLANE = PH * PWS, KQ = CIN / 4;
    static_assert(PLANE % 32 == 16, "the two channel planes a half-wave reads must sit 16 banks apart");
    __shared__ float patch[CIN * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int HD = a.Hsrc * UP, WD = a.Wsrc * UP;
    // ---- weights -> registers: B operand of tap t, channel group j: lane (kq, l16) holds W[t][4 j + kq][l16] ------------------
    float wreg[9][KQ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < KQ; ++j) wreg[t][j] = a.w[(int64_t)((a.flip ? 8 - t : t) * CIN + 4 * j + kq) * a.ldw + l16];
    // ---- stage the patch: one 16-byte channel quad per load --------------------------------------------------------------------
    const float* sb = a.src + (int64_t)b * a.Hsrc * a.Wsrc * CIN;
    for (int idx = tid; idx < PH * PW * KQ; idx += 256) {
        // thread -> (pixel, channel quad): quad-fastest gives coalesced global loads at the price of KQ-way conflicts on the transposing
        // LDS stores -- measured faster without the upsample (upconv(0,0) forward 36.0 -> 31.6 us, upconv(0,1) backward-data 48.1 -> 44.6);
        // behind the x2 upsample two neighbouring pixels share a source pixel and pixel-fastest wins (43.2 vs 44.8 us)
        const int q = (UP == 2) ? idx / (PH * PW) : idx % KQ, pix = (UP == 2) ? idx % (PH * PW) : idx / KQ;
        const int py = pix / PW, px = pix - py * PW;
        int Y = y0 - a.org + py, X = x0 - a.org + px;
        bool ok = true;
        if (REFLECT) {                                        // tiles that hang over the image edge read clamped rows (their outputs are not stored)
            Y = min(max(reflect1(Y, HD), 0), HD - 1);
            X = min(max(reflect1(X, WD), 0), WD - 1);
        } else {
            ok = Y >= 0 && Y < HD && X >= 0 && X < WD;
        }
        f4v v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *(const f4v*)(sb + ((int64_t)(Y / UP) * a.Wsrc + (X / UP)) * CIN + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) patch[(q * 4 + e) * PLANE + py * PWS + px] = v[e];
    }
    __syncthreads();
    // ---- TH * 4 groups of 16 pixels per tile, TH per wave, two at a time (independent accumulator chains) -----------------------
    const float bv = a.bias ? a.bias[l16] : 0.f;
    float* ob = a.out + (int64_t)b * a.Hd * a.Wd * 16;
#pragma unroll 1
    for (int g2 = 0; g2 < TH; g2 += 2) {
        const int g0 = wave * TH + g2, g1 = g0 + 1;
        const int r0 = g0 >> 2, c0 = (g0 & 3) * 16, r1 = g1 >> 2, c1 = (g1 & 3) * 16;
        f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* p0 = patch + kq * PLANE + r0 * PWS + c0 + l16;
        const float* p1 = patch + kq * PLANE + r1 * PWS + c1 + l16;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = (t / 3) * PWS + (t % 3);
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0[4 * j * PLANE + off], wreg[t][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1[4 * j * PLANE + off], wreg[t][j], acc1, 0, 0, 0);
            }
        }
        // D[m][n]: lane (kq, l16) holds pixels m = 4 kq + i of the group, channel n = l16
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f4v acc = h ? acc1 : acc0;
            const int y = y0 + (h ? r1 : r0), xb = x0 + (h ? c1 : c0) + 4 * kq;
            if (y >= a.Hd) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xb + i < a.Wd) ob[((int64_t)y * a.Wd + xb + i) * 16 + l16] = apply_act(acc[i] + bv, a.act);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The RGB stem (networks.py:44-50 via torchvision resnet.conv1): 7x7 / stride 2 / pad 3, 3 -> 64 channels, input normalisation
// (x - 0.45) / 0.225 and folded BatchNorm + ReLU.  As an implicit GEMM its K = 147 cannot use the 16-byte channel-quad loader
// (scalar gathers: 97 us for 2.9 GFLOP).  Patch kernel: the normalised, zero-padded input patch of a 4 x 32 output tile (14 x 69
// pixels x 3 channels, 11 KB) is staged once; every wave owns 16 of the 64 output channels with its 37 x 4 weight fragments in
// registers and sweeps the tile's 8 groups of 16 pixels: v_mfma_f32_16x16x4_f32, M = 16 output pixels, K = 4 of the 148
// (kh, kw, ci) taps -- tap k of output pixel (y, x) sits at patch offset (2 y) * ROW + 6 x + k + (k / 21) * (ROW - 21).
// ---------------------------------------------------------------------------------------------------------------------
struct StemArgs {
    const float* src;      // (B, Hs, Ws, 3) in [0, 1]
    const float* w;        // [(kh, kw, ci)][ldw]
    const float* scale;    // folded BatchNorm (may be NULL)
    const float* shift;
    float* out;            // (B, Hd, Wd, 64)
    int B, Hs, Ws, Hd, Wd, ldw, act;
    float in_sub, in_mul;
};

__global__ __launch_bounds__(256) void k_conv7x7_stem(StemArgs a) {
    constexpr int TH = 4, TW = 32, PH = 2 * TH + 6, PW = 2 * TW + 5, ROW = 208, NJ = 37;        // ROW >= 3 PW = 207
    __shared__ float patch[PH * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int co = 16 * wave + l16;
    float wreg[NJ];
    int offk[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = 4 * j + kq;
        wreg[j] = k < 147 ? a.w[(int64_t)k * a.ldw + co] : 0.f;
        offk[j] = k + (k / 21) * (ROW - 21);
    }
    const float* sb = a.src + (int64_t)b * a.Hs * a.Ws * 3;
    for (int idx = tid; idx < PH * ROW; idx += 256) {
        const int py = idx / ROW, e = idx - py * ROW;
        const int Y = 2 * y0 - 3 + py, X = 2 * x0 - 3 + e / 3;
        const bool ok = e < 3 * PW && Y >= 0 && Y < a.Hs && X >= 0 && X < a.Ws;
        patch[idx] = ok ? (sb[((int64_t)Y * a.Ws + X) * 3 + (e - (e / 3) * 3)] - a.in_sub) * a.in_mul : 0.f;     // zero padding of the NORMALISED image
    }
    __syncthreads();
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.shift ? a.shift[co] : 0.f;
    float* ob = a.out + (int64_t)b * a.Hd * a.Wd * 64;
#pragma unroll 1
    for (int g = 0; g < 2 * TH; g += 2) {                    // two groups (the two halves of a tile row) per pass: independent chains
        const int r = g >> 1;
        const float* p0 = patch + 2 * r * ROW + 6 * l16;
        const float* p1 = p0 + 6 * 16;
        f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(p0[offk[j]], wreg[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(p1[offk[j]], wreg[j], acc1, 0, 0, 0);
        }
        const int y = y0 + r;
        if (y >= a.Hd) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f4v acc = h ? acc1 : acc0;
            const int xb = x0 + 16 * h + 4 * kq;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (xb + i < a.Wd) ob[((int64_t)y * a.Wd + xb + i) * 64 + co] = apply_act(fmaf(acc[i], sc, sh), a.act);
        }
    }
}

__global__ __launch_bounds__(256) void k_conv_splitk_epilogue(const float* __restrict__ slab, int S, int64_t total, int Ncols,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* res, float* out, int act, const float* __restrict__ xin, int dact,
                                                              const float* __restrict__ pre) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += slab[(int64_t)z * total + i];        // fixed order
        const int col = (int)(i % Ncols);
        v = fmaf(v, scale ? scale[col] : 1.f, shift ? shift[col] : 0.f);
        if (pre) v += pre[i];
        if (dact) v *= act_deriv(xin[i], dact);
        if (res) v += res[i];
        out[i] = apply_act(v, act);
    }
}

// class form (stride-2 backward-data) split by tap: output pixel (b, y, x) belongs to class (y & 1, x & 1), its partial sums are row
// (b, y >> 1, x >> 1) of that class's slabs, one per tap of the class, added in tap order
__global__ __launch_bounds__(256) void k_conv_splitk_epilogue_cls(const float* __restrict__ slab, int64_t cls_rows, int B, int Hd, int Wd, int Ncols,
                                                                  int KH, int KW, int off, const float* res, float* out, const float* __restrict__ xin,
                                                                  int dact, const float* __restrict__ pre) {
    const unsigned colq = (unsigned)Ncols >> 2, totalq = (unsigned)B * Hd * Wd * colq;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < totalq; q += gridDim.x * 256u) {
        unsigned t = q;
        const unsigned cq = t % colq; t /= colq;
        const unsigned x = t % Wd; t /= Wd;
        const unsigned y = t % Hd;
        const unsigned b = t / Hd;
        const int py = y & 1, px = x & 1;
        const int kh0 = (py + off) & 1, kw0 = (px + off) & 1;
        const int ntap = ((KH - kh0 + 1) / 2) * ((KW - kw0 + 1) / 2);
        const int Hc = (Hd - py + 1) / 2, Wc = (Wd - px + 1) / 2;
        const int64_t row = ((int64_t)b * Hc + (y >> 1)) * Wc + (x >> 1);
        f4v v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < ntap; ++z) v += *(const f4v*)(slab + (((int64_t)z * 4 + (py * 2 + px)) * cls_rows + row) * Ncols + cq * 4);
        if (pre) v += ((const f4v*)pre)[q];
        if (dact) {
            const f4v xv = ((const f4v*)xin)[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] *= act_deriv(xv[k], dact);
        }
        if (res) v += ((const f4v*)res)[q];
        ((f4v*)out)[q] = v;
    }
}

// the same on 16-byte quads (Ncols % 4 == 0 and total < 2^31: every split layer of this network)
__global__ __launch_bounds__(256) void k_conv_splitk_epilogue4(const float* __restrict__ slab, int S, unsigned totalq, int Ncols,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* res, float* out, int act, const float* __restrict__ xin, int dact,
                                                               const float* __restrict__ pre) {
    const unsigned colq = (unsigned)Ncols >> 2;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < totalq; q += gridDim.x * 256u) {
        f4v v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < S; ++z) v += ((const f4v*)slab)[(int64_t)z * totalq + q];        // fixed order
        const int col = (int)(q % colq) * 4;
        const f4v sc = scale ? *(const f4v*)(scale + col) : (f4v){1.f, 1.f, 1.f, 1.f};
        const f4v sh = shift ? *(const f4v*)(shift + col) : (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], sc[k], sh[k]);
        if (pre) v += ((const f4v*)pre)[q];
        if (dact) {
            const f4v xv = ((const f4v*)xin)[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] *= act_deriv(xv[k], dact);
        }
        if (res) v += ((const f4v*)res)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], act);
        ((f4v*)out)[q] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward-weight: dW[m, n] = sum_p dZ[p, m] * X[p, n]   (m = cout, n = (kh,kw,ci) [+ 1 ones-column for the bias],
// p = output pixel) as a split-K GEMM: grid.z slices of the pixel range write partial slabs, k_wgrad_reduce adds
// them in slice order (deterministic) and scatters to torch's (Cout,Cin,KH,KW) layout.
// ---------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dz;       // [P][Cout]
    const float* src0;
    const float* src1;
    float* slabs;          // [S][Mpad][Npad]
    int B, Hs, Ws, Cin, C1, up;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad, pad_mode;
    int Ngemm;             // KH*KW*Cin (+1 when has_bias)
    int has_bias;
    int Mpad, Npad;
    int64_t pix_per_slice;
    float in_sub, in_mul;
    int vec;
};

// XCD-contiguous order for the backward-weight grids: the column tiles (x) and row tiles (y) of ONE pixel slice (z) read the same dZ rows
// and the same input rows -- they run on one XCD, back to back
#define WGRAD_BLOCK_IDS                                                                                                                   \
    unsigned bx, by, bz;                                                                                                                  \
    set_wave_priority(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));                                                   \
    {                                                                                                                                     \
        const unsigned lin = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z); \
        bx = lin % gridDim.x;                                                                                                             \
        const unsigned t_ = lin / gridDim.x;                                                                                              \
        by = t_ % gridDim.y;                                                                                                              \
        bz = t_ / gridDim.y;                                                                                                              \
    }

template <int WM, int WN, int VEC>
__global__ __launch_bounds__(256) void k_wgrad_gemm(WgradArgs a) {
    constexpr int BM = 32 * WM, BN = 32 * WN, NT = 256;     // WM x WN = 4 waves, one 32x32 accumulator each
    constexpr int A_CNT = CBK * (BM / 4), B_CNT = CBK * (BN / 4);
    constexpr int A_PER = (A_CNT + NT - 1) / NT, B_PER = (B_CNT + NT - 1) / NT;
    __shared__ float As[2][CBK][BM];
    __shared__ float Bs[2][CBK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    WGRAD_BLOCK_IDS
    const int m0 = by * BM, nn0 = bx * BN;
    const int64_t P = (int64_t)a.B * a.Ho * a.Wo;
    const int64_t p0 = (int64_t)bz * a.pix_per_slice;
    const int64_t p1 = (p0 + a.pix_per_slice < P) ? p0 + a.pix_per_slice : P;
    const int nchunks = (p1 > p0) ? (int)((p1 - p0 + CBK - 1) / CBK) : 0;
    const int Hl = a.Hs / a.up, Wl = a.Ws / a.up, C2 = a.Cin - a.C1;
    const int Kconv = a.KH * a.KW * a.Cin;

    // per-thread fixed roles.  B: (pixel row kr inside the chunk, column quad) -> the column's (tap, ci) is decoded ONCE and
    // the pixel (b,oh,ow) is advanced incrementally by CBK per chunk (no divisions in the loop).
    int b_kr[B_PER], b_col[B_PER], b_kh[B_PER], b_kw[B_PER], b_ci[B_PER];
    int pb[B_PER], poh[B_PER], pow_[B_PER];
    bool b_on[B_PER], b_conv[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int idx = tid + j * NT;
        b_on[j] = idx < B_CNT;
        b_kr[j] = idx / (BN / 4);
        b_col[j] = nn0 + (idx - b_kr[j] * (BN / 4)) * 4;
        b_conv[j] = b_col[j] < Kconv;
        const int tap = b_conv[j] ? b_col[j] / a.Cin : 0;
        b_ci[j] = b_conv[j] ? b_col[j] - tap * a.Cin : 0;
        b_kh[j] = tap / a.KW;
        b_kw[j] = tap - b_kh[j] * a.KW;
        const int64_t p = p0 + b_kr[j];
        pb[j] = (int)(p / ((int64_t)a.Ho * a.Wo));
        const int r = (int)(p - (int64_t)pb[j] * a.Ho * a.Wo);
        poh[j] = r / a.Wo;
        pow_[j] = r - poh[j] * a.Wo;
    }
    f4v areg[A_PER], breg[B_PER];
    int ld_chunk = 0;
    auto load_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + j * NT;
            const int kr = idx / (BM / 4), m = m0 + (idx - kr * (BM / 4)) * 4;
            const int64_t p = p0 + (int64_t)ld_chunk * CBK + kr;
            f4v v = (f4v){0.f, 0.f, 0.f, 0.f};
            if (idx < A_CNT && p < p1 && m < a.Cout) {
                if (a.Cout % 4 == 0) v = *(const f4v*)(a.dz + p * a.Cout + m);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (m + e < a.Cout) v[e] = a.dz[p * a.Cout + m + e];
            }
            areg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int64_t p = p0 + (int64_t)ld_chunk * CBK + b_kr[j];
            f4v v = (f4v){0.f, 0.f, 0.f, 0.f};
            if (b_on[j] && p < p1) {
                const int b = pb[j], oh = poh[j], ow = pow_[j];
                if (VEC == 4) {
                    if (b_conv[j]) {
                        int ys = oh * a.stride + b_kh[j] - a.pad, xs = ow * a.stride + b_kw[j] - a.pad;
                        bool ok = true;
                        if (a.pad_mode == 1) { ys = reflect1(ys, a.Hs); xs = reflect1(xs, a.Ws); }
                        else ok = ys >= 0 && ys < a.Hs && xs >= 0 && xs < a.Ws;
                        if (ok) {
                            const int ci = b_ci[j];
                            const float* q = (ci < a.C1)
                                ? a.src0 + (((int64_t)b * Hl + ys / a.up) * Wl + xs / a.up) * a.C1 + ci
                                : a.src1 + (((int64_t)b * a.Hs + ys) * a.Ws + xs) * C2 + (ci - a.C1);
                            v = *(const f4v*)q;
                        }
                    } else if (a.has_bias && b_col[j] == Kconv) {
                        v[0] = 1.f;                            // ones column: dW[:, Kconv] = sum_p dZ = bias gradient
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = b_col[j] + e;
                        if (n < Kconv) {
                            const int tp = n / a.Cin, c = n - tp * a.Cin, kh1 = tp / a.KW, kw1 = tp - kh1 * a.KW;
                            const int ys = oh * a.stride + kh1 - a.pad, xs = ow * a.stride + kw1 - a.pad;
                            if (ys >= 0 && ys < a.Hs && xs >= 0 && xs < a.Ws)
                                v[e] = (a.src0[(((int64_t)b * a.Hs + ys) * a.Ws + xs) * a.Cin + c] - a.in_sub) * a.in_mul;
                        } else if (a.has_bias && n == Kconv) v[e] = 1.f;
                    }
                }
            }
            breg[j] = v;
            // advance this role's pixel by one chunk
            pow_[j] += CBK;
            while (pow_[j] >= a.Wo) { pow_[j] -= a.Wo; if (++poh[j] == a.Ho) { poh[j] = 0; ++pb[j]; } }
        }
        ++ld_chunk;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + j * NT;
            if (idx >= A_CNT) continue;
            const int kr = idx / (BM / 4), mq = idx - kr * (BM / 4);
            *(f4v*)&As[buf][kr][mq * 4] = areg[j];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + j * NT;
            if (idx >= B_CNT) continue;
            const int nq = idx - b_kr[j] * (BN / 4);
            *(f4v*)&Bs[buf][b_kr[j]][nq * 4] = breg[j];
        }
    };
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    const int khalf = lane >> 5;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk();
        float av[CBK / 2], bv[CBK / 2];                           // the chunk's fragments first, then the MFMAs (see k_conv_gemm)
#pragma unroll
        for (int kk = 0; kk < CBK / 2; ++kk) {
            av[kk] = As[buf][kk * 2 + khalf][wm * 32 + (lane & 31)];
            bv[kk] = Bs[buf][kk * 2 + khalf][wn * 32 + (lane & 31)];
        }
#pragma unroll
        for (int kk = 0; kk < CBK / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, CBK / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, CBK / 2, 0);
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* slab = a.slabs + (int64_t)bz * a.Mpad * a.Npad;
    const int n = nn0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (m < a.Mpad && n < a.Npad) slab[(int64_t)m * a.Npad + n] = acc[r];
    }
}

// VEC-4 backward-weight with the lean loader of k_conv_gemm: buffer resources + 32-bit offsets (structural zeros are
// out-of-range offsets), no integer divisions in the chunk loop, CB pixels per chunk.  A concat layer reads its two
// sources through two resources: lanes of one load instruction span columns on both sides of the split, and a
// resource is wave-uniform, so each slot issues one load per source with the foreign lanes out of range.
template <int WM, int WN, int CB, bool TWO>
__global__ __launch_bounds__(256) void k_wgrad_gemm4(WgradArgs a) {
    constexpr int BM = 32 * WM, BN = 32 * WN, NT = 256;     // WM x WN = 4 waves, one 32x32 accumulator each
    constexpr int A_CNT = CB * (BM / 4), B_CNT = CB * (BN / 4);
    constexpr int A_PER = A_CNT / NT, B_PER = B_CNT / NT;
    static_assert(A_CNT % NT == 0 && B_CNT % NT == 0, "tile / thread-count mismatch");
    __shared__ float As[2][CB][BM];
    __shared__ float Bs[2][CB][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    WGRAD_BLOCK_IDS
    const int m0 = by * BM, nn0 = bx * BN;
    const int P = a.B * a.Ho * a.Wo;                          // < 2^24 (host): pixel indices are exact in fp32
    const int p0 = (int)((int64_t)bz * a.pix_per_slice);
    const int p1 = (p0 + a.pix_per_slice < P) ? (int)(p0 + a.pix_per_slice) : P;
    const int npix = (p1 > p0) ? p1 - p0 : 0;
    const int nchunks = (npix + CB - 1) / CB;
    const int sh = a.up >> 1;
    const int Hl = a.Hs >> sh, Wl = a.Ws >> sh, C2 = a.Cin - a.C1;
    const int Kconv = a.KH * a.KW * a.Cin;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)((int64_t)P * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)((int64_t)a.B * Hl * Wl * a.C1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0,
                                                                         (int)(a.src1 ? (int64_t)a.B * a.Hs * a.Ws * C2 * 4 : 0), 0x00020000);

    // A (dZ) slots: row kr of the chunk, cout quad -- the chunk rides in soffset
    int a_kr[A_PER];
    unsigned a_off[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int idx = tid + j * NT;
        a_kr[j] = idx / (BM / 4);
        const int m = m0 + (idx - a_kr[j] * (BM / 4)) * 4;
        a_off[j] = (m < a.Cout) ? (unsigned)(a_kr[j] * a.Cout + m) * 4u : OOB;      // Cout % 4 == 0
    }
    // B (gathered input) slots.  A thread's B_PER slots are B_PER column quads of ONE pixel row of the chunk (row = tid / QG, quads
    // tid % QG + j QG): the pixel -- (image, row, column) of p0 + row + c CB -- is worked out once per thread and chunk, each slot adds its
    // own tap (kh, kw) and channel quad, decoded ONCE here.  (Round 3 first gave every slot its own pixel: B_PER full decodes per chunk;
    // the backward-weight GEMM issues ~12 vector instructions per MFMA and they are NOT hidden under the MFMAs -- profiles/r03_bench_pmc_sq.txt.)
    // The pixel itself advances incrementally by CB per chunk with at most three column wraps and one row wrap -- selects, no branches --
    // when the image is at least 11 columns and 3 rows; smaller ones re-decode it by reciprocal multiplication + exact fix-up.
    const float inv_cin = 1.0f / (float)a.Cin, inv_kw = 1.0f / (float)a.KW;
    constexpr int QG = (BN / 4) / B_PER;                     // column quads per slot group
    static_assert(NT / QG == CB, "one pixel row of the chunk per thread");
    const int b_row = tid / QG;
    int b_quad[B_PER], b_kh[B_PER], b_kw[B_PER], b_ci[B_PER];
    bool b_conv[B_PER], b_ones[B_PER], b_src0[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        b_quad[j] = tid % QG + j * QG;
        const int col = nn0 + b_quad[j] * 4;
        b_conv[j] = col < Kconv;
        b_ones[j] = a.has_bias && col == Kconv;
        // (col < 2^24: quotients by reciprocal multiplication + exact fix-up instead of two 32-bit integer divisions per slot)
        const int cc = b_conv[j] ? col : 0;
        int tap = (int)((float)cc * inv_cin);
        tap -= (tap * a.Cin > cc) ? 1 : 0;
        tap += ((tap + 1) * a.Cin <= cc) ? 1 : 0;
        b_ci[j] = cc - tap * a.Cin;
        b_src0[j] = b_ci[j] < a.C1;
        int tkh = (int)((float)tap * inv_kw);
        tkh -= (tkh * a.KW > tap) ? 1 : 0;
        tkh += ((tkh + 1) * a.KW <= tap) ? 1 : 0;
        b_kh[j] = tkh - a.pad;                               // tap offset relative to the strided output position
        b_kw[j] = tap - tkh * a.KW - a.pad;
    }
    const int hw = a.Ho * a.Wo;
    const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)a.Wo;
    const bool reflect = a.pad_mode == 1;
    // two staging sets: loads run two chunks ahead (see k_conv_gemm).  The sets hold what the loads RETURN and nothing else: the merge of
    // the two sources of a concat layer and the ones column are applied when a set is written to LDS, two steps later -- applied at load
    // time (the first form) they are uses of the loaded registers, i.e. an s_waitcnt vmcnt(0) right behind every load: no prefetch at all
    // (round-3 disassembly)
    f4v areg[2][A_PER], breg[2][B_PER], breg1[2][TWO ? B_PER : 1];
    int left_of[2] = {0, 0};
    int ld_chunk = 0;
    int cur_b = 0, cur_oh = 0, cur_ow = 0;                   // the thread's pixel of the chunk being loaded
    const bool incremental = a.Wo >= 11 && a.Ho >= 3;
    auto load_chunk = [&](int set) {
        const int left = npix - ld_chunk * CB;               // pixels of the slice that remain from this chunk on (<= 0 past the end)
        left_of[set] = left;
        const int pc = p0 + ld_chunk * CB;
        const int soffz = pc * a.Cout * 4;
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            areg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsz, (a_kr[j] < left) ? a_off[j] : OOB, soffz, 0));
        // the thread's pixel of this chunk (24-bit multiplies throughout: every factor is a pixel / row / channel count below 2^24 -- the
        // host keeps P there -- and v_mul_u32_u24 / v_mad_u32_u24 issue at full rate where v_mul_lo_u32 / v_mad_u64_u32 take four times as long)
        if (!incremental || ld_chunk == 0) {
            const unsigned p = (unsigned)(pc + b_row);
            unsigned b = (unsigned)((float)p * inv_hw);      // floor(p / hw) up to +-1 ...
            b -= (__umul24(b, (unsigned)hw) > p) ? 1u : 0u;  // ... made exact
            b += (__umul24(b + 1u, (unsigned)hw) <= p) ? 1u : 0u;
            const unsigned r = p - __umul24(b, (unsigned)hw);
            unsigned oh = (unsigned)((float)r * inv_wo);
            oh -= (__umul24(oh, (unsigned)a.Wo) > r) ? 1u : 0u;
            oh += (__umul24(oh + 1u, (unsigned)a.Wo) <= r) ? 1u : 0u;
            cur_b = (int)b; cur_oh = (int)oh; cur_ow = (int)(r - __umul24(oh, (unsigned)a.Wo));
        } else {                                             // + CB pixels: <= 3 column wraps (Wo >= 11), <= 1 row wrap (Ho >= 3)
            cur_ow += CB;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool c = cur_ow >= a.Wo;
                cur_ow -= c ? a.Wo : 0;
                cur_oh += c ? 1 : 0;
            }
            const bool c = cur_oh >= a.Ho;
            cur_oh -= c ? a.Ho : 0;
            cur_b += c ? 1 : 0;
        }
        const int ybase = (int)__umul24((unsigned)cur_oh, (unsigned)a.stride), xbase = (int)__umul24((unsigned)cur_ow, (unsigned)a.stride);
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int ys0 = ybase + b_kh[j], xs0 = xbase + b_kw[j];
            // (bitwise, not short-circuit: `&&` became s_and_saveexec / s_or exec pairs around the later comparisons, and an instruction that
            //  writes exec is a wall for the instruction scheduler -- the MFMAs of the step could not be moved ahead of the gather arithmetic)
            const bool inside = ((unsigned)ys0 < (unsigned)a.Hs) & ((unsigned)xs0 < (unsigned)a.Ws);
            const int ys = reflect ? reflect1(ys0, a.Hs) : ys0, xs = reflect ? reflect1(xs0, a.Ws) : xs0;
            const bool ok = b_conv[j] & (b_row < left) & (reflect | inside);
            // (offset | out-of-range bit: written as `ok ? offset : OOB` the compiler sinks the offset arithmetic into an exec-masked branch;
            //  a structural zero may carry a negative ys / xs: its offset is garbage and never used)
            const unsigned pix0 = umad24(umad24((unsigned)cur_b, (unsigned)Hl, (unsigned)(ys >> sh)), (unsigned)Wl, (unsigned)(xs >> sh));
            const unsigned o0 = umad24(pix0, (unsigned)a.C1, (unsigned)b_ci[j]) * 4u;
            breg[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs0, o0 | ((ok && b_src0[j]) ? 0u : OOB), 0, 0));
            if (TWO) {                                       // concat layer: the lanes of one load straddle the two sources
                const unsigned pix1 = umad24(umad24((unsigned)cur_b, (unsigned)a.Hs, (unsigned)ys), (unsigned)a.Ws, (unsigned)xs);
                const unsigned o1 = umad24(pix1, (unsigned)C2, (unsigned)(b_ci[j] - a.C1)) * 4u;
                breg1[set][j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs1, o1 | ((ok && !b_src0[j]) ? 0u : OOB), 0, 0));
            }
        }
        ++ld_chunk;
    };
    auto store_chunk = [&](int buf, int set) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + j * NT;
            *(f4v*)&As[buf][a_kr[j]][(idx - a_kr[j] * (BM / 4)) * 4] = areg[set][j];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            f4v v = breg[set][j];
            if (TWO) v = b_src0[j] ? v : breg1[set][j];
            const f4v ones = {(b_row < left_of[set]) ? 1.f : 0.f, 0.f, 0.f, 0.f};     // ones column: dW[:, Kconv] = sum_p dZ = bias gradient
            *(f4v*)&Bs[buf][b_row][b_quad[j] * 4] = b_ones[j] ? ones : v;
        }
    };
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    load_chunk(0);
    load_chunk(1);
    store_chunk(0, 0);
    __syncthreads();
    const int khalf = lane >> 5;
    // software-pipelined step, as in k_conv_gemm: loads of chunk c + 2 and the LDS store of chunk c + 1 ride under the first half of
    // chunk c's MFMAs, the barrier sits in the middle, the fragments of chunk c + 1 are read under the second half into the other
    // fragment register set.  Loads / stores are unconditional (one basic block per step), steps come in pairs with the odd one behind
    // the loop (no control-flow edge from the middle of a pair back to the loop header: see k_conv_gemm).
    float fa[2][CB / 2], fb[2][CB / 2];
    auto read_frags = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
#pragma unroll
        for (int kk = 0; kk < CB / 2; ++kk) {
            fa[buf][kk] = As[buf][kk * 2 + khalf][wm * 32 + (lane & 31)];
            fb[buf][kk] = Bs[buf][kk * 2 + khalf][wn * 32 + (lane & 31)];
        }
    };
    auto pipe_step = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        load_chunk(buf);
        store_chunk(buf ^ 1, buf ^ 1);
#pragma unroll
        for (int kk = 0; kk < CB / 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][kk], fb[buf][kk], acc, 0, 0, 0);
        // the gather's address arithmetic (~100 vector instructions per step), its loads and the LDS stores are dealt out between the MFMAs of the
        // first half: hipcc's own order issued all of it ahead of the first MFMA of the step (round-4 disassembly)
#pragma unroll
        for (int g = 0; g < CB / 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(std::integral_constant<int, buf ^ 1>{});
#pragma unroll
        for (int kk = CB / 4; kk < CB / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][kk], fb[buf][kk], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < CB / 4; ++g) {                   // one MFMA, then two LDS reads, eight times
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nchunks > 0) {
        read_frags(std::integral_constant<int, 0>{});
        int c = 0;
        for (; c + 1 < nchunks; c += 2) {
            pipe_step(std::integral_constant<int, 0>{});
            pipe_step(std::integral_constant<int, 1>{});
        }
        if (c < nchunks) pipe_step(std::integral_constant<int, 0>{});
    }
    // slab store through a buffer resource: one 32-bit lane offset (beyond num_records where the tile overhangs the slab), the row of
    // accumulator element r in the scalar offset -- 16 stores and a handful of address instructions instead of 16 exec-masked blocks with
    // 64-bit pointer arithmetic each (the backward-weight kernel is the VALU-heaviest one: profiles/r03_bench_pmc_sq.txt)
    {
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.slabs + (int64_t)bz * a.Mpad * a.Npad), 0, a.Mpad * a.Npad * 4, 0x00020000);
        const int n = nn0 + wn * 32 + (lane & 31), mb = m0 + wm * 32 + 4 * khalf;
        const int rows_left = a.Mpad - mb;                   // rows of this lane's 32-row block that exist (the range check ignores soffset)
        const unsigned voff = (n < a.Npad && rows_left > 0) ? (unsigned)(mb * a.Npad + n) * 4u : OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float v = acc[r];                          // (a scalar copy first: __builtin_bit_cast applied to the vector ELEMENT stored element 0 sixteen times)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsl, (dr < rows_left) ? voff : OOB, dr * a.Npad * 4, 0);
        }
    }
}

// Backward-weight for the 16-output-channel layers (the two full-resolution decoder convolutions, networks.py:262-266 at level
// 0): a 32x32x2 tile would spend half of every MFMA on padding rows, so this variant runs v_mfma_f32_16x16x4_f32 -- M = the 16
// output channels, N = NTL 16-column tiles of (tap, ci) columns held by EVERY wave (NTL x 4 accumulator registers), and the four
// waves of a workgroup split the pixels of a chunk (K) between them; their accumulators are summed through LDS in wave order at
// the end.  One workgroup therefore reads a pixel's dZ once for all its columns, and a slab is 16 x 160 floats instead of
// 32 x 256.  Loader, slab layout and reduction are those of k_wgrad_gemm4.
template <int NTL, int CB>
__global__ __launch_bounds__(256) void k_wgrad_gemm16(WgradArgs a) {
    constexpr int BM = 16, BN = 16 * NTL, BNS = (BN % 32 == 16) ? BN : BN + 16, NT = 256;   // row stride = 16 mod 32 words: the 2 pixel
    constexpr int A_CNT = CB * (BM / 4), B_CNT = CB * (BN / 4);                            // rows a half-wave reads hit disjoint banks
    constexpr int A_PER = (A_CNT + NT - 1) / NT, B_PER = (B_CNT + NT - 1) / NT;
    static_assert(2 * CB * (BM + BNS) >= 3 * NTL * 4 * 64, "tile memory is reused for the cross-wave sum");
    __shared__ float smem[2 * CB * (BM + BNS)];
    float (*As)[CB][BM] = reinterpret_cast<float (*)[CB][BM]>(smem);
    float (*Bs)[CB][BNS] = reinterpret_cast<float (*)[CB][BNS]>(smem + 2 * CB * BM);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WGRAD_BLOCK_IDS
    (void)by;
    const int nn0 = bx * BN;
    const int64_t P = (int64_t)a.B * a.Ho * a.Wo;
    const int64_t p0 = (int64_t)bz * a.pix_per_slice;
    const int64_t p1 = (p0 + a.pix_per_slice < P) ? p0 + a.pix_per_slice : P;
    const int npix = (p1 > p0) ? (int)(p1 - p0) : 0;
    const int nchunks = (npix + CB - 1) / CB;
    const int sh = a.up >> 1;
    const int Hl = a.Hs >> sh, Wl = a.Ws >> sh, C2 = a.Cin - a.C1;
    const int Kconv = a.KH * a.KW * a.Cin;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, (int)(P * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)((int64_t)a.B * Hl * Wl * a.C1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0,
                                                                         (int)(a.src1 ? (int64_t)a.B * a.Hs * a.Ws * C2 * 4 : 0), 0x00020000);
    int a_kr[A_PER];
    unsigned a_off[A_PER];
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
        const int idx = tid + j * NT;
        a_kr[j] = idx / (BM / 4);
        const int m = (idx - a_kr[j] * (BM / 4)) * 4;
        a_off[j] = (idx < A_CNT && m < a.Cout) ? (unsigned)(a_kr[j] * a.Cout + m) * 4u : OOB;
    }
    int b_kr[B_PER], b_kh[B_PER], b_kw[B_PER], b_ci[B_PER], pb[B_PER], poh[B_PER], pow_[B_PER];
    bool b_conv[B_PER], b_ones[B_PER], b_src0[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const int idx = tid + j * NT;
        b_kr[j] = idx / (BN / 4);
        const int col = nn0 + (idx - b_kr[j] * (BN / 4)) * 4;
        b_conv[j] = idx < B_CNT && col < Kconv;
        b_ones[j] = idx < B_CNT && a.has_bias && col == Kconv;
        const int tap = b_conv[j] ? col / a.Cin : 0;
        b_ci[j] = b_conv[j] ? col - tap * a.Cin : 0;
        b_src0[j] = b_ci[j] < a.C1;
        b_kh[j] = tap / a.KW;
        b_kw[j] = tap - b_kh[j] * a.KW;
        const int64_t p = p0 + b_kr[j];
        const int hw = a.Ho * a.Wo;
        pb[j] = (int)(p / hw);
        const int r = (int)(p - (int64_t)pb[j] * hw);
        poh[j] = r / a.Wo;
        pow_[j] = r - poh[j] * a.Wo;
    }
    f4v areg[A_PER], breg[B_PER];
    int ld_chunk = 0;
    auto load_chunk = [&]() {
        const int left = npix - ld_chunk * CB;
        const int soffz = (int)((p0 + (int64_t)ld_chunk * CB) * a.Cout * 4);
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
            areg[j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsz, (a_kr[j] < left) ? a_off[j] : OOB, soffz, 0));
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            int ys = poh[j] * a.stride + b_kh[j] - a.pad, xs = pow_[j] * a.stride + b_kw[j] - a.pad;
            bool ok = b_conv[j] && b_kr[j] < left;
            if (a.pad_mode == 1) { ys = reflect1(ys, a.Hs); xs = reflect1(xs, a.Ws); }
            else ok = ok && ys >= 0 && ys < a.Hs && xs >= 0 && xs < a.Ws;
            const unsigned o0 = (unsigned)(((pb[j] * Hl + (ys >> sh)) * Wl + (xs >> sh)) * a.C1 + b_ci[j]) * 4u;
            const unsigned o1 = (unsigned)(((pb[j] * a.Hs + ys) * a.Ws + xs) * C2 + (b_ci[j] - a.C1)) * 4u;
            f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs0, (ok && b_src0[j]) ? o0 : OOB, 0, 0));
            if (a.src1 != nullptr) {
                const f4v v1 = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs1, (ok && !b_src0[j]) ? o1 : OOB, 0, 0));
                v = b_src0[j] ? v : v1;
            }
            if (b_ones[j]) v = (f4v){(b_kr[j] < left) ? 1.f : 0.f, 0.f, 0.f, 0.f};
            breg[j] = v;
            pow_[j] += CB;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2)
                if (pow_[j] >= a.Wo) { pow_[j] -= a.Wo; if (++poh[j] == a.Ho) { poh[j] = 0; ++pb[j]; } }
        }
        ++ld_chunk;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + j * NT;
            if (idx >= A_CNT) continue;
            *(f4v*)&As[buf][a_kr[j]][(idx - a_kr[j] * (BM / 4)) * 4] = areg[j];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + j * NT;
            if (idx >= B_CNT) continue;
            *(f4v*)&Bs[buf][b_kr[j]][(idx - b_kr[j] * (BN / 4)) * 4] = breg[j];
        }
    };
    f4v acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = (f4v){0.f, 0.f, 0.f, 0.f};
    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
    }
    __syncthreads();
    const int kq = lane >> 4, l16 = lane & 15;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk();
#pragma unroll
        for (int ks = 0; ks < CB / 16; ++ks) {               // this wave's CB / 4 pixel rows of the chunk, 4 per MFMA
            const int row = wave * (CB / 4) + ks * 4 + kq;
            const float av = As[buf][row][l16];
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[buf][row][t * 16 + l16], acc[t], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    float* red = smem;                                       // [3 waves][NTL * 4][64]
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave - 1) * NTL * 4 + t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (wave > 0) return;
    float* slab = a.slabs + (int64_t)bz * a.Mpad * a.Npad;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[t][i];
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) v += red[(w2 * NTL * 4 + t * 4 + i) * 64 + lane];
            const int m = 4 * kq + i, n = nn0 + t * 16 + l16;            // D[m][n]: lane (kq, l16) holds rows 4 kq .. 4 kq + 3
            if (n < a.Npad) slab[(int64_t)m * a.Npad + n] = v;
        }
}

// Backward-weight of the same thin 3x3 layers (16 output channels, 16 or 32 input channels, reflection pad, optional x2 upsample)
// from a staged patch: dW[co][(tap, ci)] = sum_p dZ[p][co] X[p + tap][ci].  One workgroup stages the dZ tile (4 x 64 pixels) and the
// input patch (6 x 66 pixels, 16 channels of the source starting at channel c0) in LDS in their natural NHWC order -- both are
// conflict-free MFMA operands that way -- and each wave sweeps one tile row: per 4 pixels ONE A read (dZ) feeds 9 MFMAs (one per tap)
// + 1 for the bias column.  The implicit-GEMM form re-reads dZ for every column tile and every input pixel nine times through L2
// (k_wgrad_gemm16: 94 us / 55 us for the two layers).  The four waves' accumulators are summed through LDS in wave order, a workgroup
// covers `tiles_x` consecutive tiles and writes ONE slab [16][npad]; k_wgrad_reduce adds the slabs in order.
struct ThinWgradArgs {
    const float* dz;       // (B, Hd, Wd, 16 MT)
    const float* src0;     // (B, Hd / up, Wd / up, C1): the (optionally x2 upsampled) first source
    const float* src1;     // (B, Hd, Wd, Cin - C1): the skip connection, or NULL
    float* slabs;          // [S][16 MT][npad]
    int B, Hd, Wd, Cin, C1, up, npad, has_bias, tiles_x, nxg;          // nxg: workgroups along x
};

template <int MT>
__global__ __launch_bounds__(256) void k_wgrad3x3_thin(ThinWgradArgs a) {
    constexpr int TH = 4, TW = 64, PH = TH + 2, PW = TW + 2, CO = 16 * MT, NA = 10 * MT;
    __shared__ float smem[PH * PW * 16 + TH * TW * CO];
    float* patch = smem;                                     // [py][px][ci]
    float* dzs = smem + PH * PW * 16;                        // [row][x][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int nhalf = a.Cin >> 4, n0 = a.C1 >> 4;
    const int b = blockIdx.z / nhalf, hf = blockIdx.z % nhalf;
    const bool first = hf < n0;                              // this workgroup's 16 input channels: of src0 (behind the upsample) or of src1
    const int sh = first ? (a.up >> 1) : 0, Cs = first ? a.C1 : a.Cin - a.C1, cs0 = (first ? hf : hf - n0) * 16;
    const int Hs = a.Hd >> sh, Ws = a.Wd >> sh;
    const int y0 = blockIdx.y * TH;
    const float* sb = (first ? a.src0 : a.src1) + (int64_t)b * Hs * Ws * Cs + cs0;
    const float* zb = a.dz + (int64_t)b * a.Hd * a.Wd * CO;
    f4v acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = (f4v){0.f, 0.f, 0.f, 0.f};
    const float one = (l16 == 0) ? 1.f : 0.f;                // B operand of the bias column: n = 0 only
    for (int tx = 0; tx < a.tiles_x; ++tx) {
        const int x0 = (blockIdx.x * a.tiles_x + tx) * TW;
        if (x0 >= a.Wd) break;                               // workgroup-uniform
        __syncthreads();                                     // the previous tile's operands are dead
        for (int idx = tid; idx < PH * PW * 4; idx += 256) { // patch: 16 channels = 4 quads per pixel, quad-fastest (64-byte runs)
            const int q = idx & 3, pix = idx >> 2;
            const int py = pix / PW, px = pix - py * PW;
            const int Y = min(max(reflect1(y0 - 1 + py, a.Hd), 0), a.Hd - 1), X = min(max(reflect1(x0 - 1 + px, a.Wd), 0), a.Wd - 1);
            *(f4v*)&patch[pix * 16 + q * 4] = *(const f4v*)(sb + ((int64_t)(Y >> sh) * Ws + (X >> sh)) * Cs + q * 4);
        }
        for (int idx = tid; idx < TH * TW * (CO / 4); idx += 256) {     // dZ tile; pixels outside the image contribute zeros
            const int q = idx % (CO / 4), pix = idx / (CO / 4);
            const int r = pix / TW, x = pix - r * TW;
            const bool ok = y0 + r < a.Hd && x0 + x < a.Wd;
            *(f4v*)&dzs[pix * CO + q * 4] = ok ? *(const f4v*)(zb + ((int64_t)(y0 + r) * a.Wd + x0 + x) * CO + q * 4) : (f4v){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        const float* ap = dzs + (wave * TW + kq) * CO + l16;             // A[m = co][k = pixel]: dZ[row = wave][4 ks + kq][co = 16 mt + l16]
        const float* bp = patch + (wave * PW + kq) * 16 + l16;           // B[k = pixel][n = ci]: X[row + kh][4 ks + kq + kw][ci = l16]
#pragma unroll 2
        for (int ks = 0; ks < TW / 4; ++ks) {
            float av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = ap[ks * 4 * CO + mt * 16];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float bv = bp[((t / 3) * PW + (t % 3)) * 16 + ks * 64];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt * 10 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv, acc[mt * 10 + t], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt * 10 + 9] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], one, acc[mt * 10 + 9], 0, 0, 0);
        }
    }
    // ---- cross-wave sum in a fixed tree ((w0 + w2) + (w1 + w3)) through LDS, then the slab ----------------------------------------
    float* red = smem;                                       // [2 waves][NA x 4][64 lanes]: the staged operands are dead after the barrier
    static_assert(2 * NA * 4 * 64 <= PH * PW * 16 + TH * TW * CO, "reduction scratch");
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave - 2) * NA * 4 + t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] += red[(wave * NA * 4 + t * 4 + i) * 64 + lane];
    }
    __syncthreads();                                         // every wave stays until the last barrier
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (wave != 0) return;
    const int64_t sid = ((int64_t)b * gridDim.y + blockIdx.y) * a.nxg + blockIdx.x;
    float* slab = a.slabs + sid * CO * a.npad;
    const int kconv = 9 * a.Cin, cg0 = hf * 16;                  // D[m = co][n]: lane (kq, l16) holds co = 16 mt + 4 kq + i, n = l16
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = acc[mt * 10 + t][i] + red[((mt * 10 + t) * 4 + i) * 64 + lane];
                const int co = mt * 16 + 4 * kq + i;
                if (t < 9) slab[co * a.npad + t * a.Cin + cg0 + l16] = v;
                else if (a.has_bias && hf == 0 && l16 == 0) slab[co * a.npad + kconv] = v;
            }
}

// Backward-weight of the RGB stem from the same staged patch: dW[co][k] = sum_p dZ[p][co] * patch(p, k), k = (kh, kw, ci) < 147.
// Every wave owns 16 output channels (M) and all ten 16-column tiles of k (N, 160 >= 148) -- no cross-wave sum; K = the tile's pixels,
// 4 per MFMA: one A read (dZ) feeds ten MFMAs.  A workgroup sweeps `tiles_x` tiles and writes one slab [64][148].
struct StemWgradArgs {
    const float* dz;       // (B, Hd, Wd, 64)
    const float* src;      // (B, Hs, Ws, 3)
    float* slabs;          // [S][64][148]
    int B, Hs, Ws, Hd, Wd, tiles_x, nxg;
    float in_sub, in_mul;
};

__global__ __launch_bounds__(256) void k_wgrad7x7_stem(StemWgradArgs a) {
    constexpr int TH = 4, TW = 32, PH = 2 * TH + 6, PW = 2 * TW + 5, ROW = 208, NPAD = 148;
    __shared__ float patch[PH * ROW];
    __shared__ float dzs[TH * TW * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int y0 = blockIdx.y * TH, b = blockIdx.z;
    int offn[10];
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
        const int k = 16 * nt + l16;                         // columns 147 .. 159 read valid patch cells and are never stored
        offn[nt] = k + (k / 21) * (ROW - 21) + 48 * kq;      // + this lane's pixel of the MFMA's 4: pixels ks, ks + 8, ks + 16, ks + 24 of the row
                                                             // (48 floats apart = 16 banks: the two k-groups of a half-wave read disjoint banks;
                                                             // adjacent pixels, 6 floats apart, gave a 39 % conflict rate)
    }
    f4v acc[10];
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc[nt] = (f4v){0.f, 0.f, 0.f, 0.f};
    const float* sb = a.src + (int64_t)b * a.Hs * a.Ws * 3;
    const float* zb = a.dz + (int64_t)b * a.Hd * a.Wd * 64;
    for (int tx = 0; tx < a.tiles_x; ++tx) {
        const int x0 = (blockIdx.x * a.tiles_x + tx) * TW;
        if (x0 >= a.Wd) break;
        __syncthreads();
        for (int idx = tid; idx < PH * ROW; idx += 256) {
            const int py = idx / ROW, e = idx - py * ROW;
            const int Y = 2 * y0 - 3 + py, X = 2 * x0 - 3 + e / 3;
            const bool ok = e < 3 * PW && Y >= 0 && Y < a.Hs && X >= 0 && X < a.Ws;
            patch[idx] = ok ? (sb[((int64_t)Y * a.Ws + X) * 3 + (e - (e / 3) * 3)] - a.in_sub) * a.in_mul : 0.f;
        }
        for (int idx = tid; idx < TH * TW * 16; idx += 256) {
            const int q = idx & 15, pix = idx >> 4;
            const int r = pix / TW, x = pix - r * TW;
            const bool ok = y0 + r < a.Hd && x0 + x < a.Wd;
            *(f4v*)&dzs[pix * 64 + q * 4] = ok ? *(const f4v*)(zb + ((int64_t)(y0 + r) * a.Wd + x0 + x) * 64 + q * 4) : (f4v){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < TH; ++r) {
            const float* ap = dzs + (r * TW + 8 * kq) * 64 + 16 * wave + l16;    // A[m = co][k = pixel ks + 8 kq]
            const float* bp = patch + 2 * r * ROW;                               // B[k = pixel][n]: patch(pixel, column 16 nt + l16)
#pragma unroll 2
            for (int ks = 0; ks < TW / 4; ++ks) {
                const float av = ap[ks * 64];
#pragma unroll
                for (int nt = 0; nt < 10; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[offn[nt] + ks * 6], acc[nt], 0, 0, 0);
            }
        }
    }
    float* slab = a.slabs + (((int64_t)b * gridDim.y + blockIdx.y) * a.nxg + blockIdx.x) * 64 * NPAD;
#pragma unroll
    for (int nt = 0; nt < 10; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 16 * nt + l16;
            if (n < 147) slab[(16 * wave + 4 * kq + i) * NPAD + n] = acc[nt][i];
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward-weight of the 3x3 stride-1 pad-1 layers with 32-channel tiles, TAP-REUSE form (round 4).
//   dW[co][ci][kh][kw] = sum over pixels  dZ[p][co] * Xpad[p + (kh - 1, kw - 1)][ci]
// The implicit-GEMM kernels above treat (tap, ci) as the GEMM's N dimension and gather every (pixel, tap, channel) operand per chunk:
// 16 MFMAs per 2 x 16 KB of staged operands, ~12 vector instructions per MFMA of address arithmetic.  Here a
// workgroup stages ONE 8 x 8 pixel patch of dZ (32 co) and the patch + 1-pixel halo of X (10 x 10, 32 ci) and runs ALL NINE taps from
// it -- a tap is a constant LDS offset: 288 MFMAs per chunk from 21 KB of operands, no per-tap addressing.  Three waves: wave w owns the
// kernel row kh = w, i.e. three accumulators (kw = 0, 1, 2) of the workgroup's 32 x 32 (co, ci) block over ALL 64 pixels -- no
// cross-wave sum, 48 accumulator registers per lane.  A k-step (2 pixels) is one dZ fragment read + three X reads + three MFMAs.
// Pixel slices (grid.z) write slabs [S][Mpad][Npad] in the layout of the kernels above (column = tap * Cin + ci, bias column 9 Cin):
// k_wgrad_reduce is shared.  (A first form with 64 x 64 tiles and nine accumulators per wave ran 30 % SLOWER than the GEMM it replaced:
// a workgroup's slab is 9 x its tile, 512 workgroups wrote 75 MB of slabs per layer against 13 MB -- profiles/r04_wgrad_taps.txt.)
// ds_read_b32 banks are per 32-lane half (lanes l and l + 32 never conflict): rows of 32 floats are conflict-free.
// X gather: zero or reflection padding, optional x2 nearest upsample of src0 and a channel-concatenated src1 (a 32-channel tile lies in
// one source: C1 % 32 == 0).
// ---------------------------------------------------------------------------------------------------------------------
struct TapWgradArgs {
    const float* dz;       // (B, H, W, Cout)
    const float* src0;     // (B, H / up, W / up, C1)
    const float* src1;     // (B, H, W, Cin - C1) or NULL
    float* slabs;          // [S][Mpad][Npad]
    int B, H, W, Cin, C1, up, Cout, reflect, has_bias, Mpad, Npad;
    int ptx, pty;          // 8 x 8 patches per image along x / y
    int nchunks, cps;      // chunks (patch of one image) in total / per slice
};

#define TAP_NT 192
template <bool BIAS>
__global__ __launch_bounds__(TAP_NT) void k_wgrad3x3_taps(TapWgradArgs a) {
    __shared__ float dzs[64 * 32];                                  // [pixel][co]
    __shared__ float xs[100 * 32];                                  // [halo pixel][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh_ = lane >> 5, l32 = lane & 31;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const bool first = ci0 < a.C1;                                  // this tile's 32 input channels: of src0 (behind the upsample) or of src1
    const int sh = first ? (a.up >> 1) : 0, Cs = first ? a.C1 : a.Cin - a.C1, cs0 = first ? ci0 : ci0 - a.C1;
    const int Hs = a.H >> sh, Ws = a.W >> sh;
    const float* sbase = first ? a.src0 : a.src1;
    const int c_begin = blockIdx.z * a.cps, c_end = min(c_begin + a.cps, a.nchunks);
    const bool want_bias = BIAS && blockIdx.x == 0 && wave == 0;
    f16v acc[3], accb;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const float one = (l32 == 0) ? 1.f : 0.f;
    f4v rz[3], rx[5];
    auto load_chunk = [&](int g) {
        const int b = g / (a.pty * a.ptx), rem = g - b * (a.pty * a.ptx), py = rem / a.ptx, px = rem - py * a.ptx;
        const int y0 = py * 8, x0 = px * 8;
        const float* zb = a.dz + (int64_t)b * a.H * a.W * a.Cout + co0;
        const float* sb = sbase + (int64_t)b * Hs * Ws * Cs + cs0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + TAP_NT * i, pix = f >> 3, q = f & 7, y = y0 + (pix >> 3), x = x0 + (pix & 7);
            const bool ok = f < 512 && y < a.H && x < a.W;
            rz[i] = ok ? *(const f4v*)(zb + ((int64_t)y * a.W + x) * a.Cout + q * 4) : (f4v){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int f = tid + TAP_NT * i;
            f4v v = {0.f, 0.f, 0.f, 0.f};
            if (f < 800) {
                const int pix = f >> 3, q = f & 7;
                int y = y0 - 1 + pix / 10, x = x0 - 1 + pix % 10;
                bool ok = true;
                if (a.reflect) {                                    // (beyond the image's far edge by more than the pad: the patch overhangs, dZ is 0 there)
                    y = min(max(reflect1(y, a.H), 0), a.H - 1);
                    x = min(max(reflect1(x, a.W), 0), a.W - 1);
                } else {
                    ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
                }
                if (ok) v = *(const f4v*)(sb + ((int64_t)(y >> sh) * Ws + (x >> sh)) * Cs + q * 4);
            }
            rx[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = tid + TAP_NT * i;
            if (f < 512) *(f4v*)&dzs[f * 4] = rz[i];                 // [pixel f >> 3][quad f & 7]: contiguous
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int f = tid + TAP_NT * i;
            if (f < 800) *(f4v*)&xs[f * 4] = rx[i];
        }
    };
    if (c_begin < c_end) load_chunk(c_begin);
    const float* ap = dzs + kh_ * 32 + l32;                             // A[m = co][k = pixel]: dZ[pixel 2 ks + (lane >> 5)][co]
    const float* bp = xs + (kh_ + wave * 10) * 32 + l32;                // B[k = pixel][n = ci]: X[halo pixel of (pixel + (kh = wave, kw))][ci]
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();                                                // the previous chunk's fragments are consumed
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);                           // in flight under the MFMAs below
        // fragments are read QB k-steps (QB dZ + 3 QB X values per lane) ahead of their 3 QB MFMAs, two register sets: the reads of batch
        // q + 1 are issued before the MFMAs of batch q.  hipcc left alone emits read -> s_waitcnt lgkmcnt(0) -> 1..3 MFMAs per k-step (the
        // LDS round trip paid 32 times per chunk); reading half a chunk ahead costs 183 registers, two waves per SIMD and a second round of
        // workgroups; QB = 4 keeps three waves per SIMD -- profiles/r04_wgrad_taps.txt
        constexpr int QB = 4, NQ = 32 / QB;
        float av[2][QB], bv[2][QB][3];
        auto read_q = [&](int set, int qt) {
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const int ks = QB * qt + j, q0 = (ks >> 2) * 10 + 2 * (ks & 3);     // halo index of the k-step's first pixel at tap (0, 0)
                av[set][j] = ap[2 * ks * 32];
#pragma unroll
                for (int t = 0; t < 3; ++t) bv[set][j][t] = bp[(q0 + t) * 32];
            }
        };
        auto pin_q = [&](int set) {              // the set's fragments exist in registers HERE (the scheduler may not sink their reads past this point)
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                asm volatile("" : "+v"(av[set][j]));
#pragma unroll
                for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(bv[set][j][t]));
            }
        };
        read_q(0, 0);
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) {
            if (qt + 1 < NQ) read_q((qt + 1) & 1, qt + 1);
            __builtin_amdgcn_sched_barrier(0);
            pin_q(qt & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < QB; ++j) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[qt & 1][j], bv[qt & 1][j][t], acc[t], 0, 0, 0);
                if (BIAS && want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(av[qt & 1][j], one, accb, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* slab = a.slabs + (int64_t)blockIdx.z * a.Mpad * a.Npad;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = co0 + (r & 3) + 8 * (r >> 2) + 4 * kh_;
            slab[(int64_t)m * a.Npad + (wave * 3 + t) * a.Cin + ci0 + l32] = acc[t][r];
        }
    if (BIAS && want_bias && l32 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(int64_t)(co0 + (r & 3) + 8 * (r >> 2) + 4 * kh_) * a.Npad + 9 * a.Cin] = accb[r];
    }
}

// sum the S slabs and scatter to dW (Cout,Cin,KH,KW) [accumulating when `accumulate`] and the bias gradient, in ONE launch:
// a thread owns 4 consecutive columns of one row (16-byte loads, Npad % 4 == 0; a wave reads 1 KB runs of a slab), the ZL waves of
// a workgroup take the slabs z = w, w + ZL, ... with 4 independent loads in flight, and the ZL partial sums are combined through
// LDS in the fixed order ((p0 + p1) + p2) + ... -- bitwise reproducible.  scale (may be NULL): per-output-channel factor applied
// to the sum (the folded BatchNorm scale, when the GEMM ran on dA instead of dZ = dA * scale).  Measured alone on this network's
// shapes (scratch/reduce_bench.hip): 4-8 us against 8-11 us for the scalar-load form on the many-slab layers, equal on the
// few-slab ones (17 us at 512x512x3x3, where the 9.4 MB scatter dominates).
// one group of 64 quads: lane e of wave w (of ZL) -- `part` is the group's [ZL][64] LDS scratch; every thread of the workgroup calls this the same
// number of times (two barriers inside)
template <int ZL>
__device__ __forceinline__ void wgrad_reduce_group(const float* __restrict__ slabs, int S, int Mpad, int Npad, int Cout, int Cin, int KH, int KW, int has_bias,
                                                   float* __restrict__ dw, float* __restrict__ dbias, int accumulate, const float* __restrict__ scale,
                                                   int64_t base, int e, int w, f4v (*part)[64]) {
    const int Kconv = KH * KW * Cin, Ng = Kconv + (has_bias ? 1 : 0);
    const int nq = (Ng + 3) / 4;
    const int64_t totalq = (int64_t)Cout * nq;
    const int64_t slab_q = (int64_t)Mpad * Npad / 4;
    const f4v zero = {0.f, 0.f, 0.f, 0.f};
    const int64_t q = base + e;
    const bool on = q < totalq;
    const int m = on ? (int)(q / nq) : 0, n = on ? (int)(q - (int64_t)m * nq) * 4 : 0;
    const f4v* src = (const f4v*)(slabs + (int64_t)m * Npad + n);
    f4v sacc = zero;
    int z = w;
    for (; z + 3 * ZL < S; z += 4 * ZL) {
        f4v v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = on ? src[(int64_t)(z + ZL * j) * slab_q] : zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc += v[j];
    }
    for (; z < S; z += ZL) sacc += on ? src[(int64_t)z * slab_q] : zero;
    part[w][e] = sacc;
    __syncthreads();
    if (w == 0 && on) {
        f4v t = part[0][e];
#pragma unroll
        for (int j = 1; j < ZL; ++j) t += part[j][e];
        if (scale) t *= scale[m];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int nn = n + k;
            if (nn < Kconv) {
                const int tap = nn / Cin, ci = nn - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
                float* d = dw + (((int64_t)m * Cin + ci) * KH + kh) * KW + kw;
                *d = accumulate ? *d + t[k] : t[k];
            } else if (nn == Kconv && dbias) {
                dbias[m] = accumulate ? dbias[m] + t[k] : t[k];
            }
        }
    }
    __syncthreads();
}

template <int ZL>
__global__ __launch_bounds__(64 * ZL) void k_wgrad_reduce(const float* __restrict__ slabs, int S, int Mpad, int Npad, int Cout, int Cin,
                                                         int KH, int KW, int has_bias, float* __restrict__ dw, float* __restrict__ dbias,
                                                         int accumulate, const float* __restrict__ scale) {
    __shared__ f4v part[ZL][64];
    const int64_t totalq = (int64_t)Cout * ((KH * KW * Cin + (has_bias ? 1 : 0) + 3) / 4);
    for (int64_t base = (int64_t)blockIdx.x * 64; base < totalq; base += (int64_t)gridDim.x * 64)
        wgrad_reduce_group<ZL>(slabs, S, Mpad, Npad, Cout, Cin, KH, KW, has_bias, dw, dbias, accumulate, scale, base, threadIdx.x & 63, threadIdx.x >> 6, part);
}

// The slab reductions of MANY layers in one launch (e2e_wgrad_reduce_batched): a backward pass of the depth network ends ~30 backward-weight GEMMs
// with a reduction launch of 5 - 15 us each, most of it launch latency and an almost empty GPU (a layer's dW is 10^4 - 10^6 floats) -- 0.29 ms of
// a 4.9 ms step.  Deferred to the end of the pass they are one grid that fills the chip.  A work item is one workgroup-load of a layer's quads:
// 64 quads shared by 8 waves (zl = 8, layers with >= 8 slabs) or 4 x 64 quads by 2 waves each (zl = 2), i.e. the SAME association of the sum as
// the per-layer launch -- the results are bit-identical to e2e_conv2d_bwd_weight_scaled.
__global__ __launch_bounds__(512) void k_wgrad_reduce_batched(const e2e_wgrad_reduce_desc* __restrict__ d, int n, long long total_items) {
    __shared__ f4v part[8][64];
    const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long item = blockIdx.x; item < total_items; item += gridDim.x) {
        int lo = 0, hi = n - 1;                                   // last descriptor whose first_item <= item (workgroup-uniform)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (d[mid].first_item <= item) lo = mid; else hi = mid - 1;
        }
        const e2e_wgrad_reduce_desc D = d[lo];
        const long long local = item - D.first_item;
        if (D.zl == 8)
            wgrad_reduce_group<8>(D.slabs, D.S, D.Mpad, D.Npad, D.Cout, D.Cin, D.KH, D.KW, D.has_bias, D.dw, D.dbias, D.accumulate, D.scale, local * 64, e, w, part);
        else
            wgrad_reduce_group<2>(D.slabs, D.S, D.Mpad, D.Npad, D.Cout, D.Cin, D.KH, D.KW, D.has_bias, D.dw, D.dbias, D.accumulate, D.scale,
                                  (local * 4 + (w >> 1)) * 64, e, w & 1, part + 2 * (w >> 1));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// helpers: weight layout transforms, activation backward, reflection-pad / upsample / concat adjoint
// ---------------------------------------------------------------------------------------------------------------------
// W (Cout,Cin,KH,KW) -> Wf [(kh,kw,ci)][ldf] (forward B operand) and Wb [(kh,kw,co)][ldb] (backward-data B operand)
__global__ __launch_bounds__(256) void k_weight_layouts(const float* __restrict__ w, int Cout, int Cin, int KH, int KW,
                                                        float* __restrict__ wf, int ldf, float* __restrict__ wb, int ldb) {
    const int64_t total = (int64_t)Cout * Cin * KH * KW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH); t /= KH;
        const int ci = (int)(t % Cin); t /= Cin;
        const int co = (int)t;
        const float v = w[i];
        if (wf) wf[((int64_t)(kh * KW + kw) * Cin + ci) * ldf + co] = v;
        if (wb) wb[((int64_t)(kh * KW + kw) * Cout + co) * ldb + ci] = v;
    }
}

// all layers of a network in ONE launch (after an optimiser step every layout is stale): blockIdx.y = layer, descriptor
// table of 10 int64 per layer on the device: {w, w_fwd, w_bwd, Cout, Cin, KH, KW, ld_fwd, ld_bwd, bwd_scale (float* or 0)}
__global__ __launch_bounds__(256) void k_weight_layouts_batched(const long long* __restrict__ desc) {
    const long long* d = desc + (int64_t)blockIdx.y * 10;
    const float* __restrict__ w = (const float*)d[0];
    float* __restrict__ wf = (float*)d[1];
    float* __restrict__ wb = (float*)d[2];
    const int Cout = (int)d[3], Cin = (int)d[4], KH = (int)d[5], KW = (int)d[6], ldf = (int)d[7], ldb = (int)d[8];
    // optional per-output-channel factor folded into the BACKWARD layout only (a folded BatchNorm scale: dX = (dA * scale) W^T is
    // computed as dA (scale W)^T, so the backward GEMM reads dA directly -- e2ehip.netplan)
    const float* __restrict__ bsc = (const float*)d[9];
    const int T = KH * KW;
    constexpr int TMAX = 9, TS = 32;
    __shared__ float tile[TMAX][TS + 1][TS + 1];           // tap stride 33 * 33 = 1 mod 32 banks: the staging store walks taps fastest (PMC: 54 % conflicts with [TS][TS + 1])
    if (T > TMAX) {                                           // the 7x7 stem (Cin = 3): element-wise, it is tiny
        const int64_t total = (int64_t)Cout * Cin * T;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int tap = (int)(i % T);
            const int64_t r = i / T;
            const int ci = (int)(r % Cin), co = (int)(r / Cin);
            const float v = w[i];
            if (wf) wf[((int64_t)tap * Cin + ci) * ldf + co] = v;
            if (wb) wb[((int64_t)tap * Cout + co) * ldb + ci] = bsc ? v * bsc[co] : v;
        }
        return;
    }
    // 32 (cout) x 32 (cin) x T tiles through LDS: the read is contiguous along (ci, tap) of one cout row, the two writes
    // are contiguous along cout (w_fwd) and along cin (w_bwd) -- a direct scatter runs at ~1 TB/s, 175 us per step
    const int nct = (Cout + TS - 1) / TS, nit = (Cin + TS - 1) / TS;
    for (int t = blockIdx.x; t < nct * nit; t += gridDim.x) {
        const int co0 = (t / nit) * TS, ci0 = (t % nit) * TS;
        const int nci = min(TS, Cin - ci0), nco = min(TS, Cout - co0);
        const int row = nci * T;                              // contiguous floats of one cout row inside the tile
        for (int e = threadIdx.x; e < nco * row; e += 256) {
            const int co_l = e / row, rem = e - co_l * row;
            const int ci_l = rem / T, tap = rem - ci_l * T;
            tile[tap][co_l][ci_l] = w[((int64_t)(co0 + co_l) * Cin + ci0) * T + rem];
        }
        __syncthreads();
        if (wf)
            for (int e = threadIdx.x; e < T * nci * nco; e += 256) {
                const int co_l = e % nco, r = e / nco;
                const int ci_l = r % nci, tap = r / nci;
                wf[((int64_t)tap * Cin + ci0 + ci_l) * ldf + co0 + co_l] = tile[tap][co_l][ci_l];
            }
        if (wb)
            for (int e = threadIdx.x; e < T * nco * nci; e += 256) {
                const int ci_l = e % nci, r = e / nci;
                const int co_l = r % nco, tap = r / nco;
                wb[((int64_t)tap * Cout + co0 + co_l) * ldb + ci0 + ci_l] = bsc ? tile[tap][co_l][ci_l] * bsc[co0 + co_l] : tile[tap][co_l][ci_l];
            }
        __syncthreads();
    }
}

// dZ (+)= dY * act'(Y) * scale[c]     (Y = the activation's OUTPUT; ELU' = y+1 for y<=0; DISP' = (y-.01)(1-(y-.01)/10));
// accumulate != 0 adds to dz (a gradient that a second consumer contributes to, e.g. the residual branch of a BasicBlock)
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ scale,
                                                 float* __restrict__ dz, int64_t n, int C, int act, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float g = dy[i];
        const float v = y[i];
        if (act == ACT_RELU) g = v > 0.f ? g : 0.f;
        else if (act == ACT_ELU) g = v > 0.f ? g : g * (v + 1.f);
        else if (act == ACT_DISP) { const float s = (v - 0.01f) * 0.1f; g = g * 10.f * s * (1.f - s); }
        if (scale) g *= scale[i % C];
        dz[i] = accumulate ? dz[i] + g : g;
    }
}

// Adjoint of the forward gather of one convolution input: dXp (B,Hp,Wp,Cin) over the padded full-resolution domain
// (Hp = Hs + 2*pp with pp = 1 for reflection padding, 0 otherwise) -> d src0 (B,Hs/up,Ws/up,C1) [+ d src1 (B,Hs,Ws,C2)].
// Every destination element GATHERS its contributors in a fixed order (no atomics): the reflect-pad copies of a pixel
// and, for src0, the up x up block of full-resolution pixels that read it.
__global__ __launch_bounds__(256) void k_gather_adjoint(const float* __restrict__ dxp, int B, int Hs, int Ws, int Cin, int C1, int up,
                                                        int pp, float* __restrict__ d0, float* __restrict__ d1, int acc0, int acc1,
                                                        const float* __restrict__ x0, int act0, const float* __restrict__ x1, int act1) {
    const int Hp = Hs + 2 * pp, Wp = Ws + 2 * pp, C2 = Cin - C1, Hl = Hs / up, Wl = Ws / up;
    const int64_t n0 = (int64_t)B * Hl * Wl * C1, n1 = (int64_t)B * Hs * Ws * C2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += (int64_t)gridDim.x * 256) {
        const bool first = i < n0;
        int64_t t = first ? i : i - n0;
        const int Cx = first ? C1 : C2, Hx = first ? Hl : Hs, Wx = first ? Wl : Ws, u = first ? up : 1;
        const int c = (int)(t % Cx); t /= Cx;
        const int x = (int)(t % Wx); t /= Wx;
        const int y = (int)(t % Hx); t /= Hx;
        const int b = (int)t;
        const int cc = first ? c : C1 + c;
        float s = 0.f;
        for (int dy = 0; dy < u; ++dy)
            for (int dx = 0; dx < u; ++dx) {
                const int fy = y * u + dy, fx = x * u + dx;                 // full-resolution pixel
                // padded positions that map to (fy,fx): itself (+pp) and its reflections across the borders
                int ys[3] = {fy + pp, -1, -1}, xs[3] = {fx + pp, -1, -1};
                if (pp) {
                    if (fy == 1) ys[1] = 0;
                    if (fy == Hs - 2) ys[2] = Hp - 1;
                    if (fx == 1) xs[1] = 0;
                    if (fx == Ws - 2) xs[2] = Wp - 1;
                }
                for (int iy = 0; iy < 3; ++iy)
                    for (int ix = 0; ix < 3; ++ix)
                        if (ys[iy] >= 0 && xs[ix] >= 0) s += dxp[(((int64_t)b * Hp + ys[iy]) * Wp + xs[ix]) * Cin + cc];
            }
        float* d = first ? d0 + i : d1 + (i - n0);
        const int acc = first ? acc0 : acc1;
        if (first ? act0 : act1) s *= act_deriv(first ? x0[i] : x1[i - n0], first ? act0 : act1);     // -> gradient wrt the producer's pre-activation
        *d = acc ? *d + s : s;
    }
}

// the same for channel counts that are multiples of 4 (every layer of this network): a thread owns a 16-byte channel quad, indices
// are 32-bit (the host checks the extents) -- the scalar form spends its time in three 64-bit divisions per float (measured
// 23 us -> see profiles/r02_*kernel_stats* for an up(0,1)-sized call)
__global__ __launch_bounds__(256) void k_gather_adjoint4(const float* __restrict__ dxp, int B, int Hs, int Ws, int Cin, int C1, int up,
                                                         int pp, float* __restrict__ d0, float* __restrict__ d1, int acc0, int acc1,
                                                         const float* __restrict__ x0, int act0, const float* __restrict__ x1, int act1) {
    const int Hp = Hs + 2 * pp, Wp = Ws + 2 * pp, C2 = Cin - C1, Hl = Hs / up, Wl = Ws / up;
    const unsigned q0 = (unsigned)B * Hl * Wl * (C1 >> 2), q1 = (unsigned)B * Hs * Ws * (C2 >> 2);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < q0 + q1; i += gridDim.x * 256u) {
        const bool first = i < q0;
        unsigned t = first ? i : i - q0;
        const unsigned Cq = (unsigned)(first ? C1 : C2) >> 2, Hx = first ? Hl : Hs, Wx = first ? Wl : Ws;
        const int u = first ? up : 1;
        const unsigned cq = t % Cq; t /= Cq;
        const unsigned x = t % Wx; t /= Wx;
        const unsigned y = t % Hx;
        const unsigned b = t / Hx;
        const int cc = (first ? 0 : C1) + (int)cq * 4;
        f4v s = {0.f, 0.f, 0.f, 0.f};
        for (int dy = 0; dy < u; ++dy)
            for (int dx = 0; dx < u; ++dx) {
                const int fy = (int)y * u + dy, fx = (int)x * u + dx;
                int ys[3] = {fy + pp, -1, -1}, xs[3] = {fx + pp, -1, -1};
                if (pp) {
                    if (fy == 1) ys[1] = 0;
                    if (fy == Hs - 2) ys[2] = Hp - 1;
                    if (fx == 1) xs[1] = 0;
                    if (fx == Ws - 2) xs[2] = Wp - 1;
                }
#pragma unroll
                for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 3; ++ix)
                        if (ys[iy] >= 0 && xs[ix] >= 0) s += *(const f4v*)(dxp + (((int64_t)b * Hp + ys[iy]) * Wp + xs[ix]) * Cin + cc);
            }
        const int64_t e = (int64_t)(first ? i : i - q0) * 4;
        const int act = first ? act0 : act1;
        if (act) {
            const f4v xv = *(const f4v*)((first ? x0 : x1) + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] *= act_deriv(xv[k], act);
        }
        f4v* d = (f4v*)((first ? d0 : d1) + e);
        if (first ? acc0 : acc1) s += *d;
        *d = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 1-channel disparity head (networks.py:271-272,289-290): Conv3x3(reflect) 16 -> 1 + 10*sigmoid+0.01.  One output
// column cannot fill an MFMA tile; it is a 144-tap dot product per pixel -- HBM-bound VALU kernels.
// ---------------------------------------------------------------------------------------------------------------------
#define HC 16   // input channels of the head this kernel family is specialised for (num_ch_dec[0], networks.py:253)
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ w /*(1,16,3,3)*/,
                                                  const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int act) {
    __shared__ float sw[9 * HC];
    for (int i = threadIdx.x; i < 9 * HC; i += 256) {        // (ci,kh,kw) -> [tap][ci]
        const int ci = i / 9, tap = i - ci * 9;
        sw[tap * HC + ci] = w[i];
    }
    __syncthreads();
    // FOUR lanes per pixel, one channel quad each: a wave's load of one tap is 16 pixels x 64 contiguous bytes = 1 KB in eight cache lines.
    // (One lane per pixel read its 64 bytes as four float4s 64 bytes apart from its neighbours': every load instruction touched 64 lines, 2304
    // line look-ups per 64 pixels against 288 now -- the kernel ran at 26 us on 39 MB, bound by the L1's tag rate, not by memory.)
    const int64_t N = (int64_t)B * H * W;
    const int q = threadIdx.x & 3;
    for (int64_t n0 = (int64_t)blockIdx.x * 64; n0 < N; n0 += (int64_t)gridDim.x * 64) {          // 64 pixels per workgroup and pass: uniform trip count
        const int64_t n = n0 + (threadIdx.x >> 2);
        const bool on = n < N;
        const int64_t nn = on ? n : N - 1;
        const int b = (int)(nn / ((int64_t)H * W));
        const int r = (int)(nn - (int64_t)b * H * W), h = r / W, ww = r - h * W;
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ys = reflect1(h + kh - 1, H), xs = reflect1(ww + kw - 1, W);
                const f4v v = *(const f4v*)(x + (((int64_t)b * H + ys) * W + xs) * HC + q * 4);
                const float* wt = &sw[(kh * 3 + kw) * HC + q * 4];
                acc = fmaf(v[0], wt[0], fmaf(v[1], wt[1], fmaf(v[2], wt[2], fmaf(v[3], wt[3], acc))));
            }
        acc += __shfl_xor(acc, 1, 64);                        // the four quads of a pixel: fixed tree
        acc += __shfl_xor(acc, 2, 64);
        if (on && q == 0) y[n] = apply_act(acc + (bias ? bias[0] : 0.f), act);
    }
}

// d/dx: every input pixel gathers the (reflect-aware) outputs that read it:  dx[p,ci] = sum_{q, tap: src(q,tap)=p} dz[q] w[tap,ci]
__global__ __launch_bounds__(256) void k_head_bwd_data(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                                                       int B, int H, int W, const float* __restrict__ xin, int dact) {
    __shared__ float sw[9 * HC];
    for (int i = threadIdx.x; i < 9 * HC; i += 256) {
        const int ci = i / 9, tap = i - ci * 9;
        sw[tap * HC + ci] = w[i];
    }
    __syncthreads();
    // four lanes per pixel, one channel quad each (see k_head_fwd): the 16-channel row of a pixel is read (x_in) and written (dx) as one 64-byte
    // piece by four neighbouring lanes; the nine scalar dZ gathers are the same for the four and hit the same line
    const int64_t N = (int64_t)B * H * W;
    const int cq = threadIdx.x & 3;
    for (int64_t n = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2); n < N; n += (int64_t)gridDim.x * 64) {
        const int b = (int)(n / ((int64_t)H * W));
        const int r = (int)(n - (int64_t)b * H * W), h = r / W, ww = r - h * W;
        f4v acc = {0.f, 0.f, 0.f, 0.f};
        // the outputs q that read input pixel p through tap (kh, kw): padded position q + (kh - 1, kw - 1) must map to p.  Along one
        // axis that is q = p - k + 1 (the direct reader) and, for p one pixel away from a border only, the reader that reaches p through
        // the reflection (p = 1: padded -1, i.e. q = -k; p = n - 2: padded n, i.e. q = n - k + 1).  At most 2 x 2 readers per tap, and
        // exactly one for interior pixels -- the first version searched a 5 x 5 neighbourhood for every pixel of a border WAVE.
        const int ry = (h == 1) ? 0 : ((h == H - 2) ? H + 1 : -100);             // reflected reader: q = ry - kh (invalid when out of range)
        const int rx = (ww == 1) ? 0 : ((ww == W - 2) ? W + 1 : -100);
        const float* gz = dz + (int64_t)b * H * W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* wt = &sw[(kh * 3 + kw) * HC + cq * 4];
                const int qy0 = h - kh + 1, qx0 = ww - kw + 1, qy1 = ry - kh, qx1 = rx - kw;
                const bool y0 = qy0 >= 0 && qy0 < H, x0 = qx0 >= 0 && qx0 < W, y1 = qy1 >= 0 && qy1 < H, x1 = qx1 >= 0 && qx1 < W;
                float g = (y0 && x0) ? gz[(int64_t)qy0 * W + qx0] : 0.f;
                if (y1 | x1) {                                                  // border pixels only (divergence is a handful of lanes per row)
                    if (y0 && x1) g += gz[(int64_t)qy0 * W + qx1];
                    if (y1 && x0) g += gz[(int64_t)qy1 * W + qx0];
                    if (y1 && x1) g += gz[(int64_t)qy1 * W + qx1];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(g, wt[c], acc[c]);
            }
        if (dact) {                                              // -> gradient wrt the pre-activation of the layer that produced x
            const f4v xv = *(const f4v*)(xin + n * HC + cq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= act_deriv(xv[e], dact);
        }
        *(f4v*)(dx + n * HC + cq * 4) = acc;
    }
}

// d/dw, d/dbias: per-workgroup partial sums of dz[p] * x[src(p,tap), ci] (145 values), fixed-order second stage.
// Four lanes per pixel, one channel quad each, as in the other two head kernels: a lane multiplies its pixel's dZ into the nine taps' quads (nine
// fully coalesced loads) and keeps 36 running sums; at the end the 16 pixel slots of a wave are folded with four fixed xor-shuffles per sum and
// the four waves through LDS in wave order.  (The previous form gave every one of the 145 COLUMNS a thread that walked the workgroup's pixels
// one after the other -- a dependent chain of 300 loads, 44 us for 39 MB.)
__global__ __launch_bounds__(256) void k_head_bwd_weight(const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ partials,
                                                         int B, int H, int W) {
    __shared__ float red[4][9 * HC + 1];
    const int64_t N = (int64_t)B * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = threadIdx.x & 3, ps = threadIdx.x >> 2;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t n0 = (int64_t)blockIdx.x * per, n1 = (n0 + per < N) ? n0 + per : N;
    float acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.f;
    float sb = 0.f;
    for (int64_t n = n0 + ps; n < n1; n += 64) {
        const float g = dz[n];
        const int b = (int)(n / ((int64_t)H * W));
        const int r = (int)(n - (int64_t)b * H * W), h = r / W, ww = r - h * W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ys = reflect1(h + kh - 1, H), xs = reflect1(ww + kw - 1, W);
                const f4v v = *(const f4v*)(x + (((int64_t)b * H + ys) * W + xs) * HC + cq * 4);       // < 2^31 elements (checked by the host)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[(kh * 3 + kw) * 4 + c] = fmaf(g, v[c], acc[(kh * 3 + kw) * 4 + c]);
            }
        sb += g;
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        acc[k] = v;
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) sb += __shfl_xor(sb, o, 64);
    if (lane < 4) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[wave][t * HC + lane * 4 + c] = acc[t * 4 + c];
        if (lane == 0) red[wave][9 * HC] = sb;
    }
    __syncthreads();
    if (threadIdx.x < 9 * HC + 1)
        partials[(int64_t)blockIdx.x * (9 * HC + 1) + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one workgroup per column of the 145-vector: 256 threads add the per-workgroup partials (fixed assignment, fixed
// combine order => deterministic)
__global__ __launch_bounds__(256) void k_head_wreduce(const float* __restrict__ partials, int nparts, float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float red[4];
    const int col = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[(int64_t)i * (9 * HC + 1) + col];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (col == 9 * HC) { if (dbias) dbias[0] = t; }
        else { const int tap = col / HC, ci = col - tap * HC; dw[ci * 9 + tap] = t; }       // (1,16,3,3)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
static inline int egrid(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// K-chunk depth: 32 when the channel count allows (a chunk never straddles a tap; a row's chunk is one 128-byte line)
#define GEMM_LAUNCH(WM, WN, TM, TN, GRID)                                                                                   \
    do {                                                                                                                    \
        if (cb == 32) hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 4, TR, 32>), GRID, dim3(64 * WM * WN), 0, st, a);     \
        else hipLaunchKernelGGL((k_conv_gemm<WM, WN, TM, TN, 4, TR, 16>), GRID, dim3(64 * WM * WN), 0, st, a);              \
    } while (0)

// A GEMM decomposition: rows x columns of the workgroup tile and the number of K slices (split-K); S < 0: stream-K on -S persistent
// workgroups (k_conv_gemm_sk, 64x64 tiles).  bm == 0: not forced.  Tuning choices travel PER CALL (the *_tuned entry points); the
// library keeps no mutable state.
struct GemmCfg { int bm, bn, S; };
#define WGRAD_TARGET 1024                   // workgroups a backward-weight launch aims at by default

// waves of a tile shape (one 32x32xTMxTN accumulator block per wave) -- 0: unsupported shape
static int tile_waves(int bm, int bn) {
    if ((bm == 64 && bn == 64) || (bm == 128 && bn == 64) || (bm == 128 && bn == 128) || (bm == 128 && bn == 32) || (bm == 32 && bn == 128)) return 4;
    if ((bm == 32 && bn == 64) || (bm == 64 && bn == 32)) return 2;
    if (bm == 32 && bn == 32) return 1;
    return 0;
}

// The decomposition of a GEMM of rows x cols x K.  Calibrated on tools/gemm_tune.py (profiles/r02_gemm_tune*.txt: every tile
// family x 1..16 K slices timed on every layer shape of the 480x640 network, forward and backward-data):
//   * 64x64 tiles (4 waves, one 32x32 block each) are within 3 % of the best family wherever the GEMM has more than 32
//     columns; thin GEMMs take 128x32 (64x32 when that leaves fewer than ~2 workgroups per CU);
//   * the number of K slices that is fastest puts ~460 workgroups on the 256 CUs (all of them resident at once, ~1.8 per CU):
//     S = round(460 / tiles), e.g. layer2 (300 tiles) 57 us unsplit -> 41 us at S = 3, upconv(4,0) (40 tiles) 34 -> 26 us at
//     S = 12 -- as long as a slice keeps at least 128 of K.  Finer tiles WITHOUT split-K (32x64, 32x32: 1200-2400 workgroups)
//     do not pay: the kernels are not bound by tile quantisation but by the phases every wave of a launch goes through at
//     the same time (operand set-up, epilogue), which more but smaller workgroups do not shorten.
static GemmCfg choose_cfg(int64_t rows, int cols, int K, int cb, bool allow_split, GemmCfg force = GemmCfg{0, 0, 0}) {
    if (force.bm) { GemmCfg f = force; if (!allow_split && f.S > 1) f.S = 1; if (f.bm == 128 && f.bn == 128 && cb != 16) f.bn = 64; return f; }
    GemmCfg c = {64, 64, 1};
    // round 3 (branch-free loaders, profiles/r03_gemm_tune_final.txt): with >= 128 columns the 32 x 128 tile (four waves side by side on ONE
    // 32-row A block) wins 5-10 % over 64 x 64 at the same slice count -- the gathered A operand is the expensive one to stage, and this
    // shape stages half as much of it per MFMA (layer3 41.1 -> 37.3 us, layer4 41.0 -> 37.3, up(3,1) 67.4 -> 62.9, up(2,1) backward-data
    // 61.5 -> 56.8 = 99.7 TF/s)
    if (cols >= 128) { c.bm = 32; c.bn = 128; }
    if (cols <= 32) {
        c.bm = 128; c.bn = 32;
        if (((rows + 127) / 128) * ((cols + 31) / 32) < 600) c.bm = 64;
    }
    if (allow_split && cols > 32) {
        const int64_t tiles = ((rows + c.bm - 1) / c.bm) * ((cols + c.bn - 1) / c.bn);
        const int nchunks = (K + cb - 1) / cb;
        // Round 2's rule aimed at ~950 workgroups whatever the depth.  With the round-3 loops a workgroup is cheaper to run and dearer to
        // start and to reduce (prologue + slab round trip are now a larger share of it), and the table (tools/gemm_tune.py,
        // profiles/r03_gemm_tune_final.txt) puts the optimum at a slice DEPTH instead: K shallower than 30 chunks of 32 runs unsplit (18
        // chunks, 300 tiles: 25.8 us against 27.9 split in three), 36 chunks take 3 slices, 72 take 4, 144 take 6 (600 x 512 x 2304:
        // 23.3 us at 3-4 slices, 25.1 at 6, 27.7 at the 8 the old rule picked); 500+ tiles run unsplit (600 tiles: 37 us, 2 slices: 43 us).
        // Guards: at least ~200 workgroups where the depth allows it, never more than ~1300, never below 9 chunks per slice, at most 12 slices.
        const int depth = nchunks * cb / 32;
        int S = tiles >= 500 ? 1 : depth < 30 ? 1 : depth < 60 ? 3 : depth < 120 ? 4 : 6;
        while (S < 12 && tiles * S < 200 && (nchunks / (S + 1)) * cb >= 288) ++S;
        while (S > 1 && tiles * S > 1300) --S;
        while (S > 1 && (nchunks / S) * cb < 288) --S;
        if (S < 1) S = 1;
        // Round 4: the depth rule ignores how the workgroups fall on the 256 CUs.  Up to 768 of them are resident at once and start together,
        // so the launch lasts as long as the CU with the most: 600 workgroups (3 on some CUs) of 18 chunks took 39.7 us where 450 (at most 2)
        // of 24 chunks took 36.3 (layer3), 304 of 18 chunks 28.9 us where 228 (one per CU) of 24 took 23.3 (l4.0.conv1, upconv(3,0)) --
        // but 456 of 24 (two per CU) are level with 228 of 48 (layer4: 35.7 / 36.4 us): a CU runs two workgroups in 1.76x the time of one,
        // three in 2.41x (fitted to those pairs; past 768 the queue evens the load out).  A neighbouring slice count replaces the depth
        // rule's where this model predicts at least 7 % (profiles/r04_gemm_tune_final.txt: every other layer keeps its choice).
        auto model = [&](int s) {
            const double wgs = (double)tiles * s, per_cu = wgs <= 256 ? 1 : wgs <= 512 ? 2 : wgs <= 768 ? 3 : 0;
            const double f = per_cu == 1 ? 1.0 : per_cu == 2 ? 1.76 : per_cu == 3 ? 2.41 : 2.41 * wgs / 768.0;
            return f * ((double)nchunks / s + 6.0) + (s > 1 ? 4.0 : 0.0);
        };
        if (tiles < 500 && S > 1) {
            int best = S;
            for (int s = S - 2; s <= S + 2; ++s) {
                if (s < 2 || s > 12 || s == S || (nchunks / s) * cb < 288 || tiles * s > 1300) continue;
                if (model(s) < 0.93 * model(best)) best = s;
            }
            S = best;
        }
        c.S = S;
    }
    return c;
}

template <bool TR>
static void launch_tile(ConvArgs& a, int cb, GemmCfg c, dim3 g, hipStream_t st) {
    if (c.bm == 64 && c.bn == 64) GEMM_LAUNCH(2, 2, 1, 1, g);
    else if (c.bm == 128 && c.bn == 64) GEMM_LAUNCH(2, 2, 2, 1, g);
    else if (c.bm == 128 && c.bn == 128) hipLaunchKernelGGL((k_conv_gemm<2, 2, 2, 2, 4, TR, 16>), g, dim3(256), 0, st, a);
    else if (c.bm == 128 && c.bn == 32) GEMM_LAUNCH(4, 1, 1, 1, g);
    else if (c.bm == 32 && c.bn == 128) GEMM_LAUNCH(1, 4, 1, 1, g);
    else if (c.bm == 32 && c.bn == 64) GEMM_LAUNCH(1, 2, 1, 1, g);
    else if (c.bm == 64 && c.bn == 32) GEMM_LAUNCH(2, 1, 1, 1, g);
    else GEMM_LAUNCH(1, 1, 1, 1, g);
}

// stream-K launch: G persistent workgroups over tiles x chunks (see k_conv_gemm_sk).  workspace = [flags | slabs]
template <bool TR>
static void launch_streamk(ConvArgs& a, int cb, int G, float* workspace, hipStream_t st) {
    const int64_t Ntot = (int64_t)a.B * a.Hd * a.Wd;
    SkArgs s;
    s.flags = (unsigned*)workspace;
    s.slabs = workspace + SK_FLAG_FLOATS;
    s.tiles_n = (a.Ncols + 63) / 64;
    s.C = a.KH * a.KW * a.Cin / cb;
    s.I = ((Ntot + 63) / 64) * s.tiles_n * (long long)s.C;
    if (G > SK_MAX_G) G = SK_MAX_G;
    if ((long long)G > s.I) G = (int)s.I;                    // every workgroup owns at least one iteration
    s.G = G;
    if (cb == 32) hipLaunchKernelGGL((k_conv_gemm_sk<TR, 32>), dim3(G), dim3(256), 0, st, a, s);
    else hipLaunchKernelGGL((k_conv_gemm_sk<TR, 16>), dim3(G), dim3(256), 0, st, a, s);
}

template <bool TR>
static void launch_gemm(ConvArgs a, int vec, float* workspace_all, GemmCfg force, hipStream_t st) {
    const int64_t Ntot = (int64_t)a.B * a.Hd * a.Wd;
    const int K = a.KH * a.KW * a.Cin;
    const int cb = (vec == 4 && a.Cin % 32 == 0 && (a.C1 == a.Cin || a.C1 % 32 == 0)) ? 32 : 16;
    float* workspace = workspace_all ? workspace_all + SK_FLAG_FLOATS : nullptr;     // split-K slabs live behind the stream-K flag region
    a.ksplit = 1; a.cps = 0; a.slab = workspace;
    a.bytes0 = (int64_t)a.B * (a.Hs / a.up) * (a.Ws / a.up) * a.C1 * 4;
    a.bytes1 = (int64_t)a.B * a.Hs * a.Ws * (a.Cin - a.C1) * 4;
    a.bytesw = (int64_t)K * a.ldw * 4;
    if (vec == 1) {
        dim3 g((unsigned)((Ntot + 127) / 128), (unsigned)((a.Ncols + 63) / 64));
        hipLaunchKernelGGL((k_conv_gemm<4, 1, 1, 2, 1, TR, 16>), g, dim3(256), 0, st, a);
        return;
    }
    if (TR && a.cls) {
        // 4 parity classes x the tiles of the largest class (ceil(Hd/2) x ceil(Wd/2) pixels per image); blockIdx.x & 3 = class.
        // The classes carry 1, 2, 2 and 4 of the 9 taps: price the decomposition on the average (K * 9/16 of a class of Nc rows x 4)
        const int64_t Nc = (int64_t)a.B * ((a.Hd + 1) / 2) * ((a.Wd + 1) / 2);
        GemmCfg c = choose_cfg(Nc * 4, a.Ncols, (K * 9 / 16 + cb - 1) / cb * cb, cb, false, force.S < 0 ? GemmCfg{0, 0, 0} : force);
        if (c.bm == 128 && c.bn == 128) c.bn = 64;
        // The classes of a 3x3 kernel carry 4, 2, 2 and 1 taps: with so few workgroups the launch lasts as long as a 4-tap one
        // (l4.0.conv1: 57 us against 26 us for the forward).  Slice every class by TAP: equal work per workgroup, partial sums in
        // per-(tap, class) slabs, added in tap order by the class epilogue.
        const int64_t total = Ntot * a.Ncols;
        const int64_t cls_wgs = 4 * ((Nc + c.bm - 1) / c.bm) * ((a.Ncols + c.bn - 1) / c.bn);     // measured: 304 workgroups 49.7 -> 38.6 us, 160: 57.6 -> 35.1, 600: 39 -> 40
        if (workspace && cls_wgs < 500 && a.KH == 3 && a.KW == 3 && a.Cin % cb == 0 && a.Ncols % 4 == 0 && total < (1ll << 31) && !a.scale && !a.shift &&
            a.act == ACT_NONE) {
            a.ksplit = 4; a.cps = a.Cin / cb; a.cls_rows = Nc;
            dim3 g((unsigned)(4 * ((Nc + c.bm - 1) / c.bm)), (unsigned)((a.Ncols + c.bn - 1) / c.bn), 4u);
            launch_tile<TR>(a, cb, c, g, st);
            hipLaunchKernelGGL(k_conv_splitk_epilogue_cls, dim3(egrid(total / 4)), dim3(256), 0, st, workspace, Nc, a.B, a.Hd, a.Wd, a.Ncols, a.KH, a.KW, a.off,
                               a.res, a.out, a.xin, a.dact, a.pre);
            return;
        }
        dim3 g((unsigned)(4 * ((Nc + c.bm - 1) / c.bm)), (unsigned)((a.Ncols + c.bn - 1) / c.bn));
        launch_tile<TR>(a, cb, c, g, st);
        return;
    }
    const GemmCfg c = choose_cfg(Ntot, a.Ncols, K, cb, workspace != nullptr, force);
    if (c.S < 0 && workspace && vec == 4 && K % cb == 0 && a.Cin % cb == 0) {      // stream-K (64x64 tiles)
        launch_streamk<TR>(a, cb, -c.S, workspace_all, st);
        return;
    }
    if (c.S > 1) {
        const int nchunks = (K + cb - 1) / cb;
        a.cps = (nchunks + c.S - 1) / c.S;
        const int Sz = (nchunks + a.cps - 1) / a.cps;
        a.ksplit = Sz;
        if (Sz > 1) {
            dim3 g((unsigned)((Ntot + c.bm - 1) / c.bm), (unsigned)((a.Ncols + c.bn - 1) / c.bn), (unsigned)Sz);
            launch_tile<TR>(a, cb, c, g, st);
            const int64_t total = Ntot * a.Ncols;
            if (a.Ncols % 4 == 0 && total < (1ll << 31))
                hipLaunchKernelGGL(k_conv_splitk_epilogue4, dim3(egrid(total / 4)), dim3(256), 0, st, workspace, Sz, (unsigned)(total / 4), a.Ncols, a.scale, a.shift,
                                   a.res, a.out, a.act, a.xin, a.dact, a.pre);
            else
                hipLaunchKernelGGL(k_conv_splitk_epilogue, dim3(egrid(total)), dim3(256), 0, st, workspace, Sz, total, a.Ncols, a.scale, a.shift, a.res, a.out, a.act, a.xin, a.dact, a.pre);
            return;
        }
        a.ksplit = 1; a.cps = 0;
    }
    dim3 g((unsigned)((Ntot + c.bm - 1) / c.bm), (unsigned)((a.Ncols + c.bn - 1) / c.bn));
    launch_tile<TR>(a, cb, c, g, st);
}

extern "C" {

int e2e_conv_weight_layouts(const float* w, int Cout, int Cin, int KH, int KW, float* w_fwd, int ld_fwd, float* w_bwd,
                            int ld_bwd, void* stream) {
    E2E_REQUIRE(w && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && (w_fwd || w_bwd), E2E_ERR_ARG, "e2e_conv_weight_layouts: bad argument");
    E2E_REQUIRE((!w_fwd || (ld_fwd >= Cout && ld_fwd % 4 == 0)) && (!w_bwd || (ld_bwd >= Cin && ld_bwd % 4 == 0)), E2E_ERR_ARG,
                "e2e_conv_weight_layouts: leading dimensions must cover the columns and be multiples of 4");
    hipLaunchKernelGGL(k_weight_layouts, dim3(egrid((int64_t)Cout * Cin * KH * KW)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KH,
                       KW, w_fwd, ld_fwd, w_bwd, ld_bwd);
    E2E_LAUNCH_CHECK("e2e_conv_weight_layouts");
    return E2E_OK;
}

int e2e_conv_weight_layouts_batched(const long long* desc, int nlayers, void* stream) {
    E2E_REQUIRE(desc && nlayers > 0 && nlayers <= 65535, E2E_ERR_ARG, "e2e_conv_weight_layouts_batched: bad argument");
    hipLaunchKernelGGL(k_weight_layouts_batched, dim3(256, nlayers), dim3(256), 0, (hipStream_t)stream, desc);
    E2E_LAUNCH_CHECK("e2e_conv_weight_layouts_batched");
    return E2E_OK;
}

/* floats of workspace a GEMM of `rows` x `cols` with reduction length K may use: the stream-K flag region (SK_FLAG_FLOATS, which the
 * owner zeroes ONCE after allocating) followed by split-K slabs or stream-K slabs, whichever is larger (0: the layer takes neither) */
int64_t e2e_conv2d_splitk_workspace_floats(int64_t rows, int cols, int K);
/* workspace of a backward-data call: the split-K slabs of the stride-1 form, or the (tap, class) slabs of the stride-2 class form */
int64_t e2e_conv2d_bwd_data_workspace_floats(int B, int Hd, int Wd, int cols, int K, int stride) {
    const int64_t plain = e2e_conv2d_splitk_workspace_floats((int64_t)B * Hd * Wd, cols, K);
    if (stride != 2) return plain;
    const int64_t cls = SK_FLAG_FLOATS + (int64_t)16 * B * ((Hd + 1) / 2) * ((Wd + 1) / 2) * cols;      // 4 tap slices x 4 classes x the largest class
    return cls > plain ? cls : plain;
}

int64_t e2e_conv2d_splitk_workspace_floats(int64_t rows, int cols, int K) {
    if (K % 16 != 0) return 0;
    int smax = 1;                                            // the largest slice count any chunk depth would choose
    bool sk = false;
    for (int cb = 16; cb <= 32; cb += 16) {
        const GemmCfg c = choose_cfg(rows, cols, K, cb, true);
        if (c.S > smax) smax = c.S;
        if (c.S < 0) sk = true;
    }
    int64_t n = smax > 1 ? (int64_t)smax * rows * cols : 0;
    if (sk && n < (int64_t)SK_MAX_G * 64 * 64) n = (int64_t)SK_MAX_G * 64 * 64;
    return n ? n + SK_FLAG_FLOATS : 0;
}

/* workspace that covers ANY tile / split-K (<= 16 slices) / stream-K choice of a *_tuned call on a GEMM of rows x cols */
int64_t e2e_conv_tuned_workspace_floats(int64_t rows, int cols) {
    int64_t n = (int64_t)16 * rows * cols;
    if (n < (int64_t)SK_MAX_G * 64 * 64) n = (int64_t)SK_MAX_G * 64 * 64;
    return n + SK_FLAG_FLOATS;
}
int e2e_conv_workspace_flag_floats(void) { return SK_FLAG_FLOATS; }
int e2e_conv_streamk_error_index(void) { return SK_ERR_INDEX; }

/* the decomposition the built-in cost model picks for a GEMM of rows x cols x K at chunk depth cb (host-only query) */
int e2e_conv_gemm_choice(int64_t rows, int cols, int K, int chunk_depth, int allow_split, int* out3) {
    E2E_REQUIRE(out3 && rows > 0 && cols > 0 && K > 0 && (chunk_depth == 16 || chunk_depth == 32), E2E_ERR_ARG, "e2e_conv_gemm_choice: bad argument");
    const GemmCfg c = choose_cfg(rows, cols, K, chunk_depth, allow_split != 0);
    out3[0] = c.bm; out3[1] = c.bn; out3[2] = c.S;
    return E2E_OK;
}

static int conv_fwd_impl(const float* src0, const float* src1, int C1, int up, const float* w_fwd, int ld_fwd, const float* scale,
                         const float* shift, const float* residual, float* out, int B, int Hs, int Ws, int Cin, int Cout, int KH,
                         int KW, int stride, int pad, int pad_mode, int act, float in_sub, float in_mul, float* workspace, GemmCfg force, void* stream) {
    E2E_REQUIRE(src0 && w_fwd && out && B > 0 && Hs > 0 && Ws > 0 && Cin > 0 && Cout > 0, E2E_ERR_ARG, "e2e_conv2d_fwd: bad argument");
    E2E_REQUIRE(up == 1 || up == 2, E2E_ERR_ARG, "e2e_conv2d_fwd: upsample factor must be 1 or 2");
    E2E_REQUIRE(C1 > 0 && C1 <= Cin && (C1 == Cin || src1), E2E_ERR_ARG, "e2e_conv2d_fwd: bad channel split");
    E2E_REQUIRE(pad_mode == 0 || (pad_mode == 1 && pad == 1 && Hs >= 2 && Ws >= 2), E2E_ERR_ARG, "e2e_conv2d_fwd: reflection padding needs pad == 1");
    E2E_REQUIRE(stride == 1 || stride == 2, E2E_ERR_ARG, "e2e_conv2d_fwd: stride must be 1 or 2");
    E2E_REQUIRE(Hs % up == 0 && Ws % up == 0 && ld_fwd % 4 == 0 && ld_fwd >= Cout, E2E_ERR_ARG, "e2e_conv2d_fwd: bad sizes");
    const int vec = (Cin % 16 == 0 && (C1 == Cin || C1 % 16 == 0)) ? 4 : 1;
    E2E_REQUIRE(vec == 4 || (C1 == Cin && up == 1), E2E_ERR_ARG, "e2e_conv2d_fwd: the scalar path (Cin %% 16 != 0) takes a single full-resolution source");
    E2E_REQUIRE((int64_t)B * Hs * Ws * Cin * 4 < (1ll << 30) && (int64_t)KH * KW * Cin * ld_fwd * 4 < (1ll << 31), E2E_ERR_ARG,
                "e2e_conv2d_fwd: activations must stay below 1 GB, weights below 2 GB (32-bit buffer offsets with out-of-range markers)");
    E2E_REQUIRE(vec != 4 || (KH <= 3 && KW <= 3), E2E_ERR_ARG, "e2e_conv2d_fwd: the channel-quad loader takes kernels up to 3 x 3");
    ConvArgs a{};
    a.src0 = src0; a.src1 = src1; a.w = w_fwd; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.C1 = C1; a.up = up;
    a.Hd = (Hs + 2 * pad - KH) / stride + 1; a.Wd = (Ws + 2 * pad - KW) / stride + 1;
    a.Ncols = Cout; a.ldw = ld_fwd; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.pad_mode = pad_mode; a.off = 0; a.act = act;
    a.in_sub = in_sub; a.in_mul = in_mul;
    // the RGB stem: patch kernel
    if (!force.bm && KH == 7 && KW == 7 && stride == 2 && pad == 3 && pad_mode == 0 && Cin == 3 && Cout == 64 && C1 == Cin && up == 1 && !residual && ld_fwd >= 64) {
        StemArgs t{src0, w_fwd, scale, shift, out, B, Hs, Ws, a.Hd, a.Wd, ld_fwd, act, in_sub, in_mul};
        hipLaunchKernelGGL(k_conv7x7_stem, dim3((a.Wd + 31) / 32, (a.Hd + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, t);
        E2E_LAUNCH_CHECK("e2e_conv2d_fwd");
        return E2E_OK;
    }
    // the two 16-output-channel layers at the decoder's last level: patch-in-LDS kernel (k_conv3x3_thin)
    if (!force.bm && KH == 3 && KW == 3 && stride == 1 && pad == 1 && pad_mode == 1 && Cout == 16 && C1 == Cin && !scale && !residual && ld_fwd >= 16 &&
        ((Cin == 16 && up == 2) || (Cin == 32 && up == 1))) {
        ThinArgs t{src0, w_fwd, shift, out, B, Hs / up, Ws / up, Hs, Ws, ld_fwd, act, 1, 0};
        if (Cin == 16) hipLaunchKernelGGL((k_conv3x3_thin<16, 2, true, 8>), dim3((Ws + 63) / 64, (Hs + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, t);
        else hipLaunchKernelGGL((k_conv3x3_thin<32, 1, true, 4>), dim3((Ws + 63) / 64, (Hs + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, t);
        E2E_LAUNCH_CHECK("e2e_conv2d_fwd");
        return E2E_OK;
    }
    launch_gemm<false>(a, vec, workspace, force, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_conv2d_fwd");
    return E2E_OK;
}

int e2e_conv2d_fwd(const float* src0, const float* src1, int C1, int up, const float* w_fwd, int ld_fwd, const float* scale,
                   const float* shift, const float* residual, float* out, int B, int Hs, int Ws, int Cin, int Cout, int KH,
                   int KW, int stride, int pad, int pad_mode, int act, float in_sub, float in_mul, float* workspace, void* stream) {
    return conv_fwd_impl(src0, src1, C1, up, w_fwd, ld_fwd, scale, shift, residual, out, B, Hs, Ws, Cin, Cout, KH, KW, stride, pad, pad_mode, act, in_sub, in_mul,
                         workspace, GemmCfg{0, 0, 0}, stream);
}

// a tuned call's decomposition: tile_m x tile_n workgroup tiles; ksplit >= 1: that many K slices, ksplit < 0: stream-K on -ksplit
// persistent workgroups (64 x 64 tiles); tile_m == 0: the library's own choice
static bool tuning_ok(int tile_m, int tile_n, int ksplit) {
    if (tile_m == 0) return true;
    if (ksplit < 0) return tile_m == 64 && tile_n == 64 && -ksplit <= SK_MAX_G;
    return tile_waves(tile_m, tile_n) != 0 && ksplit >= 1 && ksplit <= 16;
}

int e2e_conv2d_fwd_tuned(const float* src0, const float* src1, int C1, int up, const float* w_fwd, int ld_fwd, const float* scale,
                         const float* shift, const float* residual, float* out, int B, int Hs, int Ws, int Cin, int Cout, int KH,
                         int KW, int stride, int pad, int pad_mode, int act, float in_sub, float in_mul, float* workspace, int tile_m, int tile_n,
                         int ksplit, void* stream) {
    E2E_REQUIRE(tuning_ok(tile_m, tile_n, ksplit), E2E_ERR_ARG, "e2e_conv2d_fwd_tuned: unsupported decomposition %d x %d / %d", tile_m, tile_n, ksplit);
    return conv_fwd_impl(src0, src1, C1, up, w_fwd, ld_fwd, scale, shift, residual, out, B, Hs, Ws, Cin, Cout, KH, KW, stride, pad, pad_mode, act, in_sub, in_mul,
                         workspace, GemmCfg{tile_m, tile_n, ksplit}, stream);
}

static int bwd_data_impl(const float* dz, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo,
                         int KH, int KW, int stride, int pad, int pad_mode, int accumulate, const float* x_in, int in_act, const float* pre_add,
                         float* workspace, void* stream, GemmCfg force = GemmCfg{0, 0, 0}) {
    E2E_REQUIRE(pre_add == nullptr || pad_mode == 0, E2E_ERR_ARG, "e2e_conv2d_bwd_data: the pre-activation addend takes a zero-padded layer");
    E2E_REQUIRE(in_act == 0 || (x_in && (in_act == ACT_RELU || in_act == ACT_ELU) && pad_mode == 0), E2E_ERR_ARG,
                "e2e_conv2d_bwd_data: the fused input-activation derivative takes ReLU / ELU, the activation's output and a zero-padded layer");
    E2E_REQUIRE(dz && w_bwd && dxp && B > 0 && Cin > 0 && Cout > 0 && Cout % 16 == 0, E2E_ERR_ARG, "e2e_conv2d_bwd_data: bad argument (Cout %% 16 == 0)");
    E2E_REQUIRE(ld_bwd % 4 == 0 && ld_bwd >= Cin && (stride == 1 || stride == 2), E2E_ERR_ARG, "e2e_conv2d_bwd_data: bad sizes");
    E2E_REQUIRE((int64_t)B * Ho * Wo * Cout * 4 < (1ll << 30) && (int64_t)KH * KW * Cout * ld_bwd * 4 < (1ll << 31) && KH <= 3 && KW <= 3, E2E_ERR_ARG,
                "e2e_conv2d_bwd_data: gradients must stay below 1 GB, weights below 2 GB (32-bit buffer offsets), kernels up to 3 x 3");
    // 16 -> 16 channels on the padded grid of a reflection-padded layer (upconv(0,1)): dXp[q] = sum_t dZ[q - t] Wb[t], patch kernel
    if (!force.bm && KH == 3 && KW == 3 && stride == 1 && pad == 1 && pad_mode == 1 && Cin == 16 && Cout == 16 && !accumulate && !in_act && !pre_add && ld_bwd >= 16) {
        ThinArgs t{dz, w_bwd, nullptr, dxp, B, Ho, Wo, Hs + 2, Ws + 2, ld_bwd, ACT_NONE, 2, 1};
        hipLaunchKernelGGL((k_conv3x3_thin<16, 1, false, 8>), dim3((Ws + 2 + 63) / 64, (Hs + 2 + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, t);
        E2E_LAUNCH_CHECK("e2e_conv2d_bwd_data");
        return E2E_OK;
    }
    ConvArgs a{};
    a.src0 = dz; a.src1 = nullptr; a.w = w_bwd; a.out = dxp;
    a.B = B; a.Hs = Ho; a.Ws = Wo; a.Cin = Cout; a.C1 = Cout; a.up = 1;
    const int pp = pad_mode == 1 ? pad : 0;                // reflect: produce the whole padded domain, folded afterwards
    a.Hd = Hs + 2 * pp; a.Wd = Ws + 2 * pp; a.off = pad_mode == 1 ? 0 : pad;
    a.Ncols = Cin; a.ldw = ld_bwd; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.pad_mode = 0; a.act = ACT_NONE;
    // stride 2: parity classes (each input-gradient pixel only visits the taps that reach it).  A 1x1 kernel reaches one class
    // only -- the other three quarters of dxp are zeros, written by a memset node on the same stream.
    // (class lattices of 2^24 rows or more take the plain transposed form: the class epilogue decodes its rows in fp32)
    a.cls = (stride == 2 && (int64_t)B * ((a.Hd + 1) / 2) * ((a.Wd + 1) / 2) < (1 << 24)) ? 1 : 0;
    if (a.cls && (KH < 2 || KW < 2) && !accumulate)
        (void)hipMemsetAsync(dxp, 0, (size_t)B * a.Hd * a.Wd * Cin * sizeof(float), (hipStream_t)stream);
    // accumulate: dxp += result -- the epilogue's residual input reads the element it is about to overwrite (same thread)
    if (accumulate) a.res = dxp;
    a.xin = in_act ? x_in : nullptr; a.dact = in_act; a.pre = pre_add;
    launch_gemm<true>(a, 4, workspace, force, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_conv2d_bwd_data");
    return E2E_OK;
}

int e2e_conv2d_bwd_data(const float* dz, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout,
                        int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, float* workspace, void* stream) {
    return bwd_data_impl(dz, w_bwd, ld_bwd, dxp, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, 0, nullptr, 0, nullptr, workspace, stream);
}

int e2e_conv2d_bwd_data_acc(const float* dz, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout,
                            int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, float* workspace,
                            void* stream) {
    return bwd_data_impl(dz, w_bwd, ld_bwd, dxp, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate, nullptr, 0, nullptr, workspace, stream);
}

int e2e_conv2d_bwd_data_fused(const float* da, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout,
                              int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, const float* x_in,
                              int in_act, const float* pre_add, float* workspace, void* stream) {
    return bwd_data_impl(da, w_bwd, ld_bwd, dxp, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate, x_in, in_act, pre_add, workspace,
                         stream);
}

int e2e_conv2d_bwd_data_fused_tuned(const float* da, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout,
                                    int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, const float* x_in,
                                    int in_act, const float* pre_add, float* workspace, int tile_m, int tile_n, int ksplit, void* stream) {
    E2E_REQUIRE(tuning_ok(tile_m, tile_n, ksplit), E2E_ERR_ARG, "e2e_conv2d_bwd_data_fused_tuned: unsupported decomposition %d x %d / %d", tile_m, tile_n, ksplit);
    return bwd_data_impl(da, w_bwd, ld_bwd, dxp, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate, x_in, in_act, pre_add, workspace,
                         stream, GemmCfg{tile_m, tile_n, ksplit});
}

// launch the adjoint of pad + upsample + concat: 16-byte channel quads when the channel split allows, 32-bit indices
static void launch_gather_adjoint(const float* dxp, int B, int Hs, int Ws, int Cin, int C1, int up, int padded, float* d0, float* d1, int acc0, int acc1,
                                  const float* x0, int act0, const float* x1, int act1, hipStream_t st) {
    const int64_t n = (int64_t)B * (Hs / up) * (Ws / up) * C1 + (int64_t)B * Hs * Ws * (Cin - C1);
    if (Cin % 4 == 0 && C1 % 4 == 0 && n < (1ll << 31))
        hipLaunchKernelGGL(k_gather_adjoint4, dim3(egrid(n / 4)), dim3(256), 0, st, dxp, B, Hs, Ws, Cin, C1, up, padded ? 1 : 0, d0, d1, acc0, acc1, x0, act0, x1, act1);
    else
        hipLaunchKernelGGL(k_gather_adjoint, dim3(egrid(n)), dim3(256), 0, st, dxp, B, Hs, Ws, Cin, C1, up, padded ? 1 : 0, d0, d1, acc0, acc1, x0, act0, x1, act1);
}

int e2e_conv2d_gather_adjoint(const float* dxp, int B, int Hs, int Ws, int Cin, int C1, int up, int padded, float* d_src0,
                              float* d_src1, int accumulate0, int accumulate1, void* stream) {
    E2E_REQUIRE(dxp && d_src0 && B > 0 && Cin > 0 && C1 > 0 && C1 <= Cin && (C1 == Cin || d_src1) && (up == 1 || up == 2), E2E_ERR_ARG,
                "e2e_conv2d_gather_adjoint: bad argument");
    launch_gather_adjoint(dxp, B, Hs, Ws, Cin, C1, up, padded, d_src0, d_src1, accumulate0, accumulate1, nullptr, 0, nullptr, 0, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_conv2d_gather_adjoint");
    return E2E_OK;
}

int e2e_conv2d_gather_adjoint_act(const float* dxp, int B, int Hs, int Ws, int Cin, int C1, int up, int padded, float* d_src0, float* d_src1,
                                  int accumulate0, int accumulate1, const float* src0, int act0, const float* src1, int act1, void* stream) {
    E2E_REQUIRE(dxp && d_src0 && B > 0 && Cin > 0 && C1 > 0 && C1 <= Cin && (C1 == Cin || d_src1) && (up == 1 || up == 2), E2E_ERR_ARG,
                "e2e_conv2d_gather_adjoint_act: bad argument");
    E2E_REQUIRE((act0 == 0 || src0) && (act1 == 0 || src1) && act0 >= 0 && act0 <= 2 && act1 >= 0 && act1 <= 2, E2E_ERR_ARG,
                "e2e_conv2d_gather_adjoint_act: an activation derivative needs the activation's output");
    launch_gather_adjoint(dxp, B, Hs, Ws, Cin, C1, up, padded, d_src0, d_src1, accumulate0, accumulate1, src0, act0, src1, act1, (hipStream_t)stream);
    E2E_LAUNCH_CHECK("e2e_conv2d_gather_adjoint_act");
    return E2E_OK;
}

int e2e_conv2d_act_bwd(const float* dy, const float* y, const float* scale, float* dz, int64_t n, int C, int act, void* stream) {
    E2E_REQUIRE(dy && y && dz && n > 0 && C > 0 && act >= 0 && act <= 3, E2E_ERR_ARG, "e2e_conv2d_act_bwd: bad argument");
    hipLaunchKernelGGL(k_act_bwd, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, scale, dz, n, C, act, 0);
    E2E_LAUNCH_CHECK("e2e_conv2d_act_bwd");
    return E2E_OK;
}

int e2e_conv2d_act_bwd_acc(const float* dy, const float* y, const float* scale, float* dz, int64_t n, int C, int act, int accumulate,
                           void* stream) {
    E2E_REQUIRE(dy && y && dz && n > 0 && C > 0 && act >= 0 && act <= 3, E2E_ERR_ARG, "e2e_conv2d_act_bwd_acc: bad argument");
    hipLaunchKernelGGL(k_act_bwd, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, scale, dz, n, C, act, accumulate);
    E2E_LAUNCH_CHECK("e2e_conv2d_act_bwd_acc");
    return E2E_OK;
}

#define HEAD_PARTS 2048
int e2e_head_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int act, void* stream) {
    E2E_REQUIRE(x && w && y && B > 0 && H >= 2 && W >= 2 && Cin == HC, E2E_ERR_ARG, "e2e_head_fwd: bad argument (the head takes %d channels)", HC);
    hipLaunchKernelGGL(k_head_fwd, dim3(egrid((int64_t)B * H * W * 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, H, W, act);
    E2E_LAUNCH_CHECK("e2e_head_fwd");
    return E2E_OK;
}
int64_t e2e_head_workspace_floats(void) { return (int64_t)HEAD_PARTS * (9 * HC + 1); }
static int head_bwd_impl(const float* dz, const float* x, const float* w, float* dx, float* dw, float* dbias, float* workspace, int B, int H,
                         int W, int Cin, int in_act, void* stream) {
    E2E_REQUIRE(dz && x && w && workspace && B > 0 && H >= 2 && W >= 2 && Cin == HC && in_act >= 0 && in_act <= 2, E2E_ERR_ARG, "e2e_head_bwd: bad argument");
    E2E_REQUIRE((int64_t)B * H * W * HC < (1ll << 31), E2E_ERR_ARG, "e2e_head_bwd: activation too large for 32-bit element offsets");
    hipStream_t st = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(k_head_bwd_data, dim3(egrid((int64_t)B * H * W * 4)), dim3(256), 0, st, dz, w, dx, B, H, W, x, in_act);
    if (dw) {
        hipLaunchKernelGGL(k_head_bwd_weight, dim3(HEAD_PARTS), dim3(256), 0, st, dz, x, workspace, B, H, W);
        hipLaunchKernelGGL(k_head_wreduce, dim3(9 * HC + 1), dim3(256), 0, st, workspace, HEAD_PARTS, dw, dbias);
    }
    E2E_LAUNCH_CHECK("e2e_head_bwd");
    return E2E_OK;
}

int e2e_head_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* dbias, float* workspace, int B, int H,
                 int W, int Cin, void* stream) {
    return head_bwd_impl(dz, x, w, dx, dw, dbias, workspace, B, H, W, Cin, 0, stream);
}

int e2e_head_bwd_act(const float* dz, const float* x, const float* w, float* dx, float* dw, float* dbias, float* workspace, int B, int H,
                     int W, int Cin, int in_act, void* stream) {
    return head_bwd_impl(dz, x, w, dx, dw, dbias, workspace, B, H, W, Cin, in_act, stream);
}

// backward-weight decomposition shared by the workspace query and the launch: tile shape, padded GEMM size, pixel slices
struct WgradPlan { int tm, tn, Mpad, Npad; int64_t S; };
static WgradPlan wgrad_plan(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int has_bias, bool use16, int wg_target = WGRAD_TARGET) {
    WgradPlan p;
    const int Ng = KH * KW * Cin + (has_bias ? 1 : 0);
    p.tm = Cout <= 32 ? 32 : 64; p.tn = Cout <= 32 ? 128 : 64;        // 32x128 tiles for the thin layers
    if (use16) { p.tm = 16; p.tn = 160; }                              // k_wgrad_gemm16 (16x16x4 MFMA tiles)
    p.Mpad = (Cout + p.tm - 1) / p.tm * p.tm; p.Npad = (Ng + p.tn - 1) / p.tn * p.tn;
    const int64_t P = (int64_t)B * Ho * Wo;
    const int64_t tiles = (int64_t)(p.Mpad / p.tm) * (p.Npad / p.tn);
    const int target = use16 ? wg_target * 3 / 4 : wg_target;     // measured: 768 slices suit the 16-channel kernel (94 vs 110 us)
    int64_t S = (target + tiles - 1) / tiles;                // pixel slices: ~target workgroups in total
    const int64_t maxS = (P + 255) / 256;                   // at least 256 pixels per slice
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    p.S = S;
    return p;
}

// decomposition of k_wgrad3x3_thin: workgroups of `tiles_x` consecutive 4 x 64 tiles, ~1200 of them; one slab per workgroup position
struct ThinWgradPlan { int tiles_x, nxg, ny, npad; int64_t S; };
// measured against the implicit-GEMM kernels (tools/gemm_tune.py wgrad): upconv(0,1) 16 -> 16: 94 -> 66 us, upconv(0,0) 32 -> 16: 55 -> 45 us,
// upconv(1,1) 96 -> 32: 174 -> 142 us; upconv(1,0) 64 -> 32 is faster as a GEMM (39 vs 46 us) and stays there
static bool thin_wgrad_ok(int Cin, int Cout, int KH, int KW) {
    return KH == 3 && KW == 3 && ((Cout == 16 && (Cin == 16 || Cin == 32)) || (Cout == 32 && Cin == 96));
}
static ThinWgradPlan thin_wgrad_plan(int B, int Ho, int Wo, int Cin, int has_bias) {
    ThinWgradPlan p;
    const int txt = (Wo + 63) / 64;
    p.ny = (Ho + 3) / 4;
    const int64_t total = (int64_t)B * p.ny * txt * (Cin / 16);
    p.tiles_x = 1;
    while (total / p.tiles_x > 1400 && p.tiles_x < txt) ++p.tiles_x;
    p.nxg = (txt + p.tiles_x - 1) / p.tiles_x;
    p.S = (int64_t)B * p.ny * p.nxg;
    p.npad = (9 * Cin + (has_bias ? 1 : 0) + 3) / 4 * 4;
    return p;
}

// decomposition of k_wgrad3x3_taps: (Cin / 32) x (Cout / 32) tiles x S pixel slices of whole 8 x 8 patches, ~TAP_TARGET workgroups of 3 waves
#define TAP_TARGET 640
struct TapWgradPlan { int ptx, pty, nchunks, cps, S, Npad; };
static bool tap_wgrad_shape_ok(int Cin, int Cout, int KH, int KW) { return KH == 3 && KW == 3 && Cin % 32 == 0 && Cout % 32 == 0; }
static TapWgradPlan tap_wgrad_plan(int B, int H, int W, int Cin, int Cout, int has_bias) {
    TapWgradPlan p;
    p.ptx = (W + 7) / 8; p.pty = (H + 7) / 8;
    p.nchunks = B * p.ptx * p.pty;
    const int tiles = (Cin / 32) * (Cout / 32);
    static const int target = getenv("E2E_TAP_TARGET") ? atoi(getenv("E2E_TAP_TARGET")) : TAP_TARGET;      // (diagnostic sweep; read once)
    int S = (target + tiles - 1) / tiles;
    if (S > p.nchunks) S = p.nchunks;
    if (S < 1) S = 1;
    p.cps = (p.nchunks + S - 1) / S;
    p.S = (p.nchunks + p.cps - 1) / p.cps;
    p.Npad = (9 * Cin + (has_bias ? 1 : 0) + 3) / 4 * 4;
    return p;
}

int64_t e2e_conv2d_wgrad_workspace_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int has_bias) {
    const WgradPlan p = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW, has_bias, false);
    int64_t n = p.S * (int64_t)p.Mpad * p.Npad;
    if (tap_wgrad_shape_ok(Cin, Cout, KH, KW)) {
        const TapWgradPlan t = tap_wgrad_plan(B, Ho, Wo, Cin, Cout, has_bias);
        if ((int64_t)t.S * Cout * t.Npad > n) n = (int64_t)t.S * Cout * t.Npad;
    }
    if (Cout == 16) {                                                  // the 16-channel kernel is chosen at launch (it needs the lean loader)
        const WgradPlan q = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW, has_bias, true);
        if (q.S * (int64_t)q.Mpad * q.Npad > n) n = q.S * (int64_t)q.Mpad * q.Npad;
    }
    if (thin_wgrad_ok(Cin, Cout, KH, KW)) {
        const ThinWgradPlan t = thin_wgrad_plan(B, Ho, Wo, Cin, has_bias);
        if (t.S * Cout * t.npad > n) n = t.S * Cout * t.npad;
    }
    if (KH == 7 && KW == 7 && Cin == 3 && Cout == 64) {                 // k_wgrad7x7_stem: one slab [64][148] per pair of 4 x 32 tiles
        const int64_t S = (int64_t)B * ((Ho + 3) / 4) * (((Wo + 31) / 32 + 1) / 2);
        if (S * 64 * 148 > n) n = S * 64 * 148;
    }
    return n;
}

static int bwd_weight_impl(const float* dz, const float* src0, const float* src1, int C1, int up, float* dw, float* dbias, float* workspace, int B,
                           int Hs, int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, int accumulate,
                           float in_sub, float in_mul, const float* out_scale, void* stream, int wg_target = 0, e2e_wgrad_reduce_desc* defer = nullptr) {
    const bool tuned = wg_target != 0;
    if (!tuned) wg_target = WGRAD_TARGET;
    E2E_REQUIRE(dz && src0 && dw && workspace && B > 0 && Cin > 0 && Cout > 0, E2E_ERR_ARG, "e2e_conv2d_bwd_weight: bad argument");
    const int vec = (Cin % 4 == 0 && C1 % 4 == 0) ? 4 : 1;
    E2E_REQUIRE(vec == 4 || (C1 == Cin && up == 1 && pad_mode == 0), E2E_ERR_ARG, "e2e_conv2d_bwd_weight: scalar path takes one full-resolution zero-padded source");
    WgradArgs a{};
    a.dz = dz; a.src0 = src0; a.src1 = src1; a.slabs = workspace;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.C1 = C1; a.up = up; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.pad_mode = pad_mode; a.has_bias = dbias ? 1 : 0;
    a.Ngemm = KH * KW * Cin + a.has_bias;
    const int64_t P = (int64_t)B * Ho * Wo;
    // lean VEC-4 kernel: 32 pixels per chunk; needs Cout % 4 == 0, 32-bit offsets and image rows of at least 8 pixels
    const bool lean = vec == 4 && Cout % 4 == 0 && Wo >= 8 && (int64_t)B * Hs * Ws * Cin * 4 < (1ll << 31) && P * Cout * 4 < (1ll << 31) && P < (1ll << 24);
    hipStream_t st = (hipStream_t)stream;
    // the slab reduction that ends every path: launched here, or -- `defer` -- described for e2e_wgrad_reduce_batched
    auto reduce = [&](int S, int Mpad, int Npad, int zl) {
        if (defer) {
            *defer = e2e_wgrad_reduce_desc{workspace, dw, dbias, out_scale, S, Mpad, Npad, Cout, Cin, KH, KW, a.has_bias, accumulate, zl, 0};
            return;
        }
        const int64_t tq = (int64_t)Cout * ((a.Ngemm + 3) / 4);
        if (zl == 8)
            hipLaunchKernelGGL((k_wgrad_reduce<8>), dim3(egrid(tq * 4)), dim3(512), 0, st, workspace, S, Mpad, Npad, Cout, Cin, KH, KW, a.has_bias, dw, dbias,
                               accumulate, out_scale);
        else
            hipLaunchKernelGGL((k_wgrad_reduce<2>), dim3(egrid(tq * 4)), dim3(128), 0, st, workspace, S, Mpad, Npad, Cout, Cin, KH, KW, a.has_bias, dw, dbias,
                               accumulate, out_scale);
    };
    // the RGB stem: patch kernel + the common slab reduction
    if (KH == 7 && KW == 7 && stride == 2 && pad == 3 && pad_mode == 0 && Cin == 3 && Cout == 64 && C1 == Cin && up == 1 && !dbias &&
        Ho == (Hs + 6 - 7) / 2 + 1 && Wo == (Ws + 6 - 7) / 2 + 1) {
        const int nxg = ((Wo + 31) / 32 + 1) / 2, ny = (Ho + 3) / 4;
        StemWgradArgs ta{dz, src0, workspace, B, Hs, Ws, Ho, Wo, 2, nxg, in_sub, in_mul};
        hipLaunchKernelGGL(k_wgrad7x7_stem, dim3(nxg, ny, B), dim3(256), 0, st, ta);
        reduce(B * ny * nxg, 64, 148, 8);
        E2E_LAUNCH_CHECK("e2e_conv2d_bwd_weight");
        return E2E_OK;
    }
    // the thin 3x3 layers of the decoder's last two levels (reflection pad, 16 / 32 output channels): patch kernel + the common slab reduction
    if (!tuned && lean && thin_wgrad_ok(Cin, Cout, KH, KW) && stride == 1 && pad == 1 && pad_mode == 1 && C1 % 16 == 0 && Ho == Hs && Wo == Ws &&
        (C1 == Cin || src1) && Hs % up == 0 && Ws % up == 0) {
        const ThinWgradPlan t = thin_wgrad_plan(B, Ho, Wo, Cin, a.has_bias);
        ThinWgradArgs ta{dz, src0, src1, workspace, B, Ho, Wo, Cin, C1, up, t.npad, a.has_bias, t.tiles_x, t.nxg};
        const dim3 tg((unsigned)t.nxg, (unsigned)t.ny, (unsigned)(B * (Cin / 16)));
        if (Cout == 32) hipLaunchKernelGGL((k_wgrad3x3_thin<2>), tg, dim3(256), 0, st, ta);
        else hipLaunchKernelGGL((k_wgrad3x3_thin<1>), tg, dim3(256), 0, st, ta);
        reduce((int)t.S, Cout, t.npad, 8);
        E2E_LAUNCH_CHECK("e2e_conv2d_bwd_weight");
        return E2E_OK;
    }
    // the 32-channel-tile 3x3 layers (encoder stages, decoder upconv(k, 0) and the concat layers whose two sources are multiples of 32 wide):
    // tap-reuse patch kernel + the common slab reduction.  OPT-IN (E2E_WGRAD_TAPS=1): measured at parity with or behind the implicit-GEMM
    // kernels on every layer (profiles/r04_wgrad_taps.txt: 52.7 vs 48.4 us on layer1 at its best decomposition) -- it is kept, tested, as
    // the record of that experiment and as a starting point, not as the product's path
    if (!tuned && vec == 4 && tap_wgrad_shape_ok(Cin, Cout, KH, KW) && stride == 1 && pad == 1 && Ho == Hs && Wo == Ws && (up == 1 || up == 2) &&
        Hs % up == 0 && Ws % up == 0 && (C1 == Cin || (src1 && C1 % 32 == 0)) && getenv("E2E_WGRAD_TAPS") != nullptr && getenv("E2E_WGRAD_TAPS")[0] == '1') {
        const TapWgradPlan t = tap_wgrad_plan(B, Ho, Wo, Cin, Cout, a.has_bias);
        TapWgradArgs ta{dz, src0, src1, workspace, B, Ho, Wo, Cin, C1, up, Cout, pad_mode == 1 ? 1 : 0, a.has_bias, Cout, t.Npad, t.ptx, t.pty, t.nchunks, t.cps};
        const dim3 tg((unsigned)(Cin / 32), (unsigned)(Cout / 32), (unsigned)t.S);
        if (a.has_bias) hipLaunchKernelGGL(k_wgrad3x3_taps<true>, tg, dim3(TAP_NT), 0, st, ta);
        else hipLaunchKernelGGL(k_wgrad3x3_taps<false>, tg, dim3(TAP_NT), 0, st, ta);
        reduce(t.S, Cout, t.Npad, t.S >= 8 ? 8 : 2);
        E2E_LAUNCH_CHECK("e2e_conv2d_bwd_weight");
        return E2E_OK;
    }
    const bool use16 = lean && Cout == 16;
    const WgradPlan wp = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW, a.has_bias, use16, wg_target);
    const int tm = wp.tm, tn = wp.tn;
    a.Mpad = wp.Mpad; a.Npad = wp.Npad;
    a.in_sub = in_sub; a.in_mul = in_mul; a.vec = vec;
    const int64_t S = wp.S;
    const int cbp = lean ? 32 : CBK;
    a.pix_per_slice = ((P + S - 1) / S + cbp - 1) / cbp * cbp;
    const int Sz = (int)((P + a.pix_per_slice - 1) / a.pix_per_slice);
    dim3 g((unsigned)(a.Npad / tn), (unsigned)(a.Mpad / tm), (unsigned)Sz);
    if (use16) {
        hipLaunchKernelGGL((k_wgrad_gemm16<10, 32>), g, dim3(256), 0, st, a);
    } else if (lean) {
        if (tm == 32) {
            if (src1) hipLaunchKernelGGL((k_wgrad_gemm4<1, 4, 32, true>), g, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_wgrad_gemm4<1, 4, 32, false>), g, dim3(256), 0, st, a);
        } else {
            if (src1) hipLaunchKernelGGL((k_wgrad_gemm4<2, 2, 32, true>), g, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_wgrad_gemm4<2, 2, 32, false>), g, dim3(256), 0, st, a);
        }
    } else if (tm == 32) {
        if (vec == 4) hipLaunchKernelGGL((k_wgrad_gemm<1, 4, 4>), g, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_wgrad_gemm<1, 4, 1>), g, dim3(256), 0, st, a);
    } else {
        if (vec == 4) hipLaunchKernelGGL((k_wgrad_gemm<2, 2, 4>), g, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_wgrad_gemm<2, 2, 1>), g, dim3(256), 0, st, a);
    }
    reduce(Sz, a.Mpad, a.Npad, Sz >= 8 ? 8 : 2);
    E2E_LAUNCH_CHECK("e2e_conv2d_bwd_weight");
    return E2E_OK;
}

int e2e_conv2d_bwd_weight(const float* dz, const float* src0, const float* src1, int C1, int up, float* dw, float* dbias,
                          float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride,
                          int pad, int pad_mode, int accumulate, float in_sub, float in_mul, void* stream) {
    return bwd_weight_impl(dz, src0, src1, C1, up, dw, dbias, workspace, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate,
                           in_sub, in_mul, nullptr, stream);
}

int e2e_conv2d_bwd_weight_scaled(const float* da, const float* out_scale, const float* src0, const float* src1, int C1, int up, float* dw,
                                 float* dbias, float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW,
                                 int stride, int pad, int pad_mode, int accumulate, float in_sub, float in_mul, void* stream) {
    return bwd_weight_impl(da, src0, src1, C1, up, dw, dbias, workspace, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate,
                           in_sub, in_mul, out_scale, stream);
}

/* e2e_conv2d_bwd_weight_scaled WITHOUT its final launch: the partial slabs stay in `workspace` (which must stay untouched until the reduction
 * ran) and *desc_out (HOST memory) describes the reduction left to do.  A caller collects the descriptors of a whole backward pass,
 * e2e_wgrad_reduce_batch_prepare()s the array, copies it to the device once (shapes, pointers and decompositions are those of a static launch
 * plan) and ends the pass with ONE e2e_wgrad_reduce_batched launch.  Results are bit-identical to e2e_conv2d_bwd_weight_scaled. */
int e2e_conv2d_bwd_weight_scaled_deferred(const float* da, const float* out_scale, const float* src0, const float* src1, int C1, int up, float* dw,
                                          float* dbias, float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW,
                                          int stride, int pad, int pad_mode, int accumulate, float in_sub, float in_mul,
                                          e2e_wgrad_reduce_desc* desc_out, void* stream) {
    E2E_REQUIRE(desc_out, E2E_ERR_ARG, "e2e_conv2d_bwd_weight_scaled_deferred: desc_out is NULL");
    return bwd_weight_impl(da, src0, src1, C1, up, dw, dbias, workspace, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate,
                           in_sub, in_mul, out_scale, stream, 0, desc_out);
}

/* fills first_item of n descriptors in HOST memory (running total of work items); returns the total, or -1 on a malformed descriptor */
long long e2e_wgrad_reduce_batch_prepare(e2e_wgrad_reduce_desc* descs_host, int n) {
    if (!descs_host || n <= 0) return -1;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        e2e_wgrad_reduce_desc& d = descs_host[i];
        if (!d.slabs || !d.dw || d.S <= 0 || d.Mpad <= 0 || d.Npad <= 0 || d.Npad % 4 || d.Cout <= 0 || d.Cin <= 0 || d.KH <= 0 || d.KW <= 0 || (d.zl != 8 && d.zl != 2))
            return -1;
        const long long quads = (long long)d.Cout * ((d.KH * d.KW * d.Cin + (d.has_bias ? 1 : 0) + 3) / 4);
        d.first_item = total;
        total += (quads + (d.zl == 8 ? 63 : 255)) / (d.zl == 8 ? 64 : 256);
    }
    return total;
}

int e2e_wgrad_reduce_batched(const e2e_wgrad_reduce_desc* descs_dev, int n, long long total_items, void* stream) {
    E2E_REQUIRE(descs_dev && n > 0 && total_items > 0, E2E_ERR_ARG, "e2e_wgrad_reduce_batched: bad argument");
    hipLaunchKernelGGL(k_wgrad_reduce_batched, dim3((unsigned)(total_items < 4096 ? total_items : 4096)), dim3(512), 0, (hipStream_t)stream, descs_dev, n, total_items);
    E2E_LAUNCH_CHECK("e2e_wgrad_reduce_batched");
    return E2E_OK;
}

/* the same through the implicit-GEMM kernels with an explicit number of workgroups to spread the pixel slices over (64 .. 8192; tools/gemm_tune.py).
 * workspace: e2e_conv2d_wgrad_tuned_workspace_floats(...) floats */
int e2e_conv2d_bwd_weight_scaled_tuned(const float* da, const float* out_scale, const float* src0, const float* src1, int C1, int up, float* dw,
                                       float* dbias, float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW,
                                       int stride, int pad, int pad_mode, int accumulate, float in_sub, float in_mul, int target_workgroups, void* stream) {
    E2E_REQUIRE(target_workgroups >= 64 && target_workgroups <= 8192, E2E_ERR_ARG, "e2e_conv2d_bwd_weight_scaled_tuned: 64 .. 8192 workgroups");
    return bwd_weight_impl(da, src0, src1, C1, up, dw, dbias, workspace, B, Hs, Ws, Cin, Cout, Ho, Wo, KH, KW, stride, pad, pad_mode, accumulate,
                           in_sub, in_mul, out_scale, stream, target_workgroups);
}

int64_t e2e_conv2d_wgrad_tuned_workspace_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int has_bias, int target_workgroups) {
    int64_t n = 0;
    for (int u = 0; u < 2; ++u) {
        if (u == 1 && Cout != 16) break;
        const WgradPlan p = wgrad_plan(B, Ho, Wo, Cin, Cout, KH, KW, has_bias, u == 1, target_workgroups);
        if (p.S * (int64_t)p.Mpad * p.Npad > n) n = p.S * (int64_t)p.Mpad * p.Npad;
    }
    return n;
}

#ifdef E2E_CONV_STAMPS
int e2e_debug_read_stamps(unsigned long long* host, int n) {
    if (n < -(1 << 20)) return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phases), (size_t)(-n - (1 << 20)) * 8);   // the K-loop phase clocks
    if (n < 0) return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps2), (size_t)(-n) * 8);      // n < 0: the prologue stamps
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8);
}
#endif

}  // extern "C"
