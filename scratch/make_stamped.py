src=open('end-to-end-self-supervised-slam_amd/csrc/warp_photo_fused.hip').read()
src=src.replace('#include "e2e_common.h"','#include "../end-to-end-self-supervised-slam_amd/csrc/e2e_common.h"')
def stamp(i):
    return '{ unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (tid == 0) dbg[(size_t)blk_ * 8 + ' + str(i) + '] = t_; }'
def rep(a, b):
    global src
    assert a in src, a[:60]
    src = src.replace(a, b)
rep('    float* __restrict__ partials, int B, int H, int W) {\n    constexpr int LT_H','    float* __restrict__ partials, int B, int H, int W, unsigned long long* dbg) {\n    constexpr int LT_H')
rep('    const int tid = threadIdx.y * LT_W + threadIdx.x;\n    const int64_t N = (int64_t)H * W;\n    const float* dep','    const int tid = threadIdx.y * LT_W + threadIdx.x;\n    const int blk_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;\n    '+stamp(0)+'\n    const int64_t N = (int64_t)H * W;\n    const float* dep')
rep('    __syncthreads();                                  // sgeo ready','    __syncthreads();                                  // sgeo ready\n    '+stamp(1))
rep('#pragma unroll\n    for (int e = 0; e < NE; ++e) {\n        const Taps& k = kp[e];\n        const float w00','    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    '+stamp(2)+'\n#pragma unroll\n    for (int e = 0; e < NE; ++e) {\n        const Taps& k = kp[e];\n        const float w00')
rep('    // ---- phase 2: SSIM statistics','    '+stamp(3)+'\n    // ---- phase 2: SSIM statistics')
rep('    // ---- phase 3: adjoint per own pixel','    '+stamp(4)+'\n    // ---- phase 3: adjoint per own pixel')
rep('    // ---- loss: per-workgroup partial sums','    '+stamp(5)+'\n    // ---- loss: per-workgroup partial sums')
rep('g_depth_tgt, g_depth_src, workspace, B, H, W\n','g_depth_tgt, g_depth_src, workspace, B, H, W, (unsigned long long*)dbg_buf\n')
rep('float* workspace, int B, int H, int W, void* stream) {\n    E2E_REQUIRE(B > 0 && H > 1 && W > 1 && (int64_t)B * H * W * 3','float* workspace, int B, int H, int W, void* stream, void* dbg_buf) {\n    E2E_REQUIRE(B > 0 && H > 1 && W > 1 && (int64_t)B * H * W * 3')
rep('int e2e_warp_photo_lossgrad(const float* depth_tgt','int dbg_lossgrad(const float* depth_tgt')
rep('int64_t e2e_warp_photo_lossgrad_workspace_floats(int B, int H, int W) {','int64_t dbg_ws(int B, int H, int W) {')
open('scratch/lossgrad_stamped.hip','w').write(src)
