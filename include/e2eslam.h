/* e2eslam.h -- C ABI of libe2eslam_hip.so: the MI355X (gfx950) implementation of the
 * online-refinement hot path of ivanalberico/End-To-End-Self-Supervised-SLAM.
 *
 * The reference has no FFI layer: its boundary for this path is the Python import surface of
 * online_adaption.py:15-36 / train_depth.py:17-38.  These entry points are what a
 * ctypes binding of that surface calls; each one cites the reference interface it replaces.
 * The Python host side lives in end-to-end-self-supervised-slam_amd/ (same module / class /
 * function names as the reference); INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (or int64 / int32 where stated) unless marked `host`;
 *  - the caller owns every buffer (torch allocator); the library allocates nothing;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); no call synchronises;
 *  - return value 0 = OK, negative = E2E_ERR_*; e2e_last_error() gives a thread-local message;
 *  - image tensors are addressed through ELEMENT strides (sb, sc, sh, sw) so that the
 *    reference's NHWC frame stack (online_adaption.py:215-220) and its permuted NCHW views
 *    (online_adaption.py:393-394) are read in place, without a layout copy;
 *  - results are deterministic run to run: no floating-point atomics on any parity output.
 */
#ifndef E2ESLAM_H
#define E2ESLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2E_OK 0
#define E2E_ERR_ARG (-1)      /* bad argument (null pointer, non-positive size, unknown enum) */
#define E2E_ERR_LAUNCH (-2)   /* HIP reported a launch error */
#define E2E_ERR_WORKSPACE (-3)/* caller-provided workspace too small */

#define E2E_PADDING_ZEROS 0   /* MODEL.padding_mode: zeros  (configs/config.yaml:35) */
#define E2E_PADDING_BORDER 1  /* MODEL.padding_mode: border */

/* element strides of a (B,C,H,W)-indexed fp32 image */
typedef struct e2e_strides {
    int64_t sb, sc, sh, sw;
} e2e_strides;

int e2e_version(void);
const char* e2e_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* View synthesis -- depth_estimation/view_synthesis.py                                         */
/* ------------------------------------------------------------------------------------------ */

/* BackprojectDepth.forward (view_synthesis.py:34-40): depth (B,H*W), inv_K (B,4,4)
 * -> cam_points (B,4,H*W) rows x,y,z,1. */
int e2e_backproject_fwd(const float* depth, const float* inv_K, float* cam_points,
                        int B, int H, int W, void* stream);
/* its autograd: g_cam (B,4,H*W) -> g_depth (B,H*W). */
int e2e_backproject_bwd(const float* g_cam, const float* inv_K, float* g_depth,
                        int B, int H, int W, void* stream);

/* Project3D.forward (view_synthesis.py:54-78): points (B,4,H*W), K, T (B,4,4)
 * -> grid (B,H,W,2) normalised with /(W-1),/(H-1); valid (B,H,W) = max|grid|<=1 as 0/1 float;
 * z_out (B,H,W) = clamp(c2, 1e-3) when non-NULL (geometric=True, view_synthesis.py:73-76). */
int e2e_project3d_fwd(const float* points, const float* K, const float* T, float* grid, float* valid,
                      float* z_out, int B, int H, int W, void* stream);
/* autograd of the above: g_grid (B,H,W,2), g_z (B,H,W) or NULL -> g_points (B,4,H*W). */
int e2e_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                      const float* g_z, float* g_points, int B, int H, int W, void* stream);

/* F.grid_sample(input, grid, mode="bilinear", padding_mode, align_corners) as called at
 * online_adaption.py:431-439,450-453.  input (B,C,Hi,Wi) via strides, grid (B,Ho,Wo,2) contiguous,
 * out (B,C,Ho,Wo) contiguous. */
int e2e_grid_sample_fwd(const float* input, e2e_strides in_strides, const float* grid, float* out,
                        int B, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode,
                        int align_corners, void* stream);
/* g_out (B,C,Ho,Wo) contiguous -> g_grid (B,Ho,Wo,2); if g_input != NULL it must be a ZEROED
 * contiguous (B,C,Hi,Wi) buffer that receives the scatter-add (only the geometric branch,
 * online_adaption.py:436-439, needs it; this one output uses float atomics). */
int e2e_grid_sample_bwd(const float* input, e2e_strides in_strides, const float* grid,
                        const float* g_out, float* g_grid, float* g_input, int B, int C, int Hi,
                        int Wi, int Ho, int Wo, int padding_mode, int align_corners, void* stream);
/* The same with a BITWISE REPRODUCIBLE gradient with respect to the sampled image: the bilinear adjoint is a data-dependent scatter; its
 * contributions are accumulated as 2^-48 fixed-point integers (integer adds commute: no dependence on arrival order; resolution 3.6e-15)
 * in g_input_fixed (B*C*Hi*Wi + 1 int64 of scratch, zeroed here) and converted to g_input (B,C,Hi,Wi) afterwards.  Range: every
 * contribution |g_out * weight| < 4096 and every sum < 32768; a contribution that is not finite or not below 4096 is refused and raises
 * the scratch's last word, and g_input then comes back as NaN everywhere (never a wrapped or saturated finite value). */
int e2e_grid_sample_bwd_exact(const float* input, e2e_strides in_strides, const float* grid, const float* g_out,
                              float* g_grid, long long* g_input_fixed, float* g_input, int B, int C, int Hi, int Wi,
                              int Ho, int Wo, int padding_mode, int align_corners, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Photometric loss -- loss/losses.py                                                           */
/* ------------------------------------------------------------------------------------------ */

/* SSIM.forward (losses.py:23-37) and photometric_loss (losses.py:97-117) in one pass.
 * x = prediction, y = target, both (B,C,H,W) via strides.
 * ssim_out (B,C,H,W) contiguous or NULL; pmap_out (B,1,H,W) contiguous or NULL
 * (pmap = 0.85*mean_c ssim + 0.15*mean_c |y-x|). */
int e2e_photometric_fwd(const float* x, e2e_strides xs, const float* y, e2e_strides ys,
                        float* ssim_out, float* pmap_out, int B, int C, int H, int W, void* stream);
/* Gradient wrt x.  g_pmap (B,1,H,W) or NULL, g_ssim (B,C,H,W) or NULL (both contiguous; the two
 * contributions add) -> g_x (B,C,H,W) contiguous.  Gradient wrt y: call with x and y swapped
 * (SSIM is symmetric; the L1 term's sign flips with the swap, as it must). */
int e2e_photometric_bwd(const float* x, e2e_strides xs, const float* y, e2e_strides ys,
                        const float* g_pmap, const float* g_ssim, float* g_x, int B, int C, int H,
                        int W, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Fused refinement-step kernels (what the build's own driver launches)                         */
/* ------------------------------------------------------------------------------------------ */

/* Number of floats of workspace the fused forward needs for (B,H,W). */
int64_t e2e_warp_photo_workspace_floats(int B, int H, int W);

/* Forward of one refinement step's image-space part, ONE launch (+ a 1-block reduction):
 *   backproject -> project -> grid_sample(src) -> mask -> masked SSIM+L1 -> mean   [+ depth reg.]
 * reference: online_adaption.py:412-455 (novel_view_synthesis), :544-564, :482-511, :612-623.
 *   depth_tgt (B,H,W)   target depth (after median scaling)
 *   src, tgt            source / target frame, (B,3,H,W) via strides (NHWC memory is fine)
 *   K, inv_K, T         (B,4,4)
 *   synth (B,3,H,W), valid (B,H,W)   outputs kept for the backward
 *   pmap (B,H,W) or NULL             per-pixel loss map
 *   use_mask            LOSS.photometric_mask
 *   reg_kind            0 = no depth regulariser, 1 = l1, 2 = l2 (losses.py:134-148); when != 0:
 *     reg_init_tgt, reg_init_src (B,H,W)  initial depths (online_adaption.py:284-285)
 *     depth_src (B,H,W)                    refined depth of the source frame
 *   loss_out[0] = photometric mean, loss_out[1] = mean-reg(tgt) + mean-reg(src)
 *   workspace: e2e_warp_photo_workspace_floats floats. */
int e2e_warp_photo_fwd(const float* depth_tgt, const float* src, e2e_strides src_strides,
                       const float* tgt, e2e_strides tgt_strides, const float* K, const float* inv_K,
                       const float* T, float* synth, float* valid, float* pmap, int use_mask,
                       int padding_mode, int reg_kind, const float* reg_init_tgt,
                       const float* reg_init_src, const float* depth_src, float* loss_out,
                       float* workspace, int B, int H, int W, void* stream);

/* Backward of the above, ONE launch.
 *   g_loss (device, 2 floats): upstream gradients of loss_out[0] and loss_out[1]
 *   g_depth_tgt (B,H,W) = d/d(depth_tgt) of g_loss[0]*photometric + g_loss[1]*reg
 *   g_depth_src (B,H,W) = d/d(depth_src) of g_loss[1]*reg   (written only when reg_kind != 0;
 *   the warp reads the TARGET depth only: online_adaption.py:419). */
int e2e_warp_photo_bwd(const float* depth_tgt, const float* src, e2e_strides src_strides,
                       const float* tgt, e2e_strides tgt_strides, const float* K, const float* inv_K,
                       const float* T, const float* synth, const float* valid, int use_mask,
                       int padding_mode, int reg_kind, const float* reg_init_tgt,
                       const float* reg_init_src, const float* depth_src, const float* g_loss,
                       float* g_depth_tgt, float* g_depth_src, int B, int H, int W, void* stream);

/* The step the build's driver actually launches: loss AND gradient in ONE pass (the gradient of a
 * mean does not depend on its value, so forward and backward of the image-space part collapse;
 * synth / valid / the loss map never reach HBM).  Same semantics as e2e_warp_photo_fwd followed by
 * e2e_warp_photo_bwd with upstream gradients (w_photo, w_reg):
 *   loss_out[0] = photometric mean, loss_out[1] = regulariser (unweighted);
 *   g_depth_tgt = d(w_photo*loss0 + w_reg*loss1)/d(depth_tgt), g_depth_src likewise (reg_kind != 0).
 * workspace: e2e_warp_photo_lossgrad_workspace_floats floats (per-workgroup partial sums; a
 * second-stage kernel adds them in a fixed order => bitwise reproducible loss).  loss_out == NULL
 * skips that second launch (gradients only). */
int64_t e2e_warp_photo_lossgrad_workspace_floats(int B, int H, int W);
int e2e_warp_photo_lossgrad(const float* depth_tgt, const float* src, e2e_strides src_strides,
                            const float* tgt, e2e_strides tgt_strides, const float* K,
                            const float* inv_K, const float* T, int use_mask, int padding_mode,
                            int reg_kind, const float* reg_init_tgt, const float* reg_init_src,
                            const float* depth_src, float w_photo, float w_reg, float* loss_out,
                            float* g_depth_tgt, float* g_depth_src, float* workspace, int B, int H,
                            int W, void* stream);

/* Same launch for ONE keyframe pair (B = 1) when the caller already holds the pair's geometry on the HOST (poses are
 * dataset inputs): geometry12_host = rows of M = (K T)[:3,:3] inv(K)[:3,:3] (9 floats) followed by p4 = (K T)[:3,3]
 * (3 floats), so that c = d * M [x,y,1]^T + p4.  The 12 numbers travel as kernel arguments: no device round trip and no
 * workgroup barrier before the first loads. */
int e2e_warp_photo_lossgrad_hostgeo(const float* depth_tgt, const float* src, e2e_strides src_strides,
                                    const float* tgt, e2e_strides tgt_strides,
                                    const float* geometry12_host, int use_mask, int padding_mode,
                                    int reg_kind, const float* reg_init_tgt, const float* reg_init_src,
                                    const float* depth_src, float w_photo, float w_reg, float* loss_out,
                                    float* g_depth_tgt, float* g_depth_src, float* workspace, int H,
                                    int W, void* stream);

/* Chained form: ONE kernel per step.  Each workgroup adds its partial loss sums (as 2^-36 fixed
 * point: the integer total does not depend on arrival order, so the result is bitwise reproducible)
 * into slot set `set_cur` of the workspace, and the launch also turns the complete slot set
 * `set_prev` of the PREVIOUS launch into that launch's loss (loss_prev_out[0..1]) and clears it
 * (set_prev < 0: nothing to finalise).  Sets are 0..7 and must alternate; the last launch of a chain
 * is finished by e2e_warp_photo_lossgrad_chain_flush.  geometry12_host non-NULL (B == 1) replaces
 * K / inv_K / T as in e2e_warp_photo_lossgrad_hostgeo.  The workspace (same size function) must be
 * zero-filled once before the first chained launch.  A per-workgroup sum that is negative, not
 * finite or above 2.6e8 / #workgroups turns the loss into NaN (the gradients are unaffected). */
int e2e_warp_photo_lossgrad_chain(const float* depth_tgt, const float* src, e2e_strides src_strides,
                                  const float* tgt, e2e_strides tgt_strides, const float* K,
                                  const float* inv_K, const float* T, const float* geometry12_host,
                                  int use_mask, int padding_mode, int reg_kind, const float* reg_init_tgt,
                                  const float* reg_init_src, const float* depth_src, float w_photo,
                                  float w_reg, int set_cur, int set_prev, float* loss_prev_out,
                                  float* g_depth_tgt, float* g_depth_src, float* workspace, int B, int H,
                                  int W, void* stream);
int e2e_warp_photo_lossgrad_chain_flush(float* workspace, int set, int reg_kind, float* loss_out, int B,
                                        int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* RGB-D unprojection and the PointFusion map step -- gradslam (un-vendored dependency; semantics */
/* per SURVEY.md Appendix A), reference call sites online_adaption.py:347-363, :461-469, :642      */
/* ------------------------------------------------------------------------------------------ */

/* gradslam RGBDImages.vertex_map / normal_map / global_vertex_map / global_normal_map and the
 * fusion confidence alpha = exp(-|V|^2 / alpha_den) (alpha_den = 2 sigma^2 + 1e-7) in one pass.
 * depth (B,H,W); K, pose (B,4,4); outputs channels-last (B,H,W,3) / alpha (B,H,W).
 * V, Nm, Ng, alpha may be NULL.  Invalid (zero) depth gives zero rows. */
int e2e_vertex_normal_maps(const float* depth, const float* K, const float* pose, float alpha_den,
                           float* V, float* Nm, float* Vg, float* Ng, float* alpha, int B, int H,
                           int W, void* stream);
/* d/d(depth) of sum(g_V . V) + sum(g_Vg . Vg); either gradient may be NULL.  The normal map is not
 * differentiated (nothing on the reference path takes its gradient). */
int e2e_vertex_maps_bwd(const float* depth, const float* K, const float* pose, const float* g_V,
                        const float* g_Vg, float* g_depth, int B, int H, int W, void* stream);

/* gradslam.geometry.geometryutils.transform_pointcloud (online_adaption.py:642): out = R p + t for
 * (n,3) points, T (4,4).  transpose_rotation_only=1 gives R^T p (its autograd wrt the points). */
int e2e_transform_points(const float* points, const float* T, float* out, int64_t n,
                         int transpose_rotation_only, void* stream);

/* PointFusion map step (gradslam update_map_fusion) for ONE live frame against ONE resident map.
 * The map is four capacity-sized arrays (points/normals/colors (cap,3), ccounts (cap)) of which
 * the first M rows are live.  workspace: e2e_pf_workspace_bytes(cap, H, W) bytes.
 *
 * e2e_pf_associate = find_active_map_points + find_similar_map_points +
 *   find_best_unique_correspondences: per map point the pixel it projects to and its flags, per
 *   pixel the winning map point (max confidence, then min distance, then min index).
 * e2e_pf_table materialises one of the three index tables as int64 rows [n, h, w]
 *   (which = 0 active (ascending n), 1 similar (ascending n), 2 unique (ordered by pixel)); the row
 *   count goes to *count_out (device int64).  rows must hold min(M, H*W) resp. M rows.
 * e2e_pf_fuse_append = fuse_with_map: confidence-weighted merge of the matched points, then the
 *   frame's unmatched valid pixels are appended in row-major order; *new_count_out (device int64)
 *   receives the new live row count (rows beyond capacity are dropped -- size the map for the run). */
int64_t e2e_pf_workspace_bytes(int64_t map_capacity, int H, int W);
int e2e_pf_associate(const float* map_points, const float* map_normals, const float* map_ccounts,
                     int64_t M, const float* K, const float* pose, const float* Vg, const float* Ng,
                     float dist_th, float dot_th, void* workspace, int64_t map_capacity, int H, int W,
                     void* stream);
int e2e_pf_table(int which, int64_t M, void* workspace, int64_t map_capacity, int H, int W,
                 long long* rows, long long* count_out, void* stream);
int e2e_pf_fuse_append(float* map_points, float* map_normals, float* map_colors, float* map_ccounts,
                       int64_t M, int64_t map_capacity, const float* depth, const float* Vg,
                       const float* Ng, const float* rgb, const float* alpha, void* workspace, int H,
                       int W, long long* new_count_out, void* stream);

/* The same two steps on a map whose live size is DEVICE data (online_adaption.py:347-363 keeps the global
 * map on the device; gradslam reads its length on the host, which costs one synchronisation per keyframe).
 * map_count_dev: int64[3] in device memory = {M, scratch, sticky overflow flag}; e2e_pf_associate_dev
 * reads M, e2e_pf_fuse_append_dev reads M and writes the size after the append back to [0]
 * (clamped to map_capacity; [2] then holds the size that would have been needed, else stays 0).
 * Grids are sized by map_capacity, every argument is constant from keyframe to keyframe, the host reads
 * nothing: results equal e2e_pf_associate / e2e_pf_fuse_append called with the same M. */
int e2e_pf_associate_dev(const float* map_points, const float* map_normals, const float* map_ccounts,
                         const long long* map_count_dev, const float* K, const float* pose, const float* Vg,
                         const float* Ng, float dist_th, float dot_th, void* workspace, int64_t map_capacity,
                         int H, int W, void* stream);
/* Target cloud of the odometry: the map points the last e2e_pf_associate[_dev] found ACTIVE (find_active_map_points against
 * the previous frame, online_adaption.py:362 through PointFusion._localize), every dsratio-th of them in ascending map order
 * (gradslam downsample_pointclouds), gathered with their normals.  tgt_count_dev: int64[3] {targets written, active points,
 * sticky overflow flag (more targets than tgt_capacity)} -- device data like the map size. */
int e2e_pf_active_subsample_dev(const float* map_points, const float* map_normals, const long long* map_count_dev,
                                int64_t map_capacity, void* workspace, int H, int W, int dsratio, float* tgt,
                                float* tgt_normals, long long* tgt_count_dev, int64_t tgt_capacity, void* stream);
int e2e_pf_fuse_append_dev(float* map_points, float* map_normals, float* map_colors, float* map_ccounts,
                           long long* map_count_dev, int64_t map_capacity, const float* depth, const float* Vg,
                           const float* Ng, const float* rgb, const float* alpha, void* workspace, int H, int W,
                           void* stream);

/* ------------------------------------------------------------------------------------------ */
/* chamferdist.knn_points, K = 1, D = 3 (loss/losses.py:3,57)                                   */
/* ------------------------------------------------------------------------------------------ */

/* p1 (n1,3) queries, p2 (n2,3) references -> dists (n1) squared L2, idx (n1) int64 (first minimum
 * wins = smallest index among equal distances).  algorithm: 0 = auto, 1 = brute force (LDS-tiled,
 * fp32 VALU bound), 2 = exact uniform grid (counting-sort build + shell search with a provable
 * stopping rule + brute-force finish for the few unbounded queries).  Both give IDENTICAL results.
 * workspace: e2e_knn1_workspace_bytes(n1, n2) bytes. */
int64_t e2e_knn1_workspace_bytes(int64_t n1, int64_t n2);
int e2e_knn1_fwd(const float* p1, int64_t n1, const float* p2, int64_t n2, float* dists,
                 long long* idx, void* workspace, int algorithm, void* stream);
/* g_p1 = 2 g_dists (p1 - p2[idx]) */
int e2e_knn1_bwd(const float* g_dists, const float* p1, const float* p2, const long long* idx,
                 int64_t n1, float* g_p1, void* stream);
/* The other half of chamferdist's knn_points backward (un-vendored, SURVEY.md Appendix A: "d p2 = scatter of the negative"):
 *   g_p2[idx[i]] += -2 g_dists[i] (p1[i] - p2[idx[i]])        (n2,3)
 * reached by ChamferDistance(..., bidirectional=True) at train_depth.py:690-692, whose reverse term searches the
 * differentiable cloud.  Collisions are summed as 2^-48 fixed-point integers (order independent, bitwise reproducible);
 * scratch_fixed: 3*n2+1 int64, zeroed here.  A non-finite / >= 4096 contribution turns the whole result into NaN. */
int e2e_knn1_bwd_ref(const float* g_dists, const float* p1, const float* p2, const long long* idx,
                     int64_t n1, int64_t n2, long long* scratch_fixed, float* g_p2, void* stream);

/* Persistent index over one reference set: build once, query many times (ICP iterations, the
 * refinement steps of one keyframe all search the same map).  `index`: caller-owned buffer of
 * e2e_knn1_workspace_bytes(max_queries, n2) bytes; results are identical to e2e_knn1_fwd. */
int e2e_knn1_index_build(const float* p2, int64_t n2, int64_t max_queries, void* index, void* stream);
int e2e_knn1_index_query(const float* p1, int64_t n1, int64_t n2, int64_t max_queries, void* index,
                         float* dists, long long* idx, void* stream);
/* The index over a RESIDENT reference set (the global map of online_adaption.py:638-645) whose live size
 * is device data: *n2_dev points (0 < *n2_dev <= n2_capacity) are indexed, the host never reads the count.
 * `index`: e2e_knn1_index_capacity_bytes(max_queries, n2_capacity) bytes, allocated once for the run;
 * all launch arguments are constant between builds (the 3-D loss launches can sit in a captured hipGraph).
 * Results are identical to e2e_knn1_fwd on the first *n2_dev points. */
int64_t e2e_knn1_index_capacity_bytes(int64_t max_queries, int64_t n2_capacity);
int e2e_knn1_index_build_dev(const float* p2, const long long* n2_dev, int64_t n2_capacity, int64_t max_queries,
                             void* index, void* stream);
int e2e_knn1_index_query_dev(const float* p1, int64_t n1, int64_t n2_capacity, int64_t max_queries, void* index,
                             float* dists, long long* idx, void* stream);
/* The resident index at a caller-chosen resolution (cells per axis, 4 .. 256) for query sets the map-tuned rule does not fit:
 * frame-to-model odometry (PointFusion._localize behind online_adaption.py:362) asks 19 200 queries of a sparse target set,
 * possibly decimetres away while the pose is still wrong -- a coarse grid bounds the walk over empty cells.  Identical results
 * at any resolution.  ref_points / warm_idx: both NULL (cold search) or both set (candidates from an earlier search). */
int64_t e2e_knn1_index_capacity_bytes_res(int64_t max_queries, int64_t n2_capacity, int cells_per_axis);
int e2e_knn1_index_build_dev_res(const float* p2, const long long* n2_dev, int64_t n2_capacity, int64_t max_queries,
                                 void* index, int cells_per_axis, void* stream);
int e2e_knn1_index_query_dev_res(const float* p1, int64_t n1, const float* ref_points, const long long* warm_idx,
                                 int64_t n2_capacity, int64_t max_queries, void* index, int cells_per_axis,
                                 float* dists, long long* idx, void* stream);
/* The same query for queries that are the pixels of an image (row-major, `row_len` pixels per row, n1 a whole number of rows -- the
 * back-projected depth map of the 3-D loss, online_adaption.py:638-645): the lanes of a wave take 8 x 8 pixel tiles, whose points share
 * their grid cells, instead of 64 consecutive pixels of a row.  Results are identical; row_len = 0 (or sizes that are not multiples of 8)
 * falls back to the plain order. */
int e2e_knn1_index_query_dev_image(const float* p1, int64_t n1, int row_len, int64_t n2_capacity, int64_t max_queries, void* index,
                                   float* dists, long long* idx, void* stream);
/* ... with a WARM START: warm_idx[i] (NULL: none; may be the same buffer as idx) is the result of an earlier query for a point near
 * p1[i] against the same reference set -- the three refinement steps of a keyframe (online_adaption.py:259-327) query the same pixels with
 * slightly moved depths.  The candidate's distance bounds the search from the start; the result is still the exact nearest neighbour
 * (lowest index among ties), whatever the candidates are -- entries outside [0, live points) are ignored.  ref_points: the reference
 * points the index was built over (unsorted, rows of 3 floats). */
int e2e_knn1_index_query_dev_image_warm(const float* p1, int64_t n1, int row_len, const float* ref_points, const long long* warm_idx,
                                        int64_t n2_capacity, int64_t max_queries, void* index, float* dists, long long* idx, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* disp -> depth, median scaling, regulariser, metrics, optimiser                                */
/* ------------------------------------------------------------------------------------------ */

/* torch.median over all elements (online_adaption.py:295,343): the LOWER median (rank (n-1)/2) by
 * radix select; *value_out (device).  workspace: e2e_median_workspace_bytes() bytes; afterwards it
 * also holds the smallest index whose value equals the median and the number of elements that hold it (used by
 * the scale chain's autograd). */
int64_t e2e_median_workspace_bytes(void);
int e2e_median_lower(const float* x, int64_t n, float* value_out, void* workspace, void* stream);

/* online_adaption.py:282,292-298 for the n = F*H*W elements of the stacked disparities:
 *   delta = 1/disp ; ratio = median_gt / median(delta) ; depth = ratio * delta.
 * median_gt, median_delta, ratio_out are device scalars.  The backward is the exact autograd chain,
 * including the term torch routes to the element that IS the median:
 *   g_delta = ratio*g + [delta_i == median] * (-(ratio/median_delta) * sum(g*delta)) / #{delta_j == median} ;
 *   g_disp = -delta^2 g_delta      (torch.median(x) differentiates as evenly_distribute_backward: the elements that HOLD
 *   the median value share its gradient equally -- one element unless values tie exactly).
 * workspace (shared by fwd and bwd of one step): e2e_depth_scale_workspace_bytes() bytes. */
int64_t e2e_depth_scale_workspace_bytes(void);
int e2e_depth_scale_fwd(const float* disp, const float* median_gt, float* delta, float* depth,
                        float* median_delta, float* ratio_out, void* workspace, int64_t n,
                        void* stream);
int e2e_depth_scale_bwd(const float* g_depth, const float* delta, const float* median_gt,
                        const float* median_delta, float* g_disp, void* workspace, int64_t n,
                        void* stream);
/* The same chain with the elements that receive the median's gradient NAMED by the caller (device int32 indices into the
 * n stacked predictions, 1..64 of them, sharing the gradient equally; n_elements == 0: e2e_depth_scale_bwd).  For callers
 * that hold torch.median's own choice: among 614 400 fp32 depths the neighbours of the median lie ~1e-6 apart, closer than two
 * correct evaluations of the network agree, so WHICH element it is belongs to the evaluation, not to the algorithm --
 * the parity tests name the CPU evaluation's elements here and compare everything else (online_adaption.py:295-298). */
int e2e_depth_scale_bwd_at(const float* g_depth, const float* delta, const float* median_gt,
                           const float* median_delta, const int* elements, int n_elements,
                           float* g_disp, void* workspace, int64_t n, void* stream);

/* The fixed-scale form of train_depth.py:331-345 (`depth = 1 / disp` then `depth *= ABLATION.scaling_depth`): delta = 1 / disp
 * (may be NULL), depth = delta * scale; backward g_disp = -(g_depth * scale) / disp^2. */
int e2e_depth_fixed_scale_fwd(const float* disp, float scale, float* delta, float* depth, int64_t n,
                              void* stream);
int e2e_depth_fixed_scale_bwd(const float* g_depth, const float* disp, float scale, float* g_disp,
                              int64_t n, void* stream);

/* depth_reguralizer (losses.py:134-148) as a stand-alone op: out = mean |a-b| (kind 1) or
 * mean (a-b)^2 (kind 2); backward wrt b with upstream device scalar g_out.
 * workspace: e2e_reduce_workspace_floats() floats. */
int64_t e2e_reduce_workspace_floats(void);
int e2e_mean_diff_fwd(const float* a, const float* b, int64_t n, int kind, float* out,
                      float* workspace, void* stream);
int e2e_mean_diff_bwd(const float* a, const float* b, const float* g_out, int64_t n, int kind,
                      float* g_b, void* stream);

/* depth_metrics / compute_depth_errors (losses.py:162-201): out7 = abs_rel, sq_rel, rmse, rmse_log,
 * a1, a2, a3 over the kept pixels (mask_zero_gt = 1 drops gt == 0: the "TUM" rule). */
int e2e_depth_metrics(const float* gt, const float* pred, int64_t n, int mask_zero_gt, float* out7,
                      float* workspace, void* stream);

/* torch.optim.Adam step (training_utils.py:23-25; amsgrad off, no weight decay) over ONE flat,
 * 16-byte-aligned fp32 buffer per state; `step` is the 1-based step count. */
int e2e_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int step, void* stream);
/* The data-parallel form (one sequence per GPU): `grad_sums` is the flat bucket after ONE all-reduce(SUM)
 * over the ranks, `participants` (device scalar, all-reduced as the bucket's extra tail element) the number of ranks that
 * took a refinement step; the update uses grad_sums / max(participants, 1).  participants == 1: identical to
 * e2e_adam_step. */
int e2e_adam_step_mean(float* params, const float* grad_sums, const float* participants, float* exp_avg,
                       float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                       int step, void* stream);
/* The hipGraph-safe form: the step count lives on the device.  schedule (device, 8-byte aligned) holds for t = 1..len the
 * pair { lr / (1 - beta1^t), sqrt(1 - beta2^t) } exactly as the host computes it for e2e_adam_step (doubles rounded to
 * fp32); the launch uses entry step_counter[0] (1-based, clamped to len) and a second one-thread kernel advances the
 * counter, so a captured step replays through the bias corrections.  participants may be NULL (= 1). */
int e2e_adam_step_resident(float* params, const float* grad_sums, const float* participants, float* exp_avg,
                           float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                           const float* schedule, int schedule_len, int* step_counter, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Depth network convolutions -- depth_estimation/networks.py:44-57,157-189,277-292             */
/* fp32 implicit GEMM on v_mfma_f32_32x32x2_f32, NHWC activations.                               */
/* ------------------------------------------------------------------------------------------ */
#define E2E_ACT_NONE 0
#define E2E_ACT_RELU 1
#define E2E_ACT_ELU 2
#define E2E_ACT_DISP 3   /* 10*sigmoid(x)+0.01 (networks.py:290) */

/* torch weight (Cout,Cin,KH,KW) -> w_fwd [(kh,kw,ci)][ld_fwd] and/or w_bwd [(kh,kw,co)][ld_bwd]
 * (k-major GEMM operands; the padding columns must have been zeroed by the caller). */
int e2e_conv_weight_layouts(const float* w, int Cout, int Cin, int KH, int KW, float* w_fwd,
                            int ld_fwd, float* w_bwd, int ld_bwd, void* stream);

/* The same for `nlayers` layers in one launch (all layouts go stale together after an optimiser
 * step).  `desc`: DEVICE array of 10 int64 per layer {w, w_fwd, w_bwd (addresses, 0 = skip), Cout,
 * Cin, KH, KW, ld_fwd, ld_bwd, 0}. */
int e2e_conv_weight_layouts_batched(const long long* desc, int nlayers, void* stream);
/* (descriptor slot 9, "reserved" in round 1: a float* per-output-channel factor folded into w_bwd only, or 0) */

/* out (B,Ho,Wo,Cout) = act( scale[c] * conv(x) + shift[c] (+ residual) ), x = the VIRTUAL input
 * cat( nearest_upsample(src0 (B,Hs/up,Ws/up,C1), up), src1 (B,Hs,Ws,Cin-C1) ) padded by `pad`
 * (pad_mode 0 zeros, 1 reflection) -- upsample, concat and padding are gather arithmetic, never
 * materialised (networks.py:283-287, :186-188).  scale/shift: folded eval-mode BatchNorm and/or
 * bias (either may be NULL).  Cin % 16 != 0 (the RGB stem) takes a scalar-gather path that also
 * applies (v - in_sub) * in_mul to the image (networks.py:50). */
int e2e_conv2d_fwd(const float* src0, const float* src1, int C1, int up, const float* w_fwd,
                   int ld_fwd, const float* scale, const float* shift, const float* residual,
                   float* out, int B, int Hs, int Ws, int Cin, int Cout, int KH, int KW, int stride,
                   int pad, int pad_mode, int act, float in_sub, float in_mul, float* workspace,
                   void* stream);
/* Layers with few output tiles (deep, small-spatial) split the reduction over workgroups and add the
 * slices in a fixed order.  workspace for e2e_conv2d_fwd (rows = B*Ho*Wo, cols = Cout, K = KH*KW*Cin)
 * and e2e_conv2d_bwd_data (rows = B*(Hs+2p)*(Ws+2p), cols = Cin, K = KH*KW*Cout); NULL disables it. */
int64_t e2e_conv2d_splitk_workspace_floats(int64_t rows, int cols, int K);
/* floats of workspace for e2e_conv2d_bwd_data* on an input-gradient domain of (B, Hd, Wd, cols) with K = KH * KW * Cout: the split-K
 * slabs of a stride-1 layer, or the per-(tap, parity class) slabs of the stride-2 class form (4 x 4 x the largest class). */
int64_t e2e_conv2d_bwd_data_workspace_floats(int B, int Hd, int Wd, int cols, int K, int stride);
/* The head of every convolution GEMM workspace (e2e_conv_workspace_flag_floats() floats) holds the hand-off flags of the stream-K
 * kernels: zero outside a launch (a launch clears what it raised), so the owner zeroes that region ONCE after allocating the buffer.
 * Its int32 word e2e_conv_streamk_error_index() is raised, and stays raised, if a workgroup ever timed out waiting for a partial tile
 * (a bounded spin instead of a GPU hang); read it wherever the host synchronises anyway. */
int e2e_conv_workspace_flag_floats(void);
int e2e_conv_streamk_error_index(void);
/* Tuned forms (tools/gemm_tune.py, tests/test_gpu_conv.py): the same operators with the GEMM decomposition given PER CALL -- the
 * library keeps no mutable tuning state.  tile_m x tile_n in {64x64, 128x64, 128x128, 128x32, 32x128, 32x64, 64x32, 32x32};
 * ksplit >= 1: K slices (split-K slabs + reduction launch); ksplit < 0: stream-K on -ksplit persistent workgroups (64 x 64 tiles:
 * equal shares of the flattened (tile, K chunk) space, partial tiles combined in-launch in K order); tile_m == 0: the built-in choice.
 * workspace: e2e_conv_tuned_workspace_floats(GEMM rows, GEMM columns) floats, flag region zeroed.
 * e2e_conv2d_bwd_weight_scaled_tuned: the implicit-GEMM backward-weight kernels with `target_workgroups` pixel slices x tiles;
 * workspace e2e_conv2d_wgrad_tuned_workspace_floats(...). */
int64_t e2e_conv_tuned_workspace_floats(int64_t rows, int cols);
int e2e_conv2d_fwd_tuned(const float* src0, const float* src1, int C1, int up, const float* w_fwd, int ld_fwd, const float* scale,
                         const float* shift, const float* residual, float* out, int B, int Hs, int Ws, int Cin, int Cout, int KH,
                         int KW, int stride, int pad, int pad_mode, int act, float in_sub, float in_mul, float* workspace, int tile_m,
                         int tile_n, int ksplit, void* stream);
int e2e_conv2d_bwd_data_fused_tuned(const float* da, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs, int Ws, int Cin, int Cout,
                                    int Ho, int Wo, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, const float* x_in,
                                    int in_act, const float* pre_add, float* workspace, int tile_m, int tile_n, int ksplit, void* stream);
int64_t e2e_conv2d_wgrad_tuned_workspace_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW, int has_bias, int target_workgroups);
int e2e_conv2d_bwd_weight_scaled_tuned(const float* da, const float* out_scale, const float* src0, const float* src1, int C1, int up,
                                       float* dw, float* dbias, float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho, int Wo,
                                       int KH, int KW, int stride, int pad, int pad_mode, int accumulate, float in_sub, float in_mul,
                                       int target_workgroups, void* stream);
/* host-only query of that choice: out3 (HOST pointer) = {bm, bn, K slices}. */
int e2e_conv_gemm_choice(int64_t rows, int cols, int K, int chunk_depth, int allow_split, int* out3_host);

/* dZ = dY * act'(Y) * scale[c]  (Y = the op's OUTPUT; scale may be NULL). */
int e2e_conv2d_act_bwd(const float* dy, const float* y, const float* scale, float* dz, int64_t n,
                       int C, int act, void* stream);
/* the same with dz += ... when accumulate != 0 (a tensor with several consumers, e.g. a BasicBlock's input that feeds
 * both the first convolution and the residual add: networks.py -> torchvision BasicBlock.forward). */
int e2e_conv2d_act_bwd_acc(const float* dy, const float* y, const float* scale, float* dz, int64_t n,
                           int C, int act, int accumulate, void* stream);

/* gradient wrt the virtual (padded when pad_mode == 1) input: dxp (B,Hs+2p,Ws+2p,Cin). 
 * workspace: e2e_conv2d_bwd_data_workspace_floats(B, Hs + 2 pp, Ws + 2 pp, Cin, KH * KW * Cout, stride) floats (NULL: no split). */
int e2e_conv2d_bwd_data(const float* dz, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs,
                        int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride,
                        int pad, int pad_mode, float* workspace, void* stream);
/* the same with dxp += ... when accumulate != 0 (pad_mode 0 only: the padded domain of a reflection-padded layer is folded
 * by e2e_conv2d_gather_adjoint, which has its own accumulate flags). */
int e2e_conv2d_bwd_data_acc(const float* dz, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs,
                            int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride,
                            int pad, int pad_mode, int accumulate, float* workspace, void* stream);
/* Backward-data for the launch plan of a whole network (e2ehip.netplan), which keeps d loss / d PRE-activation in every gradient
 * buffer: `da` is that gradient of this layer's output (no separate dY * act'(Y) pass; a folded BatchNorm scale is folded into
 * w_bwd by e2e_conv_weight_layouts_batched), and with in_act = 1 (ReLU) / 2 (ELU) the epilogue multiplies the result by
 * act'(.) of the tensor x_in (B,Hs,Ws,Cin) the forward convolution read, so dxp is already the pre-activation gradient of the
 * layer that produced x_in (pad_mode 0 only; reflection-padded layers apply it in e2e_conv2d_gather_adjoint_act).
 * pre_add (may be NULL; (B,Hs,Ws,Cin)): a second gradient with respect to the same tensor -- the residual branch of a BasicBlock
 * (networks.py / torchvision BasicBlock.forward `out += identity`) -- added before the act' factor:
 * dxp (+)= (dA * W^T + pre_add) * act'(x_in). */
int e2e_conv2d_bwd_data_fused(const float* da, const float* w_bwd, int ld_bwd, float* dxp, int B, int Hs,
                              int Ws, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride,
                              int pad, int pad_mode, int accumulate, const float* x_in, int in_act,
                              const float* pre_add, float* workspace, void* stream);
/* adjoint of the gather: dxp -> d_src0 (B,Hs/up,Ws/up,C1) [, d_src1 (B,Hs,Ws,Cin-C1)]; every output
 * element sums its reflect-pad copies and its up x up readers in a fixed order (no atomics). */
int e2e_conv2d_gather_adjoint(const float* dxp, int B, int Hs, int Ws, int Cin, int C1, int up,
                              int padded, float* d_src0, float* d_src1, int accumulate0,
                              int accumulate1, void* stream);

/* the same, each destination multiplied by act'(.) of the tensor it is the gradient of (src0 (B,Hs/up,Ws/up,C1) with act0, src1
 * (B,Hs,Ws,Cin-C1) with act1; 0 none, 1 ReLU, 2 ELU -- derivative taken from the activation's output). */
int e2e_conv2d_gather_adjoint_act(const float* dxp, int B, int Hs, int Ws, int Cin, int C1, int up,
                                  int padded, float* d_src0, float* d_src1, int accumulate0,
                                  int accumulate1, const float* src0, int act0, const float* src1,
                                  int act1, void* stream);

/* dW (Cout,Cin,KH,KW) [and dbias (Cout) when non-NULL] = split-K GEMM over the output pixels with a
 * fixed-order slab reduction.  workspace: e2e_conv2d_wgrad_workspace_floats(...) floats. */
int64_t e2e_conv2d_wgrad_workspace_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW,
                                          int has_bias);
int e2e_conv2d_bwd_weight(const float* dz, const float* src0, const float* src1, int C1, int up,
                          float* dw, float* dbias, float* workspace, int B, int Hs, int Ws, int Cin,
                          int Cout, int Ho, int Wo, int KH, int KW, int stride, int pad,
                          int pad_mode, int accumulate, float in_sub, float in_mul, void* stream);

/* the same on `da` = the gradient BEFORE a folded BatchNorm scale: dW[co] = out_scale[co] * sum_p da[p,co] x[p,...] (the scale is
 * applied once per output element in the slab reduction instead of once per pixel in a separate pass); out_scale may be NULL. */
int e2e_conv2d_bwd_weight_scaled(const float* da, const float* out_scale, const float* src0,
                                 const float* src1, int C1, int up, float* dw, float* dbias,
                                 float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho,
                                 int Wo, int KH, int KW, int stride, int pad, int pad_mode,
                                 int accumulate, float in_sub, float in_mul, void* stream);

/* The slab reduction that ends a backward-weight call, as data (see e2e_conv2d_bwd_weight_scaled_deferred): the reference has no counterpart --
 * torch.autograd runs cuDNN's backward-filter per layer (loss.backward(), online_adaption.py:323) -- this is launch economy of the static plan. */
typedef struct e2e_wgrad_reduce_desc {
    const float* slabs;     /* S partial slabs [Mpad][Npad] */
    float* dw;              /* (Cout,Cin,KH,KW) */
    float* dbias;           /* (Cout) or NULL */
    const float* scale;     /* per-output-channel factor or NULL */
    int S, Mpad, Npad, Cout, Cin, KH, KW, has_bias, accumulate;
    int zl;                 /* 8 or 2: waves that share the slabs of one group of 64 quads (fixes the association of the sum) */
    long long first_item;   /* filled by e2e_wgrad_reduce_batch_prepare */
} e2e_wgrad_reduce_desc;
int e2e_conv2d_bwd_weight_scaled_deferred(const float* da, const float* out_scale, const float* src0,
                                          const float* src1, int C1, int up, float* dw, float* dbias,
                                          float* workspace, int B, int Hs, int Ws, int Cin, int Cout, int Ho,
                                          int Wo, int KH, int KW, int stride, int pad, int pad_mode,
                                          int accumulate, float in_sub, float in_mul,
                                          e2e_wgrad_reduce_desc* desc_out_host, void* stream);
long long e2e_wgrad_reduce_batch_prepare(e2e_wgrad_reduce_desc* descs_host, int n);
int e2e_wgrad_reduce_batched(const e2e_wgrad_reduce_desc* descs_dev, int n, long long total_items, void* stream);

/* Many device-to-device copies in one launch (launch economy of the static plan; the reference has no counterpart): n descriptors
 * {src, dst, bytes (a positive multiple of 16, both pointers 16-byte aligned), first_item} in device memory, prepared on the host. */
typedef struct e2e_copy_desc {
    const void* src;
    void* dst;
    long long bytes;
    long long first_item;   /* filled by e2e_copy_batch_prepare */
} e2e_copy_desc;
long long e2e_copy_batch_prepare(e2e_copy_desc* descs_host, int n);
int e2e_copy_batched(const e2e_copy_desc* descs_dev, int n, long long total_items, void* stream);

/* ResNet stem max-pool, nn.MaxPool2d(3, 2, 1) (networks.py:53 -> torchvision resnet.maxpool): x (B,H,W,C) NHWC ->
 * y (B,(H-1)/2+1,(W-1)/2+1,C).  The backward keeps no index tensor: every input element re-derives the first maximum
 * (ATen's scan order) of the <= 4 windows that contain it; accumulate != 0 adds to dx, mul_relu != 0 multiplies the
 * result by [x > 0] (the stem's ReLU, whose output x is).  C % 4 == 0. */
int e2e_maxpool3x3s2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int e2e_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C,
                         int accumulate, int mul_relu, void* stream);
/* The same pair with the position of each window's maximum kept in one byte per output element (argmax (B,Ho,Wo,C) uint8 = kh * 3 + kw
 * of the first maximum in ATen's scan order): the backward reads the <= 4 windows' bytes instead of re-scanning the input (the launch
 * plan's form; x is only read for mul_relu). */
int e2e_maxpool3x3s2_fwd_idx(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C, void* stream);
int e2e_maxpool3x3s2_bwd_idx(const float* x, const unsigned char* argmax, const float* dy, float* dx, int B, int H, int W, int C,
                             int accumulate, int mul_relu, void* stream);

/* eval-mode BatchNorm with a TRAINABLE affine (the reference freezes parameters whose name contains "bn",
 * online_adaption.py:182-184, so encoder.layerN.0.downsample.1 keeps training): scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale, rstd = 1 / sqrt(var + eps) (rstd may be NULL). */
int e2e_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                const float* running_var, float eps, float* scale, float* shift, float* rstd, int C,
                void* stream);
/* y[i] = relu?( z[i] * scale[i % C] + shift[i % C] + residual[i] ) over n NHWC elements (shift / residual may be NULL).
 * With C = 1 and no residual / relu this is Conv1x1(1, 1, bias) / ScaleLayer of the scale-learning experiments
 * (networks.py:191-215). */
int e2e_affine_fwd(const float* z, const float* scale, const float* shift, const float* residual, int relu,
                   float* y, int64_t n, int C, void* stream);
int64_t e2e_affine_bwd_workspace_floats(int C);
/* d gamma[c] = sum_p dy[p,c] (z[p,c] - mean[c]) rstd[c], d beta[c] = sum_p dy[p,c] over P pixels (mean / rstd NULL:
 * 0 / 1, required for C = 1); two-stage fixed-order reduction; accumulate != 0 adds to the outputs. */
int e2e_affine_bwd(const float* dy, const float* z, const float* mean, const float* rstd, int64_t P,
                   int C, float* dgamma, float* dbeta, int accumulate, float* workspace, void* stream);

/* upsample() of networks.py:218-221 [+ torch.cat with the skip tensor, :283-286]: x (B,h,w,C1), skip (B,2h,2w,C2) or
 * NULL -> y (B,2h,2w,C1+C2), all NHWC.  (Inside the decoder this is fused into the next convolution's gather.) */
int e2e_upsample2_concat(const float* x, const float* skip, float* y, int B, int h, int w, int C1,
                         int C2, void* stream);

/* The 1-channel disparity head: Conv3x3(reflect) 16 -> 1 (+ activation), networks.py:271-272,289-290.
 * x (B,H,W,16) NHWC, w (1,16,3,3), y (B,H,W).  Backward: dz = dY*act' (e2e_conv2d_act_bwd) ->
 * dx (B,H,W,16), dw (1,16,3,3), dbias (1); any of dx / dw may be NULL.
 * workspace: e2e_head_workspace_floats() floats. */
int e2e_head_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                 int Cin, int act, void* stream);
int64_t e2e_head_workspace_floats(void);
int e2e_head_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* dbias,
                 float* workspace, int B, int H, int W, int Cin, void* stream);

/* e2e_head_bwd whose dx is multiplied by act'(x) (in_act 1 ReLU / 2 ELU, from the activation's output x): the pre-activation
 * gradient of the layer that produced x (launch plan form, see e2e_conv2d_bwd_data_fused). */
int e2e_head_bwd_act(const float* dz, const float* x, const float* w, float* dx, float* dw, float* dbias,
                     float* workspace, int B, int H, int W, int Cin, int in_act, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Frame-to-model ICP odometry (gradslam odometry providers; MODEL.odom: icp | gradicp)          */
/* ------------------------------------------------------------------------------------------ */

/* One Gauss-Newton reduction of point-to-plane ICP: for every source point s_i (n,3) with nearest
 * target t_i = tgt[idx[i]] and normal n_i: A_i = [n_i, s_i x n_i], b_i = n_i . (t_i - s_i).
 * out29 (device, float64) = upper triangle of A^T A (21, row-major), A^T b (6), inlier count,
 * sum b_i^2.  Points with dists[i] >= dist_thresh^2 are skipped (dist_thresh < 0: keep all).
 * workspace: e2e_icp_workspace_bytes() bytes. */
int64_t e2e_icp_workspace_bytes(void);
int e2e_icp_normal_equations(const float* src, const float* tgt, const float* tgt_normals,
                             const long long* idx, const float* dists, float dist_thresh, int64_t n,
                             double* out29, void* workspace, void* stream);

/* The rest of an odometry iteration ON THE DEVICE (gradslam odometry providers "icp" / "gradicp" behind PointFusion.step,
 * online_adaption.py:111-122,362; configs/config.yaml:30-35 ships odom: gradicp): solve (A^T A + lambda I) xi = A^T b, the
 * se(3) exponential, GradICP's smooth damping update and T <- exp(.) T, all float64, from the 29 sums of
 * e2e_icp_normal_equations -- so the numiters iterations of a keyframe are a fixed launch sequence without a host round trip.
 *   state   : e2e_icp_state_doubles() float64 (T, xi, lambda, trace ...), initialised by e2e_icp_state_init (T = I, lambda = damp)
 *   T32     : (4,4) float32, the running transform for e2e_transform_points(src -> cur)
 *   step32  : (4,4) float32, GradICP's trial step for e2e_transform_points(cur -> next)
 *   pose_out: (4,4) float32 = T . prev_pose after every update (may be NULL)
 * mode 0 "icp":     phase 0 = solve, T <- exp(xi) T.
 * mode 1 "gradicp": phase 0 = solve, step32 <- exp(xi), remember err / cnt;  phase 1 (out29 = the reduction at the trial pose):
 *                   delta = err' / cnt' - err / cnt ; lambda *= 1/lmax + (lmax - 1/lmax) / (1 + B exp(-B2 nu delta)) ;
 *                   T <- exp(xi / (1 + exp(clip(nu delta, +-60)))) T.
 * Fewer than 6 inliers stop the iteration for good (later updates leave T alone), as the host loop's `break`. */
int64_t e2e_icp_state_doubles(void);
int e2e_icp_state_init(double* state, float* T32, float* step32, const float* prev_pose, float* pose_out,
                       double damp, void* stream);
int e2e_icp_update(const double* out29, double* state, float* T32, float* step32, const float* prev_pose,
                   float* pose_out, int mode, int phase, double lambda_max, double B, double B2, double nu,
                   void* stream);
/* e2e_icp_normal_equations followed by e2e_icp_update as TWO launches (the update folds the per-workgroup partial sums itself). */
int e2e_icp_reduce_update(const float* src, const float* tgt, const float* tgt_normals, const long long* idx,
                          const float* dists, float dist_thresh, int64_t n, void* workspace, double* state, float* T32,
                          float* step32, const float* prev_pose, float* pose_out, int mode, int phase,
                          double lambda_max, double B, double B2, double nu, void* stream);
/* Source cloud of the odometry (gradslam downsample_rgbdimages): every dsratio-th pixel in both directions of the live
 * frame's world vertex map Vg (H,W,3), row-major -> src (ceil(H/ds) * ceil(W/ds), 3).  A selected pixel without depth sets
 * *status (device int32; the resident path serves network-predicted depths, which are positive everywhere). */
int e2e_icp_source_subsample(const float* Vg, const float* depth, int H, int W, int dsratio, float* src,
                             int* status, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Off-by-default losses: loss value(s) AND the gradient for a unit upstream  */
/* gradient in one pass each; workspace: e2e_aux_workspace_floats() floats.                      */
/* ------------------------------------------------------------------------------------------ */
int64_t e2e_aux_workspace_floats(void);

/* loss/losses.py:119-132 disparity_smoothness_loss.  disp (B,1,H,W) contiguous, img (B,C,H,W)
 * through element strides.  loss_out[0] = x term, loss_out[1] = y term (the loss is their sum);
 * g_disp (B,1,H,W) = d(sum)/d(disp) or NULL. */
int e2e_smoothness_lossgrad(const float* disp, const float* img, e2e_strides img_strides, int B, int C,
                            int H, int W, float* loss_out, float* g_disp, float* workspace,
                            void* stream);

/* loss/losses.py:84-95 geometric_consistency_loss on n elements (mask as float, already expanded).
 * stats_out3 = {loss, sum(mask), gradient normaliser}; the `sum(mask) > 10000` gate is evaluated
 * on the device (loss and gradients are 0 when it is closed).  g_* both NULL or both given. */
int e2e_geometric_consistency_lossgrad(const float* warped_depth, const float* interpolated_depth,
                                       const float* mask, int64_t n, float* stats_out3,
                                       float* g_warped, float* g_interpolated, float* workspace,
                                       void* stream);

/* loss/losses.py:151-160 depth_gt_loss: mean |prediction * mask - sparse_gt| over n elements. */
int e2e_masked_l1_lossgrad(const float* prediction, const float* sparse_gt, const float* sparse_mask,
                           int64_t n, float* loss_out, float* g_prediction, float* workspace,
                           void* stream);

/* train_depth.py:657-661: mean over (b,y,x) of min over the C stacked error maps (B,C,H,W); the
 * gradient goes to the first minimal channel. */
int e2e_min_reprojection_lossgrad(const float* errors, int B, int C, int H, int W, float* loss_out,
                                  float* g_errors, float* workspace, void* stream);

/* train_depth.py's operator-by-operator loss assembly (the fused kernels cover the default flags; these serve min-reprojection /
 * auto-masking / smoothness):  out (B,C,H,W) = x (strided view) * mask (B,1,H,W) (:713-714);  channel mean of stacked error maps
 * (B,C,H,W) -> (B,1,H,W) (:630) and its adjoint (adjoint = 1: x (B,1,H,W) -> out (B,C,H,W) = x / C);  mean-normalised disparity
 * d / (mean_hw(d) + 1e-7) per image (:768-770) and, with g != NULL, its adjoint for the upstream gradient g (workspace: B * 130
 * floats). */
int e2e_mask_mul(const float* x, e2e_strides x_strides, const float* mask, int B, int C, int H, int W,
                 float* out, void* stream);
int e2e_channel_mean(const float* x, int B, int C, int H, int W, int adjoint, float* out, void* stream);
int e2e_mean_normalize(const float* d, const float* g, int B, int H, int W, float* out, float* workspace,
                       void* stream);

/* mean of values[i] over the elements whose gate[i] != 0, and weight x its gradient: the 3-D point loss
 * (online_adaption.py:638-645; loss/losses.py:57-63 `torch.mean(dists)`) over the valid-depth pixels without the
 * reference's boolean indexing (dynamic shape + host sync): gate = the depth map.  out3 = {mean, count, weight / count};
 * g_values (may be NULL) = weight * [gate != 0] / count.  workspace: e2e_aux_workspace_floats() floats. */
int e2e_masked_mean_lossgrad(const float* values, const float* gate, int64_t n, float weight, float* out3,
                             float* g_values, float* workspace, void* stream);

/* train_depth.py:224-237 process_disparity: disp_pair (2,1,H,W) = net(img), net(flip(img)) ->
 * out (1,1,H,W); bwd: g_out (H,W) -> g_disp_pair (2,1,H,W). */
int e2e_disp_blend_fwd(const float* disp_pair, int H, int W, float* out, void* stream);
int e2e_disp_blend_bwd(const float* g_out, int H, int W, float* g_disp_pair, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E2ESLAM_H */
