/*
 * smplraster.h - C ABI of libsmplraster_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the SMPL decoder + orthographic projection + visibility mask +
 * 31-part / silhouette soft-rasteriser of akashsengupta1997/indirect_learning_pose-shape.
 * The reference has no FFI: the path is a composition of TensorFlow ops inside Keras
 * Layer/Lambda callables.  Each entry point below replaces the TF graph of one of those
 * callables (cited as file:line of the reference) and is what a binding for that callable
 * would call.  The Python host side (`indirect_learning_pose-shape_amd/keras_smpl/ modules`)
 * binds them with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous row-major fp32 / int32 / int16 data,
 *    owned by the caller for the duration of the launch (the library allocates nothing);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are
 *    asynchronous, never synchronise and are HIP-graph capturable;
 *  - return value: 0 on success, a positive hipError_t if a launch failed, or a negative
 *    SMPLR_E* code for an argument error; `smplr_last_error()` describes the last failure
 *    on the calling thread;
 *  - B = meshes in the batch, V = vertices (6890), VP = ceil(V / vertex_sampling) projected
 *    vertices, W = raster width/height, P = body parts (31), J = 24 joints.
 */
#ifndef SMPLRASTER_H
#define SMPLRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPLR_ABI_VERSION 7
#define SMPLR_NJ 24            /* joints                                   */
#define SMPLR_KPAD 220         /* 10 betas + 207 pose features, padded     */
#define SMPLR_CHUNK 8          /* raster vertex-list padding granule       */

#define SMPLR_EINVAL (-1)      /* bad size / null pointer                  */
#define SMPLR_EUNSUPPORTED (-2)

int smplr_abi_version(void);
const char *smplr_last_error(void);
/* Test hooks for the per-device launch state (kernels that need more than 48 KB of dynamic LDS have their attribute
 * raised once per (kernel, device); the reference's multi_gpu_model towers, train.py:205-210, share nothing of the kind).
 *   smplr_debug_device_ordinal(n)  n >= 0: the library takes n for the current device's ordinal in that table from now on
 *                                  (a one-GPU box can then walk the table as a second device would); n < 0: hipGetDevice()
 *                                  again.  Returns the previous setting (-1 = real).
 *   smplr_debug_lds_attr_sets()    how many times the attribute has been set so far.                              */
int smplr_debug_device_ordinal(int fake_ordinal);
int smplr_debug_lds_attr_sets(void);
/* sha256 (64 hex digits) over the sources this library was built from - the .hip and .h files of csrc/ (sorted by
 * name) and this header, concatenated; the host side recomputes it from the files beside the library and refuses a stale
 * library (`_lib.load`).  No reference counterpart: the reference ships no compiled code.                      */
const char *smplr_build_id(void);

/* ---- SMPLLayer.call: keras_smpl/batch_smpl.py:96-153 --------------------------------- */

/* Rodrigues (batch_smpl.py:255-276, 230-253), pose feature (:122), joints from betas
 * (J = J_template + J_dirs * beta, algebraically :106-115) and the 24-joint kinematic chain
 * (batch_global_rigid_transformation, :168-228).
 *   x     (B, x_stride)  rows [cam(num_cam) | theta(72) | beta(10)]
 *   coef  (220, ld)      k-MAJOR, ld = smplr_coef_ld(B) = B rounded up to 32: column n =
 *                        [beta(10) | pose_feature(207) | 0,0,0] of mesh n - the blend GEMM's A operand,
 *                        laid out so that the matrix cores read it without a transpose
 *   coef3 (smplr_coef3_bytes(B) bytes) the same columns split into three bf16 terms per entry and laid
 *                        out as MFMA A-fragments - the operand of smplr_blend3_fwd.  coef or coef3 may be
 *                        NULL (at least one is required).
 *   Rs    (B,24,9)  J (B,24,3)  A (B,24,12) = rows 0..2 of the reference's (4,4) A
 *   J_transformed (B,24,3)  (batch_smpl.py:131, :216)                                      */
int smplr_coef_ld(int B);
size_t smplr_coef3_bytes(int B);
int smplr_pose_fwd(const float *x, int x_stride, int num_cam, int B,
                   const float *J_template, const float *J_dirs, const int32_t *parents,
                   float *coef, void *coef3, float *Rs, float *J, float *A, float *J_transformed,
                   void *stream);

/* Backward of the above.  dcoef (B,220), dA (B,24,12), dJ_transformed (B,24,3) or NULL,
 * dcam (B,4) or NULL (d[k_u,k_v,u0,v0] from the projection, smplr_skin_bwd).
 * Writes dx (B, x_stride) columns [0, num_cam+82): camera columns = dcam (first 4) or 0.     */
int smplr_pose_bwd(const float *x, int x_stride, int num_cam, int B,
                   const float *J_dirs, const int32_t *parents,
                   const float *Rs, const float *J, const float *A,
                   const float *dcoef, const float *dA, const float *dJ_transformed,
                   const float *dcam, float *dx, void *stream);

/* Blend shapes (batch_smpl.py:106-108 and :126-128 as ONE fp32-MFMA GEMM):
 *   v_posed (B,N3) = coef^T x blend (220,N3) + v_template (N3),  N3 = 3*V; coef (220, smplr_coef_ld(B))
 *   k-major as smplr_pose_fwd writes it.
 * blend rows 0..9 = shapedirs, 10..216 = posedirs, 217..219 = 0.                            */
int smplr_blend_fwd(const float *coef, const float *blend, const float *v_template,
                    int B, int N3, float *v_posed, void *stream);

/* dcoef (B,220) = dv_posed (B,N3) x blend^T.  blend_t (N3,224) is blend transposed with rows
 * zero-padded to 224 floats (a second constant, so that neither GEMM transposes the big
 * matrix).  workspace: smplr_blend_bwd_workspace(B,N3) bytes (split-K partials, summed in order). */
size_t smplr_blend_bwd_workspace(int B, int N3);
int smplr_blend_bwd(const float *dv_posed, const float *blend_t, int B, int N3,
                    float *dcoef, void *workspace, void *stream);

/* The same two GEMMs on the bf16 matrix cores with fp32-grade operands ("bf16x3"; the default of
 * the Python host side).  Every fp32 operand is the exact sum of three bf16 numbers (3 x 8 = 24
 * significant bits); a product is accumulated in fp32 from its six partial products of relative
 * size >= 2^-16 (what is dropped is < 2^-24 of the product, below an fp32 product's own rounding).
 * The constant is split and laid out in MFMA fragment order once:
 *   smplr_blend3_pack(blend (220,N3) fp32) -> pk_fwd (smplr_blend3_fwd_bytes(N3) bytes) and/or
 *   pk_bwd (smplr_blend3_bwd_bytes(N3) bytes); either may be NULL.
 * smplr_blend3_fwd / smplr_blend3_bwd then have the semantics of smplr_blend_fwd / smplr_blend_bwd
 * (dv_posed, v_posed, dcoef are fp32 in memory; the forward's per-step operand is smplr_pose_fwd's
 * coef3, split once per mesh instead of once per column tile; workspace: smplr_blend3_bwd_workspace). */
int smplr_coef3_pack(const float *coef, int B, void *coef3, void *stream);   /* coef (k-major) -> coef3 */
size_t smplr_blend3_fwd_bytes(int N3);
size_t smplr_blend3_bwd_bytes(int N3);
int smplr_blend3_pack(const float *blend, int N3, void *pk_fwd, void *pk_bwd, void *stream);
int smplr_blend3_fwd(const void *coef3, const void *pk_fwd, const float *v_template,
                     int B, int N3, float *v_posed, void *stream);
/* smplr_pose_fwd + smplr_blend3_fwd in ONE launch (what the decoder runs): the first ceil(B/8) workgroups are the
 * pose kernel (Rs, J, A, J_transformed as smplr_pose_fwd writes them), every GEMM wave computes the coefficient rows
 * of its own 32 meshes itself instead of reading coef3 - same Rodrigues, same three-way split, so v_posed is what
 * the two separate calls give, bit for bit - and nothing is handed between workgroups.  The pose chain's latency
 * (10 us at B = 128) disappears under the GEMM.                                                          */
int smplr_pose_blend3_fwd(const float *x, int x_stride, int num_cam, int B, const float *J_template,
                          const float *J_dirs, const int32_t *parents, const void *pk_fwd,
                          const float *v_template, int N3, float *Rs, float *J, float *A,
                          float *J_transformed, float *v_posed, void *stream);
size_t smplr_blend3_bwd_workspace(int B, int N3);
int smplr_blend3_bwd(const float *dv_posed, const void *pk_bwd, int B, int N3,
                     float *dcoef, void *workspace, void *stream);

/* Linear-blend skinning (batch_smpl.py:135-145) with the orthographic projection
 * (projection.py:54-81) as an optional epilogue.
 *   verts (B,V,3) = (sum_j w[v][j] A[b][j]) . [v_posed;1]
 *   proj  (B,VP,3) = (u0 + k_u x, v0 + k_v y, z) of vertices 0, vs, 2vs, ...  (NULL: skip)
 *   cam = x (row stride x_stride; columns 0..3 = k_u,k_v,u0,v0), may be NULL iff proj NULL.
 *   lbs_top4 (V,8) or NULL: per vertex its (up to) 4 non-zero weights followed by their joint
 *   indices as floats - the sparse form of lbs_weights (real SMPL rows have <= 4 non-zeros);
 *   when given, T sums 4 terms instead of 24, bit-identically (zero terms add exactly 0).      */
int smplr_skin_fwd(const float *v_posed, const float *lbs_weights, const float *lbs_top4,
                   const float *A, const float *cam, int x_stride, int B, int V,
                   int vertex_sampling, float *verts, float *proj, void *stream);

/* Backward of skinning (+ projection epilogue).
 *   dverts (B,V,3) or NULL;  dproj (B,VP,3) or NULL (the rasterisers write its z column as 0:
 *   z only feeds the non-differentiable mask, compute_mask.py:30);  at least one is required.
 *   dv_posed (B,V,3), dA (B,24,12) (overwritten), dcam (B,4) or NULL = d[k_u,k_v,u0,v0].
 *   workspace: smplr_skin_bwd_workspace(B,V) bytes (per-block partials, summed in fixed order). */
size_t smplr_skin_bwd_workspace(int B, int V);
int smplr_skin_bwd(const float *dverts, const float *dproj,
                   const float *v_posed, const float *lbs_weights, const float *lbs_top4,
                   const float *A, const float *cam, int x_stride, int B, int V,
                   int vertex_sampling, float *dv_posed, float *dA, float *dcam, void *workspace,
                   void *stream);

/* Fused backward of the whole SMPLLayer (+ projection epilogue): smplr_skin_bwd -> smplr_blend_bwd ->
 * smplr_pose_bwd in three launches, the partial sums of the first two folded into the third
 * (fixed summation order).  Same semantics as chaining the three entry points above.
 * The blend GEMM uses blend3_bwd (smplr_blend3_pack's pk_bwd) when it is not NULL, else blend_t.
 * seg_part / seg_vslot / seg_nsplit (NULL, NULL, 0 to omit): the segmentation rasteriser's gradient as
 * smplr_seg_bwd leaves it when called with dproj = NULL - per-row-block slot sums in its workspace - plus
 * the vertex -> slot map of the forward and smplr_seg_bwd_nsplit(B,W) (row blocks per mesh: 8 image rows each,
 * 24 while the batch gives every CU one to four row blocks).  The skinning backward then gathers
 * d(seg)/d(proj) by vertex, summing the row blocks in the order the merge kernel would have (bit-identical),
 * and adds it to dproj (if given): one launch and one (B,VP,3) round trip less.
 * workspace: smplr_smpl_bwd_workspace(B,V) bytes.                                              */
size_t smplr_smpl_bwd_workspace(int B, int V);
int smplr_smpl_bwd(const float *dverts, const float *dproj,
                   const float *seg_part, const int16_t *seg_vslot, int seg_nsplit,
                   const float *dJ_transformed, const float *x, int x_stride, int num_cam, int B, int V, int vertex_sampling,
                   const float *blend_t, const void *blend3_bwd,
                   const float *lbs_weights, const float *lbs_top4,
                   const float *J_dirs, const int32_t *parents, const float *Rs, const float *J, const float *A,
                   const float *v_posed, float *dx, void *workspace, void *stream);

/* ---- orthographic_project: keras_smpl/projection.py:54-81 ------------------------------ */
int smplr_project_fwd(const float *verts, const float *cam, int x_stride, int B, int V,
                      int vertex_sampling, float *proj, void *stream);
/* dverts (B,V,3) fully written (zeros at unsampled vertices), dcam (B,4).                   */
int smplr_project_bwd(const float *dproj, const float *verts, const float *cam, int x_stride,
                      int B, int V, int vertex_sampling, float *dverts, float *dcam,
                      void *stream);

/* ---- compute_mask: keras_smpl/compute_mask.py:12-108 (stateless semantics) ------------- */
/* mask (B,VP) in {1,500}: 1 for the arg-max-z vertex of each occupied cell of a
 * grid_wh x grid_wh grid of rounded (half-to-even) pixel positions, lowest index on ties;
 * with ref_compat != 0 vertex 1 is also visible whenever a cell is empty (:99).            */
int smplr_visibility(const float *proj, int B, int VP, int grid_wh, int ref_compat,
                     float *mask, void *stream);

/* ---- projects_to_seg: keras_smpl/projects_to_seg.py:9-69 -------------------------------- */
/* Part table, built once on the host from part_vertices.pkl (projects_to_seg.py:18-24,36-37):
 *   part_pos (K) int32: positions into the VP-long vertex list, part-major (K = 6879); every position at most
 *     once (the reference's tables are partitions; the backward keeps one record slot per vertex);
 *   part_off (P+1) int32: CSR offsets of the P parts.
 * mask values must be >= 0 (the reference produces {1, 500}).
 * workspace: smplr_seg_workspace(B,VP,W,P,K) bytes of scratch.
 * seg (B,W,W,P+1): channel 0 = 1 - clip(sum_p score_p, 0, 1); channel 1+p =
 *   max_v exp(-mask_v * |proj_v - (c,r)|); rows flipped (:68).  Scores below the fp32
 *   underflow threshold (mask*d >= 104) are exactly 0, as they are in fp32 arithmetic.
 * rec (B,S,4) fp32, S = smplr_seg_slots(P,K): the mesh's compact record list
 *   (u, v, mask^2, vertex position as int32 bits) - first the far-reaching (visible) vertices,
 *   part-major, then the nearest-pixel-only ones; the last slot is a header of int32 words:
 *   [0] used slots, [1] 1 if some far-reaching weight is not 1, [2] length of the far-reaching list (padded per
 *   part to 4: smplr_seg_raster_plan), [3] -1.  Only the used prefix and the header are written.
 * arg (B,W,W,32) int16, slot = channel: [0] = 1 iff 0 <= sum_p <= 1 (the clip's pass-through
 *   gate); [1+p] = index into rec[b] of the maximising vertex of part p, or -1 when no vertex
 *   contributes a non-zero score.  rec + arg are what the backward needs (no proj/mask/seg).
 * vslot (B,VP) int16 or NULL: for each vertex position its slot in rec[b], -1 if it has no record
 *   (needed only by the gather form of the backward, see smplr_smpl_bwd).
 * Requires P <= 31, VP <= 32767, W <= 160.                                                     */
int smplr_seg_slots(int P, int K);
size_t smplr_seg_workspace(int B, int VP, int W, int P, int K);
int smplr_seg_fwd(const float *proj, const float *mask, int B, int VP, int W,
                  const int32_t *part_pos, const int32_t *part_off, int P, int K,
                  void *workspace, float *seg, int16_t *arg, float *rec, int16_t *vslot, void *stream);

/* compute_mask + projects_to_seg in one call (the model.py:113-118 pair as the fused decoder
 * runs it): same results as smplr_visibility followed by smplr_seg_fwd, the z-buffer being built
 * inside the binning workgroup; mask (B,VP) is an OUTPUT here.  Same workspace as smplr_seg_fwd. */
int smplr_vis_seg_fwd(const float *proj, int B, int VP, int W, int grid_wh, int ref_compat,
                      const int32_t *part_pos, const int32_t *part_off, int P, int K,
                      void *workspace, float *mask, float *seg, int16_t *arg, float *rec, int16_t *vslot,
                      void *stream);

/* smplr_skin_fwd + smplr_vis_seg_fwd in ONE call of two launches (what the fused decoder runs for models whose
 * skinning rows have <= 4 non-zeros and no vertex sampling, V <= 7168): the binning workgroup of a mesh skins and
 * projects the mesh's vertices itself (batch_smpl.py:135-145 + projection.py:62-79, the arithmetic of smplr_skin_fwd
 * bit for bit) and writes verts (B,V,3) and proj (B,V,3) out, then goes on as smplr_vis_seg_fwd.  cam = the rows of x
 * (camera in columns 0..3), x_stride floats apart.  Same workspace and outputs as smplr_vis_seg_fwd.  verts, proj
 * and mask may each be NULL: the backward of the decoder needs none of them (the record list carries the projected
 * positions it uses), so a caller that only consumes seg saves their 82.7 + 82.7 + 27.6 KB per mesh of stores.
 * smplr_skin_vis_seg_fits(V, W, grid_wh) = 1 when the form applies (the mesh's staged (u, v), its z-buffer, the
 * pixel counters and the slot map fit the workgroup's LDS: W <= ~100 at V = 6890, grid_wh = 64); otherwise call
 * smplr_skin_fwd and smplr_vis_seg_fwd.                                                                            */
int smplr_skin_vis_seg_fits(int V, int W, int grid_wh);
int smplr_skin_vis_seg_fwd(const float *v_posed, const float *lbs_top4, const float *A, const float *cam,
                           int x_stride, int B, int V, int W, int grid_wh, int ref_compat,
                           const int32_t *part_pos, const int32_t *part_off, int P, int K, void *workspace,
                           float *verts, float *proj, float *mask, float *seg, int16_t *arg, float *rec,
                           int16_t *vslot, void *stream);

/* The two stages of smplr_seg_fwd / smplr_vis_seg_fwd as separate entry points (one launch each; calling
 * them back to back IS the fused call, bit for bit) - for callers that re-rasterise a binned batch and for
 * timing the pair loop (the dominant kernel) by itself:
 *   smplr_seg_bin     binning: proj (+ mask) -> rec, vslot and the workspace's per-mesh part offsets and pixel
 *                     lists.  grid_wh > 0: compute_mask runs inside (mask is an OUTPUT, as in smplr_vis_seg_fwd);
 *                     grid_wh = 0: mask is an INPUT (as in smplr_seg_fwd).
 *   smplr_seg_raster  the rasteriser proper over that workspace + rec -> seg, arg.                  */
int smplr_seg_bin(const float *proj, float *mask, int B, int VP, int W, int grid_wh, int ref_compat,
                  const int32_t *part_pos, const int32_t *part_off, int P, int K, void *workspace, float *rec,
                  int16_t *vslot, void *stream);
int smplr_seg_raster(int B, int W, int P, int K, const void *workspace, const float *rec, float *seg,
                     int16_t *arg, void *stream);
/* Measurement aid (bench.py's roofline): smplr_seg_raster with start / stop events ON the launch (hipExtLaunchKernel);
 * the call WAITS for the kernel and returns its own duration in *kernel_ms (host pointer) - begin to end on the device,
 * the quantity rocprofv3's kernel trace reports, without the dispatch gap that events recorded around a launch
 * include.  Not capturable, synchronises: never on the product path.                                            */
int smplr_seg_raster_timed(int B, int W, int P, int K, const void *workspace, const float *rec, float *seg,
                           int16_t *arg, float *kernel_ms, void *stream);

/* How smplr_seg_raster will run a batch - host arithmetic only, nothing is launched (bench.py and the tests count
 * the blocks that need more than one pass over their record list with it; the reference has no counterpart: its
 * rasteriser, projects_to_seg.py:41-56, materialises every pair).  info[8] receives
 *   [0] pair-lanes per workgroup, [1] part ranges (waves per 64 pair-lanes), [2] workgroups per mesh (tiles),
 *   [3] ints per mesh in the workspace's part-offset block, which starts at byte 0 of the workspace: entry [P] of a
 *       mesh's row is the length of its far-reaching record list, padded per part to 4 (`lbase`), entry [P + 1] is
 *       non-zero when some far-reaching weight is not 1,
 *   [4] the largest record table over the tiles, [5] the smallest, [6] 1 if some tile has more image rows under it
 *       than the table form takes (such tiles walk the list with scalar loads, whatever its length), [7] 0.
 * tile_records (tiles ints) or NULL: per tile the records ONE pass over the LDS table takes; a unit-weight mesh with
 * lbase records costs tile t ceil(lbase / tile_records[t]) passes (0 there = the tile is of kind [6]).  Returns the
 * number of tiles, 0 on bad sizes.                                                                               */
int smplr_seg_raster_plan(int B, int W, int P, int K, int32_t *info, int32_t *tile_records);

/* dproj (B,VP,3), fully written (z column and unreferenced vertices = 0).  Gradient goes to
 * the first arg-min vertex only (TF splits exact ties); it is 0 where the distance is 0 (TF:
 * NaN).  The score is recomputed from the arg-min record, so seg itself is not an input.
 * workspace: smplr_seg_bwd_workspace(B,W) bytes (per-row-block partial sums, merged in order).
 * dproj = NULL stops after the partial sums: the workspace (B, smplr_seg_bwd_nsplit(B,W), 5, 4096, 2) then IS
 * the result, to be handed to smplr_smpl_bwd together with the forward's vslot.
 * deterministic != 0 (also smplr_silh_bwd): a workgroup's per-vertex sums are accumulated as 64-bit fixed-point
 * integers (scale = a power of two from max|dseg| and the largest weight, resolution 2^-41 of the largest possible
 * term) instead of fp32 LDS atomics, whose result depends on arrival order in the last bits: the same inputs then give
 * the same gradient bit for bit on every launch (the reference's op is a pure function).  Everything downstream
 * (row-block merge, skinning, blend, pose backward) sums in a fixed order in either mode.             */
int smplr_seg_bwd_nsplit(int B, int W);
size_t smplr_seg_bwd_workspace(int B, int W);
int smplr_seg_bwd(const float *dseg, const int16_t *arg, const float *rec,
                  int B, int VP, int W, int P, int K, float *dproj, void *workspace, int deterministic,
                  void *stream);

/* ---- projects_to_seg + the loss head in one pass: keras_smpl/projects_to_seg.py:34-69 + model.py:119-120 +
 * focal_loss.py:10-46 ("fuses the seg tensor's last read into the loss") -----------------
 * The rasteriser's write-out phase holds a pixel's 32 raw scores in 8 adjacent lanes, so Reshape + softmax + the
 * focal loss at an INTEGER class map are its epilogue: labels (B, W, W) int32 as the output lies (rows flipped; an id
 * outside [0, 32) contributes no loss and no gradient, as in smplr_focal_fwd), class_w (32) or NULL, gamma >= 0
 * -> loss (B, W*W) per pixel (the value smplr_focal_fwd would return on the scores) and stats (B, W*W, 4) fp32 for
 * the backward: [k / sum_c exp(score_c) | k x (delta_0t - softmax_0 where the background clip's gate is open, else 0) |
 * k = q_t softmax_t | label bits].  seg may be NULL: the (B, W, W, 32) scores then never leave the chip (295 KB per mesh not written,
 * and not read twice by the loss kernels); arg and rec are written as by smplr_seg_fwd.  P must be 31.
 *   smplr_seg_raster_ex          = smplr_seg_raster with optional extras (after smplr_seg_bin): loss != NULL adds the
 *                                  epilogue (then labels and stats are required and seg may be NULL); vmax != NULL
 *                                  (B, W, W) receives each pixel's largest part score - the hint smplr_silh_fwd_hint
 *                                  takes; with loss = vmax = NULL it IS smplr_seg_raster
 *   smplr_skin_vis_seg_fwd_ex    = smplr_skin_vis_seg_fwd with the same extras (what the decoder runs when it is
 *                                  given a loss and / or renders the silhouette too); verts, proj, mask may each be NULL
 *   smplr_seg_loss_bwd           = smplr_seg_bwd fed with dloss (B, W*W) and stats instead of dseg: a lane rebuilds its
 *                                  channel's d loss / d score = A (delta_ct - softmax_c) - g_background from its own
 *                                  recomputed score, A = dloss q_t softmax_t (20 B per pixel read instead of 128);
 *                                  same workspace, dproj = NULL and deterministic as smplr_seg_bwd.
 * The softmax takes no max shift (scores lie in [0, 1]) and uses v_exp_f32; against smplr_focal_fwd/bwd on the written
 * scores the loss agrees to ~1e-5 relative (tests/test_gpu_loss_fused.py).                                       */
int smplr_seg_raster_ex(int B, int W, int P, int K, const void *workspace, const float *rec,
                        const int32_t *labels, const float *class_w, float gamma, float *seg, int16_t *arg,
                        float *loss, float *stats, float *vmax, void *stream);
int smplr_skin_vis_seg_fwd_ex(const float *v_posed, const float *lbs_top4, const float *A, const float *cam,
                              int x_stride, int B, int V, int W, int grid_wh, int ref_compat,
                              const int32_t *part_pos, const int32_t *part_off, int P, int K, void *workspace,
                              const int32_t *labels, const float *class_w, float gamma, float *verts, float *proj,
                              float *mask, float *seg, int16_t *arg, float *rec, int16_t *vslot, float *loss,
                              float *stats, float *vmax, void *stream);
int smplr_seg_loss_bwd(const float *dloss, const float *stats, const int16_t *arg, const float *rec, int B, int VP,
                       int W, int P, int K, float *dproj, void *workspace, int deterministic, void *stream);

/* ---- projects_to_silhouette: keras_smpl/projects_to_silhouette.py:14-44 ----------------- */
/* silh (B,W,W,2) = [1-s, s], s = max_v exp(-|proj_v-(c,r)|/1.2) over ALL VP vertices, rows
 * flipped; arg (B,W,W) int32 = maximising vertex.  workspace: smplr_silh_workspace(B,VP,W) B.  */
size_t smplr_silh_workspace(int B, int VP, int W);
int smplr_silh_fwd(const float *proj, int B, int VP, int W, float *silh, int32_t *arg,
                   void *workspace, void *stream);
/* The same with a per-pixel hint (B, W, W), laid out as the output: a score exp(-x), x >= the distance from the pixel
 * to SOME vertex - smplr_seg_raster_ex's vmax of the same meshes at the same W.  It only bounds the exact search
 * (W <= 48: the pixel-per-lane kernel; ignored otherwise): outputs are bit for bit those of smplr_silh_fwd.  0 = no hint
 * for that pixel; hint = NULL is smplr_silh_fwd.                                                                 */
int smplr_silh_fwd_hint(const float *proj, const float *hint, int B, int VP, int W, float *silh, int32_t *arg,
                        void *workspace, void *stream);
int smplr_silh_bwd(const float *dsilh, const float *silh, const int32_t *arg,
                   const float *proj, int B, int VP, int W, float *dproj, int deterministic, void *stream);

/* ---- loss head: model.py:119-120 + focal_loss.py:10-46 ----------- */
/* Reshape(W*W, C) + softmax + categorical_focal_loss fused: logits (npix, C) are the raw
 * rasteriser scores (seg: C = 32, silhouette: C = 2), loss (npix) is the per-pixel value the Keras
 * loss function returns:  sum_c y_c w_c (1 - p_c)^gamma (-log p_c),  p = clip(softmax, 1e-7, 1-1e-7).
 * Targets: EITHER labels (npix) int32 class ids (y = one-hot; an id outside [0,C) gives y = 0)
 * OR y_true (npix, C) fp32 (the reference's one-hot format, train.py:18-31); the other is NULL.
 * class_w (C) or NULL (= focal_loss.py:22-40's weights when weight_classes, else ones).
 * gamma = 0, class_w = NULL is Keras' categorical_crossentropy on the softmax output
 * (train_stage2_silhouette.py:85-86,226-229).  probs (npix, C) optional (NULL): the softmax,
 * i.e. the 'segs' model output.  bwd: dlogits = dloss[pix] * dL/dlogits, softmax recomputed;
 * the clip passes gradient only where eps <= softmax <= 1-eps (tf.clip_by_value).             */
int smplr_focal_fwd(const float *logits, const int32_t *labels, const float *y_true,
                    const float *class_w, float gamma, long long npix, int C,
                    float *loss, float *probs, void *stream);
int smplr_focal_bwd(const float *logits, const int32_t *labels, const float *y_true,
                    const float *class_w, float gamma, const float *dloss, long long npix, int C,
                    float *dlogits, void *stream);

/* ---- PReLU of the ENet encoder: encoders/encoder_enet_simple.py:21,37,50,58,79 */
/* Per-channel slope, NCHW fp32: y = x > 0 ? x : w[c] * x over x (N, C, HW).
 * bwd: gx = x > 0 ? gy : w[c] * gy;  gw[c] = sum over n, hw of (x > 0 ? 0 : gy * x), summed in a
 * fixed order through smplr_prelu_bwd_workspace(N,C,HW) bytes of partials (no atomics).          */
int smplr_prelu_fwd(const float *x, const float *w, long long N, int C, int HW, float *y, void *stream);
size_t smplr_prelu_bwd_workspace(long long N, int C, int HW);
int smplr_prelu_bwd(const float *x, const float *w, const float *gy, long long N, int C, int HW,
                    float *gx, float *gw, void *workspace, void *stream);

/* ---- BatchNormalization (+ PReLU) of the ENet encoder, training mode:
 *      encoders/encoder_enet_simple.py:19-21,35-37,48-50,56-58,76 ---------- */
/* x (N, C, HW) NCHW fp32.  Forward: batch statistics per channel over (N, HW) (biased variance), running_mean /
 * running_var updated in place as torch.nn.BatchNorm2d does (momentum = weight of the new value; unbiased
 * variance; either may be NULL), z = prelu(gamma (x - mean) rstd + beta, slope); slope (C) = NULL: no
 * activation.  save_mean / save_rstd (C) are what the backward needs besides x.
 * Backward: dz -> dx, dgamma, dbeta (C) and dslope (C, iff slope); y and x_hat are recomputed from x.
 * workspace: smplr_bn_workspace(N,C,HW) bytes (per-chunk partial sums, added in a fixed order).          */
size_t smplr_bn_workspace(long long N, int C, int HW);
int smplr_bn_fwd(const float *x, const float *gamma, const float *beta, const float *slope,
                 long long N, int C, int HW, float eps, float momentum,
                 float *running_mean, float *running_var, float *z, float *save_mean, float *save_rstd,
                 void *workspace, void *stream);
int smplr_bn_bwd(const float *x, const float *gamma, const float *beta, const float *slope,
                 const float *save_mean, const float *save_rstd, const float *dz,
                 long long N, int C, int HW, float *dx, float *dgamma, float *dbeta, float *dslope,
                 void *workspace, void *stream);

/* The tail of an ENet bottleneck in one pass (encoder_enet_simple.py:56-79: BatchNormalization ->
 * SpatialDropout2D -> Add -> PReLU):  out = prelu(plane_scale[n,c] * bn(x) + other, slope).
 * plane_scale (N*C) = the dropout factor of each (image, channel) plane (0 or 1/(1-p)); NULL = 1.
 * other (N,C,HW) = the bottleneck's other branch.  Backward: dout -> dx, dother, dgamma, dbeta, dslope.   */
int smplr_bn_res_fwd(const float *x, const float *gamma, const float *beta, const float *plane_scale,
                     const float *other, const float *slope, long long N, int C, int HW, float eps,
                     float momentum, float *running_mean, float *running_var, float *out,
                     float *save_mean, float *save_rstd, void *workspace, void *stream);
int smplr_bn_res_bwd(const float *x, const float *gamma, const float *beta, const float *plane_scale,
                     const float *other, const float *slope, const float *save_mean,
                     const float *save_rstd, const float *dout, long long N, int C, int HW,
                     float *dx, float *dother, float *dgamma, float *dbeta, float *dslope,
                     void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMPLRASTER_H */
