/* r3det_hip.h -- C ABI of libr3det_hip.so: the MI355X (gfx950) implementation of the
 * r3det custom-op hot path.
 *
 * One entry point per pybind function of the reference (paths relative to
 * /root/reference/r3det/ops); each declaration cites the interface it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 / int64 / uint8 data, row-major, contiguous;
 *     tensors are borrowed for the duration of the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every function
 *     only enqueues work on it and returns without synchronising;
 *   - no allocation happens inside the library: NMS takes a caller-provided workspace whose
 *     size comes from r3det_nms_workspace_bytes();
 *   - return value: 0 on success, a negative R3DET_E* code otherwise; nothing is thrown
 *     across the ABI.  r3det_error_string() names a code.
 *   - boxes are [cx, cy, w, h, theta(rad)] exactly as in the reference.
 */
#ifndef R3DET_HIP_H_
#define R3DET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The entry points below are the library's WHOLE dynamic symbol table: it is built with -fvisibility=hidden, and this
 * block gives the declarations default visibility (tests/test_abi.py compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define R3DET_OK 0
#define R3DET_EINVAL (-1)  /* bad argument (null pointer, negative size, bad enum)      */
#define R3DET_ELAUNCH (-2) /* hipLaunch / hipMemsetAsync reported an error              */
#define R3DET_EWS (-3)     /* workspace too small                                       */

/* geometry variants (angle conventions of the three operator families) */
#define R3DET_GEOM_V1 1 /* rbbox_geo / rnms: vertex+segment algorithm                   */
#define R3DET_GEOM_V2 2 /* mmcv / ml_nms_rotated: hull algorithm, standard vertex sign  */
#define R3DET_GEOM_V3 3 /* box_iou_rotated / nms_rotated: hull algorithm, negated sign  */

int r3det_abi_version(void);
const char* r3det_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * Rotated IoU
 * ------------------------------------------------------------------------------------- */

/* rbbox_geo_cuda.mat_iou_iof(rb1, rb2, iof)            rbbox_geo/src/rbbox_geo_cuda.cpp:13-18
 * kernel rbbox_geo/src/rbbox_geo_kernel.cu:231-268.   rb1 (n1,5), rb2 (n2,5) -> out (n1,n2). */
int r3det_rbbox_geo_mat_iou_iof(const float* rb1, int n1, const float* rb2, int n2, int iof,
                                float* out, void* ws, size_t ws_bytes, void* stream);

/* Optional scratch for the three (n1,n2) matrix entry points.  With ws == NULL (or a matrix of at most
 * 512 columns) the matrix is produced by one kernel; with a workspace of at least this many bytes wider
 * matrices take a streaming kernel + a load-balanced drain kernel over per-tile lists of the pairs that
 * may overlap (several times faster on assignment-shaped inputs).  About 1 byte per pair; it need not be
 * initialised. */
size_t r3det_iou_workspace_bytes(int n1, int n2);

/* rbbox_geo_cuda.vec_iou_iof(rb1, rb2, iof)            rbbox_geo/src/rbbox_geo_cuda.cpp:19-24
 * kernel rbbox_geo_kernel.cu:271-309.  out (max(n1,n2),), out[i] = f(rb1[i % n1], rb2[i % n2]). */
int r3det_rbbox_geo_vec_iou_iof(const float* rb1, int n1, const float* rb2, int n2, int iof,
                                float* out, void* stream);

/* box_iou_rotated_ext.overlaps(b1, b2, iou_or_iof)     box_iou_rotated/src/box_iou_rotated_ext.cpp:17-32
 * kernel box_iou_rotated/src/box_iou_rotated_cuda.cu:14-63.  iou_or_iof != 0 selects IoU,
 * 0 selects IoF (intersection / area of b1).  out (n1,n2). */
int r3det_box_iou_rotated_overlaps(const float* b1, int n1, const float* b2, int n2,
                                   int iou_or_iof, float* out, void* ws, size_t ws_bytes,
                                   void* stream);

/* obb_overlaps(bboxes1, bboxes2, mode, is_aligned=False)   ops/box_iou_rotated/box_iou_rotated_wrapper.py:8-64:
 * the call above followed by the wrapper's epilogue -- rows of b1-boxes and columns of b2-boxes with
 * min(w, h) < 1e-3 are zeroed (:53-60) -- in one more small launch instead of six framework ones (the
 * masked_fill alone rewrites the whole matrix). */
int r3det_obb_overlaps(const float* b1, int n1, const float* b2, int n2, int iou_or_iof, float* out, void* ws,
                       size_t ws_bytes, void* stream);

/* Pairwise companion of the call above: out[i] = overlap(b1[i], b2[i]), i < n.  Serves
 * obb_overlaps(is_aligned=True) (box_iou_rotated_wrapper.py:48-49), whose reference
 * implementation is a differentiable torch composition + convex_sort. */
int r3det_box_iou_rotated_overlaps_aligned(const float* b1, const float* b2, int n,
                                           int iou_or_iof, float* out, void* stream);

/* mmcv.ops.box_iou_rotated(b1, b2, mode, aligned)      call site core/bbox/iou_calculators/
 * rotate_iou2d_calculator.py:156 (third-party op; geometry restated from
 * ml_nms_rotated/src/box_iou_rotated_utils.h).  mode_flag 0 = iou, 1 = iof.
 * aligned == 0: out (n1,n2); aligned != 0: n1 must equal n2, out (n1,). */
int r3det_mmcv_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int mode_flag,
                               int aligned, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Rotated NMS.  As in mmcv's ext signature the score sort stays with the caller: `order`
 * holds the n original indices sorted by score (descending, stable); boxes are passed in
 * ORIGINAL order and gathered through `order` on the device.  `keep_out` (int64, capacity
 * n) receives original indices in score order, `count_out` (int32, 1 element, device) their
 * number.  Suppression uses the CUDA comparison IoU > thr.
 * ------------------------------------------------------------------------------------- */
size_t r3det_nms_workspace_bytes(int n);

/* rnms_ext.rnms(dets, thr)                             rnms/src/rnms_ext.cpp:11-19
 * host rnms/src/rcuda/rnms_kernel.cu:270-335.  dets6 (n,6) = [box5, score].  The reference
 * returns keep sorted ascending; with sort_ascending != 0 keep_out is rewritten in ascending
 * index order on the device (rnms_kernel.cu:331-334). */
int r3det_rnms(const float* dets6, const int64_t* order, int n, float thr, int sort_ascending,
               void* ws, size_t ws_bytes, int64_t* keep_out, int32_t* count_out, void* stream);

/* nms_rotated_ext.nms_rotated(dets, scores, thr)       nms_rotated/src/nms_rotated_ext.cpp:24-35
 * host nms_rotated/src/nms_rotated_cuda.cu:71-134.  dets5 (n,5). */
int r3det_nms_rotated(const float* dets5, const int64_t* order, int n, float thr, void* ws,
                      size_t ws_bytes, int64_t* keep_out, int32_t* count_out, void* stream);

/* ml_nms_rotated_cuda.ml_nms_rotated(dets, scores, labels, thr)
 *                                                      ml_nms_rotated/src/nms_rotated.h:23-39
 * host ml_nms_rotated/src/nms_rotated_cuda.cu:74-137.  labels (n,) int64; IoU of boxes with
 * different labels is 0 (box_iou_rotated_utils.h:316-322). */
int r3det_ml_nms_rotated(const float* dets5, const int64_t* labels, const int64_t* order, int n,
                         float thr, void* ws, size_t ws_bytes, int64_t* keep_out,
                         int32_t* count_out, void* stream);

/* mmcv.ops.nms_rotated(dets, scores, thr, labels)      call site core/post_processing/
 * bbox_nms_rotated.py:86 (third-party; v2 geometry, optional label column).
 * labels may be NULL (single-label). */
int r3det_mmcv_nms_rotated(const float* dets5, const int64_t* labels, const int64_t* order, int n,
                           float thr, void* ws, size_t ws_bytes, int64_t* keep_out,
                           int32_t* count_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused assignment.  The anchor heads hand (gt_bboxes, anchors) to mmdet's
 * MaxIoUAssigner with iou_calculator = RBboxOverlaps2D_v1 (models/dense_heads/
 * rotate_anchor_head.py:220-231 -> core/bbox/iou_calculators/rotate_iou2d_calculator.py:51-80),
 * which reduces the (n_gt x n_boxes) overlaps to per-box max / argmax, per-gt max / argmax and
 * the assignment.  This call produces those results without materialising the overlaps
 * (100 MB at 128 x 196 416).  The assigner itself is third-party (mmdet 2.19
 * core/bbox/assigners/max_iou_assigner.py, not in the reference tree); its rules as restated:
 *   a = -1;  if 0 <= max < neg_iou_thr: a = 0;  if max >= pos_iou_thr: a = argmax + 1;
 *   if match_low_quality: for gt i in order, if gt_max[i] >= min_pos_iou:
 *       gt_max_assign_all ? a[overlaps[i] == gt_max[i]] = i + 1 : a[gt_argmax[i]] = i + 1
 * argmax ties resolve to the smaller index.  geom: R3DET_GEOM_V1 / _V2 / _V3 (plain IoU, no
 * wrapper-level zeroing of thin boxes).  argmax_overlaps and the two gt_* outputs may be NULL
 * (both gt_* or neither).  ws: r3det_rbbox_assign_workspace_bytes(n_gt, n_boxes).
 * ------------------------------------------------------------------------------------- */
size_t r3det_rbbox_assign_workspace_bytes(int n_gt, int n_boxes);
int r3det_rbbox_assign(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes, float pos_iou_thr,
                       float neg_iou_thr, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                       int64_t* assigned_gt_inds, float* max_overlaps, int64_t* argmax_overlaps,
                       float* gt_max_overlaps, int64_t* gt_argmax_overlaps, void* ws, size_t ws_bytes,
                       void* stream);

/* Prepared columns.  The second operand of the training step's overlaps is the SAME anchor grid in every step
 * (models/dense_heads/rotate_anchor_head.py:220-231 -> mat_iou_iof, rbbox_geo_kernel.cu:231-268, which recomputes the
 * trigonometry of both boxes for every pair): r3det_iou_prepare_columns computes what the kernels need of a box list
 * used as columns -- the exact per-box records, the data of the conservative disjointness test, the bounding box of
 * every 256 consecutive columns -- ONCE into `prepared` (r3det_iou_prepared_bytes(n) bytes, 16-byte aligned); the
 * _prepared entry points take it next to the boxes themselves and give bit for bit the results of the plain ones.
 * geom: R3DET_GEOM_V1 / _V2 / _V3, the same in the prepare call and in every use.  The buffer begins with a 16-byte
 * header {magic, geometry, n, check} (round 6; no host-side state): every workgroup of a consumer compares it with its
 * own launch before reading anything else, and a buffer prepared for another geometry or column count -- or one that
 * was never prepared -- is NOT used: the call answers in its data, r3det_iou_mat_prepared with a matrix of NaN,
 * the assignment with max_overlaps = NaN and every box ignored (-1).  r3det_iou_prepared_check gives the same answer
 * on the host (R3DET_OK / R3DET_EINVAL; it copies the header back: a stream synchronisation -- once after preparing,
 * not per call).  A copy of a prepared buffer is a prepared buffer.  mode of r3det_iou_mat_prepared: as the plain
 * entry of the geometry takes it (v1 / v2: 1 = iof; v3: 0 = iof). */
size_t r3det_iou_prepared_bytes(int n);
int r3det_iou_prepare_columns(int geom, const float* boxes, int n, void* prepared, size_t prepared_bytes, void* stream);
int r3det_iou_prepared_check(const void* prepared, int geom, int n, void* stream);
int r3det_iou_mat_prepared(int geom, const float* b1, int n1, const float* b2, int n2, const void* prepared, int mode,
                           float* out, void* ws, size_t ws_bytes, void* stream);
int r3det_rbbox_assign_prepared(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes,
                                const void* prepared, float pos_iou_thr, float neg_iou_thr, float min_pos_iou,
                                int match_low_quality, int gt_max_assign_all, int64_t* assigned_gt_inds,
                                float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps,
                                int64_t* gt_argmax_overlaps, void* ws, size_t ws_bytes, void* stream);

/* The same call (prepared may be NULL: then r3det_rbbox_assign) that also writes mmdet's assigned LABELS: -1, and
 * gt_labels[assigned_gt_inds - 1] at the positives (mmdet 2.19 max_iou_assigner.py, the last lines of
 * assign_wrt_overlaps; called from r3det/models/dense_heads/rotate_anchor_head.py:220-231) from the kernel that writes
 * assigned_gt_inds -- the host-side form costs four elementwise launches per call.  gt_labels (n_gt,) int64 and
 * assigned_labels (n_boxes,) int64: both or neither. */
int r3det_rbbox_assign_labeled(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes, const void* prepared,
                               float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                               int gt_max_assign_all, int64_t* assigned_gt_inds, float* max_overlaps,
                               int64_t* argmax_overlaps, float* gt_max_overlaps, int64_t* gt_argmax_overlaps,
                               const int64_t* gt_labels, int64_t* assigned_labels, void* ws, size_t ws_bytes,
                               void* stream);

/* ---------------------------------------------------------------------------------------
 * Batched post-processing: multiclass_nms_rotated for all images of a step.
 * Replaces, for nms type 'v1' and boxes shared by the classes, the per-image Python sequence
 *   core/post_processing/bbox_nms_rotated.py:31-53,97-131  (threshold, mask indexing, labels)
 *   ops/rnms/rnms_wrapper.py:34-69                         (class offsets, rnms, ascending keep)
 * with the same results.  Two calls, because the suppression workspace is sized by the
 * largest per-image candidate count, which the host has to read in between:
 *
 *   r3det_mcnms_select : boxes (B,n,5), scores (B,n,K+1; last column = background).
 *       Candidates (score > score_thr) per image in row-major (anchor, class) order:
 *       cand_row / cand_label / cand_rank (int32) and cand_score, each (B, n*K) [cand_rank is
 *       scratch of the second call, which zeroes what it uses]; counts (B) int32, maxc (B) = max over the
 *       candidate boxes' five columns.  ws: r3det_mcnms_select_workspace_bytes(B, n).
 *   r3det_mcnms_v1 : cap >= max(counts), < 65536 (an image with more candidates is processed as its
 *       first cap candidates -- never out of bounds -- so a caller may guess cap, read counts
 *       afterwards and call again when the guess was short).  Per image: stable descending score sort,
 *       x, y += label * (maxc + 1), NMS v1 (IoU > iou_thr), keep ASCENDING by candidate
 *       index, first out_cap of them.  dets_out (B,out_cap,6) = [box, score], labels_out
 *       (B,out_cap) int64, keep_idx_out (B,out_cap) int64 = the kept candidate indices (may be
 *       NULL), counts_out (B) int32; rows beyond counts_out are not written.
 *       ws: r3det_mcnms_workspace_bytes(B, cap).
 *       The candidate arrays need not come from r3det_mcnms_select: with cand_row = 0..n-1,
 *       K = 1, caller-provided labels / scores and counts = n the call is batched_rnms
 *       (ops/rnms/rnms_wrapper.py:34-69) of one image: dets_out = cat(bboxes[keep], scores[keep]),
 *       keep_idx_out = keep.
 * ------------------------------------------------------------------------------------- */
size_t r3det_mcnms_select_workspace_bytes(int B, int n);
int r3det_mcnms_select(const float* boxes, const float* scores, int B, int n, int K, float score_thr,
                       int32_t* cand_row, int32_t* cand_label, float* cand_score, int32_t* cand_rank,
                       int32_t* counts, float* maxc, void* ws, size_t ws_bytes, void* stream);
size_t r3det_mcnms_workspace_bytes(int B, int cap);
int r3det_mcnms_v1(const float* boxes, int B, int n, int K, const int32_t* cand_row,
                   const int32_t* cand_label, const float* cand_score, int32_t* cand_rank,
                   const int32_t* counts, const float* maxc, int cap, float iou_thr, int out_cap, void* ws,
                   size_t ws_bytes, float* dets_out, int64_t* labels_out, int64_t* keep_idx_out,
                   int32_t* counts_out, void* stream);

/* batched_rnms(bboxes, scores, inds, nms_thr) (ops/rnms/rnms_wrapper.py:34-69) in one call on its raw inputs:
 * offset = label * (max over all five box columns + 1) on cx, cy, NMS v1 (IoU > nms_thr), keep ASCENDING.
 * bboxes (n,5), scores (n), inds (n) int64 or NULL (class-agnostic); dets_out (n,6) = [box, score] of the kept
 * rows, keep_out (n) int64, kept_out (1) int32; rows beyond kept_out are not written.  n <= 65472.
 * ws: r3det_batched_rnms_workspace_bytes(n), uninitialised. */
size_t r3det_batched_rnms_workspace_bytes(int n);
int r3det_batched_rnms(const float* bboxes, const float* scores, const int64_t* inds, int n, float nms_thr, void* ws,
                       size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, void* stream);

/* obb_batched_nms(bboxes, scores, inds, nms_thr) (ops/nms_rotated/nms_rotated_wrapper.py:78-98, five-column boxes)
 * the same way: offset = label * (max - min of the circumscribed horizontal boxes + 1) on cx, cy, boxes with
 * min(w, h) < 0.001 never kept and never suppress, NMS v3, keep in SCORE order.  Arguments and workspace as
 * r3det_batched_rnms. */
int r3det_obb_batched_nms(const float* bboxes, const float* scores, const int64_t* inds, int n, float nms_thr, void* ws,
                          size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, void* stream);

/* The same pipeline for the other nms types of multiclass_nms_rotated (bbox_nms_rotated.py:42-58):
 *   nms_type 1 : batched_rnms, identical to r3det_mcnms_v1.
 *   nms_type 3 : obb_batched_nms (nms_rotated_wrapper.py:23-59): x, y += label * extent with
 *       extent = (max - min of the candidates' circumscribed horizontal boxes) + 1 per image,
 *       boxes with min(w, h) < 0.001 never kept and never suppress, IoU v3, keep in SCORE
 *       order, first out_cap.
 *   nms_type 2 : ml_nms_rotated (ml_nms_rotated.py:6-36, kernel ml_nms_rotated_cuda.cu:11-73):
 *       no offsets, pairs of different labels never suppress, IoU v2, keep in SCORE order,
 *       first out_cap.
 * maxc is read by nms_type 1 only (may be NULL otherwise).  Outputs as r3det_mcnms_v1 except
 * the row order. */
int r3det_mcnms(int nms_type, const float* boxes, int B, int n, int K, const int32_t* cand_row,
                const int32_t* cand_label, const float* cand_score, int32_t* cand_rank, const int32_t* counts,
                const float* maxc, int cap, float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out,
                int64_t* labels_out, int64_t* keep_idx_out, int32_t* counts_out, void* stream);

/* r3det_mcnms with the PADDED result a detector hands on (round 5): what R3Det.simple_test returns per image after
 * multiclass_nms_rotated goes through rbbox2result (models/detectors/r3det.py:137-143, core/bbox/rtransforms.py:10-25)
 * or, image-parallel, through one gather of [max_per_img, 7] rows per image.  dets7_out: (B, out_cap, 7)
 * fp32 = [cx, cy, w, h, theta, score, label], `img_stride` floats from one image to the next (>= out_cap * 7: the
 * caller may leave room for a row of its own behind every image), rows beyond the kept count ZEROED; counts_out (B)
 * int32; count_f32_out (may be NULL): the count also as fp32 at count_f32_out[img * count_f32_stride]; overflow_out
 * (B) int32 (may be NULL): 1 when the image had more than `cap` candidates -- its result is then that of its first
 * cap candidates and the caller repeats the call with a larger cap.  Nothing here needs the host to look at a count
 * between r3det_mcnms_select and this call: with a fixed cap both are plain enqueues (stream capture records them).
 * Same candidate arrays, workspace and orders as r3det_mcnms; rows and values equal its (dets, labels) lists. */
int r3det_mcnms_padded(int nms_type, const float* boxes, int B, int n, int K, const int32_t* cand_row,
                       const int32_t* cand_label, const float* cand_score, int32_t* cand_rank, const int32_t* counts,
                       const float* maxc, int cap, float iou_thr, int out_cap, void* ws, size_t ws_bytes,
                       float* dets7_out, size_t img_stride, int32_t* counts_out, float* count_f32_out,
                       size_t count_f32_stride, int32_t* overflow_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Feature refinement (rotated feature-align sampler)
 * ------------------------------------------------------------------------------------- */

/* feature_refine_cuda.forward(features, best_bboxes, spatial_scale, points, output)
 *                                                      fr/src/feature_refine_cuda.cpp:24-42
 * kernel fr/src/feature_refine_kernel.cu:112-163.  features/output (N,C,H,W) NCHW,
 * best_bboxes (N*H*W,5); points in {1,5}.  output is caller-allocated and fully overwritten. */
int r3det_feature_refine_forward(const float* features, const float* best_bboxes, int N, int C,
                                 int H, int W, float spatial_scale, int points, float* output,
                                 void* ws, size_t ws_bytes, void* stream);

/* Optional scratch for the forward sampler (20 bytes per sample point).  With it the box ->
 * (tap offsets, bilinear weights) conversion runs once per position instead of once per
 * channel plane; results are bit-identical.  ws == NULL selects the workspace-free kernels. */
size_t r3det_fr_workspace_bytes(int N, int H, int W, int points);

/* Split form of the forward (points = 1; levels of 128 x 128 or 64 x 64, for which
 * r3det_fr_table_bytes returns non-zero): r3det_feature_refine_prepare turns the boxes of a level
 * into the sampler's tap table (8 bytes per position) -- e.g. for every pyramid level before the
 * module's convolutions run -- and r3det_feature_refine_forward_prepared is then a single launch.
 * Same results as r3det_feature_refine_forward.  R3DET_EINVAL for shapes / channel counts the
 * sampler kernel does not take (use the one-call form). */
size_t r3det_fr_table_bytes(int N, int H, int W);
int r3det_feature_refine_prepare(const float* best_bboxes, int N, int H, int W, float spatial_scale, float* table,
                                 void* stream);
int r3det_feature_refine_forward_prepared(const float* features, const float* table, int N, int C, int H, int W,
                                          float* output, void* stream);

/* The sampler with the module's two elementwise passes around it folded in
 * (fr/feature_refine_module.py:121-126: feat = conv_5_1(conv_1_5(x)) + conv_1_1(x);
 * out = x + fr(feat, boxes)):  output = residual + ((mixed_a + mixed_b) + sample(mixed_a + mixed_b)), same
 * operation order, bit-identical to the three launches it replaces; 3 reads + 1 write per element
 * instead of 8 passes.  Same table and shape rules as r3det_feature_refine_forward_prepared. */
int r3det_feature_refine_module_prepared(const float* mixed_a, const float* mixed_b, const float* residual,
                                         const float* table, int N, int C, int H, int W, float* output,
                                         void* stream);
/* (mixed_b may be NULL: mixed_a is then the already summed plane, output = residual + fr(mixed_a).)
 *
 * For channels_last pipelines: channels_last -> NCHW with the module's work in front of the sampler folded in,
 * out_nchw[n,c,p] = (a_nhwc[n,p,c] + bias_a[c]) + (b_nhwc[n,p,c] + bias_b[c]), a / b = the raw outputs of
 * conv_5_1 and conv_1_1 (their bias adds are separate launches after the convolution otherwise).  b_nhwc and
 * the biases may be NULL (plain layout switch, e.g. for the residual). */
int r3det_frm_mix_nchw(const float* a_nhwc, const float* b_nhwc, const float* bias_a, const float* bias_b, int N,
                       int C, int H, int W, float* out_nchw, void* stream);

/* The pre-NMS pool of ONE pyramid level for a whole batch -- what RAnchorHead._get_bboxes_single does per image
 * and level in front of multiclass_nms_rotated (models/dense_heads/rotate_anchor_head.py:626-673; with the previous
 * stage's boxes as anchors: rotate_retina_refine_head.py:147-196):
 *   scores = cls_score.permute(1, 2, 0).reshape(-1, C).sigmoid();  top nms_pre rows by scores.max(dim=1) when the
 *   level has more (in score order, ties by ascending row);  bboxes = delta2bbox_v1(anchors, deltas, max_shape).
 * cls_score (N, A*C, H, W) and bbox_pred (N, A*5, H, W) are read through their strides (4 element strides each:
 * NCHW or channels_last, no copy); anchors (H*W*A, 5), or (N, H*W*A, 5) when anchors_per_image != 0 (the refine
 * head: A = 1, the previous boxes).  Rows [row_offset, row_offset + min(nms_pre, H*W*A)) of pool_boxes (N, pool_rows,
 * 5) and pool_scores (N, pool_rows, C + 1) are written (last score column = 0, the background column
 * multiclass_nms_rotated drops): exactly the arrays r3det_mcnms_select reads.  max_x / max_y = W_img - 1 / H_img - 1
 * (centre clamp), or negative for none.  nms_pre <= 4096 and, when the level is cut, A*H*W <= 1 000 000.  ws: r3det_level_pool_workspace_bytes() bytes (0 when
 * the level keeps all rows). */
size_t r3det_level_pool_workspace_bytes(int N, int A, int H, int W, int nms_pre);
int r3det_level_pool(const float* cls_score, const long long* cls_strides, const float* bbox_pred,
                     const long long* reg_strides, const float* anchors, int anchors_per_image, int N, int A, int C,
                     int H, int W, int nms_pre, float max_ratio, float max_x, float max_y, float* pool_boxes,
                     float* pool_scores, int pool_rows, int row_offset, void* ws, size_t ws_bytes, void* stream);

/* The same for ALL pyramid levels of a head in one call (the `for` over mlvl_* of _get_bboxes_single,
 * rotate_anchor_head.py:626-673): level l's rows follow level l - 1's in the pool, exactly as the reference
 * concatenates them; row for row the result of num_levels r3det_level_pool calls, in three launches instead of
 * up to three per level.  cls_scores / bbox_preds / anchors: HOST arrays of num_levels device pointers; cls_strides /
 * reg_strides: 4 element strides per level; A / H / W: per level.  num_levels <= 8.  ws:
 * r3det_levels_pool_workspace_bytes() bytes. */
size_t r3det_levels_pool_workspace_bytes(int num_levels, int N, const int* A, const int* H, const int* W, int nms_pre);
int r3det_levels_pool(int num_levels, const float* const* cls_scores, const long long* cls_strides,
                      const float* const* bbox_preds, const long long* reg_strides, const float* const* anchors,
                      int anchors_per_image, int N, const int* A, int C, const int* H, const int* W, int nms_pre,
                      float max_ratio, float max_x, float max_y, float* pool_boxes, float* pool_scores, int pool_rows,
                      void* ws, size_t ws_bytes, void* stream);

/* channels_last (NHWC) forms of the sampler: `features` / `output` are (N, H, W, C) contiguous -- the memory of a
 * torch channels_last (N, C, H, W) tensor -- so that a channels_last pipeline needs no layout switch around
 * the FR module.  Same results, element for element, as the NCHW entry points (feature_refine_cuda.forward,
 * fr/src/feature_refine_cuda.cpp:24-42; kernel feature_refine_kernel.cu:112-163).  One wavefront per position,
 * lane <-> 4 channels: C % 4 == 0 and 16-byte aligned pointers, else R3DET_EINVAL (nothing launched).
 *   _forward_nhwc : output = features + sample(features)                      (1 read + 1 write per element)
 *   _module_nhwc  : the FeatureRefineModule tail (fr/feature_refine_module.py:121-126) in one launch:
 *                   P = (conv_a + bias_a) + (conv_b + bias_b), output = residual + (P + sample(P)), with conv_a /
 *                   conv_b the RAW outputs of conv_5_1(conv_1_5(x)) and conv_1_1(x), residual = x
 *                   (3 reads + 1 write per element; conv_b and the biases may be NULL). */
int r3det_feature_refine_forward_nhwc(const float* features, const float* best_bboxes, int N, int C, int H, int W,
                                      float spatial_scale, int points, float* output, void* stream);
int r3det_feature_refine_module_nhwc(const float* conv_a, const float* conv_b, const float* bias_a,
                                     const float* bias_b, const float* residual, const float* best_bboxes, int N,
                                     int C, int H, int W, float spatial_scale, int points, float* output,
                                     void* stream);

/* The channels_last forms for ALL pyramid levels of a pass (the `for` over the levels of
 * FeatureRefineModule.forward, fr/feature_refine_module.py:108-127) in one call: a level that takes the wide regions
 * form (level 0 of a 1024^2 input) is one launch, ALL other levels together are one more (a grid over the tile pairs
 * of every level: the coarse levels are launch-bound on their own).  Element for element the results of one
 * r3det_feature_refine_forward_nhwc / _module_nhwc call per level.  features / conv_a / conv_b / residual /
 * best_bboxes / outputs: HOST arrays of `levels` device pointers (conv_b may be NULL: no second addend); H, W,
 * spatial_scales: host arrays; N, C, points and the biases common to the levels. */
int r3det_feature_refine_forward_levels_nhwc(int levels, const float* const* features, const float* const* best_bboxes,
                                             int N, int C, const int* H, const int* W, const float* spatial_scales,
                                             int points, float* const* outputs, void* stream);
int r3det_feature_refine_module_levels_nhwc(int levels, const float* const* conv_a, const float* const* conv_b,
                                            const float* bias_a, const float* bias_b, const float* const* residual,
                                            const float* const* best_bboxes, int N, int C, const int* H, const int* W,
                                            const float* spatial_scales, int points, float* const* outputs, void* stream);

/* The TAIL of FeatureRefineModule.forward for all NCHW levels of a pass in one call (fr/feature_refine_module.py:
 * 108-127: per level `feat = conv_5_1(conv_1_5(x)) + conv_1_1(x)`, `out = x + fr(feat, boxes)` -- two elementwise passes
 * around the sampler, level by level), points = 1: outputs[l] = residual[l] + (P + sample(P)), P = conv_a[l] + conv_b[l],
 * element for element the three-step form.  The coarse levels are ONE grid (the plane sampler with both adds folded in)
 * that also builds the tap tables of the 128 x 128 / 64 x 64 levels, each of which is then one fused launch
 * (r3det_feature_refine_module_prepared's): 3 launches for a 1024^2 pyramid instead of 13.  Used by training steps (the
 * autograd node's forward) and NCHW inference alike.  ws: r3det_fr_module_levels_workspace_bytes() bytes, left holding
 * the tables.  R3DET_EINVAL: a level takes neither form (nothing was launched: run it level by level).  Pointer / shape
 * arrays are HOST arrays of `levels` <= 8 entries; all maps (N, C, H, W) contiguous, 16-byte aligned, W % 4 == 0. */
size_t r3det_fr_module_levels_workspace_bytes(int levels, int N, const int* H, const int* W);
int r3det_feature_refine_module_levels(int levels, const float* const* conv_a, const float* const* conv_b,
                                       const float* const* residual, const float* const* best_bboxes, int N, int C,
                                       const int* H, const int* W, const float* spatial_scales, int points,
                                       float* const* outputs, void* ws, size_t ws_bytes, void* stream);

/* The per-level loop of FeatureRefineModule.forward (fr/feature_refine_module.py:115-127) in one call:
 * `levels` sampler launches enqueued back to back (from Python each level costs ~10 us of host time, more
 * than the kernels of the three coarse levels).  features / best_bboxes / outputs: HOST arrays of `levels`
 * device pointers; H, W, spatial_scales: host arrays; N, C, points common to the levels.  ws: one block of
 * r3det_fr_levels_workspace_bytes() bytes (or NULL), carved per level as by r3det_feature_refine_forward. */
size_t r3det_fr_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points);
int r3det_feature_refine_forward_levels(int levels, const float* const* features, const float* const* best_bboxes,
                                        int N, int C, const int* H, const int* W, const float* spatial_scales,
                                        int points, float* const* outputs, void* ws, size_t ws_bytes, void* stream);

/* feature_refine_cuda.backward(top_grad, best_bboxes, spatial_scale, points, bottom_grad)
 *                                                      fr/src/feature_refine_cuda.cpp:44-66
 * kernel feature_refine_kernel.cu:165-230.  Accumulates into bottom_grad (the reference caller
 * zero-fills it, fr/feature_refine_module.py:36).  With overwrite != 0 the library writes the
 * full gradient instead (no zero-fill needed, no read of bottom_grad). */
int r3det_feature_refine_backward(const float* top_grad, const float* best_bboxes, int N, int C,
                                  int H, int W, float spatial_scale, int points,
                                  float* bottom_grad, int overwrite, void* stream);

/* The same call with a caller-provided device workspace of r3det_fr_backward_workspace_bytes() bytes (0 = the
 * shape has no workspace path -- a plane of more than 32768 cells, or one that does not fit the 160 KB of LDS;
 * ws may then be NULL).  With it the scatter of the reference (five float atomics per element,
 * feature_refine_kernel.cu:165-230) runs as a GATHER: the boxes of the level are turned into the inverse tap
 * index (per cell the list of {source position, weight}, sorted, so the sum has ONE order and results are
 * reproducible run to run; lists of more than 48 entries on planes the sorted index form does not take keep
 * their arrival order), re-laid in slices of 64 cells (SELL-64); the gradient pass then stages whole planes in
 * LDS and every lane sums its own cell's list -- no atomics, no zero-fill, points 1 or 5.  Same values as the
 * reference up to the summation order, which the reference's atomics leave open as well.  overwrite == 0 adds to
 * bottom_grad like the reference. */
size_t r3det_fr_backward_workspace_bytes(int N, int H, int W, int points);
int r3det_feature_refine_backward_ws(const float* top_grad, const float* best_bboxes, int N, int C, int H, int W,
                                     float spatial_scale, int points, float* bottom_grad, int overwrite, void* ws,
                                     size_t ws_bytes, void* stream);

/* Split form of the call above: the index depends on the boxes only, so a training step builds it when the
 * forward pass has the boxes (feature_refine_module.py:18-26 saves them for the backward) and the backward proper
 * is the gather alone.  _index takes the channel count of the gradient it is for (C decides how many channels the
 * gather interleaves per staged cell, and the index entries carry offsets of that layout): R3DET_EINVAL when the
 * (shape, C) has no gather form (then use r3det_feature_refine_backward).  _indexed: same N, C, H, W, points and
 * workspace as the _index call. */
int r3det_feature_refine_backward_index(const float* best_bboxes, int N, int C, int H, int W, float spatial_scale,
                                        int points, void* ws, size_t ws_bytes, void* stream);
int r3det_feature_refine_backward_indexed(const float* top_grad, int N, int C, int H, int W, int points,
                                          float* bottom_grad, int overwrite, void* ws, size_t ws_bytes, void* stream);

/* The backward of ALL pyramid levels of one FeatureRefineModule pass (fr/feature_refine_module.py:108-127 runs the
 * sampler level by level; its autograd graph then calls feature_refine_cuda.backward once per level,
 * feature_refine_module.py:28-40): one call builds the indexes of all levels when the forward pass has the boxes,
 * one call runs all gathers when the gradients arrive.  best_bboxes / top_grads / bottom_grads: HOST arrays of
 * `levels` device pointers; H, W, spatial_scales: host arrays; N, C, points common to the levels.  ws: ONE block of
 * r3det_fr_backward_levels_workspace_bytes() bytes, the same block for both calls.  A level whose (shape, C) has no
 * gather form takes the scatter kernels from its boxes inside the second call. */
size_t r3det_fr_backward_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points);
int r3det_feature_refine_backward_index_levels(int levels, const float* const* best_bboxes, int N, int C, const int* H,
                                               const int* W, const float* spatial_scales, int points, void* ws,
                                               size_t ws_bytes, void* stream);
int r3det_feature_refine_backward_levels_indexed(int levels, const float* const* top_grads,
                                                 const float* const* best_bboxes, int N, int C, const int* H,
                                                 const int* W, const float* spatial_scales, int points,
                                                 float* const* bottom_grads, int overwrite, void* ws, size_t ws_bytes,
                                                 void* stream);

/* feature_refine_cuda.backward on channels_last memory: top_grad / bottom_grad are (N, H, W, C) contiguous
 * (torch.channels_last of the (N, C, H, W) tensors), C % 4 == 0, 16-byte aligned.  Same values as
 * r3det_feature_refine_backward up to the summation order (kernel feature_refine_kernel.cu:165-230, caller
 * fr/feature_refine_module.py:28-40), points 1 or 5, any H x W with W <= 4096.  The scatter of the reference (five
 * float atomics per element) is a gather here: _index turns the boxes of a level into the inverse tap index
 * (per cell the list of {source position, weight}, sorted, so the sum has ONE order: results are reproducible
 * run to run) in one launch; the gradient pass then reads the identity row + one row per entry and writes each
 * row once -- no atomics, no zero-fill.  overwrite == 0 adds to bottom_grad like the reference.  ws: device
 * workspace of r3det_fr_backward_nhwc_workspace_bytes() bytes (0 = shape not taken, R3DET_EINVAL from the calls).
 * _nhwc = _index + _indexed in one call; the split form lets a training step build the index when the forward
 * pass has the boxes. */
size_t r3det_fr_backward_nhwc_workspace_bytes(int N, int H, int W, int points);
int r3det_feature_refine_backward_nhwc(const float* top_grad, const float* best_bboxes, int N, int C, int H, int W,
                                       float spatial_scale, int points, float* bottom_grad, int overwrite, void* ws,
                                       size_t ws_bytes, void* stream);
int r3det_feature_refine_backward_nhwc_index(const float* best_bboxes, int N, int H, int W, float spatial_scale,
                                             int points, void* ws, size_t ws_bytes, void* stream);
int r3det_feature_refine_backward_nhwc_indexed(const float* top_grad, int N, int C, int H, int W, int points,
                                               float* bottom_grad, int overwrite, void* ws, size_t ws_bytes,
                                               void* stream);
/* All pyramid levels of a channels_last FeatureRefineModule pass (feature_refine_module.py:108-127 loops over the
 * levels) in one call each: _index_levels when the forward pass has the boxes -- the bands of all levels ONE grid when
 * every level takes the sorted index form, else level by level -- and _levels_indexed when the gradients arrive: the
 * levels of at most 4096 cells ONE grid, the others one launch each.  Results: bit for bit those of the per-level
 * calls.  Pointer / shape arrays are HOST arrays of `levels` entries; every level must have a workspace form
 * (r3det_fr_backward_nhwc_workspace_bytes != 0), else R3DET_EINVAL; ws: one block of
 * r3det_fr_backward_nhwc_levels_workspace_bytes() bytes, carved in level order. */
size_t r3det_fr_backward_nhwc_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points);
int r3det_feature_refine_backward_nhwc_index_levels(int levels, const float* const* best_bboxes, int N, const int* H,
                                                    const int* W, const float* spatial_scales, int points, void* ws,
                                                    size_t ws_bytes, void* stream);
int r3det_feature_refine_backward_nhwc_levels_indexed(int levels, const float* const* top_grads, int N, int C,
                                                      const int* H, const int* W, int points,
                                                      float* const* bottom_grads, int overwrite, void* ws,
                                                      size_t ws_bytes, void* stream);

/* Training steps (round 6): the backward's index as a by-product of the forward launch.  The reference's backward
 * (fr/src/feature_refine_kernel.cu:165-230) scatters with atomics and needs no index; this library's deterministic
 * gather does, and building it from the 20-byte box records pulls 327 KB per image through every index workgroup for
 * one float in five.  A TAP TABLE of a level is, per image, [y: H*W floats][x: H*W floats]: the clamped sample point
 * of every position (the clamps of bilinear_interpolate, feature_refine_kernel.cu:22-47, applied once; a sample
 * outside the map = row H + 1) -- r3det_fr_tap_table_bytes(N, H, W) = 8 bytes per position, any shape, points = 1.
 *   - the _tab forms of the channels_last forward calls WRITE tables[l] (non-NULL entries) from the sampler launches
 *     themselves: the wave that owns a position stores two floats; no extra launch, same outputs;
 *   - r3det_feature_refine_prepare's table (NCHW) IS a tap table of its level;
 *   - the _tab forms of the index calls READ tables[l] where non-NULL (a NULL entry: that level from its boxes): the
 *     scan is 4 contiguous bytes per source.  The table must come from the same boxes and spatial_scale; the index is
 *     byte for byte the one the box form builds.  best_bboxes stay required (a band beyond the sorted form's capacity
 *     falls back to them).
 * tables: HOST array of `levels` device pointers (16-byte aligned). */
size_t r3det_fr_tap_table_bytes(int N, int H, int W);
int r3det_feature_refine_forward_levels_nhwc_tab(int levels, const float* const* features,
                                                 const float* const* best_bboxes, int N, int C, const int* H,
                                                 const int* W, const float* spatial_scales, int points,
                                                 float* const* outputs, float* const* tables, void* stream);
int r3det_feature_refine_module_levels_nhwc_tab(int levels, const float* const* conv_a, const float* const* conv_b,
                                                const float* bias_a, const float* bias_b, const float* const* residual,
                                                const float* const* best_bboxes, int N, int C, const int* H, const int* W,
                                                const float* spatial_scales, int points, float* const* outputs,
                                                float* const* tables, void* stream);
int r3det_feature_refine_backward_index_levels_tab(int levels, const float* const* best_bboxes,
                                                   const float* const* tables, int N, int C, const int* H, const int* W,
                                                   const float* spatial_scales, int points, void* ws, size_t ws_bytes,
                                                   void* stream);
int r3det_feature_refine_backward_nhwc_index_levels_tab(int levels, const float* const* best_bboxes,
                                                        const float* const* tables, int N, const int* H, const int* W,
                                                        const float* spatial_scales, int points, void* ws,
                                                        size_t ws_bytes, void* stream);

/* Producer of the FR boxes: RRetinaHead.filter_bboxes
 * (models/dense_heads/rotate_retina_head.py:117-179) and, with num_anchors = 1 and
 * anchors_per_image = 1, RRetinaRefineHead.refine_bboxes (rotate_retina_refine_head.py:56-97),
 * both ending in delta2bbox_v1 (core/bbox/coder/delta_xywha_rbbox_coder.py:142-211, means 0,
 * stds 1, no max_shape).  cls_score (N, A*C, H, W) and bbox_pred (N, A*5, H, W) are read through
 * their element strides {n, c, h, w} (host arrays of 4), so NCHW and channels_last maps both work;
 * cls_score may be NULL when num_anchors == 1.  anchors: (H*W*A, 5), or (N, H*W*A, 5) with
 * anchors_per_image != 0.  Per position: best anchor = first argmax over anchors of the max class
 * logit; boxes_out (N, H*W, 5) = decode(anchor, deltas), the layout FR reads.
 * max_ratio = |log(wh_ratio_clip)|. */
int r3det_filter_bboxes(const float* cls_score, const long long* cls_strides, const float* bbox_pred,
                        const long long* pred_strides, const float* anchors, int anchors_per_image, int N,
                        int num_anchors, int num_classes, int H, int W, float max_ratio, float* boxes_out,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * Exported ops outside the shipped training / inference configs
 * ------------------------------------------------------------------------------------- */

/* polygon_geo_cpu.polygon_iou(a, b)                 polygon_geo/src/polygon_geo_cpu.cpp:272-287
 * (CPU-only in the reference).  a (na,8), b (nb,8) = 4 vertices each -> out (na,nb). */
int r3det_polygon_iou(const float* polys_a, int na, const float* polys_b, int nb, float* out, void* stream);

/* nms_rotated_ext.nms_poly(dets9, thr)              nms_rotated/src/nms_rotated_ext.cpp:38-51,
 * kernel + host scan nms_rotated/src/poly_nms_cuda.cu:142-261.  dets9 (n,9) = 8 coordinates +
 * score, in ORIGINAL order; order = indices by descending score (stable).  keep_out (int64,
 * capacity n) = original indices in score order, count_out (int32, device).  Suppression on
 * devPolyIoU > thr.  r3det_poly_iou_mat exposes that IoU as a matrix (rows of `stride` floats,
 * the first 8 are used) for tests. */
size_t r3det_poly_nms_workspace_bytes(int n);
int r3det_nms_poly(const float* dets9, const int64_t* order, int n, float thr, void* ws, size_t ws_bytes,
                   int64_t* keep_out, int32_t* count_out, void* stream);
int r3det_poly_iou_mat(const float* a, int na, int stride_a, const float* b, int nb, int stride_b, float* out,
                       void* stream);

/* convex_ext.convex_sort(pts, masks, circular)       convex/src/convex_ext.cpp:13-27, kernel + tensor
 * prologue convex/src/convex_cuda.cu:13-123.  pts (B,P,2) fp32, masks (B,P) bytes (0 / non-0) ->
 * index_out (B, P + (circular != 0)) int64, padded with -1.  Equal sort keys are visited in index
 * order (the reference leaves that to torch.argsort).  ws: B * P * 4 bytes. */
int r3det_convex_sort(const float* pts, const unsigned char* masks, int B, int P, int circular, void* ws,
                      size_t ws_bytes, int64_t* index_out, void* stream);

/* Convolution epilogue of the inference model around the hot path (not one of the reference's
 * extension ops): y = act(y + bias[c] (+ residual)) in place, one pass.  The reference's benchmark
 * folds BatchNorm into the convolutions (tools/analysis_tools/benchmark.py:88-89, mmcv fuse_conv_bn);
 * this replaces the bias add / ReLU / residual add launches that remain.  y holds outer x C x inner
 * elements (NCHW: outer = N, inner = H*W; channels_last: outer = N*H*W, inner = 1, C % 4 == 0);
 * residual has the same shape and layout or is NULL.  Returns R3DET_EINVAL for shapes it does not
 * take (the caller then uses its own elementwise ops). */
int r3det_bias_act(float* y, const float* bias, const float* residual, long long outer, int C, long long inner,
                   int relu, void* stream);

/* Kernel-selection switches for the tests of rarely taken paths (not part of the reference surface).  Each is read
 * once per library call.  What the PRODUCT library knows:
 *   ("fr_impl", 0 auto | 1 generic | 2 lds-plane | 10 cell)      NCHW sampler forward; backward: != 0 = scatter kernels
 *   ("fr_dbg", 0 auto | 8 wide regions | 9 4 x 4 tile pairs)     channels_last sampler forward form
 *   ("fr_walk", n)       strip height of the tile-pair launch order of the channels_last kernels (0 row-major, default 8)
 *   ("fr_profile", 0 | 1 every kernel | 2 first start and last stop only)      see r3det_fr_profile_read
 *   ("frb_impl", 0 auto | 1 general index form | 2 unpaired NHWC gather | 3 SELL rows by their own launch |
 *                5 SELL rows and CSR lists from one launch | 6 pyramid levels one launch each: indexes, coarse-level
 *                samplers and gathers)
 *   ("iou_impl", 0 auto | 1 one thread per pair | 2 one-launch tile kernel | 4 stream + drain always | 3 the fused
 *                assignment on its global pair queue of rounds 2-4 instead of the matrix path's tile queue; a plain matrix call reads 3 as 0),
 *   ("iou_small", columns from which the pipeline runs), ("iou_qcap", n: per-wave survivor capacity, small values
 *   force the dense-tile path), ("iou_dwgs", drain workgroups), ("nms_impl", 0 | 1 tiles | 2 one reducer workgroup | 4 the batched
 *   pipeline's reducer as a walk in score order, one wavefront per image and label group | 5 the candidate selection as
 *   the two launches of rounds 2-4 (count, then write) instead of one | 6 the batched pipeline's sorted-chunk form
 *   (ranks by search in sorted 1024-chunks, pair tests in x order with tiles dropped by their extents) whatever the pool's
 *   size | 7 never that form; by itself it runs beyond 10 240 candidates; same results),
 *   ("nms_qcap", n: entries per queue region, small values force the redo-tile path),
 *   ("clip_impl", 0 the straight-line v1 pair clip of the drains (round 5, csrc/r3_clip.h) | 1 the LDS-list form of
 *                 rounds 2-4: same results bit for bit, kept for the A/B in tools/clip_ab.sh),
 *   ("iou_dyn", 1 the IoU drain's wavefronts draw their blocks of 64 queue entries behind the first from atomic tickets
 *               when there are three or more per wavefront | 0 always the static stride; same results, tools/iou_dyn_ab.sh),
 *   ("iou_order", -1 every workgroup of the matrix stream kernel zeroes its tile before its tests | b in 0..30: those
 *                 with bit b of their linear index set, the others after their tests (default 8) | 31 all after; same
 *                 results, tools/iou_order_ab.sh).
 * Thread safety: the switches are process-wide relaxed atomics, each read once per operator call -- a call sees one
 * consistent value of every switch it reads, whichever thread sets them; a caller that needs "this call with that
 * setting" still has to order the two itself.
 * Launch variants that were measured and not shipped, and clock stamps inside kernels, exist only in the probes build
 * of the same sources (`make probes`: libr3det_hip_probes.so, -DR3_PROBES; tools/ load it through R3DET_HIP_LIB);
 * the product library reads their option values as 0. */
int r3det_set_option(const char* name, int value);

/* Measurement aid for bench.py (not part of the reference surface).  With option "fr_profile" = 1
 * the launches of the cell path of r3det_feature_refine_forward carry their own start / stop
 * events (up to 512 calls).  This call waits for them, writes one record of 5 floats per call
 * {N, H, table_kernel_us, cell_kernel_us, span_us = table start -> cell stop} (at most `capacity`
 * records), empties the ring and returns the number of records written.  A start/stop event pair
 * reads ~4 us longer than rocprofv3's duration of the same kernel, so span_us (one pair for both
 * kernels, their gap included) is the figure that agrees with the profiler's sum. */
int r3det_fr_profile_read(float* records, int capacity);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* R3DET_HIP_H_ */
