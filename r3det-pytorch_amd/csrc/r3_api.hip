// r3_api.hip -- extern "C" entry points declared in include/r3det_hip.h.
// Argument validation + dispatch only; kernels live in r3_iou.hip / r3_nms.hip / r3_fr.hip.
#include <vector>
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/r3det_hip.h"
#include "r3_kernels.h"

R3Option g_r3_iou_impl{0};
R3Option g_r3_iou_small{0};
R3Option g_r3_clip_impl{0};
R3Option g_r3_iou_qcap{0};
R3Option g_r3_iou_dwgs{0};
R3Option g_r3_iou_nfill{0};  // (iou_impl 5) fill workgroups of the one-launch drain: 0 = one per compute unit
R3Option g_r3_iou_order{8};  // (measured: 18.8-19.0 us against 19.3-19.5 with every workgroup early, three alternations on one box)
R3Option g_r3_iou_dyn{1};
R3Option g_r3_nms_impl{0};
R3Option g_r3_nms_qcap{0};

namespace {
__global__ __launch_bounds__(256) void r3_zero_kernel(unsigned* __restrict__ p, size_t words, size_t head, size_t quads) {
  // [0, head) single words up to the first 16-byte boundary, then `quads` uint4, then the tail
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (size_t k = i; k < quads; k += stride) q[k] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t k = i; k < head; k += stride) p[k] = 0u;
  for (size_t k = head + 4 * quads + i; k < words; k += stride) p[k] = 0u;
}
}  // namespace

int r3k_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  if (!p || (reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return -2;
  const size_t words = bytes / 4;
  size_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;
  if (head > words) head = words;
  const size_t quads = (words - head) / 4;
  size_t blocks = (quads + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(r3_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<unsigned*>(p), words, head, quads);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int r3_cu_count() {
  static int n_cu[R3_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= R3_MAX_DEVICES) return 256;
  if (n_cu[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cu[dev] = n;
  }
  return n_cu[dev];
}

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// The device of a call is the device of its stream (the reference's extensions take a CUDAGuard on their tensors'
// device, e.g. box_iou_rotated_cuda.cu:77, nms_rotated_cuda.cu:79): kernels are launched, attributes set and the
// compute-unit count read on THAT device, whatever device is current for the caller, and the caller's current device
// is restored on return.  The null stream belongs to the current device.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(void* stream) {
    if (!stream) return;
    hipDevice_t dev = 0;
    int cur = 0;
    if (hipStreamGetDevice(S(stream), &dev) != hipSuccess || hipGetDevice(&cur) != hipSuccess || dev == cur) return;
    if (hipSetDevice(dev) == hipSuccess) prev = cur;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
inline int rc(int k) {
  switch (k) {
    case 0: return R3DET_OK;
    case -1: return R3DET_EINVAL;
    case -2: return R3DET_ELAUNCH;
    case -3: return R3DET_EWS;
    default: return R3DET_EINVAL;
  }
}
inline bool bad_iou_args(const float* b1, int n1, const float* b2, int n2, const float* out) {
  if (n1 < 0 || n2 < 0) return true;
  if (n1 > 0 && n2 > 0 && (!b1 || !b2 || !out)) return true;
  return false;
}
}  // namespace

extern "C" {

int r3det_abi_version(void) { return 1; }

const char* r3det_error_string(int code) {
  switch (code) {
    case R3DET_OK: return "ok";
    case R3DET_EINVAL: return "invalid argument";
    case R3DET_ELAUNCH: return "HIP launch failure";
    case R3DET_EWS: return "workspace too small";
    default: return "unknown error";
  }
}

size_t r3det_iou_workspace_bytes(int n1, int n2) { return r3k_iou_workspace_bytes(n1, n2); }

int r3det_rbbox_geo_mat_iou_iof(const float* rb1, int n1, const float* rb2, int n2, int iof,
                                float* out, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(rb1, n1, rb2, n2, out)) return R3DET_EINVAL;
  return rc(r3k_iou_mat(R3DET_GEOM_V1, iof != 0, rb1, n1, rb2, n2, out, ws, ws_bytes, S(stream)));
}

int r3det_rbbox_geo_vec_iou_iof(const float* rb1, int n1, const float* rb2, int n2, int iof,
                                float* out, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(rb1, n1, rb2, n2, out)) return R3DET_EINVAL;
  return rc(r3k_iou_vec(R3DET_GEOM_V1, iof != 0, rb1, n1, rb2, n2, out, S(stream)));
}

int r3det_box_iou_rotated_overlaps(const float* b1, int n1, const float* b2, int n2,
                                   int iou_or_iof, float* out, void* ws, size_t ws_bytes,
                                   void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(b1, n1, b2, n2, out)) return R3DET_EINVAL;
  return rc(r3k_iou_mat(R3DET_GEOM_V3, iou_or_iof == 0, b1, n1, b2, n2, out, ws, ws_bytes, S(stream)));
}

int r3det_obb_overlaps(const float* b1, int n1, const float* b2, int n2, int iou_or_iof, float* out, void* ws,
                       size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(b1, n1, b2, n2, out)) return R3DET_EINVAL;
  // (round 6: the stream + drain pipeline applies the thin-box rule itself -- a thin row is skipped, a thin column never
  // survives -- and the epilogue launch is only needed behind the one-launch forms of narrow matrices)
  int thin_done = 0;
  const int r = r3k_iou_mat(R3DET_GEOM_V3, iou_or_iof == 0, b1, n1, b2, n2, out, ws, ws_bytes, S(stream), nullptr, &thin_done);
  if (r) return rc(r);
  if (thin_done) return R3DET_OK;
  return rc(r3k_iou_zero_thin(b1, n1, b2, n2, out, S(stream)));
}

int r3det_box_iou_rotated_overlaps_aligned(const float* b1, const float* b2, int n,
                                           int iou_or_iof, float* out, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(b1, n, b2, n, out)) return R3DET_EINVAL;
  return rc(r3k_iou_vec(R3DET_GEOM_V3, iou_or_iof == 0, b1, n, b2, n, out, S(stream)));
}

int r3det_mmcv_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int mode_flag,
                               int aligned, float* out, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(b1, n1, b2, n2, out)) return R3DET_EINVAL;
  if (mode_flag != 0 && mode_flag != 1) return R3DET_EINVAL;
  if (aligned) {
    if (n1 != n2) return R3DET_EINVAL;
    return rc(r3k_iou_vec(R3DET_GEOM_V2, mode_flag, b1, n1, b2, n2, out, S(stream)));
  }
  return rc(r3k_iou_mat(R3DET_GEOM_V2, mode_flag, b1, n1, b2, n2, out, ws, ws_bytes, S(stream)));
}

size_t r3det_iou_prepared_bytes(int n) { return r3k_iou_prepared_bytes(n); }

// What a prepared buffer was built for (the buffer is opaque device memory, a mismatched one silently gave
// wrong IoUs).  Round 6: the buffer carries a 16-byte header {magic, geometry, n, check} written by the prepare kernel;
// every workgroup of the consumers' stream kernel compares it with its own launch before it reads anything else and, on
// a mismatch, answers in the DATA -- a NaN matrix, NaN max_overlaps with every anchor ignored -- instead of using the
// buffer.  No host-side state (rounds 4-5: a process-global map keyed by the device address, fooled by a freed and
// reused address).  r3det_iou_prepared_check is the host-side answer for callers who want one: it copies
// the header back (a stream synchronisation: once after preparing, not per call).
int r3det_iou_prepare_columns(int geom, const float* boxes, int n, void* prepared, size_t prepared_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (geom < 1 || geom > 3) return R3DET_EINVAL;
  return rc(r3k_iou_prepare_columns(geom, boxes, n, prepared, prepared_bytes, S(stream)));
}

int r3det_iou_prepared_check(const void* prepared, int geom, int n, void* stream) {
  const DeviceGuard guard(stream);
  if (!prepared || geom < 1 || geom > 3 || n <= 0) return R3DET_EINVAL;
  int h[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(h, prepared, sizeof(h), hipMemcpyDeviceToHost, S(stream)) != hipSuccess) return R3DET_ELAUNCH;
  if (hipStreamSynchronize(S(stream)) != hipSuccess) return R3DET_ELAUNCH;
  const int magic = 0x52335043;  // "R3PC" (csrc/r3_iou.hip: COLPREP_MAGIC)
  return (h[0] == magic && h[1] == geom && h[2] == n && h[3] == (magic ^ geom ^ n)) ? R3DET_OK : R3DET_EINVAL;
}

int r3det_iou_mat_prepared(int geom, const float* b1, int n1, const float* b2, int n2, const void* prepared, int mode,
                           float* out, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (bad_iou_args(b1, n1, b2, n2, out) || geom < 1 || geom > 3 || (n2 > 0 && !prepared)) return R3DET_EINVAL;
  // (mode: as the reference entry of the geometry takes it -- v1 / v2: 1 = iof; v3: 0 = iof)
  const int iof = geom == 3 ? (mode == 0) : (mode != 0);
  return rc(r3k_iou_mat(geom, iof, b1, n1, b2, n2, out, ws, ws_bytes, S(stream), prepared));
}

int r3det_rbbox_assign_prepared(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes,
                                const void* prepared, float pos_iou_thr, float neg_iou_thr, float min_pos_iou,
                                int match_low_quality, int gt_max_assign_all, int64_t* assigned_gt_inds,
                                float* max_overlaps, int64_t* argmax_overlaps, float* gt_max_overlaps,
                                int64_t* gt_argmax_overlaps, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (n_boxes > 0 && !prepared) return R3DET_EINVAL;
  return rc(r3k_iou_assign(geom, gts, n_gt, boxes, n_boxes, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality,
                           gt_max_assign_all, assigned_gt_inds, max_overlaps, argmax_overlaps, gt_max_overlaps,
                           gt_argmax_overlaps, ws, ws_bytes, S(stream), prepared));
}

size_t r3det_rbbox_assign_workspace_bytes(int n_gt, int n_boxes) {
  return r3k_iou_assign_workspace_bytes(n_gt, n_boxes);
}

int r3det_rbbox_assign(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes, float pos_iou_thr,
                       float neg_iou_thr, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                       int64_t* assigned_gt_inds, float* max_overlaps, int64_t* argmax_overlaps,
                       float* gt_max_overlaps, int64_t* gt_argmax_overlaps, void* ws, size_t ws_bytes,
                       void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_iou_assign(geom, gts, n_gt, boxes, n_boxes, pos_iou_thr, neg_iou_thr, min_pos_iou,
                           match_low_quality, gt_max_assign_all, assigned_gt_inds, max_overlaps, argmax_overlaps,
                           gt_max_overlaps, gt_argmax_overlaps, ws, ws_bytes, S(stream)));
}

int r3det_rbbox_assign_labeled(int geom, const float* gts, int n_gt, const float* boxes, int n_boxes, const void* prepared,
                               float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                               int gt_max_assign_all, int64_t* assigned_gt_inds, float* max_overlaps,
                               int64_t* argmax_overlaps, float* gt_max_overlaps, int64_t* gt_argmax_overlaps,
                               const int64_t* gt_labels, int64_t* assigned_labels, void* ws, size_t ws_bytes,
                               void* stream) {
  const DeviceGuard guard(stream);
  if ((gt_labels == nullptr) != (assigned_labels == nullptr)) return R3DET_EINVAL;
  return rc(r3k_iou_assign(geom, gts, n_gt, boxes, n_boxes, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality,
                           gt_max_assign_all, assigned_gt_inds, max_overlaps, argmax_overlaps, gt_max_overlaps,
                           gt_argmax_overlaps, ws, ws_bytes, S(stream), prepared, gt_labels, assigned_labels));
}

size_t r3det_nms_workspace_bytes(int n) { return r3k_nms_workspace_bytes(n); }

int r3det_rnms(const float* dets6, const int64_t* order, int n, float thr, int sort_ascending,
               void* ws, size_t ws_bytes, int64_t* keep_out, int32_t* count_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_nms(R3DET_GEOM_V1, dets6, 6, nullptr, order, n, thr, sort_ascending, ws, ws_bytes,
                    keep_out, count_out, S(stream)));
}

int r3det_nms_rotated(const float* dets5, const int64_t* order, int n, float thr, void* ws,
                      size_t ws_bytes, int64_t* keep_out, int32_t* count_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_nms(R3DET_GEOM_V3, dets5, 5, nullptr, order, n, thr, 0, ws, ws_bytes, keep_out,
                    count_out, S(stream)));
}

int r3det_ml_nms_rotated(const float* dets5, const int64_t* labels, const int64_t* order, int n,
                         float thr, void* ws, size_t ws_bytes, int64_t* keep_out,
                         int32_t* count_out, void* stream) {
  const DeviceGuard guard(stream);
  if (n > 0 && !labels) return R3DET_EINVAL;
  return rc(r3k_nms(R3DET_GEOM_V2, dets5, 5, labels, order, n, thr, 0, ws, ws_bytes, keep_out,
                    count_out, S(stream)));
}

int r3det_mmcv_nms_rotated(const float* dets5, const int64_t* labels, const int64_t* order, int n,
                           float thr, void* ws, size_t ws_bytes, int64_t* keep_out,
                           int32_t* count_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_nms(R3DET_GEOM_V2, dets5, 5, labels, order, n, thr, 0, ws, ws_bytes, keep_out,
                    count_out, S(stream)));
}

size_t r3det_mcnms_select_workspace_bytes(int B, int n) { return r3k_mcnms_select_workspace_bytes(B, n); }

int r3det_mcnms_select(const float* boxes, const float* scores, int B, int n, int K, float score_thr,
                       int32_t* cand_row, int32_t* cand_label, float* cand_score, int32_t* cand_rank,
                       int32_t* counts, float* maxc, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_mcnms_select(boxes, scores, B, n, K, score_thr, cand_row, cand_label, cand_score, cand_rank, counts,
                             maxc, ws, ws_bytes, S(stream)));
}

size_t r3det_mcnms_workspace_bytes(int B, int cap) { return r3k_mcnms_workspace_bytes(B, cap); }

int r3det_mcnms_v1(const float* boxes, int B, int n, int K, const int32_t* cand_row,
                   const int32_t* cand_label, const float* cand_score, int32_t* cand_rank,
                   const int32_t* counts, const float* maxc, int cap, float iou_thr, int out_cap, void* ws,
                   size_t ws_bytes, float* dets_out, int64_t* labels_out, int64_t* keep_idx_out,
                   int32_t* counts_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_mcnms_v1(boxes, B, n, K, cand_row, cand_label, cand_score, cand_rank, counts, maxc, cap, iou_thr,
                         out_cap, ws, ws_bytes, dets_out, labels_out, keep_idx_out, counts_out, S(stream)));
}

int r3det_mcnms(int nms_type, const float* boxes, int B, int n, int K, const int32_t* cand_row,
                const int32_t* cand_label, const float* cand_score, int32_t* cand_rank, const int32_t* counts,
                const float* maxc, int cap, float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out,
                int64_t* labels_out, int64_t* keep_idx_out, int32_t* counts_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_mcnms_run(nms_type, boxes, B, n, K, cand_row, cand_label, cand_score, cand_rank, counts, maxc, cap,
                          iou_thr, out_cap, ws, ws_bytes, dets_out, labels_out, keep_idx_out, counts_out, S(stream)));
}

int r3det_mcnms_padded(int nms_type, const float* boxes, int B, int n, int K, const int32_t* cand_row,
                       const int32_t* cand_label, const float* cand_score, int32_t* cand_rank, const int32_t* counts,
                       const float* maxc, int cap, float iou_thr, int out_cap, void* ws, size_t ws_bytes,
                       float* dets7_out, size_t img_stride, int32_t* counts_out, float* count_f32_out,
                       size_t count_f32_stride, int32_t* overflow_out, void* stream) {
  const DeviceGuard guard(stream);
  const R3kMcPadded pd{img_stride, count_f32_out, count_f32_stride, overflow_out};
  return rc(r3k_mcnms_run(nms_type, boxes, B, n, K, cand_row, cand_label, cand_score, cand_rank, counts, maxc, cap,
                          iou_thr, out_cap, ws, ws_bytes, dets7_out, nullptr, nullptr, counts_out, S(stream), &pd));
}

size_t r3det_batched_rnms_workspace_bytes(int n) { return r3k_batched_rnms_workspace_bytes(n); }

int r3det_batched_rnms(const float* bboxes, const float* scores, const int64_t* inds, int n, float nms_thr, void* ws,
                       size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, void* stream) {
  const DeviceGuard guard(stream);
  if (n == 0) return R3DET_OK;
  return rc(r3k_batched_nms(1, bboxes, scores, inds, n, nms_thr, ws, ws_bytes, dets_out, keep_out, kept_out, S(stream)));
}

int r3det_obb_batched_nms(const float* bboxes, const float* scores, const int64_t* inds, int n, float nms_thr, void* ws,
                          size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, void* stream) {
  const DeviceGuard guard(stream);
  if (n == 0) return R3DET_OK;
  return rc(r3k_batched_nms(3, bboxes, scores, inds, n, nms_thr, ws, ws_bytes, dets_out, keep_out, kept_out, S(stream)));
}

size_t r3det_fr_workspace_bytes(int N, int H, int W, int points) {
  return r3k_fr_workspace_bytes(N, H, W, points);
}

int r3det_feature_refine_forward(const float* features, const float* best_bboxes, int N, int C,
                                 int H, int W, float spatial_scale, int points, float* output,
                                 void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N < 0 || C < 0 || H < 0 || W < 0) return R3DET_EINVAL;
  if ((size_t)N * C * H * W > 0 && (!features || !best_bboxes || !output)) return R3DET_EINVAL;
  return rc(r3k_fr_forward(features, best_bboxes, N, C, H, W, spatial_scale, points, output, ws,
                           ws_bytes, S(stream)));
}

size_t r3det_fr_table_bytes(int N, int H, int W) { return r3k_fr_table_bytes(N, H, W); }

int r3det_feature_refine_prepare(const float* best_bboxes, int N, int H, int W, float spatial_scale, float* table,
                                 void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_fr_prepare(best_bboxes, N, H, W, spatial_scale, table, S(stream)));
}

int r3det_feature_refine_forward_prepared(const float* features, const float* table, int N, int C, int H, int W,
                                          float* output, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_fr_forward_prepared(features, nullptr, nullptr, table, N, C, H, W, output, S(stream)));
}

int r3det_frm_mix_nchw(const float* a_nhwc, const float* b_nhwc, const float* bias_a, const float* bias_b, int N,
                       int C, int H, int W, float* out_nchw, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_mix_to_nchw(a_nhwc, b_nhwc, bias_a, bias_b, N, C, H, W, out_nchw, S(stream)));
}

int r3det_feature_refine_module_prepared(const float* mixed_a, const float* mixed_b, const float* residual,
                                         const float* table, int N, int C, int H, int W, float* output,
                                         void* stream) {
  const DeviceGuard guard(stream);
  if (!mixed_a || !residual) return R3DET_EINVAL;
  return rc(r3k_fr_forward_prepared(mixed_a, mixed_b, residual, table, N, C, H, W, output, S(stream)));
}

size_t r3det_level_pool_workspace_bytes(int N, int A, int H, int W, int nms_pre) {
  return r3k_level_pool_workspace_bytes(N, A, H, W, nms_pre);
}

int r3det_level_pool(const float* cls_score, const long long* cls_strides, const float* bbox_pred,
                     const long long* reg_strides, const float* anchors, int anchors_per_image, int N, int A, int C,
                     int H, int W, int nms_pre, float max_ratio, float max_x, float max_y, float* pool_boxes,
                     float* pool_scores, int pool_rows, int row_offset, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_level_pool(cls_score, cls_strides, bbox_pred, reg_strides, anchors, anchors_per_image, N, A, C, H, W,
                           nms_pre, max_ratio, max_x, max_y, pool_boxes, pool_scores, pool_rows, row_offset, ws, ws_bytes,
                           S(stream)));
}

size_t r3det_levels_pool_workspace_bytes(int num_levels, int N, const int* A, const int* H, const int* W, int nms_pre) {
  return r3k_levels_pool_workspace_bytes(num_levels, N, A, H, W, nms_pre);
}

int r3det_levels_pool(int num_levels, const float* const* cls_scores, const long long* cls_strides,
                      const float* const* bbox_preds, const long long* reg_strides, const float* const* anchors,
                      int anchors_per_image, int N, const int* A, int C, const int* H, const int* W, int nms_pre,
                      float max_ratio, float max_x, float max_y, float* pool_boxes, float* pool_scores, int pool_rows,
                      void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_levels_pool(num_levels, cls_scores, cls_strides, bbox_preds, reg_strides, anchors, anchors_per_image, N, A,
                            C, H, W, nms_pre, max_ratio, max_x, max_y, pool_boxes, pool_scores, pool_rows, 0, ws, ws_bytes,
                            S(stream)));
}

int r3det_feature_refine_forward_nhwc(const float* features, const float* best_bboxes, int N, int C, int H, int W,
                                      float spatial_scale, int points, float* output, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_fr_forward_nhwc(features, nullptr, nullptr, nullptr, nullptr, best_bboxes, N, C, H, W, spatial_scale,
                                points, output, S(stream)));
}

int r3det_feature_refine_module_nhwc(const float* conv_a, const float* conv_b, const float* bias_a,
                                     const float* bias_b, const float* residual, const float* best_bboxes, int N,
                                     int C, int H, int W, float spatial_scale, int points, float* output,
                                     void* stream) {
  const DeviceGuard guard(stream);
  if (!conv_a || !residual) return R3DET_EINVAL;
  return rc(r3k_fr_forward_nhwc(conv_a, conv_b, bias_a, bias_b, residual, best_bboxes, N, C, H, W, spatial_scale,
                                points, output, S(stream)));
}

namespace {
// (tables: null, or per level a device pointer / null -- the tap tables the launches also write, points = 1)
int fr_levels_nhwc(int levels, const float* const* a, const float* const* b, const float* bias_a, const float* bias_b,
                   const float* const* res, const float* const* best_bboxes, int N, int C, const int* H, const int* W,
                   const float* spatial_scales, int points, float* const* outputs, float* const* tables, void* stream) {
  if (tables && points != 1) return R3DET_EINVAL;
  return rc(r3k_fr_forward_nhwc_levels(levels, a, b, bias_a, bias_b, res, best_bboxes, N, C, H, W, spatial_scales, points,
                                       outputs, S(stream), tables));
}
}  // namespace

int r3det_feature_refine_forward_levels_nhwc(int levels, const float* const* features, const float* const* best_bboxes,
                                             int N, int C, const int* H, const int* W, const float* spatial_scales,
                                             int points, float* const* outputs, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || (points != 1 && points != 5)) return R3DET_EINVAL;
  return fr_levels_nhwc(levels, features, nullptr, nullptr, nullptr, nullptr, best_bboxes, N, C, H, W, spatial_scales,
                        points, outputs, nullptr, stream);
}

int r3det_feature_refine_forward_levels_nhwc_tab(int levels, const float* const* features,
                                                 const float* const* best_bboxes, int N, int C, const int* H,
                                                 const int* W, const float* spatial_scales, int points,
                                                 float* const* outputs, float* const* tables, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || points != 1 || !tables) return R3DET_EINVAL;
  return fr_levels_nhwc(levels, features, nullptr, nullptr, nullptr, nullptr, best_bboxes, N, C, H, W, spatial_scales,
                        points, outputs, tables, stream);
}

int r3det_feature_refine_module_levels_nhwc(int levels, const float* const* conv_a, const float* const* conv_b,
                                            const float* bias_a, const float* bias_b, const float* const* residual,
                                            const float* const* best_bboxes, int N, int C, const int* H, const int* W,
                                            const float* spatial_scales, int points, float* const* outputs, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || (points != 1 && points != 5) || (levels > 0 && (!conv_a || !residual)))
    return R3DET_EINVAL;
  for (int l = 0; l < levels; l++)
    if (!conv_a[l] || !residual[l] || (conv_b && conv_b[0] && !conv_b[l])) return R3DET_EINVAL;
  return fr_levels_nhwc(levels, conv_a, conv_b, bias_a, bias_b, residual, best_bboxes, N, C, H, W, spatial_scales, points,
                        outputs, nullptr, stream);
}

int r3det_feature_refine_module_levels_nhwc_tab(int levels, const float* const* conv_a, const float* const* conv_b,
                                                const float* bias_a, const float* bias_b, const float* const* residual,
                                                const float* const* best_bboxes, int N, int C, const int* H, const int* W,
                                                const float* spatial_scales, int points, float* const* outputs,
                                                float* const* tables, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || points != 1 || !tables || (levels > 0 && (!conv_a || !residual)))
    return R3DET_EINVAL;
  for (int l = 0; l < levels; l++)
    if (!conv_a[l] || !residual[l] || (conv_b && conv_b[0] && !conv_b[l])) return R3DET_EINVAL;
  return fr_levels_nhwc(levels, conv_a, conv_b, bias_a, bias_b, residual, best_bboxes, N, C, H, W, spatial_scales, points,
                        outputs, tables, stream);
}

size_t r3det_fr_tap_table_bytes(int N, int H, int W) { return r3k_fr_tap_table_bytes(N, H, W); }

size_t r3det_fr_module_levels_workspace_bytes(int levels, int N, const int* H, const int* W) {
  if (levels < 0 || !H || !W) return 0;
  size_t total = 0;
  for (int l = 0; l < levels; l++) total += (r3k_fr_table_bytes(N, H[l], W[l]) + 255) & ~(size_t)255;
  return total;
}

int r3det_feature_refine_module_levels(int levels, const float* const* conv_a, const float* const* conv_b,
                                       const float* const* residual, const float* const* best_bboxes, int N, int C,
                                       const int* H, const int* W, const float* spatial_scales, int points,
                                       float* const* outputs, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 1 || levels > 8 || N <= 0 || C <= 0 || points != 1 || !conv_a || !conv_b || !residual || !best_bboxes || !H ||
      !W || !spatial_scales || !outputs)
    return R3DET_EINVAL;
  const size_t need = r3det_fr_module_levels_workspace_bytes(levels, N, H, W);
  if (ws_bytes < need || (need && !ws)) return R3DET_EWS;
  float* tables[8];
  char* p = static_cast<char*>(ws);
  for (int l = 0; l < levels; l++) {
    const size_t part = (r3k_fr_table_bytes(N, H[l], W[l]) + 255) & ~(size_t)255;
    tables[l] = part ? reinterpret_cast<float*>(p) : nullptr;
    p += part;
  }
  return rc(r3k_fr_module_levels(levels, conv_a, conv_b, residual, best_bboxes, N, C, H, W, spatial_scales, outputs, tables,
                                 S(stream)));
}

size_t r3det_fr_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points) {
  size_t total = 0;
  for (int l = 0; l < levels; l++) total += r3k_fr_workspace_bytes(N, H[l], W[l], points);
  return total;
}

int r3det_feature_refine_forward_levels(int levels, const float* const* features, const float* const* best_bboxes,
                                        int N, int C, const int* H, const int* W, const float* spatial_scales,
                                        int points, float* const* outputs, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N < 0 || C < 0 || (levels > 0 && (!features || !best_bboxes || !H || !W || !spatial_scales || !outputs)))
    return R3DET_EINVAL;
  if (ws && ws_bytes < r3det_fr_levels_workspace_bytes(levels, N, H, W, points)) return R3DET_EWS;
  for (int l = 0; l < levels; l++) {
    if (H[l] < 0 || W[l] < 0 || (points != 1 && points != 5)) return R3DET_EINVAL;
    if ((size_t)N * C * H[l] * W[l] > 0 && (!features[l] || !best_bboxes[l] || !outputs[l])) return R3DET_EINVAL;
  }
  return rc(r3k_fr_forward_levels(levels, features, best_bboxes, N, C, H, W, spatial_scales, points, outputs, ws, ws_bytes,
                                  S(stream)));
}

int r3det_feature_refine_backward(const float* top_grad, const float* best_bboxes, int N, int C,
                                  int H, int W, float spatial_scale, int points,
                                  float* bottom_grad, int overwrite, void* stream) {
  const DeviceGuard guard(stream);
  if (N < 0 || C < 0 || H < 0 || W < 0) return R3DET_EINVAL;
  if ((size_t)N * C * H * W > 0 && (!top_grad || !best_bboxes || !bottom_grad)) return R3DET_EINVAL;
  return rc(r3k_fr_backward(top_grad, best_bboxes, N, C, H, W, spatial_scale, points, bottom_grad,
                            overwrite, nullptr, 0, 0, S(stream)));
}

size_t r3det_fr_backward_workspace_bytes(int N, int H, int W, int points) {
  return r3k_fr_backward_workspace_bytes(N, H, W, points);
}

int r3det_feature_refine_backward_ws(const float* top_grad, const float* best_bboxes, int N, int C, int H, int W,
                                     float spatial_scale, int points, float* bottom_grad, int overwrite, void* ws,
                                     size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N < 0 || C < 0 || H < 0 || W < 0) return R3DET_EINVAL;
  if ((size_t)N * C * H * W > 0 && (!top_grad || !best_bboxes || !bottom_grad)) return R3DET_EINVAL;
  return rc(r3k_fr_backward(top_grad, best_bboxes, N, C, H, W, spatial_scale, points, bottom_grad,
                            overwrite, ws, ws_bytes, 0, S(stream)));
}

int r3det_feature_refine_backward_index(const float* best_bboxes, int N, int C, int H, int W, float spatial_scale,
                                        int points, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (points != 1 && points != 5)) return R3DET_EINVAL;
  return rc(r3k_frn_index(best_bboxes, N, C, H, W, spatial_scale, points, ws, ws_bytes, S(stream)));
}

int r3det_feature_refine_backward_indexed(const float* top_grad, int N, int C, int H, int W, int points,
                                          float* bottom_grad, int overwrite, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (points != 1 && points != 5) || !top_grad || !bottom_grad || !ws)
    return R3DET_EINVAL;
  return rc(r3k_frn_gather(top_grad, N, C, H, W, points, bottom_grad, overwrite, ws, ws_bytes, S(stream)));
}

// ---- all pyramid levels of one FeatureRefineModule pass in one call each (index when the forward has the boxes,
//      gather when the gradient arrives): the per-level workspaces are one block, carved in level order
namespace {
inline size_t lvl_bytes(int N, int H, int W, int points) {
  return (r3k_fr_backward_workspace_bytes(N, H, W, points) + 255) & ~(size_t)255;
}
}  // namespace

size_t r3det_fr_backward_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points) {
  if (levels < 0 || !H || !W) return 0;
  size_t total = 0;
  for (int l = 0; l < levels; l++) total += lvl_bytes(N, H[l], W[l], points);
  return total;
}

namespace {
int frn_index_levels(int levels, const float* const* best_bboxes, const float* const* tables, int N, int C, const int* H,
                     const int* W, const float* spatial_scales, int points, void* ws, size_t ws_bytes, void* stream) {
  if (levels < 0 || N <= 0 || C <= 0 || (points != 1 && points != 5) ||
      (levels > 0 && (!best_bboxes || !H || !W || !spatial_scales)) || (tables && points != 1))
    return R3DET_EINVAL;
  if (ws_bytes < r3det_fr_backward_levels_workspace_bytes(levels, N, H, W, points)) return R3DET_EWS;
  char* p = static_cast<char*>(ws);
  if (levels >= 1 && levels <= 8) {  // the bands of all levels as one grid, when every level takes that form
    void* wl[8];
    size_t wb[8];
    char* q = p;
    for (int l = 0; l < levels; l++) {
      wl[l] = q, wb[l] = lvl_bytes(N, H[l], W[l], points);
      q += wb[l];
    }
    const int k = r3k_frn_index_levels(levels, best_bboxes, N, C, H, W, spatial_scales, points, wl, wb, S(stream), tables);
    if (k <= 0) return rc(k);
  }
  for (int l = 0; l < levels; l++) {
    const size_t part = lvl_bytes(N, H[l], W[l], points);
    if (part) {  // (a level without a gather form has no index: its gradient pass reads the boxes)
      const int k = r3k_frn_index(best_bboxes[l], N, C, H[l], W[l], spatial_scales[l], points, p, part, S(stream),
                                  tables ? tables[l] : nullptr);
      if (k != 0 && k != -1) return rc(k);
    }
    p += part;
  }
  return R3DET_OK;
}
}  // namespace

int r3det_feature_refine_backward_index_levels(int levels, const float* const* best_bboxes, int N, int C, const int* H,
                                               const int* W, const float* spatial_scales, int points, void* ws,
                                               size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  return frn_index_levels(levels, best_bboxes, nullptr, N, C, H, W, spatial_scales, points, ws, ws_bytes, stream);
}

int r3det_feature_refine_backward_index_levels_tab(int levels, const float* const* best_bboxes,
                                                   const float* const* tables, int N, int C, const int* H, const int* W,
                                                   const float* spatial_scales, int points, void* ws, size_t ws_bytes,
                                                   void* stream) {
  const DeviceGuard guard(stream);
  if (!tables) return R3DET_EINVAL;
  return frn_index_levels(levels, best_bboxes, tables, N, C, H, W, spatial_scales, points, ws, ws_bytes, stream);
}

int r3det_feature_refine_backward_levels_indexed(int levels, const float* const* top_grads,
                                                 const float* const* best_bboxes, int N, int C, const int* H,
                                                 const int* W, const float* spatial_scales, int points,
                                                 float* const* bottom_grads, int overwrite, void* ws, size_t ws_bytes,
                                                 void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || (points != 1 && points != 5) ||
      (levels > 0 && (!top_grads || !best_bboxes || !H || !W || !spatial_scales || !bottom_grads)))
    return R3DET_EINVAL;
  if (ws_bytes < r3det_fr_backward_levels_workspace_bytes(levels, N, H, W, points)) return R3DET_EWS;
  std::vector<void*> parts(levels);
  std::vector<size_t> part_bytes(levels);
  std::vector<int> taken(levels);
  char* p = static_cast<char*>(ws);
  for (int l = 0; l < levels; l++) {
    part_bytes[l] = lvl_bytes(N, H[l], W[l], points);
    parts[l] = part_bytes[l] ? p : nullptr;
    p += part_bytes[l];
  }
  const int k = r3k_frn_gather_levels(levels, top_grads, N, C, H, W, points, bottom_grads, overwrite, parts.data(), part_bytes.data(),
                                      taken.data(),                                      S(stream));
  if (k) return rc(k);
  for (int l = 0; l < levels; l++) {
    if (taken[l]) continue;  // no gather form for this (shape, C): the scatter kernels, from the boxes
    const int k2 = r3k_fr_backward(top_grads[l], best_bboxes[l], N, C, H[l], W[l], spatial_scales[l], points, bottom_grads[l],
                                   overwrite, nullptr, 0, 0, S(stream));
    if (k2) return rc(k2);
  }
  return R3DET_OK;
}

size_t r3det_fr_backward_nhwc_workspace_bytes(int N, int H, int W, int points) {
  return r3k_frb_workspace_bytes(N, H, W, points);
}

int r3det_feature_refine_backward_nhwc(const float* top_grad, const float* best_bboxes, int N, int C, int H, int W,
                                       float spatial_scale, int points, float* bottom_grad, int overwrite, void* ws,
                                       size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N < 0 || C < 0 || H < 0 || W < 0 || (points != 1 && points != 5)) return R3DET_EINVAL;
  if ((size_t)N * C * H * W == 0) return R3DET_OK;
  if (!top_grad || !best_bboxes || !bottom_grad || !ws) return R3DET_EINVAL;
  return rc(r3k_frb_backward(top_grad, best_bboxes, N, C, H, W, spatial_scale, points, bottom_grad, overwrite, ws,
                             ws_bytes, 0, S(stream)));
}

int r3det_feature_refine_backward_nhwc_index(const float* best_bboxes, int N, int H, int W, float spatial_scale,
                                             int points, void* ws, size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (N <= 0 || H <= 0 || W <= 0) return R3DET_EINVAL;
  return rc(r3k_frb_index(best_bboxes, N, H, W, spatial_scale, points, ws, ws_bytes, S(stream)));
}

int r3det_feature_refine_backward_nhwc_indexed(const float* top_grad, int N, int C, int H, int W, int points,
                                               float* bottom_grad, int overwrite, void* ws, size_t ws_bytes,
                                               void* stream) {
  const DeviceGuard guard(stream);
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || !top_grad || !bottom_grad || !ws) return R3DET_EINVAL;
  return rc(r3k_frb_backward(top_grad, nullptr, N, C, H, W, 0.f, points, bottom_grad, overwrite, ws, ws_bytes, 1,
                             S(stream)));
}

namespace {
inline size_t nhwc_lvl_bytes(int N, int H, int W, int points) {
  return (r3k_frb_workspace_bytes(N, H, W, points) + 255) & ~(size_t)255;
}
}  // namespace

size_t r3det_fr_backward_nhwc_levels_workspace_bytes(int levels, int N, const int* H, const int* W, int points) {
  if (levels < 0 || !H || !W) return 0;
  size_t total = 0;
  for (int l = 0; l < levels; l++) {
    const size_t part = nhwc_lvl_bytes(N, H[l], W[l], points);
    if (!part) return 0;  // (a level without a workspace form: no levels call)
    total += part;
  }
  return total;
}

namespace {
int frb_index_levels(int levels, const float* const* best_bboxes, const float* const* tables, int N, const int* H,
                     const int* W, const float* spatial_scales, int points, void* ws, size_t ws_bytes, void* stream) {
  if (levels < 0 || N <= 0 || (points != 1 && points != 5) || (levels > 0 && (!best_bboxes || !H || !W || !spatial_scales)) ||
      (tables && points != 1))
    return R3DET_EINVAL;
  const size_t need = r3det_fr_backward_nhwc_levels_workspace_bytes(levels, N, H, W, points);
  if (levels > 0 && need == 0) return R3DET_EINVAL;
  if (ws_bytes < need || (need && !ws)) return R3DET_EWS;
  std::vector<void*> parts(levels);
  std::vector<size_t> part_bytes(levels);
  char* p = static_cast<char*>(ws);
  for (int l = 0; l < levels; l++) {
    if (!best_bboxes[l]) return R3DET_EINVAL;
    parts[l] = p, part_bytes[l] = nhwc_lvl_bytes(N, H[l], W[l], points);
    p += part_bytes[l];
  }
  const int k = r3k_frb_index_levels(levels, best_bboxes, N, H, W, spatial_scales, points, parts.data(), part_bytes.data(),
                                     S(stream), tables);
  if (k <= 0) return rc(k);
  for (int l = 0; l < levels; l++) {  // (not every level takes the sorted form: level by level)
    const int k2 = r3k_frb_index(best_bboxes[l], N, H[l], W[l], spatial_scales[l], points, parts[l], part_bytes[l], S(stream),
                                 tables ? tables[l] : nullptr);
    if (k2) return rc(k2);
  }
  return R3DET_OK;
}
}  // namespace

int r3det_feature_refine_backward_nhwc_index_levels(int levels, const float* const* best_bboxes, int N, const int* H,
                                                    const int* W, const float* spatial_scales, int points, void* ws,
                                                    size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  return frb_index_levels(levels, best_bboxes, nullptr, N, H, W, spatial_scales, points, ws, ws_bytes, stream);
}

int r3det_feature_refine_backward_nhwc_index_levels_tab(int levels, const float* const* best_bboxes,
                                                        const float* const* tables, int N, const int* H, const int* W,
                                                        const float* spatial_scales, int points, void* ws,
                                                        size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (!tables) return R3DET_EINVAL;
  return frb_index_levels(levels, best_bboxes, tables, N, H, W, spatial_scales, points, ws, ws_bytes, stream);
}

int r3det_feature_refine_backward_nhwc_levels_indexed(int levels, const float* const* top_grads, int N, int C,
                                                      const int* H, const int* W, int points,
                                                      float* const* bottom_grads, int overwrite, void* ws,
                                                      size_t ws_bytes, void* stream) {
  const DeviceGuard guard(stream);
  if (levels < 0 || N <= 0 || C <= 0 || (points != 1 && points != 5) ||
      (levels > 0 && (!top_grads || !H || !W || !bottom_grads)))
    return R3DET_EINVAL;
  const size_t need = r3det_fr_backward_nhwc_levels_workspace_bytes(levels, N, H, W, points);
  if (levels > 0 && need == 0) return R3DET_EINVAL;
  if (ws_bytes < need || (need && !ws)) return R3DET_EWS;
  std::vector<void*> parts(levels);
  std::vector<size_t> part_bytes(levels);
  char* p = static_cast<char*>(ws);
  for (int l = 0; l < levels; l++) {
    if (!top_grads[l] || !bottom_grads[l]) return R3DET_EINVAL;
    parts[l] = p, part_bytes[l] = nhwc_lvl_bytes(N, H[l], W[l], points);
    p += part_bytes[l];
  }
  return rc(r3k_frb_gather_levels(levels, top_grads, N, C, H, W, points, bottom_grads, overwrite, parts.data(),
                                  part_bytes.data(), S(stream)));
}

int r3det_filter_bboxes(const float* cls_score, const long long* cls_strides, const float* bbox_pred,
                        const long long* pred_strides, const float* anchors, int anchors_per_image, int N,
                        int num_anchors, int num_classes, int H, int W, float max_ratio, float* boxes_out,
                        void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_filter_bboxes(cls_score, cls_strides, bbox_pred, pred_strides, anchors, anchors_per_image, N,
                              num_anchors, num_classes, H, W, max_ratio, boxes_out, S(stream)));
}

int r3det_polygon_iou(const float* polys_a, int na, const float* polys_b, int nb, float* out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_polygon_iou(polys_a, na, polys_b, nb, out, S(stream)));
}

int r3det_poly_iou_mat(const float* a, int na, int stride_a, const float* b, int nb, int stride_b, float* out,
                       void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_poly_iou_mat(a, na, stride_a, b, nb, stride_b, out, S(stream)));
}

size_t r3det_poly_nms_workspace_bytes(int n) { return r3k_poly_nms_workspace_bytes(n); }

int r3det_nms_poly(const float* dets9, const int64_t* order, int n, float thr, void* ws, size_t ws_bytes,
                   int64_t* keep_out, int32_t* count_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_poly_nms(dets9, order, n, thr, ws, ws_bytes, keep_out, count_out, S(stream)));
}

int r3det_convex_sort(const float* pts, const unsigned char* masks, int B, int P, int circular, void* ws,
                      size_t ws_bytes, int64_t* index_out, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_convex_sort(pts, masks, B, P, circular, ws, ws_bytes, index_out, S(stream)));
}

int r3det_bias_act(float* y, const float* bias, const float* residual, long long outer, int C, long long inner,
                   int relu, void* stream) {
  const DeviceGuard guard(stream);
  return rc(r3k_bias_act(y, bias, residual, outer, C, inner, relu, S(stream)));
}

int r3det_fr_profile_read(float* records, int capacity) {
  if (capacity < 0 || (capacity > 0 && !records)) return 0;
  return r3k_fr_profile_read(records, capacity);
}

int r3det_set_option(const char* name, int value) {
  if (!name) return R3DET_EINVAL;
  if (!strcmp(name, "fr_impl")) g_r3_fr_impl = value;
  else if (!strcmp(name, "fr_dbg")) g_r3_fr_dbg = value;
  else if (!strcmp(name, "fr_walk")) g_r3_fr_walk = value < 0 ? 0 : value > 1024 && !R3_HAS_PROBES ? 1024 : value;
  else if (!strcmp(name, "fr_profile")) g_r3_fr_profile = value;
  else if (!strcmp(name, "frb_impl")) g_r3_frb_impl = value;
  else if (!strcmp(name, "frn_stamps_lo")) g_r3_frn_stamps = (g_r3_frn_stamps & 0xffffffff00000000ull) | (unsigned)value;
  else if (!strcmp(name, "frn_stamps_hi")) g_r3_frn_stamps = (g_r3_frn_stamps & 0xffffffffull) | ((unsigned long long)(unsigned)value << 32);
  else if (!strcmp(name, "iou_impl")) g_r3_iou_impl = value;
  else if (!strcmp(name, "iou_small")) g_r3_iou_small = value;
  else if (!strcmp(name, "clip_impl")) g_r3_clip_impl = value;
  else if (!strcmp(name, "iou_qcap")) g_r3_iou_qcap = value;
  else if (!strcmp(name, "iou_dwgs")) g_r3_iou_dwgs = value;
  else if (!strcmp(name, "iou_dyn")) g_r3_iou_dyn = value != 0;
  else if (!strcmp(name, "iou_nfill")) g_r3_iou_nfill = value < 0 ? 0 : value;
  else if (!strcmp(name, "iou_order")) g_r3_iou_order = value < -1 ? -1 : value > 31 ? 31 : value;
  else if (!strcmp(name, "nms_impl")) g_r3_nms_impl = value;
  else if (!strcmp(name, "nms_qcap")) g_r3_nms_qcap = value;
  else return R3DET_EINVAL;
  return R3DET_OK;
}

}  // extern "C"
