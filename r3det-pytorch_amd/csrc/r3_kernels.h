// r3_kernels.h -- internal launch functions behind the C ABI (include/r3det_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

// The switches of r3det_set_option: process-wide, written by any thread, read (once per call) by whichever thread runs an
// operator -- relaxed atomics, so that two threads flipping options and calling operators do not race (round 5).
struct R3Option {
  std::atomic<int> v;
  explicit constexpr R3Option(int x) : v(x) {}
  operator int() const { return v.load(std::memory_order_relaxed); }
  R3Option& operator=(int x) {
    v.store(x, std::memory_order_relaxed);
    return *this;
  }
};
struct R3Option64 {
  std::atomic<unsigned long long> v;
  explicit constexpr R3Option64(unsigned long long x) : v(x) {}
  unsigned long long get() const { return v.load(std::memory_order_relaxed); }
  operator unsigned long long() const { return v.load(std::memory_order_relaxed); }
  R3Option64& operator=(unsigned long long x) {
    v.store(x, std::memory_order_relaxed);
    return *this;
  }
};

// Per-device state.  The dynamic-LDS opt-in (hipFuncSetAttribute) belongs to the CURRENT DEVICE's copy of a kernel and
// the compute-unit count to the device: a process that drives several GPUs (the reference's ops run under
// MMDataParallel) needs both per device, not per process.
constexpr int R3_MAX_DEVICES = 64;
struct R3DeviceOnce {
  bool done[R3_MAX_DEVICES] = {};
  bool first() {  // true the first time it is asked on the current device (always, beyond R3_MAX_DEVICES)
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= R3_MAX_DEVICES) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};
int r3_cu_count();  // compute units of the current device (r3_api.hip)
// Zero `bytes` bytes at p (both multiples of 4) with a KERNEL.  Not hipMemsetAsync: a memset node inside a captured HIP
// graph faulted on replay ("Memory access fault by GPU") whenever an ordinary kernel or copy had been enqueued on the
// stream in front of the graph launch (round 5, ROCm 7.2: tools/dense_graph_bisect.py -- the graph of the whole dense
// step replays cleanly up to the refine head, faults with the pre-NMS pool's one memset in it, and is clean again with
// this kernel in its place).  Returns 0 / -2.
int r3k_zero_async(void* p, size_t bytes, hipStream_t stream);

// Experiment-only forms -- the A/B kernels of tools/ (launch variants that were measured and not shipped), clock
// stamps inside kernels -- exist only in a library built with -DR3_PROBES (`make probes`: libr3det_hip_probes.so,
// loaded when R3DET_HIP_LIB names it).  The product library keeps the switches its tests need (documented with
// r3det_set_option in include/r3det_hip.h) and reads each ONCE per call.
#ifdef R3_PROBES
constexpr bool R3_HAS_PROBES = true;
#else
constexpr bool R3_HAS_PROBES = false;
#endif

// return 0 ok, -1 bad argument, -2 launch failure, -3 workspace too small
size_t r3k_iou_workspace_bytes(int n1, int n2);
// ws may be null (single-kernel path); with a workspace the stream + drain pipeline runs
// prepared: r3k_iou_prepare_columns of b2 for this geom (or null)
int r3k_iou_mat(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                void* ws, size_t ws_bytes, hipStream_t stream, const void* prepared = nullptr,
                int* thin_done = nullptr);  // thin_done (geom 3): apply obb_overlaps' thin-box rule in the launch where it can; 1 = done
size_t r3k_iou_prepared_bytes(int n2);
int r3k_iou_prepare_columns(int geom, const float* b2, int n2, void* prepared, size_t bytes, hipStream_t stream);
// obb_overlaps' epilogue: zero the rows / columns of boxes with min(w, h) < 1e-3
int r3k_iou_zero_thin(const float* b1, int n1, const float* b2, int n2, float* out, hipStream_t stream);
int r3k_iou_vec(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                hipStream_t stream);

// fused MaxIoU assignment on the (gts x boxes) overlaps without materialising them
size_t r3k_iou_assign_workspace_bytes(int n1, int n2);
int r3k_iou_assign(int geom, const float* gts, int n1, const float* boxes, int n2, float pos_thr, float neg_thr,
                   float min_pos_iou, int match_low, int assign_all, int64_t* assigned, float* max_overlaps,
                   int64_t* argmax, float* gt_max, int64_t* gt_argmax, void* ws, size_t ws_bytes,
                   hipStream_t stream, const void* prepared = nullptr, const int64_t* gt_labels = nullptr,
                   int64_t* labels = nullptr);  // gt_labels / labels: both or neither -- mmdet's assigned_labels too

size_t r3k_nms_workspace_bytes(int n);
// dets: (n, det_stride) original order; labels: int64 (n,) or null; order: int64 (n,)
int r3k_nms(int geom, const float* dets, int det_stride, const int64_t* labels,
            const int64_t* order, int n, float thr, int sort_ascending, void* ws, size_t ws_bytes,
            int64_t* keep_out, int32_t* count_out, hipStream_t stream);

// batched multiclass NMS (v1): select -> [host reads counts] -> sort / prepare / stream / drain /
// reduce / finish for B images at once.  Candidate arrays have stride n * K per image.
size_t r3k_mcnms_select_workspace_bytes(int B, int n);
int r3k_mcnms_select(const float* boxes, const float* scores, int B, int n, int K, float score_thr,
                     int* cand_row, int* cand_label, float* cand_score, int* cand_rank, int* counts,
                     float* maxc, void* ws, size_t ws_bytes, hipStream_t stream);
size_t r3k_mcnms_workspace_bytes(int B, int cap);
size_t r3k_batched_rnms_workspace_bytes(int n);
int r3k_batched_nms(int geom, const float* boxes, const float* scores, const int64_t* inds, int n, float thr, void* ws,
                    size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, hipStream_t stream);
int r3k_mcnms_v1(const float* boxes, int B, int n, int K, const int* cand_row, const int* cand_label,
                 const float* cand_score, int* cand_rank, const int* counts, const float* maxc, int cap,
                 float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out, int64_t* labels_out,
                 int64_t* keep_idx_out, int32_t* counts_out, hipStream_t stream);
// geom 1 = v1 (same as above), 3 = obb_batched_nms, 2 = ml_nms_rotated; 2/3 emit score order
// padded (or null): dets_out is (B, out_cap, 7) = [box, score, label] with img_stride floats between images, rows beyond
// the count zeroed; the count also as fp32 at count_f32[img * count_f32_stride]; overflow[img] = counts[img] > cap
struct R3kMcPadded {
  size_t img_stride;
  float* count_f32;
  size_t count_f32_stride;
  int32_t* overflow;
};
int r3k_mcnms_run(int geom, const float* boxes, int B, int n, int K, const int* cand_row, const int* cand_label,
                  const float* cand_score, int* cand_rank, const int* counts, const float* maxc, int cap,
                  float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out, int64_t* labels_out,
                  int64_t* keep_idx_out, int32_t* counts_out, hipStream_t stream, const R3kMcPadded* padded = nullptr,
                  int scale_parts = 0);  // (scale_parts > 0: maxc holds rnms_begin_kernel's partial results, r3_nms.hip)

size_t r3k_fr_workspace_bytes(int N, int H, int W, int points);
// ws may be null (taps derived per channel plane); with a workspace: tap table + unpack kernel
int r3k_fr_forward(const float* feat, const float* boxes, int N, int C, int H, int W, float scale,
                   int points, float* out, void* ws, size_t ws_bytes, hipStream_t stream);
int r3k_fr_forward_levels(int levels, const float* const* feat, const float* const* boxes, int N, int C, const int* H,
                          const int* W, const float* scales, int points, float* const* out, void* ws, size_t ws_bytes,
                          hipStream_t stream);
// ws may be null; with r3k_fr_backward_workspace_bytes() of workspace the call is the gather over the inverse tap
// index (r3_frb.hip); index_ready: ws holds r3k_frn_index of these boxes
size_t r3k_fr_backward_workspace_bytes(int N, int H, int W, int points);
int r3k_fr_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W,
                    float scale, int points, float* bottom_grad, int overwrite, void* ws, size_t ws_bytes,
                    int index_ready, hipStream_t stream);
// NCHW gather backward (r3_frb.hip): index = CSR + SELL-64 of the boxes, then the gather alone
size_t r3k_frn_workspace_bytes(int N, int H, int W, int points);
// tab / tabs[l]: the level's tap table ([N][y: HW][x: HW] floats, r3k_fr_tap_table_bytes; written by the forward pass
// of the same boxes) or null -- the index kernel's scan then reads 4 contiguous bytes per source instead of the boxes
int r3k_frn_index(const float* boxes, int N, int C, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                  hipStream_t stream, const float* tab = nullptr);
int r3k_frn_index_levels(int levels, const float* const* boxes, int N, int C, const int* H, const int* W,
                         const float* scales, int points, void* const* ws, const size_t* ws_bytes, hipStream_t stream,
                         const float* const* tabs = nullptr);
int r3k_frn_gather(const float* top_grad, int N, int C, int H, int W, int points, float* bottom_grad, int overwrite,
                   void* ws, size_t ws_bytes, hipStream_t stream);
int r3k_frb_index_levels(int levels, const float* const* boxes, int N, const int* H, const int* W, const float* scales,
                         int points, void* const* ws, const size_t* ws_bytes, hipStream_t stream,
                         const float* const* tabs = nullptr);
int r3k_frb_gather_levels(int levels, const float* const* top_grad, int N, int C, const int* H, const int* W, int points,
                          float* const* bottom_grad, int overwrite, void* const* ws, const size_t* ws_bytes,
                          hipStream_t stream);
int r3k_frn_gather_levels(int levels, const float* const* top_grad, int N, int C, const int* H, const int* W, int points,
                          float* const* bottom_grad, int overwrite, void* const* ws, const size_t* ws_bytes, int* taken,
                          hipStream_t stream);

// polygon ops outside the shipped configs (r3_poly.hip)
int r3k_nms_reduce_dense(const unsigned long long* mask, int n, int cb, const int64_t* order, int64_t* keep_out,
                         int32_t* count_out, hipStream_t stream);
int r3k_polygon_iou(const float* a, int na, const float* b, int nb, float* out, hipStream_t stream);
int r3k_poly_iou_mat(const float* a, int na, int sa, const float* b, int nb, int sb, float* out, hipStream_t stream);
size_t r3k_poly_nms_workspace_bytes(int n);
int r3k_poly_nms(const float* dets9, const int64_t* order, int n, float thr, void* ws, size_t ws_bytes,
                 int64_t* keep_out, int32_t* count_out, hipStream_t stream);
int r3k_convex_sort(const float* pts, const unsigned char* masks, int B, int P, int circular, void* ws,
                    size_t ws_bytes, int64_t* out, hipStream_t stream);

// FR box producers: best anchor per position + delta2bbox_v1, strided (NCHW or channels_last) inputs
int r3k_filter_bboxes(const float* cls, const long long* cls_strides, const float* reg,
                      const long long* reg_strides, const float* anchors, int per_image, int N, int A, int C,
                      int H, int W, float max_ratio, float* out, hipStream_t stream);

// pre-NMS pool of one level for a batch: sigmoid, per-image top-nms_pre by the best class score (score order),
// decode with the centre clamp, written at row_offset of the (N, pool_rows, 5) / (N, pool_rows, C + 1) arrays
size_t r3k_level_pool_workspace_bytes(int N, int A, int H, int W, int nms_pre);
int r3k_level_pool(const float* cls, const long long* cls_strides, const float* reg, const long long* reg_strides,
                   const float* anchors, int per_image, int N, int A, int C, int H, int W, int nms_pre, float max_ratio,
                   float clamp_x, float clamp_y, float* boxes, float* scores, int pool_rows, int row_offset, void* ws,
                   size_t ws_bytes, hipStream_t stream);
// ... all levels of a head in one set of launches (levels in pool order; row offsets follow from the level sizes)
#define R3K_POOL_MAX_LEVELS 8
size_t r3k_levels_pool_workspace_bytes(int nlevels, int N, const int* A, const int* H, const int* W, int nms_pre);
int r3k_levels_pool(int nlevels, const float* const* cls, const long long* cls_strides, const float* const* reg,
                    const long long* reg_strides, const float* const* anchors, int per_image, int N, const int* A, int C,
                    const int* H, const int* W, int nms_pre, float max_ratio, float clamp_x, float clamp_y, float* boxes,
                    float* scores, int pool_rows, int row_offset, void* ws, size_t ws_bytes, hipStream_t stream);

// convolution epilogue of the inference model: y = act(y + bias[c] (+ residual)), in place
// channels_last -> NCHW: out = (a + bias_a[c]) (+ (b + bias_b[c])); b and the biases may be null
int r3k_mix_to_nchw(const float* a, const float* b, const float* bias_a, const float* bias_b, int N, int C, int H, int W,
                    float* out, hipStream_t stream);
int r3k_bias_act(float* y, const float* bias, const float* residual, long long outer, int C, long long inner, int relu,
                 hipStream_t stream);

// split form of the FR forward cell path: tap table ahead of time, then the sampler kernel alone
size_t r3k_fr_table_bytes(int N, int H, int W);
int r3k_fr_prepare(const float* boxes, int N, int H, int W, float scale, float* table, hipStream_t stream);
// feat2 / res null: the sampler alone; both given: out = res + ((feat + feat2) + sample(feat + feat2))
int r3k_fr_forward_prepared(const float* feat, const float* feat2, const float* res, const float* table, int N, int C,
                            int H, int W, float* out, hipStream_t stream);

// the module tail of all NCHW levels of a pass (points = 1): one grid for the coarse levels + the cell levels' tables,
// one fused cell launch per cell level; tables[l]: r3k_fr_table_bytes of storage per cell level.  -1: nothing launched
int r3k_fr_module_levels(int levels, const float* const* a, const float* const* b, const float* const* res,
                         const float* const* boxes, int N, int C, const int* H, const int* W, const float* scales,
                         float* const* out, float* const* tables, hipStream_t stream);
size_t r3k_fr_table_bytes(int N, int H, int W);

// channels_last (N, H, W, C) sampler; b / biases / res non-null: the module tail out = res + (P + sample(P)),
// P = (a + bias_a) + (b + bias_b)
// tab (points = 1 only): the launch also writes the level's tap table (r3k_fr_tap_table_bytes) for the backward's index
int r3k_fr_forward_nhwc(const float* a, const float* b, const float* bias_a, const float* bias_b, const float* res,
                        const float* boxes, int N, int C, int H, int W, float scale, int points, float* out,
                        hipStream_t stream, float* tab = nullptr);
inline size_t r3k_fr_tap_table_bytes(int N, int H, int W) {
  return (N > 0 && H > 0 && W > 0) ? (size_t)N * H * W * 2 * sizeof(float) : 0;
}

// ... all pyramid levels of a pass: levels without the wide form as ONE grid (host arrays of device pointers)
int r3k_fr_forward_nhwc_levels(int levels, const float* const* a, const float* const* b, const float* bias_a,
                               const float* bias_b, const float* const* res, const float* const* boxes, int N, int C,
                               const int* H, const int* W, const float* scales, int points, float* const* out,
                               hipStream_t stream, float* const* tabs = nullptr);

// channels_last backward (r3_frb.hip): inverse tap index of the boxes + gather; points 1 or 5, any H x W with W <= 4096
size_t r3k_frb_workspace_bytes(int N, int H, int W, int points);
int r3k_frb_index(const float* boxes, int N, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                  hipStream_t stream, const float* tab = nullptr);
int r3k_frb_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W, float scale, int points,
                     float* bottom_grad, int overwrite, void* ws, size_t ws_bytes, int index_ready,
                     hipStream_t stream);

// profiling ring of the FR cell path (see r3det_fr_profile_read)
int r3k_fr_profile_read(float* records, int capacity);
extern R3Option g_r3_fr_profile;

// A/B knobs (r3det_set_option)
extern R3Option g_r3_fr_impl;   // 0 auto, 1 generic, 2 lds-plane, 3/4 tap-table, 5/6 persistent
extern R3Option g_r3_fr_dbg;    // NHWC forward: 0 auto | 8 wide regions | 9 tile pairs; other values: probes builds only
// (the value a call works with: what the product library does not know reads as 0)
inline int r3_fr_dbg() {
  const int v = g_r3_fr_dbg;
  return (R3_HAS_PROBES || v == 8 || v == 9) ? v : 0;
}
extern R3Option g_r3_fr_walk;   // strip height of the tile-pair walk (0: row-major)
extern R3Option64 g_r3_frn_stamps;
extern R3Option g_r3_frb_impl;  // 0 auto; 1 general index form always; 2 unpaired gather
extern R3Option g_r3_iou_impl;  // 0 auto, 1 one thread per pair, 2 one-launch compact kernel, 4 prep + stream + drain pipeline always
extern R3Option g_r3_clip_impl; // v1 pair clip of the drains: 0 straight-line form (r3_clip.h), 1 the LDS-list form (r3_geom_lds.h, rounds 2-4)
extern R3Option g_r3_iou_dyn;   // drain: 1 a wavefront's blocks behind its first are handed out by an atomic ticket | 0 static stride
extern R3Option g_r3_iou_nfill; // (iou_impl 5) fill workgroups of the one-launch drain; 0: one per compute unit
extern R3Option g_r3_iou_order; // stream kernel: -1 all workgroups zero their tile early; b >= 0: those with bit b of the linear index
extern R3Option g_r3_iou_dwgs;  // 0 default (2048); > 0: workgroups of the IoU drain kernel (tuning)
extern R3Option g_r3_iou_qcap;  // 0 default; > 0 caps the IoU pipeline's global pair queue (tests the overflow path)
extern R3Option g_r3_iou_small; // 0 default (513); > 0: column count from which the pipeline runs
extern R3Option g_r3_nms_impl;  // 0 auto (queue pipeline), 1 tile kernels
extern R3Option g_r3_nms_qcap;  // 0 default; > 0 caps the global pair queue (tests the overflow path)
