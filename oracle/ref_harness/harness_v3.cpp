// harness_v3.cpp -- C-ABI driver around the REFERENCE's own CPU functions
//   box_iou_rotated_cpu   (r3det/ops/box_iou_rotated/src/box_iou_rotated_cpu.cpp:23-38)
//   nms_rotated_cpu       (r3det/ops/nms_rotated/src/nms_rotated_cpu.cpp:63-74)
// Those two .cpp files are compiled from where they lie under /root/reference
// by oracle/build_ref.py and linked with this file (one .so per op, so the two
// differing box_iou_rotated_utils.h copies never meet in one image).
//
// TEST INFRASTRUCTURE ONLY (see oracle/r3_oracle.cpp header).
#include <torch/types.h>

#include <cstdint>
#include <cstring>

#ifdef HARNESS_IOU
at::Tensor box_iou_rotated_cpu(const at::Tensor& boxes1, const at::Tensor& boxes2,
                               const bool iou_or_iof);

extern "C" void ref_v3_iou_mat(const float* b1, int n1, const float* b2, int n2, int iou_or_iof,
                               float* out) {
  auto t1 = at::from_blob(const_cast<float*>(b1), {n1, 5}, at::kFloat).clone();
  auto t2 = at::from_blob(const_cast<float*>(b2), {n2, 5}, at::kFloat).clone();
  auto r = box_iou_rotated_cpu(t1, t2, iou_or_iof != 0).contiguous();
  std::memcpy(out, r.data_ptr<float>(), sizeof(float) * (size_t)n1 * n2);
}
#endif

#ifdef HARNESS_NMS
at::Tensor nms_rotated_cpu(const at::Tensor& dets, const at::Tensor& scores,
                           const float iou_threshold);

extern "C" int ref_v3_nms(const float* dets5, const float* scores, int n, float thr,
                          int64_t* keep) {
  auto d = at::from_blob(const_cast<float*>(dets5), {n, 5}, at::kFloat).clone();
  auto s = at::from_blob(const_cast<float*>(scores), {n}, at::kFloat).clone();
  auto r = nms_rotated_cpu(d, s, thr).contiguous();
  int k = (int)r.numel();
  if (k) std::memcpy(keep, r.data_ptr<int64_t>(), sizeof(int64_t) * k);
  return k;
}
#endif
