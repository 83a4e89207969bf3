// harness_rank4.cpp -- C-ABI drivers around the REFERENCE's own CPU code for the "rank 4" ops
//   convex_sort_cpu   (r3det/ops/convex/src/convex_cpu.cpp:93-108), linked from where it lies;
//   polygon_iou       (r3det/ops/polygon_geo/src/polygon_geo_cpu.cpp:272-283), #included
//                     (its PYBIND11_MODULE only needs a module name).
// Built by oracle/build.py into oracle/_ref/.  TEST INFRASTRUCTURE ONLY.
#include <torch/extension.h>

#include <cstdint>
#include <cstring>

#ifdef HARNESS_CONVEX
at::Tensor convex_sort_cpu(const at::Tensor& pts, const at::Tensor& masks, const bool circular);

extern "C" void ref_convex_sort(const float* pts, const uint8_t* masks, int B, int P, int circular, int64_t* out) {
  auto p = at::from_blob(const_cast<float*>(pts), {B, P, 2}, at::kFloat).clone();
  auto m = at::from_blob(const_cast<uint8_t*>(masks), {B, P}, at::kByte).clone().to(at::kBool);
  auto r = convex_sort_cpu(p, m, circular != 0).contiguous();
  std::memcpy(out, r.data_ptr<int64_t>(), sizeof(int64_t) * (size_t)r.numel());
}
#endif

#ifdef HARNESS_POLYGON
#include REF_POLYGON_CPP

extern "C" void ref_polygon_iou(const float* a, int na, const float* b, int nb, float* out) {
  auto ta = at::from_blob(const_cast<float*>(a), {na, 8}, at::kFloat).clone();
  auto tb = at::from_blob(const_cast<float*>(b), {nb, 8}, at::kFloat).clone();
  auto r = polygon_iou(ta, tb).contiguous();
  std::memcpy(out, r.data_ptr<float>(), sizeof(float) * (size_t)na * nb);
}
#endif
