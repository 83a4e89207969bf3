// harness_v2.cpp -- C-ABI driver around the REFERENCE header
//   r3det/ops/ml_nms_rotated/src/box_iou_rotated_utils.h
// (#included from where it lies; path injected as REF_ML_UTILS_H).  That header
// is the in-tree statement of the mmcv/detectron2 vertex convention used by
// RBboxOverlaps2D_v2 and nms type 'v2'/'mmcv'.
//
// ml_nms_rotated/src/nms_rotated_cpu.cpp itself does not compile against
// torch 2.10 (AT_DISPATCH_FLOATING_TYPES(dets.type(), ...) at :67), so only
// the header's single_box_iou_rotated<float> (ml utils.h:314-347, label guard
// :316-322) is driven here; the greedy loop of nms_rotated_cpu.cpp:35-56 is
// token-identical to the v3 one already covered by harness_v3.cpp and is
// re-driven below over this header's IoU.
//
// TEST INFRASTRUCTURE ONLY (see oracle/r3_oracle.cpp header).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include REF_ML_UTILS_H

extern "C" {

// boxes carry 6 floats: [x, y, w, h, a, label]
void ref_v2_iou_mat(const float* b1, int n1, const float* b2, int n2, float* out) {
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++)
      out[(size_t)i * n2 + j] = single_box_iou_rotated<float>(b1 + (size_t)i * 6, b2 + (size_t)j * 6);
}

int ref_v2_nms(const float* dets6, const float* scores, int n, float thr, int64_t* keep) {
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
  std::vector<uint8_t> sup(n, 0);
  int cnt = 0;
  for (int _i = 0; _i < n; _i++) {
    auto i = order[_i];
    if (sup[i]) continue;
    keep[cnt++] = i;
    for (int _j = _i + 1; _j < n; _j++) {
      auto j = order[_j];
      if (sup[j]) continue;
      auto ovr = single_box_iou_rotated<float>(dets6 + i * 6, dets6 + j * 6);
      if (ovr >= thr) sup[j] = 1;
    }
  }
  return cnt;
}

}  // extern "C"
