// r3_boxes.hip -- producers of the boxes the Feature Refinement sampler consumes.
//
// RRetinaHead.filter_bboxes (models/dense_heads/rotate_retina_head.py:117-179): per image and
// level, permute cls_score to (H*W, A, C), take the best class score of every anchor, pick the
// best anchor of every position, gather its 5 deltas and its anchor, decode with
// delta2bbox_v1 (core/bbox/coder/delta_xywha_rbbox_coder.py:142-211, means 0 / stds 1).
// The reference (and a plain torch port) runs ~10 elementwise / reduce / gather launches per level
// and materialises the permuted maps; here one thread per position reads its A*C scores and
// 5 deltas straight from the head outputs -- through their strides, so NCHW and channels_last
// tensors both work without a copy -- and writes the (N*H*W, 5) box array FR reads.
// RRetinaRefineHead.refine_bboxes (rotate_retina_refine_head.py:56-97) is the A = 1 case with
// the previous boxes as anchors (no scores).
#include <hip/hip_runtime.h>

#include "r3_kernels.h"

namespace {

struct Strides4 {
  long long n, c, h, w;  // in elements
};

// cls: (N, A*C, H, W) logits (any strides), may be null when A == 1; reg: (N, A*5, H, W);
// anchors: per_image ? (N, H*W*A, 5) : (H*W*A, 5); out: (N, H*W, 5)
__global__ __launch_bounds__(256) void filter_bboxes_kernel(const float* __restrict__ cls, Strides4 sc,
                                                            const float* __restrict__ reg, Strides4 sr,
                                                            const float* __restrict__ anchors, int per_image,
                                                            int N, int A, int C, int H, int W, float max_ratio,
                                                            float* __restrict__ out) {
  const int HW = H * W;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)N * HW) return;
  const int n = (int)(t / HW), p = (int)(t - (long long)n * HW);
  const int h = p / W, w = p - h * W;
  int best_a = 0;
  if (A > 1) {
    // cls.max(dim=-1)[0].argmax(dim=-1): first maximum on ties
    const float* cb = cls + n * sc.n + h * sc.h + w * sc.w;
    float best = -INFINITY;
    for (int a = 0; a < A; a++) {
      float m = -INFINITY;
      for (int c = 0; c < C; c++) m = fmaxf(m, cb[(long long)(a * C + c) * sc.c]);
      if (m > best || a == 0) {
        best = m;
        best_a = a;
      }
    }
  }
  const float* rb = reg + n * sr.n + h * sr.h + w * sr.w + (long long)(best_a * 5) * sr.c;
  const float d0 = rb[0], d1 = rb[sr.c], d2 = rb[2 * sr.c], d3 = rb[3 * sr.c], d4 = rb[4 * sr.c];
  const float* an = anchors + (((long long)(per_image ? n : 0) * HW + p) * A + best_a) * 5;
  const float ax = an[0], ay = an[1], aw = an[2], ah = an[3], aa = an[4];
  // delta2bbox_v1 (delta_xywha_rbbox_coder.py:171-196): dw / dh clamped to +-|log(wh_ratio_clip)|
  const float dw = fminf(fmaxf(d2, -max_ratio), max_ratio);
  const float dh = fminf(fmaxf(d3, -max_ratio), max_ratio);
  float* o = out + t * 5;
  o[0] = ax + aw * d0;
  o[1] = ay + ah * d1;
  o[2] = aw * expf(dw);
  o[3] = ah * expf(dh);
  o[4] = aa + d4;
}

}  // namespace

int r3k_filter_bboxes(const float* cls, const long long* cls_strides, const float* reg,
                      const long long* reg_strides, const float* anchors, int per_image, int N, int A, int C,
                      int H, int W, float max_ratio, float* out, hipStream_t stream) {
  if (N <= 0 || A <= 0 || H <= 0 || W <= 0 || !reg || !reg_strides || !anchors || !out) return -1;
  if (A > 1 && (!cls || !cls_strides || C <= 0)) return -1;
  Strides4 sc{0, 0, 0, 0}, sr{reg_strides[0], reg_strides[1], reg_strides[2], reg_strides[3]};
  if (cls_strides) sc = Strides4{cls_strides[0], cls_strides[1], cls_strides[2], cls_strides[3]};
  const long long total = (long long)N * H * W;
  hipLaunchKernelGGL(filter_bboxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, cls, sc, reg, sr,
                     anchors, per_image, N, A, C, H, W, max_ratio, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
