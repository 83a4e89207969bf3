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
#include <stdint.h>

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

// channels_last heads (dense NHWC, A > 1): a position's A*C logits are contiguous, so one thread per
// position reads with a 4*A*C-byte stride between lanes (0.45 TB/s at level 0).  Here a workgroup stages
// the logits of 64 consecutive positions -- one contiguous chunk -- in LDS with coalesced 16-byte loads,
// each thread then walks its own record (LDS stride A*C, odd for the shipped 9 x 15: conflict-free), and
// the 5 deltas of the winning anchor come through the same buffer.
constexpr int FB_POS = 64;

// AT, CT: compile-time A, C (0 = use the runtime values): the shipped heads (9 anchors x 15 classes) get
// fully unrolled loops -- with runtime bounds every LDS read of the 135-long scan waits for the previous one
template <int AT, int CT>
__global__ __launch_bounds__(FB_POS) void filter_bboxes_nhwc_kernel(const float* __restrict__ cls,
                                                                    const float* __restrict__ reg,
                                                                    const float* __restrict__ anchors, int per_image,
                                                                    long long total, int HW, int A_rt, int C_rt,
                                                                    float max_ratio, float* __restrict__ out) {
  const int A = AT ? AT : A_rt, C = CT ? CT : C_rt;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int tid = threadIdx.x;
  const long long pos0 = (long long)blockIdx.x * FB_POS;
  const int npos = (int)min((long long)FB_POS, total - pos0);
  const int AC = A * C, A5 = A * 5;
  {
    const float* src = cls + pos0 * AC;  // 16-byte aligned: FB_POS * AC * 4 is a multiple of 16
    const int n4 = npos * AC / 4, nflt = npos * AC;
#pragma unroll 8
    for (int i = tid; i < n4; i += FB_POS) reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int i = n4 * 4 + tid; i < nflt; i += FB_POS) tile[i] = src[i];
  }
  __syncthreads();
  int best_a = 0;
  if (tid < npos) {
    const float* cb = tile + tid * AC;
    float best = -INFINITY;
#pragma unroll
    for (int a = 0; a < A; a++) {
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) m = fmaxf(m, cb[a * C + c]);
      if (m > best || a == 0) {  // first maximum on ties, as argmax
        best = m;
        best_a = a;
      }
    }
  }
  __syncthreads();
  {
    const float* src = reg + pos0 * A5;
    const int n4 = npos * A5 / 4, nflt = npos * A5;
#pragma unroll 8
    for (int i = tid; i < n4; i += FB_POS) reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int i = n4 * 4 + tid; i < nflt; i += FB_POS) tile[i] = src[i];
  }
  __syncthreads();
  if (tid >= npos) return;
  const long long t = pos0 + tid;
  const float* rb = tile + tid * A5 + best_a * 5;
  const float d0 = rb[0], d1 = rb[1], d2 = rb[2], d3 = rb[3], d4 = rb[4];
  const long long p = per_image ? t : t % HW;
  const float* an = anchors + (p * A + best_a) * 5;
  const float ax = an[0], ay = an[1], aw = an[2], ah = an[3], aa = an[4];
  const float dw = fminf(fmaxf(d2, -max_ratio), max_ratio);
  const float dh = fminf(fmaxf(d3, -max_ratio), max_ratio);
  float* o = out + t * 5;
  o[0] = ax + aw * d0;
  o[1] = ay + ah * d1;
  o[2] = aw * expf(dw);
  o[3] = ah * expf(dh);
  o[4] = aa + d4;
}

inline bool dense_nhwc(const Strides4& s, int ch, int H, int W) {
  return s.c == 1 && s.w == ch && s.h == (long long)W * ch && s.n == (long long)H * W * ch;
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
  const size_t tile_bytes = (size_t)FB_POS * A * (C > 5 ? C : 5) * sizeof(float);
  if (A > 1 && dense_nhwc(sc, A * C, H, W) && dense_nhwc(sr, A * 5, H, W) && tile_bytes <= 64 * 1024 &&
      (reinterpret_cast<uintptr_t>(cls) & 15) == 0 && (reinterpret_cast<uintptr_t>(reg) & 15) == 0 &&
      (FB_POS * A * C) % 4 == 0 && (FB_POS * A * 5) % 4 == 0) {
    const dim3 grid((unsigned)((total + FB_POS - 1) / FB_POS));
    if (A == 9 && C == 15)
      hipLaunchKernelGGL((filter_bboxes_nhwc_kernel<9, 15>), grid, dim3(FB_POS), tile_bytes, stream, cls, reg, anchors,
                         per_image, total, H * W, A, C, max_ratio, out);
    else
      hipLaunchKernelGGL((filter_bboxes_nhwc_kernel<0, 0>), grid, dim3(FB_POS), tile_bytes, stream, cls, reg, anchors,
                         per_image, total, H * W, A, C, max_ratio, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  hipLaunchKernelGGL(filter_bboxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, cls, sc, reg, sr,
                     anchors, per_image, N, A, C, H, W, max_ratio, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
