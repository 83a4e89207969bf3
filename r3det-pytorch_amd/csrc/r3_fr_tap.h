// r3_fr_tap.h -- sample-point arithmetic of the Feature Refinement sampler, shared by the forward / backward
// kernels (r3_fr.hip) and the channels_last backward (r3_frb.hip).
// Restates bilinear_interpolate / bilinear_interpolate_gradient's coordinate logic
// (fr/src/feature_refine_kernel.cu:16-52,67-106) and the sample points of a position (:125-151).
#pragma once
#include <hip/hip_runtime.h>

#include "r3_trig.h"

namespace {

struct Tap {
  int o00, o01, o10, o11;  // offsets inside a plane with row pitch `pitch`
  float w1, w2, w3, w4;
  bool valid;
};

// bilinear_interpolate / _gradient coordinate logic (feature_refine_kernel.cu:16-52,67-106)
__device__ __forceinline__ Tap make_tap(int height, int width, int pitch, float y, float x) {
  Tap t;
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    t.valid = false;
    t.o00 = t.o01 = t.o10 = t.o11 = 0;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  t.valid = true;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = (float)(1. - (double)ly);
  float hx = (float)(1. - (double)lx);
  t.w1 = hy * hx;
  t.w2 = hy * lx;
  t.w3 = ly * hx;
  t.w4 = ly * lx;
  t.o00 = y_low * pitch + x_low;
  t.o01 = y_low * pitch + x_high;
  t.o10 = y_high * pitch + x_low;
  t.o11 = y_high * pitch + x_high;
  return t;
}

// sample points of one position (feature_refine_kernel.cu:125-151)
template <int POINTS>
__device__ __forceinline__ void make_taps(const float* __restrict__ box, float scale, int H, int W,
                                          int pitch, Tap* taps) {
  float roi_y = box[0] * scale;  // sic: row <- x_ctr
  float roi_x = box[1] * scale;  //      col <- y_ctr
  taps[0] = make_tap(H, W, pitch, roi_y, roi_x);
  if (POINTS > 1) {
    float roi_w = box[2] * scale;
    float roi_h = box[3] * scale;
    float roi_a = box[4];
    float w_2 = roi_w / 2, h_2 = roi_h / 2;
    float sina, cosa;
    r3_sincos(roi_a, sina, cosa);
    float wx = cosa * w_2, wy = sina * w_2;
    float hx = -sina * h_2, hy = cosa * h_2;
    taps[1] = make_tap(H, W, pitch, roi_y + wy + hy, roi_x + wx + hx);
    taps[2] = make_tap(H, W, pitch, roi_y - wy + hy, roi_x - wx + hx);
    taps[3] = make_tap(H, W, pitch, roi_y - wy - hy, roi_x - wx - hx);
    taps[4] = make_tap(H, W, pitch, roi_y + wy - hy, roi_x + wx - hx);
  }
}

// The same logic with the four cells as (row, column) pairs instead of plane offsets -- for code that addresses
// cells, not a plane (the channels_last backward's inverse index).  Same operations in the same order as make_tap:
// the weights are bit-identical.
struct TapYX {
  int yl, xl, yh, xh;
  float w[4];  // w1 (yl, xl), w2 (yl, xh), w3 (yh, xl), w4 (yh, xh)
  bool valid;
};

__device__ __forceinline__ TapYX make_tap_yx(int height, int width, float y, float x) {
  TapYX t;
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    t.valid = false;
    t.yl = t.xl = t.yh = t.xh = 0;
    t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
    return t;
  }
  t.valid = true;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = (float)(1. - (double)ly);
  float hx = (float)(1. - (double)lx);
  t.w[0] = hy * hx;
  t.w[1] = hy * lx;
  t.w[2] = ly * hx;
  t.w[3] = ly * lx;
  t.yl = y_low; t.xl = x_low; t.yh = y_high; t.xh = x_high;
  return t;
}

// sample points of one position as cells (feature_refine_kernel.cu:125-151)
template <int POINTS>
__device__ __forceinline__ void make_taps_yx(const float* __restrict__ box, float scale, int H, int W, TapYX* taps) {
  float roi_y = box[0] * scale;  // sic: row <- x_ctr
  float roi_x = box[1] * scale;  //      col <- y_ctr
  taps[0] = make_tap_yx(H, W, roi_y, roi_x);
  if (POINTS > 1) {
    float roi_w = box[2] * scale;
    float roi_h = box[3] * scale;
    float roi_a = box[4];
    float w_2 = roi_w / 2, h_2 = roi_h / 2;
    float sina, cosa;
    r3_sincos(roi_a, sina, cosa);
    float wx = cosa * w_2, wy = sina * w_2;
    float hx = -sina * h_2, hy = cosa * h_2;
    taps[1] = make_tap_yx(H, W, roi_y + wy + hy, roi_x + wx + hx);
    taps[2] = make_tap_yx(H, W, roi_y - wy + hy, roi_x - wx + hx);
    taps[3] = make_tap_yx(H, W, roi_y - wy - hy, roi_x - wx - hx);
    taps[4] = make_tap_yx(H, W, roi_y + wy - hy, roi_x + wx - hx);
  }
}

// Walk of the strict tile pairs (pi < pj) of a tiles x tiles grid for the kernels that give one workgroup a 4 x 4
// tile and its transpose.  A pair's out-of-tile taps fall in the tiles of the pairs (pi +- 1, pj) and (pi, pj +- 1);
// a row fetched for one of them is still in the XCD's L2 for the other only when the two workgroups start within
// a few microseconds of each other, i.e. a few dozen places apart in launch order.  Row-major over the triangle,
// (pi, pj +- 1) are up to `tiles` places apart and the earlier one's rows are gone.  Here: strips of `strip` rows
// of pj, column-major inside a strip, so that (pi, pj +- 1) are neighbours and (pi +- 1, pj) are `strip` apart;
// only a strip's first / last row has a far neighbour.  strip <= 0: the row-major walk.  All scalar arithmetic.
__device__ __forceinline__ void pair_walk(int tt, int tiles, int strip, int& pi, int& pj) {
  if (strip <= 0) strip = tiles;  // one strip: plain row-major over the triangle
  int s0 = 0, base = 0, hs, body;
  for (;;) {  // strip of rows [s0, s0 + hs): hs * s0 full columns' worth, then the hs x hs triangle at the diagonal
    hs = min(strip, tiles - s0);
    body = hs * s0;
    const int cnt = body + hs * (hs - 1) / 2;
    if (tt < base + cnt || s0 + hs >= tiles) break;
    base += cnt;
    s0 += hs;
  }
  const int u = tt - base;
  if (u < body) {
    pi = u / hs;
    pj = s0 + (u - pi * hs);
  } else {
    const int v = u - body;
    int b = 1;
    while ((b + 1) * b / 2 <= v) b++;
    pi = s0 + v - b * (b - 1) / 2;
    pj = s0 + b;
  }
}

}  // namespace
