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

}  // namespace
