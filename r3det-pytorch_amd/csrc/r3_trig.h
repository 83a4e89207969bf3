// r3_trig.h -- deterministic sin/cos for the rotated-box kernels (gfx950).
//
// The reference calls cosf/sinf (rbbox_geo_kernel.cu:147, feature_refine_kernel.cu:142) and
// (T)cos(double)/(T)sin(double) (box_iou_rotated_utils.h:61-63).  Device libm results are
// not reproducible on a host, so the kernels evaluate one fixed double-precision routine:
// Cody-Waite reduction by pi/2 and Taylor polynomials in Horner form, IEEE + - * only
// (the library is built with -ffp-contract=off).  Its error (< 1e-15) is far below half a
// float ulp, so the float it rounds to is the correctly rounded sine/cosine except in
// ~1e-8 of inputs; it is per-box work (n + m evaluations, never n * m).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void r3_sincos(float a, float& s_out, float& c_out) {
  if (a == 0.f) {  // (anchor grids: the polynomial below gives exactly (a, 1) for +-0; a wave of such boxes skips it)
    s_out = a;
    c_out = 1.f;
    return;
  }
  double x = (double)a;
  if (!(fabs(x) < 1.0e9)) {
    float q = a - a;  // NaN for inf/nan input, mirrors "no finite answer"
    s_out = q;
    c_out = q;
    return;
  }
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;
  const double pio2_lo = 6.07710050650619224932e-11;
  double k = rint(x * two_over_pi);
  double r = (x - k * pio2_hi) - k * pio2_lo;
  double z = r * r;
  double ps = -7.6471637318198164759e-13;
  ps = ps * z + 1.6059043836821614599e-10;
  ps = ps * z + -2.5052108385441718775e-08;
  ps = ps * z + 2.7557319223985890653e-06;
  ps = ps * z + -1.9841269841269841270e-04;
  ps = ps * z + 8.3333333333333332177e-03;
  ps = ps * z + -1.6666666666666665741e-01;
  double sr = r + r * (z * ps);
  double pc = 4.7794773323873852974e-14;
  pc = pc * z + -1.1470745597729724714e-11;
  pc = pc * z + 2.0876756987868098979e-09;
  pc = pc * z + -2.7557319223985888276e-07;
  pc = pc * z + 2.4801587301587301566e-05;
  pc = pc * z + -1.3888888888888889419e-03;
  pc = pc * z + 4.1666666666666664354e-02;
  pc = pc * z + -5.0000000000000000000e-01;
  double cr = 1.0 + z * pc;
  long long ki = (long long)k;
  int n = (int)(ki & 3);
  double s, c;
  if (n == 0) {
    s = sr;
    c = cr;
  } else if (n == 1) {
    s = cr;
    c = -sr;
  } else if (n == 2) {
    s = -sr;
    c = -cr;
  } else {
    s = -cr;
    c = sr;
  }
  s_out = (float)s;
  c_out = (float)c;
}
