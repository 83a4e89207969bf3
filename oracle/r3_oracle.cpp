// r3_oracle.cpp -- CPU restatement of the r3det custom-op hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under r3det-pytorch_amd/ may include,
// link, import or execute this file.  It is the checker used by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg, never the product.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/r3det/ops).  The text below is a fresh restatement in plain
// C++ (no torch); the reference's own CPU sources are compiled separately into
// oracle/_ref/ (see oracle/ref_harness/) and this file is pinned against them
// by tests/test_oracle_vs_ref.py and by the fixtures in tests/golden/.
//
// Pinning status:
//   IoU v1 / NMS v1 : pinned against rnms/src/rcpu/rnms_cpu.cpp (reference CPU)
//   IoU v3 / NMS v3 : pinned against box_iou_rotated/src + nms_rotated/src (CPU)
//   IoU v2 / NMS v2 : pinned against ml_nms_rotated/src/box_iou_rotated_utils.h
//                     (the in-tree statement of the mmcv convention); the
//                     arithmetic of mmcv.ops.{box_iou_rotated,nms_rotated}
//                     itself lives in mmcv-full 1.3.15..1.5.0 which is NOT
//                     under /root/reference  => "parity unpinned" for mmcv.
//   FR fwd/bwd      : the reference has only a CUDA implementation and no
//                     tests => "parity unpinned" by any reference fixture; the
//                     restatement follows fr/src/feature_refine_kernel.cu.
//   polygon_iou     : pinned against polygon_geo/src/polygon_geo_cpu.cpp (CPU)
//   convex_sort     : pinned against convex/src/convex_cpu.cpp (CPU) on inputs
//                     whose sort keys are distinct; for equal keys the reference
//                     inherits torch.argsort's unstable order, here: index order
//   poly_nms        : CUDA-only in the reference (poly_nms_cpu.cpp is a stub)
//                     => "parity unpinned"; follows nms_rotated/src/poly_nms_cuda.cu.
//
// Two switches make the oracle usable both as a reference-faithful CPU model
// and as the bit-exact twin of the HIP kernels:
//   trig mode  0 = libm (sinf/cosf, cos/sin as the reference CPU code calls)
//              1 = the deterministic double-precision sincos shared with the
//                  HIP kernels (csrc/r3_trig.h restates the same polynomial)
//   hull sort  0 = host branch (std::sort)          box_iou_rotated_utils.h:217-231
//              1 = device branch (exchange sort)    box_iou_rotated_utils.h:193-216
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

int g_trig_mode = 0;   // 0 libm, 1 deterministic twin
int g_hull_sort = 0;   // 0 host std::sort, 1 device exchange sort

// ---------------------------------------------------------------------------
// Deterministic sincos (twin of csrc/r3_trig.h).  Only IEEE + - * in double,
// no FMA (build with -ffp-contract=off), so CPU and gfx950 agree bit for bit.
// ---------------------------------------------------------------------------
void det_sincos(float a, float* s_out, float* c_out) {
  double x = (double)a;
  if (!(std::fabs(x) < 1.0e9)) {  // inf / nan / absurd: deterministic NaN
    float q = a - a;
    *s_out = q; *c_out = q;
    return;
  }
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11;   // pi/2 - pio2_hi
  double k = std::nearbyint(x * two_over_pi);
  double r = (x - k * pio2_hi) - k * pio2_lo;
  double z = r * r;
  double ps = -7.6471637318198164759e-13;              // -1/15!
  ps = ps * z + 1.6059043836821614599e-10;             //  1/13!
  ps = ps * z + -2.5052108385441718775e-08;            // -1/11!
  ps = ps * z + 2.7557319223985890653e-06;             //  1/9!
  ps = ps * z + -1.9841269841269841270e-04;            // -1/7!
  ps = ps * z + 8.3333333333333332177e-03;             //  1/5!
  ps = ps * z + -1.6666666666666665741e-01;            // -1/3!
  double sr = r + r * (z * ps);
  double pc = 4.7794773323873852974e-14;               //  1/16!
  pc = pc * z + -1.1470745597729724714e-11;            // -1/14!
  pc = pc * z + 2.0876756987868098979e-09;             //  1/12!
  pc = pc * z + -2.7557319223985888276e-07;            // -1/10!
  pc = pc * z + 2.4801587301587301566e-05;             //  1/8!
  pc = pc * z + -1.3888888888888889419e-03;            // -1/6!
  pc = pc * z + 4.1666666666666664354e-02;             //  1/4!
  pc = pc * z + -5.0000000000000000000e-01;            // -1/2!
  double cr = 1.0 + z * pc;
  long long ki = (long long)k;
  int n = (int)(ki & 3);
  double s, c;
  if (n == 0)      { s = sr;  c = cr;  }
  else if (n == 1) { s = cr;  c = -sr; }
  else if (n == 2) { s = -sr; c = -cr; }
  else             { s = -cr; c = sr;  }
  *s_out = (float)s;
  *c_out = (float)c;
}

// sinf/cosf as rbbox_geo_kernel.cu:147 / rnms_cpu.cpp calls them.
inline void trig_f32(float a, float* s, float* c) {
  if (g_trig_mode == 0) { *s = sinf(a); *c = cosf(a); }
  else det_sincos(a, s, c);
}
// (T)cos(double theta), (T)sin(double theta) as box_iou_rotated_utils.h:61-63.
inline void trig_f64_to_f32(float a, float* s, float* c) {
  if (g_trig_mode == 0) {
    double th = (double)a;
    *c = (float)std::cos(th);
    *s = (float)std::sin(th);
  } else det_sincos(a, s, c);
}

// ===========================================================================
// v1 geometry  (rbbox_geo/src/rbbox_geo_kernel.cu:43-268, token-identical CPU
// twin rnms/src/rcpu/rnms_cpu.cpp:11-221).
// ===========================================================================
struct P1 {
  float x, y;
};
inline float dot1(P1 a, P1 b) { return a.x * b.x + a.y * b.y; }            // :48-50
inline float cross1(P1 a, P1 b) { return a.x * b.y - b.x * a.y; }          // :51-53
inline P1 sub1(P1 a, P1 b) { return P1{a.x - b.x, a.y - b.y}; }            // :54-57
inline P1 add1(P1 a, P1 b) { return P1{a.x + b.x, a.y + b.y}; }            // :64-67
inline P1 mul1(float k, P1 p) { return P1{k * p.x, k * p.y}; }             // :82-86
// operator<  (:74-79): origin sorts first, otherwise counter-clockwise order.
inline bool less1(P1 a, P1 b) {
  if ((a.x == 0 && a.y == 0) && (b.x != 0 || b.y != 0)) return true;
  return cross1(a, b) > 0;
}

// rbbox2points (:143-155)
void v1_points(const float* rb, P1* vs) {
  float x = rb[0], y = rb[1], w_2 = rb[2] / 2, h_2 = rb[3] / 2, a = rb[4];
  float sina, cosa;
  trig_f32(a, &sina, &cosa);
  float wx = cosa * w_2, wy = sina * w_2;
  float hx = -sina * h_2, hy = cosa * h_2;
  vs[0] = P1{x + wx + hx, y + wy + hy};
  vs[1] = P1{x - wx + hx, y - wy + hy};
  vs[2] = P1{x - wx - hx, y - wy - hy};
  vs[3] = P1{x + wx - hx, y + wy - hy};
}

const int V1_CAP = 16;  // scratch capacity of the reference (:196,:241)

// vertex_in_rbbox (:157-175): vertices of v1 strictly inside box v2.
int v1_vertex_in(const P1* v1, const P1* v2, P1* ps, int room) {
  P1 center = mul1(0.5f, add1(v2[0], v2[2]));
  P1 w_vec = mul1(0.5f, sub1(v2[1], v2[0]));
  P1 h_vec = mul1(0.5f, sub1(v2[2], v2[1]));
  float h_vec_2 = dot1(h_vec, h_vec);
  float w_vec_2 = dot1(w_vec, w_vec);
  int cnt = 0;
  for (int i = 0; i < 4; i++) {
    P1 pr = sub1(v1[i], center);
    if (std::abs(dot1(pr, h_vec)) < h_vec_2 && std::abs(dot1(pr, w_vec)) < w_vec_2) {
      if (cnt < room) ps[cnt] = v1[i];
      cnt++;
    }
  }
  return cnt;
}

// LinSeg::InterSectWith (:94-140)
int v1_seg(P1 a1, P1 a2, P1 b1, P1 b2, P1* ps) {
  P1 A = sub1(a2, a1), B = sub1(b2, b1), C = sub1(a1, b1);
  if (C.x == 0 && C.y == 0) {
    ps[0] = a1;
    return 1;
  }
  float D = -cross1(A, B);
  if (D != 0) {
    float s = cross1(C, B) / D;
    float t = -cross1(A, C) / D;
    if (0 <= s && s < 1 && 0 <= t && t < 1) {
      ps[0] = add1(a1, mul1(s, A));
      return 1;
    }
    return 0;
  }
  if (cross1(A, C) != 0) return 0;
  int cnt = 0;
  float BdtC = dot1(B, C);
  float BdtB = dot1(B, B);
  float AdtnC = -dot1(A, C);
  float AdtA = dot1(A, A);
  if (BdtC >= 0 && BdtC < BdtB) ps[cnt++] = a1;
  if (AdtnC >= 0 && AdtnC < AdtA) ps[cnt++] = b1;
  return cnt;
}

// rbbox_border_intsec (:177-191): 4x4 ordered edge pairs.
int v1_border(const P1* v1, const P1* v2, P1* ps, int room) {
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      P1 tmp[2];
      int k = v1_seg(v1[i], v1[(i + 1) & 3], v2[j], v2[(j + 1) & 3], tmp);
      for (int q = 0; q < k; q++) {
        if (cnt < room) ps[cnt] = tmp[q];
        cnt++;
      }
    }
  return cnt;
}

// area (:193-228): translate to first point, 1e-2 de-dup, insertion sort
// around the origin with slot 0 as sentinel, shoelace.
float v1_area(P1* dirty, int n_dirty) {
  const float numthres = (float)1e-2;
  P1 vs[V1_CAP];
  vs[0] = P1{0, 0};
  int n = 1;
  for (int i = 1; i < n_dirty; i++) {
    bool clean = true;
    dirty[i] = sub1(dirty[i], dirty[0]);
    for (int j = 0; j < n; j++) {
      P1 d = sub1(dirty[i], vs[j]);
      if (std::abs(d.x) < numthres && std::abs(d.y) < numthres) {
        clean = false;
        break;
      }
    }
    if (clean) vs[n++] = dirty[i];
  }
  for (int i = 1; i < n; i++) {
    vs[0] = vs[i];
    int j;
    for (j = i - 1; less1(vs[0], vs[j]); j--) vs[j + 1] = vs[j];
    vs[j + 1] = vs[0];
  }
  float a = 0;
  vs[0] = P1{0, 0};
  for (int i = 1; i < n; i++) a += cross1(vs[i], vs[(i + 1) % n]);
  return a / 2;
}

// body of mat_iou_iof_kernel (:238-266) == nmsr_cpu_kernel's pair block
// (rnms_cpu.cpp:254-276).  Points beyond the 16-slot scratch are dropped (the
// reference would write out of bounds there).
float v1_pair(const float* rb1, const float* rb2, bool iof) {
  P1 v1[4], v2[4], u[V1_CAP];
  v1_points(rb1, v1);
  v1_points(rb2, v2);
  int cnt = 0;
  cnt += v1_vertex_in(v1, v2, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  cnt += v1_vertex_in(v2, v1, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  cnt += v1_border(v1, v2, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  if (cnt >= 3) {
    float s1 = rb1[2] * rb1[3];
    float s2 = rb2[2] * rb2[3];
    float su = v1_area(u, cnt);
    su = std::min(su, s1);
    su = std::min(su, s2);
    su = std::max(su, 0.0f);
    return iof ? su / s1 : su / (s1 + s2 - su);
  }
  return 0.0f;
}

// ===========================================================================
// Hull geometry: v3 = box_iou_rotated/src/box_iou_rotated_utils.h (and its
// copy nms_rotated/src/box_iou_rotated_utils.h); v2 = ml_nms_rotated/src/
// box_iou_rotated_utils.h (standard vertex sign, label guard, fused scan test).
// ===========================================================================
struct P3 {
  float x, y;
};
inline P3 sub3(P3 a, P3 b) { return P3{a.x - b.x, a.y - b.y}; }
inline P3 add3(P3 a, P3 b) { return P3{a.x + b.x, a.y + b.y}; }
inline P3 mul3(P3 a, float k) { return P3{a.x * k, a.y * k}; }
inline float dot3(P3 a, P3 b) { return a.x * b.x + a.y * b.y; }             // :43-46
inline float cross3(P3 a, P3 b) { return a.x * b.y - b.x * a.y; }           // :49-53

struct RBox {
  float x, y, w, h, a;
};

// get_rotated_vertices: v3 signs utils.h:55-74, v2 signs ml utils.h:56-76.
void hull_vertices(const RBox& b, P3* pts, bool v2) {
  float st, ct;
  trig_f64_to_f32(b.a, &st, &ct);
  float cosTheta2 = ct * 0.5f;
  float sinTheta2 = st * 0.5f;
  if (!v2) {
    pts[0].x = b.x + sinTheta2 * b.h + cosTheta2 * b.w;
    pts[0].y = b.y + cosTheta2 * b.h - sinTheta2 * b.w;
    pts[1].x = b.x - sinTheta2 * b.h + cosTheta2 * b.w;
    pts[1].y = b.y - cosTheta2 * b.h - sinTheta2 * b.w;
  } else {
    pts[0].x = b.x - sinTheta2 * b.h - cosTheta2 * b.w;
    pts[0].y = b.y + cosTheta2 * b.h - sinTheta2 * b.w;
    pts[1].x = b.x + sinTheta2 * b.h - cosTheta2 * b.w;
    pts[1].y = b.y - cosTheta2 * b.h - sinTheta2 * b.w;
  }
  pts[2].x = 2 * b.x - pts[0].x;
  pts[2].y = 2 * b.y - pts[0].y;
  pts[3].x = 2 * b.x - pts[1].x;
  pts[3].y = 2 * b.y - pts[1].y;
}

// get_intersection_points (utils.h:76-155)
int hull_candidates(const P3* pts1, const P3* pts2, P3* out) {
  P3 vec1[4], vec2[4];
  for (int i = 0; i < 4; i++) {
    vec1[i] = sub3(pts1[(i + 1) % 4], pts1[i]);
    vec2[i] = sub3(pts2[(i + 1) % 4], pts2[i]);
  }
  int num = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      float det = cross3(vec2[j], vec1[i]);
      if (std::fabs((double)det) <= 1e-14) continue;
      P3 vec12 = sub3(pts2[j], pts1[i]);
      float t1 = cross3(vec2[j], vec12) / det;
      float t2 = cross3(vec1[i], vec12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f)
        out[num++] = add3(pts1[i], mul3(vec1[i], t1));
    }
  }
  {
    P3 AB = vec2[0], DA = vec2[3];
    float ABdotAB = dot3(AB, AB), ADdotAD = dot3(DA, DA);
    for (int i = 0; i < 4; i++) {
      P3 AP = sub3(pts1[i], pts2[0]);
      float APdotAB = dot3(AP, AB);
      float APdotAD = -dot3(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD)
        out[num++] = pts1[i];
    }
  }
  {
    P3 AB = vec1[0], DA = vec1[3];
    float ABdotAB = dot3(AB, AB), ADdotAD = dot3(DA, DA);
    for (int i = 0; i < 4; i++) {
      P3 AP = sub3(pts2[i], pts1[0]);
      float APdotAB = dot3(AP, AB);
      float APdotAD = -dot3(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD)
        out[num++] = pts2[i];
    }
  }
  return num;
}

// convex_hull_graham with shift_to_zero=true (utils.h:157-289; v2 variant
// ml utils.h:158-270).  Returns the number of hull points left in q.
int hull_graham(const P3* p, int num_in, P3* q, bool v2) {
  int t = 0;
  for (int i = 1; i < num_in; i++)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  P3 start = p[t];
  for (int i = 0; i < num_in; i++) q[i] = sub3(p[i], start);
  P3 tmp = q[0];
  q[0] = q[t];
  q[t] = tmp;

  float dist[24];
  if (g_hull_sort == 1) {
    // device branch: dist before the sort, sorted together (utils.h:196-216)
    for (int i = 0; i < num_in; i++) dist[i] = dot3(q[i], q[i]);
    for (int i = 1; i < num_in - 1; i++)
      for (int j = i + 1; j < num_in; j++) {
        float cp = cross3(q[i], q[j]);
        if (((double)cp < -1e-6) || (std::fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
          P3 qt = q[i]; q[i] = q[j]; q[j] = qt;
          float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
        }
      }
  } else {
    // host branch (utils.h:219-231).  The ml/v2 copy computes dist BEFORE the
    // sort and never refreshes it (ml utils.h:193-196,216-226).
    if (v2)
      for (int i = 0; i < num_in; i++) dist[i] = dot3(q[i], q[i]);
    std::sort(q + 1, q + num_in, [](const P3& A, const P3& B) -> bool {
      float temp = cross3(A, B);
      if (std::fabs((double)temp) < 1e-6) return dot3(A, A) < dot3(B, B);
      return temp > 0;
    });
    if (!v2)
      for (int i = 0; i < num_in; i++) dist[i] = dot3(q[i], q[i]);
  }

  int k;
  for (k = 1; k < num_in; k++)
    if ((double)dist[k] > 1e-8) break;
  if (k == num_in) {
    q[0] = p[t];
    return 1;
  }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num_in; i++) {
    while (m > 1) {
      P3 q1 = sub3(q[i], q[m - 2]), q2 = sub3(q[m - 1], q[m - 2]);
      bool pop;
      if (!v2) pop = (q1.x * q2.y >= q2.x * q1.y);        // utils.h:264
      else pop = (cross3(q1, q2) >= 0);                    // ml utils.h:253
      if (pop) m--; else break;
    }
    q[m++] = q[i];
  }
  return m;
}

// polygon_area (utils.h:291-303)
float hull_area(const P3* q, int m) {
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; i++)
    area += std::fabs(cross3(sub3(q[i], q[0]), sub3(q[i + 1], q[0])));
  return (float)(area / 2.0);
}

// rotated_boxes_intersection (utils.h:305-328)
float hull_intersection(const RBox& b1, const RBox& b2, bool v2) {
  P3 inter[24], ordered[24];
  P3 pts1[4], pts2[4];
  hull_vertices(b1, pts1, v2);
  hull_vertices(b2, pts2, v2);
  int num = hull_candidates(pts1, pts2, inter);
  if (num <= 2) return 0.0f;
  int m = hull_graham(inter, num, ordered, v2);
  return hull_area(ordered, m);
}

// single_box_iou_rotated: v3 utils.h:331-361 (iou_or_iof: true = IoU);
// v2 ml utils.h:314-347 (label guard on raw[5] when with_label).
float hull_pair(const float* r1, const float* r2, bool v2, bool iou_mode, bool with_label) {
  if (with_label && r1[5] != r2[5]) return 0.0f;
  RBox b1, b2;
  double csx = (double)(r1[0] + r2[0]) / 2.0;
  double csy = (double)(r1[1] + r2[1]) / 2.0;
  b1.x = (float)((double)r1[0] - csx);
  b1.y = (float)((double)r1[1] - csy);
  b1.w = r1[2]; b1.h = r1[3]; b1.a = r1[4];
  b2.x = (float)((double)r2[0] - csx);
  b2.y = (float)((double)r2[1] - csy);
  b2.w = r2[2]; b2.h = r2[3]; b2.a = r2[4];
  float area1 = b1.w * b1.h;
  float area2 = b2.w * b2.h;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;
  float inter = hull_intersection(b1, b2, v2);
  return iou_mode ? inter / (area1 + area2 - inter) : inter / area1;
}

// ===========================================================================
// Feature refinement (fr/src/feature_refine_kernel.cu)
// ===========================================================================
// bilinear_interpolate (:16-65)
float fr_bilinear(const float* plane, int height, int width, float y, float x) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) return 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
  else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
  else x_high = x_low + 1;
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = 1. - ly;
  float hx = 1. - lx;
  float lt = plane[y_low * width + x_low];
  float rt = plane[y_low * width + x_high];
  float lb = plane[y_high * width + x_low];
  float rb = plane[y_high * width + x_high];
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return (w1 * lt + w2 * rt + w3 * lb + w4 * rb);
}

// bilinear_interpolate_gradient (:67-110)
void fr_bilinear_grad(int height, int width, float y, float x, float* w, int* xl, int* xh,
                      int* yl, int* yh) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    w[0] = w[1] = w[2] = w[3] = 0.;
    *xl = *xh = *yl = *yh = -1;
    return;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
  else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
  else x_high = x_low + 1;
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = 1. - ly;
  float hx = 1. - lx;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
  *xl = x_low; *xh = x_high; *yl = y_low; *yh = y_high;
}

// sample points of one position (:125-151): note roi_y <- box[0], roi_x <- box[1].
void fr_points(const float* box, float scale, int points, float* px, float* py) {
  float roi_y = box[0] * scale;
  float roi_x = box[1] * scale;
  px[0] = roi_x; py[0] = roi_y;
  for (int i = 1; i < 5; i++) { px[i] = 0; py[i] = 0; }
  if (points > 1) {
    float roi_w = box[2] * scale;
    float roi_h = box[3] * scale;
    float roi_a = box[4];
    float w_2 = roi_w / 2, h_2 = roi_h / 2;
    float sina, cosa;
    trig_f32(roi_a, &sina, &cosa);
    float wx = cosa * w_2, wy = sina * w_2;
    float hx = -sina * h_2, hy = cosa * h_2;
    px[1] = roi_x + wx + hx; py[1] = roi_y + wy + hy;
    px[2] = roi_x - wx + hx; py[2] = roi_y - wy + hy;
    px[3] = roi_x - wx - hx; py[3] = roi_y - wy - hy;
    px[4] = roi_x + wx - hx; py[4] = roi_y + wy - hy;
  }
}

// ===========================================================================
// polygon_iou (polygon_geo/src/polygon_geo_cpu.cpp): the v1 machinery on general
// 4-point polygons.  Pinned by oracle/_ref (the reference file compiles verbatim).
// ===========================================================================
// polygon2points (:137-158): vertices 2 and 3 insertion-sorted around vertex 0.
void poly_points(const float* poly, P1* vs) {
  for (int i = 0; i < 4; i++) vs[i] = P1{poly[2 * i], poly[2 * i + 1]};
  for (int i = 2; i < 4; i++) {
    P1 pt = vs[i];
    int j;
    for (j = i - 1; less1(sub1(pt, vs[0]), sub1(vs[j], vs[0])); j--) vs[j + 1] = vs[j];
    vs[j + 1] = pt;
  }
}

// vertex_in_polygon (:160-184): v1[i] is kept unless some edge of v2 has it on its right.
int poly_vertex_in(const P1* v1, const P1* v2, P1* ps, int room) {
  int cnt = 0;
  for (int i = 0; i < 4; i++) {
    bool inside = true;
    for (int j = 0; j < 4; j++) {
      P1 pr = sub1(v1[i], v2[j]);
      P1 pb = sub1(v2[(j + 1) % 4], v2[j]);
      if (less1(pr, pb)) {
        inside = false;
        break;
      }
    }
    if (inside) {
      if (cnt < room) ps[cnt] = v1[i];
      cnt++;
    }
  }
  return cnt;
}

// polygon_iou_kernel pair body (:241-266); candidate points beyond the reference's 16-slot
// scratch are dropped (it would write out of bounds there), as in v1_pair.
float poly_pair(const float* a, const float* b) {
  P1 v1[4], v2[4], u[V1_CAP];
  poly_points(a, v1);
  poly_points(b, v2);
  int cnt = 0;
  cnt += poly_vertex_in(v1, v2, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  cnt += poly_vertex_in(v2, v1, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  cnt += v1_border(v1, v2, u + cnt, V1_CAP - cnt);
  if (cnt > V1_CAP) cnt = V1_CAP;
  if (cnt >= 3) {
    float s1 = v1_area(v1, 4);
    float s2 = v1_area(v2, 4);
    float su = v1_area(u, cnt);
    su = std::min(su, s1);
    su = std::min(su, s2);
    su = std::max(su, 0.0f);
    return su / (s1 + s2 - su);
  }
  return 0.0f;
}

// ===========================================================================
// poly_nms IoU (nms_rotated/src/poly_nms_cuda.cu:21-140): signed triangle-fan clipping.
// CUDA-only in the reference (poly_nms_cpu.cpp is a stub): PARITY UNPINNED, restated from
// the source text.  float arithmetic, eps compared in double as the source does.
// ===========================================================================
struct F2 {
  float x, y;
};
const double PN_EPS = 1E-8;
inline int pn_sig(float d) { return (d > PN_EPS) - (d < -PN_EPS); }
inline bool pn_eq(F2 a, F2 b) { return pn_sig(a.x - b.x) == 0 && pn_sig(a.y - b.y) == 0; }
inline float pn_cross(F2 o, F2 a, F2 b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
inline float pn_area(F2* ps, int n) {
  ps[n] = ps[0];
  float res = 0;
  for (int i = 0; i < n; i++) res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return res / 2.0;
}
inline int pn_line_cross(F2 a, F2 b, F2 c, F2 d, F2& p) {
  float s1 = pn_cross(a, b, c), s2 = pn_cross(a, b, d);
  if (pn_sig(s1) == 0 && pn_sig(s2) == 0) return 2;
  if (pn_sig(s2 - s1) == 0) return 0;
  p.x = (c.x * s2 - d.x * s1) / (s2 - s1);
  p.y = (c.y * s2 - d.y * s1) / (s2 - s1);
  return 1;
}
inline void pn_cut(F2* p, int& n, F2 a, F2 b, F2* pp) {
  int m = 0;
  p[n] = p[0];
  for (int i = 0; i < n; i++) {
    if (pn_sig(pn_cross(a, b, p[i])) > 0) pp[m++] = p[i];
    if (pn_sig(pn_cross(a, b, p[i])) != pn_sig(pn_cross(a, b, p[i + 1]))) pn_line_cross(a, b, p[i], p[i + 1], pp[m++]);
  }
  n = 0;
  for (int i = 0; i < m; i++)
    if (!i || !pn_eq(pp[i], pp[i - 1])) p[n++] = pp[i];
  while (n > 1 && pn_eq(p[n - 1], p[0])) n--;
}
inline float pn_tri(F2 a, F2 b, F2 c, F2 d) {
  F2 o{0, 0};
  int s1 = pn_sig(pn_cross(o, a, b)), s2 = pn_sig(pn_cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) std::swap(a, b);
  if (s2 == -1) std::swap(c, d);
  F2 p[10] = {o, a, b};
  int n = 3;
  F2 pp[10] = {};  // (the source leaves it uninitialised; a skipped lineCross then keeps garbage)
  pn_cut(p, n, o, c, pp);
  pn_cut(p, n, c, d, pp);
  pn_cut(p, n, d, o, pp);
  float res = std::fabs(pn_area(p, n));
  if (s1 * s2 == -1) res = -res;
  return res;
}
float pn_iou(const float* p, const float* q) {
  F2 ps1[10], ps2[10];
  for (int i = 0; i < 4; i++) {
    ps1[i] = F2{p[2 * i], p[2 * i + 1]};
    ps2[i] = F2{q[2 * i], q[2 * i + 1]};
  }
  if (pn_area(ps1, 4) < 0) std::reverse(ps1, ps1 + 4);
  if (pn_area(ps2, 4) < 0) std::reverse(ps2, ps2 + 4);
  ps1[4] = ps1[0];
  ps2[4] = ps2[0];
  float inter = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) inter += pn_tri(ps1[i], ps1[i + 1], ps2[j], ps2[j + 1]);
  float uni = std::fabs(pn_area(ps1, 4)) + std::fabs(pn_area(ps2, 4)) - inter;
  return uni == 0 ? (inter + 1) / (uni + 1) : inter / uni;
}

inline float pair_iou(int geom, const float* a, const float* b, bool iof, bool with_label) {
  if (geom == 1) return v1_pair(a, b, iof);
  return hull_pair(a, b, geom == 2, !iof, with_label);
}

}  // namespace

extern "C" {

void orc_set_trig_mode(int m) { g_trig_mode = m; }
void orc_set_hull_sort(int m) { g_hull_sort = m; }
int orc_get_trig_mode() { return g_trig_mode; }
int orc_get_hull_sort() { return g_hull_sort; }
int orc_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_sincos(const float* a, int n, float* s, float* c) {
  for (int i = 0; i < n; i++) det_sincos(a[i], s + i, c + i);
}

// geom: 1 = v1 (rbbox_geo), 2 = v2 (mmcv convention), 3 = v3 (box_iou_rotated).
// mat: mat_iou_iof_kernel rbbox_geo_kernel.cu:231-268 / box_iou_rotated_cpu.cpp:7-21.
// stride = floats per row of b1/b2 (5, or 6 when a score/label column follows).
void orc_iou_mat(int geom, int iof, const float* b1, int n1, int s1, const float* b2, int n2,
                 int s2, float* out, int threads) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++)
      out[(size_t)i * n2 + j] = pair_iou(geom, b1 + (size_t)i * s1, b2 + (size_t)j * s2, iof != 0, false);
}

// vec: vec_iou_iof_kernel rbbox_geo_kernel.cu:271-309 (modulo broadcast).
void orc_iou_vec(int geom, int iof, const float* b1, int n1, int s1, const float* b2, int n2,
                 int s2, float* out) {
  int n = n1 > n2 ? n1 : n2;
  for (int i = 0; i < n; i++)
    out[i] = pair_iou(geom, b1 + (size_t)(i % n1) * s1, b2 + (size_t)(i % n2) * s2, iof != 0, false);
}

// Greedy NMS over score-sorted boxes.
//   rnms_cpu.cpp:223-282 (v1, dets n x 6, suppress on >=, result ascending),
//   nms_rotated_cpu.cpp:9-61 (v3), ml nms_rotated_cpu.cpp:7-58 (v2, label in col 5).
// strict != 0 selects the CUDA comparison ">" (rnms_kernel.cu:260,
// nms_rotated_cuda.cu:60-61).  keep_out receives original indices in score
// order (sorted ascending afterwards when ascending != 0).  Returns the count.
// Sort = stable descending, as torch's CPU sort behaves.
int orc_nms(int geom, const float* boxes, int stride, const float* scores, int n, float thr,
            int strict, int with_label, int ascending, int64_t* keep_out) {
  if (n == 0) return 0;
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
  std::vector<uint8_t> sup(n, 0);
  int cnt = 0;
  for (int _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep_out[cnt++] = i;
    for (int _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      float ov = pair_iou(geom, boxes + i * stride, boxes + j * stride, false, with_label != 0);
      if (strict ? (ov > thr) : (ov >= thr)) sup[j] = 1;
    }
  }
  if (ascending) std::sort(keep_out, keep_out + cnt);
  return cnt;
}

// polygon_iou (polygon_geo_cpu.cpp:231-287): (na, 8) x (nb, 8) -> (na, nb)
void orc_polygon_iou(const float* a, int na, const float* b, int nb, float* out) {
  for (int i = 0; i < na; i++)
    for (int j = 0; j < nb; j++) out[(size_t)i * nb + j] = poly_pair(a + (size_t)i * 8, b + (size_t)j * 8);
}

// devPolyIoU as a matrix (test aid) and poly_nms_cuda (poly_nms_cuda.cu:142-261): dets (n, 9) =
// 8 coordinates + score; stable descending score sort; suppress on IoU > thr; keep in score order.
void orc_poly_iou_mat(const float* a, int na, int sa, const float* b, int nb, int sb, float* out) {
  for (int i = 0; i < na; i++)
    for (int j = 0; j < nb; j++) out[(size_t)i * nb + j] = pn_iou(a + (size_t)i * sa, b + (size_t)j * sb);
}

int orc_poly_nms(const float* dets9, int n, float thr, int64_t* keep_out) {
  if (n == 0) return 0;
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return dets9[a * 9 + 8] > dets9[b * 9 + 8]; });
  std::vector<uint8_t> sup(n, 0);
  int cnt = 0;
  for (int _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep_out[cnt++] = i;
    for (int _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (!sup[j] && pn_iou(dets9 + i * 9, dets9 + j * 9) > thr) sup[j] = 1;
    }
  }
  return cnt;
}

// convex_sort (convex/src/convex_cpu.cpp:8-90): pts (B, P, 2), masks (B, P) as 0/1 floats ->
// (B, P + circular) int64 padded with -1.  The tensor prologue (:19-31) in fp32: start = first
// argmin of the masked y; key = (x - sx) / sqrt((x - sx)^2 + (y - sy)^2 + 1e-6); stable
// descending order of the keys; then the scan (:42-86).
void orc_convex_sort(const float* pts, const float* masks, int B, int P, int circular, int64_t* out) {
  const int isz = circular ? P + 1 : P;
  const float INF_ = 10000000.f, EPS_ = 0.000001f;
  std::vector<float> key(P);
  std::vector<int64_t> order(P);
  for (int b = 0; b < B; b++) {
    const float* p = pts + (size_t)b * P * 2;
    const float* m = masks + (size_t)b * P;
    int64_t* ci = out + (size_t)b * isz;
    for (int k = 0; k < isz; k++) ci[k] = -1;
    if (P == 0) continue;
    int64_t start = 0;
    float best = 0;
    for (int k = 0; k < P; k++) {
      float my = m[k] * p[2 * k + 1] + (1 - m[k]) * INF_;
      if (k == 0 || my < best) {
        best = my;
        start = k;
      }
    }
    const float sx = p[2 * start], sy = p[2 * start + 1];
    for (int k = 0; k < P; k++) {
      float dx = p[2 * k] - sx, dy = p[2 * k + 1] - sy;
      key[k] = dx / std::sqrt(dx * dx + dy * dy + EPS_);
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return key[a] > key[c]; });
    ci[0] = start;
    int64_t c_i = 0;
    for (int _j = 0; _j < P; _j++) {
      const int64_t j = order[_j];
      if (j == start) continue;
      if (m[j] < 0.5) continue;
      const float x0 = p[2 * j], y0 = p[2 * j + 1];
      float x1 = p[2 * ci[c_i]], y1 = p[2 * ci[c_i] + 1];
      const float d = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
      if (d < 0.000001) continue;  // (the source compares with the double literal)
      if (c_i < 2) {
        ci[++c_i] = j;
      } else {
        float x2 = p[2 * ci[c_i - 1]], y2 = p[2 * ci[c_i - 1] + 1];
        while (true) {
          const float t = (x1 - x2) * (y0 - y2) - (y1 - y2) * (x0 - x2);
          if (t >= 0) {
            ci[++c_i] = j;
            break;
          }
          if (c_i <= 1) {
            ci[c_i] = j;
            break;
          }
          c_i--;
          x1 = p[2 * ci[c_i]];
          y1 = p[2 * ci[c_i] + 1];
          x2 = p[2 * ci[c_i - 1]];
          y2 = p[2 * ci[c_i - 1] + 1];
        }
      }
    }
    if (circular) ci[++c_i] = ci[0];
  }
}

// feature_refine_forward_kernel (feature_refine_kernel.cu:112-163)
void orc_fr_forward(const float* feat, const float* boxes, int N, int C, int H, int W, float scale,
                    int points, float* out, int threads) {
#pragma omp parallel for collapse(2) num_threads(threads > 0 ? threads : 1)
  for (int n = 0; n < N; n++)
    for (int c = 0; c < C; c++) {
      const float* plane = feat + ((size_t)n * C + c) * H * W;
      float* oplane = out + ((size_t)n * C + c) * H * W;
      for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
          float px[5], py[5];
          fr_points(boxes + (((size_t)n * H + h) * W + w) * 5, scale, points, px, py);
          float v = plane[h * W + w];
          for (int i = 0; i < points; i++) v += fr_bilinear(plane, H, W, py[i], px[i]);
          oplane[h * W + w] = v;
        }
    }
}

// feature_refine_backward_kernel (:165-230).  Accumulates into bottom_grad
// (caller zero-fills, feature_refine_module.py:36).  Serial order = index order.
void orc_fr_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W,
                     float scale, int points, float* bottom_grad) {
  for (int n = 0; n < N; n++)
    for (int c = 0; c < C; c++) {
      const float* tplane = top_grad + ((size_t)n * C + c) * H * W;
      float* bplane = bottom_grad + ((size_t)n * C + c) * H * W;
      for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
          float px[5], py[5];
          fr_points(boxes + (((size_t)n * H + h) * W + w) * 5, scale, points, px, py);
          float g = tplane[h * W + w];
          bplane[h * W + w] += g;
          for (int i = 0; i < points; i++) {
            float wt[4];
            int xl, xh, yl, yh;
            fr_bilinear_grad(H, W, py[i], px[i], wt, &xl, &xh, &yl, &yh);
            float g1 = g * wt[0], g2 = g * wt[1], g3 = g * wt[2], g4 = g * wt[3];
            if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
              bplane[yl * W + xl] += g1;
              bplane[yl * W + xh] += g2;
              bplane[yh * W + xl] += g3;
              bplane[yh * W + xh] += g4;
            }
          }
        }
    }
}

}  // extern "C"
