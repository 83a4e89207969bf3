// r3_geom.h -- per-box records and pair geometry for the rotated IoU / NMS kernels.
//
// Design (not a translation of the reference kernels):
//   * everything that depends on ONE box (trig, half-extent products, vertices, area,
//     circumscribed radius) is computed once per box into a 12-float record
//     (n + m work instead of n * m; the reference recomputes sin/cos per pair,
//     rbbox_geo_kernel.cu:147, box_iou_rotated_utils.h:61-63);
//   * a pair first takes a circumscribed-circle test: when the circles are separated with
//     a safety margin both reference algorithms find no candidate point and return exactly
//     0 (v1: p_cnt < 3, rbbox_geo_kernel.cu:250,264-266; hull: num <= 2,
//     box_iou_rotated_utils.h:320-322), so the clipping is skipped with identical result;
//   * the surviving pairs run the reference arithmetic operation for operation (same
//     association order, IEEE ops, no FMA contraction) so the result is bit-identical to
//     the CPU restatement in oracle/ run in "twin" mode.
//
// geometry ids: 1 = v1 (rbbox_geo/rnms), 2 = v2 (mmcv/ml_nms_rotated), 3 = v3
// (box_iou_rotated/nms_rotated).
#pragma once
#include <hip/hip_runtime.h>

#include "r3_trig.h"

#define R3_REC 16  // floats per box record (64 B, four 16-B loads)

struct Pt {
  float x, y;
};
__device__ __forceinline__ float dotp(Pt a, Pt b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float crossp(Pt a, Pt b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ Pt subp(Pt a, Pt b) { return Pt{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ Pt addp(Pt a, Pt b) { return Pt{a.x + b.x, a.y + b.y}; }

// Box record.
//   v1  : f[0..7] = vertices (x0,y0,..,x3,y3)  f[8] = w*h
//   hull: f[0],f[1] = cx,cy  f[2] = sin/2*h  f[3] = cos/2*w  f[4] = cos/2*h  f[5] = sin/2*w
//         f[6] = w*h  f[7] = label (as float)
//   both: f[9],f[10] = cx,cy   f[11] = inflated circumscribed radius
//         f[12],f[13] = inflated half extents of the axis-aligned bounding box
struct BoxRec {
  float f[R3_REC];
};

// Radius of the circumscribed circle, inflated so that rounding in the vertex arithmetic
// (relative to the box size and to the magnitude of the coordinates) can never turn a
// circle-separated pair into one with candidate points.
__device__ __forceinline__ float r3_radius(float cx, float cy, float w, float h) {
  float r = 0.5f * sqrtf(w * w + h * h);
  return r * 1.001f + 2e-6f * (fabsf(cx) + fabsf(cy)) + 1e-6f;
}

template <int GEOM>
__device__ __forceinline__ void make_record(const float* __restrict__ b, float label, BoxRec& r) {
  float x = b[0], y = b[1], w = b[2], h = b[3], a = b[4];
  float s, c;
  r3_sincos(a, s, c);
  if (GEOM == 1) {
    // rbbox2points, rbbox_geo_kernel.cu:143-155
    float w_2 = w / 2, h_2 = h / 2;
    float wx = c * w_2, wy = s * w_2;
    float hx = -s * h_2, hy = c * h_2;
    r.f[0] = x + wx + hx; r.f[1] = y + wy + hy;
    r.f[2] = x - wx + hx; r.f[3] = y - wy + hy;
    r.f[4] = x - wx - hx; r.f[5] = y - wy - hy;
    r.f[6] = x + wx - hx; r.f[7] = y + wy - hy;
    r.f[8] = w * h;
  } else {
    // get_rotated_vertices, box_iou_rotated_utils.h:61-63 (products only; sums are per pair)
    float cosTheta2 = c * 0.5f;
    float sinTheta2 = s * 0.5f;
    r.f[0] = x; r.f[1] = y;
    r.f[2] = sinTheta2 * h;
    r.f[3] = cosTheta2 * w;
    r.f[4] = cosTheta2 * h;
    r.f[5] = sinTheta2 * w;
    r.f[6] = w * h;
    r.f[7] = label;
    r.f[8] = 0.f;
  }
  r.f[9] = x;
  r.f[10] = y;
  r.f[11] = r3_radius(x, y, w, h);
  // axis-aligned half extents |c|*w/2 + |s|*h/2, |s|*w/2 + |c|*h/2 with the same inflation
  float ac = fabsf(c), as = fabsf(s), aw = 0.5f * fabsf(w), ah = 0.5f * fabsf(h);
  float slack = 2e-6f * (fabsf(x) + fabsf(y)) + 1e-6f;
  r.f[12] = (ac * aw + as * ah) * 1.001f + slack;
  r.f[13] = (as * aw + ac * ah) * 1.001f + slack;
  r.f[14] = 0.f;
  r.f[15] = 0.f;
}

// Conservative disjointness test on two records: separated circumscribed circles OR
// separated axis-aligned bounding boxes (each inflated).  NaN / Inf => false.
__device__ __forceinline__ bool boxes_apart(const float* A, const float* B) {
  float dx = A[9] - B[9], dy = A[10] - B[10];
  float rr = A[11] + B[11];
  return (dx * dx + dy * dy > rr * rr) | (fabsf(dx) > A[12] + B[12]) | (fabsf(dy) > A[13] + B[13]);
}

// true when the pair certainly has no candidate intersection point (result exactly 0).
// NaN/Inf anywhere makes the comparison false, i.e. the slow path decides.
__device__ __forceinline__ bool circles_apart(float ax, float ay, float ar, float bx, float by,
                                              float br) {
  float dx = ax - bx, dy = ay - by;
  float rr = ar + br;
  return dx * dx + dy * dy > rr * rr;
}

// ----------------------------------------------------------------------------------------
// v1: vertex + segment algorithm (rbbox_geo_kernel.cu:88-268)
// ----------------------------------------------------------------------------------------
#define R3_V1_CAP 16

__device__ __forceinline__ void v1_push(Pt* u, int& cnt, Pt p) {
  if (cnt < R3_V1_CAP) u[cnt++] = p;
}

// vertex_in_rbbox (:157-175): vertices of `v` strictly inside the box with vertices `box`.
__device__ __forceinline__ void v1_vertex_in(const Pt* v, const Pt* box, Pt* u, int& cnt) {
  Pt s02 = addp(box[0], box[2]);
  Pt center = Pt{0.5f * s02.x, 0.5f * s02.y};
  Pt d10 = subp(box[1], box[0]);
  Pt w_vec = Pt{0.5f * d10.x, 0.5f * d10.y};
  Pt d21 = subp(box[2], box[1]);
  Pt h_vec = Pt{0.5f * d21.x, 0.5f * d21.y};
  float h2 = dotp(h_vec, h_vec);
  float w2 = dotp(w_vec, w_vec);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Pt pr = subp(v[i], center);
    if (fabsf(dotp(pr, h_vec)) < h2 && fabsf(dotp(pr, w_vec)) < w2) v1_push(u, cnt, v[i]);
  }
}

// LinSeg::InterSectWith (:94-140)
__device__ __forceinline__ void v1_segment(Pt a1, Pt a2, Pt b1, Pt b2, Pt* u, int& cnt) {
  Pt A = subp(a2, a1), B = subp(b2, b1), C = subp(a1, b1);
  if (C.x == 0 && C.y == 0) {
    v1_push(u, cnt, a1);
    return;
  }
  float D = -crossp(A, B);
  if (D != 0) {
    float s = crossp(C, B) / D;
    float t = -crossp(A, C) / D;
    if (0 <= s && s < 1 && 0 <= t && t < 1) v1_push(u, cnt, Pt{a1.x + s * A.x, a1.y + s * A.y});
    return;
  }
  if (crossp(A, C) != 0) return;
  float BdtC = dotp(B, C);
  float BdtB = dotp(B, B);
  float AdtnC = -dotp(A, C);
  float AdtA = dotp(A, A);
  if (BdtC >= 0 && BdtC < BdtB) v1_push(u, cnt, a1);
  if (AdtnC >= 0 && AdtnC < AdtA) v1_push(u, cnt, b1);
}

// operator< (:74-79)
__device__ __forceinline__ bool v1_less(Pt a, Pt b) {
  if ((a.x == 0 && a.y == 0) && (b.x != 0 || b.y != 0)) return true;
  return crossp(a, b) > 0;
}

// area (:193-228)
__device__ __forceinline__ float v1_area(Pt* dirty, int n_dirty) {
  const float numthres = (float)1e-2;
  Pt vs[R3_V1_CAP];
  vs[0] = Pt{0.f, 0.f};
  int n = 1;
  for (int i = 1; i < n_dirty; i++) {
    bool clean = true;
    Pt d = subp(dirty[i], dirty[0]);
    for (int j = 0; j < n; j++) {
      Pt df = subp(d, vs[j]);
      if (fabsf(df.x) < numthres && fabsf(df.y) < numthres) {
        clean = false;
        break;
      }
    }
    if (clean) vs[n++] = d;
  }
  for (int i = 1; i < n; i++) {
    vs[0] = vs[i];
    int j;
    for (j = i - 1; v1_less(vs[0], vs[j]); j--) vs[j + 1] = vs[j];
    vs[j + 1] = vs[0];
  }
  float a = 0;
  vs[0] = Pt{0.f, 0.f};
  for (int i = 1; i < n; i++) {
    int nx = (i + 1 == n) ? 0 : i + 1;
    a += crossp(vs[i], vs[nx]);
  }
  return a / 2;
}

// body of mat_iou_iof_kernel (:238-266) on two records
__device__ __noinline__ float v1_pair_slow(const BoxRec& A, const BoxRec& B, bool iof) {
  Pt v1[4], v2[4], u[R3_V1_CAP];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v1[i] = Pt{A.f[2 * i], A.f[2 * i + 1]};
    v2[i] = Pt{B.f[2 * i], B.f[2 * i + 1]};
  }
  int cnt = 0;
  v1_vertex_in(v1, v2, u, cnt);
  v1_vertex_in(v2, v1, u, cnt);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) v1_segment(v1[i], v1[(i + 1) & 3], v2[j], v2[(j + 1) & 3], u, cnt);
  if (cnt >= 3) {
    float s1 = A.f[8], s2 = B.f[8];
    float su = v1_area(u, cnt);
    su = (s1 < su) ? s1 : su;
    su = (s2 < su) ? s2 : su;
    su = (su < 0.f) ? 0.f : su;
    return iof ? su / s1 : su / (s1 + s2 - su);
  }
  return 0.f;
}

// ----------------------------------------------------------------------------------------
// hull algorithm (box_iou_rotated_utils.h:55-361; device branch of the sort :193-216)
// ----------------------------------------------------------------------------------------
template <bool V2>
__device__ __forceinline__ void hull_vertices(float x, float y, const BoxRec& r, Pt* pts) {
  float sh = r.f[2], cw = r.f[3], ch = r.f[4], sw = r.f[5];
  if (!V2) {  // box_iou_rotated/src/box_iou_rotated_utils.h:66-69
    pts[0].x = x + sh + cw;
    pts[0].y = y + ch - sw;
    pts[1].x = x - sh + cw;
    pts[1].y = y - ch - sw;
  } else {  // ml_nms_rotated/src/box_iou_rotated_utils.h:67-70
    pts[0].x = x - sh - cw;
    pts[0].y = y + ch - sw;
    pts[1].x = x + sh - cw;
    pts[1].y = y - ch - sw;
  }
  pts[2].x = 2 * x - pts[0].x;
  pts[2].y = 2 * y - pts[0].y;
  pts[3].x = 2 * x - pts[1].x;
  pts[3].y = 2 * y - pts[1].y;
}

template <bool V2>
__device__ __noinline__ float hull_pair_slow(const BoxRec& A, const BoxRec& B, bool iou_mode) {
  // single_box_iou_rotated, box_iou_rotated_utils.h:331-361
  float ax = A.f[0], ay = A.f[1], bx = B.f[0], by = B.f[1];
  double csx = (double)(ax + bx) / 2.0;
  double csy = (double)(ay + by) / 2.0;
  float x1 = (float)((double)ax - csx), y1 = (float)((double)ay - csy);
  float x2 = (float)((double)bx - csx), y2 = (float)((double)by - csy);
  float area1 = A.f[6], area2 = B.f[6];
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

  Pt pts1[4], pts2[4];
  hull_vertices<V2>(x1, y1, A, pts1);
  hull_vertices<V2>(x2, y2, B, pts2);

  // get_intersection_points, :76-155
  Pt inter[24];
  int num = 0;
  Pt vec1[4], vec2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    vec1[i] = subp(pts1[(i + 1) & 3], pts1[i]);
    vec2[i] = subp(pts2[(i + 1) & 3], pts2[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float det = crossp(vec2[j], vec1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      Pt vec12 = subp(pts2[j], pts1[i]);
      float t1 = crossp(vec2[j], vec12) / det;
      float t2 = crossp(vec1[i], vec12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f)
        inter[num++] = Pt{pts1[i].x + vec1[i].x * t1, pts1[i].y + vec1[i].y * t1};
    }
  }
  {
    Pt AB = vec2[0], DA = vec2[3];
    float ABdotAB = dotp(AB, AB), ADdotAD = dotp(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = subp(pts1[i], pts2[0]);
      float APdotAB = dotp(AP, AB);
      float APdotAD = -dotp(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD)
        inter[num++] = pts1[i];
    }
  }
  {
    Pt AB = vec1[0], DA = vec1[3];
    float ABdotAB = dotp(AB, AB), ADdotAD = dotp(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = subp(pts2[i], pts1[0]);
      float APdotAB = dotp(AP, AB);
      float APdotAD = -dotp(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD)
        inter[num++] = pts2[i];
    }
  }
  float intersection = 0.f;
  if (num > 2) {
    // convex_hull_graham(shift_to_zero = true), :157-289
    Pt q[24];
    float dist[24];
    int t = 0;
    for (int i = 1; i < num; i++)
      if (inter[i].y < inter[t].y || (inter[i].y == inter[t].y && inter[i].x < inter[t].x)) t = i;
    Pt start = inter[t];
    for (int i = 0; i < num; i++) q[i] = subp(inter[i], start);
    Pt tmp = q[0];
    q[0] = q[t];
    q[t] = tmp;
    for (int i = 0; i < num; i++) dist[i] = dotp(q[i], q[i]);
    for (int i = 1; i < num - 1; i++)
      for (int j = i + 1; j < num; j++) {
        float cp = crossp(q[i], q[j]);
        if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && dist[i] > dist[j])) {
          Pt qt = q[i]; q[i] = q[j]; q[j] = qt;
          float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
        }
      }
    int k;
    for (k = 1; k < num; k++)
      if ((double)dist[k] > 1e-8) break;
    int m;
    if (k == num) {
      m = 1;
    } else {
      q[1] = q[k];
      m = 2;
      for (int i = k + 1; i < num; i++) {
        while (m > 1) {
          Pt q1 = subp(q[i], q[m - 2]), q2 = subp(q[m - 1], q[m - 2]);
          bool pop = V2 ? (crossp(q1, q2) >= 0)            // ml utils.h:253
                        : (q1.x * q2.y >= q2.x * q1.y);    // utils.h:264
          if (pop) m--; else break;
        }
        q[m++] = q[i];
      }
    }
    // polygon_area, :291-303
    if (m > 2) {
      float area = 0;
      for (int i = 1; i < m - 1; i++) area += fabsf(crossp(subp(q[i], q[0]), subp(q[i + 1], q[0])));
      intersection = (float)((double)area / 2.0);
    }
  }
  return iou_mode ? intersection / (area1 + area2 - intersection) : intersection / area1;
}

// Full pair evaluation on two records.  LABEL: hull variants compare f[7] first
// (ml box_iou_rotated_utils.h:316-322).
template <int GEOM, bool LABEL>
__device__ __forceinline__ float pair_iou(const BoxRec& A, const BoxRec& B, bool iof) {
  if (GEOM != 1 && LABEL && A.f[7] != B.f[7]) return 0.f;
  if (circles_apart(A.f[9], A.f[10], A.f[11], B.f[9], B.f[10], B.f[11])) return 0.f;
  if (GEOM == 1) return v1_pair_slow(A, B, iof);
  if (GEOM == 2) return hull_pair_slow<true>(A, B, !iof);
  return hull_pair_slow<false>(A, B, !iof);
}
