// r3_geom_lds.h -- pair geometry for the COMPACTED slow path: every lane clips a different
// pair, so the candidate-point list (<= 16 points v1, <= 24 points hull) is indexed
// dynamically per lane.  It lives in LDS as one float2 array per lane, laid out
// [slot][lane] (slot stride = workgroup size): consecutive lanes touch consecutive 8-byte
// words, i.e. conflict-free ds_read_b64 / ds_write_b64, and nothing spills to scratch.
//
// Arithmetic is the reference's, operation for operation (see r3_geom.h); three things are
// restructured without changing any result:
//   * the half-open / closed unit-interval tests on s = num / D are decided from the signs
//     of num and D (exact for correctly rounded division, see unit_halfopen()), so a
//     division is only executed for accepted intersections (<= 8 instead of 32 per pair);
//   * v1's de-dup + insertion sort (rbbox_geo_kernel.cu:193-228) and the hull's Graham scan
//     (box_iou_rotated_utils.h:157-289) run in place in the one point array;
//   * the hull's dist[] (squared length, swapped along with the points in the device-branch
//     sort :196-216) is a pure function of the point and is recomputed on demand.
#pragma once
#include "r3_geom.h"

// (0 <= fl(num / D) && fl(num / D) < 1) for D != 0, without dividing.
// Proof sketch (round-to-nearest division, D > 0): fl(q) < 1 <=> num < D because the largest
// float below D gives q <= 1 - 2^-24 = pred(1); 0 <= fl(q) <=> num >= 0 unless q underflows
// to -0.  The underflow / infinite-D corner falls back to the real division.
__device__ __forceinline__ bool unit_halfopen(float num, float D) {
  bool weird = !(fabsf(D) < 3.0e38f) | ((num != 0) & (fabsf(num) < fabsf(D) * 1.0e-30f));
  if (weird) {
    float q = num / D;
    return 0 <= q && q < 1;
  }
  return (D > 0) ? (num >= 0 && num < D) : (num <= 0 && num > D);
}
// (fl(num / D) >= 0 && fl(num / D) <= 1) for |D| > 1e-14
__device__ __forceinline__ bool unit_closed(float num, float D) {
  bool weird = !(fabsf(D) < 3.0e38f) | ((num != 0) & (fabsf(num) < fabsf(D) * 1.0e-30f));
  if (weird) {
    float q = num / D;
    return q >= 0.0f && q <= 1.0f;
  }
  return (D > 0) ? (num >= 0 && num <= D) : (num <= 0 && num >= D);
}

template <int STRIDE>
struct LanePts {
  float2* base;  // already offset by the lane
  __device__ __forceinline__ Pt get(int s) const {
    float2 v = base[s * STRIDE];
    return Pt{v.x, v.y};
  }
  __device__ __forceinline__ void set(int s, Pt p) const { base[s * STRIDE] = make_float2(p.x, p.y); }
};

// ----------------------------------------------------------------------------------------
// v1
// ----------------------------------------------------------------------------------------
// CAP = R3_V1_CAP: the reference's 16-point scratch (further candidates are dropped, r3_geom.h).  CAP = 8: the
// short form -- 8 slots per lane, half the LDS, twice the resident waves; a 9th candidate (only possible with
// coincident candidates: two convex quadrilaterals in general position give at most 8) raises `over` and the
// caller redoes the pair with CAP = 16.
template <int STRIDE, int CAP>
__device__ __forceinline__ void v1l_push(const LanePts<STRIDE>& u, int& cnt, bool& over, Pt p) {
  if (cnt < CAP) {
    u.set(cnt, p);
    cnt++;
  } else {
    over = true;
  }
}

template <int STRIDE, int CAP>
__device__ __forceinline__ void v1l_vertex_in(const Pt* v, const Pt* box, const LanePts<STRIDE>& u,
                                              int& cnt, bool& over) {
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
    if (fabsf(dotp(pr, h_vec)) < h2 && fabsf(dotp(pr, w_vec)) < w2) v1l_push<STRIDE, CAP>(u, cnt, over, v[i]);
  }
}

template <int STRIDE, int CAP = R3_V1_CAP>
__device__ __forceinline__ float v1_pair_lds(const float* __restrict__ A, const float* __restrict__ B,
                                             bool iof, const LanePts<STRIDE>& u, bool* overflow = nullptr) {
  Pt v1[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v1[i] = Pt{A[2 * i], A[2 * i + 1]};
    v2[i] = Pt{B[2 * i], B[2 * i + 1]};
  }
  int cnt = 0;
  bool over = false;
  v1l_vertex_in<STRIDE, CAP>(v1, v2, u, cnt, over);
  v1l_vertex_in<STRIDE, CAP>(v2, v1, u, cnt, over);
  Pt e1[4], e2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    e1[i] = subp(v1[(i + 1) & 3], v1[i]);
    e2[i] = subp(v2[(i + 1) & 3], v2[i]);
  }
  // rbbox_border_intsec (:177-191) x LinSeg::InterSectWith (:94-140)
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const Pt a1 = v1[i], b1 = v2[j], Av = e1[i], Bv = e2[j];
      const Pt Cv = subp(a1, b1);
      if (Cv.x == 0 && Cv.y == 0) {
        v1l_push<STRIDE, CAP>(u, cnt, over, a1);
        continue;
      }
      const float D = -crossp(Av, Bv);
      if (D != 0) {
        const float ns = crossp(Cv, Bv);
        const float nt = -crossp(Av, Cv);
        if (unit_halfopen(ns, D) && unit_halfopen(nt, D)) {
          const float s = ns / D;
          v1l_push<STRIDE, CAP>(u, cnt, over, Pt{a1.x + s * Av.x, a1.y + s * Av.y});
        }
        continue;
      }
      if (crossp(Av, Cv) != 0) continue;
      const float BdtC = dotp(Bv, Cv);
      const float BdtB = dotp(Bv, Bv);
      const float AdtnC = -dotp(Av, Cv);
      const float AdtA = dotp(Av, Av);
      if (BdtC >= 0 && BdtC < BdtB) v1l_push<STRIDE, CAP>(u, cnt, over, a1);
      if (AdtnC >= 0 && AdtnC < AdtA) v1l_push<STRIDE, CAP>(u, cnt, over, b1);
    }
  }
  if (CAP < R3_V1_CAP) {
    if (overflow) *overflow = over;
    if (over) return 0.f;
  }
  if (cnt < 3) return 0.f;

  // area (:193-228), in place: slot 0 stands for the origin / sentinel, p0 stays in registers.
  // Every loop below is a chain of dependent LDS reads (the drain kernel spent half of its wave cycles in
  // s_waitcnt on them), so each one keeps several reads in flight: the next candidate is requested before the
  // current one is examined, the cleaned list is compared four entries at a time, the insertion sort requests
  // the entry below the one it is comparing.  Results and operation order are unchanged.
  const float numthres = (float)1e-2;
  const Pt p0 = u.get(0);
  int n = 1;
  Pt cand = u.get(cnt > 1 ? 1 : 0);
  for (int i = 1; i < cnt; i++) {
    const Pt d = subp(cand, p0);
    cand = u.get(i + 1 < cnt ? i + 1 : i);  // slot i + 1 > n: not touched by the set below
    bool clean = !(fabsf(d.x) < numthres && fabsf(d.y) < numthres);  // against vs[0] = origin
    for (int j = 1; clean && j < n; j += 4) {
      const Pt q0 = u.get(j), q1 = u.get(min(j + 1, CAP - 1)), q2 = u.get(min(j + 2, CAP - 1)),
               q3 = u.get(min(j + 3, CAP - 1));
      const Pt f0 = subp(d, q0), f1 = subp(d, q1), f2 = subp(d, q2), f3 = subp(d, q3);
      const bool hit = (fabsf(f0.x) < numthres && fabsf(f0.y) < numthres) |
                       ((j + 1 < n) & (fabsf(f1.x) < numthres && fabsf(f1.y) < numthres)) |
                       ((j + 2 < n) & (fabsf(f2.x) < numthres && fabsf(f2.y) < numthres)) |
                       ((j + 3 < n) & (fabsf(f3.x) < numthres && fabsf(f3.y) < numthres));
      if (hit) clean = false;
    }
    if (clean) {
      u.set(n, d);
      n++;
    }
  }
  for (int i = 2; i < n; i++) {  // i = 1 is a no-op (vs[0] < vs[0] is false)
    const Pt key = u.get(i);
    int j = i - 1;
    Pt o = u.get(j);
    for (;;) {
      const Pt below = u.get(j > 1 ? j - 1 : 1);
      if (!v1_less(key, o)) break;
      u.set(j + 1, o);
      j--;
      if (j < 1) break;
      o = below;
    }
    u.set(j + 1, key);
  }
  float a = 0;
  Pt cur = u.get(n > 1 ? 1 : 0);
  if (n <= 1) cur = Pt{0.f, 0.f};
  for (int i = 1; i < n; i += 4) {
    const Pt q1 = u.get(min(i + 1, CAP - 1)), q2 = u.get(min(i + 2, CAP - 1)),
             q3 = u.get(min(i + 3, CAP - 1)), q4 = u.get(min(i + 4, CAP - 1));
    const Pt z = Pt{0.f, 0.f};
    {
      const Pt nx = (i + 1 == n) ? z : q1;
      a += crossp(cur, nx);
      cur = nx;
    }
    if (i + 1 < n) {
      const Pt nx = (i + 2 == n) ? z : q2;
      a += crossp(cur, nx);
      cur = nx;
    }
    if (i + 2 < n) {
      const Pt nx = (i + 3 == n) ? z : q3;
      a += crossp(cur, nx);
      cur = nx;
    }
    if (i + 3 < n) {
      const Pt nx = (i + 4 == n) ? z : q4;
      a += crossp(cur, nx);
      cur = nx;
    }
  }
  float su = a / 2;
  const float s1 = A[8], s2 = B[8];
  su = (s1 < su) ? s1 : su;
  su = (s2 < su) ? s2 : su;
  su = (su < 0.f) ? 0.f : su;
  return iof ? su / s1 : su / (s1 + s2 - su);
}

// ----------------------------------------------------------------------------------------
// hull (v2 / v3)
// ----------------------------------------------------------------------------------------
// CAP < 24: the point list of a lane holds CAP slots; a pair with more candidate points sets *over and returns 0
// (the caller redoes it with the full 24): two rectangles in general position give at most 8 crossings, and vertices
// inside the other box only replace crossings, so 12 slots cover all but degenerate (touching / duplicate-vertex) pairs.
template <bool V2, int STRIDE, int CAP = 24>
__device__ __forceinline__ float hull_pair_lds(const float* __restrict__ A, const float* __restrict__ B,
                                               bool iou_mode, const LanePts<STRIDE>& q, bool* over = nullptr) {
  const float ax = A[0], ay = A[1], bx = B[0], by = B[1];
  const double csx = (double)(ax + bx) / 2.0;
  const double csy = (double)(ay + by) / 2.0;
  const float x1 = (float)((double)ax - csx), y1 = (float)((double)ay - csy);
  const float x2 = (float)((double)bx - csx), y2 = (float)((double)by - csy);
  const float area1 = A[6], area2 = B[6];
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

  BoxRec ra, rb;  // only f[2..5] are read by hull_vertices
#pragma unroll
  for (int i = 2; i < 6; i++) {
    ra.f[i] = A[i];
    rb.f[i] = B[i];
  }
  Pt pts1[4], pts2[4];
  hull_vertices<V2>(x1, y1, ra, pts1);
  hull_vertices<V2>(x2, y2, rb, pts2);

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
      const float det = crossp(vec2[j], vec1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      const Pt vec12 = subp(pts2[j], pts1[i]);
      const float n1 = crossp(vec2[j], vec12);
      const float n2 = crossp(vec1[i], vec12);
      if (unit_closed(n1, det) && unit_closed(n2, det)) {
        const float t1 = n1 / det;
        if (CAP == 24 || num < CAP) q.set(num, Pt{pts1[i].x + vec1[i].x * t1, pts1[i].y + vec1[i].y * t1});
        num++;
      }
    }
  }
  {
    const Pt AB = vec2[0], DA = vec2[3];
    const float ABdotAB = dotp(AB, AB), ADdotAD = dotp(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const Pt AP = subp(pts1[i], pts2[0]);
      const float APdotAB = dotp(AP, AB);
      const float APdotAD = -dotp(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) {
        if (CAP == 24 || num < CAP) q.set(num, pts1[i]);
        num++;
      }
    }
  }
  {
    const Pt AB = vec1[0], DA = vec1[3];
    const float ABdotAB = dotp(AB, AB), ADdotAD = dotp(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const Pt AP = subp(pts2[i], pts1[0]);
      const float APdotAB = dotp(AP, AB);
      const float APdotAD = -dotp(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) {
        if (CAP == 24 || num < CAP) q.set(num, pts2[i]);
        num++;
      }
    }
  }
  if (CAP < 24 && num > CAP) {
    *over = true;
    return 0.f;
  }
  float intersection = 0.f;
  if (num > 2) {
    int t = 0;
    Pt best = q.get(0);
    for (int i = 1; i < num; i++) {
      const Pt p = q.get(i);
      if (p.y < best.y || (p.y == best.y && p.x < best.x)) {
        t = i;
        best = p;
      }
    }
    // q[i] = p[i] - start, then swap slot 0 <-> t
    const Pt first = subp(q.get(0), best);
    for (int i = 1; i < num; i++) q.set(i, subp(q.get(i), best));
    q.set(0, Pt{best.x - best.x, best.y - best.y});  // p[t] - start
    if (t != 0) q.set(t, first);
    // device-branch exchange sort (:203-216)
    for (int i = 1; i < num - 1; i++) {
      Pt qi = q.get(i);
      for (int j = i + 1; j < num; j++) {
        const Pt qj = q.get(j);
        const float cp = crossp(qi, qj);
        bool sw = ((double)cp < -1e-6);
        if (!sw && fabs((double)cp) < 1e-6) sw = dotp(qi, qi) > dotp(qj, qj);
        if (sw) {
          q.set(j, qi);
          qi = qj;
        }
      }
      q.set(i, qi);
    }
    int k;
    for (k = 1; k < num; k++) {
      const Pt p = q.get(k);
      if ((double)dotp(p, p) > 1e-8) break;
    }
    int m = 1;
    if (k < num) {
      q.set(1, q.get(k));
      m = 2;
      for (int i = k + 1; i < num; i++) {
        const Pt qi = q.get(i);
        while (m > 1) {
          const Pt base = q.get(m - 2);
          const Pt q1 = subp(qi, base), q2 = subp(q.get(m - 1), base);
          const bool pop = V2 ? (crossp(q1, q2) >= 0) : (q1.x * q2.y >= q2.x * q1.y);
          if (pop) m--; else break;
        }
        q.set(m, qi);
        m++;
      }
    }
    if (m > 2) {
      float area = 0;
      const Pt q0 = q.get(0);
      Pt prev = subp(q.get(1), q0);
      for (int i = 1; i < m - 1; i++) {
        const Pt nxt = subp(q.get(i + 1), q0);
        area += fabsf(crossp(prev, nxt));
        prev = nxt;
      }
      intersection = (float)((double)area / 2.0);
    }
  }
  return iou_mode ? intersection / (area1 + area2 - intersection) : intersection / area1;
}

template <int GEOM, int STRIDE>
__device__ __forceinline__ float pair_slow_lds(const float* __restrict__ A, const float* __restrict__ B,
                                               bool iof, const LanePts<STRIDE>& pts) {
  if (GEOM == 1) return v1_pair_lds<STRIDE>(A, B, iof, pts);
  if (GEOM == 2) return hull_pair_lds<true, STRIDE>(A, B, !iof, pts);
  return hull_pair_lds<false, STRIDE>(A, B, !iof, pts);
}
template <int GEOM>
constexpr int pts_slots() { return GEOM == 1 ? 16 : 24; }
