// r3_clip.h -- the v1 pair clip (rbbox_geo_kernel.cu:88-268 == rnms_kernel.cu:15-200) as STRAIGHT-LINE code for
// pairs in general position; everything else is flagged and redone by the exact branchy form (r3_geom_lds.h).
//
// Round 5.  The drains (iou_drain3 / nms_drain / assign_drain) were bound by the instructions they issue: per wave
// and clip ~1500 vector + ~1400 scalar instructions, the scalar half exec-mask bookkeeping of lane-divergent
// branches, and long dependent chains through LDS (de-dup, insertion sort, shoelace walk the candidate list slot by
// slot).  This form issues ~40 % fewer vector instructions and almost no scalar ones:
//
//   * WHICH edge pairs cross is decided from signs, without dividing and without the 16 branchy segment tests.
//     With ns(i,j) = cross(v1[i] - v2[j], B_j) and nt(i,j) = -cross(A_i, v1[i] - v2[j]) -- the very numerators the
//     reference divides (rbbox_geo_kernel.cu:111-112) -- exact arithmetic gives ns(i+1,j) = ns(i,j) - D(i,j) and
//     nt(i,j+1) = nt(i,j) - D(i,j), D = -cross(A_i, B_j).  So 0 <= ns/D < 1 <=> ns(i,j), ns(i+1,j) have opposite
//     signs, and likewise for nt.  In floating point this holds whenever all 32 numerators are larger in magnitude
//     than 3 delta, delta = 8.2 u max(|C|,|E|) |E| the rounding-error bound of a numerator / of D (u = 2^-24; |C|
//     the largest vertex difference, |E| the largest edge component) -- proof in DESIGN_HISTORY 4.1.  Such a pair has no
//     coincident vertices (C == 0 => ns == 0), no parallel-and-collinear edges (D == 0 needs nt == 0 to produce a
//     point) and no quotient near 0 or 1, i.e. none of the reference's special branches (:99-103, :118-139) can fire.
//     A pair with a smaller numerator is FLAGGED (`redo`) and the caller runs the exact form on it.
//   * the accepted crossings (<= 2 per edge of box 1 in general position; a third flags the pair) are selected with
//     conditional moves, divided (the reference's s = ns / D, correctly rounded) and appended in the reference's order:
//     vertices of 1 inside 2, vertices of 2 inside 1 (the reference's own dot-product tests, :157-175), crossings
//     edge-major.  The only use of LDS is as the dynamic index it is good at: a candidate is written at slot `cnt`
//     and the <= 8 candidates are read back into registers at static slots (two convex quadrilaterals in general
//     position meet in <= 8 points; a 9th flags the pair).
//   * de-dup (:196-209), insertion sort (:210-215) and shoelace (:216-227) on registers, fully unrolled: the
//     de-dup is the reference's sequential rule evaluated with predicates; the sort is a RANK: one cross product per
//     unordered pair (cross(b, a) is exactly -cross(a, b) in floating point) says which of the two the stable
//     insertion sort leaves in front; if the 21 answers form a total order (every rank taken once) the insertion
//     sort, which only ever asks those questions, must produce it -- otherwise the pair is flagged.  The kept
//     candidates go to LDS slot `rank`, come back in order, and the shoelace sum adds them in the reference's order.
//
// Every float produced here that reaches the result is produced by the reference's own operation on the reference's
// own operands (IEEE + - * /, no contraction: -ffp-contract=off), so an unflagged pair is bit-identical to the exact
// form.  The functions are __host__ __device__: tests/test_clip_host.py compiles this header with the host compiler
// and checks 10^6s of pairs against the oracle without a GPU.
#pragma once
#include <hip/hip_runtime.h>

#define R3_CLIP_SLOTS 9  // 8 candidates + the dump slot that takes the writes of rejected candidates

// candidate store of ONE pair: slot -> point.  Device: wave-private LDS, [slot][lane] (conflict-free 8-byte words).
template <int STRIDE>
struct ClipLds {
  float2* base;  // already offset by the lane
  __device__ __forceinline__ void set(int s, float x, float y) const { base[s * STRIDE] = make_float2(x, y); }
  __device__ __forceinline__ void get(int s, float& x, float& y) const {
    const float2 v = base[s * STRIDE];
    x = v.x;
    y = v.y;
  }
  // true in every lane when the condition holds in any lane: rare paths are wave-uniform branches
  __device__ __forceinline__ bool any(bool c) const { return __builtin_amdgcn_ballot_w64(c) != 0; }
};
struct ClipHost {  // (the host harness of the tests)
  mutable float px[R3_CLIP_SLOTS], py[R3_CLIP_SLOTS];
  void set(int s, float x, float y) const { px[s] = x, py[s] = y; }
  void get(int s, float& x, float& y) const { x = px[s], y = py[s]; }
  bool any(bool c) const { return c; }
};

// A, B: v1 records (r3_geom.h): f[0..7] the vertices (x0, y0, .., x3, y3), f[8] = w * h.
// Returns the IoU / IoF of an unflagged pair; `redo` = the pair needs the exact form (the value is then meaningless).
template <class Store>
__host__ __device__ __forceinline__ float v1_clip_fast(const float* __restrict__ A, const float* __restrict__ B,
                                                       const bool iof, const Store& st, bool& redo) {
  float ax[4], ay[4], bx[4], by[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    ax[i] = A[2 * i], ay[i] = A[2 * i + 1];
    bx[i] = B[2 * i], by[i] = B[2 * i + 1];
  }
  float Ax[4], Ay[4], Bx[4], By[4];  // edges: v[i + 1] - v[i] (:183-186 hand a1 = v[i], a2 = v[i + 1] to the segment test)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Ax[i] = ax[(i + 1) & 3] - ax[i], Ay[i] = ay[(i + 1) & 3] - ay[i];
    Bx[i] = bx[(i + 1) & 3] - bx[i], By[i] = by[(i + 1) & 3] - by[i];
  }
  int cnt = 0;
  bool bad = false;
  // ---- vertices strictly inside the other box (vertex_in_rbbox, :157-175): the reference's own arithmetic
  {
    const float cx = 0.5f * (bx[0] + bx[2]), cy = 0.5f * (by[0] + by[2]);
    const float wx = 0.5f * Bx[0], wy = 0.5f * By[0], hx = 0.5f * Bx[1], hy = 0.5f * By[1];
    const float h2 = hx * hx + hy * hy, w2 = wx * wx + wy * wy;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float px = ax[i] - cx, py = ay[i] - cy;
      const bool in = (fabsf(px * hx + py * hy) < h2) & (fabsf(px * wx + py * wy) < w2);
      st.set(in ? cnt : 8, ax[i], ay[i]);
      cnt += in ? 1 : 0;
    }
  }
  {
    const float cx = 0.5f * (ax[0] + ax[2]), cy = 0.5f * (ay[0] + ay[2]);
    const float wx = 0.5f * Ax[0], wy = 0.5f * Ay[0], hx = 0.5f * Ax[1], hy = 0.5f * Ay[1];
    const float h2 = hx * hx + hy * hy, w2 = wx * wx + wy * wy;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float px = bx[i] - cx, py = by[i] - cy;
      const bool in = (fabsf(px * hx + py * hy) < h2) & (fabsf(px * wx + py * wy) < w2);
      st.set(in ? cnt : 8, bx[i], by[i]);
      cnt += in ? 1 : 0;
    }
  }
  // ---- the 32 numerators (rbbox_geo_kernel.cu:105-112: C = a1 - b1, s = cross(C, B) / D, t = -cross(A, C) / D)
  float ns[4][4], nt[4][4];
  float cmax = 0.f, emax = 0.f, nmin = 3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    emax = fmaxf(emax, fmaxf(fmaxf(fabsf(Ax[i]), fabsf(Ay[i])), fmaxf(fabsf(Bx[i]), fabsf(By[i]))));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float Cx = ax[i] - bx[j], Cy = ay[i] - by[j];
      ns[i][j] = Cx * By[j] - Bx[j] * Cy;
      nt[i][j] = -(Ax[i] * Cy - Cx * Ay[i]);
      cmax = fmaxf(cmax, fmaxf(fabsf(Cx), fabsf(Cy)));
      nmin = fminf(nmin, fminf(fabsf(ns[i][j]), fabsf(nt[i][j])));
    }
  }
  // general position: every numerator beyond 3.5 x the rounding bound (`unc` otherwise).  Flagged outright: a scale
  // that is not sane (overflow / underflow of the products) and NaN / Inf (fmaxf / fminf drop NaN operands, so a sum
  // of all coordinates times zero carries them).
  bool unc;
  {
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) z += (ax[i] + ay[i]) + (bx[i] + by[i]);
    const float big = fmaxf(cmax, emax);
    const float lim = 1.8e-6f * big * emax;  // 3.5 * 8.2 * 2^-24 = 1.71e-6
    unc = !(nmin > lim);
    bad = bad | !(z * 0.f == 0.f) | !(big < 1.0e15f) | !(emax > 1.0e-12f);
  }
  // ---- which edge pairs cross (rbbox_border_intsec, :177-191 x LinSeg::InterSectWith, :94-140)
  bool acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // opposite signs <=> the reference's 0 <= s < 1 (and t), in general position
      const bool so = (ns[i][j] < 0.f) != (ns[(i + 1) & 3][j] < 0.f);
      const bool to = (nt[i][j] < 0.f) != (nt[i][(j + 1) & 3] < 0.f);
      acc[i][j] = so & to;
    }
  }
  // A numerator too small to trust its sign (~4e-4 of overlapping pairs, 2-3 % of wavefronts): the reference's own
  // tests, with its divisions, for every edge pair of the wavefront's pairs (for the lanes in general position they
  // give the sign rule's answer).  What stays flagged is what makes the reference leave its main branch: coincident
  // vertices (C == 0, :99-103) and parallel collinear edges (D == 0 and cross(A, C) == 0, :118-139).
  if (st.any(unc)) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float Cx = ax[i] - bx[j], Cy = ay[i] - by[j];
        const float D = -(Ax[i] * By[j] - Bx[j] * Ay[i]);
        const float s = ns[i][j] / D, t = nt[i][j] / D;
        acc[i][j] = (D != 0.f) & (0.f <= s) & (s < 1.f) & (0.f <= t) & (t < 1.f);
        bad = bad | ((Cx == 0.f) & (Cy == 0.f)) | ((D == 0.f) & (nt[i][j] == 0.f));
      }
    }
  }
  // ---- crossings, edge-major: the (at most two) accepted edge pairs of every edge of box 1
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int na = (int)acc[i][0] + (int)acc[i][1] + (int)acc[i][2] + (int)acc[i][3];
    bad = bad | (na > 2);
    // first / second accepted j
    const bool f0 = acc[i][0], f1 = !acc[i][0] & acc[i][1], f2 = !acc[i][0] & !acc[i][1] & acc[i][2];
    const float n0 = f0 ? ns[i][0] : f1 ? ns[i][1] : f2 ? ns[i][2] : ns[i][3];
    const float bx0 = f0 ? Bx[0] : f1 ? Bx[1] : f2 ? Bx[2] : Bx[3];
    const float by0 = f0 ? By[0] : f1 ? By[1] : f2 ? By[2] : By[3];
    const bool s1 = acc[i][0] & acc[i][1], s2 = (acc[i][0] | acc[i][1]) & acc[i][2] & !s1;
    const float n1 = s1 ? ns[i][1] : s2 ? ns[i][2] : ns[i][3];
    const float bx1 = s1 ? Bx[1] : s2 ? Bx[2] : Bx[3];
    const float by1 = s1 ? By[1] : s2 ? By[2] : By[3];
    {
      const float D = -(Ax[i] * by0 - bx0 * Ay[i]);
      const float s = n0 / D;
      const bool ok = na >= 1;
      st.set(ok ? (cnt < 8 ? cnt : 8) : 8, ax[i] + s * Ax[i], ay[i] + s * Ay[i]);
      cnt += ok ? 1 : 0;
    }
    {
      const float D = -(Ax[i] * by1 - bx1 * Ay[i]);
      const float s = n1 / D;
      const bool ok = na >= 2;
      st.set(ok ? (cnt < 8 ? cnt : 8) : 8, ax[i] + s * Ax[i], ay[i] + s * Ay[i]);
      cnt += ok ? 1 : 0;
    }
  }
  bad = bad | (cnt > 8);
  // ---- area (:193-228) on registers
  float ux[8], uy[8];
#pragma unroll
  for (int k = 0; k < 8; k++) st.get(k, ux[k], uy[k]);
  const float numthres = (float)1e-2;
  float dx[8], dy[8];
  bool kept[8];
  int before[8];  // kept points in front of point i in candidate order (the origin, slot 0, not counted)
  kept[0] = true;
  dx[0] = dy[0] = 0.f;
  int n = 1;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    dx[i] = ux[i] - ux[0], dy[i] = uy[i] - uy[0];
    bool clean = (i < cnt) & !((fabsf(dx[i]) < numthres) & (fabsf(dy[i]) < numthres));  // against vs[0] = the origin
#pragma unroll
    for (int j = 1; j < i; j++) {
      const float fx = dx[i] - dx[j], fy = dy[i] - dy[j];
      clean = clean & !(kept[j] & (fabsf(fx) < numthres) & (fabsf(fy) < numthres));
    }
    kept[i] = clean;
    before[i] = n - 1;
    n += clean ? 1 : 0;
  }
  // Rank of every kept point in the order the stable insertion sort (operator< :74-79 = cross > 0) leaves: point j
  // (the later key) ends in front of point i < j  <=>  cross(d_j, d_i) > 0  <=>  cross(d_i, d_j) < 0.  Dropped points
  // take part as (0, 0): their cross products are +-0, never < 0.  The seven counters "moved in front of me - I moved in
  // front of" share one register, four bits each, biased by 8.
  float zx[8], zy[8];
#pragma unroll
  for (int i = 1; i < 8; i++) zx[i] = kept[i] ? dx[i] : 0.f, zy[i] = kept[i] ? dy[i] : 0.f;
  unsigned packed = 0x88888880u;
#pragma unroll
  for (int i = 1; i < 8; i++) {
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      const float c = zx[i] * zy[j] - zx[j] * zy[i];  // cross(d_i, d_j); cross(d_j, d_i) is exactly -c
      packed += (c < 0.f) ? (1u << (4 * i)) - (1u << (4 * j)) : 0u;
    }
  }
  unsigned taken = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    const int rank = before[i] + (int)((packed >> (4 * i)) & 15u) - 7;  // 1 + before + (field - 8)
    taken += kept[i] ? (1u << (rank & 31)) : 0u;  // (a sum: a rank taken twice carries into a wrong pattern or beyond)
    st.set(kept[i] ? (rank & 7) : 8, dx[i], dy[i]);
  }
  // (the 21 answers form a total order <=> every rank 1 .. n - 1 is taken once; the insertion sort, which only ever asks
  // these questions, then produces that order)
  redo = bad | (taken != (1u << n) - 2u);
  float vx[8], vy[8];
#pragma unroll
  for (int k = 1; k < 8; k++) st.get(k, vx[k], vy[k]);
  float a = 0.f;
#pragma unroll
  for (int k = 1; k < 7; k++) {
    const float t = vx[k] * vy[k + 1] - vx[k + 1] * vy[k];
    a = (k + 1 < n) ? a + t : a;
  }
  // (the closing term cross(vs[n - 1], origin) is +-0 and a is never -0: adding it changes nothing)
  float su = a / 2;
  const float s1 = A[8], s2 = B[8];
  su = (s1 < su) ? s1 : su;
  su = (s2 < su) ? s2 : su;
  su = (su < 0.f) ? 0.f : su;
  const float r = iof ? su / s1 : su / (s1 + s2 - su);
  return cnt >= 3 ? r : 0.f;
}

// ----------------------------------------------------------------------------------------------------------------
// The hull geometry (box_iou_rotated_utils.h:55-361, v3; ml_nms_rotated's copy, v2) the same way (round 5).
//   * crossings from signs: with n1(i,j) = cross(vec2[j], pts2[j] - pts1[i]) and n2(i,j) = cross(vec1[i], pts2[j] - pts1[i])
//     -- the numerators of t1, t2 (:99-100) -- exact arithmetic gives n1(i+1,j) = n1(i,j) - det and n2(i,j+1) =
//     n2(i,j) - det, det = cross(vec2[j], vec1[i]): 0 <= t <= 1 <=> opposite signs, under the same 3.5 x rounding-bound
//     limit as v1 (a numerator under it: the reference's own tests with its divisions, its |det| <= 1e-14 skip included,
//     in one wave-uniform block; the closed interval only matters at numerators that are exactly zero there);
//   * vertex-in-box by the reference's own dot products (:113-152); at most two crossings per edge, eight points;
//   * Graham (:157-289, device branch of the sort :193-216) on registers: the pivot (min y, then min x), the other points
//     RANKED by the sign of their pairwise cross products -- the exchange sort's answer whenever every |cross| >= 1e-6
//     (its tolerance branch never fires) and the signs form a total order; the scan's pop tests are EVALUATED on that
//     order, not executed: all false => nothing is popped and the stack is the order itself.  A point within 1e-8 of
//     the pivot (squared), a |cross| under 1e-6, a pop, an inconsistent order flag the pair for the exact list form.
//   * the fan area (:291-303) in the scan's order.
// A, B: hull records (r3_geom.h): f[0], f[1] the centre, f[2..5] = sin/2 h, cos/2 w, cos/2 h, sin/2 w, f[6] = w h.
// ----------------------------------------------------------------------------------------------------------------
template <bool V2, class Store>
__host__ __device__ __forceinline__ float hull_clip_fast(const float* __restrict__ A, const float* __restrict__ B,
                                                         const bool iou_mode, const Store& st, bool& redo) {
  const float cax = A[0], cay = A[1], cbx = B[0], cby = B[1];
  const double csx = (double)(cax + cbx) / 2.0;
  const double csy = (double)(cay + cby) / 2.0;
  const float x1 = (float)((double)cax - csx), y1 = (float)((double)cay - csy);
  const float x2 = (float)((double)cbx - csx), y2 = (float)((double)cby - csy);
  const float area1 = A[6], area2 = B[6];
  redo = false;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;  // (:353; wave-divergent early exit: rare)
  float ax[4], ay[4], bx[4], by[4];
  {
    const float sh = A[2], cw = A[3], ch = A[4], sw = A[5];
    if (!V2) { ax[0] = x1 + sh + cw; ay[0] = y1 + ch - sw; ax[1] = x1 - sh + cw; ay[1] = y1 - ch - sw; }
    else { ax[0] = x1 - sh - cw; ay[0] = y1 + ch - sw; ax[1] = x1 + sh - cw; ay[1] = y1 - ch - sw; }
    ax[2] = 2 * x1 - ax[0]; ay[2] = 2 * y1 - ay[0]; ax[3] = 2 * x1 - ax[1]; ay[3] = 2 * y1 - ay[1];
  }
  {
    const float sh = B[2], cw = B[3], ch = B[4], sw = B[5];
    if (!V2) { bx[0] = x2 + sh + cw; by[0] = y2 + ch - sw; bx[1] = x2 - sh + cw; by[1] = y2 - ch - sw; }
    else { bx[0] = x2 - sh - cw; by[0] = y2 + ch - sw; bx[1] = x2 + sh - cw; by[1] = y2 - ch - sw; }
    bx[2] = 2 * x2 - bx[0]; by[2] = 2 * y2 - by[0]; bx[3] = 2 * x2 - bx[1]; by[3] = 2 * y2 - by[1];
  }
  float Ax[4], Ay[4], Bx[4], By[4];  // vec1, vec2 (:85-88)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Ax[i] = ax[(i + 1) & 3] - ax[i], Ay[i] = ay[(i + 1) & 3] - ay[i];
    Bx[i] = bx[(i + 1) & 3] - bx[i], By[i] = by[(i + 1) & 3] - by[i];
  }
  // ---- the numerators of t1, t2 (:97-100)
  float n1[4][4], n2[4][4];
  float cmax = 0.f, emax = 0.f, nmin = 3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    emax = fmaxf(emax, fmaxf(fmaxf(fabsf(Ax[i]), fabsf(Ay[i])), fmaxf(fabsf(Bx[i]), fabsf(By[i]))));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float vx = bx[j] - ax[i], vy = by[j] - ay[i];  // vec12
      n1[i][j] = Bx[j] * vy - vx * By[j];
      n2[i][j] = Ax[i] * vy - vx * Ay[i];
      cmax = fmaxf(cmax, fmaxf(fabsf(vx), fabsf(vy)));
      nmin = fminf(nmin, fminf(fabsf(n1[i][j]), fabsf(n2[i][j])));
    }
  }
  bool bad = false, unc;
  {
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) z += (ax[i] + ay[i]) + (bx[i] + by[i]);
    const float big = fmaxf(cmax, emax);
    const float lim = 1.8e-6f * big * emax;
    unc = !(nmin > lim);
    bad = !(z * 0.f == 0.f) | !(big < 1.0e15f) | !(lim > 1.0e-12f);  // (the last: 3 delta stays far above the 1e-14 skip)
  }
  bool acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool so = (n1[i][j] < 0.f) != (n1[(i + 1) & 3][j] < 0.f);
      const bool to = (n2[i][j] < 0.f) != (n2[i][(j + 1) & 3] < 0.f);
      acc[i][j] = so & to;
    }
  }
  if (st.any(unc)) {  // the reference's own tests (:91-110) for every edge pair of the wavefront's pairs
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float det = Bx[j] * Ay[i] - Ax[i] * By[j];
        const float t1 = n1[i][j] / det, t2 = n2[i][j] / det;
        acc[i][j] = !(fabs((double)det) <= 1e-14) & (t1 >= 0.0f) & (t1 <= 1.0f) & (t2 >= 0.0f) & (t2 <= 1.0f);
      }
    }
  }
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int na = (int)acc[i][0] + (int)acc[i][1] + (int)acc[i][2] + (int)acc[i][3];
    bad = bad | (na > 2);
    const bool f0 = acc[i][0], f1 = !acc[i][0] & acc[i][1], f2 = !acc[i][0] & !acc[i][1] & acc[i][2];
    const float m0 = f0 ? n1[i][0] : f1 ? n1[i][1] : f2 ? n1[i][2] : n1[i][3];
    const float bx0 = f0 ? Bx[0] : f1 ? Bx[1] : f2 ? Bx[2] : Bx[3];
    const float by0 = f0 ? By[0] : f1 ? By[1] : f2 ? By[2] : By[3];
    const bool s1 = acc[i][0] & acc[i][1], s2 = (acc[i][0] | acc[i][1]) & acc[i][2] & !s1;
    const float m1 = s1 ? n1[i][1] : s2 ? n1[i][2] : n1[i][3];
    const float bx1 = s1 ? Bx[1] : s2 ? Bx[2] : Bx[3];
    const float by1 = s1 ? By[1] : s2 ? By[2] : By[3];
    {
      const float det = bx0 * Ay[i] - Ax[i] * by0;
      const float t1 = m0 / det;
      const bool ok = na >= 1;
      st.set(ok ? (cnt < 8 ? cnt : 8) : 8, ax[i] + Ax[i] * t1, ay[i] + Ay[i] * t1);
      cnt += ok ? 1 : 0;
    }
    {
      const float det = bx1 * Ay[i] - Ax[i] * by1;
      const float t1 = m1 / det;
      const bool ok = na >= 2;
      st.set(ok ? (cnt < 8 ? cnt : 8) : 8, ax[i] + Ax[i] * t1, ay[i] + Ay[i] * t1);
      cnt += ok ? 1 : 0;
    }
  }
  // ---- vertices inside the other box (:113-152): the reference's own arithmetic, closed comparisons
  {
    const float ABx = Bx[0], ABy = By[0], DAx = Bx[3], DAy = By[3];
    const float ABdotAB = ABx * ABx + ABy * ABy, ADdotAD = DAx * DAx + DAy * DAy;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float APx = ax[i] - bx[0], APy = ay[i] - by[0];
      const float APdotAB = APx * ABx + APy * ABy;
      const float APdotAD = -(APx * DAx + APy * DAy);
      const bool in = (APdotAB >= 0) & (APdotAD >= 0) & (APdotAB <= ABdotAB) & (APdotAD <= ADdotAD);
      st.set(in ? (cnt < 8 ? cnt : 8) : 8, ax[i], ay[i]);
      cnt += in ? 1 : 0;
    }
  }
  {
    const float ABx = Ax[0], ABy = Ay[0], DAx = Ax[3], DAy = Ay[3];
    const float ABdotAB = ABx * ABx + ABy * ABy, ADdotAD = DAx * DAx + DAy * DAy;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float APx = bx[i] - ax[0], APy = by[i] - ay[0];
      const float APdotAB = APx * ABx + APy * ABy;
      const float APdotAD = -(APx * DAx + APy * DAy);
      const bool in = (APdotAB >= 0) & (APdotAD >= 0) & (APdotAB <= ABdotAB) & (APdotAD <= ADdotAD);
      st.set(in ? (cnt < 8 ? cnt : 8) : 8, bx[i], by[i]);
      cnt += in ? 1 : 0;
    }
  }
  bad = bad | (cnt > 8);
  const int num = cnt < 8 ? cnt : 8;
  float px[8], py[8];
#pragma unroll
  for (int k = 0; k < 8; k++) st.get(k, px[k], py[k]);
  // ---- the pivot: lowest y, then lowest x, the first such index (:166-176)
  float sx = px[0], sy = py[0];
  int tix = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    const bool better = (i < num) & ((py[i] < sy) | ((py[i] == sy) & (px[i] < sx)));
    sx = better ? px[i] : sx;
    sy = better ? py[i] : sy;
    tix = better ? i : tix;
  }
  // the other points relative to it; a dropped slot (beyond num, or the pivot itself) takes part as (0, 0)
  float qx[8], qy[8];
  bool use[8];
  int before[8];
  int nb = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    use[i] = (i < num) & (i != tix);
    const float dx = px[i] - sx, dy = py[i] - sy;
    qx[i] = use[i] ? dx : 0.f;
    qy[i] = use[i] ? dy : 0.f;
    bad = bad | (use[i] & !((double)(dx * dx + dy * dy) > 1e-8));  // (:233-238: a point on the pivot is skipped there)
    before[i] = nb;
    nb += use[i] ? 1 : 0;
  }
  // rank by the sign of the pairwise cross products: the exchange sort (:203-216) puts q_j in front of q_i <=> cross(q_i, q_j) < -1e-6
  unsigned packed = 0x88888888u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      const float c = qx[i] * qy[j] - qx[j] * qy[i];
      bad = bad | (use[i] & use[j] & (fabs((double)c) < 1e-6));  // (the sort's tolerance branch: distance decides there)
      packed += (c < 0.f) ? (1u << (4 * i)) - (1u << (4 * j)) : 0u;
    }
  }
  unsigned taken = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int rank = before[i] + (int)((packed >> (4 * i)) & 15u) - 7;  // 1 + before + (field - 8): 1 .. num - 1
    taken += use[i] ? (1u << (rank & 31)) : 0u;
    st.set(use[i] ? (rank & 7) : 8, qx[i], qy[i]);
  }
  bad = bad | ((num > 2) & (taken != (1u << num) - 2u));
  float vx[8], vy[8];
  vx[0] = vy[0] = 0.f;  // the pivot, shifted to zero
#pragma unroll
  for (int k = 1; k < 8; k++) st.get(k, vx[k], vy[k]);
  // the scan's pop tests (:253 ml / :264) on that order: none may fire
#pragma unroll
  for (int i = 2; i < 8; i++) {
    const float q1x = vx[i] - vx[i - 2], q1y = vy[i] - vy[i - 2], q2x = vx[i - 1] - vx[i - 2], q2y = vy[i - 1] - vy[i - 2];
    const bool pop = V2 ? (q1x * q2y - q2x * q1y >= 0) : (q1x * q2y >= q2x * q1y);
    bad = bad | ((i < num) & pop);
  }
  float area = 0.f;
#pragma unroll
  for (int i = 1; i < 7; i++) {
    const float t = fabsf(vx[i] * vy[i + 1] - vx[i + 1] * vy[i]);
    area = (i + 1 < num) ? area + t : area;
  }
  redo = bad;
  const float inter = num > 2 ? (float)((double)area / 2.0) : 0.f;
  return iou_mode ? inter / (area1 + area2 - inter) : inter / area1;
}
