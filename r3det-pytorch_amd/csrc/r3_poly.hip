// r3_poly.hip -- the exported ops outside the shipped training / inference configs:
//   polygon_iou  (polygon_geo/src/polygon_geo_cpu.cpp: CPU-only in the reference; DOTA evaluation)
//   poly_nms     (nms_rotated/src/poly_nms_cuda.cu: 8-coordinate polygon NMS; DOTA result merge)
//   convex_sort  (convex/src/convex_cuda.cu: Graham-style ordering used by the differentiable
//                 aligned overlaps and core/bbox/rtransforms.py)
// None of them is performance critical (tens to thousands of items): one thread per pair / item,
// the arithmetic restated operation by operation so that results match the reference's.
#include <hip/hip_runtime.h>

#include "r3_geom.h"
#include "r3_kernels.h"

namespace {

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------ polygon_iou
// polygon2points (polygon_geo_cpu.cpp:137-158): vertices 2, 3 insertion-sorted around vertex 0
__device__ __forceinline__ void poly_points(const float* __restrict__ poly, Pt* vs) {
#pragma unroll
  for (int i = 0; i < 4; i++) vs[i] = Pt{poly[2 * i], poly[2 * i + 1]};
  for (int i = 2; i < 4; i++) {
    const Pt pt = vs[i];
    int j;
    for (j = i - 1; v1_less(subp(pt, vs[0]), subp(vs[j], vs[0])); j--) vs[j + 1] = vs[j];
    vs[j + 1] = pt;
  }
}

// vertex_in_polygon (:160-184)
__device__ __forceinline__ void poly_vertex_in(const Pt* v1, const Pt* v2, Pt* u, int& cnt) {
  for (int i = 0; i < 4; i++) {
    bool inside = true;
    for (int j = 0; j < 4; j++) {
      if (v1_less(subp(v1[i], v2[j]), subp(v2[(j + 1) & 3], v2[j]))) {
        inside = false;
        break;
      }
    }
    if (inside) v1_push(u, cnt, v1[i]);
  }
}

// pair body of polygon_iou_kernel (:241-266)
__device__ float polygon_pair(const float* __restrict__ a, const float* __restrict__ b) {
  Pt v1[4], v2[4], u[R3_V1_CAP];
  poly_points(a, v1);
  poly_points(b, v2);
  int cnt = 0;
  poly_vertex_in(v1, v2, u, cnt);
  poly_vertex_in(v2, v1, u, cnt);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) v1_segment(v1[i], v1[(i + 1) & 3], v2[j], v2[(j + 1) & 3], u, cnt);
  if (cnt < 3) return 0.f;
  const float s1 = v1_area(v1, 4), s2 = v1_area(v2, 4);
  float su = v1_area(u, cnt);
  su = (s1 < su) ? s1 : su;
  su = (s2 < su) ? s2 : su;
  su = (su < 0.f) ? 0.f : su;
  return su / (s1 + s2 - su);
}

__global__ __launch_bounds__(256) void polygon_iou_kernel(const float* __restrict__ a, int na,
                                                          const float* __restrict__ b, int nb,
                                                          float* __restrict__ out) {
  const long long total = (long long)na * nb;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t / nb), j = (int)(t - (long long)i * nb);
    out[t] = polygon_pair(a + (size_t)i * 8, b + (size_t)j * 8);
  }
}

// --------------------------------------------------------------------------------------- poly_nms
// devPolyIoU (poly_nms_cuda.cu:21-140): signed triangle-fan clipping; eps compared in double.
#define PN_EPS 1E-8
__device__ __forceinline__ int pn_sig(float d) { return (d > PN_EPS) - (d < -PN_EPS); }
__device__ __forceinline__ bool pn_eq(float2 a, float2 b) { return pn_sig(a.x - b.x) == 0 && pn_sig(a.y - b.y) == 0; }
__device__ __forceinline__ float pn_cross(float2 o, float2 a, float2 b) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}
__device__ __forceinline__ float pn_area(float2* ps, int n) {
  ps[n] = ps[0];
  float res = 0;
  for (int i = 0; i < n; i++) res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return (float)(res / 2.0);
}
__device__ __forceinline__ void pn_line_cross(float2 a, float2 b, float2 c, float2 d, float2& p) {
  const float s1 = pn_cross(a, b, c), s2 = pn_cross(a, b, d);
  if (pn_sig(s1) == 0 && pn_sig(s2) == 0) return;
  if (pn_sig(s2 - s1) == 0) return;
  p.x = (c.x * s2 - d.x * s1) / (s2 - s1);
  p.y = (c.y * s2 - d.y * s1) / (s2 - s1);
}
__device__ __forceinline__ void pn_cut(float2* p, int& n, float2 a, float2 b, float2* pp) {
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
__device__ float pn_tri(float2 a, float2 b, float2 c, float2 d) {
  const float2 o = make_float2(0.f, 0.f);
  const int s1 = pn_sig(pn_cross(o, a, b)), s2 = pn_sig(pn_cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.f;
  if (s1 == -1) { const float2 t = a; a = b; b = t; }
  if (s2 == -1) { const float2 t = c; c = d; d = t; }
  float2 p[10], pp[10];
  for (int i = 0; i < 10; i++) p[i] = pp[i] = make_float2(0.f, 0.f);
  p[0] = o; p[1] = a; p[2] = b;
  int n = 3;
  pn_cut(p, n, o, c, pp);
  pn_cut(p, n, c, d, pp);
  pn_cut(p, n, d, o, pp);
  float res = fabsf(pn_area(p, n));
  if (s1 * s2 == -1) res = -res;
  return res;
}
__device__ float pn_iou(const float* __restrict__ p, const float* __restrict__ q) {
  float2 ps1[10], ps2[10];
  for (int i = 0; i < 4; i++) {
    ps1[i] = make_float2(p[2 * i], p[2 * i + 1]);
    ps2[i] = make_float2(q[2 * i], q[2 * i + 1]);
  }
  if (pn_area(ps1, 4) < 0) { float2 t = ps1[0]; ps1[0] = ps1[3]; ps1[3] = t; t = ps1[1]; ps1[1] = ps1[2]; ps1[2] = t; }
  if (pn_area(ps2, 4) < 0) { float2 t = ps2[0]; ps2[0] = ps2[3]; ps2[3] = t; t = ps2[1]; ps2[1] = ps2[2]; ps2[2] = t; }
  ps1[4] = ps1[0];
  ps2[4] = ps2[0];
  float inter = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) inter += pn_tri(ps1[i], ps1[i + 1], ps2[j], ps2[j + 1]);
  const float uni = fabsf(pn_area(ps1, 4)) + fabsf(pn_area(ps2, 4)) - inter;
  return uni == 0 ? (inter + 1) / (uni + 1) : inter / uni;
}

// poly_nms_kernel (:142-196): 64 x 64 tiles, upper triangle only (the greedy scan never reads below it)
__global__ __launch_bounds__(64) void poly_mask_kernel(const float* __restrict__ dets9,
                                                       const int64_t* __restrict__ order, int n, int cb, float thr,
                                                       u64* __restrict__ mask) {
  __shared__ float cols[64][8];
  const int rb = blockIdx.y, cblk = blockIdx.x, lane = threadIdx.x;
  if (cblk < rb) return;
  const int col_size = min(n - cblk * 64, 64);
  if (lane < col_size) {
    const float* s = dets9 + (size_t)order[cblk * 64 + lane] * 9;
    for (int k = 0; k < 8; k++) cols[lane][k] = s[k];
  }
  __syncthreads();
  const int row = rb * 64 + lane;
  if (row >= n) return;
  float me[8];
  const float* s = dets9 + (size_t)order[row] * 9;
  for (int k = 0; k < 8; k++) me[k] = s[k];
  u64 t = 0;
  for (int i = (rb == cblk) ? lane + 1 : 0; i < col_size; i++)
    if (pn_iou(me, cols[i]) > thr) t |= 1ULL << i;
  mask[(size_t)row * cb + cblk] = t;
}

__global__ __launch_bounds__(256) void poly_iou_mat_kernel(const float* __restrict__ a, int na, int sa,
                                                           const float* __restrict__ b, int nb, int sb,
                                                           float* __restrict__ out) {
  const long long total = (long long)na * nb;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t / nb), j = (int)(t - (long long)i * nb);
    out[t] = pn_iou(a + (size_t)i * sa, b + (size_t)j * sb);
  }
}

// ------------------------------------------------------------------------------------ convex_sort
// convex_sort_cuda (convex_cuda.cu:13-123) with its tensor prologue folded in: start = first argmin
// of the masked y, key = (x - sx) / sqrt((x - sx)^2 + (y - sy)^2 + 1e-6), points visited by STABLE
// descending key (the reference's argsort leaves the order of equal keys to the sort
// implementation), then the scan.  `work` = B x P ints (the visiting order).
__global__ __launch_bounds__(128) void convex_sort_kernel(const float* __restrict__ pts,
                                                          const unsigned char* __restrict__ masks, int B, int P,
                                                          int circular, int* __restrict__ work,
                                                          int64_t* __restrict__ out) {
  const int b = blockIdx.x * 128 + threadIdx.x;
  if (b >= B) return;
  const int isz = circular ? P + 1 : P;
  const float* p = pts + (size_t)b * P * 2;
  const unsigned char* m = masks + (size_t)b * P;
  int64_t* ci = out + (size_t)b * isz;
  int* ord = work + (size_t)b * P;
  for (int k = 0; k < isz; k++) ci[k] = -1;
  if (P == 0) return;
  int start = 0;
  float best = 0.f;
  for (int k = 0; k < P; k++) {
    const float mk = m[k] ? 1.f : 0.f;
    const float my = mk * p[2 * k + 1] + (1.f - mk) * 10000000.f;
    if (k == 0 || my < best) {
      best = my;
      start = k;
    }
  }
  const float sx = p[2 * start], sy = p[2 * start + 1];
  auto key = [&](int k) {
    const float dx = p[2 * k] - sx, dy = p[2 * k + 1] - sy;
    return dx / sqrtf(dx * dx + dy * dy + 0.000001f);
  };
  // rank by counting (P is small): position of k = #{q : key_q > key_k or (== and q < k)}
  for (int k = 0; k < P; k++) {
    const float kk = key(k);
    int r = 0;
    for (int q = 0; q < P; q++) {
      const float kq = key(q);
      r += (kq > kk) | ((kq == kk) & (q < k));
    }
    ord[r] = k;
  }
  ci[0] = start;
  int c_i = 0;
  for (int _j = 0; _j < P; _j++) {
    const int j = ord[_j];
    if (j == start || !m[j]) continue;
    const float x0 = p[2 * j], y0 = p[2 * j + 1];
    float x1 = p[2 * ci[c_i]], y1 = p[2 * ci[c_i] + 1];
    const float d = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
    if ((double)d < 0.000001) continue;
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

}  // namespace

int r3k_polygon_iou(const float* a, int na, const float* b, int nb, float* out, hipStream_t stream) {
  if (na < 0 || nb < 0) return -1;
  if (na == 0 || nb == 0) return 0;
  if (!a || !b || !out) return -1;
  long long blocks = ((long long)na * nb + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(polygon_iou_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, na, b, nb, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int r3k_poly_iou_mat(const float* a, int na, int sa, const float* b, int nb, int sb, float* out, hipStream_t stream) {
  if (na <= 0 || nb <= 0 || !a || !b || !out || sa < 8 || sb < 8) return -1;
  long long blocks = ((long long)na * nb + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(poly_iou_mat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, na, sa, b, nb, sb, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t r3k_poly_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  const size_t cb = (n + 63) / 64;
  return (size_t)n * cb * 8 + 256;
}

int r3k_poly_nms(const float* dets9, const int64_t* order, int n, float thr, void* ws, size_t ws_bytes,
                 int64_t* keep_out, int32_t* count_out, hipStream_t stream) {
  if (n < 0 || !count_out) return -1;
  if (n == 0) return r3k_zero_async(count_out, sizeof(int32_t), stream);
  if (!dets9 || !order || !ws || !keep_out) return -1;
  if (ws_bytes < r3k_poly_nms_workspace_bytes(n)) return -3;
  const int cb = (n + 63) / 64;
  u64* mask = (u64*)ws;
  hipLaunchKernelGGL(poly_mask_kernel, dim3(cb, cb), dim3(64), 0, stream, dets9, order, n, cb, thr, mask);
  return r3k_nms_reduce_dense(mask, n, cb, order, keep_out, count_out, stream);
}

int r3k_convex_sort(const float* pts, const unsigned char* masks, int B, int P, int circular, void* ws,
                    size_t ws_bytes, int64_t* out, hipStream_t stream) {
  if (B < 0 || P < 0) return -1;
  if (B == 0) return 0;
  if (!out || (P > 0 && (!pts || !masks || !ws))) return -1;
  if (ws_bytes < (size_t)B * P * sizeof(int)) return -3;
  hipLaunchKernelGGL(convex_sort_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, pts, masks, B, P, circular,
                     (int*)ws, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
