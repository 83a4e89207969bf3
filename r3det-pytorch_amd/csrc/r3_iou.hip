// r3_iou.hip -- rotated IoU matrix / vector kernels for gfx950.
//
// Matrix kernel (replaces mat_iou_iof_kernel rbbox_geo_kernel.cu:231-268 and
// box_iou_rotated_cuda_kernel box_iou_rotated_cuda.cu:14-63).
//
// Shape of the problem: in anchor assignment (K GT x 196 416 anchors) > 95 % of the pairs are
// disjoint and the kernel is bound by WRITING 4 bytes per pair; the few overlapping pairs cost
// ~1000x more ALU each.  A one-thread-per-pair kernel therefore idles 63 lanes of a wavefront
// whenever one lane clips.  Design:
//   * tile = 32 rows x 1024 columns per workgroup; lane <-> 4 adjacent columns, so a
//     wavefront writes 1 KB contiguous per output row with 16-byte stores (the reference maps
//     threadIdx.x to the ROW and writes with stride n2);
//   * phase A (streaming): per pair a conservative disjointness test (inflated circumscribed
//     circles + axis-aligned bounds) -> store 0; surviving pairs are NOT computed in place but
//     appended to an LDS work queue (wave-aggregated: one LDS atomic per wavefront);
//   * phase B (compacted): the queue is drained with all 256 lanes busy, one pair per lane,
//     candidate points in LDS ([slot][lane], conflict-free), result scattered with a 4-byte
//     store.  Every output element is written exactly once.
//   * row operand records are staged once per workgroup in LDS (broadcast reads); trig is
//     evaluated per box, never per pair.
#include <hip/hip_runtime.h>

#include "r3_geom_lds.h"
#include "r3_kernels.h"

namespace {

// ------------------------------------------------------------------ v1 kernel (simple, kept for A/B)
constexpr int IOU_BLOCK = 256;
constexpr int IOU_ROWS = 128;

template <int GEOM>
__global__ __launch_bounds__(IOU_BLOCK) void iou_mat_kernel(const float* __restrict__ b1, int n1,
                                                            const float* __restrict__ b2, int n2,
                                                            int iof, float* __restrict__ out) {
  __shared__ BoxRec rows[IOU_ROWS];
  const int col = blockIdx.x * IOU_BLOCK + threadIdx.x;
  const int row0 = blockIdx.y * IOU_ROWS;
  const int nrows = min(IOU_ROWS, n1 - row0);
  for (int r = threadIdx.x; r < nrows; r += IOU_BLOCK) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)(row0 + r) * 5, 0.f, rec);
    rows[r] = rec;
  }
  BoxRec mine;
  if (col < n2) make_record<GEOM>(b2 + (size_t)col * 5, 0.f, mine);
  __syncthreads();
  if (col >= n2) return;
  float* o = out + (size_t)row0 * n2 + col;
  for (int r = 0; r < nrows; r++) {
    const BoxRec& A = rows[r];
    float v;
    if (circles_apart(A.f[9], A.f[10], A.f[11], mine.f[9], mine.f[10], mine.f[11])) {
      v = 0.f;
    } else {
      BoxRec a = A;
      if (GEOM == 1) v = v1_pair_slow(a, mine, iof != 0);
      else if (GEOM == 2) v = hull_pair_slow<true>(a, mine, iof == 0);
      else v = hull_pair_slow<false>(a, mine, iof == 0);
    }
    o[(size_t)r * n2] = v;
  }
}

// ------------------------------------------------------------------ compacted kernel
// One launch, no workspace: tile = ROWS rows x (256 * CPT) columns per workgroup.  CPT / ROWS are picked per
// shape (launch_mat): wide tiles for wide matrices (16-byte stores), one column per lane and few rows per tile
// when the matrix is small, so that a 1000 x 128 call still spreads over ~125 workgroups.
constexpr int T_THREADS = 256;
constexpr int T_CPT = 4;                       // columns per thread (wide form)
constexpr int T_COLS = T_THREADS * T_CPT;      // 1024
constexpr int T_SUB = 8;                       // rows per phase-A/phase-B round

template <int GEOM, bool VEC, int CPT, int ROWS>
__global__ __launch_bounds__(T_THREADS) void iou_mat_compact_kernel(const float* __restrict__ b1, int n1,
                                                                    const float* __restrict__ b2, int n2,
                                                                    int iof, float* __restrict__ out) {
  constexpr int COLS = T_THREADS * CPT;
  constexpr int CSH = CPT == 4 ? 10 : (CPT == 2 ? 9 : 8);  // log2(COLS)
  constexpr int QCAP = T_SUB * COLS;                        // a round can never overflow
  __shared__ __attribute__((aligned(16))) float rows[ROWS][R3_REC];
  __shared__ unsigned short queue[QCAP];
  __shared__ int qcount[2];
  __shared__ float2 pts[pts_slots<GEOM>() * T_THREADS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int colbase = blockIdx.x * COLS;
  const int col0 = colbase + tid * CPT;
  const int row0 = blockIdx.y * ROWS;
  const int nrows = min(ROWS, n1 - row0);

  if (tid < nrows) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)(row0 + tid) * 5, 0.f, rec);
#pragma unroll
    for (int k = 0; k < R3_REC; k++) rows[tid][k] = rec.f[k];
  }
  if (tid < 2) qcount[tid] = 0;

  // reject data of my columns (cx, cy, radius, aabb half extents)
  float cq[CPT][5];
  bool cvalid[CPT];
#pragma unroll
  for (int c = 0; c < CPT; c++) {
    cvalid[c] = (col0 + c) < n2;
    if (cvalid[c]) {
      const float* b = b2 + (size_t)(col0 + c) * 5;
      float x = b[0], y = b[1], w = b[2], h = b[3], a = b[4];
      float s, co;
      r3_sincos(a, s, co);
      float ac = fabsf(co), as = fabsf(s), aw = 0.5f * fabsf(w), ah = 0.5f * fabsf(h);
      float slack = 2e-6f * (fabsf(x) + fabsf(y)) + 1e-6f;
      cq[c][0] = x;
      cq[c][1] = y;
      cq[c][2] = r3_radius(x, y, w, h);
      cq[c][3] = (ac * aw + as * ah) * 1.001f + slack;
      cq[c][4] = (as * aw + ac * ah) * 1.001f + slack;
    } else {
      cq[c][0] = cq[c][1] = cq[c][2] = cq[c][3] = cq[c][4] = 0.f;
    }
  }
  const bool all_valid = cvalid[CPT - 1];
  __syncthreads();

  const LanePts<T_THREADS> lp{pts + tid};
  for (int sub = 0; sub * T_SUB < nrows; sub++) {
    const int rbase = sub * T_SUB;
    const int rcount = min(T_SUB, nrows - rbase);
    int* qc = &qcount[sub & 1];
    // ---------------- phase A: stream zeros, enqueue survivors
    for (int r = 0; r < rcount; r++) {
      const float* A = rows[rbase + r];
      const float ax = A[9], ay = A[10], ar = A[11], aex = A[12], aey = A[13];
      bool pend[CPT];
      bool any = false;
#pragma unroll
      for (int c = 0; c < CPT; c++) {
        float dx = ax - cq[c][0], dy = ay - cq[c][1];
        float rr = ar + cq[c][2];
        bool apart = (dx * dx + dy * dy > rr * rr) | (fabsf(dx) > aex + cq[c][3]) |
                     (fabsf(dy) > aey + cq[c][4]);
        pend[c] = cvalid[c] && !apart;
        any |= pend[c];
      }
      float* o = out + (size_t)(row0 + rbase + r) * n2 + col0;
      if (VEC && CPT == 4 && all_valid && !any) {
        *reinterpret_cast<float4*>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int c = 0; c < CPT; c++)
          if (cvalid[c] && !pend[c]) o[c] = 0.f;
      }
      if (__ballot(any)) {
#pragma unroll
        for (int c = 0; c < CPT; c++) {
          unsigned long long m = __ballot(pend[c]);
          if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(qc, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (pend[c]) {
              int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
              queue[slot] = (unsigned short)((r << CSH) | (tid * CPT + c));
            }
          }
        }
      }
    }
    __syncthreads();
    // ---------------- phase B: drain the queue, one pair per lane
    const int total = *qc;
    if (tid == 0) qcount[(sub + 1) & 1] = 0;
    for (int q = tid; q < total; q += T_THREADS) {
      const unsigned e = queue[q];
      const int r = e >> CSH;
      const int col = colbase + (int)(e & (unsigned)(COLS - 1));
      BoxRec Bc;
      make_record<GEOM>(b2 + (size_t)col * 5, 0.f, Bc);
      const float v = pair_slow_lds<GEOM, T_THREADS>(rows[rbase + r], Bc.f, iof != 0, lp);
      out[(size_t)(row0 + rbase + r) * n2 + col] = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ global-queue pipeline
// The compacted kernel above balances lanes inside a workgroup, but pairs that survive the
// disjointness test are clustered (all 576 coarse-level anchors overlap every GT), so a few
// workgroups own most of the clipping and set the kernel's duration.  With a caller-provided
// workspace the work is split in two launches:
//   stream : phase A only -- zeros are streamed out, survivors go to a per-workgroup LDS
//            queue that is flushed to ONE global queue (one global atomic per workgroup);
//            workgroup row 0 / column 0 also publish the prepared box records;
//   drain  : a grid-stride loop over the global queue, one pair per lane, perfectly
//            balanced over the chip.
constexpr int S_ROWS = 16;
constexpr int S_QCAP = S_ROWS * T_COLS;  // 16384 entries (u16): a tile can never overflow

template <int GEOM, bool VEC>
__global__ __launch_bounds__(T_THREADS) void iou_stream_kernel(const float* __restrict__ b1, int n1,
                                                               const float* __restrict__ b2, int n2,
                                                               float* __restrict__ out,
                                                               BoxRec* __restrict__ recsA,
                                                               BoxRec* __restrict__ recsB,
                                                               unsigned* __restrict__ gqueue,
                                                               unsigned* __restrict__ counter) {
  __shared__ __attribute__((aligned(16))) float rows[S_ROWS][8];  // cx, cy, rad, ex, ey
  __shared__ unsigned short queue[S_QCAP];
  __shared__ int qcount;
  __shared__ unsigned qbase;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int colbase = blockIdx.x * T_COLS;
  const int col0 = colbase + tid * T_CPT;
  const int row0 = blockIdx.y * S_ROWS;
  const int nrows = min(S_ROWS, n1 - row0);

  if (tid < nrows) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)(row0 + tid) * 5, 0.f, rec);
#pragma unroll
    for (int k = 0; k < 5; k++) rows[tid][k] = rec.f[9 + k];
    if (blockIdx.x == 0) recsA[row0 + tid] = rec;
  }
  if (tid == 0) qcount = 0;

  float cq[T_CPT][5];
  bool cvalid[T_CPT];
#pragma unroll
  for (int c = 0; c < T_CPT; c++) {
    cvalid[c] = (col0 + c) < n2;
    if (cvalid[c]) {
      BoxRec rec;
      make_record<GEOM>(b2 + (size_t)(col0 + c) * 5, 0.f, rec);
#pragma unroll
      for (int k = 0; k < 5; k++) cq[c][k] = rec.f[9 + k];
      if (blockIdx.y == 0) recsB[col0 + c] = rec;
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) cq[c][k] = 0.f;
    }
  }
  const bool all_valid = cvalid[T_CPT - 1];
  __syncthreads();

  for (int r = 0; r < nrows; r++) {
    const float* A = rows[r];
    const float ax = A[0], ay = A[1], ar = A[2], aex = A[3], aey = A[4];
    bool pend[T_CPT];
    bool any = false;
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      float dx = ax - cq[c][0], dy = ay - cq[c][1];
      float rr = ar + cq[c][2];
      bool apart = (dx * dx + dy * dy > rr * rr) | (fabsf(dx) > aex + cq[c][3]) |
                   (fabsf(dy) > aey + cq[c][4]);
      pend[c] = cvalid[c] && !apart;
      any |= pend[c];
    }
    if (out) {  // (the fused assignment has no matrix: out == nullptr)
      float* o = out + (size_t)(row0 + r) * n2 + col0;
      if (VEC && all_valid && !any) {
        *reinterpret_cast<float4*>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int c = 0; c < T_CPT; c++)
          if (cvalid[c] && !pend[c]) o[c] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      unsigned long long m = __ballot(pend[c]);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&qcount, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (pend[c]) {
          int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
          queue[slot] = (unsigned short)((r << 10) | (tid * T_CPT + c));
        }
      }
    }
  }
  __syncthreads();
  const int total = qcount;
  if (total == 0) return;
  if (tid == 0) qbase = atomicAdd(counter, (unsigned)total);
  __syncthreads();
  const unsigned base = qbase;
  for (int q = tid; q < total; q += T_THREADS) {
    const unsigned e = queue[q];
    gqueue[base + q] = (unsigned)(row0 + (e >> 10)) * (unsigned)n2 + (unsigned)(colbase + (e & 1023u));
  }
}

// ------------------------------------------------------------------ prep + stream + drain pipeline
// The matrix path for large problems (the 128 x 196 416 assignment shape).
//   prep   : one small launch builds the row records and, ONCE per column, the 20 bytes the conservative test
//            reads (the round-1 stream kernel rebuilt them -- a double-precision sincos each -- in every one of
//            the n1/16 row tiles), laid out so that the stream kernel's per-lane loads are 64 + 16 contiguous
//            bytes, and zeroes the queue counter (no memset node);
//   stream : tile = SROWS rows x 1024 columns: conservative test -> zeros streamed out with 16-byte stores,
//            survivors -> LDS queue -> one of 64 bounded regions of the global queue (one atomicAdd per
//            workgroup on that region's counter); entries that no longer fit are clipped by the workgroup
//            itself, so the workspace is 0.5 B per pair, not 4 B;
//   drain  : grid-stride over the concatenation of the regions, one pair per lane, balanced over the chip.
// Measured and NOT shipped: a single persistent kernel in which workgroups alternate between
// streaming tiles and clipping chunks published by other workgroups (tickets in global memory).  On gfx950 an
// agent-scope release / acquire is an L2 write-back / invalidate of the whole XCD (the eight L2s are not
// coherent with each other), so every published tile flushed the freshly written zeros: 725 us instead of 75.
constexpr int F_NREG = 64;             // the global queue is split into 64 regions, each with its own counter:
constexpr int F_CSTRIDE = 32;          // 1536 workgroups adding to ONE address serialise (device-scope atomics are
constexpr int F_CTL = F_NREG * F_CSTRIDE;  // performed memory-side); counters sit 128 B apart

template <int GEOM>
__global__ __launch_bounds__(256) void iou_prep_kernel(const float* __restrict__ b1, int n1,
                                                       const float* __restrict__ b2, int n2,
                                                       BoxRec* __restrict__ recsA, float4* __restrict__ rejB,
                                                       float* __restrict__ radB, unsigned* __restrict__ ctl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < F_CTL) ctl[i] = 0;
  if (i < n1) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)i * 5, 0.f, rec);
    recsA[i] = rec;
  }
  if (i < n2) {
    // columns: only the 20 bytes the stream kernel tests with (coalesced 16-byte + 4-byte stores).  Their full
    // 64-byte records would be 12.5 MB of lane-strided stores at 196 416 anchors (13.5 us measured for this
    // launch); the drain kernel rebuilds the record of a surviving pair's column instead (~1.7 % of the pairs).
    const float* b = b2 + (size_t)i * 5;
    const float x = b[0], y = b[1], w = b[2], h = b[3];
    float sn, cs;
    r3_sincos(b[4], sn, cs);
    const float ac = fabsf(cs), as = fabsf(sn), aw = 0.5f * fabsf(w), ah = 0.5f * fabsf(h);
    const float slack = 2e-6f * (fabsf(x) + fabsf(y)) + 1e-6f;
    rejB[i] = make_float4(x, y, (ac * aw + as * ah) * 1.001f + slack, (as * aw + ac * ah) * 1.001f + slack);
    radB[i] = r3_radius(x, y, w, h);
  }
}

template <int GEOM, bool VEC, int SROWS>
__global__ __launch_bounds__(T_THREADS) void iou_stream2_kernel(const BoxRec* __restrict__ recsA, int n1,
                                                                const float* __restrict__ b2,
                                                                const float4* __restrict__ rejB,
                                                                const float* __restrict__ radB, int n2, int iof,
                                                                float* __restrict__ out, unsigned* __restrict__ counter,
                                                                unsigned* __restrict__ gqueue, unsigned qcap) {
  __shared__ __attribute__((aligned(16))) float rows[SROWS][8];  // cx, cy, rad, ex, ey
  __shared__ unsigned short queue[SROWS * T_COLS];
  __shared__ int qcount;
  __shared__ unsigned qbase;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int colbase = blockIdx.x * T_COLS;
  const int col0 = colbase + tid * T_CPT;
  const int row0 = blockIdx.y * SROWS;
  const int nrows = min(SROWS, n1 - row0);
  if (tid < nrows) {
    const float* a = recsA[row0 + tid].f;
#pragma unroll
    for (int k = 0; k < 5; k++) rows[tid][k] = a[9 + k];
  }
  if (tid == 0) qcount = 0;
  float4 cq[T_CPT];
  float cr[T_CPT];
  bool cvalid[T_CPT];
  if (col0 + T_CPT <= n2) {
    const float4 r4 = *reinterpret_cast<const float4*>(radB + col0);
    cr[0] = r4.x; cr[1] = r4.y; cr[2] = r4.z; cr[3] = r4.w;
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      cq[c] = rejB[col0 + c];
      cvalid[c] = true;
    }
  } else {
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      cvalid[c] = (col0 + c) < n2;
      cq[c] = cvalid[c] ? rejB[col0 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
      cr[c] = cvalid[c] ? radB[col0 + c] : 0.f;
    }
  }
  const bool all_valid = cvalid[T_CPT - 1];
  // Bounding box of this WAVE's 256 columns (inflated extents included).  Anchors and refined boxes come in
  // spatial order, so for most (row, wave) combinations the row's box lies outside it: one wave-uniform test
  // then replaces the 4 per-column tests (~14 VALU operations each: the stream kernel was co-limited by them,
  // 34 us against 15 us for a plain fill of the same 100 MB) and the row segment is stored as zeros.
  // Waves with an invalid or non-finite column never take the shortcut.
  float bx0 = cq[0].x - cq[0].z, bx1 = cq[0].x + cq[0].z, by0 = cq[0].y - cq[0].w, by1 = cq[0].y + cq[0].w;
  bool fin = all_valid;
#pragma unroll
  for (int c = 0; c < T_CPT; c++) {
    bx0 = fminf(bx0, cq[c].x - cq[c].z);
    bx1 = fmaxf(bx1, cq[c].x + cq[c].z);
    by0 = fminf(by0, cq[c].y - cq[c].w);
    by1 = fmaxf(by1, cq[c].y + cq[c].w);
    fin = fin && (fabsf(cq[c].x) < 3.0e38f) && (fabsf(cq[c].y) < 3.0e38f) && (cq[c].z < 3.0e38f) && (cq[c].w < 3.0e38f);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    bx0 = fminf(bx0, __shfl_xor(bx0, d));
    bx1 = fmaxf(bx1, __shfl_xor(bx1, d));
    by0 = fminf(by0, __shfl_xor(by0, d));
    by1 = fmaxf(by1, __shfl_xor(by1, d));
  }
  const bool wave_ok = VEC && (__ballot(fin) == ~0ULL);
  __syncthreads();
  for (int r = 0; r < nrows; r++) {
    const float* A = rows[r];
    const float ax = A[0], ay = A[1], ar = A[2], aex = A[3], aey = A[4];
    if (wave_ok && ((ax - aex > bx1) | (ax + aex < bx0) | (ay - aey > by1) | (ay + aey < by0))) {
      *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * n2 + col0) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    bool pend[T_CPT];
    bool any = false;
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      float dx = ax - cq[c].x, dy = ay - cq[c].y;
      float rr = ar + cr[c];
      bool apart = (dx * dx + dy * dy > rr * rr) | (fabsf(dx) > aex + cq[c].z) | (fabsf(dy) > aey + cq[c].w);
      pend[c] = cvalid[c] && !apart;
      any |= pend[c];
    }
    float* o = out + (size_t)(row0 + r) * n2 + col0;
    if (VEC && all_valid && !any) {
      *reinterpret_cast<float4*>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int c = 0; c < T_CPT; c++)
        if (cvalid[c] && !pend[c]) o[c] = 0.f;
    }
    if (__ballot(any)) {
#pragma unroll
      for (int c = 0; c < T_CPT; c++) {
        unsigned long long m = __ballot(pend[c]);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&qcount, __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          if (pend[c]) {
            int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
            queue[slot] = (unsigned short)((r << 10) | (tid * T_CPT + c));
          }
        }
      }
    }
  }
  __syncthreads();
  const int total = qcount;
  if (total == 0) return;
  const unsigned reg = (blockIdx.x + blockIdx.y * 7u) % (unsigned)F_NREG;  // heavy column tiles spread over regions
  if (tid == 0) qbase = atomicAdd(counter + reg * F_CSTRIDE, (unsigned)total);
  __syncthreads();
  const unsigned base = qbase;
  // entries [0, fit) go to this region of the global queue; the rest (only when the bounded region is full:
  // dense inputs) are clipped here, one pair per lane without LDS staging
  const int fit = base >= qcap ? 0 : (int)min((unsigned)total, qcap - base);
  unsigned* region = gqueue + (size_t)reg * qcap;
  for (int q = tid; q < fit; q += T_THREADS) {
    const unsigned e = queue[q];
    region[base + q] = (unsigned)(row0 + (int)(e >> 10)) * (unsigned)n2 + (unsigned)(colbase + (int)(e & 1023u));
  }
  for (int q = fit + tid; q < total; q += T_THREADS) {
    const unsigned e = queue[q];
    const unsigned r = (unsigned)row0 + (e >> 10), c = (unsigned)colbase + (e & 1023u);
    BoxRec A = recsA[r];
    BoxRec B;
    make_record<GEOM>(b2 + (size_t)c * 5, 0.f, B);
    float v;
    if (GEOM == 1) v = v1_pair_slow(A, B, iof != 0);
    else if (GEOM == 2) v = hull_pair_slow<true>(A, B, iof == 0);
    else v = hull_pair_slow<false>(A, B, iof == 0);
    out[(size_t)r * n2 + c] = v;
  }
}

template <int GEOM>
__global__ __launch_bounds__(T_THREADS) void iou_drain_kernel(const BoxRec* __restrict__ recsA,
                                                              const float* __restrict__ b2, int n2,
                                                              int iof, const unsigned* __restrict__ gqueue,
                                                              const unsigned* __restrict__ counter, unsigned qcap,
                                                              float* __restrict__ out) {
  __shared__ float2 pts[pts_slots<GEOM>() * T_THREADS];
  __shared__ unsigned pre[F_NREG + 1];  // exclusive prefix of the regions' (clamped) counts
  const LanePts<T_THREADS> lp{pts + threadIdx.x};
  if (threadIdx.x < 64) {
    unsigned v = min(counter[threadIdx.x * F_CSTRIDE], qcap);
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(incl, d);
      if ((int)threadIdx.x >= d) incl += t;
    }
    pre[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) pre[0] = 0;
  }
  __syncthreads();
  const unsigned total = pre[F_NREG];
  for (unsigned q = blockIdx.x * T_THREADS + threadIdx.x; q < total; q += gridDim.x * T_THREADS) {
    int lo = 0;  // region of entry q: largest lo with pre[lo] <= q
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1)
      if (pre[lo + step] <= q) lo += step;
    const unsigned e = gqueue[(size_t)lo * qcap + (q - pre[lo])];
    const unsigned r = e / (unsigned)n2;
    const unsigned c = e - r * (unsigned)n2;
    const BoxRec A = recsA[r];
    BoxRec B;
    make_record<GEOM>(b2 + (size_t)c * 5, 0.f, B);
    out[e] = pair_slow_lds<GEOM, T_THREADS>(A.f, B.f, iof != 0, lp);
  }
}

// ---------------------------------------------------------------------------------------------
// Fused assignment: what MaxIoUAssigner needs from the K x N overlap matrix of
// (gt, anchors) -- per-anchor max / argmax, per-gt max / argmax and the low-quality matches --
// without ever writing the matrix (100 MB at 128 x 196 416).  stream (no zero fill) -> drain:
// every clipped pair updates a packed (iou bits << 32 | ~index) key per column and per row with
// a 64-bit atomicMax (larger IoU wins, then the SMALLER index: the torch-CPU / numpy tie rule)
// and keeps its IoU next to its queue entry; a second sweep over the queue finds the pairs that
// equal their gt's maximum; the last kernel applies the thresholds.
typedef unsigned long long u64k;

__device__ __forceinline__ u64k pack_key(float iou, unsigned idx) {
  return ((u64k)__float_as_uint(iou) << 32) | (u64k)(0xffffffffu - idx);
}
__device__ __forceinline__ float key_iou(u64k k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned key_idx(u64k k) { return 0xffffffffu - (unsigned)(k & 0xffffffffu); }

__global__ __launch_bounds__(256) void assign_init_kernel(u64k* __restrict__ rowkey, int n1, u64k* __restrict__ colkey,
                                                          int* __restrict__ lowq, int n2,
                                                          unsigned* __restrict__ counter) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *counter = 0;
  if (i < n1) rowkey[i] = pack_key(0.f, 0);  // all-zero row / column: max 0 at index 0
  if (i < n2) {
    colkey[i] = pack_key(0.f, 0);
    lowq[i] = 0;
  }
}

template <int GEOM>
__global__ __launch_bounds__(T_THREADS) void assign_drain_kernel(const BoxRec* __restrict__ recsA,
                                                                 const BoxRec* __restrict__ recsB, int n2,
                                                                 const unsigned* __restrict__ gqueue,
                                                                 const unsigned* __restrict__ counter,
                                                                 float* __restrict__ qiou, u64k* __restrict__ rowkey,
                                                                 u64k* __restrict__ colkey, int n1_lds) {
  __shared__ float2 pts[pts_slots<GEOM>() * T_THREADS];
  extern __shared__ __attribute__((aligned(16))) u64k rowbest[];  // n1_lds entries (0 = none): per-workgroup row maxima
  const LanePts<T_THREADS> lp{pts + threadIdx.x};
  const unsigned total = *counter;
  for (int i = threadIdx.x; i < n1_lds; i += T_THREADS) rowbest[i] = 0;
  __syncthreads();
  for (unsigned q = blockIdx.x * T_THREADS + threadIdx.x; q < total; q += gridDim.x * T_THREADS) {
    const unsigned e = gqueue[q];
    const unsigned r = e / (unsigned)n2;
    const unsigned c = e - r * (unsigned)n2;
    const BoxRec A = recsA[r];
    const BoxRec B = recsB[c];
    const float v = pair_slow_lds<GEOM, T_THREADS>(A.f, B.f, false, lp);
    qiou[q] = v;
    if (v > 0.f) {
      // The few hundred gt rows take ~5 k updates each: global atomics on 128 addresses serialise
      // (0.6 ms).  Rows are reduced in LDS first and flushed once per workgroup; columns (anchors)
      // are many and rarely contended: look (the keys only grow), then atomicMax.
      const u64k kc = pack_key(v, r), kr = pack_key(v, c);
      if (kc > __builtin_nontemporal_load(&colkey[c])) atomicMax(&colkey[c], kc);
      if ((int)r < n1_lds) atomicMax(&rowbest[r], kr);
      else if (kr > __builtin_nontemporal_load(&rowkey[r])) atomicMax(&rowkey[r], kr);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n1_lds; i += T_THREADS) {
    const u64k k = rowbest[i];
    if (k && k > __builtin_nontemporal_load(&rowkey[i])) atomicMax(&rowkey[i], k);
  }
}

// pairs whose IoU equals their gt's maximum (MaxIoUAssigner step 4): the LAST such gt wins in the
// reference's sequential loop = the largest gt index = an atomicMax of (index + 1)
__global__ __launch_bounds__(256) void assign_lowq_kernel(const unsigned* __restrict__ gqueue,
                                                          const unsigned* __restrict__ counter,
                                                          const float* __restrict__ qiou, int n2,
                                                          const u64k* __restrict__ rowkey, float min_pos_iou,
                                                          int assign_all, int* __restrict__ lowq) {
  const unsigned total = *counter;
  for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < total; q += gridDim.x * 256) {
    const float v = qiou[q];
    if (!(v > 0.f)) continue;
    const unsigned e = gqueue[q];
    const unsigned r = e / (unsigned)n2;
    const unsigned c = e - r * (unsigned)n2;
    const u64k rk = rowkey[r];
    if (v == key_iou(rk) && v >= min_pos_iou && (assign_all || key_idx(rk) == c)) atomicMax(&lowq[c], (int)r + 1);
  }
}

__global__ __launch_bounds__(256) void assign_final_kernel(const u64k* __restrict__ rowkey, int n1,
                                                           const u64k* __restrict__ colkey,
                                                           const int* __restrict__ lowq, int n2, float pos_thr,
                                                           float neg_thr, float min_pos_iou, int match_low,
                                                           int assign_all, int64_t* __restrict__ assigned,
                                                           float* __restrict__ max_overlaps,
                                                           int64_t* __restrict__ argmax,
                                                           float* __restrict__ gt_max, int64_t* __restrict__ gt_argmax) {
  __shared__ int zmax;  // largest gt index whose best IoU is 0 (it matches every anchor with IoU 0 = all)
  if (threadIdx.x == 0) zmax = -1;
  __syncthreads();
  int z = -1;
  for (int i = threadIdx.x; i < n1; i += 256)
    if (key_iou(rowkey[i]) == 0.f) z = max(z, i);
  if (z >= 0) atomicMax(&zmax, z);
  __syncthreads();
  z = zmax;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n1 && gt_max) {
    gt_max[j] = key_iou(rowkey[j]);
    gt_argmax[j] = key_idx(rowkey[j]);
  }
  if (j >= n2) return;
  const u64k ck = colkey[j];
  const float mo = key_iou(ck);
  int a = -1;
  if (mo >= 0.f && mo < neg_thr) a = 0;
  if (mo >= pos_thr) a = (int)key_idx(ck) + 1;
  if (match_low) {
    int cand = lowq[j];
    // a gt with no overlap at all has max 0 >= min_pos_iou when min_pos_iou <= 0: `overlaps[i] == 0`
    // then selects every anchor (or, without gt_max_assign_all, its argmax = anchor 0)
    if (z >= 0 && 0.f >= min_pos_iou && (assign_all || j == 0)) cand = max(cand, z + 1);
    if (cand > 0) a = cand;
  }
  assigned[j] = a;
  max_overlaps[j] = mo;
  if (argmax) argmax[j] = key_idx(ck);
}

// vec_iou_iof_kernel (rbbox_geo_kernel.cu:271-309): out[i] = f(b1[i % n1], b2[i % n2]).
template <int GEOM>
__global__ __launch_bounds__(256) void iou_vec_kernel(const float* __restrict__ b1, int n1,
                                                      const float* __restrict__ b2, int n2,
                                                      int iof, float* __restrict__ out) {
  __shared__ float2 pts[pts_slots<GEOM>() * 256];
  const LanePts<256> lp{pts + threadIdx.x};
  const int n = max(n1, n2);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    BoxRec A, B;
    make_record<GEOM>(b1 + (size_t)(i % n1) * 5, 0.f, A);
    make_record<GEOM>(b2 + (size_t)(i % n2) * 5, 0.f, B);
    float v = 0.f;
    if (!boxes_apart(A.f, B.f)) v = pair_slow_lds<GEOM, 256>(A.f, B.f, iof != 0, lp);
    out[i] = v;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PipeLayout {
  unsigned* ctl;
  BoxRec* recsA;
  float4* rejB;
  float* radB;
  unsigned* gqueue;
  unsigned qcap;
};

// bounded global queue: 64 regions of pairs / 512 entries each = one eighth of the pairs in total (assignment
// shapes have < 2 % survivors; a workgroup whose region is full clips in place), at least 1 K entries per region
inline size_t pipe_layout(int n1, int n2, void* ws, PipeLayout* L) {
  const unsigned long long pairs = (unsigned long long)n1 * (unsigned long long)n2;
  unsigned long long qcap = pairs / (8 * F_NREG);
  if (qcap < 1024) qcap = 1024;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* ctl = take((size_t)F_CTL * 4);
  char* ra = take((size_t)n1 * sizeof(BoxRec));
  char* rj = take((size_t)n2 * 16);
  char* rd = take((size_t)n2 * 4 + 16);
  char* gq = take((size_t)qcap * F_NREG * 4);
  if (L) {
    L->ctl = (unsigned*)ctl; L->recsA = (BoxRec*)ra; L->rejB = (float4*)rj; L->radB = (float*)rd;
    L->gqueue = (unsigned*)gq; L->qcap = (unsigned)qcap;
  }
  return off + 256;
}

template <int GEOM, int CPT, int ROWS>
void launch_compact(bool vec, int iof, const float* b1, int n1, const float* b2, int n2, float* out, hipStream_t stream) {
  dim3 grid((n2 + T_THREADS * CPT - 1) / (T_THREADS * CPT), (n1 + ROWS - 1) / ROWS);
  if (vec)
    hipLaunchKernelGGL((iou_mat_compact_kernel<GEOM, true, CPT, ROWS>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2, iof, out);
  else
    hipLaunchKernelGGL((iou_mat_compact_kernel<GEOM, false, CPT, ROWS>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2, iof, out);
}

template <int GEOM>
int launch_mat(int iof, const float* b1, int n1, const float* b2, int n2, float* out, void* ws,
               size_t ws_bytes, hipStream_t stream) {
  const bool vec = (n2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (g_r3_iou_impl == 1) {
    dim3 grid((n2 + IOU_BLOCK - 1) / IOU_BLOCK, (n1 + IOU_ROWS - 1) / IOU_ROWS);
    hipLaunchKernelGGL(iou_mat_kernel<GEOM>, grid, dim3(IOU_BLOCK), 0, stream, b1, n1, b2, n2, iof, out);
    return 0;
  }
  const unsigned long long pairs = (unsigned long long)n1 * (unsigned long long)n2;
  const bool fits32 = pairs < 0xffffffffULL;
  // Wide matrices (assignment shapes: few gt rows x tens of thousands of anchors, < 2 % overlapping pairs) take
  // the prep + stream + drain pipeline.  Everything else is ONE launch (no prep, no queue): tiles narrow enough
  // that the grid still fills the CUs; dense square problems (2000 x 2000: 129 us here, 329 us through the
  // pipeline, whose drain then carries most pairs) and small ones (1000 x 128: 17 us) both prefer it.
  const int wide = g_r3_iou_small > 0 ? g_r3_iou_small : 16384;
  const bool piped = ws && fits32 && ws_bytes >= r3k_iou_workspace_bytes(n1, n2) && g_r3_iou_impl != 2 &&
                     (n2 >= wide || g_r3_iou_impl == 4);
  if (!piped) {
    if (g_r3_iou_impl == 2 || pairs > 2000000ULL) launch_compact<GEOM, 4, 32>(vec, iof, b1, n1, b2, n2, out, stream);
    else if (n2 <= 512) launch_compact<GEOM, 1, 8>(vec, iof, b1, n1, b2, n2, out, stream);
    else launch_compact<GEOM, 4, 8>(vec, iof, b1, n1, b2, n2, out, stream);
    return 0;
  }
  PipeLayout L;
  pipe_layout(n1, n2, ws, &L);
  const int nmax = (n1 > n2 ? n1 : n2) > F_CTL ? (n1 > n2 ? n1 : n2) : F_CTL;
  hipLaunchKernelGGL(iou_prep_kernel<GEOM>, dim3((nmax + 255) / 256), dim3(256), 0, stream, b1, n1, b2, n2, L.recsA,
                     L.rejB, L.radB, L.ctl);
  const unsigned qcap = g_r3_iou_qcap > 0 && (unsigned)g_r3_iou_qcap < L.qcap ? (unsigned)g_r3_iou_qcap : L.qcap;
  constexpr int SR = 16;
  dim3 grid((n2 + T_COLS - 1) / T_COLS, (n1 + SR - 1) / SR);
  if (vec)
    hipLaunchKernelGGL((iou_stream2_kernel<GEOM, true, SR>), grid, dim3(T_THREADS), 0, stream, L.recsA, n1, b2, L.rejB,
                       L.radB, n2, iof, out, L.ctl, L.gqueue, qcap);
  else
    hipLaunchKernelGGL((iou_stream2_kernel<GEOM, false, SR>), grid, dim3(T_THREADS), 0, stream, L.recsA, n1, b2, L.rejB,
                       L.radB, n2, iof, out, L.ctl, L.gqueue, qcap);
  // drain: enough workgroups to fill the chip at the kernel's occupancy; grid-stride inside
  int blocks = (int)((pairs + T_THREADS - 1) / T_THREADS);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(iou_drain_kernel<GEOM>, dim3(blocks), dim3(T_THREADS), 0, stream, L.recsA, b2, n2, iof, L.gqueue,
                     L.ctl, qcap, out);
  return 0;
}

}  // namespace

size_t r3k_iou_workspace_bytes(int n1, int n2) {
  if (n1 <= 0 || n2 <= 0) return 256;
  return pipe_layout(n1, n2, nullptr, nullptr);
}

int r3k_iou_mat(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n1 == 0 || n2 == 0) return 0;
  int rc;
  switch (geom) {
    case 1: rc = launch_mat<1>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream); break;
    case 2: rc = launch_mat<2>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream); break;
    case 3: rc = launch_mat<3>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream); break;
    default: return -1;
  }
  if (rc) return rc;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

namespace {

struct AssignLayout {
  unsigned* counter;
  BoxRec *recsA, *recsB;
  unsigned* gqueue;
  float* qiou;
  u64k *rowkey, *colkey;
  int* lowq;
};

inline size_t assign_layout(int n1, int n2, void* ws, AssignLayout* L) {
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* counter = take(256);
  char* ra = take((size_t)n1 * sizeof(BoxRec));
  char* rb = take((size_t)n2 * sizeof(BoxRec));
  char* gq = take((size_t)n1 * n2 * 4);
  char* qi = take((size_t)n1 * n2 * 4);
  char* rk = take((size_t)n1 * 8);
  char* ck = take((size_t)n2 * 8);
  char* lq = take((size_t)n2 * 4);
  if (L) {
    L->counter = (unsigned*)counter; L->recsA = (BoxRec*)ra; L->recsB = (BoxRec*)rb; L->gqueue = (unsigned*)gq;
    L->qiou = (float*)qi; L->rowkey = (u64k*)rk; L->colkey = (u64k*)ck; L->lowq = (int*)lq;
  }
  return off + 256;
}

template <int GEOM>
void launch_assign(const float* gts, int n1, const float* boxes, int n2, const AssignLayout& L, float min_pos_iou,
                   int match_low, int assign_all, hipStream_t stream) {
  const int nmax = n1 > n2 ? n1 : n2;
  hipLaunchKernelGGL(assign_init_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, L.rowkey, n1, L.colkey,
                     L.lowq, n2, L.counter);
  dim3 grid((n2 + T_COLS - 1) / T_COLS, (n1 + S_ROWS - 1) / S_ROWS);
  hipLaunchKernelGGL((iou_stream_kernel<GEOM, false>), grid, dim3(T_THREADS), 0, stream, gts, n1, boxes, n2,
                     (float*)nullptr, L.recsA, L.recsB, L.gqueue, L.counter);
  unsigned long long pairs = (unsigned long long)n1 * n2;
  int blocks = (int)((pairs + T_THREADS - 1) / T_THREADS);
  if (blocks > 1024) blocks = 1024;
  const int n1_lds = n1 < 2048 ? n1 : 2048;  // rows reduced in LDS (16 KB); the rest goes straight to global
  hipLaunchKernelGGL(assign_drain_kernel<GEOM>, dim3(blocks < 512 ? blocks : 512), dim3(T_THREADS),
                     (size_t)n1_lds * sizeof(u64k), stream, L.recsA, L.recsB, n2, L.gqueue, L.counter, L.qiou, L.rowkey,
                     L.colkey, n1_lds);
  if (match_low)
    hipLaunchKernelGGL(assign_lowq_kernel, dim3(blocks), dim3(256), 0, stream, L.gqueue, L.counter, L.qiou, n2,
                       L.rowkey, min_pos_iou, assign_all, L.lowq);
}

}  // namespace

size_t r3k_iou_assign_workspace_bytes(int n1, int n2) {
  if (n1 <= 0 || n2 <= 0) return 256;
  return assign_layout(n1, n2, nullptr, nullptr);
}

int r3k_iou_assign(int geom, const float* gts, int n1, const float* boxes, int n2, float pos_thr, float neg_thr,
                   float min_pos_iou, int match_low, int assign_all, int64_t* assigned, float* max_overlaps,
                   int64_t* argmax, float* gt_max, int64_t* gt_argmax, void* ws, size_t ws_bytes,
                   hipStream_t stream) {
  if (n1 <= 0 || n2 <= 0 || !gts || !boxes || !assigned || !max_overlaps || !ws) return -1;
  if ((gt_max == nullptr) != (gt_argmax == nullptr)) return -1;
  if ((unsigned long long)n1 * (unsigned long long)n2 >= 0xffffffffULL) return -1;
  if (ws_bytes < r3k_iou_assign_workspace_bytes(n1, n2)) return -3;
  AssignLayout L;
  assign_layout(n1, n2, ws, &L);
  switch (geom) {
    case 1: launch_assign<1>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream); break;
    case 2: launch_assign<2>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream); break;
    case 3: launch_assign<3>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream); break;
    default: return -1;
  }
  const int nmax = n1 > n2 ? n1 : n2;
  hipLaunchKernelGGL(assign_final_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, L.rowkey, n1, L.colkey, L.lowq,
                     n2, pos_thr, neg_thr, min_pos_iou, match_low, assign_all, assigned, max_overlaps, argmax, gt_max,
                     gt_argmax);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int r3k_iou_vec(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                hipStream_t stream) {
  if (n1 == 0 || n2 == 0) return 0;
  int n = n1 > n2 ? n1 : n2;
  dim3 grid(min((n + 255) / 256, 2048)), block(256);
  switch (geom) {
    case 1: hipLaunchKernelGGL(iou_vec_kernel<1>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 2: hipLaunchKernelGGL(iou_vec_kernel<2>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 3: hipLaunchKernelGGL(iou_vec_kernel<3>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
