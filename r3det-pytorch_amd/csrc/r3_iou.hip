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

#include <algorithm>

#include "r3_clip.h"
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
  // (round 5: the candidate list in LDS, [slot][lane], as everywhere else -- the register form of r3_geom.h took
  // 416-640 B of scratch per lane here)
  __shared__ float2 pts[pts_slots<GEOM>() * IOU_BLOCK];
  const LanePts<IOU_BLOCK> lp{pts + threadIdx.x};
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
      v = pair_slow_lds<GEOM, IOU_BLOCK>(A.f, mine.f, iof != 0, lp);
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

// LDS of one tile routine call: rows[ROWS][R3_REC] floats, queue[SUB * 256 * CPT] u16 (a round can never overflow),
// qcount[2], pts[pts_slots * 256]
template <int GEOM, bool VEC, int CPT, int ROWS, int SUB>
__device__ __forceinline__ void compact_tile(const int bx, const int by, const float* __restrict__ b1, int n1,
                                             const float* __restrict__ b2, int n2, int iof,
                                             float* __restrict__ out, float (*rows)[R3_REC], unsigned short* queue,
                                             int* qcount, float2* pts) {
  constexpr int COLS = T_THREADS * CPT;
  constexpr int CSH = CPT == 4 ? 10 : (CPT == 2 ? 9 : 8);  // log2(COLS)
  constexpr int T_SUB = SUB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int colbase = bx * COLS;
  const int col0 = colbase + tid * CPT;
  const int row0 = by * ROWS;
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
    const int wv = tid >> 6;
    for (int qb = wv * 64; qb < total; qb += T_THREADS) {  // (wave-uniform trips: the hull form's redo is a wave-level pass)
      const int q = qb + lane;
      const bool valid = q < total;
      const unsigned e = valid ? queue[q] : 0u;
      const int r = e >> CSH;
      const int col = colbase + (int)(e & (unsigned)(COLS - 1));
      if constexpr (GEOM == 1) {  // round 5: the straight-line clip; what it flags takes the exact list form (same region)
        if (valid) {
          BoxRec Bc;
          make_record<GEOM>(b2 + (size_t)col * 5, 0.f, Bc);
          bool flagged = false;
          float v = v1_clip_fast(rows[rbase + r], Bc.f, iof != 0, ClipLds<T_THREADS>{pts + tid}, flagged);
          if (flagged) v = pair_slow_lds<GEOM, T_THREADS>(rows[rbase + r], Bc.f, iof != 0, lp);
          out[(size_t)(row0 + rbase + r) * n2 + col] = v;
        }
      } else {
        // hull: 12 of the 24 slots in the wave's private region; a pair with a 13th point is redone by lanes 0..31
        // with 24 slots in the same region (round 5: 49 KB -> 24.5 KB of LDS per workgroup)
        bool over = false;
        if (valid) {
          BoxRec Bc;
          make_record<GEOM>(b2 + (size_t)col * 5, 0.f, Bc);
          // (round 5: the straight-line hull clip; what it flags takes the redo below)
          const float v = hull_clip_fast<GEOM == 2>(rows[rbase + r], Bc.f, iof == 0, ClipLds<64>{pts + wv * (64 * 12) + lane}, over);
          if (!over) out[(size_t)(row0 + rbase + r) * n2 + col] = v;
        }
        unsigned long long m = __ballot(over);
        while (m) {
          int src = -1, seen = 0;
          for (unsigned long long t2 = m; t2; t2 &= t2 - 1) {
            if (seen == lane) src = __builtin_ctzll(t2);
            seen++;
          }
          const int rr = __shfl(r, src < 0 ? 0 : src), cc = __shfl(col, src < 0 ? 0 : src);
          if (lane < 32 && src >= 0) {
            BoxRec Bc;
            make_record<GEOM>(b2 + (size_t)cc * 5, 0.f, Bc);
            out[(size_t)(row0 + rbase + rr) * n2 + cc] =
                hull_pair_lds<GEOM == 2, 32, 24>(rows[rbase + rr], Bc.f, iof == 0, LanePts<32>{pts + wv * (64 * 12) + lane});
          }
          for (int k = 0; k < 32 && m; k++) m &= m - 1;
        }
      }
    }
    __syncthreads();
  }
}

// (R3_COMPACT_WAVES / R3_VEC_WAVES: minimum waves per SIMD asked of the compiler for the v1 forms.  Without a bound they
// take 185 / 188 registers = occupancy 2; bounded to 3 they compile to 168 registers with no scratch.  Measured
// (profiles/r06_iou_occupancy_ab.txt): the one-launch tile kernel at 1000 x 128 9.94 us either way -- its 125 workgroups
// are a latency chain, not an occupancy problem --, the aligned form over 196 416 pairs 13.1 -> 12.4 us.  The 32-row
// tile form (212 registers) keeps its occupancy 2: bounded it spills 180 B / lane.)
#ifndef R3_COMPACT_WAVES
#define R3_COMPACT_WAVES 3
#endif
#ifndef R3_VEC_WAVES
#define R3_VEC_WAVES 3
#endif
template <int GEOM, bool VEC, int CPT, int ROWS>
__global__ __launch_bounds__(T_THREADS, (GEOM == 1 && ROWS == 8) ? R3_COMPACT_WAVES : 1) void iou_mat_compact_kernel(const float* __restrict__ b1, int n1,
                                                                    const float* __restrict__ b2, int n2,
                                                                    int iof, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float rows[ROWS][R3_REC];
  __shared__ unsigned short queue[T_SUB * T_THREADS * CPT];
  __shared__ int qcount[2];
  __shared__ float2 pts[(GEOM == 1 ? R3_V1_CAP : 12) * T_THREADS];
  compact_tile<GEOM, VEC, CPT, ROWS, T_SUB>(blockIdx.x, blockIdx.y, b1, n1, b2, n2, iof, out, rows, queue, qcount, pts);
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
                                                               unsigned* __restrict__ counter,
                                                               const float4* __restrict__ prej = nullptr,
                                                               const float* __restrict__ prad = nullptr) {
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
    if (cvalid[c] && prej) {  // prepared columns: the records exist (recsB IS the prepared array)
      const float4 q = prej[col0 + c];
      cq[c][0] = q.x, cq[c][1] = q.y, cq[c][2] = prad[col0 + c], cq[c][3] = q.z, cq[c][4] = q.w;
    } else if (cvalid[c]) {
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

// ------------------------------------------------------------------ stream + drain pipeline
// The matrix path for wide problems (the 128 x 196 416 assignment shape).  TWO launches and nothing in front of
// them: on this part a launch that does nothing still occupies ~4.5 us of the stream (measured: an empty
// 512-workgroup kernel), so the prep kernel (records + counter zeroing, 5 us), and the overflow kernel (4.4 us,
// normally empty) of the earlier forms were a sixth of the op.
//   stream : tile = 8 rows x 1024 columns (measured: 4 rows 24.5 us, 8 rows 21.5, 16 rows 24, 32 rows 29).  The
//            conservative-test data (centre, circumscribed radius, AABB half extents) is derived in the kernel from the raw boxes with hardware sine / cosine -- the test only
//            has to be conservative, so the extents are inflated by the approximation's error bound instead of
//            evaluating the exact double-precision sincos; zeros are streamed out with 16-byte stores;
//            survivors -> per-wave LDS segments (no atomics) -> the tile's OWN slot of the workspace (u16
//            tile-local entries, 1 B per pair of workspace) + the tile's count.  No global atomics, so nothing
//            has to be zeroed first.  A tile with a wave more than half full is marked dense (count -1).
//   drain  : every workgroup rebuilds the prefix over the tile counts (grouped, <= 1024 groups in LDS), then
//            grid-strides over the concatenated entries, one pair per lane, balanced over the chip; a dense
//            tile counts as its 8 x 1024 pairs, each tested with the exact records and clipped or zeroed.
// Measured and NOT shipped (DESIGN_HISTORY 4.1): a single persistent kernel in which workgroups alternate between
// streaming tiles and clipping chunks published by other workgroups (tickets in global memory).  On gfx950 an
// agent-scope release / acquire is an L2 write-back / invalidate of the whole XCD (the eight L2s are not
// coherent with each other), so every published tile flushed the freshly written zeros: 725 us instead of 75.
constexpr int P_ROWS = 8;  // (rows per tile, measured at 128 x 196 416 with prepared columns: 4 -> 52, 8 -> 48, 16 -> 50, 32 -> 55 us)
constexpr int P_WSEG = P_ROWS * 128;   // entries of a wave's LDS segment: half of its 8 x 256 pairs
constexpr int P_SLOT = 4 * P_WSEG;     // u16 entries of a tile's slot in the workspace (16 KB)
constexpr int P_GROUPS = 3072;         // tile groups whose prefix a drain workgroup keeps in LDS (12 KB; round 5: 1024 -> 3072, so
                                       // that at 128 x 196 416 a group IS a tile -- no walk through global counts behind the search)
constexpr int P_GPT = P_GROUPS / T_THREADS;  // groups per thread of the prefix phase
static_assert(P_GROUPS % (4 * T_THREADS) == 0, "a thread's groups are whole int4 loads");
// ints of the tile-count array: the prefix phase reads P_GROUPS * G of them (those beyond the tiles count as empty)
__host__ __device__ inline size_t tcount_ints(long long tiles) {
  const long long G = (tiles + P_GROUPS - 1) / P_GROUPS;
  return (size_t)(tiles > P_GROUPS * G ? tiles : P_GROUPS * G);
}
// ... and behind them the drain's TICKET: a drain wavefront's first 64 entries are its own (wave index), every further
// block of 64 is handed out by an atomic counter, zeroed by the stream kernel (option iou_dyn 0: the static stride of the
// first half of the round).  Static shares are equal in entries, not in time: clips differ (rare block, redo), and a SIMD
// whose four wavefronts drew the long ones was the kernel's tail.
constexpr int P_TICKETS = 64;      // counters, 128 B apart: wavefront w draws from counter w % 64 (ONE counter for the 7 k blocks of
                                   // 128 x 196 416 made the drain 93 us instead of 23: a same-address returning atomic is ~10 ns, in turn)
constexpr int P_TICKET_PAD = P_TICKETS * 32;  // ints kept behind the counts
constexpr int P_MAX_TILES = 16384;     // beyond: the one-launch form (a group would span > 16 tiles)

// centre / radius / AABB half extents for the conservative test, from hardware sine / cosine (|error| < 1e-3 on
// |angle| < 64; larger finite angles use the circle's AABB; non-finite angles give NaN = "never apart")
__device__ __forceinline__ void reject_data(const float x, const float y, const float w, const float h, const float a,
                                            float& rad, float& ex, float& ey) {
  // (r3_radius with the hardware square root: 1 ulp against the 1e-3 inflation, a denormal sum of squares against the
  // absolute 1e-6 -- the correctly rounded sqrtf is ~20 instructions, 5 of them per thread were a third of the prologue)
  const float slack = 2e-6f * (fabsf(x) + fabsf(y)) + 1e-6f;
  rad = 0.5f * __builtin_amdgcn_sqrtf(w * w + h * h) * 1.001f + slack;
  const float aw = 0.5f * fabsf(w), ah = 0.5f * fabsf(h);
  if (fabsf(a) < 64.f) {
    const float ac = fabsf(__cosf(a)) + 1e-3f, as = fabsf(__sinf(a)) + 1e-3f;
    ex = (ac * aw + as * ah) * 1.001f + slack;
    ey = (as * aw + ac * ah) * 1.001f + slack;
  } else {
    ex = ey = (fabsf(a) < 3.0e38f) ? rad : __builtin_nanf("");
  }
}

// ---- prepared columns.  The second operand of the assignment is the same anchor grid in every step
// (rotate_anchor_head.py:220-231 -> rbbox_geo_kernel.cu:231-268 recomputes everything about it per pair): its exact
// records (deterministic sincos, 64 B), the conservative-test data of its columns (centre + AABB half extents, radius)
// and the bounding box of every 256 consecutive columns (a stream wavefront's columns) are computed ONCE
// (r3det_iou_prepare_columns) and read back by every call -- the stream kernel's per-wavefront prologue was 45 % of its
// instructions, the drain rebuilt a column's record for every surviving pair.
// Round 6: the buffer begins with a 16-byte HEADER {magic, geometry, n, magic ^ geometry ^ n} written by the prepare
// kernel and looked at by every workgroup of the consumers' stream kernel (one scalar load): a buffer prepared for another
// geometry or another column count -- or not prepared at all -- is refused ON THE DEVICE, before a byte of it is used:
// the matrix form fills its tile with NaN, the assignment's keys become NaN (max_overlaps NaN, every anchor ignored).
// No host-side state (rounds 4-5 kept a process-global map keyed by the device address, which a freed and reused
// address could fool: ADVICE r5); r3det_iou_prepared_check copies the header back for callers who want a host answer.
constexpr int COLPREP_MAGIC = 0x52335043;  // "R3PC"
struct ColPrep {
  const int4* hdr;     // {magic, geom, n, magic ^ geom ^ n}
  const BoxRec* rec;   // [n2]
  const float4* rej;   // [n2] cx, cy, inflated AABB half extents (= rec.f[9], f[10], f[12], f[13])
  const float* rad;    // [n2] inflated circumscribed radius (= rec.f[11])
  const float4* wbox;  // [ceil(n2 / 256)] min x, max x, min y, max y over the 256 columns; NaN: no shortcut
};

inline size_t colprep_layout(int n2, const void* p, ColPrep* L) {
  const size_t nw = ((size_t)n2 + 255) / 256;
  const size_t o_rec = 256, o_rej = (o_rec + (size_t)n2 * sizeof(BoxRec) + 255) & ~(size_t)255;
  const size_t o_rad = (o_rej + (size_t)n2 * 16 + 255) & ~(size_t)255, o_wb = (o_rad + (size_t)n2 * 4 + 255) & ~(size_t)255;
  if (L) {
    const char* c = static_cast<const char*>(p);
    L->hdr = reinterpret_cast<const int4*>(c);
    L->rec = reinterpret_cast<const BoxRec*>(c + o_rec);
    L->rej = reinterpret_cast<const float4*>(c + o_rej);
    L->rad = reinterpret_cast<const float*>(c + o_rad);
    L->wbox = reinterpret_cast<const float4*>(c + o_wb);
  }
  return o_wb + nw * 16 + 256;
}

template <int GEOM>
__global__ __launch_bounds__(256) void iou_prepare_kernel(const float* __restrict__ b2, int n2, BoxRec* __restrict__ rec,
                                                          float4* __restrict__ rej, float* __restrict__ rad,
                                                          float4* __restrict__ wbox, int4* __restrict__ hdr) {
  __shared__ float red[4][4];
  __shared__ int bad;
  const int c = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) *hdr = make_int4(COLPREP_MAGIC, GEOM, n2, COLPREP_MAGIC ^ GEOM ^ n2);
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  float x0 = 3.0e38f, x1 = -3.0e38f, y0 = 3.0e38f, y1 = -3.0e38f;
  bool fin = false;
  if (c < n2) {
    BoxRec r;
    make_record<GEOM>(b2 + (size_t)c * 5, 0.f, r);
    rec[c] = r;
    rej[c] = make_float4(r.f[9], r.f[10], r.f[12], r.f[13]);
    rad[c] = r.f[11];
    x0 = r.f[9] - r.f[12], x1 = r.f[9] + r.f[12], y0 = r.f[10] - r.f[13], y1 = r.f[10] + r.f[13];
    fin = (fabsf(r.f[9]) < 3.0e38f) && (fabsf(r.f[10]) < 3.0e38f) && (r.f[12] < 3.0e38f) && (r.f[13] < 3.0e38f);
  }
  if (!fin) bad = 1;  // (a column beyond the list, or a non-finite one: its 256 columns never take the shortcut)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, d));
    x1 = fmaxf(x1, __shfl_xor(x1, d));
    y0 = fminf(y0, __shfl_xor(y0, d));
    y1 = fmaxf(y1, __shfl_xor(y1, d));
  }
  if (lane == 0) red[wave][0] = x0, red[wave][1] = x1, red[wave][2] = y0, red[wave][3] = y1;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nan = __builtin_nanf("");
    float4 b = make_float4(fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])),
                           fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1])),
                           fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2])),
                           fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])));
    if (bad) b = make_float4(nan, nan, nan, nan);
    wbox[blockIdx.x] = b;
  }
}

// The fused assignment's key arrays, zeroed by the stream kernel's first row / column of workgroups (the drain, a later
// launch, is their first user): all-zero row / column = maximum 0 at index 0 = key 0x00000000ffffffff (pack_key(0, 0)).
struct AssignZero {
  unsigned long long* rowkey;  // [n1] or nullptr
  unsigned long long* colkey;  // [n2]
  int* lowq;                   // [n2]
};

// BITS (round 6, option iou_impl 5; VEC only): the tests alone -- no matrix stores -- plus, per tile and lane, which of
// the lane's 8 x 4 elements SURVIVED (bit 4 r + c of bits[tile * 256 + tid]; a dense tile: all ones).  The zeros are
// then written by the drain launch's fill blocks, which skip exactly those elements: fill and clip write disjoint
// addresses and can share one launch with no ordering between them.
// obb_overlaps' rule (box_iou_rotated_wrapper.py:53-60): a box with min(w, h) < 1e-3 has an all-zero row / column.
// torch.min propagates NaN and NaN < 1e-3 is false: a box with a NaN side is not thin.
__device__ __forceinline__ bool iou_thin_box(const float w, const float h) {
  return !(w != w || h != h) && fminf(w, h) < 0.001f;
}

// thin (GEOM 3, round 6): obb_overlaps' epilogue inside the pipeline -- a thin row is skipped, a thin column never
// survives, so their elements keep the tile's zeros and the separate epilogue launch (4.7 us at 128 x 196 416) is not
// needed; the drain applies the same rule to the pairs of a dense tile, which it enumerates itself.
template <int GEOM, bool VEC, bool PREP = false, bool BITS = false>
__global__ __launch_bounds__(T_THREADS) void iou_stream3_kernel(const float* __restrict__ b1, int n1,
                                                                const float* __restrict__ b2, int n2,
                                                                float* __restrict__ out, BoxRec* __restrict__ recsA,
                                                                int* __restrict__ tcount,
                                                                unsigned short* __restrict__ slots, int wcap,
                                                                const ColPrep prep = ColPrep(), const int probe = 0, const int order = -1,
                                                                const AssignZero az = AssignZero(),
                                                                unsigned* __restrict__ bits = nullptr, const int thin = 0) {
  // (probe: probes build only, tools/iou_stream_phases.sh -- 1 leave after the zeros, 2 after the prologue, 3 before the
  // queue flush: what each part of the kernel adds to the plain fill)
  __shared__ __attribute__((aligned(16))) float rows[P_ROWS][12];  // cx, cy, rad, ex, ey, -, -, -, cx - ex, cx + ex, cy - ey, cy + ey
  __shared__ unsigned short queue[4 * P_WSEG];
  __shared__ int wcount[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // column tiles are walked from the END of the box list: anchor lists run from the fine to the coarse level
  // and the coarse anchors (the last few tiles) carry most survivors -- started last they would be the tail
  const int bx = (int)(gridDim.x - 1 - blockIdx.x);
  const int colbase = bx * T_COLS;
  const int col0 = colbase + tid * T_CPT;
  const int row0 = blockIdx.y * P_ROWS;
  const int nrows = min(P_ROWS, n1 - row0);
  // Round 5: the tile's zeros go out EARLY and unconditionally -- their addresses depend on the block index alone, so
  // they drain while the boxes are tested (they used to wait behind each row's tests).  A surviving pair's element is
  // zeroed too: the drain, a later launch, overwrites it.  They are issued right AFTER the box loads: gfx9 counts
  // loads and stores in one in-order vmcnt, so a load issued behind the stores waits for all of them to be
  // acknowledged (measured, tools/iou_stream_phases.sh: stores first = fill 14.2 + prologue 2.8 + row loop 4.4 us,
  // nothing overlapped).
  // order: -1 every workgroup stores early; 0..30 the workgroups whose linear index has that bit set store early and
  // the others after their tests (so a CU holds both kinds: one kind's stores run under the other's tests); 31 all late
  // (option iou_order, default 8: 18.8-19.0 us against 19.3-19.5 for -1 in three alternations on one box; other bits 18.6 ..
  // 20.2.  Measured and not kept, tools/iou_order_ab.sh: one or two EXTRA wavefronts per workgroup that do nothing but
  // store the zeros while the four others only test -- 21.3 / 19.8 us.  Where the zeros are issued is not what the
  // kernel's time is made of.)
  const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
  const bool early = order < 0 || ((lin >> order) & 1u);
  auto zero_tile = [&](const bool late) {
    if (!late) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loads have landed, nothing below waits on vmcnt again
    if (!out || late == early || (R3_HAS_PROBES && probe >= 16)) return;  // (probe 16 + p: phase p without the zeros)
    if (VEC && col0 + T_CPT <= n2) {
#pragma unroll
      for (int r = 0; r < P_ROWS; r++)
        if (r < nrows) *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * n2 + col0) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int r = 0; r < P_ROWS; r++)
#pragma unroll
        for (int c = 0; c < T_CPT; c++)
          if (r < nrows && col0 + c < n2) out[(size_t)(row0 + r) * n2 + col0 + c] = 0.f;
    }
  };
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < P_TICKETS) tcount[tcount_ints((long long)gridDim.x * gridDim.y) + tid * 32] = 0;  // the drain's tickets
  if (az.colkey) {
    if (blockIdx.y == 0) {
#pragma unroll
      for (int c = 0; c < T_CPT; c++)
        if (col0 + c < n2) az.colkey[col0 + c] = 0xffffffffULL, az.lowq[col0 + c] = 0;
    }
    if (blockIdx.x == 0 && tid < nrows) az.rowkey[row0 + tid] = 0xffffffffULL;
  }
  if (PREP) {
    // the buffer's header against this launch (uniform: one scalar load per workgroup)
    const int4 h = *prep.hdr;
    if (h.x != COLPREP_MAGIC || h.y != GEOM || h.z != n2 || h.w != (COLPREP_MAGIC ^ GEOM ^ n2)) {
      const float nan = __builtin_nanf("");
      if (out && col0 < n2) {  // (VEC: n2 % 4 == 0)
        for (int r = 0; r < nrows; r++) *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * n2 + col0) = make_float4(nan, nan, nan, nan);
      }
      if (az.colkey) {  // the assignment: NaN keys -> max_overlaps NaN, every anchor ignored (-1)
        const unsigned long long nk = ((unsigned long long)__float_as_uint(nan) << 32) | 0xffffffffULL;
        if (blockIdx.y == 0)
          for (int c = 0; c < T_CPT; c++)
            if (col0 + c < n2) az.colkey[col0 + c] = nk, az.lowq[col0 + c] = 0;
        if (blockIdx.x == 0 && tid < nrows) az.rowkey[row0 + tid] = nk;
      }
      if (tid == 0) tcount[blockIdx.y * gridDim.x + bx] = 0;  // nothing for the drain
      if (BITS) bits[(size_t)(blockIdx.y * gridDim.x + bx) * T_THREADS + tid] = 0xffffffffu;
      return;
    }
  }
  float rowraw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < nrows) {
    const float* b = b1 + (size_t)(row0 + tid) * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) rowraw[k] = b[k];
  }
  if (R3_HAS_PROBES && (probe & 15) == 1) {
    zero_tile(false);
    if (tid == 0) tcount[blockIdx.y * gridDim.x + bx] = 0;
    return;
  }
  float cx[T_CPT], cy[T_CPT], cr[T_CPT], cex[T_CPT], cey[T_CPT];
  bool cvalid[T_CPT];
  float bx0, bx1, by0, by1;
  bool fin;
  if (PREP) {  // (VEC form only: n2 % 4 == 0, so a lane's four columns are all valid or all beyond the list)
    const bool v = col0 < n2;
    const int cs = v ? col0 : 0;
    const float4 r4 = *reinterpret_cast<const float4*>(prep.rad + cs);
    cr[0] = r4.x, cr[1] = r4.y, cr[2] = r4.z, cr[3] = r4.w;
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      const float4 q = prep.rej[cs + c];
      cx[c] = q.x, cy[c] = q.y, cex[c] = q.z, cey[c] = q.w;
      cvalid[c] = v;
    }
    const float4 wb = prep.wbox[(colbase >> 8) + __builtin_amdgcn_readfirstlane(wave)];  // (NaN: never the shortcut)
    zero_tile(false);
    bx0 = wb.x, bx1 = wb.y, by0 = wb.z, by1 = wb.w;
    fin = v;  // (a wavefront with a lane beyond the list never takes the wave shortcut: its wbox entry may be padding)
  } else {
  {
    float raw[T_CPT * 5];
    if (col0 + T_CPT <= n2 && (reinterpret_cast<uintptr_t>(b2) & 15) == 0) {
      const float4* src = reinterpret_cast<const float4*>(b2 + (size_t)col0 * 5);  // 80 contiguous bytes per lane
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const float4 v = src[k];
        raw[4 * k] = v.x; raw[4 * k + 1] = v.y; raw[4 * k + 2] = v.z; raw[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int c = 0; c < T_CPT; c++) cvalid[c] = true;
    } else {
#pragma unroll
      for (int c = 0; c < T_CPT; c++) {
        cvalid[c] = (col0 + c) < n2;
#pragma unroll
        for (int k = 0; k < 5; k++) raw[c * 5 + k] = cvalid[c] ? b2[(size_t)(col0 + c) * 5 + k] : 0.f;
      }
    }
    zero_tile(false);
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      cx[c] = raw[c * 5];
      cy[c] = raw[c * 5 + 1];
      reject_data(cx[c], cy[c], raw[c * 5 + 2], raw[c * 5 + 3], raw[c * 5 + 4], cr[c], cex[c], cey[c]);
      if (GEOM == 3 && thin) cvalid[c] = cvalid[c] && !iou_thin_box(raw[c * 5 + 2], raw[c * 5 + 3]);
    }
  }
  // Bounding box of this LANE's 4 columns (inflated extents included).  Anchors and refined boxes come in spatial
  // order, so for most (row, wave) combinations the row's box lies outside every lane's box: one compare per lane
  // and a ballot then replace the 4 per-column tests.  A lane with a non-finite column never counts as outside.
  // (Until round 5 the box was reduced over the wave -- 24 ds_bpermute + 24 min/max per thread for a uniform test
  // that costs the same four compares.)
  bx0 = cx[0] - cex[0], bx1 = cx[0] + cex[0], by0 = cy[0] - cey[0], by1 = cy[0] + cey[0];
  float fsum = (bx0 + bx1) + (by0 + by1);
#pragma unroll
  for (int c = 1; c < T_CPT; c++) {
    const float x0 = cx[c] - cex[c], x1 = cx[c] + cex[c], y0 = cy[c] - cey[c], y1 = cy[c] + cey[c];
    bx0 = fminf(bx0, x0);
    bx1 = fmaxf(bx1, x1);
    by0 = fminf(by0, y0);
    by1 = fmaxf(by1, y1);
    fsum += (x0 + x1) + (y0 + y1);
  }
  fin = fabsf(fsum) < 3.0e38f;  // (NaN / inf in any centre or extent, or an overflowing sum: never outside)
  }
  if (tid < nrows) {
    float rad, ex, ey;
    reject_data(rowraw[0], rowraw[1], rowraw[2], rowraw[3], rowraw[4], rad, ex, ey);
    rows[tid][0] = rowraw[0]; rows[tid][1] = rowraw[1]; rows[tid][2] = rad; rows[tid][3] = ex; rows[tid][4] = ey;
    if (GEOM == 3) rows[tid][5] = (thin && iou_thin_box(rowraw[2], rowraw[3])) ? 1.f : 0.f;
    rows[tid][8] = rowraw[0] - ex; rows[tid][9] = rowraw[0] + ex; rows[tid][10] = rowraw[1] - ey; rows[tid][11] = rowraw[1] + ey;
  }
  // the exact row records for the drain kernel: written once, by the middle column tile's workgroups
  if (blockIdx.x == gridDim.x / 2 && tid >= 64 && tid < 64 + nrows) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)(row0 + tid - 64) * 5, 0.f, rec);
    recsA[row0 + tid - 64] = rec;
  }
  // survivors: one PRIVATE LDS segment per wave, its fill count in a wave-uniform register -- no LDS atomics,
  // nothing to wait for (a shared queue with one atomicAdd per ballot cost 12 of the kernel's 32 us)
  unsigned short* wq = queue + wave * P_WSEG;
  int cnt = 0;
  bool full = false;
  unsigned bm = 0;  // (BITS) this lane's survivors: bit 4 r + c
  __syncthreads();
  if (R3_HAS_PROBES && (probe & 15) == 2) {
    if (tid == 0) tcount[blockIdx.y * gridDim.x + bx] = bx0 + bx1 + by0 + by1 + cx[0] + cr[1] == 12345.f;  // (keeps the prologue alive)
    return;
  }
  for (int r = 0; r < nrows; r++) {
    const float* A = rows[r];
    if (GEOM == 3 && A[5] != 0.f) continue;  // (a thin row: uniform)
    const float4 ab = *reinterpret_cast<const float4*>(A + 8);
    if (__builtin_amdgcn_ballot_w64(fin & ((ab.x > bx1) | (ab.y < bx0) | (ab.z > by1) | (ab.w < by0))) == ~0ULL) continue;
    const float ax = A[0], ay = A[1], ar = A[2], aex = A[3], aey = A[4];
    bool pend[T_CPT];
    bool any = false;
#pragma unroll
    for (int c = 0; c < T_CPT; c++) {
      float dx = ax - cx[c], dy = ay - cy[c];
      float rr = ar + cr[c];
      bool apart = (dx * dx + dy * dy > rr * rr) | (fabsf(dx) > aex + cex[c]) | (fabsf(dy) > aey + cey[c]);
      pend[c] = cvalid[c] && !apart;
      any |= pend[c];
      if (BITS) bm |= (pend[c] ? 1u : 0u) << (4 * r + c);
    }
    if (__ballot(any)) {
      if (cnt + 256 > wcap) {
        full = true;
      } else {
#pragma unroll
        for (int c = 0; c < T_CPT; c++) {
          const unsigned long long m = __ballot(pend[c]);
          if (pend[c]) wq[cnt + __popcll(m & ((1ULL << lane) - 1ULL))] = (unsigned short)((r << 10) | (tid * T_CPT + c));
          cnt += __popcll(m);
        }
      }
    }
  }
  if (R3_HAS_PROBES && (probe & 15) == 3) {
    if (lane == 0) tcount[blockIdx.y * gridDim.x + bx] = cnt == 123456789;
    return;
  }
  if (lane == 0) wcount[wave] = full ? -1 : cnt;
  __syncthreads();
  const int c0 = wcount[0], c1 = wcount[1], c2 = wcount[2], c3 = wcount[3];
  const bool dense = (c0 | c1 | c2 | c3) < 0;
  const int tile = (int)(blockIdx.y * gridDim.x) + bx;
  if (tid == 0) tcount[tile] = dense ? -1 : c0 + c1 + c2 + c3;
  zero_tile(true);
  if (BITS) bits[(size_t)tile * T_THREADS + tid] = dense ? 0xffffffffu : bm;  // (a dense tile: the drain writes every element)
  if (dense) return;
  unsigned short* slot = slots + (size_t)tile * P_SLOT + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
  for (int q = lane; q < cnt; q += 64) slot[q] = wq[q];
}

// (the fused assignment's packed (IoU bits << 32 | ~index) keys: larger IoU wins a 64-bit atomicMax, then the SMALLER index)
typedef unsigned long long u64k;

__device__ __forceinline__ u64k pack_key(float iou, unsigned idx) {
  return ((u64k)__float_as_uint(iou) << 32) | (u64k)(0xffffffffu - idx);
}
__device__ __forceinline__ float key_iou(u64k k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned key_idx(u64k k) { return 0xffffffffu - (unsigned)(k & 0xffffffffu); }

// ASSIGN (round 5): the fused assignment's drain on the same queue -- instead of the matrix element a clipped pair
// updates the per-column / per-row keys and leaves its IoU in siou[tile * 8192 + entry] for the low-quality sweep.
struct AssignOut {
  float* siou;     // [tiles][P_ROWS * T_COLS]
  u64k* rowkey;    // [n1]
  u64k* colkey;    // [n2]
  int n1_lds;      // rows whose maxima are reduced in LDS first (dynamic LDS: n1_lds u64)
  int probe;       // (probes build, tools/assign_emit_ab.sh: 1 no column keys, 2 a look before the column atomic, 3 no siou, 4 no row keys, 5 no row flush, 6 a look before the flush's atomics)
};

// FILL (round 6, option iou_impl 5, matrix form only): the launch also writes the matrix's zeros.  Its first `nfill`
// workgroups begin as FILL blocks -- tile by tile, every element the stream kernel's survivor bits do not name -- and
// then join the drain; clips write the named elements only, so the two kinds of store never meet and need no order.
// Every workgroup has the drain's footprint (128 VGPRs, 18 KB of LDS: four per compute unit), so a fill block is a
// drain block that starts late: the blocks behind a wavefront's first are then always dealt by ticket.
// (the v3 hull form at 4 waves per SIMD = 128 registers spilled 12 B / lane; bounded to 3 it takes 164 registers and no
// scratch: 28.1 against 28.6 us at 128 x 196 416, profiles/r06_iou_occupancy_ab.txt)
#ifndef R3_DRAIN3_V3_WAVES
#define R3_DRAIN3_V3_WAVES 3
#endif
template <int GEOM, bool FAST = false, bool ASSIGN = false, bool FILL = false>
__global__ __launch_bounds__(T_THREADS, FAST ? (GEOM == 3 ? R3_DRAIN3_V3_WAVES : 4) : 1) void iou_drain3_kernel(const float* __restrict__ b1, int n1,
                                                               const float* __restrict__ b2, int n2, int iof,
                                                               const BoxRec* __restrict__ recsA,
                                                               const int* __restrict__ tcount,
                                                               const unsigned short* __restrict__ slots, int tiles_x,
                                                               int tiles, float* __restrict__ out,
                                                               const BoxRec* __restrict__ recsB = nullptr,
                                                               unsigned long long* __restrict__ stamps = nullptr,
                                                               const AssignOut ao = AssignOut(), const int dyn = 0,
                                                               const unsigned* __restrict__ bits = nullptr,
                                                               const int nfill = 0, const int thin = 0) {
  // (probes build, tools/iou_drain_stamps.py: wave 0 of every workgroup stamps its phases with the 100 MHz clock)
#ifdef R3_PROBES
#define R3_DSTAMP(k)                                                                             \
  if (stamps && threadIdx.x == 0) {                                                              \
    __builtin_amdgcn_s_waitcnt(0);                                                               \
    stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                     \
  }
#else
#define R3_DSTAMP(k)
#endif
  R3_DSTAMP(0)
  // One clip takes a wave ~15 us from first load to store (long dependent chains through LDS), the ALUs are idle
  // most of that time, so what counts is how many waves a CU holds.  v1: 8 candidate slots per lane instead of
  // the reference's 16 (wave-private [slot][lane] regions of 4 KB) => 18 KB of LDS per workgroup, 8 workgroups
  // per CU; the rare pair with a 9th candidate is redone by lanes 0..31 with 16 slots in the same region.
  // v2 / v3 (hull): 12 of the 24 slots, redo with 24 by lanes 0..31 likewise => 24.5 KB instead of 49 KB per workgroup
  // (the 49 KB form ran at occupancy 3: that was the 41 us v3 drain)
  // Round 5, v1 (FAST): the straight-line clip of r3_clip.h -- crossings decided from signs, candidates in registers,
  // LDS only as the dynamic index (9 slots per lane: 8 candidates + the dump slot); a pair it flags (not in general
  // position: ~2e-4 of random pairs) is redone by lanes 0..31 with the exact 16-slot form in the same region.
  constexpr bool SHORT = true;
  constexpr int CAPS = GEOM == 1 ? (FAST ? R3_CLIP_SLOTS : 8) : 12, CAPF = GEOM == 1 ? R3_V1_CAP : 24;  // short / full slots per lane
  constexpr int D_PAIRS = P_ROWS * T_COLS;  // a dense tile is enumerated pair by pair
  __shared__ float2 pts[CAPS * T_THREADS];
  __shared__ unsigned pre[P_GROUPS + 1];
  __shared__ unsigned wsum[4];
  extern __shared__ __attribute__((aligned(16))) u64k rowbest[];  // ASSIGN: n1_lds entries (0 = none)
  if (ASSIGN) {
    for (int i = threadIdx.x; i < ao.n1_lds; i += T_THREADS) rowbest[i] = 0;  // (visible behind the prefix barriers below)
  }
  // what a clipped pair leaves behind: the matrix element, or (ASSIGN) its IoU for the low-quality sweep and the keys --
  // the few hundred gt rows take ~5 k updates each (global atomics on 128 addresses serialise: 0.6 ms), so rows are
  // reduced in LDS and flushed once per workgroup; columns are many and rarely contended: look, then atomicMax
  auto emit = [&](const unsigned r, const unsigned c, const float v, const unsigned t, const unsigned off) {
    if (!ASSIGN) {
      out[(size_t)r * n2 + c] = v;
    } else {
      const int pb = R3_HAS_PROBES ? ao.probe : 0;
      if (pb != 3) ao.siou[(size_t)t * D_PAIRS + off] = v;
      if (v > 0.f) {
        const u64k kc = pack_key(v, r), kr = pack_key(v, c);
        // (no look at the key first: the nontemporal load in front of the atomic was 8 of the drain's 44 us, and 16 more
        // in front of the row flush below -- tools/assign_emit_ab.sh; probes 2 / 6 bring the looks back)
        if (pb == 2) {
          if (kc > __builtin_nontemporal_load(&ao.colkey[c])) atomicMax(&ao.colkey[c], kc);
        } else if (pb != 1) {
          atomicMax(&ao.colkey[c], kc);
        }
        if (pb == 4) return;
        if ((int)r < ao.n1_lds) atomicMax(&rowbest[r], kr);
        else if (kr > __builtin_nontemporal_load(&ao.rowkey[r])) atomicMax(&ao.rowkey[r], kr);
      }
    }
  };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (FILL && (int)blockIdx.x < nfill) {
    // (n2 % 4 == 0 and a 16-byte aligned matrix: the launcher; a lane's word = the stream kernel's four columns)
    // The bits and counts of FB tiles are requested together, THEN their zeros go out: gfx9 counts loads and stores in
    // one in-order vmcnt, so a tile's loads issued behind the previous tile's stores waited for every one of them to be
    // acknowledged -- one memory round trip per tile, 128 fill blocks ran 56 us where 256 ran 44.
    constexpr int FB = 8;
    for (int t0 = blockIdx.x; t0 < tiles; t0 += nfill * FB) {
      unsigned bmv[FB];
      int tcv[FB];
#pragma unroll
      for (int k = 0; k < FB; k++) {
        const int t = t0 + k * nfill;
        const int tc = min(t, tiles - 1);
        bmv[k] = bits[(size_t)tc * T_THREADS + tid];
        tcv[k] = tcount[tc];
      }
#pragma unroll
      for (int k = 0; k < FB; k++) {
        const int t = t0 + k * nfill;
        if (t >= tiles || tcv[k] < 0) continue;  // dense tile: enumerated pair by pair below, every element written there
        const int by = t / tiles_x, bx = t - by * tiles_x;
        const int col0 = bx * T_COLS + tid * T_CPT, row0 = by * P_ROWS;
        if (col0 >= n2) continue;
#pragma unroll
        for (int r = 0; r < P_ROWS; r++) {
          if (row0 + r >= n1) break;
          float* p = out + (size_t)(row0 + r) * n2 + col0;
          const unsigned m = (bmv[k] >> (4 * r)) & 15u;
          if (m == 0u) {
            *reinterpret_cast<float4*>(p) = make_float4(0.f, 0.f, 0.f, 0.f);
          } else {
#pragma unroll
            for (int c = 0; c < T_CPT; c++)
              if (!((m >> c) & 1u)) p[c] = 0.f;
          }
        }
      }
    }
  }
  // prefix over the tile counts in groups of G consecutive tiles; thread t owns groups P_GPT * t .. P_GPT * t + P_GPT - 1.
  // Round 5: a thread's counts are P_GPT * G consecutive ints -- at G = 1 (up to 3072 tiles) three 16-byte loads in
  // flight together; the loop of scalar loads it replaces was 12 L2 round trips one after the other, 2.4 us in front of
  // every workgroup's first clip.
  const int G = (tiles + P_GROUPS - 1) / P_GROUPS;
  unsigned gs[P_GPT];
  unsigned mine = 0;
  if (G == 1) {
    const int4* p4 = reinterpret_cast<const int4*>(tcount) + tid * (P_GPT / 4);
    int4 v[P_GPT / 4];
#pragma unroll
    for (int j = 0; j < P_GPT / 4; j++) v[j] = p4[j];
#pragma unroll
    for (int j = 0; j < P_GPT / 4; j++) {
      const int c[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int t = tid * P_GPT + j * 4 + e;
        gs[j * 4 + e] = t < tiles ? (c[e] < 0 ? (unsigned)D_PAIRS : (unsigned)c[e]) : 0u;
        mine += gs[j * 4 + e];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < P_GPT; k++) {
      unsigned sum = 0;
      const int t0 = (tid * P_GPT + k) * G;
      if ((G & 3) == 0) {  // (512 x 196 416: G = 4)
        for (int t = t0; t < t0 + G; t += 4) {
          const int4 v = *reinterpret_cast<const int4*>(tcount + t);
          const int c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; e++) sum += t + e < tiles ? (c[e] < 0 ? (unsigned)D_PAIRS : (unsigned)c[e]) : 0u;
        }
      } else {
        for (int t = t0; t < t0 + G && t < tiles; t++) {
          const int c = tcount[t];
          sum += c < 0 ? (unsigned)D_PAIRS : (unsigned)c;
        }
      }
      gs[k] = sum;
      mine += sum;
    }
  }
  unsigned incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned base = incl - mine;
  for (int w = 0; w < wave; w++) base += wsum[w];
#pragma unroll
  for (int k = 0; k < P_GPT; k++) {
    pre[tid * P_GPT + k] = base;
    base += gs[k];
  }
  if (tid == T_THREADS - 1) pre[P_GROUPS] = base;
  __syncthreads();
  const unsigned total = pre[P_GROUPS];
  R3_DSTAMP(1)
  int trips = 0;
  const unsigned nwaves = gridDim.x * (T_THREADS / 64);
  const unsigned wid = blockIdx.x * (T_THREADS / 64) + wave;
  // (blocks are dealt in P_TICKETS classes, block b in class b % P_TICKETS: a wavefront's first block is its own index, the
  // class's further blocks come from the class's counter)
  // (only where a wavefront has three blocks or more: at 1.7 blocks per wavefront -- 128 x 196 416 -- there is nothing to even
  // out, 23.3 us both ways; 512 x 196 416, 7 blocks: 104 -> 92-95 us; tools/iou_dyn_ab.sh)
  unsigned* const ticket = (dyn && nwaves % P_TICKETS == 0 && (FILL || total >= 3u * 64u * nwaves))
      ? reinterpret_cast<unsigned*>(const_cast<int*>(tcount)) + tcount_ints(tiles) + (wid % P_TICKETS) * 32 : nullptr;
  for (unsigned qb = wid * 64; qb < total;) {  // wave-uniform
    // the block after this one: by ticket (requested here, looked at behind the clip) or at the static stride
    unsigned tk = 0;
    if (ticket && lane == 0) tk = atomicAdd(ticket, 1u);
    const unsigned q = qb + lane;
    bool valid = q < total;
    bool dense = false;
    unsigned r = 0, c = 0, et = 0, eoff = 0;  // (et, eoff: the entry's tile and offset in it)
    if (valid) {
      int g = 0;  // group of entry q: largest g with pre[g] <= q
#pragma unroll
      for (int step = 2048; step >= 1; step >>= 1)
        if (g + step <= P_GROUPS && pre[g + step] <= q) g += step;
      unsigned off = q - pre[g];
      int t = g * G;
      if (G == 1) dense = pre[g + 1] - pre[g] == (unsigned)D_PAIRS;  // (a tile's queue holds at most 4 * P_WSEG < D_PAIRS entries)
      else for (;;) {  // tile inside the group
        // (round 5, measured and dropped: the tile counts kept in LDS as well, so that this walk needs no global load --
        // the lookup 1.8 -> 1.5 us, the prefix phase 2.4 -> 3.3 us)
        const int tc = tcount[t];
        const unsigned cn = tc < 0 ? (unsigned)D_PAIRS : (unsigned)tc;
        if (off < cn) {
          dense = tc < 0;
          break;
        }
        off -= cn;
        t++;
      }
      const unsigned e = dense ? off : slots[(size_t)t * P_SLOT + off];
      et = (unsigned)t, eoff = off;
      const int by = t / tiles_x, bx = t - by * tiles_x;
      r = (unsigned)(by * P_ROWS) + (e >> 10);
      c = (unsigned)(bx * T_COLS) + (e & 1023u);
      valid = r < (unsigned)n1 && c < (unsigned)n2;  // dense edge tiles enumerate beyond the matrix
    }
    bool over = false;
    if ((trips & 0xffff) == 0) { R3_DSTAMP(2) }
    if (valid) {
      const BoxRec A = recsA[r];
      BoxRec B;
      if (recsB) B = recsB[c];  // (prepared columns)
      else make_record<GEOM>(b2 + (size_t)c * 5, 0.f, B);
      if ((trips & 0xffff) == 0) { R3_DSTAMP(3) }
      float v = 0.f;
      // a dense tile's pairs were never tested with exact records: do it here (apart => 0, as in every form)
      bool dead = dense && boxes_apart(A.f, B.f);
      if (GEOM == 3 && dense && thin && !dead)  // (obb_overlaps' thin rows / columns inside a dense tile)
        dead = iou_thin_box(b1[(size_t)r * 5 + 2], b1[(size_t)r * 5 + 3]) || iou_thin_box(b2[(size_t)c * 5 + 2], b2[(size_t)c * 5 + 3]);
      if (!dead) {
        const LanePts<64> lp{pts + wave * (64 * CAPS) + lane};  // wave-private [slot][lane] region
        if constexpr (GEOM == 1 && FAST) v = v1_clip_fast(A.f, B.f, iof != 0, ClipLds<64>{pts + wave * (64 * CAPS) + lane}, over);
        else if constexpr (GEOM == 1) v = v1_pair_lds<64, CAPS>(A.f, B.f, iof != 0, lp, &over);
        else if constexpr (FAST) v = hull_clip_fast<GEOM == 2>(A.f, B.f, iof == 0, ClipLds<64>{pts + wave * (64 * CAPS) + lane}, over);
        else v = hull_pair_lds<GEOM == 2, 64, CAPS>(A.f, B.f, iof == 0, lp, &over);
      }
      if ((trips & 0xffff) == 0) { R3_DSTAMP(4) }
      if (!over) emit(r, c, v, et, eoff);
    } else if (ASSIGN && q < total) {
      ao.siou[(size_t)et * D_PAIRS + eoff] = 0.f;  // (a dense edge tile's entry beyond the matrix: nothing for the sweep)
    }
    if ((trips & 0xffff) == 0) { R3_DSTAMP(5) }
    trips++;
    if (SHORT) {
      unsigned long long m = __ballot(over);
      if (m) trips += 0x10000;  // (probe: redo passes in the high half)
      while (m) {  // rare: lane k < 32 redoes the k-th flagged pair with the full 16 slots
        int src = -1, seen = 0;
        for (unsigned long long t2 = m; t2; t2 &= t2 - 1) {
          if (seen == lane) src = __builtin_ctzll(t2);
          seen++;
        }
        const int sl = src < 0 ? 0 : src;
        const unsigned rr = __shfl(r, sl), cc = __shfl(c, sl), tt = __shfl(et, sl), oo = __shfl(eoff, sl);
        if (lane < 32 && src >= 0) {
          const BoxRec A = recsA[rr];
          BoxRec B;
          if (recsB) B = recsB[cc];
          else make_record<GEOM>(b2 + (size_t)cc * 5, 0.f, B);
          const LanePts<32> lp{pts + wave * (64 * CAPS) + lane};  // (32 lanes x CAPF slots = the same region)
          emit(rr, cc, GEOM == 1 ? v1_pair_lds<32, CAPF>(A.f, B.f, iof != 0, lp)
                                 : hull_pair_lds<GEOM == 2, 32, CAPF>(A.f, B.f, iof == 0, lp), tt, oo);
        }
        for (int k = 0; k < 32 && m; k++) m &= m - 1;
      }
    }
    qb = ticket ? (nwaves + (unsigned)__builtin_amdgcn_readfirstlane((int)tk) * P_TICKETS + wid % P_TICKETS) * 64u : qb + nwaves * 64u;
  }
  R3_DSTAMP(6)
#ifdef R3_PROBES
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)trips;
#endif
#undef R3_DSTAMP
  if (ASSIGN) {
    __syncthreads();
    for (int i = threadIdx.x; i < ao.n1_lds; i += T_THREADS) {
      const u64k k = rowbest[i];
      const int pb = R3_HAS_PROBES ? ao.probe : 0;  // (5: no flush, 6: flush without the look)
      if (pb == 5) continue;
      if (k && (pb != 6 || k > __builtin_nontemporal_load(&ao.rowkey[i]))) atomicMax(&ao.rowkey[i], k);
    }
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

template <int GEOM, bool FAST = false>
__global__ __launch_bounds__(T_THREADS, FAST ? 4 : 1) void assign_drain_kernel(const BoxRec* __restrict__ recsA,
                                                                 const BoxRec* __restrict__ recsB, int n2,
                                                                 const unsigned* __restrict__ gqueue,
                                                                 const unsigned* __restrict__ counter,
                                                                 float* __restrict__ qiou, u64k* __restrict__ rowkey,
                                                                 u64k* __restrict__ colkey, int n1_lds) {
  // v1: the 8-slot clip of the IoU drain (wave-private [slot][lane] regions, 16 KB per workgroup instead of 32: twice
  // the resident waves; a pair with a 9th candidate is redone by lanes 0..31 with 16 slots in the same region)
  // Round 5 (FAST): the straight-line clip (r3_clip.h), 9 slots per lane; flagged pairs take the same redo
  // v2 / v3 (hull): 12 of the 24 slots per lane with the same redo (round 5: 49 KB -> 24.5 KB of LDS, occupancy 3 -> 5+)
  constexpr bool SHORT = true;
  constexpr int CAPS = GEOM == 1 ? (FAST ? R3_CLIP_SLOTS : 8) : 12;
  __shared__ float2 pts[CAPS * T_THREADS];
  extern __shared__ __attribute__((aligned(16))) u64k rowbest[];  // n1_lds entries (0 = none): per-workgroup row maxima
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned total = *counter;
  for (int i = threadIdx.x; i < n1_lds; i += T_THREADS) rowbest[i] = 0;
  __syncthreads();
  // (Measured: without the key updates below this kernel takes 27 us instead of 52 -- the ~0.37 M device-scope
  // atomicMax on the column keys execute memory-side, ~12 ns each per channel.)
  // The few hundred gt rows take ~5 k updates each: global atomics on 128 addresses serialise (0.6 ms).  Rows are
  // reduced in LDS first and flushed once per workgroup; columns (anchors) are many and rarely contended: look (the
  // keys only grow), then atomicMax.
  auto record = [&](const unsigned q, const unsigned r, const unsigned c, const float v) {
    qiou[q] = v;
    if (v > 0.f) {
      const u64k kc = pack_key(v, r), kr = pack_key(v, c);
      atomicMax(&colkey[c], kc);
      if ((int)r < n1_lds) atomicMax(&rowbest[r], kr);
      else if (kr > __builtin_nontemporal_load(&rowkey[r])) atomicMax(&rowkey[r], kr);
    }
  };
  for (unsigned qb = blockIdx.x * T_THREADS + wave * 64; qb < total; qb += gridDim.x * T_THREADS) {  // wave-uniform
    const unsigned q = qb + lane;
    const bool valid = q < total;
    unsigned r = 0, c = 0;
    bool over = false;
    if (valid) {
      const unsigned e = gqueue[q];
      r = e / (unsigned)n2;
      c = e - r * (unsigned)n2;
      const BoxRec A = recsA[r];
      const BoxRec B = recsB[c];
      float v;
      if constexpr (GEOM == 1 && FAST) {
        v = v1_clip_fast(A.f, B.f, false, ClipLds<64>{pts + wave * (64 * CAPS) + lane}, over);
      } else if constexpr (GEOM == 1) {
        const LanePts<64> lp8{pts + wave * (64 * CAPS) + lane};
        v = v1_pair_lds<64, 8>(A.f, B.f, false, lp8, &over);
      } else {
        const LanePts<64> lp12{pts + wave * (64 * CAPS) + lane};
        v = hull_pair_lds<GEOM == 2, 64, 12>(A.f, B.f, true, lp12, &over);
      }
      if (!over) record(q, r, c, v);
    }
    if (SHORT) {
      unsigned long long m = __ballot(over);
      while (m) {  // rare: lane k < 32 redoes the k-th flagged pair with the full 16 slots
        int src = -1, seen = 0;
        for (unsigned long long t2 = m; t2; t2 &= t2 - 1) {
          if (seen == lane) src = __builtin_ctzll(t2);
          seen++;
        }
        const int sl = src < 0 ? 0 : src;
        const unsigned qq = __shfl(q, sl), rr = __shfl(r, sl), cc = __shfl(c, sl);
        if (lane < 32 && src >= 0) {
          const BoxRec A = recsA[rr];
          const BoxRec B = recsB[cc];
          const LanePts<32> lpf{pts + wave * (64 * CAPS) + lane};  // (32 lanes x the full slots = the same region)
          if constexpr (GEOM == 1) record(qq, rr, cc, v1_pair_lds<32, R3_V1_CAP>(A.f, B.f, false, lpf));
          else record(qq, rr, cc, hull_pair_lds<GEOM == 2, 32, 24>(A.f, B.f, true, lpf));
        }
        for (int k = 0; k < 32 && m; k++) m &= m - 1;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n1_lds; i += T_THREADS) {
    const u64k k = rowbest[i];
    if (k) atomicMax(&rowkey[i], k);
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

// ... the same sweep over the stream + drain queue (round 5): one workgroup per tile, its entries' IoUs in siou
__global__ __launch_bounds__(256) void assign_lowq3_kernel(const int* __restrict__ tcount,
                                                           const unsigned short* __restrict__ slots, int tiles_x,
                                                           const float* __restrict__ siou, int n1, int n2,
                                                           const u64k* __restrict__ rowkey, float min_pos_iou,
                                                           int assign_all, int* __restrict__ lowq) {
  const int t = blockIdx.x;  // (blockIdx.y: a quarter of the tile's entries -- most tiles have none and leave at once)
  const int tc = tcount[t];
  if (tc == 0) return;
  const int cnt = tc < 0 ? P_ROWS * T_COLS : tc;
  const int by = t / tiles_x, bx = t - by * tiles_x;
  for (int off = blockIdx.y * 256 + threadIdx.x; off < cnt; off += 256 * (int)gridDim.y) {
    const float v = siou[(size_t)t * (P_ROWS * T_COLS) + off];
    if (!(v > 0.f)) continue;
    const unsigned e = tc < 0 ? (unsigned)off : slots[(size_t)t * P_SLOT + off];
    const unsigned r = (unsigned)(by * P_ROWS) + (e >> 10), c = (unsigned)(bx * T_COLS) + (e & 1023u);
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
                                                           float* __restrict__ gt_max, int64_t* __restrict__ gt_argmax,
                                                           const int64_t* __restrict__ gt_labels,
                                                           int64_t* __restrict__ labels) {
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
  // (mmdet's assigned_labels: -1, and gt_labels[assigned - 1] at the positives -- max_iou_assigner.py's last lines)
  if (labels) labels[j] = a > 0 ? gt_labels[a - 1] : -1;
}

// vec_iou_iof_kernel (rbbox_geo_kernel.cu:271-309): out[i] = f(b1[i % n1], b2[i % n2]).
template <int GEOM>
__global__ __launch_bounds__(256, GEOM == 1 ? R3_VEC_WAVES : 1) void iou_vec_kernel(const float* __restrict__ b1, int n1,
                                                      const float* __restrict__ b2, int n2,
                                                      int iof, float* __restrict__ out) {
  // v1: the straight-line clip (9 slots), the exact list form (16) for what it flags; hull: 12 of the 24 slots in
  // wave-private regions, a pair with a 13th point redone by lanes 0..31 with 24 slots (round 5: 49 KB -> 24.5 KB)
  __shared__ float2 pts[(GEOM == 1 ? R3_V1_CAP : 12) * 256];
  const LanePts<256> lp{pts + threadIdx.x};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = max(n1, n2);
  for (int i0 = blockIdx.x * 256 + wave * 64; i0 < n; i0 += 256 * (int)gridDim.x) {  // (wave-uniform trips)
    const int i = i0 + lane;
    const bool valid = i < n;
    BoxRec A, B;
    bool apart = true;
    if (valid) {
      make_record<GEOM>(b1 + (size_t)(i % n1) * 5, 0.f, A);
      make_record<GEOM>(b2 + (size_t)(i % n2) * 5, 0.f, B);
      apart = boxes_apart(A.f, B.f);
    }
    if constexpr (GEOM == 1) {
      if (valid) {
        float v = 0.f;
        if (!apart) {
          bool flagged = false;
          v = v1_clip_fast(A.f, B.f, iof != 0, ClipLds<256>{pts + threadIdx.x}, flagged);
          if (flagged) v = pair_slow_lds<GEOM, 256>(A.f, B.f, iof != 0, lp);
        }
        out[i] = v;
      }
    } else {
      bool over = false;
      if (valid) {
        float v = 0.f;
        if (!apart) v = hull_clip_fast<GEOM == 2>(A.f, B.f, iof == 0, ClipLds<64>{pts + wave * (64 * 12) + lane}, over);
        if (!over) out[i] = v;
      }
      unsigned long long m = __ballot(over);
      while (m) {
        int src = -1, seen = 0;
        for (unsigned long long t2 = m; t2; t2 &= t2 - 1) {
          if (seen == lane) src = __builtin_ctzll(t2);
          seen++;
        }
        const int ii = __shfl(i, src < 0 ? 0 : src);
        if (lane < 32 && src >= 0) {
          BoxRec A2, B2;
          make_record<GEOM>(b1 + (size_t)(ii % n1) * 5, 0.f, A2);
          make_record<GEOM>(b2 + (size_t)(ii % n2) * 5, 0.f, B2);
          out[ii] = hull_pair_lds<GEOM == 2, 32, 24>(A2.f, B2.f, iof == 0, LanePts<32>{pts + wave * (64 * 12) + lane});
        }
        for (int k = 0; k < 32 && m; k++) m &= m - 1;
      }
    }
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PipeLayout {
  int* tcount;
  BoxRec* recsA;
  unsigned short* slots;
  unsigned* bits;  // [tiles][256] survivor bits (option iou_impl 5)
  int tiles_x, tiles_y;
};

// workspace: the tile counts, the row records and one 8 KB slot per tile (1 B per pair)
inline size_t pipe_layout(int n1, int n2, void* ws, PipeLayout* L) {
  const int tx = (n2 + T_COLS - 1) / T_COLS, ty = (n1 + P_ROWS - 1) / P_ROWS;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* tc = take((tcount_ints((long long)tx * ty) + P_TICKET_PAD) * 4);
  char* ra = take((size_t)n1 * sizeof(BoxRec));
  char* sl = take((size_t)tx * ty * P_SLOT * 2);
  char* bt = take((size_t)tx * ty * T_THREADS * 4);
  if (L) {
    L->tcount = (int*)tc; L->recsA = (BoxRec*)ra; L->slots = (unsigned short*)sl; L->bits = (unsigned*)bt;
    L->tiles_x = tx; L->tiles_y = ty;
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
               size_t ws_bytes, hipStream_t stream, const void* prepared = nullptr, int* thin_done = nullptr) {
  // thin_done (GEOM 3): the caller wants obb_overlaps' thin-box rule; set to 1 when this launch applied it itself (the
  // pipeline on plain columns), left 0 when the caller still has to run the epilogue kernel
  const bool vec = (n2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // (every switch is read ONCE per call -- a concurrent r3det_set_option cannot pair one form's grid with
  // the other form's kernel)
  const int impl = g_r3_iou_impl, clip_impl = g_r3_clip_impl;
  if (impl == 1) {
    dim3 grid((n2 + IOU_BLOCK - 1) / IOU_BLOCK, (n1 + IOU_ROWS - 1) / IOU_ROWS);
    hipLaunchKernelGGL(iou_mat_kernel<GEOM>, grid, dim3(IOU_BLOCK), 0, stream, b1, n1, b2, n2, iof, out);
    return 0;
  }
  const unsigned long long pairs = (unsigned long long)n1 * (unsigned long long)n2;
  // More than 512 columns: the stream + drain pipeline (needs the workspace).  Narrower matrices: ONE launch of the
  // tile kernel with one column per lane (no queue in memory).  Measured (tools/iou_crossover.py, random boxes,
  // 1.8 % overlapping; us: tile kernel / pipeline): 500 x 500: 16 / 26, 2000 x 512: 17 / 26, 128 x 1000: 31 / 25,
  // 1000 x 1000: 31 / 26, 2000 x 2000: 87 / 30, 4000 x 4000: 96 / 55; dense 2000 x 2000 (40 % overlapping): 364 / 71.
  const int wide = g_r3_iou_small > 0 ? g_r3_iou_small : 513;
  PipeLayout L;
  const size_t need = pipe_layout(n1, n2, ws, &L);
  const long long tiles = (long long)L.tiles_x * L.tiles_y;
  const bool piped = ws && ws_bytes >= need && tiles <= P_MAX_TILES && impl != 2 &&
                     (n2 >= wide || impl == 4 || impl == 5);  // (5: the pipeline always, in its one-launch form where it applies)
  if (!piped) {
    if (impl == 2) launch_compact<GEOM, 4, 32>(vec, iof, b1, n1, b2, n2, out, stream);
    else if (n2 <= 512) launch_compact<GEOM, 1, 8>(vec, iof, b1, n1, b2, n2, out, stream);
    else if (pairs > 2000000ULL) launch_compact<GEOM, 4, 32>(vec, iof, b1, n1, b2, n2, out, stream);  // (no workspace)
    else launch_compact<GEOM, 4, 8>(vec, iof, b1, n1, b2, n2, out, stream);
    return 0;
  }
  const int qcap_o = g_r3_iou_qcap;
  const int wcap = qcap_o > 0 && qcap_o < P_WSEG ? qcap_o : P_WSEG;
  const dim3 grid(L.tiles_x, L.tiles_y);
  // drain: enough workgroups to fill the chip at the kernel's occupancy (6 per CU measured best: 1024 -> 33 us, 1536 -> 28, 1792 and more -> 32); grid-stride inside
  int blocks = (int)((pairs + T_THREADS - 1) / T_THREADS);
  // (round 5, straight-line clip: 4 waves per SIMD of registers => 4 workgroups per CU are resident; a larger grid's
  // second wave of workgroups pays the 6 us prefix / lookup / record-load preamble again: 1024 -> 23.4 us of stamps, 1536 -> 26.7)
  const bool fast = clip_impl == 0;
  const int dwgs = g_r3_iou_dwgs;
  const int maxb = dwgs > 0 ? dwgs : fast ? 4 * r3_cu_count() : 1536;
  if (blocks > maxb) blocks = maxb;
  if (!vec) prepared = nullptr;  // (the buffer is read only behind the stream kernel that checks its header: the VEC form)
  const int thin = (GEOM == 3 && thin_done && !prepared) ? 1 : 0;  // (prepared columns carry no widths: the epilogue stays)
  if (thin) *thin_done = 1;
  ColPrep P = ColPrep();
  if (prepared) colprep_layout(n2, prepared, &P);
  const int sorder = (int)g_r3_iou_order;
  const int sprobe = R3_HAS_PROBES ? (int)g_r3_fr_walk - 1000 : 0;  // (probes build: option fr_walk 1001 / 1002 / 1003 = stream phase probe)
  // (probes build: the stamp buffer named by the frn_stamps_lo / _hi options, shared with the FR gather's probe)
  unsigned long long* const dstamps = R3_HAS_PROBES ? reinterpret_cast<unsigned long long*>(g_r3_frn_stamps.get()) : nullptr;
  if (impl == 5 && vec && fast && blocks >= 8 && !prepared) {
    // Round 6, option iou_impl 5: fill and clip in ONE launch on disjoint addresses.  K1 = the
    // tests alone (no matrix stores) + the survivor bits; K2 = the drain whose first `nfill` workgroups write every
    // element the bits do not name before they start clipping.  A/B: profiles/r06_iou_one_launch_ab.txt.
    const int nf_o = g_r3_iou_nfill;
    const int nfill = std::min(blocks / 2, nf_o > 0 ? nf_o : r3_cu_count());
    if (prepared)
      hipLaunchKernelGGL((iou_stream3_kernel<GEOM, true, true, true>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2,
                         (float*)nullptr, L.recsA, L.tcount, L.slots, wcap, P, 0, -1, AssignZero(), L.bits);
    else
      hipLaunchKernelGGL((iou_stream3_kernel<GEOM, true, false, true>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2,
                         (float*)nullptr, L.recsA, L.tcount, L.slots, wcap, P, 0, -1, AssignZero(), L.bits, thin);
    hipLaunchKernelGGL((iou_drain3_kernel<GEOM, true, false, true>), dim3(blocks), dim3(T_THREADS), 0, stream, b1, n1, b2, n2,
                       iof, L.recsA, L.tcount, L.slots, L.tiles_x, (int)tiles, out,
                       prepared ? P.rec : (const BoxRec*)nullptr, dstamps, AssignOut(), 1, L.bits, nfill, thin);
    return 0;
  }
  if (vec && prepared)
    hipLaunchKernelGGL((iou_stream3_kernel<GEOM, true, true>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2, out, L.recsA,
                       L.tcount, L.slots, wcap, P, sprobe, sorder);
  else if (vec)
    hipLaunchKernelGGL((iou_stream3_kernel<GEOM, true>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2, out, L.recsA,
                       L.tcount, L.slots, wcap, P, sprobe, sorder, AssignZero(), (unsigned*)nullptr, thin);
  else
    hipLaunchKernelGGL((iou_stream3_kernel<GEOM, false>), grid, dim3(T_THREADS), 0, stream, b1, n1, b2, n2, out, L.recsA,
                       L.tcount, L.slots, wcap, P, sprobe, sorder, AssignZero(), (unsigned*)nullptr, thin);
  if (fast)
    hipLaunchKernelGGL((iou_drain3_kernel<GEOM, true>), dim3(blocks), dim3(T_THREADS), 0, stream, b1, n1, b2, n2, iof, L.recsA,
                       L.tcount, L.slots, L.tiles_x, (int)tiles, out, prepared ? P.rec : (const BoxRec*)nullptr, dstamps, AssignOut(),
                       (int)g_r3_iou_dyn, (const unsigned*)nullptr, 0, thin);
  else
    hipLaunchKernelGGL((iou_drain3_kernel<GEOM, false>), dim3(blocks), dim3(T_THREADS), 0, stream, b1, n1, b2, n2, iof, L.recsA,
                       L.tcount, L.slots, L.tiles_x, (int)tiles, out, prepared ? P.rec : (const BoxRec*)nullptr, dstamps, AssignOut(),
                       (int)g_r3_iou_dyn, (const unsigned*)nullptr, 0, thin);
  return 0;
}

}  // namespace

namespace {
// obb_overlaps' epilogue (box_iou_rotated_wrapper.py:53-60): rows of b1-boxes and columns of b2-boxes with
// min(w, h) < 1e-3 are zeroed.  One thread looks at one line (row or column); a wave zeroes its (rare) thin lines
// together.  torch.min propagates NaN, and NaN < 1e-3 is false: a box with a NaN side is not thin.
__global__ __launch_bounds__(256) void iou_zero_thin_kernel(const float* __restrict__ b1, int n1,
                                                            const float* __restrict__ b2, int n2,
                                                            float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  bool thin = false;
  if (t < n1 + n2) {
    const float* b = t < n1 ? b1 + (size_t)t * 5 : b2 + (size_t)(t - n1) * 5;
    const float w = b[2], h = b[3];
    thin = !(w != w || h != h) && fminf(w, h) < 0.001f;
  }
  unsigned long long m = __ballot(thin);
  while (m) {
    const int line = t - lane + __builtin_ctzll(m);  // wave-uniform
    m &= m - 1;
    if (line < n1) {
      float* o = out + (size_t)line * n2;
      for (int c = lane; c < n2; c += 64) o[c] = 0.f;
    } else {
      float* o = out + (line - n1);
      for (int r = lane; r < n1; r += 64) o[(size_t)r * n2] = 0.f;
    }
  }
}
}  // namespace

int r3k_iou_zero_thin(const float* b1, int n1, const float* b2, int n2, float* out, hipStream_t stream) {
  if (n1 <= 0 || n2 <= 0) return 0;
  hipLaunchKernelGGL(iou_zero_thin_kernel, dim3((unsigned)(((long long)n1 + n2 + 255) / 256)), dim3(256), 0, stream, b1, n1,
                     b2, n2, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t r3k_iou_workspace_bytes(int n1, int n2) {
  if (n1 <= 0 || n2 <= 0) return 256;
  return pipe_layout(n1, n2, nullptr, nullptr);
}

int r3k_iou_mat(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                void* ws, size_t ws_bytes, hipStream_t stream, const void* prepared, int* thin_done) {
  if (thin_done) *thin_done = 0;
  if (n1 == 0 || n2 == 0) return 0;
  int rc;
  switch (geom) {
    case 1: rc = launch_mat<1>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream, prepared); break;
    case 2: rc = launch_mat<2>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream, prepared); break;
    case 3: rc = launch_mat<3>(iof, b1, n1, b2, n2, out, ws, ws_bytes, stream, prepared, thin_done); break;
    default: return -1;
  }
  if (rc) return rc;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t r3k_iou_prepared_bytes(int n2) { return n2 > 0 ? colprep_layout(n2, nullptr, nullptr) : 256; }

// the exact records, conservative-test data and 256-column bounding boxes of a box list used as the COLUMNS of many
// matrices / assignments (the anchor grid)
int r3k_iou_prepare_columns(int geom, const float* b2, int n2, void* prepared, size_t bytes, hipStream_t stream) {
  if (n2 <= 0 || !b2 || !prepared || (reinterpret_cast<uintptr_t>(prepared) & 15)) return -1;
  if (bytes < r3k_iou_prepared_bytes(n2)) return -3;
  ColPrep P;
  colprep_layout(n2, prepared, &P);
  const dim3 grid((n2 + 255) / 256), block(256);
#define R3_PREP(G) \
  hipLaunchKernelGGL(iou_prepare_kernel<G>, grid, block, 0, stream, b2, n2, const_cast<BoxRec*>(P.rec), \
                     const_cast<float4*>(P.rej), const_cast<float*>(P.rad), const_cast<float4*>(P.wbox), \
                     const_cast<int4*>(P.hdr))
  switch (geom) {
    case 1: R3_PREP(1); break;
    case 2: R3_PREP(2); break;
    case 3: R3_PREP(3); break;
    default: return -1;
  }
#undef R3_PREP
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
  // round 5, the stream + drain queue of the matrix path (tiles <= P_MAX_TILES): gqueue / qiou hold these instead
  bool tiled;
  int tiles_x, tiles_y;
  int* tcount;
  unsigned short* slots;
  float* siou;
};

// Two forms share one workspace size (the larger): the tile queue of the matrix path (tile counts, one 8 KB slot and
// 8192 IoUs per 8 x 1024 tile), taken when the matrix has at most P_MAX_TILES tiles, else the global queue of rounds 2-4.
inline size_t assign_layout(int n1, int n2, void* ws, AssignLayout* L, const bool tiled);
// (a tile costs 40 KB however empty it is -- a skinny problem, n1 < 8 or few columns, needed several times
// the global queue's bytes.  The tile form is taken only where its layout is no larger than the global queue's, or
// small in absolute terms; the workspace size follows the same rule, so callers that sized it once stay valid.)
inline bool assign_tiled_fits(int n1, int n2) {
  const long long tx = ((long long)n2 + T_COLS - 1) / T_COLS, ty = ((long long)n1 + P_ROWS - 1) / P_ROWS;
  if (tx * ty > P_MAX_TILES) return false;
  const size_t tiled = assign_layout(n1, n2, nullptr, nullptr, true), global_q = assign_layout(n1, n2, nullptr, nullptr, false);
  return tiled <= std::max(global_q, (size_t)1 << 20);
}
inline bool assign_tiled(int n1, int n2) {
  return assign_tiled_fits(n1, n2) && g_r3_iou_impl != 3;  // (iou_impl 3: the global-queue form whatever the size, for the A/B)
}

inline size_t assign_layout(int n1, int n2, void* ws, AssignLayout* L, const bool tiled) {
  const long long tx = ((long long)n2 + T_COLS - 1) / T_COLS, ty = ((long long)n1 + P_ROWS - 1) / P_ROWS;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* counter = take(256);
  char* ra = take((size_t)n1 * sizeof(BoxRec));
  char* rb = take((size_t)n2 * sizeof(BoxRec));
  char* rk = take((size_t)n1 * 8);
  char* ck = take((size_t)n2 * 8);
  char* lq = take((size_t)n2 * 4);
  char *gq = nullptr, *qi = nullptr, *tc = nullptr, *sl = nullptr, *si = nullptr;
  if (tiled) {
    tc = take((tcount_ints(tx * ty) + P_TICKET_PAD) * 4);
    sl = take((size_t)(tx * ty) * P_SLOT * 2);
    si = take((size_t)(tx * ty) * P_ROWS * T_COLS * 4);
  } else {
    gq = take((size_t)n1 * n2 * 4);
    qi = take((size_t)n1 * n2 * 4);
  }
  if (L) {
    L->counter = (unsigned*)counter; L->recsA = (BoxRec*)ra; L->recsB = (BoxRec*)rb; L->gqueue = (unsigned*)gq;
    L->qiou = (float*)qi; L->rowkey = (u64k*)rk; L->colkey = (u64k*)ck; L->lowq = (int*)lq;
    L->tiled = tiled; L->tiles_x = (int)tx; L->tiles_y = (int)ty; L->tcount = (int*)tc; L->slots = (unsigned short*)sl;
    L->siou = (float*)si;
  }
  return off + 256;
}

template <int GEOM>
void launch_assign(const float* gts, int n1, const float* boxes, int n2, const AssignLayout& L, float min_pos_iou,
                   int match_low, int assign_all, hipStream_t stream, const void* prepared) {
  // (the prepared buffer is read only behind the stream kernel that checks its header: the tile form with n2 % 4 == 0;
  // the other forms rebuild the columns' data from the boxes)
  const bool use_prep = prepared && L.tiled && n2 % 4 == 0;
  ColPrep P = ColPrep();
  if (use_prep) colprep_layout(n2, prepared, &P);
  const BoxRec* recsB = L.recsB;
  const int nmax = n1 > n2 ? n1 : n2;
  if (!L.tiled)
    hipLaunchKernelGGL(assign_init_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, L.rowkey, n1, L.colkey,
                       L.lowq, n2, L.counter);
  if (L.tiled) {
    // round 5: the matrix path's queue -- stream3 (per-wave segments, no atomics, the wave-level bounding box; no
    // matrix: out = nullptr) -> the drain with the keys as its result -> the low-quality sweep tile by tile.
    // (iou_stream_kernel + one global queue behind an atomic counter: 37 us of stream at 128 x 196 416, this: ~20)
    const int tiles = L.tiles_x * L.tiles_y;
    const dim3 sgrid(L.tiles_x, L.tiles_y);
    const int qcap3 = g_r3_iou_qcap;  // (one read per call)
    const int wcap3 = qcap3 > 0 && qcap3 < P_WSEG ? qcap3 : P_WSEG;  // (small: the dense-tile path)
    const AssignZero az{L.rowkey, L.colkey, L.lowq};  // (round 5: assign_init_kernel's work, one launch less)
    if (use_prep)
      hipLaunchKernelGGL((iou_stream3_kernel<GEOM, true, true>), sgrid, dim3(T_THREADS), 0, stream, gts, n1, boxes, n2,
                         (float*)nullptr, L.recsA, L.tcount, L.slots, wcap3, P, 0, -1, az);
    else
      hipLaunchKernelGGL((iou_stream3_kernel<GEOM, false>), sgrid, dim3(T_THREADS), 0, stream, gts, n1, boxes, n2,
                         (float*)nullptr, L.recsA, L.tcount, L.slots, wcap3, P, 0, -1, az);
    const bool fast = g_r3_clip_impl == 0;
    unsigned long long pairs3 = (unsigned long long)n1 * n2;
    int blocks3 = (int)((pairs3 + T_THREADS - 1) / T_THREADS);
    const int dwgs3 = g_r3_iou_dwgs;  // (one read per call)
    const int maxb3 = dwgs3 > 0 ? dwgs3 : fast ? 4 * r3_cu_count() : 1536;
    if (blocks3 > maxb3) blocks3 = maxb3;
    const int n1_lds3 = n1 < 2048 ? n1 : 2048;
    const AssignOut ao{L.siou, L.rowkey, L.colkey, n1_lds3, R3_HAS_PROBES ? (int)g_r3_fr_walk - 2000 : 0};
    if (fast)
      // (v3 with the keys as its result: the straight-line form spills 12 B per lane at 128 registers -- the list form there)
      hipLaunchKernelGGL((iou_drain3_kernel<GEOM, GEOM != 3, true>), dim3(blocks3), dim3(T_THREADS),
                         (size_t)n1_lds3 * sizeof(u64k), stream, gts, n1, boxes, n2, 0, L.recsA, L.tcount, L.slots,
                         L.tiles_x, tiles, (float*)nullptr, use_prep ? P.rec : (const BoxRec*)nullptr,
                         (unsigned long long*)nullptr, ao, (int)g_r3_iou_dyn);
    else
      hipLaunchKernelGGL((iou_drain3_kernel<GEOM, false, true>), dim3(blocks3), dim3(T_THREADS),
                         (size_t)n1_lds3 * sizeof(u64k), stream, gts, n1, boxes, n2, 0, L.recsA, L.tcount, L.slots,
                         L.tiles_x, tiles, (float*)nullptr, use_prep ? P.rec : (const BoxRec*)nullptr,
                         (unsigned long long*)nullptr, ao, (int)g_r3_iou_dyn);
    if (match_low)
      hipLaunchKernelGGL(assign_lowq3_kernel, dim3(tiles, 4), dim3(256), 0, stream, L.tcount, L.slots, L.tiles_x, L.siou, n1,
                         n2, L.rowkey, min_pos_iou, assign_all, L.lowq);
    return;
  }
  dim3 grid((n2 + T_COLS - 1) / T_COLS, (n1 + S_ROWS - 1) / S_ROWS);
  hipLaunchKernelGGL((iou_stream_kernel<GEOM, false>), grid, dim3(T_THREADS), 0, stream, gts, n1, boxes, n2,
                     (float*)nullptr, L.recsA, L.recsB, L.gqueue, L.counter, P.rej, P.rad);
  unsigned long long pairs = (unsigned long long)n1 * n2;
  int blocks = (int)((pairs + T_THREADS - 1) / T_THREADS);
  if (blocks > 2048) blocks = 2048;
  // rows reduced in LDS (16 KB); the rest goes straight to global.  (Look + global atomicMax for the rows as for the
  // columns: 243 us instead of 55 -- a few hundred addresses take every pair's update.)
  const int n1_lds = n1 < 2048 ? n1 : 2048;
  // (LDS-list clip, measured: 512 -> 62 us, 1024 -> 58, 1536 -> 55, 2048 -> 53; the straight-line clip holds 4 workgroups per CU)
  const bool fast_g = GEOM == 1 && g_r3_clip_impl == 0;  // (one read per call)
  const int dwgs_g = g_r3_iou_dwgs;
  const int dmax = dwgs_g > 0 ? dwgs_g : fast_g ? 4 * r3_cu_count() : 2048;
  if (fast_g)
    hipLaunchKernelGGL((assign_drain_kernel<GEOM, GEOM == 1>), dim3(blocks < dmax ? blocks : dmax), dim3(T_THREADS),
                       (size_t)n1_lds * sizeof(u64k), stream, L.recsA, recsB, n2, L.gqueue, L.counter, L.qiou, L.rowkey,
                       L.colkey, n1_lds);
  else
    hipLaunchKernelGGL((assign_drain_kernel<GEOM, false>), dim3(blocks < dmax ? blocks : dmax), dim3(T_THREADS),
                       (size_t)n1_lds * sizeof(u64k), stream, L.recsA, recsB, n2, L.gqueue, L.counter, L.qiou, L.rowkey,
                       L.colkey, n1_lds);
  if (match_low)
    hipLaunchKernelGGL(assign_lowq_kernel, dim3(blocks), dim3(256), 0, stream, L.gqueue, L.counter, L.qiou, n2,
                       L.rowkey, min_pos_iou, assign_all, L.lowq);
}

}  // namespace

size_t r3k_iou_assign_workspace_bytes(int n1, int n2) {
  if (n1 <= 0 || n2 <= 0) return 256;
  // (either form must fit: the option that picks the form may change between this query and the call)
  const size_t global_q = assign_layout(n1, n2, nullptr, nullptr, false);
  return assign_tiled_fits(n1, n2) ? std::max(global_q, assign_layout(n1, n2, nullptr, nullptr, true)) : global_q;
}

int r3k_iou_assign(int geom, const float* gts, int n1, const float* boxes, int n2, float pos_thr, float neg_thr,
                   float min_pos_iou, int match_low, int assign_all, int64_t* assigned, float* max_overlaps,
                   int64_t* argmax, float* gt_max, int64_t* gt_argmax, void* ws, size_t ws_bytes,
                   hipStream_t stream, const void* prepared, const int64_t* gt_labels, int64_t* labels) {
  if (n1 <= 0 || n2 <= 0 || !gts || !boxes || !assigned || !max_overlaps || !ws) return -1;
  if ((gt_labels == nullptr) != (labels == nullptr)) return -1;
  if ((gt_max == nullptr) != (gt_argmax == nullptr)) return -1;
  if ((unsigned long long)n1 * (unsigned long long)n2 >= 0xffffffffULL) return -1;
  if (ws_bytes < r3k_iou_assign_workspace_bytes(n1, n2)) return -3;
  AssignLayout L;
  assign_layout(n1, n2, ws, &L, assign_tiled(n1, n2));
  switch (geom) {
    case 1: launch_assign<1>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream, prepared); break;
    case 2: launch_assign<2>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream, prepared); break;
    case 3: launch_assign<3>(gts, n1, boxes, n2, L, min_pos_iou, match_low, assign_all, stream, prepared); break;
    default: return -1;
  }
  const int nmax = n1 > n2 ? n1 : n2;
  hipLaunchKernelGGL(assign_final_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, L.rowkey, n1, L.colkey, L.lowq,
                     n2, pos_thr, neg_thr, min_pos_iou, match_low, assign_all, assigned, max_overlaps, argmax, gt_max,
                     gt_argmax, gt_labels, labels);
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
