// r3_frb.hip -- Feature Refinement sampler, BACKWARD on channels_last memory (N, H, W, C).
//
// Replaces feature_refine_backward_kernel (fr/src/feature_refine_kernel.cu:165-230; called from
// feature_refine_module.py:28-40) for channels_last pipelines.  The reference scatters: one thread per
// element, five global float atomics (the identity term + the four bilinear taps).  Here the scatter is
// turned into a GATHER, which needs no atomics and no zero-fill and sums in a fixed order:
//
//   grad_in[q] = grad_out[q] + sum over the entries e of cell q:  w_e * grad_out[src_e]
//
// where the entries of a cell are the (source position, tap) pairs that touch it.  In channels_last a
// position's gradient is one contiguous row of C floats (1 KB at C = 256): a wavefront owns a cell, a lane
// owns 4 channels, every row access is one 16-byte load per lane.
//
//   index   (boxes only, once per level and step, independent of C): the inverse tap index in CSR form --
//           per cell {start, len}, per entry {key, weight}, key = (source row * 8192 + source column) * 32 +
//           point * 4 + tap -- built in ONE launch.  A workgroup owns a band of cell rows.  It scans the
//           sample rows of all sources of its image (a level's boxes are 320 KB: L2-resident), keeps the few
//           whose taps reach the band, and derives the band's base in the entry array from a private count of
//           the entries that fall in earlier rows -- so no cross-workgroup prefix and no global atomics.
//           points = 1 (frb_index_sort_kernel): the band's entries are collected in LDS in a
//           fixed (wave, source, tap) order and stably sorted by cell: the sorted list IS the band's CSR slice,
//           every cell's entries in one fixed order, no LDS atomics (they retire about one lane per 3 cycles and
//           serialise on piles of sources sampling one cell -- what a trained detector produces around every
//           object).  A band with more entries than the
//           LDS list holds, and points = 5, take the general form (frb_index_general: LDS counters, fill, sort).
//   gather  the sum above.  A 512-thread workgroup owns a 4 x 4 tile of cells and its TRANSPOSE (the reference
//           samples row <- x_ctr, column <- y_ctr: the sources of tile (i, j) lie in tile (j, i)); the two
//           tiles' gradient rows are shared through LDS, so most entries are one LDS read instead of a row
//           load, and the second use of every row meets the first in the same compute unit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <hip/hip_ext.h>

#include "r3_fr_tap.h"
#include "r3_kernels.h"

namespace {

typedef unsigned long long u64;
typedef float frb_v4 __attribute__((ext_vector_type(4)));  // (the non-temporal builtins do not take HIP's float4)

constexpr int IX_T = 1024;         // threads of an index workgroup
constexpr int IX_WAVES = IX_T / 64;
constexpr int IX_MAXCELLS = 4096;  // cells of one band (general form)
constexpr int IX_SORT_MAX = 48;    // general form: lists up to this length are sorted (longer ones keep the fill order)
constexpr int IXS_CELLS = 256;     // sort form: cells per band (an average of 1024 entries)
constexpr int IXS_CAP = 8192;      // sort form: entries a band sorts in LDS
constexpr int IXS_SEG = 256;       // sort form: sources per wave that reach the band

// An entry's first word tells the gather where the source's gradient row is: bits 0..25 the source position
// sy * W + sx; bit 31 set: the source lies in the 4 x 4 cell tile of the entry's own cell (bits 26..29 = its slot
// (sy & 3) * 4 + (sx & 3), bit 30 clear) or in that tile's partner in the gather's paired launch (bit 30 set) --
// the transposed tile (tx, ty), for a diagonal tile its neighbour ty ^ 1 -- whose rows the workgroup holds in LDS.
__device__ __forceinline__ int frb_entry_code(int sy, int sx, int y, int x, int W) {
  const int s = sy * W + sx, ty = y >> 2, tx = x >> 2, syt = sy >> 2, sxt = sx >> 2;
  const int slot = (sy & 3) * 4 + (sx & 3);
  if (syt == ty && sxt == tx) return (int)(0x80000000u | ((unsigned)slot << 26) | (unsigned)s);
  const int oy = ty != tx ? tx : (ty ^ 1), ox = ty != tx ? ty : (ty ^ 1);
  if (syt == oy && sxt == ox) return (int)(0xc0000000u | ((unsigned)slot << 26) | (unsigned)s);
  return s;
}

struct FrbLayout {
  int2* cellinfo;  // [N][HW] {start (in the image's entry array), len}
  int2* entries;   // [N][4 * points * HW] {key, weight bits}
  size_t bytes;
};

inline FrbLayout frb_layout(void* ws, int N, int H, int W, int points) {
  FrbLayout L;
  const size_t hw = (size_t)H * W;
  const size_t ci = (N * hw * sizeof(int2) + 255) & ~(size_t)255;
  L.cellinfo = reinterpret_cast<int2*>(ws);
  L.entries = reinterpret_cast<int2*>(static_cast<char*>(ws) + ci);
  L.bytes = ci + N * hw * 4 * points * sizeof(int2) + 256;
  return L;
}

inline int general_band_rows(int H, int W) {
  int r = IX_T / W;  // one cell per thread in the common shapes
  if (r < 1) r = 1;
  return r > H ? H : r;
}

inline int sort_band_rows(int H, int W) {
  int r = IXS_CELLS / W;
  if (r < 1) r = 1;
  return r > H ? H : r;
}

// ------------------------------------------------------------------------------------------------
// general form: any band up to IX_MAXCELLS cells, points 1 or 5
// ------------------------------------------------------------------------------------------------
template <int POINTS>
__device__ __forceinline__ void frb_index_general_body(const float* __restrict__ bx, float scale, int H, int W, int r0,
                                                       int r1, int* ix, int2* __restrict__ ci, int2* __restrict__ en) {
  const int CB = (r1 - r0) * W;
  int* cnt = ix;           // [CB] entries per cell, later the fill cursor
  int* start = cnt + CB;   // [CB] exclusive prefix inside the band
  int* part = start + CB;  // [IX_WAVES] wave totals
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = H * W;
  for (int i = tid; i < CB; i += IX_T) cnt[i] = 0;
  __syncthreads();
  // pass A: entries per cell of the band; entries in earlier rows (private count)
  int before = 0;
  for (int s = tid; s < HW; s += IX_T) {
    TapYX taps[POINTS];
    make_taps_yx<POINTS>(bx + (size_t)s * 5, scale, H, W, taps);
#pragma unroll
    for (int p = 0; p < POINTS; p++) {
      const TapYX& t = taps[p];
      if (!t.valid) continue;
      before += (t.yl < r0 ? 2 : 0) + (t.yh < r0 ? 2 : 0);
      if (t.yl >= r0 && t.yl < r1) {
        atomicAdd(&cnt[(t.yl - r0) * W + t.xl], 1);
        atomicAdd(&cnt[(t.yl - r0) * W + t.xh], 1);
      }
      if (t.yh >= r0 && t.yh < r1) {
        atomicAdd(&cnt[(t.yh - r0) * W + t.xl], 1);
        atomicAdd(&cnt[(t.yh - r0) * W + t.xh], 1);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) part[wave] = before;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < IX_WAVES; w++) base += part[w];
  __syncthreads();
  // exclusive prefix of cnt over the band's cells (CPT consecutive cells per thread)
  const int CPT = (CB + IX_T - 1) / IX_T;
  const int lo = min(tid * CPT, CB), hi = min(lo + CPT, CB);
  int mine = 0;
  for (int i = lo; i < hi; i++) mine += cnt[i];
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; w++) wbase += part[w];
  int run = base + wbase + incl - mine;
  for (int i = lo; i < hi; i++) {
    const int c = cnt[i];
    start[i] = run;
    ci[r0 * W + i] = make_int2(run, c);
    run += c;
  }
  __syncthreads();
  for (int i = tid; i < CB; i += IX_T) cnt[i] = 0;  // fill cursors
  __syncthreads();
  // pass B: the entries at their places (order inside a list = arrival order; sorted below)
  for (int s = tid; s < HW; s += IX_T) {
    TapYX taps[POINTS];
    make_taps_yx<POINTS>(bx + (size_t)s * 5, scale, H, W, taps);
    const int sy = s / W, sx = s - sy * W;
#pragma unroll
    for (int p = 0; p < POINTS; p++) {
      const TapYX& t = taps[p];
      if (!t.valid) continue;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int y = (d >> 1) ? t.yh : t.yl, x = (d & 1) ? t.xh : t.xl;
        if (y >= r0 && y < r1) {
          const int c = (y - r0) * W + x;
          const int rank = atomicAdd(&cnt[c], 1);
          en[start[c] + rank] = make_int2(frb_entry_code(sy, sx, y, x, W), __float_as_int(t.w[d]));
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // every list in (code, weight bits) order: the gather then sums in one fixed order (entries equal in both commute)
  for (int i = lo; i < hi; i++) {
    const int len = cnt[i];
    if (len < 2 || len > IX_SORT_MAX) continue;
    int2* e = en + start[i];
    for (int a = 1; a < len; a++) {
      const int2 v = e[a];
      int b = a - 1;
      while (b >= 0) {
        const int2 u = e[b];
        if (u.x < v.x || (u.x == v.x && u.y <= v.y)) break;
        e[b + 1] = u;
        b--;
      }
      e[b + 1] = v;
    }
  }
}

template <int POINTS>
__global__ __launch_bounds__(IX_T) void frb_index_general_kernel(const float* __restrict__ boxes, float scale, int H,
                                                                 int W, int R, int2* __restrict__ cellinfo,
                                                                 int2* __restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) int ix[];
  const int n = blockIdx.y, r0 = blockIdx.x * R, r1 = min(H, r0 + R);
  const size_t HW = (size_t)H * W;
  frb_index_general_body<POINTS>(boxes + n * HW * 5, scale, H, W, r0, r1, ix, cellinfo + n * HW,
                                 entries + n * HW * 4 * POINTS);
}

__device__ __attribute__((noinline)) void frb_index_general_call(const float* __restrict__ bx, float scale, int H, int W,
                                                                 int r0, int r1, int* ix, int2* __restrict__ ci,
                                                                 int2* __restrict__ en) {
  frb_index_general_body<1>(bx, scale, H, W, r0, r1, ix, ci, en);
}

// ------------------------------------------------------------------------------------------------
// sort form (points = 1)
// ------------------------------------------------------------------------------------------------
struct IxsLds {
  u64 sk[IXS_CAP];              // (cell in the band << 56) | (source << 34) | (tap << 32) | weight bits
  int seg[IX_WAVES * IXS_SEG];  // sources that reach the band, per wave
  int cst[IXS_CELLS], cend[IXS_CELLS];
  int hist[256 * 2 * IX_WAVES + IX_WAVES];  // the sort's block counts (one 8-bit pass: E <= 2; two 4-bit passes: E <= 8)
  int wcnt[IX_WAVES];
  int part[IX_WAVES];
  int total, over;
};

// Stable sort of E * 1024 keys by their top byte (the cell), E per thread (element e * 1024 + tid): two passes of a
// 4-bit counting sort.  A wavefront ranks its 64 keys with 16 ballots (no LDS atomics), the 16 x (16 E) block counts are
// scanned digit-major in LDS, every key goes to its place.  Stable, so the order inside a cell is the list's own
// (deterministic) order.  (A full bitonic sort of (cell, source, tap) was 18-34 us of this kernel: 55 stages for
// 1024 keys, 0.3-0.6 us each.)
template <int E>
__device__ __forceinline__ void ixs_sort(u64* sk, int* hist, const int tid) {
  constexpr int NB = E * IX_WAVES;  // 64-key blocks
  const int lane = tid & 63, wave = tid >> 6;
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = sk[e * IX_T + tid];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    int rank[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int dg = (int)(v[e] >> (56 + 4 * pass)) & 15;
      int r = 0, c = 0;
#pragma unroll
      for (int d = 0; d < 16; d++) {
        const u64 m = __ballot(dg == d);
        if (dg == d) r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == d) c = __popcll(m);
      }
      rank[e] = r;
      if (lane < 16) hist[lane * NB + e * IX_WAVES + wave] = c;
    }
    __syncthreads();
    // exclusive scan of the 16 NB counts (digit-major): CPT consecutive counts per thread
    {
      constexpr int CNT = 16 * NB, CPT = (CNT + IX_T - 1) / IX_T;
      int c[CPT], mine = 0;
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        const int i = tid * CPT + q;
        c[q] = i < CNT ? hist[i] : 0;
        mine += c[q];
      }
      int incl = mine;
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
      }
      if (lane == 63) hist[CNT + wave] = incl;
      __syncthreads();
      int run = incl - mine;
      for (int w = 0; w < wave; w++) run += hist[CNT + w];
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        const int i = tid * CPT + q;
        if (i < CNT) hist[i] = run;
        run += c[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int dg = (int)(v[e] >> (56 + 4 * pass)) & 15;
      sk[hist[dg * NB + e * IX_WAVES + wave] + rank[e]] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = sk[e * IX_T + tid];
    __syncthreads();
  }
}

// The same sort in ONE pass over the whole byte, for E <= 2 (up to 2048 keys: 256 x 32 block counts fit the table):
// a wavefront finds the lanes that hold its key's byte with 8 ballots (one per bit; the two 4-bit passes took 16 ballots
// each with a rank and a count per digit), the lowest such lane writes the count.  Half the barriers, a quarter of the
// ranking instructions (clock stamps, tools/probes/frb_index_probe.hip: the sort was a third of the kernel).
template <int E>
__device__ __forceinline__ void ixs_sort8(u64* sk, int* hist, const int tid) {
  constexpr int NB = E * IX_WAVES;  // 64-key blocks
  constexpr int CNT = 256 * NB, CPT = CNT / IX_T;
  const int lane = tid & 63, wave = tid >> 6;
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = sk[e * IX_T + tid];
#pragma unroll
  for (int q = 0; q < CPT; q++) hist[q * IX_T + tid] = 0;
  __syncthreads();
  int rank[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int dg = (int)(v[e] >> 56);
    u64 m = ~0ULL;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (dg >> b) & 1;
      const u64 bb = __ballot(bit);
      m &= bit ? bb : ~bb;
    }
    rank[e] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (rank[e] == 0) hist[dg * NB + e * IX_WAVES + wave] = __popcll(m);
  }
  __syncthreads();
  {  // exclusive scan of the 256 NB counts (digit-major): CPT consecutive counts per thread
    int c[CPT], mine = 0;
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      c[q] = hist[tid * CPT + q];
      mine += c[q];
    }
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) hist[CNT + wave] = incl;
    __syncthreads();
    int run = incl - mine;
    // (all 16 wave totals requested together: `for (w < wave)` was a serial chain of up to 15 LDS round trips)
#pragma unroll
    for (int w = 0; w < IX_WAVES; w++) {
      const int t = hist[CNT + w];
      run += w < wave ? t : 0;
    }
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      hist[tid * CPT + q] = run;
      run += c[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) sk[hist[(int)(v[e] >> 56) * NB + e * IX_WAVES + wave] + rank[e]] = v[e];
  __syncthreads();
}

// Where a band's SELL-64 rows go when the index kernel re-lays them itself (the NCHW gather's form of the lists,
// frn_gather_kernel below): hdr == nullptr: CSR only.
struct FrbSellOut {
  int* hdr;    // [N][slices] padded list length | tail flag << 16
  int2* rows;  // [N][slices][cap / 8 batches][4 row pairs][64 lanes][2 entries] {weight, source cell byte offset}
  int ascale, cap, pitch, slices;
};

// entry r of the cell in lane `lane` of slice `slice` (one image's rows)
__device__ __forceinline__ size_t frb_sell_at(int slice, int lane, int r, int cap) {
  return ((((size_t)slice * (cap >> 3) + (r >> 3)) * 4 + ((r >> 1) & 3)) * 64 + lane) * 2 + (r & 1);
}

// One slice's rows from the CSR lists of its 64 cells (one wavefront): rows in PAIRS -- {weight, cell, weight, cell}
// of entries 2p and 2p + 1, 16 bytes per lane, 1 KB per wavefront load; a batch is four row pairs; a slice's batches
// lie at a fixed stride.  ascale = 4 * CP: the entry carries the byte offset of the source cell in the gather's LDS
// plane.  Lists are padded to the slice's longest (rounded up to even) with {zero cell, weight 0}.
__device__ __forceinline__ void frb_sell_slice(const int2* __restrict__ cellinfo_n, const int2* __restrict__ entries_n,
                                               int slice, int lane, int HW, int W, int P, int zero_cell, int ascale,
                                               int cap, int* __restrict__ slicehdr_n, int4* __restrict__ sell_n) {
  const int q = slice * 64 + lane;
  int2 ci = make_int2(0, 0);
  if (q < HW) ci = cellinfo_n[q];
  int m = ci.y;
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  const int mp = min(cap, (m + 1) & ~1);
  if (lane == 0) slicehdr_n[slice] = mp | ((m > cap) << 16);
  const int2* en = entries_n + ci.x;
  int4* out = sell_n + (size_t)slice * (cap >> 3) * 256 + lane;
  auto entry = [&](int e) -> int2 {
    if (e >= ci.y) return make_int2(zero_cell * ascale, 0);
    const int2 v = en[e];
    const int s = v.x & 0x3ffffff, sy = s / W, sx = s - sy * W;
    return make_int2((sy * P + sx) * ascale, v.y);
  };
  for (int e = 0; e < mp; e += 2) {
    const int2 e0 = entry(e), e1 = entry(e + 1);
    out[(size_t)(e >> 1) * 64] = make_int4(e0.y, e0.x, e1.y, e1.x);
  }
}

// CSR = false (only with SELL output): the band writes its SELL rows and nothing else -- no {start, len} / entry
// arrays, so no count of the entries in earlier rows and no column test in the scan (sources out of range in x are
// dropped with their taps in phase B): the scan loads one float per source instead of two.  A band with a list longer
// than the SELL capacity takes the general form, which writes the CSR lists of ITS cells (the gather reads them for
// exactly those cells).
// TAB (round 6): the sources' sample points come from the level's TAP TABLE -- per image [y: HW floats][x: HW floats],
// the clamped sample point of every position (cell_tap: r3_fr.hip; an invalid sample = row H + 1), written by the
// forward pass of the same boxes (fr_cell_table_kernel, or the channels_last samplers' `tab` output) -- instead of
// from the 20-byte box records: the scan reads 4 contiguous bytes per source (64 KB per image at level 0, coalesced)
// where the box form pulls all 327 KB of records through the compute unit for one float in five, and needs no scale,
// no validity test and no column.  make_tap_yx is idempotent on its own clamps (the table's (y, x) give the same cells
// and bit-identical weights), and the sources are walked in the same order: the index is byte-identical to the box
// form's (tests/test_gpu_fr_keys.py).
template <bool CSR, bool TAB = false>
__device__ __forceinline__ void frb_index_sort_body(const float* __restrict__ boxes, float scale, int H, int W, int R,
                                                    int2* __restrict__ cellinfo, int2* __restrict__ entries,
                                                    const FrbSellOut& so, const int band, const int bands, const int n,
                                                    u64* __restrict__ stamps, const float* __restrict__ tab = nullptr) {
  extern __shared__ __attribute__((aligned(16))) int ix[];
  IxsLds& S = *reinterpret_cast<IxsLds*>(ix);
  // (tools/probes/frb_index_probe.hip: clock stamps of one workgroup at the phase boundaries)
  auto stamp = [&](int k) {
#ifdef R3_PROBES
    if (stamps && band == bands / 2 && n == 0 && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();
#endif
  };
  stamp(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = band * R, r1 = min(H, r0 + R);
  const int HW = H * W, CB = (r1 - r0) * W;
  const float* bx = boxes + (size_t)n * HW * 5;
  const float* ty_g = TAB ? tab + (size_t)n * 2 * HW : nullptr;  // [y: HW][x: HW] of this image
  // the sample point of source s as make_tap_yx takes it (row, column)
  auto tap_of = [&](const int s) -> TapYX {
    if (TAB) return make_tap_yx(H, W, ty_g[s], ty_g[HW + s]);
    return make_tap_yx(H, W, bx[(size_t)s * 5] * scale, bx[(size_t)s * 5 + 1] * scale);
  };
  int2* ci = cellinfo + (size_t)n * HW;
  int2* en = entries + (size_t)n * HW * 4;
  if (tid == 0) S.over = 0;
  for (int i = tid; i < IXS_CELLS; i += IX_T) S.cst[i] = S.cend[i] = 0;
  __syncthreads();
  // A: the sample rows of every source: entries in earlier rows (the band's base), sources that reach the band
  const float band_lo = (float)(r0 - 1);
  const float band_hi = r1 < H ? (float)r1 : __int_as_float(__float_as_int((float)H) + 1);  // (last band: y <= H)
  int before = 0, wc = 0;
  bool over = false;
  constexpr int UA = 8;  // sources per thread and round: their UA loads are in flight together (one workgroup per
                         // compute unit: a round is one L2 latency).  The phase is bound by its ~28 instructions per
                         // source at 4 waves per SIMD (16.0 k cycles with guarded loads and a validity branch ->
                         // 13.3 k with clamped unconditional loads and a branch-free body; a pre-pass that stored the
                         // sample rows as one int per source made it no shorter; round 6, table form: all 16 of a
                         // level-0 map in one round trip: 14.5 -> 14.1 us, inside the run-to-run spread -- the phase
                         // is its instruction issue: ~15 per source and wave at 4 waves per SIMD).
  for (int s0 = 0; s0 < HW; s0 += IX_T * UA) {
    float yv[UA], xv[UA];
#pragma unroll
    for (int u = 0; u < UA; u++) {
      // (unconditional loads at a clamped index: behind `if (s < HW)` every load waited for itself -- 16 L2 round
      // trips in a row, the whole of this phase)
      const int s = s0 + u * IX_T + tid;
      if (TAB) {
        const float y = ty_g[min(s, HW - 1)];  // (a wavefront: 256 contiguous bytes)
        yv[u] = s < HW ? y : 3.0e38f;          // (beyond the map: as an invalid sample, below every band's upper end)
        xv[u] = 0.f;
        continue;
      }
      const float* bp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bx) + (unsigned)min(s, HW - 1) * 20u);
      const float y = bp[0], x = CSR ? bp[1] : 0.f;
      yv[u] = s < HW ? y : -3.0e38f;  // (beyond the map: out of range below)
      xv[u] = s < HW ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UA; u++) {
      const int s = s0 + u * IX_T + tid;
      // (branch-free: every workgroup of the image walks all sources, so this body is the kernel's instruction
      // budget; with the validity test as a branch a third of it was exec-mask bookkeeping)
      // The sample rows yl = min((int)max(y, 0), H - 1), yh = min(yl + 1, H - 1) of a valid sample (y in [-1, H],
      // feature_refine_kernel.cu:72-79) meet the band's rows [r0, r1) <=> r0 - 1 <= max(y, 0) < r1 (r1 = H: <= H), all in
      // float: five vector instructions per source where the integer form took twenty (scan phase 10.9 k -> 9.5 k clocks).  (A NaN y is "valid" in the
      // reference and samples row 0: v_max(NaN, 0) = 0 and the negated compare keep that.)
      // (the CSR form also counts the entries in earlier rows and tests the column: there the integer form below is
      // the shorter one -- 13.2 k against 14.8 k clocks for the phase)
      bool pass;
      if (TAB) {
        // the table's y is the clamped sample row (0 <= y <= H - 1, or H + 1 for a sample outside the map in y OR x):
        // yl = (int)y, yh = min(yl + 1, H - 1)
        // (a NaN stays "valid" and samples row 0, as in the box form and the reference)
        if (!CSR) {
          const float yc = __builtin_fmaxf(yv[u], 0.f);
          pass = yc >= band_lo && yc < band_hi;
        } else {
          // (measured and not kept: the count of entries in earlier rows as two float compares, 2 [yc < r0] + 2 [yc <
          // r0 - 1] -- 14.2-14.6 us against 13.3-13.6 for this integer form)
          const float yc = yv[u];
          const int yl = (int)yc, yh = min(yl + 1, H - 1);
          const bool valid = !(yc > (float)H);
          before += valid ? (yl < r0 ? 2 : 0) + (yh < r0 ? 2 : 0) : 0;
          pass = valid && yl < r1 && yh >= r0;
        }
      } else if (!CSR) {
        const float y = yv[u] * scale;
        const float yc = __builtin_fmaxf(y, 0.f);
        pass = !(y < -1.0f) && yc >= band_lo && yc < band_hi;
      } else {
        float y = yv[u] * scale;
        const float x = xv[u] * scale;  // sic: row <- x_ctr, column <- y_ctr
        const bool valid = !(y < -1.0 || y > H || x < -1.0 || x > W);  // (feature_refine_kernel.cu:72-79)
        y = y <= 0 ? 0.f : y;
        const int yl = min((int)y, H - 1), yh = min(yl + 1, H - 1);  // (= the reference's clamp: yl >= H - 1 -> both H - 1)
        before += valid ? (yl < r0 ? 2 : 0) + (yh < r0 ? 2 : 0) : 0;
        pass = valid && yl < r1 && yh >= r0;
      }
      const u64 m = __ballot(pass);
      if (m == 0ULL) continue;
      if (pass) {
        const int at = wc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (at < IXS_SEG) S.seg[wave * IXS_SEG + at] = s; else over = true;
      }
      wc += __popcll(m);
    }
  }
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) S.part[wave] = before;
  if (over) S.over = 1;
  __syncthreads();
  stamp(1);
  int base = 0;
  for (int w = 0; w < IX_WAVES; w++) base += S.part[w];
  // B: every wave its own reaching sources: the four taps; entries of this band to the LDS list, in (wave, source,
  // tap) order: first the wave's count, then -- behind the waves before it -- the entries
  const int wcc = min(wc, IXS_SEG);
  int wtot = 0;
  // (the taps of a wave's first 64 reaching sources -- usually all of them -- are computed once and kept for the write
  // pass behind the barrier: computed again there they cost a second round trip to the boxes)
  TapYX t_first;
  t_first.valid = false;
  int s_first = 0;
  if (lane < wcc) {
    s_first = S.seg[wave * IXS_SEG + lane];
    t_first = tap_of(s_first);
  }
  for (int i0 = 0; i0 < wcc; i0 += 64) {
    const int i = i0 + lane;
    TapYX t = t_first;
    if (i0 > 0) {
      t.valid = false;
      if (i < wcc) {
        const int s = S.seg[wave * IXS_SEG + i];
        t = tap_of(s);
      }
    }
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const int y = (d >> 1) ? t.yh : t.yl;
      wtot += __popcll(__ballot(t.valid && y >= r0 && y < r1));
    }
  }
  if (lane == 0) S.wcnt[wave] = wtot;
  __syncthreads();
  int at0 = 0, total = 0;
  for (int w = 0; w < IX_WAVES; w++) {
    const int c = S.wcnt[w];
    if (w < wave) at0 += c;
    total += c;
  }
  if (!S.over && total <= IXS_CAP) {
    // A wave's entries in (tap, position in its source list) order -- the tap outermost, so that the order does not
    // depend on where the list's 64-source chunks begin: sources that reach the band with their row but turn out to
    // have no valid sample (the box form's scan does not look at the column; the table form never lists them) take
    // list slots without contributing entries, and the two forms must still build the same index.  (Up to 64 sources
    // -- nearly always -- the taps are the registers of t_first; a longer list computes a chunk's taps once per tap.)
#pragma unroll
    for (int d = 0; d < 4; d++) {
      for (int i0 = 0; i0 < wcc; i0 += 64) {
        const int i = i0 + lane;
        TapYX t = t_first;
        int s = s_first;
        if (i0 > 0) {
          t.valid = false;
          s = 0;
          if (i < wcc) {
            s = S.seg[wave * IXS_SEG + i];
            t = tap_of(s);
          }
        }
        const int y = (d >> 1) ? t.yh : t.yl, x = (d & 1) ? t.xh : t.xl;
        const bool in = t.valid && y >= r0 && y < r1;
        const u64 m = __ballot(in);
        if (in) {
          const int at = at0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          S.sk[at] = ((u64)(unsigned)((y - r0) * W + x) << 56) | ((u64)(unsigned)s << 34) | ((u64)d << 32) |
                     (unsigned)__float_as_int(t.w[d]);
        }
        at0 += __popcll(m);
      }
    }
  }
  __syncthreads();
  stamp(2);
  if (S.over || total > IXS_CAP) {  // more than the LDS list holds: the general form on the same band
    __syncthreads();
    frb_index_general_call(bx, scale, H, W, r0, r1, ix, ci, en);
    if (so.hdr) {  // ... and the band's SELL rows from the lists it just wrote
      __threadfence_block();
      __syncthreads();
      const int s0 = (r0 * W) >> 6, ns = CB >> 6;
      if ((tid >> 6) < ns)
        frb_sell_slice(ci, en, s0 + (tid >> 6), lane, HW, W, so.pitch, H * so.pitch, so.ascale, so.cap,
                       so.hdr + (size_t)n * so.slices,
                       reinterpret_cast<int4*>(so.rows) + (size_t)n * so.slices * (so.cap >> 3) * 256);
    }
    return;
  }
  // C: stable sort by cell
  const int npad = total <= IX_T ? IX_T : total <= 2 * IX_T ? 2 * IX_T : total <= 4 * IX_T ? 4 * IX_T : 8 * IX_T;
  for (int i = total + tid; i < npad; i += IX_T) S.sk[i] = ~0ull;
  __syncthreads();
  if (npad == IX_T) ixs_sort8<1>(S.sk, S.hist, tid);
  else if (npad == 2 * IX_T) ixs_sort8<2>(S.sk, S.hist, tid);
  else if (npad == 4 * IX_T) ixs_sort<4>(S.sk, S.hist, tid);
  else ixs_sort<8>(S.sk, S.hist, tid);
  stamp(3);
  // D: the sorted list is the band's slice of the entry array; run boundaries give {start, len} per cell
  for (int i = tid; i < total; i += IX_T) {
    const u64 k = S.sk[i];
    const int c = (int)(k >> 56);
    if (i == 0 || (int)(S.sk[i - 1] >> 56) != c) S.cst[c] = i;
    if (i == total - 1 || (int)(S.sk[i + 1] >> 56) != c) S.cend[c] = i + 1;
    if (CSR) {
      const int s = (int)((k >> 34) & 0x3fffff);
      const int sy = s / W, sx = s - sy * W, y = r0 + c / W, x = c - (c / W) * W;
      en[base + i] = make_int2(frb_entry_code(sy, sx, y, x, W), (int)(unsigned)k);
    }
  }
  __syncthreads();
  if (CSR)
    for (int c = tid; c < CB; c += IX_T) ci[r0 * W + c] = make_int2(base + S.cst[c], S.cend[c] - S.cst[c]);
  if (!CSR) {  // a list beyond the SELL capacity: the band through the general form (CSR lists + SELL rows from them)
    if (tid < CB && S.cend[tid] - S.cst[tid] > so.cap) S.over = 1;
    __syncthreads();
    if (S.over) {
      frb_index_general_call(bx, scale, H, W, r0, r1, ix, ci, en);
      __threadfence_block();
      __syncthreads();
      const int s0 = (r0 * W) >> 6, ns = CB >> 6;
      if ((tid >> 6) < ns)
        frb_sell_slice(ci, en, s0 + (tid >> 6), lane, HW, W, so.pitch, H * so.pitch, so.ascale, so.cap,
                       so.hdr + (size_t)n * so.slices,
                       reinterpret_cast<int4*>(so.rows) + (size_t)n * so.slices * (so.cap >> 3) * 256);
      return;
    }
  }
  if (so.hdr) {
    // E: the band's SELL-64 rows straight from the sorted list (the launcher sends only bands of whole slices here):
    // entry r of a cell at its (batch, row pair, half); every list padded to its slice's longest (rounded up to even)
    // with {weight 0, zero cell}
    int2* rows = so.rows + (size_t)n * so.slices * (so.cap >> 3) * 512;
    const int s0 = (r0 * W) >> 6;
    for (int i = tid; i < total; i += IX_T) {
      const u64 k = S.sk[i];
      const int c = (int)(k >> 56), r = i - S.cst[c];
      if (r < so.cap) {
        const int s = (int)((k >> 34) & 0x3fffff);
        const int sy = s / W, sx = s - sy * W;
        rows[frb_sell_at(s0 + (c >> 6), c & 63, r, so.cap)] = make_int2((int)(unsigned)k, (sy * so.pitch + sx) * so.ascale);
      }
    }
    if (tid < CB) {  // (CB <= 256: wavefront w = slice s0 + w)
      const int len = S.cend[tid] - S.cst[tid];
      int m = len;
      for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
      const int mp = min(so.cap, (m + 1) & ~1);
      if (lane == 0) so.hdr[(size_t)n * so.slices + s0 + (tid >> 6)] = mp | ((m > so.cap) << 16);
      for (int r = len; r < mp; r++)
        rows[frb_sell_at(s0 + (tid >> 6), lane, r, so.cap)] = make_int2(0, H * so.pitch * so.ascale);
    }
  }
  stamp(4);
#ifdef R3_PROBES
  if (stamps && band == bands / 2 && n == 0 && threadIdx.x == 0) {
    stamps[5] = (u64)total;
    stamps[6] = (u64)wc;
  }
#endif
}

template <bool CSR, bool TAB = false>
__global__ __launch_bounds__(IX_T) void frb_index_sort_kernel(const float* __restrict__ boxes, float scale, int H, int W,
                                                              int R, int2* __restrict__ cellinfo,
                                                              int2* __restrict__ entries, FrbSellOut so,
                                                              u64* __restrict__ stamps = nullptr,
                                                              const float* __restrict__ tab = nullptr) {
  frb_index_sort_body<CSR, TAB>(boxes, scale, H, W, R, cellinfo, entries, so, blockIdx.x, gridDim.x, blockIdx.y, stamps,
                                tab);
}

// The bands of ALL pyramid levels of a FeatureRefineModule pass as one grid (levels in kernel arguments; a block finds
// its level from the block ranges): the coarse levels' index kernels are 9-12 us of latency each on their own, here
// they run beside level 0's bands.
constexpr int FRB_MAX_LEVELS = 8;
struct FrbLevelArgs {
  const float* boxes;
  const float* tab;  // the level's tap table (TAB kernels; null: this level from its boxes)
  int2* cellinfo;
  int2* entries;
  FrbSellOut so;
  float scale;
  int H, W, R, first;  // first: the level's first block
};
struct FrbLevelsArgs {
  FrbLevelArgs l[FRB_MAX_LEVELS];
  int n;
};

template <bool CSR, bool TAB = false>
__global__ __launch_bounds__(IX_T) void frb_index_sort_levels_kernel(const FrbLevelsArgs A) {
  int lv = 0;
#pragma unroll
  for (int i = 1; i < FRB_MAX_LEVELS; i++)
    if (i < A.n && (int)blockIdx.x >= A.l[i].first) lv = i;
  // (a uniform index into the by-value argument block: scalar loads)
  const FrbLevelArgs& L = A.l[lv];
  const int bands = (lv + 1 < A.n ? A.l[lv + 1].first : (int)gridDim.x) - L.first;
  // (TAB: every level has its table -- the launcher takes the box kernel when one is missing: both bodies in one kernel
  // cost the latency-bound workgroups more than the table saves)
  frb_index_sort_body<CSR, TAB>(L.boxes, L.scale, L.H, L.W, L.R, L.cellinfo, L.entries, L.so, (int)blockIdx.x - L.first,
                                bands, blockIdx.y, nullptr, L.tab);
}

// ------------------------------------------------------------------------------------------------
// gather
// ------------------------------------------------------------------------------------------------
// PAIRED (square tile grids): 8 waves = the 4 x 4 cell tile (i, j) and its transpose (j, i), four waves each;
// the diagonal tiles two by two (every workgroup has two full halves).  Tiles are dealt to the XCDs in
// contiguous bands (blockIdx & 7 = XCD under round-robin dispatch).
// All index arithmetic in 32 bits (the launcher refuses IMAGES of 4 GB and more), everything about a cell's list in
// scalar registers: the first version of this kernel spent 640 scalar instructions per wavefront on 64-bit address
// arithmetic and register spills and was bound by their issue (PMC: 10.5 M scalar of 15.5 M instructions).
#ifdef R3_PROBES
__device__ unsigned long long* d_frb_stamps = nullptr;  // (tools/frb_stamps.py: clock stamps of thread 0 of every workgroup)
#endif

template <bool ACCUM, bool PAIRED>
__device__ __forceinline__ void frb_gather_body(const float* __restrict__ top, const int2* __restrict__ cellinfo,
                                                const int2* __restrict__ entries, int C, int H, int W, int EPI,
                                                int tiles_xs, int tiles_per_img, int T, float* __restrict__ bottom,
                                                const unsigned block) {
  const int tiles_x = tiles_xs & 0xfffff, strip = tiles_xs >> 20;  // (the pair walk's strip height rides in the top bits)
  auto stamp = [&](int i) {
#ifdef R3_PROBES
    unsigned long long* st = d_frb_stamps;
    if (st && threadIdx.x == 0) st[(size_t)block * 8 + i] = __builtin_amdgcn_s_memrealtime();
#endif
  };
  stamp(0);
  __shared__ float4 Gs[PAIRED ? 32 : 16][64];  // slot (half * 16 + row of the tile * 4 + column) x lane
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  unsigned t = block;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands of tiles
  const int n = (int)(t / (unsigned)tiles_per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)tiles_per_img);
  int ty, tx;
  bool idle = false;   // (an idle half still takes part in the barriers)
  if (PAIRED) {
    const int off = tiles_x * (tiles_x - 1) / 2;
    if (tt < off) {
      int pi, pj;
      pair_walk(tt, tiles_x, strip, pi, pj);
      ty = half ? pj : pi;
      tx = half ? pi : pj;
    } else {
      ty = tx = 2 * (tt - off) + half;
      idle = ty >= tiles_x;
    }
  } else {
    ty = tt / tiles_x;
    tx = tt - ty * tiles_x;
  }
  const int h = ty * 4 + wave;
  idle = idle || h >= H;
  const int HW = H * W, C4 = C >> 2;
  const int w0 = tx * 4, cnt = idle ? 0 : min(4, W - w0);
  // per-image bases (the only 64-bit arithmetic); below, byte offsets are unsigned 32-bit
  const char* gI = reinterpret_cast<const char*>(top) + (size_t)n * HW * C * 4;
  char* oI = reinterpret_cast<char*>(bottom) + (size_t)n * HW * C * 4;
  const char* ciI = reinterpret_cast<const char*>(cellinfo + (size_t)n * HW);
  const char* enI = reinterpret_cast<const char*>(entries + (size_t)n * EPI);
  const unsigned rowB = (unsigned)C * 4u;       // bytes of a gradient row
  const unsigned q0 = idle ? 0u : (unsigned)(h * W + w0);
  const int own = half * 16 + wave * 4;         // this wave's first slot
  const int halfX = PAIRED ? half << 4 : 0;     // an entry's slot is relative to its own cell's tile
  auto entry = [&](unsigned k) -> int2 { return *reinterpret_cast<const int2*>(enI + k * 8u); };
  for (int c0 = 0; c0 < C4; c0 += 64) {  // (wave-uniform trip count: the barriers are inside)
    const bool cl = c0 + lane < C4;
    const unsigned laneB = (unsigned)(c0 + lane) * 16u;  // this lane's 16 bytes inside a row
    struct E4 { int2 e[4]; };
    auto first4 = [&](const int st_i) -> E4 {  // (entries st .. st + 3 are inside the array, or the slack behind it)
      // (one 32-byte scalar load instead of four 8-byte ones: the entries of a cell are contiguous)
      typedef int frb_i8 __attribute__((ext_vector_type(8), aligned(8)));
      const frb_i8 v = *reinterpret_cast<const frb_i8*>(enI + (unsigned)st_i * 8u);
      E4 r;
      r.e[0] = make_int2(v[0], v[1]); r.e[1] = make_int2(v[2], v[3]);
      r.e[2] = make_int2(v[4], v[5]); r.e[3] = make_int2(v[6], v[7]);
      return r;
    };
    // phase 1: the gradient rows of the wave's own 4 cells and their {start, len}, all in flight together; rows to LDS
    int st[4], len[4];
    E4 en_first;
    {
      float4 gi[4];
#pragma unroll
      for (int i = 0; i < 4; i++) gi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      // The 2 x 2 interior of a tile: its rows are sources of this workgroup's two tiles only, as long as a box samples
      // within one cell of its transposed position -- non-temporal, like the output rows below (the forward kernel
      // gained 7 % from the same hint, r3_fr.hip).  The four loads as ONE straight-line block per kind of wave (the
      // interior positions are a compile-time mask; clamped rows: a short last tile loads a row twice): with a
      // scalar branch per row the compiler put a wait between the loads.
      auto load4 = [&](auto mask_tag) {
        constexpr int M = decltype(mask_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned q = q0 + (unsigned)min(i, max(cnt - 1, 0));
          if ((M >> i) & 1) {
            const frb_v4 t4 = __builtin_nontemporal_load(reinterpret_cast<const frb_v4*>(gI + (q * rowB + laneB)));
            gi[i] = make_float4(t4.x, t4.y, t4.z, t4.w);
          } else {
            gi[i] = *reinterpret_cast<const float4*>(gI + (q * rowB + laneB));
          }
        }
      };
      if (cl && cnt > 0) {
        if (wave == 1 || wave == 2) load4(std::integral_constant<int, 6>{});
        else load4(std::integral_constant<int, 0>{});
      }
      // (the cells' {start, len} behind the row loads: in front of them the wave waited for these scalar loads -- an
      // L2 round trip -- before it requested its rows; the compiler barrier keeps the scheduler from hoisting them back)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = q0 + (unsigned)min(i, max(cnt - 1, 0));
        const int2 ci = *reinterpret_cast<const int2*>(ciI + q * 8u);
        st[i] = i < cnt ? ci.x : 0;
        len[i] = i < cnt ? ci.y : 0;
      }
      // (... and a use right here keeps it from sinking them behind the barrier)
      asm volatile("" ::"s"(st[0]), "s"(st[1]), "s"(st[2]), "s"(st[3]), "s"(len[0]), "s"(len[1]), "s"(len[2]), "s"(len[3]));
      en_first = first4(st[0]);  // (the first cell's first entries too: the third dependent round trip of a wave)
      asm volatile("" ::"s"(en_first.e[0].x), "s"(en_first.e[1].x), "s"(en_first.e[2].x), "s"(en_first.e[3].x));
#pragma unroll
      for (int i = 0; i < 4; i++) Gs[own + i][lane] = gi[i];
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    // an entry's gradient row: from either tile of the workgroup (LDS) or from memory
    auto G = [&](const int code) -> float4 {  // wave-uniform
      if (code < 0 && (PAIRED || !(code & 0x40000000))) return Gs[((code >> 26) & 31) ^ halfX][lane];
      return cl ? *reinterpret_cast<const float4*>(gI + (((unsigned)code & 0x3ffffffu) * rowB + laneB))
                : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto cell = [&](const int i, const int st_i, const int len_i, const E4& ef) {
      const unsigned q = q0 + (unsigned)i;
      float4 acc = Gs[own + i][lane];
      if (ACCUM && cl) {
        const float4 o = *reinterpret_cast<const float4*>(oI + (q * rowB + laneB));
        acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
      }
      int k = 0;
      for (; k + 4 <= len_i; k += 4) {  // four rows requested, then four updates
        int2 e[4];
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = k == 0 ? ef.e[u] : entry((unsigned)(st_i + k + u));
        // (the rows from memory first, all of them, then the rows from LDS: with `r[u] = G(e[u].x)` entry by entry
        // the LDS read of one entry waited for the pending row load of the entry before it -- vmcnt(0) in front of
        // every ds_read -- and the batch's loads ran one after the other)
        bool mem[4];
#pragma unroll
        for (int u = 0; u < 4; u++) mem[u] = !(e[u].x < 0 && (PAIRED || !(e[u].x & 0x40000000)));
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (mem[u])
            r[u] = cl ? *reinterpret_cast<const float4*>(gI + (((unsigned)e[u].x & 0x3ffffffu) * rowB + laneB))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (!mem[u]) r[u] = Gs[((e[u].x >> 26) & 31) ^ halfX][lane];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float w = __int_as_float(e[u].y);
          acc.x += w * r[u].x; acc.y += w * r[u].y; acc.z += w * r[u].z; acc.w += w * r[u].w;
        }
      }
      for (; k < len_i; k++) {
        const int2 e = k == 0 ? ef.e[0] : k == 1 ? ef.e[1] : k == 2 ? ef.e[2] : entry((unsigned)(st_i + k));
        const float4 r = G(e.x);
        const float w = __int_as_float(e.y);
        acc.x += w * r.x; acc.y += w * r.y; acc.z += w * r.z; acc.w += w * r.w;
      }
      if (cl) {
        const frb_v4 t4 = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(t4, reinterpret_cast<frb_v4*>(oI + (q * rowB + laneB)));
      }
    };
    // (the first entries of cell i + 1 are requested before cell i is worked on)
    E4 en_next = en_first;
#pragma unroll 1
    for (int i = 0; i < cnt; i++) {
      const int st_i = i == 0 ? st[0] : i == 1 ? st[1] : i == 2 ? st[2] : st[3];
      const int len_i = i == 0 ? len[0] : i == 1 ? len[1] : i == 2 ? len[2] : len[3];
      const E4 en_cur = en_next;
      en_next = first4(i == 0 ? st[1] : i == 1 ? st[2] : st[3]);
      cell(i, st_i, len_i, en_cur);
      if (i == 0) stamp(3);
    }
    stamp(4);
    if (c0 + 64 < C4) __syncthreads();  // the next channel block overwrites Gs
  }
}

// (Residency, measured with clock stamps and tools/probes/occupancy_probe2.hip: with its 94 scalar registers a
// wavefront is allocated 112 -- 16 beyond the granule of 16 are set aside per wavefront on this stack -- so a SIMD holds
// 7 of them and a compute unit THREE of these 8-wave workgroups, not the four the runtime's occupancy query
// answers; capped at 78 (amdgpu_num_sgpr(80)) four are resident, each lives 10.4 us instead of 8.1 and the launch
// takes the same 31-32 us at N = 4 (17.5 instead of 18.8 at N = 2): the launch is bound by what it moves.  Not capped.)
template <bool ACCUM, bool PAIRED>
__global__ __launch_bounds__(PAIRED ? 512 : 256) void frb_gather_kernel(const float* __restrict__ top,
                                                                        const int2* __restrict__ cellinfo,
                                                                        const int2* __restrict__ entries, int C, int H,
                                                                        int W, int EPI, int tiles_xs, int tiles_per_img,
                                                                        int T, float* __restrict__ bottom) {
  frb_gather_body<ACCUM, PAIRED>(top, cellinfo, entries, C, H, W, EPI, tiles_xs, tiles_per_img, T, bottom, blockIdx.x);
}

// Several levels' gathers (square tile grids: the paired form) as ONE grid: levels in the kernel arguments, a block
// finds its level from block ranges; the body is the per-level kernel's.
constexpr int FRBG_MAX = 8;
struct FrbGatherLevel {
  const float* top;
  const int2* cellinfo;
  const int2* entries;
  float* bottom;
  int H, W, EPI, tiles_xs, tiles_per_img, T, first;
};
struct FrbGatherLevels {
  FrbGatherLevel l[FRBG_MAX];
  int n;
};

template <bool ACCUM>
__global__ __launch_bounds__(512) void frb_gather_levels_kernel(const FrbGatherLevels A, int C) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < FRBG_MAX; i++)
    if (i < A.n && (int)blockIdx.x >= A.l[i].first) k = i;
  const FrbGatherLevel& L = A.l[k];
  // (T | 1: no XCD remap -- a level's blocks are a slice of the grid, not a multiple of 8 from block 0)
  frb_gather_body<ACCUM, true>(L.top, L.cellinfo, L.entries, C, L.H, L.W, L.EPI, L.tiles_xs, L.tiles_per_img, L.T | 1, L.bottom,
                               blockIdx.x - (unsigned)L.first);
}

// ------------------------------------------------------------------------------------------------
// NCHW gather (the reference's layout, feature_refine_kernel.cu:165-230): the same sum over the same index,
// plane by plane.  In NCHW a cell's gradient is one float per plane, and the reference's row <- x_ctr,
// column <- y_ctr swap puts the sources of a ROW of cells in a COLUMN of the plane: the only access pattern that
// reads a plane coalesced is the whole plane, so a workgroup stages whole gradient planes in LDS -- CP channels
// interleaved per cell (8 or 16 bytes: one ds_read serves CP planes and one index entry is amortised over them) --
// and every lane owns a cell: bottom[q] = top[q] + sum over the entries of q.
// The lists come from the inverse tap index re-laid as SELL-64 (frb_sell_kernel): for a slice of 64 consecutive
// cells the k-th entries of all 64 lists are contiguous, padded to the slice's longest list with {zero cell,
// weight 0} -- coalesced loads, no per-lane offsets, no divergence inside a slice -- and the slices of a wavefront
// (K consecutive ones) are ONE packed stream of 4 KB batches, so that the gather's loads run ahead of its sums with
// nothing but a pointer increment.  Lists longer than the SELL capacity keep their tail in the CSR array and the
// owning lane walks it (piles of hundreds of sources on one cell: slow, but exact and in the same fixed order).
// No atomics, no zero-fill, one summation order.
// What shaped it (rocprofv3 PMC on the first working version, 46 us at level 0, N = 4): the SIMDs issued
// instructions 27 % of the time and half of those were scalar -- cursors over (slice, row), clamps, 64-bit address
// products -- with 2.5 k vector instructions per wavefront of which the sums were a third; every load behind a
// branch, and every `current = next` register copy, made the compiler wait for the load right there.
// ------------------------------------------------------------------------------------------------
constexpr int FRN_T = 1024;  // threads of a gather workgroup: 16 wavefronts, K consecutive slices each

struct FrnLayout {
  FrbLayout csr;
  int* slicehdr;  // [N][slices]  padded list length of the slice (even, <= cap) | tail flag << 16
  int4* sell;     // [N][slices][cap / 8 batches][4 row pairs][64 lanes] {weight, cell byte offset} x 2 (+ 1 batch)
  int slices, cap, pitch, K, cp;
  size_t bytes;
};

inline int frn_pitch(int W) { return W | 1; }  // odd: the column walk of a regular field hits 64 different banks
inline int frn_cap(int points) { return points == 1 ? 32 : 96; }
constexpr size_t FRN_LDS_MAX = 160 * 1024;
constexpr size_t FRN_TAB_BYTES = 16 * 128 * 4;  // the wavefronts' batch lists behind the plane

// channels interleaved per staged cell (as many as fit the 160 KB of a compute unit) and slices per wavefront;
// only the (K, CP) pairs the gather is instantiated for
inline bool frn_config(int C, int H, int W, int& K, int& cp) {
  const size_t cells = (size_t)H * frn_pitch(W) + 1;
  const long long HW = (long long)H * W;
  if (C <= 0 || HW > 32 * FRN_T) return false;
  cp = 0;
  for (int c = 4; c >= 1; c >>= 1)
    if (C % c == 0 && cells * 4 * c + FRN_TAB_BYTES <= FRN_LDS_MAX) {
      cp = c;
      break;
    }
  if (cp == 0) return false;
  K = 1;
  while ((long long)K * FRN_T < HW) K *= 2;
  if (cp < 4 && K < 8) K = 8;
  return true;
}

inline FrnLayout frn_layout(void* ws, int N, int C, int H, int W, int points) {
  FrnLayout L;
  L.csr = frb_layout(ws, N, H, W, points);
  L.slices = (H * W + 63) / 64;
  L.cap = frn_cap(points);
  L.pitch = frn_pitch(W);
  L.K = 1, L.cp = 1;
  frn_config(C, H, W, L.K, L.cp);
  const size_t csr = (L.csr.bytes + 255) & ~(size_t)255;
  const size_t hdr = ((size_t)N * L.slices * sizeof(int) + 255) & ~(size_t)255;
  L.slicehdr = reinterpret_cast<int*>(static_cast<char*>(ws) + csr);
  L.sell = reinterpret_cast<int4*>(static_cast<char*>(ws) + csr + hdr);
  // (+ 1 batch: the gather's loads run one batch ahead of its sums)
  L.bytes = csr + hdr + ((size_t)N * L.slices * (L.cap / 8) + 1) * 4096;
  return L;
}

// CSR -> SELL-64 as a launch of its own (planes whose bands the index kernels do not re-lay themselves)
__global__ __launch_bounds__(256) void frb_sell_kernel(const int2* __restrict__ cellinfo, const int2* __restrict__ entries,
                                                       int HW, int W, int P, int zero_cell, int ascale, int EPI, int cap,
                                                       int slices, int* __restrict__ slicehdr, int4* __restrict__ sell) {
  const int lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
  if (slice >= slices) return;
  frb_sell_slice(cellinfo + (size_t)n * HW, entries + (size_t)n * EPI, slice, lane, HW, W, P, zero_cell, ascale, cap,
                 slicehdr + (size_t)n * slices, sell + (size_t)n * slices * (cap >> 3) * 256);
}

template <int CP> struct FrnVec;
template <> struct FrnVec<1> { typedef float type; };
template <> struct FrnVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct FrnVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int CP>
__device__ __forceinline__ float frn_get(const typename FrnVec<CP>::type& v, int c) {
  if constexpr (CP == 1) return v; else return v[c];
}

// A wavefront owns K consecutive slices (cells (wave * K + k) * 64 + lane): every wavefront sees all row phases of
// a periodic field (with slices dealt round-robin the waves of rows 4b + 1, 4b + 2 did all the work of a field
// whose 4 x 4 blocks regress to one centre).  A workgroup owns ONE group of CP channels of one image: the plane
// loads of a second group, requested while the first is gathered, sit in front of every index load in the
// wavefront's in-order memory queue -- the gather then waits for all of them before its first sum (measured: two
// groups per workgroup took exactly twice one).
typedef unsigned int frn_u4 __attribute__((ext_vector_type(4)));

// DEEP: three batches in flight behind the one being summed (four register sets) and the results stored slice by slice
// instead of held for one burst -- the registers of the held results pay for the fourth set.
template <int K, int CP, bool DEEP = false>
__device__ __forceinline__ void frn_gather_body(const float* __restrict__ top, const int* __restrict__ slicehdr,
                                                const int4* __restrict__ sell, const int2* __restrict__ cellinfo,
                                                const int2* __restrict__ entries, int C, int H, int W, int wshift,
                                                int cap, int EPI, int accum, int xcd, float* __restrict__ bottom,
                                                unsigned long long* __restrict__ stamps, const unsigned block,
                                                const unsigned nblocks) {
  typedef typename FrnVec<CP>::type V;
  // (probes builds: clock stamps of wavefront 0 at the phase boundaries, tools/frn_stamps.py; s_memrealtime: 100 MHz,
  // one clock for the chip)
  auto stamp = [&](int i) {
#ifdef R3_PROBES
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
#endif
  };
  stamp(0);
  extern __shared__ __attribute__((aligned(16))) float frn_lds[];  // [(H * P + 1)][CP]; the last cell stays zero
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = H * W, P = W | 1, slices = (HW + 63) >> 6, BPS = cap >> 3;
  const int CG = C / CP;  // channel groups = workgroups per image
  // workgroups of one image on as few XCDs as possible (blockIdx & 7 = XCD under round-robin dispatch): an XCD's L2
  // then holds the index of one or two images instead of streaming all of them
  unsigned bid = block;
  if (xcd && (nblocks & 7) == 0) bid = (bid & 7u) * (nblocks >> 3) + (bid >> 3);
  const int n = (int)bid / CG, g0 = (int)bid - n * CG;
  // LDS index of cell q: a shift for the power-of-two widths of a pyramid; else by the float quotient (q < 2^24:
  // off by at most one)
  const float invW = 1.f / (float)W;
  auto own_cell = [&](int q) -> int {
    if (wshift >= 0) return q + (q >> wshift);  // (P = W + 1)
    int y = (int)((float)q * invW);
    const int r = q - y * W;
    y += r < 0 ? -1 : r >= W ? 1 : 0;
    return q + y * (P - W);
  };
  if (tid < CP) frn_lds[(size_t)H * P * CP + tid] = 0.f;
  const int sl0 = wave * K;  // this wavefront's first slice
  const int kend = min(K, slices - sl0);  // (<= 0: a wavefront beyond the map)
  const size_t pl = ((size_t)n * C + (size_t)g0 * CP) * HW;
  // Every memory operation below goes through a buffer descriptor (scalar base + bounds), a scalar offset and ONE
  // per-lane offset register: no per-access address arithmetic in the vector unit (the first version of this kernel
  // spent two thirds of its 2.1 k vector instructions per wavefront on 64-bit addresses, exec-mask bookkeeping of
  // bounds tests and register copies); a lane beyond the plane reads 0 and its store is dropped by the bounds check.
  __amdgpu_buffer_rsrc_t rtop[CP], rbot[CP];
#pragma unroll
  for (int c = 0; c < CP; c++) {
    rtop[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(top + pl + (size_t)c * HW), 0, HW * 4, 0x00020000);
    rbot[c] = __builtin_amdgcn_make_buffer_rsrc(bottom + pl + (size_t)c * HW, 0, HW * 4, 0x00020000);
  }
  const __amdgpu_buffer_rsrc_t rsell = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int4*>(sell + (size_t)n * slices * BPS * 256), 0, (slices * BPS + 1) * 4096, 0x00020000);
  const int lane4 = lane * 4, lane16 = lane * 16;
  // the slices' padded list lengths, lane i holding slice sl0 + i's (read back with v_readlane: the walk below is
  // all scalar); K <= 32
  int hv = 0;
  if (lane < kend) hv = slicehdr[(size_t)n * slices + sl0 + lane];
  {  // staging: the group's planes, channels interleaved per cell (KC cells per round)
    constexpr int KC = K * CP <= 32 ? (K < 16 ? K : 16) : 16 / CP;
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += KC) {
      float v[KC][CP];
      const int so = (sl0 + k0) * 256;
#pragma unroll
      for (int k = 0; k < KC; k++)
#pragma unroll
        for (int c = 0; c < CP; c++)
          v[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rtop[c], lane4 + k * 256, so, 2));
#pragma unroll
      for (int k = 0; k < KC; k++) {
        V t;
#pragma unroll
        for (int c = 0; c < CP; c++) {
          if constexpr (CP == 1) t = v[k][0]; else t[c] = v[k][c];
        }
        const int q = (sl0 + k0 + k) * 64 + lane;
        if (q < HW) *reinterpret_cast<V*>(frn_lds + (size_t)own_cell(q) * CP) = t;
      }
    }
  }
  stamp(1);
  if constexpr (!DEEP) {  // (DEEP: the barrier comes behind the first index loads, below)
    __syncthreads();
    stamp(2);
    if (kend <= 0) return;
  }
  // The wavefront's batches (four row pairs = eight entries per lane, 4 KB; a slice without entries still has one,
  // never looked at) are listed ONCE, by a scalar walk over the slices, in two registers -- lane t: the t-th batch's
  // offset and what the sums need to know about it -- so that the loop proper is: v_readlane, four loads two batches
  // AHEAD of the sums (unconditional, into register sets that rotate: a `current = next` copy would wait for the loads
  // it copies, and so did every load behind a branch), the sums.  The first version walked (slice, row) cursors on
  // both sides: 50 scalar instructions per batch, and a compute unit issues ONE scalar instruction per cycle for all
  // its wavefronts (PMC: 9.7 M scalar instructions per launch = 16 us of a 44 us kernel).
  struct B8 { frn_u4 p[4]; };
  int seq_so = 0, seq_f = 0;
  // Listing the batches is vector work, once per wavefront (as a scalar walk it was 25 scalar instructions per
  // batch): lane k < kend = slice k, a prefix sum of the batch counts over the lanes, every slice writes its (at most
  // cap / 8) batches at their places of a small LDS table behind the plane, lane t reads entry t back.  More than 64
  // batches (very long lists): window after window.
  const int nbv = lane < kend ? max(1, ((hv & 0xffff) + 7) >> 3) : 0;
  int startv = nbv;  // inclusive prefix over the lanes ...
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int up = __shfl_up(startv, o);
    if (lane >= o) startv += up;
  }
  const int total_all = kend > 0 ? __builtin_amdgcn_readlane(startv, max(kend, 1) - 1) : 0;
  startv -= nbv;  // ... exclusive
  int* seq_tab = reinterpret_cast<int*>(frn_lds + ((size_t)H * P + 1) * CP) + wave * 128;
  auto build = [&](int w0) -> int {  // the batches [w0, w0 + 64) of the wavefront; returns how many there are
    const int mp = hv & 0xffff;
    for (int b = 0; b < BPS; b++) {
      const int idx = startv + b - w0;
      if (b < nbv && idx >= 0 && idx < 64) {
        const int left = mp - 8 * b;
        const int np = left >= 8 ? 4 : (left >> 1);
        seq_tab[idx] = ((sl0 + lane) * BPS + b) * 4096;
        seq_tab[64 + idx] = np | ((b == 0) << 3) | ((b == nbv - 1) << 4) | (lane << 8);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int cnt = min(64, total_all - w0);
    seq_so = lane < cnt ? seq_tab[lane] : 0;
    seq_f = lane < cnt ? seq_tab[64 + lane] : 0;
    __builtin_amdgcn_wave_barrier();
    return cnt;
  };
  auto load_batch = [&](int t, B8& r) {  // (t beyond the list: some batch again)
    const int so = __builtin_amdgcn_readlane(seq_so, t & 63);
    const int np = __builtin_amdgcn_readlane(seq_f, t & 63) & 7;  // (only the batch's real row pairs: 1 KB each)
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (u < np) r.p[u] = __builtin_amdgcn_raw_buffer_load_b128(rsell, lane16 + u * 1024, so, 0);
  };
  // (the dynamic LDS starts at 0 -- no static __shared__ here -- so a byte offset IS the LDS address: without this
  // the compiler adds the base, a `v_add 0`, in front of every ds_read)
  typedef const V __attribute__((address_space(3))) * LdsV;
  V acc;
  auto fma_v = [&](float w, const V& g) {  // (one rounding per term instead of two: inside the backward's tolerance)
    if constexpr (CP == 1) acc = __builtin_fmaf(w, g, acc);
    else acc = __builtin_elementwise_fma(V(w), g, acc);
  };
  auto sum_pairs = [&](const B8& b, auto npairs) {
    constexpr int NP = decltype(npairs)::value;
    constexpr int UB = CP == 4 ? 2 : 4;  // row pairs whose rows are requested from the LDS together (registers)
#pragma unroll
    for (int u0 = 0; u0 < NP; u0 += UB) {
      V g[2 * UB];
#pragma unroll
      for (int u = u0; u < NP && u < u0 + UB; u++) {
        g[2 * (u - u0)] = *reinterpret_cast<LdsV>((uintptr_t)b.p[u].y);
        g[2 * (u - u0) + 1] = *reinterpret_cast<LdsV>((uintptr_t)b.p[u].w);
      }
#pragma unroll
      for (int u = u0; u < NP && u < u0 + UB; u++) {
        // (__uint_as_float: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
        fma_v(__uint_as_float(b.p[u].x), g[2 * (u - u0)]);
        fma_v(__uint_as_float(b.p[u].z), g[2 * (u - u0) + 1]);
      }
    }
  };
  // one batch of the sums; f: its entry of the list
  // The results of a wavefront's slices wait in registers and leave in ONE burst behind the sums (HOLD; up to 32
  // registers): a store between the index loads sits in the same in-order memory queue, and with the chip's write
  // bandwidth saturated (every compute unit is in this phase at the same time) the loads behind it waited -- the
  // gather phase took the index time PLUS the write time.  Issued at the end, the writes drain while the compute
  // unit's next workgroup stages its planes.
  constexpr bool HOLD = !DEEP && K * CP <= 32 && !(K == 8 && CP == 4);
  V res[HOLD ? K : 1];
  const int sl0_256 = sl0 * 256, zero_b = H * P * 4 * CP;
  auto step = [&](const B8& b, const int f) {
    const int np = f & 7;
    const int so = sl0_256 + (f & 0x7fffff00);  // byte offset of the slice in a plane
    if (f & 8) {  // first batch of a slice: the cell's own gradient (the zero cell beyond the map)
      const int q = (so >> 2) + lane;
      acc = *reinterpret_cast<LdsV>((uintptr_t)min(own_cell(q) * (4 * CP), zero_b));
      if (accum) {
#pragma unroll
        for (int ch = 0; ch < CP; ch++) {
          const float o = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbot[ch], lane4, so, 0));
          if constexpr (CP == 1) acc = o + acc; else acc[ch] = o + acc[ch];
        }
      }
    }
    if (np == 4) sum_pairs(b, std::integral_constant<int, 4>{});
    else if (np == 3) sum_pairs(b, std::integral_constant<int, 3>{});
    else if (np == 2) sum_pairs(b, std::integral_constant<int, 2>{});
    else if (np == 1) sum_pairs(b, std::integral_constant<int, 1>{});
    if (!(f & 16)) return;
    // last batch of the slice
    if (__builtin_amdgcn_readlane(hv, f >> 8) >> 16) {  // lists beyond the SELL capacity: the rest from the CSR array
      const int q = (so >> 2) + lane;
      int2 ci = make_int2(0, 0);
      if (q < HW) ci = cellinfo[(size_t)n * HW + q];
      const int2* en = entries + (size_t)n * EPI + ci.x;
      for (int t = cap; t < ci.y; t++) {
        const int2 e = en[t];
        const int s = e.x & 0x3ffffff, sy = s / W, sx = s - sy * W;
        fma_v(__int_as_float(e.y), *reinterpret_cast<const V*>(frn_lds + (size_t)(sy * P + sx) * CP));
      }
    }
    if (HOLD) {  // (a register array under a uniform index: s_set_gpr_idx, no scratch)
      res[HOLD ? (f >> 8) & (K - 1) : 0] = acc;
      return;
    }
#pragma unroll
    for (int ch = 0; ch < CP; ch++) {
      const unsigned bits = __builtin_bit_cast(unsigned, frn_get<CP>(acc, ch));
      __builtin_amdgcn_raw_buffer_store_b32(bits, rbot[ch], lane4, so, 2);
    }
  };
  if constexpr (DEEP) {
    // The index stream is a latency x depth product: with two 4 KB batches in flight per wavefront a compute unit has
    // 128 KB on the way and gets the 42 B / clock the stamps showed; with three, 128 x 128 at N = 4: 44.3 -> 41.7 us,
    // N = 2: 23.2 -> 21.2 us (tools/frn_ab.py; four in flight: no better -- a wavefront has ~19 batches -- and with the
    // results held as well the kernel spills).
    // The first batches are requested BEFORE the barrier behind the staging (they do not depend on the planes): a
    // wavefront waited 2.7 us there for the slowest one and then one more round trip for its first index rows.
    B8 A, B, Cc, D;
    int total0 = 0;
    if (total_all > 0) {
      total0 = build(0);
      load_batch(0, A);
      load_batch(1, B);
      load_batch(2, Cc);
    }
    __syncthreads();
    stamp(2);
    if (kend <= 0) return;
    for (int w0 = 0; w0 < total_all; w0 += 64) {
      int total = total0;
      if (w0 > 0) {
        total = build(w0);
        load_batch(0, A);
        load_batch(1, B);
        load_batch(2, Cc);
      }
#pragma unroll 1
      for (int t = 0; t < total; t += 4) {
        load_batch(t + 3, D);
        step(A, __builtin_amdgcn_readlane(seq_f, t));
        load_batch(t + 4, A);
        if (t + 1 < total) step(B, __builtin_amdgcn_readlane(seq_f, (t + 1) & 63));
        load_batch(t + 5, B);
        if (t + 2 < total) step(Cc, __builtin_amdgcn_readlane(seq_f, (t + 2) & 63));
        load_batch(t + 6, Cc);
        if (t + 3 < total) step(D, __builtin_amdgcn_readlane(seq_f, (t + 3) & 63));
      }
    }
  } else
  for (int w0 = 0; w0 < total_all; w0 += 64) {  // (one window unless lists are very long)
    const int total = build(w0);
    B8 A, B, D;
    load_batch(0, A);
    load_batch(1, B);
#pragma unroll 1
    for (int t = 0; t < total; t += 3) {  // (three sets: two batches in flight behind the one being summed)
      load_batch(t + 2, D);
      step(A, __builtin_amdgcn_readlane(seq_f, t));
      load_batch(t + 3, A);
      if (t + 1 < total) step(B, __builtin_amdgcn_readlane(seq_f, (t + 1) & 63));
      load_batch(t + 4, B);
      if (t + 2 < total) step(D, __builtin_amdgcn_readlane(seq_f, (t + 2) & 63));
    }
  }
  if (HOLD) {  // (a slice beyond the map: its offset lies beyond the plane's descriptor, the store is dropped)
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int ch = 0; ch < CP; ch++)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, frn_get<CP>(res[HOLD ? k : 0], ch)), rbot[ch],
                                              lane4, sl0_256 + k * 256, 2);
  }
  stamp(3);
#ifdef R3_PROBES
  if (stamps) {
    __builtin_amdgcn_s_waitcnt(0);
    stamp(4);
    __syncthreads();
    stamp(5);
  }
#endif
}

template <int K, int CP>
__global__ __launch_bounds__(FRN_T) void frn_gather_kernel(const float* __restrict__ top, const int* __restrict__ slicehdr,
                                                           const int4* __restrict__ sell,
                                                           const int2* __restrict__ cellinfo,
                                                           const int2* __restrict__ entries, int C, int H, int W,
                                                           int wshift, int cap, int EPI, int accum, int xcd,
                                                           float* __restrict__ bottom,
                                                           unsigned long long* __restrict__ stamps) {
  // (the deep form where it was measured: whole 128 x 128 planes of two channels)
  frn_gather_body<K, CP, K == 16 && CP == 2>(top, slicehdr, sell, cellinfo, entries, C, H, W, wshift, cap, EPI, accum, xcd,
                                             bottom, stamps, blockIdx.x, gridDim.x);
}

// The coarse levels of a pyramid (planes of at most 4096 cells: one or four slices per wavefront, four channels per
// workgroup) as ONE grid: each is a 5-9 us launch of latency on its own.  Levels in the kernel arguments, a block finds its level
// from block ranges; the body is the per-level kernel's.
constexpr int FRNL_MAX = 8;
struct FrnLevel {
  const float* top;
  const int* slicehdr;
  const int4* sell;
  const int2* cellinfo;
  const int2* entries;
  float* bottom;
  int H, W, wshift, cap, EPI, first, K;
};
struct FrnLevels {
  FrnLevel l[FRNL_MAX];
  int n;
};

__global__ __launch_bounds__(FRN_T) void frn_gather_levels_kernel(const FrnLevels A, int C, int accum) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < FRNL_MAX; i++)
    if (i < A.n && (int)blockIdx.x >= A.l[i].first) k = i;
  const FrnLevel& L = A.l[k];
  if (L.K == 4)
    frn_gather_body<4, 4>(L.top, L.slicehdr, L.sell, L.cellinfo, L.entries, C, L.H, L.W, L.wshift, L.cap, L.EPI, accum, 0,
                          L.bottom, nullptr, blockIdx.x - (unsigned)L.first, 0u);
  else
    frn_gather_body<1, 4>(L.top, L.slicehdr, L.sell, L.cellinfo, L.entries, C, L.H, L.W, L.wshift, L.cap, L.EPI, accum, 0,
                          L.bottom, nullptr, blockIdx.x - (unsigned)L.first, 0u);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K>
inline void allow_big_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

R3Option g_r3_frb_impl{0};
R3Option64 g_r3_frn_stamps{0};  // device address of a stamp buffer (probes), 0 = none
  // 0 auto; 1: the general index form whatever the shape; 2: unpaired gather (A/B runs, tests)

size_t r3k_frb_workspace_bytes(int N, int H, int W, int points) {
  if (N <= 0 || H <= 0 || W <= 0 || (points != 1 && points != 5) || W > IX_MAXCELLS || (long long)H * W > (1LL << 26)) return 0;
  return frb_layout(nullptr, N, H, W, points).bytes;
}

// the inverse tap index of a level (depends on the boxes only)
namespace {
// the sort form re-lays a band's lists as SELL rows itself when its bands are whole slices
inline bool frb_sort_form(int H, int W, int points) {
  return points == 1 && W <= IXS_CELLS && (long long)H * W <= (1LL << 22) && g_r3_frb_impl != 1;
}
inline bool frb_fuses_sell(int H, int W, int points) {
  if (!frb_sort_form(H, W, points) || g_r3_frb_impl == 3) return false;  // (frb_impl 3: the SELL launch always; tests)
  const int R = sort_band_rows(H, W);
  return (R * W) % 64 == 0 && R * W <= 256 && (H * W) % 64 == 0;  // (a last, shorter band is whole slices too)
}
int frb_index_launch(const float* boxes, int N, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                     const FrbSellOut& so, hipStream_t stream, const float* tab);
}  // namespace

// tab: the level's tap table (r3k_fr_table_any_bytes: [N][y: HW][x: HW], written by the forward pass of the SAME boxes
// and scale) or null; the sort form then scans the table instead of the box records (the general form reads the boxes)
int r3k_frb_index(const float* boxes, int N, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                  hipStream_t stream, const float* tab) {
  FrbSellOut none;
  none.hdr = nullptr, none.rows = nullptr, none.ascale = none.cap = none.pitch = none.slices = 0;
  return frb_index_launch(boxes, N, H, W, scale, points, ws, ws_bytes, none, stream, tab);
}

namespace {
int frb_index_launch(const float* boxes, int N, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                     const FrbSellOut& so, hipStream_t stream, const float* tab) {
  const size_t need = r3k_frb_workspace_bytes(N, H, W, points);
  if (need == 0 || !boxes || !ws || !aligned16(ws) || (tab && !aligned16(tab))) return -1;
  if (ws_bytes < need) return -3;
  const FrbLayout L = frb_layout(ws, N, H, W, points);
  if (frb_sort_form(H, W, points)) {
    static R3DeviceOnce once;
    if (once.first()) {
      allow_big_lds(frb_index_sort_kernel<true>, (int)sizeof(IxsLds));
      allow_big_lds(frb_index_sort_kernel<false>, (int)sizeof(IxsLds));
      allow_big_lds((frb_index_sort_kernel<true, true>), (int)sizeof(IxsLds));
      allow_big_lds((frb_index_sort_kernel<false, true>), (int)sizeof(IxsLds));
    }
    const int R = sort_band_rows(H, W);
    const dim3 grid((H + R - 1) / R, N);
    // (frb_impl 5: SELL rows AND the CSR lists from the one launch; tests)
    const bool sell = so.hdr && g_r3_frb_impl != 5;
#define R3_IX(CSRF, TABF) \
  hipLaunchKernelGGL((frb_index_sort_kernel<CSRF, TABF>), grid, dim3(IX_T), sizeof(IxsLds), stream, boxes, scale, H, W, R, \
                     L.cellinfo, L.entries, so, (u64*)nullptr, tab)
    if (sell) { if (tab) R3_IX(false, true); else R3_IX(false, false); }
    else { if (tab) R3_IX(true, true); else R3_IX(true, false); }
#undef R3_IX
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  const int R = general_band_rows(H, W);
  const dim3 grid((H + R - 1) / R, N);
  const size_t lds = ((size_t)2 * R * W + IX_WAVES + 8) * sizeof(int);
  if (points == 1)
    hipLaunchKernelGGL(frb_index_general_kernel<1>, grid, dim3(IX_T), lds, stream, boxes, scale, H, W, R, L.cellinfo,
                       L.entries);
  else
    hipLaunchKernelGGL(frb_index_general_kernel<5>, grid, dim3(IX_T), lds, stream, boxes, scale, H, W, R, L.cellinfo,
                       L.entries);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}  // namespace

// bottom_grad (N, H, W, C) = [bottom_grad +] backward(top_grad); index_ready: ws holds r3k_frb_index of these boxes
int r3k_frb_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W, float scale, int points,
                     float* bottom_grad, int overwrite, void* ws, size_t ws_bytes, int index_ready,
                     hipStream_t stream) {
  if (!top_grad || !bottom_grad || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (C & 3) || !aligned16(top_grad) ||
      !aligned16(bottom_grad))
    return -1;
  // (32-bit byte offsets inside an IMAGE, 64-bit image bases; tested before anything is launched: -1 = nothing ran)
  if ((unsigned long long)H * W * C * 4ull >= (1ull << 32)) return -1;
  if (!index_ready) {
    const int rc = r3k_frb_index(boxes, N, H, W, scale, points, ws, ws_bytes, stream, nullptr);
    if (rc) return rc;
  } else if (!ws || !aligned16(ws) || !r3k_frb_workspace_bytes(N, H, W, points) ||
             ws_bytes < r3k_frb_workspace_bytes(N, H, W, points)) {
    return -1;
  }
  const FrbLayout L = frb_layout(ws, N, H, W, points);
  const int tiles_x = (W + 3) / 4, tiles_y = (H + 3) / 4;
  const bool paired = tiles_x == tiles_y && g_r3_frb_impl != 2;
  const int tpi = paired ? tiles_x * (tiles_x - 1) / 2 + (tiles_x + 1) / 2 : tiles_x * tiles_y;
  const long long T = (long long)tpi * N;
  if (T > 0x7fffffffLL) return -1;
  const int EPI = H * W * 4 * points;
  const dim3 grid((unsigned)T), block(paired ? 512 : 256);
#define R3_ARGS top_grad, L.cellinfo, L.entries, C, H, W, EPI, tiles_x | (g_r3_fr_walk << 20), tpi, (int)T, bottom_grad
#ifdef R3_PROBES
  {  // (clock stamps, tools/frb_stamps.py: the buffer set through frn_stamps_lo / _hi)
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(g_r3_frn_stamps.get());
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_frb_stamps), &sp, sizeof(sp), 0, hipMemcpyHostToDevice, stream);
  }
#endif
  if (paired) {
    if (overwrite) hipLaunchKernelGGL((frb_gather_kernel<false, true>), grid, block, 0, stream, R3_ARGS);
    else hipLaunchKernelGGL((frb_gather_kernel<true, true>), grid, block, 0, stream, R3_ARGS);
  } else {
    if (overwrite) hipLaunchKernelGGL((frb_gather_kernel<false, false>), grid, block, 0, stream, R3_ARGS);
    else hipLaunchKernelGGL((frb_gather_kernel<true, false>), grid, block, 0, stream, R3_ARGS);
  }
#undef R3_ARGS
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The channels_last gathers of several levels over their indexes (ws[l]: r3k_frb_index / r3k_frb_index_levels): the
// levels of at most 4096 cells with square tile grids as ONE grid, the others one launch each.  Pointer arrays are HOST
// arrays.
int r3k_frb_gather_levels(int levels, const float* const* top_grad, int N, int C, const int* H, const int* W, int points,
                          float* const* bottom_grad, int overwrite, void* const* ws, const size_t* ws_bytes,
                          hipStream_t stream) {
  FrbGatherLevels A;
  A.n = 0;
  int blocks = 0;
  bool grouped[FRBG_MAX] = {};
  for (int l = 0; l < levels && levels <= FRBG_MAX && g_r3_frb_impl != 6 && g_r3_frb_impl != 2; l++) {
    const size_t need = r3k_frb_workspace_bytes(N, H[l], W[l], points);
    const int tiles_x = (W[l] + 3) / 4, tiles_y = (H[l] + 3) / 4;
    if (need == 0 || !top_grad[l] || !bottom_grad[l] || !ws[l] || !aligned16(ws[l]) || ws_bytes[l] < need ||
        N <= 0 || C <= 0 || (C & 3) || !aligned16(top_grad[l]) || !aligned16(bottom_grad[l]) || tiles_x != tiles_y ||
        (long long)H[l] * W[l] > 4096 || (unsigned long long)H[l] * W[l] * C * 4ull >= (1ull << 32))
      continue;
    const FrbLayout L = frb_layout(ws[l], N, H[l], W[l], points);
    FrbGatherLevel& a = A.l[A.n++];
    a.top = top_grad[l], a.cellinfo = L.cellinfo, a.entries = L.entries, a.bottom = bottom_grad[l];
    a.H = H[l], a.W = W[l], a.EPI = H[l] * W[l] * 4 * points, a.tiles_xs = tiles_x | (g_r3_fr_walk << 20);
    a.tiles_per_img = tiles_x * (tiles_x - 1) / 2 + (tiles_x + 1) / 2;
    a.T = a.tiles_per_img * N, a.first = blocks;
    blocks += a.T;
    grouped[l] = true;
  }
  if (A.n < 2) {
    A.n = 0;
    for (int l = 0; l < FRBG_MAX; l++) grouped[l] = false;
  }
  for (int l = 0; l < levels; l++) {
    if (l < FRBG_MAX && grouped[l]) continue;
    const int k = r3k_frb_backward(top_grad[l], nullptr, N, C, H[l], W[l], 0.f, points, bottom_grad[l], overwrite, ws[l],
                                   ws_bytes[l], 1, stream);
    if (k) return k;
  }
  if (A.n) {
    for (int i = A.n; i < FRBG_MAX; i++) A.l[i] = A.l[A.n - 1];
    if (overwrite) hipLaunchKernelGGL(frb_gather_levels_kernel<false>, dim3(blocks), dim3(512), 0, stream, A, C);
    else hipLaunchKernelGGL(frb_gather_levels_kernel<true>, dim3(blocks), dim3(512), 0, stream, A, C);
    if (hipGetLastError() != hipSuccess) return -2;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// NCHW entry points: index (CSR + SELL) and gather
// ------------------------------------------------------------------------------------------------
namespace {

template <int K, int CP>
inline int frn_launch(const float* top, const FrnLayout& L, int N, int C, int H, int W, int points, int accum, float* bottom,
                      hipStream_t stream) {
  const size_t lds = ((size_t)H * L.pitch + 1) * 4 * CP + FRN_TAB_BYTES;
  static R3DeviceOnce once;  // (one per instantiation)
  if (once.first()) allow_big_lds(frn_gather_kernel<K, CP>, (int)FRN_LDS_MAX);
  int wshift = -1;
  if ((W & (W - 1)) == 0)
    for (wshift = 0; (1 << wshift) < W; wshift++) {}
  hipLaunchKernelGGL((frn_gather_kernel<K, CP>), dim3((unsigned)(N * (C / CP))), dim3(FRN_T), lds, stream, top, L.slicehdr,
                     L.sell, L.csr.cellinfo, L.csr.entries, C, H, W, wshift, L.cap, H * W * 4 * points, accum,
                     (R3_HAS_PROBES && g_r3_frb_impl == 4) ? 0 : 1, bottom,  // (probes: frb_impl 4 = no XCD remap)
                     R3_HAS_PROBES ? reinterpret_cast<unsigned long long*>(g_r3_frn_stamps.get()) : nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// 0: this shape has no NCHW gather form (for any C)
size_t r3k_frn_workspace_bytes(int N, int H, int W, int points) {
  int K, cp;
  if (r3k_frb_workspace_bytes(N, H, W, points) == 0 || !frn_config(1, H, W, K, cp)) return 0;
  return frn_layout(nullptr, N, 1, H, W, points).bytes;
}

// the index of the boxes for gradients of C channels (C decides how many channels the gather interleaves, and the
// entries carry byte offsets of that layout)
int r3k_frn_index(const float* boxes, int N, int C, int H, int W, float scale, int points, void* ws, size_t ws_bytes,
                  hipStream_t stream, const float* tab) {
  const size_t need = r3k_frn_workspace_bytes(N, H, W, points);
  int K, cp;
  if (need == 0 || !boxes || !ws || !aligned16(ws) || !frn_config(C, H, W, K, cp)) return -1;
  if (ws_bytes < need) return -3;
  const FrnLayout L = frn_layout(ws, N, C, H, W, points);
  FrbSellOut so;
  so.hdr = L.slicehdr, so.rows = reinterpret_cast<int2*>(L.sell), so.ascale = 4 * L.cp, so.cap = L.cap, so.pitch = L.pitch,
  so.slices = L.slices;
  const bool fused = frb_fuses_sell(H, W, points);
  if (!fused) so.hdr = nullptr;
  const int rc = frb_index_launch(boxes, N, H, W, scale, points, ws, L.csr.bytes, so, stream, tab);
  if (rc || fused) return rc;
  hipLaunchKernelGGL(frb_sell_kernel, dim3((L.slices + 3) / 4, N), dim3(256), 0, stream, L.csr.cellinfo, L.csr.entries,
                     H * W, W, L.pitch, H * L.pitch, 4 * L.cp, H * W * 4 * points, L.cap, L.slices, L.slicehdr, L.sell);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The indexes of several levels (each with its own workspace, as r3k_frn_index would fill it) from ONE launch; 1 when
// the levels do not all take the fused sort form (the caller then indexes level by level), else 0 / -2.
int r3k_frn_index_levels(int levels, const float* const* boxes, int N, int C, const int* H, const int* W,
                         const float* scales, int points, void* const* ws, const size_t* ws_bytes, hipStream_t stream,
                         const float* const* tabs) {
  if (levels < 1 || levels > FRB_MAX_LEVELS || g_r3_frb_impl == 5 || g_r3_frb_impl == 6) return 1;  // (6: per level; tests)
  FrbLevelsArgs A;
  A.n = levels;
  int blocks = 0;
  bool all_tab = true;  // (the table kernel only when every level brings its table)
  for (int l = 0; l < levels; l++) {
    int K, cp;
    const size_t need = r3k_frn_workspace_bytes(N, H[l], W[l], points);
    if (need == 0 || !boxes[l] || !ws[l] || !aligned16(ws[l]) || ws_bytes[l] < need || !frn_config(C, H[l], W[l], K, cp) ||
        !frb_fuses_sell(H[l], W[l], points))
      return 1;
    const FrnLayout L = frn_layout(ws[l], N, C, H[l], W[l], points);
    FrbLevelArgs& a = A.l[l];
    a.boxes = boxes[l], a.cellinfo = L.csr.cellinfo, a.entries = L.csr.entries;
    a.tab = tabs ? tabs[l] : nullptr;
    if (a.tab && !aligned16(a.tab)) return -1;
    all_tab &= a.tab != nullptr;
    a.so.hdr = L.slicehdr, a.so.rows = reinterpret_cast<int2*>(L.sell), a.so.ascale = 4 * L.cp, a.so.cap = L.cap,
    a.so.pitch = L.pitch, a.so.slices = L.slices;
    a.scale = scales[l], a.H = H[l], a.W = W[l], a.R = sort_band_rows(H[l], W[l]), a.first = blocks;
    blocks += (H[l] + a.R - 1) / a.R;
  }
  for (int l = levels; l < FRB_MAX_LEVELS; l++) A.l[l] = A.l[levels - 1];
  static R3DeviceOnce once;
  if (once.first()) {
    allow_big_lds(frb_index_sort_levels_kernel<false>, (int)sizeof(IxsLds));
    allow_big_lds((frb_index_sort_levels_kernel<false, true>), (int)sizeof(IxsLds));
  }
  if (all_tab)
    hipLaunchKernelGGL((frb_index_sort_levels_kernel<false, true>), dim3(blocks, N), dim3(IX_T), sizeof(IxsLds), stream, A);
  else
    hipLaunchKernelGGL(frb_index_sort_levels_kernel<false>, dim3(blocks, N), dim3(IX_T), sizeof(IxsLds), stream, A);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The same for the channels_last gather's CSR lists (r3k_frb_index per level): ws[l] as r3k_frb_workspace_bytes sizes it.
int r3k_frb_index_levels(int levels, const float* const* boxes, int N, const int* H, const int* W, const float* scales,
                         int points, void* const* ws, const size_t* ws_bytes, hipStream_t stream,
                         const float* const* tabs) {
  if (levels < 1 || levels > FRB_MAX_LEVELS || g_r3_frb_impl == 6) return 1;
  FrbLevelsArgs A;
  A.n = levels;
  int blocks = 0;
  bool all_tab = true;  // (the table kernel only when every level brings its table)
  for (int l = 0; l < levels; l++) {
    const size_t need = r3k_frb_workspace_bytes(N, H[l], W[l], points);
    if (need == 0 || !boxes[l] || !ws[l] || !aligned16(ws[l]) || ws_bytes[l] < need || !frb_sort_form(H[l], W[l], points))
      return 1;
    const FrbLayout L = frb_layout(ws[l], N, H[l], W[l], points);
    FrbLevelArgs& a = A.l[l];
    a.boxes = boxes[l], a.cellinfo = L.cellinfo, a.entries = L.entries;
    a.tab = tabs ? tabs[l] : nullptr;
    if (a.tab && !aligned16(a.tab)) return -1;
    all_tab &= a.tab != nullptr;
    a.so.hdr = nullptr, a.so.rows = nullptr, a.so.ascale = a.so.cap = a.so.pitch = a.so.slices = 0;
    a.scale = scales[l], a.H = H[l], a.W = W[l], a.R = sort_band_rows(H[l], W[l]), a.first = blocks;
    blocks += (H[l] + a.R - 1) / a.R;
  }
  for (int l = levels; l < FRB_MAX_LEVELS; l++) A.l[l] = A.l[levels - 1];
  static R3DeviceOnce once;
  if (once.first()) {
    allow_big_lds(frb_index_sort_levels_kernel<true>, (int)sizeof(IxsLds));
    allow_big_lds((frb_index_sort_levels_kernel<true, true>), (int)sizeof(IxsLds));
  }
  if (all_tab)
    hipLaunchKernelGGL((frb_index_sort_levels_kernel<true, true>), dim3(blocks, N), dim3(IX_T), sizeof(IxsLds), stream, A);
  else
    hipLaunchKernelGGL(frb_index_sort_levels_kernel<true>, dim3(blocks, N), dim3(IX_T), sizeof(IxsLds), stream, A);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The gathers of several levels over their indexes (ws[l]: r3k_frn_index / r3k_frn_index_levels for this C): the
// levels of at most 4096 cells as ONE grid, the others one launch each.  taken[l] = 0 for a level without a gather
// form (the caller's scatter kernels), else 1.  Pointer arrays are HOST arrays.
int r3k_frn_gather_levels(int levels, const float* const* top_grad, int N, int C, const int* H, const int* W, int points,
                          float* const* bottom_grad, int overwrite, void* const* ws, const size_t* ws_bytes, int* taken,
                          hipStream_t stream) {
  FrnLevels A;
  A.n = 0;
  int blocks = 0;
  size_t lds = 0;
  bool grouped[FRNL_MAX] = {};
  for (int l = 0; l < levels && levels <= FRNL_MAX && g_r3_frb_impl != 6; l++) {
    int K, cp;
    const size_t need = r3k_frn_workspace_bytes(N, H[l], W[l], points);
    if (need == 0 || !top_grad[l] || !bottom_grad[l] || !ws[l] || !aligned16(ws[l]) || ws_bytes[l] < need ||
        !frn_config(C, H[l], W[l], K, cp) || (K != 1 && K != 4) || cp != 4)
      continue;
    const FrnLayout L = frn_layout(ws[l], N, C, H[l], W[l], points);
    FrnLevel& a = A.l[A.n++];
    a.top = top_grad[l], a.slicehdr = L.slicehdr, a.sell = L.sell, a.cellinfo = L.csr.cellinfo, a.entries = L.csr.entries;
    a.bottom = bottom_grad[l], a.H = H[l], a.W = W[l], a.cap = L.cap, a.EPI = H[l] * W[l] * 4 * points, a.first = blocks;
    a.K = K, a.wshift = -1;
    if ((W[l] & (W[l] - 1)) == 0)
      for (a.wshift = 0; (1 << a.wshift) < W[l]; a.wshift++) {}
    blocks += N * (C / 4);
    lds = std::max(lds, ((size_t)H[l] * L.pitch + 1) * 4 * 4 + FRN_TAB_BYTES);
    grouped[l] = true;
  }
  if (A.n < 2) {
    A.n = 0;
    for (int l = 0; l < FRNL_MAX; l++) grouped[l] = false;
  }
  for (int l = 0; l < levels; l++) {
    taken[l] = 1;
    if (l < FRNL_MAX && grouped[l]) continue;
    const int k = ws[l] ? r3k_frn_gather(top_grad[l], N, C, H[l], W[l], points, bottom_grad[l], overwrite, ws[l], ws_bytes[l],
                                         stream)
                        : -1;
    if (k == -1) taken[l] = 0;
    else if (k) return k;
  }
  if (A.n) {
    for (int i = A.n; i < FRNL_MAX; i++) A.l[i] = A.l[A.n - 1];
    static R3DeviceOnce once;
    if (once.first()) allow_big_lds(frn_gather_levels_kernel, (int)FRN_LDS_MAX);
    hipLaunchKernelGGL(frn_gather_levels_kernel, dim3(blocks), dim3(FRN_T), lds, stream, A, C, overwrite ? 0 : 1);
    if (hipGetLastError() != hipSuccess) return -2;
  }
  return 0;
}

// bottom_grad (N, C, H, W) = [bottom_grad +] backward(top_grad) over the index in ws (r3k_frn_index for this C)
int r3k_frn_gather(const float* top_grad, int N, int C, int H, int W, int points, float* bottom_grad, int overwrite,
                   void* ws, size_t ws_bytes, hipStream_t stream) {
  const size_t need = r3k_frn_workspace_bytes(N, H, W, points);
  int K, cp;
  if (need == 0 || !top_grad || !bottom_grad || !ws || !aligned16(ws) || !frn_config(C, H, W, K, cp)) return -1;
  if (ws_bytes < need) return -3;
  const FrnLayout L = frn_layout(ws, N, C, H, W, points);
  const int accum = overwrite ? 0 : 1;
#define R3_FRN(KK, CC) \
  if (K == KK && cp == CC) return frn_launch<KK, CC>(top_grad, L, N, C, H, W, points, accum, bottom_grad, stream)
  R3_FRN(1, 4); R3_FRN(2, 4); R3_FRN(4, 4); R3_FRN(8, 4); R3_FRN(16, 4);
  R3_FRN(8, 2); R3_FRN(16, 2); R3_FRN(32, 2);
  R3_FRN(8, 1); R3_FRN(16, 1); R3_FRN(32, 1);
#undef R3_FRN
  return -1;
}
