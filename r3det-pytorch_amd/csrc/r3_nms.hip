// r3_nms.hip -- rotated NMS on gfx950: wavefront-bitmask suppression over score-sorted
// boxes with the greedy reduction done on the device.
//
// Replaces nmsr_kernel + host scan (rnms/src/rcuda/rnms_kernel.cu:229-335),
// nms_rotated_cuda_kernel + host scan (nms_rotated/src/nms_rotated_cuda.cu:13-134) and the
// ml variant (ml_nms_rotated/src/nms_rotated_cuda.cu:14-137).  Pipeline (all on one stream,
// nothing but the keep count ever crosses PCIe):
//
//   prepare : gather boxes through `order`, one prepared record per box (trig per box);
//   stream  : every 64x64 tile with col_block >= row_block (the reference computes both
//             triangles, nms_rotated_cuda.cu:23) runs only the conservative disjointness
//             test (labels, circles, axis-aligned bounds); surviving pairs go to an LDS
//             queue and from there to one global queue;
//   drain   : the global queue is clipped one pair per lane, chip-wide balanced; IoU > thr
//             appends i to the list of SUPPRESSORS of j (elist[j], up to 32 entries; beyond that
//             bit i of row j of a transposed overflow mask) (mark_pair);
//   reduce  : one workgroup per problem resolves the greedy scan by DEPENDENCY ROUNDS over all
//             rows at once (nms_reduce_rounds_kernel): a row is removed as soon as one of its
//             suppressors is known kept, kept as soon as all of them are known removed; kept /
//             removed bit sets live in LDS, every thread gathers the few listed words of its rows.
//             Suppression chains are short on detection pools (2-5 rounds); after 32 rounds a
//             sequential in-order finish takes over (adversarial chains).  Emits the keep list and
//             its length.  The reference copies the whole n x n/64 mask to the host (9.2 MB at
//             n = 8576) and scans it there; round 1 of this build walked 64-row blocks in score
//             order (two barriers per block: 36 us per 4 x 3.3 k boxes, 260 us at n = 8576).
//   ascending (v1 only): rnms returns keep sorted by index (rnms_kernel.cu:331-334).
// The batched detection pipeline at the end of the file (r3det_mcnms_*) runs select / sort /
// offsets / these kernels with blockIdx.z = image / finish for all images of a step.
//
// The original tile kernel (mask computed in place, dense reduction) is kept as impl 1: it
// serves thr < 0, n >= 65536 and the A/B measurements.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "r3_clip.h"
#include "r3_geom_lds.h"
#include "r3_kernels.h"

namespace {

typedef unsigned long long u64;

constexpr int TILE = 64;        // one wavefront = one 64-wide bitmask word
constexpr int MASK_WAVES = 4;   // tiles per workgroup of the tile kernels
constexpr int NT = TILE * MASK_WAVES;
// The global pair queue is split into Q_NREG regions, each with its own fill counter 128 B from the next: device-
// scope atomics execute memory-side, one address takes ~12 ns per operation, and the stream kernel issues one per
// tile (9 045 tiles at n = 8576 were 108 us on ONE counter).  Control block of one problem: the region counters,
// then the redo-list counter.
constexpr int Q_NREG = 64;
constexpr int Q_CSTRIDE = 32;
constexpr int Q_REDO = Q_NREG * Q_CSTRIDE;   // word index of the redo-list counter
constexpr int Q_XFLAG = Q_REDO + 1;            // set by the drain when a suppressor edge joins two label groups (batched runs)
constexpr int RG_GROUPS = 16;                  // label groups of the batched reducer: group = label mod 16
constexpr int Q_CTL_WORDS = Q_REDO + Q_CSTRIDE;

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }


// Batched launches (blockIdx.z = image of a detection batch): every per-image array has the same
// capacity, `counts[img]` is that image's box count and the strides (in elements) separate the
// images.  counts == nullptr: one problem, n is the kernel argument (the plain operators).
struct Batch {
  const int* counts;
  size_t recs, mask, nz, counter, queue, keep;
  size_t redo;  // stride of the redo-tile lists
  size_t rows;  // row capacity of one image's arrays (= n for a single problem); `nz` = stride of the side tables
  const uint8_t* rlab;  // batched runs: label of every sorted row (stride `rows`), or nullptr
  u64* gbits;           // batched runs: per image RG_GROUPS bitmaps of cb words -- bit p of bitmap g: sorted row p has label
                        // group g -- or nullptr.  Written for the reducer: by the stream kernel's diagonal tiles (a
                        // word per 64 rows and group, plain stores) or, in the sorted-chunk form, by the ranking
                        // kernel (atomicOr into words the chunk-sort kernel zeroed)
};

inline Batch single_problem(int n) {
  Batch b{};
  b.rows = (size_t)n;
  return b;
}

// In-edge lists of one image (in the same allocation as, and zeroed with, the overflow mask):
//   ecnt[r]  : number of boxes i < r (score order) with IoU(i, r) > thr -- the suppressors of r
//   elist[r] : the first EL of them (u16 sorted positions, arrival order; 64 bytes per row)
// Suppressors beyond EL (very dense clusters) go to row r of the TRANSPOSED mask instead (bit i of
// maskT[r][i / 64]), which the reducer scans only for rows with ecnt > EL.
constexpr int EL = 32;

//   msup[r]  : 65535 - (smallest suppressor index of r), 0 = none.  The highest-scored suppressor of a box is
//              very often the kept head of its cluster: the reducer looks at it first and thereby decides most
//              removed boxes without their lists (the lists hold the first EL ARRIVALS, in no order: in a
//              cluster of a few hundred near-duplicates the head is rarely among them).
struct Side {
  int* ecnt;
  int* msup;
  unsigned short* elist;
};

__host__ __device__ inline size_t side_words(size_t rows) {  // u64 words of the lists
  return (rows * EL + 3) / 4 + 2 * ((rows + 1) / 2);
}

__host__ __device__ inline Side side_tables(u64* side, size_t rows) {
  Side s;
  s.elist = reinterpret_cast<unsigned short*>(side);  // first: rows * 64 B keeps every row 64-byte aligned
  s.ecnt = reinterpret_cast<int*>(side + (rows * EL + 3) / 4);
  s.msup = reinterpret_cast<int*>(side + (rows * EL + 3) / 4 + (rows + 1) / 2);
  return s;
}

// pair (i, j), i < j in score order, suppresses: i joins the suppressors of j
__device__ __forceinline__ void mark_pair(u64* maskT, const Side& sd, unsigned i, unsigned j, int cb) {
  atomicMax(&sd.msup[j], 65535 - (int)i);
  const int s = atomicAdd(&sd.ecnt[j], 1);
  if (s < EL) sd.elist[(size_t)j * EL + s] = (unsigned short)i;
  else atomicOr(&maskT[(size_t)j * cb + (i >> 6)], 1ULL << (i & 63u));
}

template <int GEOM>
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* __restrict__ dets,
                                                          int det_stride,
                                                          const int64_t* __restrict__ labels,
                                                          const int64_t* __restrict__ order, int n,
                                                          BoxRec* __restrict__ recs,
                                                          unsigned* __restrict__ counter) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (counter)
    for (int k = i; k < Q_CTL_WORDS; k += gridDim.x * blockDim.x) counter[k] = 0;  // region fills, redo-list fill
  if (i >= n) return;
  int64_t src = order[i];
  float lab = labels ? (float)labels[src] : 0.f;  // at::cat({dets, labels}) promotes to float
  BoxRec r;
  make_record<GEOM>(dets + (size_t)src * det_stride, lab, r);
  recs[i] = r;
}

// ---------------------------------------------------------------------------- impl 1 (tiles)
// mask[row * cb + c] bit i  <=>  IoU(box_row, box_{64c+i}) > thr, for c >= row / 64.
template <int GEOM, bool LABEL>
__global__ __launch_bounds__(NT) void nms_mask_kernel(const BoxRec* __restrict__ recs, int n, int cb,
                                                      float thr, u64* __restrict__ mask) {
  __shared__ BoxRec cols[MASK_WAVES][TILE];
  // (round 5: the candidate list in LDS, [slot][lane] -- the register form of r3_geom.h spilled 416-640 B per lane here)
  __shared__ float2 pts[pts_slots<GEOM>() * NT];
  const LanePts<NT> lp{pts + threadIdx.x};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.y;
  const int cblk = blockIdx.x * MASK_WAVES + wave;
  const bool active = (cblk < cb) && (cblk >= rb);
  int col_size = 0;
  if (active) {
    col_size = min(n - cblk * TILE, TILE);
    if (lane < col_size) cols[wave][lane] = recs[cblk * TILE + lane];
  }
  __syncthreads();
  if (!active) return;
  const int row = rb * TILE + lane;
  if (row >= n) return;
  BoxRec A = recs[row];
  u64 t = 0;
  int start = (rb == cblk) ? lane + 1 : 0;
  // thr < 0 would make IoU == 0 suppress; the disjointness shortcut is only valid for thr >= 0
  const bool shortcut = thr >= 0.f;
  for (int i = start; i < col_size; i++) {
    const BoxRec& B = cols[wave][i];
    if (GEOM != 1 && LABEL && A.f[7] != B.f[7]) {
      if (0.f > thr) t |= 1ULL << i;
      continue;
    }
    if (shortcut && circles_apart(A.f[9], A.f[10], A.f[11], B.f[9], B.f[10], B.f[11])) continue;
    const float v = pair_slow_lds<GEOM, NT>(A.f, B.f, false, lp);
    if (v > thr) t |= 1ULL << i;
  }
  mask[(size_t)row * cb + cblk] = t;
}

__device__ __forceinline__ u64 readlane64(u64 v, int k) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffULL), k);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), k);
  return ((u64)hi << 32) | lo;
}

// greedy scan of one 64-row block on its diagonal word, all-scalar (host loops
// rnms_kernel.cu:316-327, nms_rotated_cuda.cu:117-128).  `cur` = bits already removed.
__device__ __forceinline__ u64 scan_block(u64 diag, u64 cur, int nvalid) {
  u64 kb = 0;
  for (int k = 0; k < nvalid; k++) {
    u64 dk = readlane64(diag, k);
    if (!((cur >> k) & 1ULL)) {
      kb |= 1ULL << k;
      cur |= dk;
    }
  }
  return kb;
}

__global__ __launch_bounds__(1024) void nms_reduce_dense_kernel(const u64* __restrict__ mask, int n,
                                                                int cb, const int64_t* __restrict__ order,
                                                                int64_t* __restrict__ keep_out,
                                                                int32_t* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) u64 remv[];  // cb words + 1 (kept-bits slot)
  u64* kb_slot = remv + cb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int j = tid; j < cb; j += blockDim.x) remv[j] = 0;
  __syncthreads();
  int cnt = 0;
  for (int b = 0; b < cb; b++) {
    if (wave == 0) {
      const int row = b * TILE + lane;
      u64 diag = (row < n) ? mask[(size_t)row * cb + b] : 0ULL;
      u64 cur = readlane64(remv[b], 0);
      u64 kb = scan_block(diag, cur, min(TILE, n - b * TILE));
      if ((kb >> lane) & 1ULL) {
        int pos = cnt + __popcll(kb & ((1ULL << lane) - 1ULL));
        keep_out[pos] = order[row];
      }
      cnt += __popcll(kb);
      if (lane == 0) *kb_slot = kb;
    }
    __syncthreads();
    const u64 kb = *kb_slot;
    const int nwords = cb - (b + 1);
    if (nwords > 0) {
      int idx = 0;
      u64 rem = kb;
      while (rem) {
        int k = __ffsll((long long)rem) - 1;
        rem &= rem - 1;
        if ((idx % nw) == wave) {
          const u64* p = mask + (size_t)(b * TILE + k) * cb + (b + 1);
          for (int j = lane; j < nwords; j += 64) {
            u64 v = p[j];
            if (v) atomicOr(&remv[b + 1 + j], v);
          }
        }
        idx++;
      }
    }
    __syncthreads();
  }
  if (tid == 0) *count_out = cnt;
}

// ---------------------------------------------------------------------------- queue pipeline
// stream: reject tests only.  One wave = one 64 x 64 tile (rows rb, columns cblk >= rb, sorted positions).
//   * The test loop is what the kernel costs (n = 8576: 18 M pairs), so it is as short as it gets: per column
//     one 16-byte LDS broadcast and the AABB test (two packed adds, two comparisons) into a per-lane bit, all 64
//     columns unrolled; the circle test, the label guard and the queue push only run for a lane's candidates.
//     (Measured and dropped: one scalar branch per column on the 64-row ballot -- with 1-2 % near pairs most
//     columns have SOME near row, and the dependent LDS-read -> compare -> branch chain cost 200 ns per column.)
//   * survivors -> the wave's private LDS segment (count in a scalar register, no atomics), flushed per wave
//     with ONE global atomic, all or nothing.
//   * a tile that does not fit (segment full: > 1/4 of its pairs survive; or the global queue is full) goes to
//     the REDO list and is enumerated pair by pair by the drain kernel.  Nothing is clipped here, so the kernel
//     needs no scratch and no point arrays.
// Entry in the global queue: (i << 16) | j, i < j sorted positions; 0xffffffff = hole (skipped by the drain).
constexpr int SQ_WSEG = 1024;  // u16 entries per wave segment (2 KB)

// PERM (round 6, pools beyond P_MIN_CAP): rows and columns are slots of the permutation `perm` (the candidates in x
// order, see mc_chunk_sort_kernel) instead of sorted positions; `ranges` holds the extent of every 64 slots, and a tile
// whose two extents are apart has no pair to test.  What is queued are sorted positions, smaller first, as before.
// The grid is one-dimensional there: a wavefront first looks at up to 64 tiles, one per lane (tile t = lane * waves +
// wave index, so that the band of tiles along the diagonal spreads over all wavefronts), and then runs the few that
// remain one after the other.  (One wavefront per tile that leaves after two scalar loads, the first form: 61 us at
// 32 768 rows, almost all of it the launch of 65 536 workgroups.)
template <int GEOM, bool LABEL, bool PERM = false>
__global__ __launch_bounds__(NT) void nms_stream_kernel(const BoxRec* __restrict__ recs, int n, int cb,
                                                        unsigned* __restrict__ gqueue, unsigned qcap,
                                                        unsigned* __restrict__ counter, unsigned* __restrict__ redo,
                                                        Batch bt, const unsigned short* __restrict__ perm = nullptr,
                                                        const float4* __restrict__ ranges = nullptr,
                                                        size_t perm_stride = 0, size_t ranges_stride = 0,
                                                        int tile_waves = 0) {
  __shared__ __attribute__((aligned(16))) float4 colsA[MASK_WAVES][TILE];  // cx, cy, ex, ey
  __shared__ float2 colsB[MASK_WAVES][TILE];                               // radius, label
  __shared__ unsigned short queue[MASK_WAVES][SQ_WSEG];
  __shared__ unsigned short ppos[PERM ? MASK_WAVES : 1][2][TILE];          // PERM: the tile's rows / columns as positions
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // tells the compiler it is wave-uniform: scalar control flow
  const int rb0 = PERM ? 0 : (int)blockIdx.y;
  if (bt.counts) {  // cb stays the row pitch of mask; the image's own block count bounds the tiles
    const int img = blockIdx.z;
    n = bt.counts[img];
    recs += img * bt.recs;
    gqueue += img * bt.queue;
    counter += img * bt.counter;
    redo += img * bt.redo;
    if (PERM) {
      perm += img * perm_stride;
      ranges += img * ranges_stride;
    }
    if (rb0 * TILE >= n) return;
  }
  const int cbn = (n + TILE - 1) / TILE;
  // one tile (waves are independent: no workgroup barrier in here)
  auto tile = [&](const int rb, const int cblk) {
  if (!PERM && rb == cblk && bt.counts && bt.gbits && bt.rlab) {
    // the diagonal tile of a block of 64 rows also leaves the block's word of every label group's row bitmap (what the
    // reducer's workgroups start from): 16 ballots, a plain store by lane g -- every word written once, nothing to zero
    const int r = rb * TILE + lane;
    const int lg = r < n ? (int)(bt.rlab[(size_t)blockIdx.z * bt.rows + r] & (RG_GROUPS - 1)) : -1;
    u64 mine = 0;
#pragma unroll
    for (int g = 0; g < RG_GROUPS; g++) {
      const u64 mg = __ballot(lg == g);
      if (lane == g) mine = mg;
    }
    if (lane < RG_GROUPS) bt.gbits[((size_t)blockIdx.z * RG_GROUPS + lane) * cb + rb] = mine;
  }
  const int col_size = min(n - cblk * TILE, TILE);
  int pcol_l = cblk * TILE + lane;
  if (PERM) {
    pcol_l = lane < col_size ? (int)perm[cblk * TILE + lane] : 0;
    ppos[wave][1][lane] = (unsigned short)pcol_l;
  }
  if (lane < col_size) {
    // the reject data sit in the record's last two 16-byte quads (f[9..13]), the label in f[7]
    const float4* f4 = reinterpret_cast<const float4*>(recs[pcol_l].f);
    const float4 q2 = f4[2], q3 = f4[3];
    colsA[wave][lane] = make_float4(q2.y, q2.z, q3.x, q3.y);
    colsB[wave][lane] = make_float2(q2.w, (GEOM != 1) ? f4[1].w : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave reads what its own lanes wrote: no workgroup barrier
  __builtin_amdgcn_wave_barrier();
  const int row = rb * TILE + lane;
  int rr_ = row < n ? row : n - 1;
  if (PERM) {
    rr_ = (int)perm[rr_];
    ppos[wave][0][lane] = (unsigned short)rr_;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (read by other lanes at the flush)
    __builtin_amdgcn_wave_barrier();
  }
  const float4* f4 = reinterpret_cast<const float4*>(recs[rr_].f);
  const float4 q2 = f4[2], q3 = f4[3];
  const float ax = q2.y, ay = q2.z, ar = q2.w, aex = q3.x, aey = q3.y;
  const float alab = (GEOM != 1) ? f4[1].w : 0.f;
  const bool diag = rb == cblk;
  unsigned short* wq = queue[wave];
  int cnt = 0;        // wave-uniform
  bool full = false;  // wave-uniform
  // 64 AABB tests into one per-lane bit mask: straight-line code, the LDS broadcasts run ahead of the tests
  // (8 columns per step: unrolling all 64 lets the compiler hoist every LDS read -- 256 VGPRs, 2 waves per SIMD)
  unsigned long long pm = 0;
#pragma unroll 1
  for (int g = 0; g < TILE / 8; g++) {
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float4 B = colsA[wave][g * 8 + k];
      const bool near = !(fabsf(ax - B.x) > aex + B.z) && !(fabsf(ay - B.y) > aey + B.w);
      bits |= near ? (1u << k) : 0u;
    }
    pm |= (unsigned long long)bits << (8 * g);
  }
  if (col_size < TILE) pm &= (1ULL << col_size) - 1ULL;   // columns beyond the pool hold stale LDS
  if (diag) pm &= ~((2ULL << lane) - 1ULL);              // pairs above the diagonal only: column > row
  if (row >= n) pm = 0;
  // the candidates (a few per lane): circle test, label guard, push.  Wave-uniform loop, one candidate per lane
  // and turn, so that the fill count stays in a scalar register.
  while (__builtin_amdgcn_ballot_w64(pm != 0)) {
    const bool has = pm != 0;
    const int c = has ? __builtin_ctzll(pm) : 0;
    pm &= pm - 1;
    const float4 B = colsA[wave][c];
    const float2 B2 = colsB[wave][c];
    const float dx = ax - B.x, dy = ay - B.y, rr = ar + B2.x;
    bool p = has && !(dx * dx + dy * dy > rr * rr);
    if (GEOM != 1 && LABEL) p = p && (alab == B2.y);
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p);
    if (m2 == 0) continue;
    if (cnt + 64 > SQ_WSEG) {
      full = true;
      break;
    }
    if (p) wq[cnt + __popcll(m2 & ((1ULL << lane) - 1ULL))] = (unsigned short)((lane << 6) | c);
    cnt += __popcll(m2);
  }
  if (!full && cnt == 0) return;
  const unsigned reg = ((unsigned)rb * 5u + (unsigned)cblk) % (unsigned)Q_NREG;
  unsigned* region = gqueue + (size_t)reg * qcap;  // qcap: entries per region
  unsigned base = 0;
  if (!full) {
    if (lane == 0) base = atomicAdd(counter + reg * Q_CSTRIDE, (unsigned)cnt);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + (unsigned)cnt > qcap) {  // region exhausted: plug what was reserved, redo the tile
      for (unsigned q = base + lane; q < base + (unsigned)cnt && q < qcap; q += 64) region[q] = 0xffffffffu;
      full = true;
    }
  }
  if (full) {
    if (lane == 0) redo[atomicAdd(counter + Q_REDO, 1u)] = ((unsigned)rb << 16) | (unsigned)cblk;
    return;
  }
  for (int q = lane; q < cnt; q += 64) {
    const unsigned e = wq[q];
    if (PERM) {
      const unsigned i = ppos[wave][0][e >> 6], j = ppos[wave][1][e & 63u];
      region[base + q] = (min(i, j) << 16) | max(i, j);
    } else {
      region[base + q] = ((unsigned)(rb * TILE) + (e >> 6)) << 16 | ((unsigned)(cblk * TILE) + (e & 63u));
    }
  }
  };
  if (!PERM) {
    const int cblk = blockIdx.x * MASK_WAVES + wave;
    if (cblk >= cbn || cblk < rb0) return;
    tile(rb0, cblk);
    return;
  }
  // Tiles are numbered DIAGONAL BY DIAGONAL (t = (column chunk - row chunk) * cbn + row chunk): in x order the tiles
  // that remain hug the diagonal, so they are the first few cbn numbers and t = lane * waves + wave index deals them out
  // evenly.  (Row by row, the diagonal's tiles are cbn + 1 apart and pile up on the wavefronts whose count shares a
  // factor with that: cbn = 191 and 2048 wavefronts left 288 busy with six dense tiles each -- 50 us instead of 13.)
  const int nwaves = tile_waves, wv = (int)blockIdx.x * MASK_WAVES + wave;
  if (wv >= nwaves) return;
  const unsigned t = (unsigned)lane * (unsigned)nwaves + (unsigned)wv;  // (cbn <= 1024: below 2^20 + 64 * waves)
  bool live = t < (unsigned)(cbn * cbn);
  int trb = 0, tcb = 0;
  if (live) {
    const int d = (int)(t / (unsigned)cbn);
    trb = (int)(t - (unsigned)d * (unsigned)cbn);
    tcb = trb + d;
    live = tcb < cbn;
  }
  if (live && trb != tcb) {  // (a NaN extent compares false: the tile is kept)
    const float4 ra = ranges[trb], rc = ranges[tcb];
    live = !(ra.y < rc.x || rc.y < ra.x || ra.w < rc.z || rc.w < ra.z);
  }
  unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
  while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the previous tile's LDS reads are behind us)
    __builtin_amdgcn_wave_barrier();
    tile(__builtin_amdgcn_readlane(trb, src), __builtin_amdgcn_readlane(tcb, src));
  }
}

template <int GEOM, bool LABEL, bool FAST = false>
__global__ __launch_bounds__(256, FAST ? 4 : 1) void nms_drain_kernel(const BoxRec* __restrict__ recs, int n, int cb, float thr,
                                                        const unsigned* __restrict__ gqueue, unsigned qcap,
                                                        unsigned* __restrict__ counter,
                                                        const unsigned* __restrict__ redo,
                                                        u64* __restrict__ mask, u64* __restrict__ side, Batch bt,
                                                        const unsigned short* __restrict__ perm = nullptr,
                                                        size_t perm_stride = 0) {
  // v1: 8 candidate slots per lane in wave-private [slot][lane] regions (half the LDS of the reference's 16 slots:
  // twice the resident waves; one clip is ~15 us of latency, so an image's pairs should take ONE trip); the rare
  // pair with a 9th candidate is redone by lanes 0..31 with 16 slots in the same region (as in the IoU drain)
  // Round 5 (FAST): the straight-line clip (r3_clip.h), 9 slots per lane; flagged pairs take the same redo
  // Round 5, v2 / v3 (hull): 12 of the 24 slots per lane and the same redo (a pair with a 13th point is redone by lanes
  // 0..31 with 24 slots in the same region): 24.5 KB of LDS instead of 49 KB, occupancy 3 -> 5+ (the IoU drain's form)
  constexpr bool SHORT = true;
  constexpr int CAPS = GEOM == 1 ? (FAST ? R3_CLIP_SLOTS : 8) : 12;
  __shared__ float2 pts[CAPS * 256];
  __shared__ unsigned pre[Q_NREG + 1];  // exclusive prefix of the regions' (clamped) fills
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bt.counts) {
    const int img = blockIdx.z;
    n = bt.counts[img];
    recs += img * bt.recs;
    gqueue += img * bt.queue;
    counter += img * bt.counter;
    redo += img * bt.redo;
    mask += img * bt.mask;
    side += img * bt.nz;
    if (perm) perm += img * perm_stride;
  }
  const uint8_t* rlab = (bt.counts && bt.rlab) ? bt.rlab + (size_t)blockIdx.z * bt.rows : nullptr;
  const Side sd = side_tables(side, bt.rows);
  if (threadIdx.x < 64) {
    const unsigned v = min(counter[threadIdx.x * Q_CSTRIDE], qcap);
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
  // one pair (i, j), valid or not, per lane: clip, mark, and redo what did not fit 8 slots (wave-level: every lane
  // of the wave comes through here together)
  auto clip_pair = [&](const bool valid, const unsigned i, const unsigned j) {
    bool over = false;
    if (valid) {
      const BoxRec A = recs[i];
      const BoxRec B = recs[j];
      float v;
      if constexpr (GEOM == 1 && FAST) {
        v = v1_clip_fast(A.f, B.f, false, ClipLds<64>{pts + wave * (64 * CAPS) + lane}, over);
      } else if constexpr (GEOM == 1) {
        const LanePts<64> lp8{pts + wave * (64 * CAPS) + lane};
        v = v1_pair_lds<64, 8>(A.f, B.f, false, lp8, &over);
      } else if constexpr (FAST) {
        v = hull_clip_fast<GEOM == 2>(A.f, B.f, true, ClipLds<64>{pts + wave * (64 * CAPS) + lane}, over);
      } else {
        const LanePts<64> lp12{pts + wave * (64 * CAPS) + lane};
        v = hull_pair_lds<GEOM == 2, 64, 12>(A.f, B.f, true, lp12, &over);
      }
      if (!over && v > thr) {
        mark_pair(mask, sd, i, j, cb);
        if (rlab && ((rlab[i] ^ rlab[j]) & (RG_GROUPS - 1))) counter[Q_XFLAG] = 1u;
      }
    }
    if (SHORT) {
      unsigned long long m = __ballot(over);
      while (m) {
        int src = -1, seen = 0;
        for (unsigned long long t2 = m; t2; t2 &= t2 - 1) {
          if (seen == lane) src = __builtin_ctzll(t2);
          seen++;
        }
        const unsigned ii = __shfl(i, src < 0 ? 0 : src), jj = __shfl(j, src < 0 ? 0 : src);
        if (lane < 32 && src >= 0) {
          const BoxRec A = recs[ii];
          const BoxRec B = recs[jj];
          const LanePts<32> lpf{pts + wave * (64 * CAPS) + lane};  // (32 lanes x the full slots = the same region)
          float v;
          if constexpr (GEOM == 1) v = v1_pair_lds<32, R3_V1_CAP>(A.f, B.f, false, lpf);
          else v = hull_pair_lds<GEOM == 2, 32, 24>(A.f, B.f, true, lpf);
          if (v > thr) {
            mark_pair(mask, sd, ii, jj, cb);
            if (rlab && ((rlab[ii] ^ rlab[jj]) & (RG_GROUPS - 1))) counter[Q_XFLAG] = 1u;
          }
        }
        for (int k = 0; k < 32 && m; k++) m &= m - 1;
      }
    }
  };
  const unsigned total = pre[Q_NREG];
  for (unsigned qb = blockIdx.x * 256 + wave * 64; qb < total; qb += gridDim.x * 256) {  // wave-uniform trips
    const unsigned q = qb + lane;
    bool valid = q < total;
    unsigned i = 0, j = 0;
    if (valid) {
      int lo = 0;  // region of entry q: largest lo with pre[lo] <= q
#pragma unroll
      for (int step = Q_NREG / 2; step >= 1; step >>= 1)
        if (pre[lo + step] <= q) lo += step;
      const unsigned e = gqueue[(size_t)lo * qcap + (q - pre[lo])];
      valid = e != 0xffffffffu;
      i = e >> 16;
      j = e & 0xffffu;
    }
    clip_pair(valid, i, j);
  }
  // redo tiles (dense clusters, exhausted queue): all 64 x 64 pairs, 256 per step, tested with the records
  const unsigned units = counter[Q_REDO] * 16u;
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned t = redo[u >> 4];
    const unsigned p = (u & 15u) * 256u + threadIdx.x;
    unsigned i = (t >> 16) * TILE + (p >> 6), j = (t & 0xffffu) * TILE + (p & 63u);
    bool valid = i < (unsigned)n && j < (unsigned)n && i < j;
    if (perm && valid) {  // (the sorted-chunk form: the tile names slots of the permutation; the pair is their positions)
      const unsigned a = perm[i], b = perm[j];
      i = min(a, b);
      j = max(a, b);
    }
    if (valid) {
      const BoxRec A = recs[i];
      const BoxRec B = recs[j];
      valid = !boxes_apart(A.f, B.f) && !(GEOM != 1 && LABEL && A.f[7] != B.f[7]);
    }
    clip_pair(valid, i, j);
  }
}


// ---------------------------------------------------------------------------- reduce by dependency rounds
// Greedy NMS in score order: row r is kept  <=>  no kept row i < r has IoU(i, r) > thr.  elist[r] (+ the
// overflow row of the transposed mask) lists exactly those i.  Kept (K) and removed (R) bit sets live in LDS
// and only ever receive FINAL decisions: r -> R once some suppressor is in K, r -> K once all suppressors are
// in R, so the order in which threads look is irrelevant and the fixed point is the sequential answer.  One
// workgroup per problem, thread <-> rows tid, tid + 1024, ...; a row's count and its 64-byte list are five
// independent loads at computable addresses; every thread keeps the counts and first 8 suppressors of its
// (up to 9) rows in registers for all rounds, so a round is LDS bit lookups plus on-demand loads for the rows
// with more than 8 suppressors.  Measured alternatives: gathering mask WORDS through a per-row word
// list (three dependent round trips per row: 189 us at n = 8576); round 1's walk over 64-row blocks with two
// barriers per block (36 us per 4 x 3.3 k boxes, 100 us at n = 8576).
// On detection pools (clusters of near-duplicates) almost every row decides in rounds 1-3; a row can stay
// undecided only as long as a suppression CHAIN runs through it, so R_MAX_ROUNDS bounds the parallel part
// and wave 0 finishes adversarial inputs row by row in score order.
constexpr int RTHREADS = 1024;
constexpr int R_MAX_ROUNDS = 32;
constexpr int R_HANDOVER = 3;  // (in-register form of the label-group reducer) rows beyond their list go to the wavefronts' pass behind this round
constexpr int R_CACHE = 9;     // rows per thread whose count and first 8 suppressors stay in registers (n <= 9216)
constexpr int R_BLIST = 8192;  // rows on the long-list worklist (u16 each)

// LDS of the reducer (dynamic): state bytes | K words | R words | block prefix | long-list worklist
__host__ __device__ inline size_t reduce_lds_bytes(int n, int cb) {
  return (size_t)((n + 15) & ~15) + (size_t)cb * 8 * 2 + (size_t)cb * 4 + (size_t)R_BLIST * 2 + 16;
}

// Measured inside the kernel (wall clock, n = 8576, 70 us before): round 1's pass over the register-cached rows
// 9.5 us -- 144 bit lookups per thread, two LDS reads each, serialised by bank conflicts; the rows with more than
// 8 suppressors 11.5 us per round -- nine serial iterations at ~10 % lane use, two dependent global loads each;
// the keep list 8 us -- one thread per 64-row block writing its kept rows one by one.  Hence:
//   * one STATE BYTE per row in LDS (0 undecided, 1 kept, 2 removed): one read per lookup, written by the row's
//     owner only (no atomics); the K / R bit words are kept as well (ballots) for the overflow-mask scans;
//   * the long-list rows are compacted ONCE into a worklist and visited one per thread and round;
//   * the keep list is written one row per thread from the block prefix.
__global__ __launch_bounds__(RTHREADS) void nms_reduce_rounds_kernel(const u64* __restrict__ maskT,
                                                                     const u64* __restrict__ side, int n, int cb,
                                                                     const int64_t* __restrict__ order,
                                                                     int64_t* __restrict__ keep_out,
                                                                     int32_t* __restrict__ count_out, Batch bt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  __shared__ int s_und, s_nbig;
  __shared__ int wsum[RTHREADS / 64];
  if (bt.counts) {
    const int img = blockIdx.z;
    n = bt.counts[img];
    maskT += img * bt.mask;
    side += img * bt.nz;
    keep_out += img * bt.keep;
    count_out += img;
  }
  const int cbn = (n + TILE - 1) / TILE;
  const int nb = (n + 15) & ~15;
  unsigned char* st = smem8;                               // row state
  u64* Kb = reinterpret_cast<u64*>(smem8 + nb);            // cb words: rows known kept
  u64* Rb = Kb + cb;                                       // cb words: rows known removed
  int* bpre = reinterpret_cast<int*>(Rb + cb);             // exclusive prefix of the kept counts per block
  unsigned short* blist = reinterpret_cast<unsigned short*>(bpre + cb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Side sd = side_tables(const_cast<u64*>(side), bt.rows);

  // prologue: every thread fetches the counts and first chunks of its rows (all loads in flight together)
  int cnt[R_CACHE];
  uint4 c0[R_CACHE];
#pragma unroll
  for (int u = 0; u < R_CACHE; u++) {
    const int r = tid + u * RTHREADS;
    const int rr = r < n ? r : 0;
    cnt[u] = sd.ecnt[rr];
    c0[u] = *reinterpret_cast<const uint4*>(sd.elist + (size_t)rr * EL);
  }
  if (tid == 0) s_nbig = 0;
  for (int j = tid; j < cb; j += RTHREADS) {
    Kb[j] = 0;
    Rb[j] = 0;
  }
  __syncthreads();
  // rows without suppressors are kept: round 1 then already sees the cluster heads.  Rows with more than 8
  // suppressors (or beyond the register cache) go to the worklist; what does not fit stays with its owner.
  unsigned own = 0;  // cached rows with a long list that did not fit the worklist
#pragma unroll
  for (int u = 0; u < R_CACHE; u++) {
    const int r = tid + u * RTHREADS;
    const bool kept0 = r < n && cnt[u] == 0;
    if (r < n) st[r] = kept0 ? 1 : 0;
    const u64 k0 = __ballot(kept0);
    if (lane == 0 && k0) Kb[r >> 6] = k0;
    const bool isbig = r < n && cnt[u] > 8;
    const u64 mb = __ballot(isbig);
    if (mb) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_nbig, __popcll(mb));
      base = __builtin_amdgcn_readfirstlane(base);
      const int slot = base + __popcll(mb & ((1ULL << lane) - 1ULL));
      if (isbig) {
        if (slot < R_BLIST) blist[slot] = (unsigned short)r;
        else own |= 1u << u;
      }
    }
  }
  for (int u = R_CACHE; wave * 64 + u * RTHREADS < n; u++) {
    const int r = tid + u * RTHREADS;
    const bool kept0 = r < n && sd.ecnt[r < n ? r : 0] == 0;
    if (r < n) st[r] = kept0 ? 1 : 0;
    const u64 k0 = __ballot(kept0);
    if (lane == 0 && k0) Kb[r >> 6] = k0;
  }
  __syncthreads();
  const int nbig = min(s_nbig, R_BLIST);

  // one row against the K / R sets; returns 0 undecided, 1 kept, 2 removed
  auto long_row = [&](const int r) -> int {
    const unsigned m = 65535u - (unsigned)sd.msup[r];  // its highest-scored suppressor first
    if (m < 65535u && st[m] == 1) return 2;
    const int c = sd.ecnt[r];
    const uint4* lp = reinterpret_cast<const uint4*>(sd.elist + (size_t)r * EL);
    const uint4 t[4] = {lp[0], lp[1], lp[2], lp[3]};
    const int listed = min(c, EL);
    bool anyK = false, allR = true;
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned wv[4] = {t[c4].x, t[c4].y, t[c4].z, t[c4].w};
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const unsigned i = (wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
        const bool on = 8 * c4 + q < listed;
        const unsigned char v = st[on ? i : 0];
        anyK |= on && v == 1;
        allR &= !on || v == 2;
      }
    }
    if (c > EL && !anyK) {
      // suppressors beyond the list: scan the overflow row, 8 independent word loads per step
      const u64* row = maskT + (size_t)r * cb;
      const int w = r >> 6;
      for (int q0 = 0; q0 <= w && !anyK; q0 += 8) {
        u64 mm[8];
#pragma unroll
        for (int e = 0; e < 8; e++) mm[e] = row[min(q0 + e, w)];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int q = min(q0 + e, w);
          anyK |= (mm[e] & Kb[q]) != 0ULL;
          allR &= (mm[e] & ~Rb[q]) == 0ULL;
        }
      }
    }
    return anyK ? 2 : (allR ? 1 : 0);
  };

  int round = 0;
  for (; round < R_MAX_ROUNDS; round++) {
    if (tid == 0) s_und = 0;
    __syncthreads();
    int und = 0;
    // pass A: the cached rows with at most 8 suppressors, straight from registers
#pragma unroll
    for (int u = 0; u < R_CACHE; u++) {
      const int r = tid + u * RTHREADS;
      const bool act = r < n && cnt[u] <= 8 && cnt[u] > 0 && st[r] == 0;
      if (__ballot(act) == 0ULL) continue;
      bool anyK = false, allR = true;
      if (act) {
        unsigned wv[4] = {c0[u].x, c0[u].y, c0[u].z, c0[u].w};
        // opaque copies: without them the compiler hoists the 16 derived indices / shifts of every cached
        // row out of the round loop and spills (34 registers per row instead of 5)
        asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]));
        unsigned char v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {  // eight independent byte reads, then the logic
          const unsigned i = (wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
          v[q] = st[q < cnt[u] ? i : 0];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const bool on = q < cnt[u];
          anyK |= on && v[q] == 1;
          allR &= !on || v[q] == 2;
        }
      }
      const bool toR = act && anyK, toK = act && !anyK && allR;
      if (toR | toK) st[r] = toR ? 2 : 1;
      const u64 mr = __ballot(toR), mk = __ballot(toK);
      und += act && !anyK && !allR;
      if (lane == 0) {
        if (mr) Rb[r >> 6] |= mr;
        if (mk) Kb[r >> 6] |= mk;
      }
    }
    // pass B: the worklist of long rows, one per thread.  (A wave's rows are not one 64-row block here: the bit
    // words take LDS atomics -- few rows, spread over the words.)
    for (int k = tid; k < nbig; k += RTHREADS) {
      const int r = blist[k];
      if (st[r] != 0) continue;
      const int d = long_row(r);
      if (d) {
        st[r] = (unsigned char)d;
        atomicOr(d == 1 ? &Kb[r >> 6] : &Rb[r >> 6], 1ULL << (r & 63));
      } else {
        und++;
      }
    }
    // pass C: cached long rows that did not fit the worklist, and rows beyond the register cache (n > 9216)
    if (own || n > R_CACHE * RTHREADS) {
      for (int u = 0; wave * 64 + u * RTHREADS < n; u++) {
        const int r = tid + u * RTHREADS;
        const bool mine = u < R_CACHE ? ((own >> u) & 1u) : true;
        if (!(r < n && mine && st[r] == 0)) continue;
        if (u >= R_CACHE && sd.ecnt[r] == 0) continue;
        const int d = long_row(r);
        if (d) {
          st[r] = (unsigned char)d;
          atomicOr(d == 1 ? &Kb[r >> 6] : &Rb[r >> 6], 1ULL << (r & 63));
        } else {
          und++;
        }
      }
    }
    if (und) atomicAdd(&s_und, und);
    __syncthreads();
    const int left = s_und;
    __syncthreads();
    if (left == 0) break;
  }
  if (round == R_MAX_ROUNDS && tid < 64) {
    // a suppression chain longer than the round budget: wave 0 finishes in score order (every earlier row is
    // decided when a row's turn comes, so "no kept suppressor" decides it); lanes split the row's suppressors
    for (int b = 0; b < cbn; b++) {
      const int nvalid = min(TILE, n - b * TILE);
      const u64 valid = nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1ULL);
      u64 todo = valid & ~(Kb[b] | Rb[b]);
      while (todo) {
        const int k = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int r = b * TILE + k;
        const int cnt = sd.ecnt[r];
        bool hit = false;
        if (lane < min(cnt, EL)) {
          const unsigned i = sd.elist[(size_t)r * EL + lane];
          hit = (Kb[i >> 6] >> (i & 63)) & 1ULL;
        }
        if (cnt > EL) {
          const u64* row = maskT + (size_t)r * cb;
          for (int q = lane; q <= b; q += 64) hit |= (row[q] & Kb[q]) != 0ULL;
        }
        const bool removed = __ballot(hit) != 0ULL;
        if (lane == 0) {
          if (removed) Rb[b] |= 1ULL << k;
          else Kb[b] |= 1ULL << k;
          st[r] = removed ? 2 : 1;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      }
    }
  }
  __syncthreads();
  // keep list: block counts -> exclusive scan (wave shuffles + 16 partials) -> one kept row per thread
  int total = 0;
  for (int b0 = 0; b0 < cbn; b0 += RTHREADS) {  // (cbn <= 1024: one trip)
    const int b = b0 + tid;
    const int c = b < cbn ? __popcll(Kb[b]) : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RTHREADS / 64; w++) {
      const int t = wsum[w];
      if (w < wave) woff += t;
      tot += t;
    }
    if (b < cbn) bpre[b] = total + woff + incl - c;
    total += tot;
    __syncthreads();
  }
  for (int r = tid; r < n; r += RTHREADS) {
    const u64 kb = Kb[r >> 6];
    if ((kb >> (r & 63)) & 1ULL)
      keep_out[bpre[r >> 6] + __popcll(kb & ((1ULL << (r & 63)) - 1ULL))] = order ? order[r] : (int64_t)r;
  }
  if (tid == 0) *count_out = total;
}

// ---------------------------------------------------------------------------- batched runs: one reducer per label
// The batched pipeline (r3det_mcnms*: per-class NMS through class offsets or the label guard) has no suppressor edge
// between boxes of different labels -- by construction for the label guard (v2), and for the offsets (v1 / v3)
// whenever no box reaches across an offset, which the drain checks edge by edge (counter[Q_XFLAG]).  Then the greedy
// reduction of an image falls apart into one independent problem per label: grid = (labels, 1, images), every
// workgroup compacts the rows of its label (score order kept) and runs the dependency rounds over those ~n / 15 rows
// only -- one workgroup per image walked all 8576 rows for 38 us.  (Groups = labels mod 16: any number of classes, a
// fixed grid.)  Flag set: workgroup 0 of the image runs the rounds over all rows, the others leave.  Output: the kept rows as bits in `kbits` (zeroed with the side tables), which the
// finish kernels turn into the detections; no keep list.
constexpr int RG_MAXN = 32768;  // largest row capacity with grouping (LDS: state bytes + row list + worklist: 124 KB there)

__host__ __device__ inline size_t reduce_groups_lds_bytes(int n, int cb, bool with_row_list) {
  return (size_t)((n + 15) & ~15) + (size_t)cb * 8 * 3 + (with_row_list ? (size_t)n * 2 : 0) + (size_t)R_BLIST * 2 + 64;
}

u64* g_nms_stamps = nullptr;  // tools/probes/nms_reduce_probe.hip: clock stamps of reducer workgroup (0, 0) at its phases

template <bool GROUPED>
__device__ __forceinline__ void reduce_groups_body(const u64* __restrict__ maskT, const Side& sd, const int n,
                                                   const int bt_rows, const int cb, const u64* __restrict__ gbm,
                                                   const int group,
                                                   u64* __restrict__ kbits, const int* __restrict__ svals,
                                                   u64* __restrict__ fbits, unsigned char* smem8, int* s_und,
                                                   int* s_nbig, int* s_m, int* wsum, u64* __restrict__ stamps) {
  // (probe: stamps[15] names the group whose workgroup, image 0, writes the stamps)
  const bool stamp_on = stamps && blockIdx.z == 0 && threadIdx.x == 0 && blockIdx.x == (unsigned)stamps[15];
  auto stamp = [&](int k) {
    if (stamp_on) stamps[k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  constexpr int OVQ = 2048;  // rows with more than EL suppressors the wavefronts' pass can take: every row of an in-register label group
  __shared__ unsigned short ovq[OVQ];
  __shared__ unsigned char ovl[OVQ];  // (general form: what the row's list said -- all of its entries removed)
  __shared__ int hcnt[2];    // [0]: rows handed over to the wavefronts' pass (in-register form; they stay on its list)
  __shared__ int hdec[RTHREADS / 64];  // ... and how many of them each wavefront left undecided in the round
  const int cbn = (n + TILE - 1) / TILE;
  const int nb = (n + 15) & ~15;
  unsigned char* st = smem8;                               // row state: 0 undecided, 1 kept, 2 removed
  u64* Kb = reinterpret_cast<u64*>(smem8 + nb);            // cb words: rows known kept
  u64* Rb = Kb + cb;                                       // cb words: rows known removed
  u64* Own = Rb + cb;                                      // cb words: rows of this workgroup
  unsigned short* rows_l = reinterpret_cast<unsigned short*>(Own + cb);  // this workgroup's rows, ascending
  unsigned short* blist = rows_l + (gridDim.x > 1 ? bt_rows : 0);         // (the list exists in multi-group launches)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < cb; j += RTHREADS) {
    Kb[j] = 0;
    Rb[j] = 0;
    Own[j] = 0;
  }
  if (tid == 0) {
    *s_nbig = 0;
    *s_m = 0;
    hcnt[0] = hcnt[1] = 0;
  }
  __syncthreads();
  int m = n;
  if (GROUPED) {
    // The label group's rows, ascending, from its BITMAP (round 6; gbm: bit p of word p / 64 = sorted row p belongs to
    // this group -- one atomicOr by the kernel that placed the record): a thread per word, the words' popcounts scanned
    // over the workgroup, every thread writes its word's rows.  (Until then every group's workgroup read ALL rows'
    // labels, 1024 per trip with a ballot each: 26 k of a group's 70-90 k clocks at 32 768 rows, a third of the kernel at
    // every size.)  The bitmap is also `Own`.
    const u64 w = tid < cbn ? gbm[tid] : 0ULL;  // (cbn <= 512: RG_MAXN rows)
    if (tid < cbn) Own[tid] = w;
    const int pc = __popcll(w);
    int incl = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    int* wtab = reinterpret_cast<int*>(blist);  // 16 wave totals (the worklist is not in use yet)
    if (lane == 63) wtab[wave] = incl;
    __syncthreads();
    int o = incl - pc, total = 0;
#pragma unroll
    for (int q = 0; q < RTHREADS / 64; q++) {
      const int t = wtab[q];
      o += q < wave ? t : 0;
      total += t;
    }
    for (u64 x = w; x; x &= x - 1) rows_l[o++] = (unsigned short)(tid * 64 + __builtin_ctzll(x));
    m = total;
    __syncthreads();
  }
  stamp(1);
  auto row_of = [&](const int k) -> int { return GROUPED ? (int)rows_l[k] : k; };
  auto set_bit = [&](u64* words, const int r) { atomicOr(&words[r >> 6], 1ULL << (r & 63)); };

  int round = 0;
  bool fbits_done = false;
  if (GROUPED && m <= 2 * RTHREADS) {
    // A label's rows fit 1 or 2 per thread (n / 15 rows at 15 balanced classes; a pool dominated by one class --
    // the random-init bench model: 89 % of its 2170 candidates in one label -- fits at 2; 4 rows per thread need
    // more than the 128 registers a 1024-thread workgroup has): count, best suppressor
    // and the WHOLE 32-entry list stay in registers, a round is 32 LDS byte reads per row and one barrier.  (The
    // general form below re-reads the lists of rows with more than 8 suppressors from global memory in every round
    // -- most rows of a detector's clustered pool: 8.5 us for two rounds.)
    auto in_regs = [&](auto rpt_tag) -> int {
      constexpr int RPT = decltype(rpt_tag)::value;
      int rr[RPT], c[RPT], state[RPT], sv[RPT];
      unsigned ms[RPT];
      uint4 t[RPT][4];
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        const int k = tid + u * RTHREADS;
        const bool has = k < m;
        // (all of a row's loads unconditional, at row 0 for a thread without a row: one batch, no branch in between.
        // Measured and not kept: the same loads requested during the label compaction for the thread's first two rows
        // of the label and handed over through LDS (with an LDS-only barrier, so that they stay in flight): the 2 %
        // of threads with a third row of the label still wait for memory here, in four waves of five, and the extra
        // registers spill: 33 k -> 54 k cycles.)
        rr[u] = has ? (int)rows_l[k] : 0;
        const int sv_l = svals[rr[u]];  // (the row's candidate index: for the keep bits behind the rounds)
        const int c_l = sd.ecnt[rr[u]];
        const int ms_l = sd.msup[rr[u]];
        const uint4* lp = reinterpret_cast<const uint4*>(sd.elist + (size_t)rr[u] * EL);
        t[u][0] = lp[0]; t[u][1] = lp[1]; t[u][2] = lp[2]; t[u][3] = lp[3];
        sv[u] = sv_l;
        c[u] = has ? c_l : 0;
        ms[u] = has ? 65535u - (unsigned)ms_l : 65535u;
        state[u] = has ? 0 : 3;
      }
      // Rows with more suppressors than the 32-entry list (mark_pair keeps the rest as bits of the row's overflow mask)
      // are ENLISTED in a round that leaves them undecided, and a wavefront takes each: a lane per mask word, the K / R
      // bits of the word's 64 rows rebuilt from their state bytes (8 LDS reads).  History: the owner re-scanning the mask
      // row from global memory in every round (round 4: a dozen such rows cost 30 us of a 48 us kernel on the bench
      // model's pool); the bits turned into u16 LDS lists once, up front (round 5: 45-73 k clocks in front of the first
      // round), then behind round 0 and for undecided rows only (round 6: still 5-11 k clocks per label group -- the
      // conversion is a round trip to the mask plus two dependent cross-lane scans per row -- and the owner walking its
      // list alone kept the workgroup at the round's barrier for 6-10 k clocks).
      // Round 6: a look at the HEADS first.  A heavy row is nearly always a member of a cluster whose head -- its
      // highest-scored suppressor -- has no suppressor itself and is kept from the start: such a row is removed here.
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        if (state[u] == 0 && c[u] == 0) state[u] = 1;
        if (state[u] != 3) st[rr[u]] = (unsigned char)state[u];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < RPT; u++)
        if (state[u] == 0 && ms[u] < 65535u && st[ms[u]] == 1) {
          state[u] = 2;
          st[rr[u]] = 2;
        }
      __syncthreads();
      stamp(2);
      // (no K / R bit sets in this loop: 64 lanes deciding rows of one or two mask words are 64 same-address LDS
      // atomics; the sets are rebuilt from the state bytes once, behind the loop)
      int rnd = 0, left_before = -1;
      // one round; true: the loop is over (every row decided, or stuck / out of budget: rnd = R_MAX_ROUNDS, the tail below)
      bool had_pass = false;  // (uniform) the last round had a wavefronts' pass: owners look their rows up again
      bool stalled = false;   // (uniform) a round decided nothing: the heavy rows are handed over at once
      bool hv[RPT];           // the row was handed over to the wavefronts' pass: its owner only waits for the answer
#pragma unroll
      for (int u = 0; u < RPT; u++) hv[u] = false;
      auto one_round = [&]() -> bool {
        int wund = 0;  // undecided ROWS of this wavefront (uniform).  (Threads with an undecided row, the count until the
                       // wavefronts' pass subtracted the ROWS it decides: a thread with two enlisted rows counted once,
                       // both decided by the pass took the count of a pile of near-duplicates to zero with rows left --
                       // two keeps missing in one call of ten)
        if (stamp_on && rnd < 8) stamps[16 + 4 * rnd] = __builtin_amdgcn_s_memtime();
        // What a round costs is LDS reads of scattered state bytes (bank conflicts: ~12 cycles per wave read, 16 waves
        // on one LDS) and the slowest wave in front of the barrier.  So: a wave whose rows are all decided reads
        // nothing; the best suppressor and the first 8 entries are read together (most rows have fewer than 8);
        // entries 8..31 in one batch, only in waves with an undecided row that long; the overflow entries 16 at a time.
#pragma unroll
        for (int u = 0; u < RPT; u++) {
          if (had_pass && state[u] == 0) {  // (a row the wavefronts' pass of the last round decided: its owner learns it here)
            const unsigned char s0 = st[rr[u]];
            if (s0) state[u] = s0;
          }
          const bool act = state[u] == 0 && !hv[u];  // (a handed-over row: looked at, and counted, by the pass)
          if (__ballot(act) == 0ULL) continue;
          if (rnd >= R_MAX_ROUNDS) { wund += __popcll(__ballot(act)); continue; }
          const unsigned r = (unsigned)rr[u];
          const int listed = act ? min(c[u], EL) : 0;
          unsigned char v0, v8[8];
          v0 = st[(act && ms[u] < 65535u) ? ms[u] : r];
          {
            const unsigned wv[4] = {t[u][0].x, t[u][0].y, t[u][0].z, t[u][0].w};
#pragma unroll
            for (int q = 0; q < 8; q++) v8[q] = st[q < listed ? ((wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu) : r];
          }
          bool aK = act && ms[u] < 65535u && v0 == 1, aR = true;  // its highest-scored suppressor first
          bool pending = false;                                   // (left to the wavefronts' pass below)
#pragma unroll
          for (int q = 0; q < 8; q++) {
            aK |= q < listed && v8[q] == 1;
            aR &= !(q < listed) || v8[q] == 2;
          }
          if (__ballot(act && !aK && listed > 8) != 0ULL) {
            unsigned char w[24];
#pragma unroll
            for (int c4 = 1; c4 < 4; c4++) {
              const unsigned wv[4] = {t[u][c4].x, t[u][c4].y, t[u][c4].z, t[u][c4].w};
#pragma unroll
              for (int q = 0; q < 8; q++)
                w[8 * (c4 - 1) + q] = st[8 * c4 + q < listed ? ((wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu) : r];
            }
#pragma unroll
            for (int q = 0; q < 24; q++) {
              aK |= 8 + q < listed && w[q] == 1;
              aR &= !(8 + q < listed) || w[q] == 2;
            }
            const bool ovf = act && !aK && c[u] > EL;
            if (__ballot(ovf) != 0ULL) {
              // HANDED OVER behind round R_HANDOVER (most heavy rows fall to what their lists show in the first rounds;
              // measured on the model's own pool: the NMS chain 93.0 us handing over from round 0, 90.9 from 1, 89.4
              // from 2, 88.4 from 3, 87.8 from 4; with the stall rule, one box: 89.2 / 88.0 / 88.4 from 2 / 3 / 4) or as soon as a
              // round decides nothing: from then on
              // the pass below looks at the row -- list and overflow mask -- and its owner only waits for the answer.
              // (The owner walking its 32 entries every round on top of the pass: 2-3 k clocks per round in the
              // wavefronts with heavy rows, the ones the others wait for.)  No room on the pass's list: the row stays
              // with its owner, who cannot keep it -- the rounds get stuck and the tail takes over.
              const bool want = ovf && (rnd > R_HANDOVER || stalled);
              const u64 need = __ballot(want);
              if (need) {
                int hb = 0;
                if (lane == 0) hb = atomicAdd(&hcnt[0], __popcll(need));
                hb = __builtin_amdgcn_readfirstlane(hb);
                const int he = hb + __popcll(need & ((1ULL << lane) - 1ULL));
                if (want && he < OVQ) {
                  ovq[he] = (unsigned short)r;
                  hv[u] = true;
                }
              }
              pending = ovf;  // (never decided by its owner: its list is not all of its suppressors)
            }
          }
          bool row_und = false;
          if (act) {
            if (!pending && (aK || aR)) {
              state[u] = aK ? 2 : 1;
              st[r] = (unsigned char)state[u];
            } else {
              row_und = !hv[u];  // (a row handed over in this round is counted by the pass)
            }
          }
          wund += __popcll(__ballot(row_und));
        }
        if (stamp_on && rnd < 8) stamps[17 + 4 * rnd] = __builtin_amdgcn_s_memtime();
        // undecided rows of the workgroup: wave popcounts through a double-buffered table, ONE barrier per round
        // (__syncthreads_count costs ~1900 cycles however little the waves do)
        int* wtab2 = wsum;  // (2 x 16 ints)
        if (lane == 0) wtab2[(rnd & 1) * 16 + wave] = wund;
        __syncthreads();
        int left = 0;
#pragma unroll
        for (int w = 0; w < RTHREADS / 64; w++) left += wtab2[(rnd & 1) * 16 + w];
        // The handed-over rows, once there are any (uniform): a wavefront per row -- lanes 0 .. 31 an entry of its list
        // each, every lane a word of its overflow mask (the K / R bits of the word's 64 rows rebuilt from their state
        // bytes) -- and two more barriers, in the rounds of such label groups only.
        const int nh = min(hcnt[0], OVQ);
        had_pass = nh > 0;
        if (nh > 0) {
          int und_h = 0;
          for (int e = wave; e < nh; e += RTHREADS / 64) {
            const int hr = ovq[e];
            if (st[hr] != 0) continue;  // (decided in an earlier round; uniform)
            const u64* row = maskT + (size_t)hr * cb;
            const int nw = (hr >> 6) + 1;
            const unsigned ent = lane < EL ? (unsigned)sd.elist[(size_t)hr * EL + lane] : 0u;  // (more than EL suppressors: the list is full)
            const u64 m0 = lane < nw ? row[lane] : 0ULL;
            const unsigned char lv = lane < EL ? st[ent] : (unsigned char)2;
            bool hK = lv == 1, hR = lv == 2;
            for (int q0 = 0; q0 < nw && __ballot(hK) == 0ULL; q0 += 64) {
              const int q = q0 + lane;
              const u64 mm = q0 == 0 ? m0 : (q < nw ? row[q] : 0ULL);
              if (mm) {  // (few words of a row carry bits: the others cost their lane nothing)
                const u64* sp = reinterpret_cast<const u64*>(st + (size_t)q * 64);
                u64 kw = 0, rw = 0;
#pragma unroll
                for (int b8 = 0; b8 < 8; b8++) {
                  const u64 x = sp[b8];
                  const u64 k1 = x & ~(x >> 1) & 0x0101010101010101ULL, r1 = (x >> 1) & ~x & 0x0101010101010101ULL;
                  kw |= ((k1 * 0x0102040810204080ULL) >> 56) << (8 * b8);
                  rw |= ((r1 * 0x0102040810204080ULL) >> 56) << (8 * b8);
                }
                hK |= (mm & kw) != 0ULL;
                hR &= (mm & ~rw) == 0ULL;
              }
            }
            const bool k_any = __ballot(hK) != 0ULL, r_all = __ballot(!hR) == 0ULL;
            if (lane == 0) {
              if (k_any || r_all) st[hr] = k_any ? 2 : 1;
              else und_h++;
            }
          }
          if (lane == 0) hdec[wave] = und_h;
          __syncthreads();
#pragma unroll
          for (int w = 0; w < RTHREADS / 64; w++) left += hdec[w];
          __syncthreads();  // (hdec is written again in the next round)
        }
        if (stamp_on && rnd < 8) stamps[18 + 4 * rnd] = __builtin_amdgcn_s_memtime();
        if (stamp_on && rnd < 8) stamps[19 + 4 * rnd] = __builtin_amdgcn_s_memtime();
        if (left == 0) return true;                                               // every row decided
        // stuck or out of budget: the tail below -- but a first round without a decision only means that what is left are
        // heavy rows not yet handed over (their owners cannot keep them): they go to the pass, and the rounds go on
        if (rnd >= R_MAX_ROUNDS || (left == left_before && stalled)) { rnd = R_MAX_ROUNDS; return true; }
        stalled = stalled || left == left_before;
        left_before = left;
        return false;
      };
      while (!one_round()) rnd++;
      // K / R as bits, from the state bytes of this workgroup's rows (8 bytes at a time: byte == 1 / == 2 -> one bit)
      for (int b = tid; b < cbn; b += RTHREADS) {
        const u64 own = Own[b];
        u64 kb = 0, rb = 0;
        if (own) {
          const u64* sp = reinterpret_cast<const u64*>(st + (size_t)b * 64);
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const u64 x = sp[q];
            const u64 k1 = x & ~(x >> 1) & 0x0101010101010101ULL, r1 = (x >> 1) & ~x & 0x0101010101010101ULL;
            kb |= ((k1 * 0x0102040810204080ULL) >> 56) << (8 * q);
            rb |= ((r1 * 0x0102040810204080ULL) >> 56) << (8 * q);
          }
        }
        Kb[b] = kb & own;
        Rb[b] = rb & own;
      }
      if (fbits && rnd < R_MAX_ROUNDS) {  // kept candidates as bits, from the registers (the tail's rows: below)
#pragma unroll
        for (int u = 0; u < RPT; u++) {
          if (state[u] == 0) state[u] = st[rr[u]];  // (decided by the wavefronts' pass of the last round)
          if (state[u] == 1) atomicOr(&fbits[sv[u] >> 6], 1ULL << (sv[u] & 63));
        }
        fbits_done = true;
      }
      __syncthreads();
      return rnd;
    };
    // (three rows per thread -- a 32 768-row pool has ~2 200 rows per label of 15 -- measured in round 6: 120 B of scratch
    // per lane and 99 us against the general form's 95)
    if (m <= RTHREADS) round = in_regs(std::integral_constant<int, 1>{});
    else round = in_regs(std::integral_constant<int, 2>{});
  } else {
  // (two forms of the general reducer, round 6: up to 3 rows per thread -- every grouped launch of a multi-class pool --
  // the worklist's rows stay in registers too; beyond, the register cache of 9 rows per thread needs those registers)
  auto general = [&](auto rc_tag, auto wl_tag) {
  constexpr int RC = decltype(rc_tag)::value;
  constexpr bool WL = decltype(wl_tag)::value;
  // prologue: counts and first chunks of this workgroup's rows (all loads in flight together)
  int cnt[RC];
  uint4 c0[RC];
#pragma unroll
  for (int u = 0; u < RC; u++) {
    const int k = tid + u * RTHREADS;
    const int rr = k < m ? row_of(k) : 0;
    cnt[u] = sd.ecnt[rr];
    c0[u] = *reinterpret_cast<const uint4*>(sd.elist + (size_t)rr * EL);
  }
  unsigned own = 0;  // cached rows with a long list that did not fit the worklist
#pragma unroll
  for (int u = 0; u < RC; u++) {
    const int k = tid + u * RTHREADS;
    const int r = k < m ? row_of(k) : 0;
    const bool kept0 = k < m && cnt[u] == 0;
    if (k < m) st[r] = kept0 ? 1 : 0;
    if (GROUPED) {
      if (kept0) set_bit(Kb, r);
    } else {
      const u64 k0 = __ballot(kept0);
      if (lane == 0 && k0) Kb[r >> 6] = k0;
    }
    const bool isbig = k < m && cnt[u] > 8;
    const u64 mb = __ballot(isbig);
    if (mb) {
      int base = 0;
      if (lane == 0) base = atomicAdd(s_nbig, __popcll(mb));
      base = __builtin_amdgcn_readfirstlane(base);
      const int slot = base + __popcll(mb & ((1ULL << lane) - 1ULL));
      if (isbig) {
        if (slot < R_BLIST) blist[slot] = (unsigned short)r;
        else own |= 1u << u;
      }
    }
  }
  for (int u = RC; wave * 64 + u * RTHREADS < m; u++) {  // rows beyond the register cache
    const int k = tid + u * RTHREADS;
    const int r = k < m ? row_of(k) : 0;
    const bool kept0 = k < m && sd.ecnt[r] == 0;
    if (k < m) st[r] = kept0 ? 1 : 0;
    if (kept0) set_bit(Kb, r);
  }
  __syncthreads();
  stamp(2);
  const int nbig = min(*s_nbig, R_BLIST);

  // one row against the K / R sets; returns 0 undecided, 1 kept, 2 removed (-1: enlisted for the overflow pass)
  auto long_row_core = [&](const int r, const unsigned ms, const int c, const uint4 (&t)[4]) -> int {
    if (ms < 65535u && st[ms] == 1) return 2;  // its highest-scored suppressor first
    const int listed = min(c, EL);
    bool anyK = false, allR = true;
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned wv[4] = {t[c4].x, t[c4].y, t[c4].z, t[c4].w};
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const unsigned i = (wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
        const bool on = 8 * c4 + q < listed;
        const unsigned char v = st[on ? i : (unsigned)r];
        anyK |= on && v == 1;
        allR &= !on || v == 2;
      }
    }
    if (c > EL && !anyK) {
      // Suppressors beyond the list: the row's overflow mask, words 0 .. r / 64.  Enlisted for the wavefronts (lanes <->
      // words, behind the passes below); one thread walking a row of a 32 768-row pool alone -- 512 words, 8 loads per
      // step -- was 20-30 us of every round there (round 6).  No room on the list: the walk.
      const int e = atomicAdd(s_m, 1);
      if (e < OVQ) {
        ovq[e] = (unsigned short)r;
        ovl[e] = allR ? 1 : 0;
        return -1;
      }
      const u64* row = maskT + (size_t)r * cb;
      const int w = r >> 6;
      for (int q0 = 0; q0 <= w && !anyK; q0 += 8) {
        u64 mm[8];
#pragma unroll
        for (int e = 0; e < 8; e++) mm[e] = row[min(q0 + e, w)];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int q = min(q0 + e, w);
          anyK |= (mm[e] & Kb[q]) != 0ULL;
          allR &= (mm[e] & ~Rb[q]) == 0ULL;
        }
      }
    }
    return anyK ? 2 : (allR ? 1 : 0);
  };
  auto long_row = [&](const int r) -> int {
    const unsigned ms = 65535u - (unsigned)sd.msup[r];
    if (ms < 65535u && st[ms] == 1) return 2;
    const int c = sd.ecnt[r];
    const uint4* lp = reinterpret_cast<const uint4*>(sd.elist + (size_t)r * EL);
    const uint4 t[4] = {lp[0], lp[1], lp[2], lp[3]};
    return long_row_core(r, ms, c, t);
  };
  // The worklist's first 1024 rows -- all of it in a grouped launch -- keep suppressor, count and list in registers for
  // every round (round 6: two dependent trips to memory per long row and round were most of a round at 32 768 rows,
  // 450 long rows per label group there)
  int wl_r = 0, wl_c = 0;
  unsigned wl_ms = 65535u;
  uint4 wl_t[4] = {};
  if (WL && tid < nbig) {
    wl_r = blist[tid];
    wl_ms = 65535u - (unsigned)sd.msup[wl_r];
    wl_c = sd.ecnt[wl_r];
    const uint4* lp = reinterpret_cast<const uint4*>(sd.elist + (size_t)wl_r * EL);
    wl_t[0] = lp[0]; wl_t[1] = lp[1]; wl_t[2] = lp[2]; wl_t[3] = lp[3];
  }
  auto decide = [&](const int r, const int d) {
    st[r] = (unsigned char)d;
    set_bit(d == 1 ? Kb : Rb, r);
  };

  for (; round < R_MAX_ROUNDS; round++) {
    if (tid == 0) {
      *s_und = 0;
      *s_m = 0;
    }
    __syncthreads();
    int und = 0;
    if (stamp_on && round < 8) stamps[16 + 4 * round] = __builtin_amdgcn_s_memtime();
    // pass A: the cached rows with at most 8 suppressors, straight from registers
#pragma unroll
    for (int u = 0; u < RC; u++) {
      const int k = tid + u * RTHREADS;
      const int r = k < m ? row_of(k) : 0;
      const bool act = k < m && cnt[u] <= 8 && cnt[u] > 0 && st[r] == 0;
      if (__ballot(act) == 0ULL) continue;
      bool anyK = false, allR = true;
      if (act) {
        unsigned wv[4] = {c0[u].x, c0[u].y, c0[u].z, c0[u].w};
        asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]));  // (see nms_reduce_rounds_kernel)
        unsigned char v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const unsigned i = (wv[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
          v[q] = st[q < cnt[u] ? i : (unsigned)r];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const bool on = q < cnt[u];
          anyK |= on && v[q] == 1;
          allR &= !on || v[q] == 2;
        }
      }
      const bool toR = act && anyK, toK = act && !anyK && allR;
      und += act && !anyK && !allR;
      if (GROUPED) {
        if (toR | toK) decide(r, toR ? 2 : 1);
      } else {
        if (toR | toK) st[r] = toR ? 2 : 1;
        const u64 mr = __ballot(toR), mk = __ballot(toK);
        if (lane == 0) {
          if (mr) Rb[r >> 6] |= mr;
          if (mk) Kb[r >> 6] |= mk;
        }
      }
    }
    if (stamp_on && round < 8) stamps[17 + 4 * round] = __builtin_amdgcn_s_memtime();
    // pass B: the worklist of long rows, one per thread
    if (WL && tid < nbig && st[wl_r] == 0) {
      const int d = long_row_core(wl_r, wl_ms, wl_c, wl_t);
      if (d > 0) decide(wl_r, d);
      else if (d == 0) und++;
    }
    for (int k = tid + (WL ? RTHREADS : 0); k < nbig; k += RTHREADS) {
      const int r = blist[k];
      if (st[r] != 0) continue;
      const int d = long_row(r);
      if (d > 0) decide(r, d);
      else if (d == 0) und++;
    }
    // pass C: cached long rows that did not fit the worklist, and rows beyond the register cache
    if (own || m > RC * RTHREADS) {
      for (int u = 0; wave * 64 + u * RTHREADS < m; u++) {
        const int k = tid + u * RTHREADS;
        const bool mine = u < RC ? ((own >> u) & 1u) : true;
        if (!(k < m && mine)) continue;
        const int r = row_of(k);
        if (st[r] != 0) continue;
        if (u >= RC && sd.ecnt[r] == 0) continue;
        const int d = long_row(r);
        if (d > 0) decide(r, d);
        else if (d == 0) und++;
      }
    }
    // the enlisted overflow rows: a wavefront per row, a lane per word
    if (stamp_on && round < 8) stamps[18 + 4 * round] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    {
      const int nov = min(*s_m, OVQ);
      for (int e = wave; e < nov; e += RTHREADS / 64) {
        const int r = ovq[e];
        const u64* row = maskT + (size_t)r * cb;
        const int w = r >> 6;
        bool anyK = false, allR = ovl[e] != 0;
        for (int q0 = 0; q0 <= w; q0 += 256) {  // (four independent loads per lane and step)
          u64 mm[4];
#pragma unroll
          for (int x = 0; x < 4; x++) mm[x] = q0 + 64 * x + lane <= w ? row[q0 + 64 * x + lane] : 0ULL;
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const int q = min(q0 + 64 * x + lane, w);
            anyK |= (mm[x] & Kb[q]) != 0ULL;
            allR &= (mm[x] & ~Rb[q]) == 0ULL;
          }
        }
        const bool k_any = __ballot(anyK) != 0ULL, r_all = __ballot(!allR) == 0ULL;
        if (lane == 0) {
          if (k_any) decide(r, 2);
          else if (r_all) decide(r, 1);
          else und++;
        }
      }
    }
    if (und) atomicAdd(s_und, und);
    __syncthreads();
    if (stamp_on && round < 8) stamps[19 + 4 * round] = __builtin_amdgcn_s_memtime();
    const int left = *s_und;
    __syncthreads();
    if (left == 0) break;
  }
  };
  if (m <= 3 * RTHREADS) general(std::integral_constant<int, 3>{}, std::true_type{});
  else general(std::integral_constant<int, R_CACHE>{}, std::false_type{});
  }
  stamp(3);
  if (stamp_on) stamps[7] = (u64)round;
  if (round == R_MAX_ROUNDS && tid < 64) {
    // a suppression chain longer than the round budget: wave 0 finishes this workgroup's rows in score order
    for (int b = 0; b < cbn; b++) {
      const int nvalid = min(TILE, n - b * TILE);
      const u64 valid = nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1ULL);
      u64 todo = (GROUPED ? Own[b] : valid) & ~(Kb[b] | Rb[b]);
      while (todo) {
        const int k = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int r = b * TILE + k;
        const int c = sd.ecnt[r];
        bool hit = false;
        if (lane < min(c, EL)) {
          const unsigned i = sd.elist[(size_t)r * EL + lane];
          hit = (Kb[i >> 6] >> (i & 63)) & 1ULL;
        }
        if (c > EL) {
          const u64* row = maskT + (size_t)r * cb;
          for (int q = lane; q <= b; q += 64) hit |= (row[q] & Kb[q]) != 0ULL;
        }
        const bool removed = __ballot(hit) != 0ULL;
        if (lane == 0) {
          if (removed) Rb[b] |= 1ULL << k;
          else Kb[b] |= 1ULL << k;
          st[r] = removed ? 2 : 1;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      }
    }
  }
  __syncthreads();
  stamp(4);
  // the kept rows of this workgroup, as bits
  for (int b = tid; b < cbn; b += RTHREADS) {
    const u64 kb = Kb[b];
    if (GROUPED) {
      if (kb) atomicOr(&kbits[b], kb);  // (a word holds rows of several labels)
    } else {
      kbits[b] = kb;
    }
  }
  // ... and over the CANDIDATE indices (the index-ordered finish reads these: no translation pass there)
  if (fbits && !fbits_done) {
    for (int k = tid; k < m; k += RTHREADS) {
      const int r = row_of(k);
      if (st[r] == 1) {
        const int c = svals[r];
        atomicOr(&fbits[c >> 6], 1ULL << (c & 63));
      }
    }
  }
  stamp(5);
}

__global__ __launch_bounds__(RTHREADS) void nms_reduce_groups_kernel(const u64* __restrict__ maskT,
                                                                     u64* __restrict__ side, int cb,
                                                                     const unsigned* __restrict__ counter,
                                                                     u64* __restrict__ kbits, size_t kbits_stride,
                                                                     const int* __restrict__ svals,
                                                                     u64* __restrict__ fbits, Batch bt,
                                                                     u64* __restrict__ stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  __shared__ int s_und, s_nbig, s_m;
  __shared__ int wsum[2 * RTHREADS / 64];
  const int img = blockIdx.z;
  const int n = bt.counts[img];
  const bool grouped = gridDim.x > 1 && bt.gbits && counter[img * bt.counter + Q_XFLAG] == 0u;
  if (!grouped && blockIdx.x != 0) return;
  const Side sd = side_tables(side + img * bt.nz, bt.rows);
  if (grouped)
    reduce_groups_body<true>(maskT + img * bt.mask, sd, n, (int)bt.rows, cb,
                             bt.gbits + ((size_t)img * RG_GROUPS + blockIdx.x) * cb, (int)blockIdx.x,
                             kbits + img * kbits_stride, svals + (size_t)img * bt.rows,
                             fbits ? fbits + img * kbits_stride : nullptr, smem8, &s_und, &s_nbig, &s_m, wsum, stamps);
  else
    reduce_groups_body<false>(maskT + img * bt.mask, sd, n, (int)bt.rows, cb, nullptr, 0, kbits + img * kbits_stride,
                              svals + (size_t)img * bt.rows, fbits ? fbits + img * kbits_stride : nullptr, smem8, &s_und,
                              &s_nbig, &s_m, wsum, stamps);
}

// ---------------------------------------------------------------------------- reduce by a walk in score order (round 5)
// The dependency-round reducer above pays a workgroup barrier and a pass over its rows per ROUND; on the detector's own
// pool (boxes decoded from neighbouring positions, iou_thr 0.1: suppression chains of 8-10 links) that was 65 us of the
// step's 140 us of NMS with 64 workgroups of 1024 threads mostly waiting (profiles/r05_bench_kernel_stats.txt).  Here
// ONE WAVEFRONT owns a label group of an image (the groups share no suppressor edge -- the drain raises Q_XFLAG
// otherwise and group 0's wavefront then walks the whole image) and walks its rows in score order, 64 at a time:
//   * the group's rows are compacted once (ascending = score order) with their position in the group (posmap);
//   * a block of 64 rows: every lane stages its row's suppressor list (64 B) in LDS and sorts the entries in two --
//     suppressors in EARLIER blocks are final (one LDS bit test each: any kept => removed), suppressors in the SAME
//     block become a 64-bit lane mask D;
//   * the block resolves with ballots alone: a lane is removed once a lane of D is kept, kept once every lane of D is
//     decided and none kept; the lowest undecided lane always decides, so the loop ends -- whatever the chain depth,
//     an iteration is a handful of scalar instructions, not a barrier;
//   * rows with more than EL suppressors also scan their overflow row of the transposed mask.
// Greedy NMS in score order has ONE answer (row kept <=> no kept suppressor), so the result is the rounds reducer's.
constexpr int WALK_MAXN = 12288;  // rows per image (static LDS: 2 x u16 per row + the kept bits = 51 KB)

__global__ __launch_bounds__(64) void nms_reduce_walk_kernel(const u64* __restrict__ maskT, u64* __restrict__ side, int cb,
                                                             const unsigned* __restrict__ counter,
                                                             u64* __restrict__ kbits, size_t kbits_stride,
                                                             const int* __restrict__ svals, u64* __restrict__ fbits,
                                                             Batch bt) {
  __shared__ u64 Kb[WALK_MAXN / 64];
  __shared__ unsigned short posmap[WALK_MAXN];
  __shared__ unsigned short rows_l[WALK_MAXN];
  const int img = blockIdx.z, group = blockIdx.x, lane = threadIdx.x;
  const int n = bt.counts[img];
  const bool grouped = gridDim.x > 1 && bt.rlab && counter[img * bt.counter + Q_XFLAG] == 0u;
  if (!grouped && group != 0) return;
  if (n <= 0) return;
  maskT += img * bt.mask;
  kbits += img * kbits_stride;
  svals += (size_t)img * bt.rows;
  if (fbits) fbits += img * kbits_stride;
  const Side sd = side_tables(side + img * bt.nz, bt.rows);
  const uint8_t* rlab = grouped ? bt.rlab + (size_t)img * bt.rows : nullptr;
  const int cbn = (n + TILE - 1) / TILE;
  for (int w = lane; w < cbn; w += 64) Kb[w] = 0;
  // ---- the group's rows, ascending, and each row's position among them
  int m = 0;
  if (grouped) {
    // every label of the image is requested at once -- a lane takes four consecutive rows of every 256 -- and only then
    // looked at (a loop with its loads inside is one memory round trip per 256 rows: 34 us at n = 8576)
    constexpr int KMAX = WALK_MAXN / 256;
    unsigned lb[KMAX];
#pragma unroll
    for (int u = 0; u < KMAX; u++) {
      const int r4 = u * 256 + 4 * lane;
      lb[u] = r4 < n ? *reinterpret_cast<const unsigned*>(rlab + r4) : 0xffffffffu;  // (rows are padded to 64: in bounds)
    }
#pragma unroll
    for (int u = 0; u < KMAX; u++) {
      if (u * 256 >= n) break;
      const int r4 = u * 256 + 4 * lane;
      bool mine[4];
      int cl4 = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        mine[k] = (r4 + k < n) && (int)((lb[u] >> (8 * k)) & (RG_GROUPS - 1)) == group;
        cl4 += mine[k] ? 1 : 0;
      }
      int incl = cl4;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      int pos = m + incl - cl4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (mine[k]) {
          rows_l[pos] = (unsigned short)(r4 + k);
          posmap[r4 + k] = (unsigned short)pos;
          pos++;
        }
      }
      m += __shfl(incl, 63);
    }
  } else {
    m = n;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS only: a fence would also wait for the NEXT block's loads)
  __builtin_amdgcn_wave_barrier();
  // A block's loads -- the row's count, its best suppressor, its 64-byte list -- depend on the row only: the NEXT block's
  // are issued before this block is resolved, so a block costs the resolution, not three memory round trips.
  struct Blk {
    int r, c, i0;
    bool valid;
    uint4 l[4];
  };
  auto load_blk = [&](const int t0) {
    Blk b;
    const int k = t0 + lane;
    b.valid = k < m;
    b.r = b.valid ? (grouped ? (int)rows_l[k] : k) : 0;
    b.c = 0;
    b.i0 = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) b.l[q] = make_uint4(0u, 0u, 0u, 0u);
    if (b.valid) {
      b.c = sd.ecnt[b.r];
      b.i0 = 65535 - sd.msup[b.r];
      const uint4* src = reinterpret_cast<const uint4*>(sd.elist + (size_t)b.r * EL);
#pragma unroll
      for (int q = 0; q < 4; q++) b.l[q] = src[q];  // (a row's 64 bytes: entries beyond its count are stale, never read)
    }
    return b;
  };
  Blk nxt = load_blk(0);
  for (int t0 = 0; t0 < m; t0 += 64) {
    const Blk cur = nxt;
    if (t0 + 64 < m) nxt = load_blk(t0 + 64);
    const bool valid = cur.valid;
    const int r = cur.r;
    const int firstrow = grouped ? (int)rows_l[t0] : t0;  // suppressors below it are final
    int c = cur.c;
    bool rem = false;
    if (valid && c > 0) {
      // the highest-scored suppressor first: it is very often the kept head of the row's cluster, and a row it
      // removes needs neither its list nor its overflow row
      if (cur.i0 < firstrow) rem = (Kb[cur.i0 >> 6] >> (cur.i0 & 63)) & 1ULL;
      if (rem) c = 0;
    }
    const int cl = min(c, EL);
    int maxc = cl;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxc = max(maxc, __shfl_xor(maxc, d));
    u64 D = 0;
    // the list from registers, eight entries (one 16-byte quad) per wave-uniform step: the LDS lookups of a step do not
    // depend on each other
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (maxc <= q * 8) break;
      const unsigned w4[4] = {cur.l[q].x, cur.l[q].y, cur.l[q].z, cur.l[q].w};
#pragma unroll
      for (int h = 0; h < 8; h++) {
        const int e = q * 8 + h;
        const int i = min((int)((h & 1) ? (w4[h >> 1] >> 16) : (w4[h >> 1] & 0xffffu)), n - 1);
        const bool early = (Kb[i >> 6] >> (i & 63)) & 1ULL;
        const int pl = (grouped ? (int)posmap[i] : i) - t0;
        if (e < cl) {
          if (i < firstrow) rem |= early;
          else D |= 1ULL << (pl & 63);
        }
      }
    }
    // the suppressors beyond the list (rows of dense clusters the shortcut did not settle): row r of the transposed
    // mask (bits i < r), one row at a time, the wavefront's lanes a word each
    u64 ovm = __ballot(valid && !rem && c > EL);
    while (ovm) {
      const int srcl = __builtin_ctzll(ovm);
      ovm &= ovm - 1;
      const int rs = __shfl(r, srcl);
      const u64* row = maskT + (size_t)rs * cb;
      const int fw = firstrow >> 6;
      bool hit = false;
      u64 dpart = 0;
      for (int q = lane; q <= (rs >> 6); q += 64) {
        u64 w = row[q];
        if (q < fw) {
          hit |= (w & Kb[q]) != 0ULL;
        } else {
          while (w) {
            const int i = q * 64 + __builtin_ctzll(w);
            w &= w - 1;
            if (i < firstrow) hit |= (Kb[q] >> (i & 63)) & 1ULL;
            else if (i < rs) dpart |= 1ULL << ((grouped ? (int)posmap[i] : i) - t0);
          }
        }
      }
      const bool anyhit = __ballot(hit) != 0ULL;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(dpart & 0xffffffffULL), d), hi = __shfl_xor((unsigned)(dpart >> 32), d);
        dpart |= ((u64)hi << 32) | lo;
      }
      if (lane == srcl) {
        rem |= anyhit;
        D |= dpart;
      }
    }
    // ---- the block resolves with ballots
    u64 und = __ballot(valid && !rem), Kc = 0;
    while (und) {
      const bool mineu = (und >> lane) & 1ULL;
      const u64 nk = __ballot(mineu && !(D & Kc) && !(D & und));
      const u64 nr = __ballot(mineu && (D & Kc));
      Kc |= nk;
      und &= ~(nk | nr);
    }
    if ((Kc >> lane) & 1ULL) {
      atomicOr(&Kb[r >> 6], 1ULL << (r & 63));
      if (fbits) {
        const int cand = svals[r];
        atomicOr(&fbits[cand >> 6], 1ULL << (cand & 63));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS only: a fence would also wait for the NEXT block's loads)
    __builtin_amdgcn_wave_barrier();
  }
  for (int w = lane; w < cbn; w += 64) {
    const u64 kb = Kb[w];
    if (kb) atomicOr(&kbits[w], kb);  // (a word holds rows of several label groups)
  }
}

// rnms returns keep sorted by original index (rnms_kernel.cu:331-334): mark kept originals,
// then an ordered compaction by one workgroup.
__global__ __launch_bounds__(1024) void nms_ascending_kernel(int n, uint8_t* __restrict__ flags,
                                                             int64_t* __restrict__ keep_out,
                                                             const int32_t* __restrict__ count) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int cnt = *count;
  for (int i = tid; i < n; i += blockDim.x) flags[i] = 0;
  __syncthreads();
  for (int i = tid; i < cnt; i += blockDim.x) flags[keep_out[i]] = 1;
  __syncthreads();
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int c = 0;
  for (int i = lo; i < hi; i++) c += flags[i];
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - c;
  for (int i = lo; i < hi; i++)
    if (flags[i]) keep_out[pos++] = i;
}

struct Layout {
  BoxRec* recs;
  u64* mask;
  u64* nz;
  unsigned* counter;
  unsigned* gqueue;
  unsigned* redo;
  unsigned qcap;
  uint8_t* flags;
  int cb;
};

inline size_t queue_entries(int n) {
  // all regions together: generous for real pools (tens of candidate pairs per box); denser inputs go through
  // the redo list.  A region holds queue_entries / Q_NREG entries.
  size_t tri = (size_t)n * (size_t)(n - 1) / 2;
  size_t cap = (size_t)n * 64 + 65536;
  if (tri < cap) cap = tri < (size_t)Q_NREG * 64 ? (size_t)Q_NREG * 64 : tri;
  return cap / Q_NREG * Q_NREG;
}

inline size_t layout(int n, void* ws, Layout* L) {
  const size_t cb = (n + TILE - 1) / TILE;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* recs = take((size_t)n * sizeof(BoxRec));
  char* mask = take((size_t)n * cb * sizeof(u64));  // mask and nz are zeroed together
  char* nz = take(side_words((size_t)n) * sizeof(u64));  // side tables (zeroed with the mask)
  char* counter = take((size_t)Q_CTL_WORDS * 4);
  char* gq = take(queue_entries(n) * sizeof(unsigned));
  char* rd = take(cb * cb * sizeof(unsigned));  // redo-tile list
  char* flags = take((size_t)n);
  if (L) {
    L->recs = (BoxRec*)recs; L->mask = (u64*)mask; L->nz = (u64*)nz; L->counter = (unsigned*)counter;
    L->gqueue = (unsigned*)gq; L->redo = (unsigned*)rd; L->qcap = (unsigned)(queue_entries(n) / Q_NREG); L->flags = (uint8_t*)flags;
    // test hook: a tiny capacity forces the redo path of the stream / drain kernels
    const int qcap_o = g_r3_nms_qcap;  // (one read)
    if (qcap_o > 0 && (unsigned)qcap_o < L->qcap) L->qcap = (unsigned)qcap_o;
    L->cb = (int)cb;
  }
  return off + 256;
}

// (chip_wgs: what the chip holds at once -- 2048 for the LDS-list clips (16 KB of LDS per workgroup, 8 per CU); the
// straight-line v1 clip takes 4 waves per SIMD of registers: 4 workgroups per CU)
inline int drain_blocks(size_t qcap, int chip_wgs = 2048) {
  size_t blocks = (qcap + 255) / 256;
  return blocks > (size_t)chip_wgs ? chip_wgs : blocks < 1 ? 1 : (int)blocks;
}

// greedy reduction of `images` problems (blockIdx.z)
inline void launch_reduce(int images, const u64* mask, const u64* side, int n, int cb, const int64_t* order,
                          int64_t* keep_out, int32_t* count_out, const Batch& bt, hipStream_t stream) {
  const size_t lds = reduce_lds_bytes(n > 0 ? n : (int)bt.rows, cb);  // (batched: n = 0, row capacity in bt)
  static R3DeviceOnce raised;  // the default cap on dynamic LDS is 64 KB; the opt-in is per device
  if (lds > 64 * 1024 && raised.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_reduce_rounds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipLaunchKernelGGL(nms_reduce_rounds_kernel, dim3(1, 1, images), dim3(RTHREADS), lds, stream, mask, side, n, cb,
                     order, keep_out, count_out, bt);
}

template <int GEOM, bool LABEL>
int run_nms(const float* dets, int det_stride, const int64_t* labels, const int64_t* order, int n,
            float thr, const Layout& L, int64_t* keep_out, int32_t* count_out, hipStream_t stream) {
  const int cb = L.cb;
  const bool clip_fast = g_r3_clip_impl == 0;  // (one read per call)
  const bool tiles = (g_r3_nms_impl == 1) || !(thr >= 0.f) || n >= 65536;
  hipLaunchKernelGGL((nms_prepare_kernel<GEOM>), dim3((n + 255) / 256), dim3(256), 0, stream, dets,
                     det_stride, labels, order, n, L.recs, tiles ? nullptr : L.counter);
  dim3 grid((cb + MASK_WAVES - 1) / MASK_WAVES, cb);
  if (tiles) {
    hipLaunchKernelGGL((nms_mask_kernel<GEOM, LABEL>), grid, dim3(NT), 0, stream, L.recs, n, cb, thr, L.mask);
    size_t lds = (size_t)(cb + 1) * sizeof(u64);
    if (lds > 64 * 1024) return -1;  // n > ~524 k boxes: not a rotated-NMS workload
    hipLaunchKernelGGL(nms_reduce_dense_kernel, dim3(1), dim3(1024), lds, stream, L.mask, n, cb, order,
                       keep_out, count_out);
    return 0;
  }
  // mask + nz are adjacent: one fill
  size_t zbytes = (size_t)((char*)L.counter - (char*)L.mask);
  if (r3k_zero_async(L.mask, zbytes, stream) != 0) return -2;
  hipLaunchKernelGGL((nms_stream_kernel<GEOM, LABEL>), grid, dim3(NT), 0, stream, L.recs, n, cb, L.gqueue, L.qcap,
                     L.counter, L.redo, single_problem(n));
  if (clip_fast)
    hipLaunchKernelGGL((nms_drain_kernel<GEOM, LABEL, true>), dim3(drain_blocks((size_t)L.qcap * Q_NREG, 4 * r3_cu_count())), dim3(256), 0, stream, L.recs, n, cb,
                       thr, L.gqueue, L.qcap, L.counter, L.redo, L.mask, L.nz, single_problem(n));
  else
    hipLaunchKernelGGL((nms_drain_kernel<GEOM, LABEL, false>), dim3(drain_blocks((size_t)L.qcap * Q_NREG)), dim3(256), 0, stream, L.recs, n, cb,
                       thr, L.gqueue, L.qcap, L.counter, L.redo, L.mask, L.nz, single_problem(n));
  launch_reduce(1, L.mask, L.nz, n, cb, order, keep_out, count_out, single_problem(n), stream);
  return 0;
}

// ------------------------------------------------------------------ batched detection pipeline
// multiclass_nms_rotated (core/post_processing/bbox_nms_rotated.py:7-131) + batched_rnms
// (ops/rnms/rnms_wrapper.py:34-69) for ALL images of a step in one pass of launches:
//   select  : scores > score_thr, candidates in row-major (anchor, class) order [= boolean-mask
//             indexing / nonzero()], per-image count and max over the candidate boxes' columns;
//   (host reads the B counts: the mask workspace is sized by the largest image)
//   sort    : rank by counting, descending and stable (= torch.sort(stable=True)); mc_rank_kernel;
//   prepare : x, y += label * (max + 1) in fp32 exactly as the wrapper does, then the v1 record;
//   stream / drain / reduce : the kernels above with blockIdx.z = image;
//   finish  : rnms returns keep ascending (rnms_kernel.cu:331-334) and the caller keeps the first
//             max_num of THAT order: flag kept candidates, ordered compaction, gather
//             [box, score] and label of the survivors.
constexpr int SEL_T = 1024;  // rows per workgroup of the select kernels

__device__ __forceinline__ int block_exclusive_scan(int v, int* wsum, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < SEL_T / 64; w++) {
    const int t = wsum[w];
    if (w < wave) woff += t;
    total += t;
  }
  return woff + incl - v;
}

// scores above the threshold in one row of K class scores (+ the background column).  Rows of K + 1 = 16 floats
// (15 DOTA classes) are 64 aligned bytes: four 16-byte loads instead of 15 scalar ones at a 64-byte lane stride.
__device__ __forceinline__ int count_above(const float* __restrict__ s, int K, float thr) {
  int cnt = 0;
  if (K == 15 && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
    // (DOTA: the row's four loads in flight together -- the loop below waits for each of its loads in turn, four L2
    // round trips per row: that was most of mc_count_kernel's 6.5 us)
    const float4* s4 = reinterpret_cast<const float4*>(s);
    const float4 a = s4[0], b = s4[1], c = s4[2], d = s4[3];
    return (a.x > thr) + (a.y > thr) + (a.z > thr) + (a.w > thr) + (b.x > thr) + (b.y > thr) + (b.z > thr) + (b.w > thr) +
           (c.x > thr) + (c.y > thr) + (c.z > thr) + (c.w > thr) + (d.x > thr) + (d.y > thr) + (d.z > thr);
  }
  if (((K + 1) & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s);
    const int nq = (K + 1) >> 2;
    for (int q = 0; q < nq; q++) {
      const float4 v = s4[q];
      const int k = q * 4;
      cnt += (v.x > thr) + (v.y > thr) + (v.z > thr) + ((k + 3 < K) & (v.w > thr));
    }
  } else {
    for (int k = 0; k < K; k++) cnt += s[k] > thr;
  }
  return cnt;
}

// select, pass 1: candidates per 1024-row part of an image (+ max over their boxes' columns)
__global__ __launch_bounds__(SEL_T) void mc_count_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores, int n, int K,
                                                         float thr, int parts, int* __restrict__ part_cnt,
                                                         float* __restrict__ part_max) {
  __shared__ int wsum[SEL_T / 64];
  __shared__ float wmax[SEL_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.y, row = blockIdx.x * SEL_T + tid;
  int cnt = 0;
  float mx = -INFINITY;
  if (row < n) {
    const float* s = scores + ((size_t)img * n + row) * (K + 1);  // last column = background
    cnt = count_above(s, K, thr);
    if (cnt) {
      const float* b = boxes + ((size_t)img * n + row) * 5;
      mx = fmaxf(fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])), b[4]);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    cnt += __shfl_xor(cnt, d);
    mx = fmaxf(mx, __shfl_xor(mx, d));
  }
  if (lane == 0) {
    wsum[wave] = cnt;
    wmax[wave] = mx;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SEL_T / 64; w++) {
      cnt += wsum[w];
      mx = fmaxf(mx, wmax[w]);
    }
    part_cnt[img * parts + blockIdx.x] = cnt;
    part_max[img * parts + blockIdx.x] = mx;
  }
}

// select, pass 2: ordered compaction (rows ascending, classes ascending inside a row)
__global__ __launch_bounds__(SEL_T) void mc_write_kernel(const float* __restrict__ scores, int n, int K, float thr,
                                                         int parts, const int* __restrict__ part_cnt,
                                                         const float* __restrict__ part_max, int cand_stride,
                                                         int* __restrict__ cand_row, int* __restrict__ cand_label,
                                                         float* __restrict__ cand_score, int* __restrict__ cand_rank,
                                                         int* __restrict__ counts, float* __restrict__ maxc) {
  __shared__ int wsum[SEL_T / 64];
  const int tid = threadIdx.x;
  const int img = blockIdx.y, row = blockIdx.x * SEL_T + tid;
  int base = 0;
  for (int p = 0; p < (int)blockIdx.x; p++) base += part_cnt[img * parts + p];
  if (blockIdx.x == 0 && tid == 0) {
    int total = 0;
    float mx = -INFINITY;
    for (int p = 0; p < parts; p++) {
      total += part_cnt[img * parts + p];
      mx = fmaxf(mx, part_max[img * parts + p]);
    }
    counts[img] = total;
    maxc[img] = mx;
  }
  const float* s = scores + ((size_t)img * n + row) * (K + 1);
  int cnt = 0;
  if (row < n) cnt = count_above(s, K, thr);
  int total;
  int pos = base + block_exclusive_scan(cnt, wsum, total);
  if (cnt) {
    const size_t cb = (size_t)img * cand_stride;
    for (int k = 0; k < K; k++) {
      const float v = s[k];
      if (v > thr) {
        cand_row[cb + pos] = row;
        cand_label[cb + pos] = k;
        cand_score[cb + pos] = v;
        cand_rank[cb + pos] = 0;
        pos++;
      }
    }
  }
}

// select in ONE launch (round 5; mc_count_kernel + mc_write_kernel above remain for the A/B, option nms_impl 5): a part's
// workgroup counts the candidates of the parts in front of it itself -- one row per thread and earlier part, all loads
// independent -- instead of waiting for a count launch (4.9 us + the gap behind it); the LAST part, which looks at every
// other part anyway, writes the image's count and box maximum.  The part's own rows are read once: the write loop works on the
// registers the count came from (it used to re-read the K scores one dependent load after the other).
__global__ __launch_bounds__(SEL_T) void mc_select_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          int n, int K, float thr, int parts, int cand_stride,
                                                          int* __restrict__ cand_row, int* __restrict__ cand_label,
                                                          float* __restrict__ cand_score, int* __restrict__ cand_rank,
                                                          int* __restrict__ counts, float* __restrict__ maxc) {
  __shared__ int wsum[SEL_T / 64];
  __shared__ int wbef[SEL_T / 64];
  __shared__ float wmax[SEL_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.y, part = blockIdx.x, row = part * SEL_T + tid;
  const bool last = part == (int)gridDim.x - 1;  // (it looks at every other part anyway: the image's count and box maximum are its job)
  auto box_max = [&](const int r) {
    const float* b = boxes + ((size_t)img * n + r) * 5;
    return fmaxf(fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])), b[4]);
  };
  // the part's own row: 16 floats in registers when a row is 64 aligned bytes (15 classes + background)
  const float* s = scores + ((size_t)img * n + row) * (K + 1);
  const bool quad = K == 15 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
  float4 v4[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                  make_float4(0.f, 0.f, 0.f, 0.f)};
  if (quad && row < n) {
#pragma unroll
    for (int q = 0; q < 4; q++) v4[q] = reinterpret_cast<const float4*>(s)[q];
  }
  int before = 0;
  float mx = -INFINITY;
  const int q0 = 0, q1 = part;
  auto count16 = [&](const float4* w) {
    return (w[0].x > thr) + (w[0].y > thr) + (w[0].z > thr) + (w[0].w > thr) + (w[1].x > thr) + (w[1].y > thr) +
           (w[1].z > thr) + (w[1].w > thr) + (w[2].x > thr) + (w[2].y > thr) + (w[2].z > thr) + (w[2].w > thr) +
           (w[3].x > thr) + (w[3].y > thr) + (w[3].z > thr);
  };
  if (quad) {
    // four earlier rows per step, their sixteen loads requested together at clamped
    // addresses: no branch between a load and the next (with one, every row waited for its own round trip)
    for (int qb = q0; qb < q1; qb += 4) {
      float4 w[4][4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int r = (qb + u) * SEL_T + tid;
        ok[u] = qb + u < q1 && r < n;
        const int rc = r < n ? r : n - 1;
        const float4* p4 = reinterpret_cast<const float4*>(scores + ((size_t)img * n + rc) * 16);
#pragma unroll
        for (int k = 0; k < 4; k++) w[u][k] = p4[k];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = ok[u] ? count16(w[u]) : 0;
        before += c;
        if (last && c) mx = fmaxf(mx, box_max((qb + u) * SEL_T + tid));  // (candidate rows only: a box is 20 B at a 20 B stride)
      }
    }
  } else {
    for (int q = q0; q < q1; q++) {
      const int r = q * SEL_T + tid;
      if (r < n) {
        const int c = count_above(scores + ((size_t)img * n + r) * (K + 1), K, thr);
        before += c;
        if (last && c) mx = fmaxf(mx, box_max(r));
      }
    }
  }
  int cnt = 0;
  if (row < n) {
    if (quad) {
      cnt = count16(v4);
    } else {
      cnt = count_above(s, K, thr);
    }
    if (last && cnt) mx = fmaxf(mx, box_max(row));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    before += __shfl_xor(before, d);
    mx = fmaxf(mx, __shfl_xor(mx, d));
  }
  if (lane == 0) {
    wbef[wave] = before;
    wmax[wave] = mx;
  }
  int total;
  const int excl = block_exclusive_scan(cnt, wsum, total);  // (its barrier also publishes wbef / wmax)
  int bsum = 0;
#pragma unroll
  for (int w = 0; w < SEL_T / 64; w++) bsum += wbef[w];
  if (last && tid == 0) {
    float m = wmax[0];
    for (int w = 1; w < SEL_T / 64; w++) m = fmaxf(m, wmax[w]);
    counts[img] = bsum + total;
    maxc[img] = m;
  }
  int pos = bsum + excl;
  if (cnt) {
    const size_t cb = (size_t)img * cand_stride;
    auto put = [&](const int k, const float v) {
      if (v > thr) {
        cand_row[cb + pos] = row;
        cand_label[cb + pos] = k;
        cand_score[cb + pos] = v;
        cand_rank[cb + pos] = 0;
        pos++;
      }
    };
    if (quad) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        put(q * 4, v4[q].x);
        put(q * 4 + 1, v4[q].y);
        put(q * 4 + 2, v4[q].z);
        if (q < 3) put(q * 4 + 3, v4[q].w);
      }
    } else {
      for (int k = 0; k < K; k++) put(k, s[k]);
    }
  }
}

// Stable descending sort of an image's candidates by COUNTING: rank(i) = #{j : s_j > s_i or
// (s_j == s_i and j < i)} = position of i under torch.sort(descending=True, stable=True).
// O(M^2) compares, but spread over the chip in one launch with device-side M (M = 3 k: ~5 us,
// where a segmented radix sort of 4 segments takes 97 us on one workgroup each).

__device__ __forceinline__ unsigned order_key(float f) {  // monotone: a < b  <=>  key(a) < key(b)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// v3 class offsets (obb_batched_nms, nms_rotated_wrapper.py:78-98): label * (hbb.max() - hbb.min() + 1)
// over the circumscribed horizontal boxes of ALL candidates (obb2hbb :7-20, fp32 cosf / sinf like
// the torch ops it mirrors).  One workgroup per image; extent[img] = (max - min) + 1.
__global__ __launch_bounds__(1024) void mc_hbb_extent_kernel(const float* __restrict__ boxes, int n,
                                                             const int* __restrict__ cand_row, int cand_stride,
                                                             const int* __restrict__ counts, int cap,
                                                             float* __restrict__ extent) {
  __shared__ float smin[16], smax[16];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = min(counts[img], cap);
  float lo = INFINITY, hi = -INFINITY;
  for (int c = tid; c < M; c += 1024) {
    const float* b = boxes + ((size_t)img * n + cand_row[(size_t)img * cand_stride + c]) * 5;
    const float cs = cosf(b[4]), sn = sinf(b[4]);
    const float xb = fabsf(b[2] / 2 * cs) + fabsf(b[3] / 2 * sn);
    const float yb = fabsf(b[2] / 2 * sn) + fabsf(b[3] / 2 * cs);
    lo = fminf(lo, fminf(b[0] - xb, b[1] - yb));
    hi = fmaxf(hi, fmaxf(b[0] + xb, b[1] + yb));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d));
    hi = fmaxf(hi, __shfl_xor(hi, d));
  }
  if (lane == 0) {
    smin[wave] = lo;
    smax[wave] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; w++) {
      lo = fminf(lo, smin[w]);
      hi = fmaxf(hi, smax[w]);
    }
    extent[img] = hi - lo + 1.f;
  }
}

// the class offsets' scale: one float per image, or (sparts > 0, the one-pool entries, round 5) the partial results of
// rnms_begin_kernel's sparts workgroups -- v1: their maxima (NaN where a workgroup saw one); v3: minima, then maxima
template <int GEOM>
__device__ __forceinline__ float mc_scale_of(const float* __restrict__ scale, const int img, const int sparts) {
  if (sparts <= 0) return scale[img];
  if (GEOM == 1) {
    float m = scale[0];
    bool bad = m != m;
    for (int g = 1; g < sparts; g++) {
      const float v = scale[g];
      bad |= v != v;
      m = fmaxf(m, v);
    }
    return bad ? __builtin_nanf("") : m;
  }
  float lo = scale[0], hi = scale[sparts];
  for (int g = 1; g < sparts; g++) {
    lo = fminf(lo, scale[g]);
    hi = fmaxf(hi, scale[sparts + g]);
  }
  return hi - lo + 1.f;
}

// candidate c's box as the kernels see it (class offsets applied; a v3 box thinner than 1e-3 moved out of reach)
template <int GEOM>
__device__ __forceinline__ bool mc_offset_box(const float* __restrict__ b, const float lab, const int c,
                                              const float* __restrict__ scale, const int img, const int sparts,
                                              float (&d)[5]) {
  d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; d[3] = b[3]; d[4] = b[4];
  bool is_dead = false;
  if (GEOM == 1) {
    const float off = lab * (mc_scale_of<GEOM>(scale, img, sparts) + 1.f);
    d[0] += off;
    d[1] += off;
  } else if (GEOM == 3) {
    const float off = lab * mc_scale_of<GEOM>(scale, img, sparts);
    d[0] += off;
    d[1] += off;
    is_dead = fminf(b[2], b[3]) < 0.001f;
    if (is_dead) {  // disjoint from every live box and from each other by the stream kernel's tests
      d[0] = 1e30f;
      d[1] = (float)c * 1e27f;
    }
  }
  return is_dead;
}

// candidate c of image img: its record (class offsets applied), whether it is a dead v3 box, its label
template <int GEOM>
__device__ __forceinline__ void mc_build_record(const float* __restrict__ boxes, const int n,
                                                const int* __restrict__ cand_row, const int* __restrict__ cand_label,
                                                const size_t cbase, const int c, const int img,
                                                const float* __restrict__ scale, const int sparts, BoxRec& r,
                                                bool& is_dead, int& label) {
  const float* b = boxes + ((size_t)img * n + cand_row[cbase + c]) * 5;
  label = cand_label[cbase + c];
  const float lab = (float)label;
  float d[5];
  is_dead = mc_offset_box<GEOM>(b, lab, c, scale, img, sparts, d);
  make_record<GEOM>(d, GEOM == 2 ? lab : 0.f, r);
}

// ... goes to sorted position pos
template <int GEOM>
__device__ __forceinline__ void mc_store_record(const BoxRec& r, const bool is_dead, const int label, const int c,
                                                const int pos, const int img, BoxRec* __restrict__ recs,
                                                const size_t recs_stride, int* __restrict__ sorted_vals,
                                                uint8_t* __restrict__ dead, uint8_t* __restrict__ rlab,
                                                u64* __restrict__ gbits, const int cap) {
  recs[img * recs_stride + pos] = r;
  sorted_vals[img * recs_stride + pos] = c;
  rlab[img * recs_stride + pos] = (uint8_t)label;  // (the drain's cross-group test reads the low bits)
  if (gbits) {
    // the row in its label group's bitmap (sorted-chunk form only: there the words were zeroed by the launch in front.
    // The counting form zeroes them in ITS OWN launch -- an atomicOr here would race with another workgroup's fill --
    // and leaves the bitmaps to the stream kernel's diagonal tiles.)
    const size_t cbw = (size_t)((cap + TILE - 1) / TILE);
    atomicOr(&gbits[((size_t)img * RG_GROUPS + (label & (RG_GROUPS - 1))) * cbw + (pos >> 6)], 1ULL << (pos & 63));
  }
  if (GEOM == 3) dead[img * recs_stride + c] = is_dead;
}

template <int GEOM>
__device__ __forceinline__ void mc_place_record(const float* __restrict__ boxes, const int n,
                                                const int* __restrict__ cand_row, const int* __restrict__ cand_label,
                                                const size_t cbase, const int c, const int pos, const int img,
                                                const float* __restrict__ scale, const int sparts,
                                                BoxRec* __restrict__ recs, const size_t recs_stride,
                                                int* __restrict__ sorted_vals, uint8_t* __restrict__ dead,
                                                uint8_t* __restrict__ rlab, u64* __restrict__ gbits, const int cap) {
  BoxRec r;
  bool is_dead;
  int label;
  mc_build_record<GEOM>(boxes, n, cand_row, cand_label, cbase, c, img, scale, sparts, r, is_dead, label);
  mc_store_record<GEOM>(r, is_dead, label, c, pos, img, recs, recs_stride, sorted_vals, dead, rlab, gbits, cap);
}

// one thread per candidate: record at its sorted position, and the inverse permutation.
//   GEOM 1: x, y += label * (max + 1)          (batched_rnms, rnms_wrapper.py:58-63)
//   GEOM 3: x, y += label * extent; boxes thinner than 1e-3 never take part (obb_nms removes them
//           before its kernel, nms_rotated_wrapper.py:40-46): flagged dead and moved out of reach
//   GEOM 2: no offsets, the label rides in the record (ml_nms_rotated: IoU = 0 across labels)
// begin + rank + prepare in ONE launch (round 2: three launches of <= 12 us of work that each took their ~4.5 us
// of the stream, and 0.3 M atomics into the rank array).  A workgroup owns RP_C = 32 candidates of one image and
// ranks them itself against ALL M candidates, tile by tile through LDS (rank by counting, as the rank kernel it
// replaces): RP_P threads per candidate, each a slice of every tile -- M = 8576 is 1072 workgroups of 268 compares
// per thread (measured at 8576: 256 candidates per workgroup, one thread each = 34 workgroups: 84 us; 32 candidates
// x 8 threads = 268 workgroups: 22 us).  Then the records go to their
// ranks; on the way the grid zeroes the overflow masks + side tables and the queue counters the next kernels expect.
constexpr int RP_TJ = 1024;  // keys per LDS tile (measured at n = 2000 / 8576, round 5: 512 -> 5.5 / 16.1 us, 1024 -> 5.6 / 16.8, 2048 -> 7.9 / 21.9, 4096 -> 8.4 / 28.8; without the ranking loop the kernel takes 6.1 us at 8576: zero fill + records)
constexpr int RP_C = 8;      // candidates per workgroup
constexpr int RP_P = 32;     // threads per candidate

// CT (round 6): candidates per thread.  A tile's keys are read from LDS once per thread whatever CT is, so the LDS
// traffic of the ranking loop falls by CT; pools beyond 16 384 candidates take CT = 4 (M = 32 768: 4096 workgroups
// pulling all keys through LDS for 8 candidates each -> 1024 for 32).  Measured: 109 -> 100 us only -- at that size the
// kernel is the 134 MB zero fill of the dense M x M / 64 suppressor mask it carries (the reducer's input), not the
// ranking; a pool of that size wants sparse suppressor lists instead of the mask.
template <int GEOM, int CT = 1>
__global__ __launch_bounds__(256) void mc_sort_prepare_kernel(
    const float* __restrict__ boxes, int n, const int* __restrict__ cand_row, const int* __restrict__ cand_label,
    const float* __restrict__ cand_score, int cand_stride, const int* __restrict__ counts_raw, int cap,
    int* __restrict__ ccounts, const float* __restrict__ scale, BoxRec* __restrict__ recs, size_t recs_stride,
    int* __restrict__ sorted_vals, uint8_t* __restrict__ dead, uint8_t* __restrict__ rlab, u64* __restrict__ gbits,
    unsigned* __restrict__ counter, size_t counter_stride, uint4* __restrict__ zero, size_t zero16, int sparts) {
  __shared__ __attribute__((aligned(16))) unsigned keys[RP_TJ];
  __shared__ int partial[RP_P][RP_C * CT];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int M = min(counts_raw[img], cap);  // (an image with more candidates than cap is its first cap candidates)
  if (blockIdx.x == 0 && tid == 0) ccounts[img] = M;
  {
    // The zeroed region starts with the overflow masks: gridDim.y x cap rows of cb words.  Row r is only ever read (and
    // written: mark_pair, i < r) in its words 0 .. r / 64, so the fill leaves the rest of the row alone -- half the
    // mask's bytes (round 6; at cap = 32 768 the mask is 134 MB and its fill was most of this kernel).  A 16-byte store
    // is two words: skipped when its first word is beyond the row's last and its second is in the same row.
    const size_t nthreads = (size_t)gridDim.x * gridDim.y * 256;
    const unsigned cb = (unsigned)((cap + TILE - 1) / TILE);
    const size_t mwords = (size_t)gridDim.y * (size_t)cap * cb;
    const size_t mask16 = mwords < ((size_t)1 << 32) ? mwords / 2 : 0;  // (32-bit word indices below; beyond: the plain fill)
    for (size_t k = (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; k < zero16; k += nthreads) {
      if (k < mask16) {
        const unsigned w0 = (unsigned)(2 * k), R = w0 / cb, wi = w0 - R * cb;
        const unsigned r = gridDim.y > 1 ? R % (unsigned)cap : R;
        if (wi > (r >> 6) && wi + 1 < cb) continue;
      }
      zero[k] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  for (int k = blockIdx.x * 256 + tid; k < Q_CTL_WORDS; k += gridDim.x * 256) counter[img * counter_stride + k] = 0;
  constexpr int WC = RP_C * CT;  // candidates of a workgroup
  const int i0 = blockIdx.x * WC;
  if (i0 >= M) return;
  const size_t cbase = (size_t)img * cand_stride;
  const float* sc = cand_score + cbase;
  const int ci = tid & (RP_C - 1), part = tid / RP_C;
  // the thread's candidates: i0 + ci + k * RP_C
  unsigned ui[CT];
  int cnt[CT];
#pragma unroll
  for (int k = 0; k < CT; k++) {
    const int ck = i0 + ci + k * RP_C;
    ui[k] = ck < M ? order_key(sc[ck]) : 0xffffffffu;
    cnt[k] = 0;
  }
  constexpr int SL4 = RP_TJ / RP_P / 4;  // uint4 reads of a thread per tile
  const uint4* k4 = reinterpret_cast<const uint4*>(keys) + part * SL4;
  // (the next tile's scores are requested before the current tile is counted: a tile is 4 keys per thread)
  float nxt[RP_TJ / 256];
#pragma unroll
  for (int u = 0; u < RP_TJ / 256; u++) nxt[u] = (u * 256 + tid) < M ? sc[u * 256 + tid] : 0.f;
  for (int j0 = 0; j0 < M; j0 += RP_TJ) {
    const int jn = min(RP_TJ, M - j0);
    __syncthreads();
    // key 0 (the bit pattern of a NaN, never a candidate) pads the tile: it is below every real key
#pragma unroll
    for (int u = 0; u < RP_TJ / 256; u++) keys[u * 256 + tid] = (u * 256 + tid) < jn ? order_key(nxt[u]) : 0u;
#pragma unroll
    for (int u = 0; u < RP_TJ / 256; u++) {
      const int j = j0 + RP_TJ + u * 256 + tid;
      nxt[u] = j < M ? sc[j] : 0.f;
    }
    __syncthreads();
    // a tile entirely before / after this workgroup's candidates needs one compare per key (ties count / do not
    // count); only a tile that overlaps them needs the per-lane tie rule
    if (j0 + RP_TJ <= i0) {
#pragma unroll 8
      for (int q = 0; q < SL4; q++) {  // (half-wave broadcast b128 reads: 4 keys per LDS instruction)
        const uint4 u = k4[q];
#pragma unroll
        for (int k = 0; k < CT; k++) cnt[k] += (u.x >= ui[k]) + (u.y >= ui[k]) + (u.z >= ui[k]) + (u.w >= ui[k]);
      }
    } else if (j0 >= i0 + WC) {
#pragma unroll 8
      for (int q = 0; q < SL4; q++) {
        const uint4 u = k4[q];
#pragma unroll
        for (int k = 0; k < CT; k++) cnt[k] += (u.x > ui[k]) + (u.y > ui[k]) + (u.z > ui[k]) + (u.w > ui[k]);
      }
    } else {
      int before[CT];  // j < before[k]  <=>  this slice's key j precedes candidate k
#pragma unroll
      for (int k = 0; k < CT; k++) before[k] = (i0 + ci + k * RP_C) - j0 - part * (RP_TJ / RP_P);
#pragma unroll 4
      for (int q = 0; q < SL4; q++) {
        const uint4 u = k4[q];
        const int j = q * 4;
#pragma unroll
        for (int k = 0; k < CT; k++) {
          cnt[k] += (u.x > ui[k]) | ((u.x == ui[k]) & (j < before[k]));
          cnt[k] += (u.y > ui[k]) | ((u.y == ui[k]) & (j + 1 < before[k]));
          cnt[k] += (u.z > ui[k]) | ((u.z == ui[k]) & (j + 2 < before[k]));
          cnt[k] += (u.w > ui[k]) | ((u.w == ui[k]) & (j + 3 < before[k]));
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CT; k++) partial[part][ci + k * RP_C] = cnt[k];
  __syncthreads();
  if (tid >= WC) return;        // (one thread per candidate from here on: thread t = candidate i0 + t)
  int total_before = 0;
#pragma unroll
  for (int q = 0; q < RP_P; q++) total_before += partial[q][tid];
  const int c = i0 + tid;
  if (c >= M) return;
  mc_place_record<GEOM>(boxes, n, cand_row, cand_label, cbase, c, total_before, img, scale, sparts, recs, recs_stride,
                        sorted_vals, dead, rlab, nullptr, cap);
}

// ---------------------------------------------------------------------------- large pools: sorted chunks (round 6)
// Rank by counting is M^2 compares and the stream kernel's tile loop M^2 / 2 box tests: at M = 32 768 they were 100 and
// 150 us of the call's 530.  Beyond P_MIN_CAP = 10 240 candidates (option nms_impl 6: always, 7: never) the pipeline runs on
// SORTED CHUNKS instead:
//   mc_chunk_sort_kernel   a workgroup sorts 1024 candidates twice in LDS (bitonic, both sorts side by side): by score
//                          and by the x of the box centre as the kernels see it (class offsets applied), and leaves the
//                          sorted keys and every candidate's place in its chunk;
//   mc_sort_prepare_p_kernel  a candidate's rank = its place in its own chunk + one binary search per other chunk
//                          (10 LDS reads for 1024 keys where counting took 1024 compares), for both orders: the score
//                          rank is the sorted position the rest of the pipeline works in (unchanged), the x rank its
//                          slot in the permutation P; the x extent of every 64 slots of P is collected on the way;
//   nms_stream_kernel<.., true>  walks P x P instead of position x position: a 64 x 64 tile whose two x extents are
//                          apart leaves at once, the others test their pairs as before and queue them by POSITION
//                          (smaller first), so the drain and the reducer see nothing new.  With the classes offset along
//                          the diagonal and the boxes of a class spread over the image, 1-2 % of the tiles remain.
// The drain's redo tiles (a tile too dense for its queue segment) name chunks of P and are enumerated through P.
constexpr int CS_N = 1024;           // candidates per sorted chunk
constexpr int P_MIN_CAP = 10240;     // pools above this take the sorted-chunk form (measured, per call: 8576 99 ↔ 108 us, 10 240 106 ↔ 108, 12 211 129 ↔ 120, 16 384 173 ↔ 134)
#ifndef R3_PP_C
#define R3_PP_C 64
#endif
constexpr int PP_CANDS = R3_PP_C;    // candidates per workgroup of mc_sort_prepare_p_kernel (x PP_PARTS threads)
constexpr int PP_PARTS = 4;          // chunks in LDS at a time = threads per candidate
constexpr size_t PP_LDS_BYTES = (size_t)2 * 2 * PP_PARTS * CS_N * 4;  // two buffers of score + x chunks
constexpr int CS_FILL_WGS = 256;     // workgroups of mc_chunk_sort_kernel that zero the masks and side tables meanwhile

struct PSort {
  unsigned* skeys;        // per image: sorted score keys, chunk after chunk (descending)
  unsigned* xkeys;        // the same for the x keys
  unsigned* xraw;         // candidate -> its x key
  unsigned short* slr;    // candidate -> place in its chunk, score order
  unsigned short* xlr;    // the same, x order
  unsigned short* perm;   // P: slot -> sorted position
  float4* ranges;         // per 64 slots of P: lo x, hi x, lo y, hi y of the boxes' inflated axis-aligned bounds
  float4* slotbox;        // per slot of P: the same four numbers of its one box (mc_ranges_kernel reduces them)
  size_t stride;          // elements between images in the arrays above (cap rounded up to CS_N)
  size_t rstride;         // int4 between images in ranges
  u64* stamps;            // (probe, tools/probes/nms_reduce_probe.hip: clock stamps of workgroup (0, 0) of the ranking kernel)
};

// Where sorted place p of a chunk is stored: places 0 .. CS_N - 2 as a complete binary search tree in LEVEL ORDER (root
// = place 511, then 255 and 767, ...), place CS_N - 1 last.  A binary search over the sorted array itself reads, at every
// level, addresses that differ by a multiple of the level's stride -- 64 lanes on one or two LDS banks: up to 32-way
// conflicts, ~100 ns per read, 2.5 us for the two searches of a group (stamps).  In level order a level's pivots are
// neighbours.
__device__ __forceinline__ int chunk_tree_index(const int p) {
  if (p == CS_N - 1) return CS_N - 1;
  const int q = p + 1, tz = __builtin_ctz(q);
  return (CS_N >> (tz + 1)) - 1 + (q >> (tz + 1));
}

template <int GEOM>
__global__ __launch_bounds__(CS_N) void mc_chunk_sort_kernel(const float* __restrict__ boxes, int n,
                                                             const int* __restrict__ cand_row,
                                                             const int* __restrict__ cand_label,
                                                             const float* __restrict__ cand_score, int cand_stride,
                                                             const int* __restrict__ counts_raw, int cap,
                                                             const float* __restrict__ scale, int sparts, PSort ps,
                                                             uint4* __restrict__ zero, size_t zero16, int nchunks) {
  __shared__ u64 el[2][2][CS_N];  // the exchange buffers of the steps across wavefronts (double-buffered), per order
  const int img = blockIdx.y, tid = threadIdx.x;
  if ((int)blockIdx.x >= nchunks) {
    // The workgroups behind the chunks: the fill that mc_sort_prepare_kernel carries in the other form (overflow masks
    // -- words 0 .. r / 64 of row r only -- then side tables and keep bits).  The sort is 32 workgroups waiting on LDS
    // and barriers; the memory system is free meanwhile (in the ranking kernel the fill was 30 of its 45 us at 32 768).
    const size_t nthreads = (size_t)(gridDim.x - nchunks) * gridDim.y * CS_N;
    const unsigned cb = (unsigned)((cap + TILE - 1) / TILE);
    const size_t mwords = (size_t)gridDim.y * (size_t)cap * cb;
    const size_t mask16 = mwords < ((size_t)1 << 32) ? mwords / 2 : 0;
    for (size_t k = ((size_t)blockIdx.y * (gridDim.x - nchunks) + (blockIdx.x - nchunks)) * CS_N + tid; k < zero16; k += nthreads) {
      if (k < mask16) {
        const unsigned w0 = (unsigned)(2 * k), R = w0 / cb, wi = w0 - R * cb;
        const unsigned r = gridDim.y > 1 ? R % (unsigned)cap : R;
        if (wi > (r >> 6) && wi + 1 < cb) continue;
      }
      zero[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const int M = min(counts_raw[img], cap);
  const int c0 = blockIdx.x * CS_N;
  if (c0 >= M) return;
  const size_t cbase = (size_t)img * cand_stride;
  const int c = c0 + tid;
  unsigned ks = 0u, kx = 0u;  // (key 0 pads the chunk: below every real key -- the bit pattern of a NaN)
  if (c < M) {
    ks = order_key(cand_score[cbase + c]);
    float d[5];
    const float* b = boxes + ((size_t)img * n + cand_row[cbase + c]) * 5;
    mc_offset_box<GEOM>(b, (float)cand_label[cbase + c], c, scale, img, sparts, d);
    kx = order_key(d[0]);
    if (kx == 0u) kx = 1u;  // (a real candidate never carries the pad key)
    if (ks == 0u) ks = 1u;
    ps.xraw[img * ps.stride + c] = kx;
  }
  // Bitonic network with one element of each order per thread: a step whose partner is within the wavefront (distance
  // < 64: 45 of the 55 steps) is two shuffles per element, the ten others go through LDS, double-buffered -- one
  // workgroup barrier each.  (The first form kept both arrays in LDS with a compare-exchange per thread and a barrier
  // per step: 18 us for a kernel of 32 workgroups; with the wavefront-local steps behind a wavefront fence: 15.7.)
  u64 vs = ((u64)ks << 32) | (u64)(CS_N - 1 - tid);  // (key << 32) | (1023 - place): descending = key down, index up on ties
  u64 vx = ((u64)kx << 32) | (u64)(CS_N - 1 - tid);
  auto sx = [&](const u64 v, const int j) -> u64 {
    const unsigned lo = __shfl_xor((unsigned)v, j), hi = __shfl_xor((unsigned)(v >> 32), j);
    return ((u64)hi << 32) | lo;
  };
  int cur = 0;
#pragma unroll 1
  for (int k = 2; k <= CS_N; k <<= 1) {
    const bool down = (tid & k) == 0;
#pragma unroll 1
    for (int j = k >> 1; j >= 1; j >>= 1) {
      u64 os, ox;
      if (j >= TILE) {
        el[cur][0][tid] = vs;
        el[cur][1][tid] = vx;
        __syncthreads();
        os = el[cur][0][tid ^ j];
        ox = el[cur][1][tid ^ j];
        cur ^= 1;
      } else {
        os = sx(vs, j);
        ox = sx(vx, j);
      }
      const bool keep_max = ((tid & j) == 0) == down;  // (the lower index of a pair in a descending block keeps the larger)
      vs = keep_max ? (vs > os ? vs : os) : (vs < os ? vs : os);
      vx = keep_max ? (vx > ox ? vx : ox) : (vx < ox ? vx : ox);
    }
  }
  ps.skeys[img * ps.stride + c0 + chunk_tree_index(tid)] = (unsigned)(vs >> 32);  // (the ranking kernel's search tree)
  ps.xkeys[img * ps.stride + c0 + chunk_tree_index(tid)] = (unsigned)(vx >> 32);
  ps.slr[img * ps.stride + c0 + (CS_N - 1 - (int)(vs & (CS_N - 1)))] = (unsigned short)tid;
  ps.xlr[img * ps.stride + c0 + (CS_N - 1 - (int)(vx & (CS_N - 1)))] = (unsigned short)tid;
}

// numbers of keys > thr in two descending chunks of CS_N keys in LDS (tree layout), searched side by side
__device__ __forceinline__ void chunk_counts_above(const unsigned* __restrict__ ka, const unsigned ta,
                                                   const unsigned* __restrict__ kb, const unsigned tb, int& ca, int& cb) {
  int ia = 0, ib = 0;
#pragma unroll
  for (int level = 0; level < 10; level++) {  // (CS_N = 2^10)
    const unsigned va = ka[ia], vb = kb[ib];
    ia = 2 * ia + 1 + (va > ta ? 1 : 0);  // a key above the threshold: everything in front of it is too -- to the right
    ib = 2 * ib + 1 + (vb > tb ? 1 : 0);
  }
  ca = ia - (CS_N - 1);  // the leaf's number = keys above among the first CS_N - 1
  cb = ib - (CS_N - 1);
  if (ca == CS_N - 1) ca += ka[CS_N - 1] > ta ? 1 : 0;
  if (cb == CS_N - 1) cb += kb[CS_N - 1] > tb ? 1 : 0;
}

// workgroup barrier for LDS traffic alone: global loads in flight stay in flight (__syncthreads waits for them: the
// next group's keys, requested right in front of it, cost 1.2 us of waiting per group)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int GEOM, int PP_C>
__global__ __launch_bounds__(PP_C * PP_PARTS) void mc_sort_prepare_p_kernel(
    const float* __restrict__ boxes, int n, const int* __restrict__ cand_row, const int* __restrict__ cand_label,
    const float* __restrict__ cand_score, int cand_stride, const int* __restrict__ counts_raw, int cap,
    int* __restrict__ ccounts, const float* __restrict__ scale, BoxRec* __restrict__ recs, size_t recs_stride,
    int* __restrict__ sorted_vals, uint8_t* __restrict__ dead, uint8_t* __restrict__ rlab, u64* __restrict__ gbits,
    unsigned* __restrict__ counter, size_t counter_stride, int sparts, PSort ps) {
  // dynamic LDS, 64 KB: [2 buffers][score | x][PP_PARTS chunks][CS_N keys] -- a group of chunks is searched while the next
  // one lands in the other buffer (one buffer: two barriers per group and the wait for its keys in between, 3.1 us per
  // group of which 1.3 were the searches)
  extern __shared__ __attribute__((aligned(16))) unsigned keys_dyn[];
  __shared__ int partial[2][PP_PARTS][PP_C];
  auto kbuf = [&](const int b, const int o) -> unsigned* { return keys_dyn + (size_t)(b * 2 + o) * PP_PARTS * CS_N; };
  const int img = blockIdx.y, tid = threadIdx.x;
  const int M = min(counts_raw[img], cap);
  if (blockIdx.x == 0 && tid == 0) ccounts[img] = M;
  constexpr int NTH = PP_C * PP_PARTS;
  constexpr int LPT = PP_PARTS * CS_N / 4 / NTH;  // 16-byte loads per thread, array and group of chunks
  for (int k = blockIdx.x * NTH + tid; k < Q_CTL_WORDS; k += gridDim.x * NTH) counter[img * counter_stride + k] = 0;
  const int i0 = blockIdx.x * PP_C;
  const bool stamp_on = ps.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  auto stamp = [&](const int k) {
    if (stamp_on) ps.stamps[k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  if (i0 < M) {
  const size_t cbase = (size_t)img * cand_stride;
  const int ci = tid & (PP_C - 1), part = tid / PP_C;  // (a wavefront = the 64 candidates against one chunk)
  const int c = i0 + ci;
  const bool has = c < M;
  const unsigned us = has ? max(order_key(cand_score[cbase + c]), 1u) : 1u;
  const unsigned ux = has ? ps.xraw[img * ps.stride + c] : 1u;
  const int own = c / CS_N;
  const int nch = (M + CS_N - 1) / CS_N;
  const unsigned* gs = ps.skeys + img * ps.stride;
  const unsigned* gx = ps.xkeys + img * ps.stride;
  int cnt_s = 0, cnt_x = 0;
  // (the next group of chunks is requested before the current one is searched: 4 chunks x 2 orders = 32 keys per thread)
  uint4 nxt[2][LPT];
  auto request = [&](const int t0) {
#pragma unroll
    for (int q = 0; q < LPT; q++) {
      const int idx = tid + q * NTH;  // 16-byte element of the group: chunk idx / 256, element idx % 256
      const bool on = t0 + idx / (CS_N / 4) < nch;
      nxt[0][q] = on ? reinterpret_cast<const uint4*>(gs + (size_t)t0 * CS_N)[idx] : make_uint4(0u, 0u, 0u, 0u);
      nxt[1][q] = on ? reinterpret_cast<const uint4*>(gx + (size_t)t0 * CS_N)[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  // (the workgroups start at different groups of chunks: all of them asking for the same 32 KB at the same time queue
  // up at the few L2 channels that hold it)
  const int ngrp = (nch + PP_PARTS - 1) / PP_PARTS;
  const int rot = (int)(blockIdx.x % (unsigned)ngrp);
  auto group_of = [&](const int g) { const int x = g + rot; return (x >= ngrp ? x - ngrp : x) * PP_PARTS; };
  auto store = [&](const int b) {
#pragma unroll
    for (int q = 0; q < LPT; q++) {
      reinterpret_cast<uint4*>(kbuf(b, 0))[tid + q * NTH] = nxt[0][q];
      reinterpret_cast<uint4*>(kbuf(b, 1))[tid + q * NTH] = nxt[1][q];
    }
  };
  stamp(1);
  request(group_of(0));
  // (the candidate's record is built by its first thread BEFORE the ranks are known, under the first keys' flight -- box, class offset, trigonometry:
  // two dependent trips to memory that used to follow the searches -- and only stored behind them)
  BoxRec rec;
  bool rec_dead = false;
  int rec_label = 0;
  if (part == 0 && has)
    mc_build_record<GEOM>(boxes, n, cand_row, cand_label, cbase, c, img, scale, sparts, rec, rec_dead, rec_label);
  store(0);
  if (ngrp > 1) request(group_of(1));
  lds_barrier();
  stamp(2);
  for (int g = 0; g < ngrp; g++) {
    const int t0 = group_of(g);
    if (g == 2) stamp(5);
    if (g + 1 < ngrp) {
      store((g + 1) & 1);  // (last searched in the previous turn, behind its barrier)
      if (g + 2 < ngrp) request(group_of(g + 2));
    }
    if (g == 2) stamp(6);
    const int t = t0 + part;
    if (t < nch && t != own) {
      // chunks in front of the candidate's own count ties (their candidates come first), chunks behind it do not:
      // keys >= u  <=>  keys > u - 1 (a candidate's key is never 0)
      const unsigned dec = t < own ? 1u : 0u;
      int a_s, a_x;
      chunk_counts_above(kbuf(g & 1, 0) + part * CS_N, us - dec, kbuf(g & 1, 1) + part * CS_N, ux - dec, a_s, a_x);
      cnt_s += a_s;
      cnt_x += a_x;
    }
    if (g == 2) stamp(7);
    lds_barrier();
    if (g == 2) stamp(8);
  }
  stamp(3);
  partial[0][part][ci] = cnt_s;
  partial[1][part][ci] = cnt_x;
  __syncthreads();
  if (tid < PP_C && has) {  // (one thread per candidate from here on)
  int pos = ps.slr[img * ps.stride + c], slot = ps.xlr[img * ps.stride + c];
#pragma unroll
  for (int q = 0; q < PP_PARTS; q++) {
    pos += partial[0][q][tid];
    slot += partial[1][q][tid];
  }
  mc_store_record<GEOM>(rec, rec_dead, rec_label, c, pos, img, recs, recs_stride, sorted_vals, dead, rlab, gbits, cap);
  const float rj[4] = {rec.f[9], rec.f[10], rec.f[12], rec.f[13]};  // (what the stream kernel's axis-aligned test reads)
  ps.perm[img * ps.stride + slot] = (unsigned short)pos;
  // (the extents of every 64 slots come from these in mc_ranges_kernel.  Collected here with atomicMin / atomicMax they
  // were 14 of this kernel's 31 us at 32 768 candidates: a group's 64 candidates sit in 64 workgroups that all finish
  // together, and their 256 atomics on one 16-byte line are served one after the other)
  ps.slotbox[img * ps.stride + slot] = make_float4(rj[0] - rj[2], rj[0] + rj[2], rj[1] - rj[3], rj[1] + rj[3]);
  }
  stamp(4);
  }
}

// the extents of every 64 slots of P: a wavefront per group, a lane per slot (fminf / fmaxf drop a NaN: such a box has no
// pair that matters, and the others' extent stands)
__global__ __launch_bounds__(256) void mc_ranges_kernel(const int* __restrict__ counts, PSort ps) {
  const int img = blockIdx.y, lane = threadIdx.x & 63;
  const int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int M = counts[img];  // (the clamped counts: written by the ranking kernel)
  if (grp * TILE >= M) return;
  const int slot = grp * TILE + lane;
  const float inf = __builtin_inff();
  float4 v = slot < M ? ps.slotbox[img * ps.stride + slot] : make_float4(inf, -inf, inf, -inf);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    v.x = fminf(v.x, __shfl_xor(v.x, d));
    v.y = fmaxf(v.y, __shfl_xor(v.y, d));
    v.z = fminf(v.z, __shfl_xor(v.z, d));
    v.w = fmaxf(v.w, __shfl_xor(v.w, d));
  }
  if (lane == 0) ps.ranges[img * ps.rstride + grp] = v;
}

// Where the finish kernels write.  List form (the reference's return values, bbox_nms_rotated.py:127-131): dets
// (B, out_cap, 6) = [box, score], labels int64, rows beyond the count untouched.  Padded form (round 5: the buffer the
// step's exchange sends, core/bbox/rtransforms.py:10-25 rbbox2result's input): cols = 7 = [box, score, label as fp32],
// `img_stride` floats between images, the rows beyond the count ZEROED, the count also as fp32 at
// count_f32[img * count_f32_stride] (the row gather_detections appends) and overflow[img] = (the image had more
// candidates than `cap`: the result is that of its first cap candidates and the caller redoes the step).
struct McOut {
  float* dets;
  int cols;           // 6 | 7
  size_t img_stride;  // floats
  int64_t* labels;    // may be null when cols == 7
  int64_t* keep_idx;  // may be null
  int32_t* counts;
  float* count_f32;   // may be null
  size_t count_f32_stride;
  int32_t* overflow;  // may be null
  const int* raw_counts;
  int cap;
  int pad;  // zero the rows beyond the count
};

__device__ __forceinline__ void mc_emit(const McOut& o, int img, int out_cap, int pos, const float* __restrict__ b,
                                        float score, int label, int cand) {
  float* d = o.dets + (size_t)img * o.img_stride + (size_t)pos * o.cols;
  d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; d[3] = b[3]; d[4] = b[4];
  d[5] = score;
  if (o.cols == 7) d[6] = (float)label;
  if (o.labels) o.labels[(size_t)img * out_cap + pos] = label;
  if (o.keep_idx) o.keep_idx[(size_t)img * out_cap + pos] = cand;
}

// one workgroup of an image: the count (both forms), the overflow flag, the zero rows behind the kept ones
__device__ __forceinline__ void mc_tail(const McOut& o, int img, int out_cap, int total) {
  const int kept = min(total, out_cap);
  if (threadIdx.x == 0) {
    o.counts[img] = kept;
    if (o.count_f32) o.count_f32[(size_t)img * o.count_f32_stride] = (float)kept;
    if (o.overflow) o.overflow[img] = o.raw_counts[img] > o.cap ? 1 : 0;
  }
  if (o.pad) {
    float* d = o.dets + (size_t)img * o.img_stride;
    for (int e = kept * o.cols + (int)threadIdx.x; e < out_cap * o.cols; e += (int)blockDim.x) d[e] = 0.f;
  }
}

__global__ __launch_bounds__(1024) void mc_finish_kernel(const float* __restrict__ boxes, int n,
                                                         const int* __restrict__ cand_row,
                                                         const int* __restrict__ cand_label,
                                                         const float* __restrict__ cand_score, int cand_stride,
                                                         const int* __restrict__ counts,
                                                         const u64* __restrict__ fbits, size_t fbits_stride, int out_cap,
                                                         const McOut o) {
  // Ascending keep: the reducer left a bit per kept CANDIDATE (fbits).  grid = (chunks of 1024 candidates, images):
  // a workgroup counts the kept candidates in front of its chunk (popcounts of at most 1023 words), then every
  // thread emits its own candidate.  (One workgroup per image that first translated the sorted rows to candidates
  // and then walked all of them: 14.5 us at 8576 candidates.)
  __shared__ int wsum[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.y;
  const int M = counts[img];
  const int i0 = blockIdx.x * 1024;
  if (i0 >= M && blockIdx.x != 0) return;
  const size_t cbase = (size_t)img * cand_stride;
  fbits += img * fbits_stride;
  const int words = (M + 63) >> 6;
  const int first = blockIdx.x * 16, wmine = first + wave;  // the chunk's words; the wave's own word
  int before = 0, all = 0;  // kept candidates in front of the chunk; all of them (workgroup 0: counts_out)
  for (int w = tid; w < words; w += 1024) {  // (M < 65536: one trip)
    const int c = __popcll(fbits[w]);
    all += c;
    if (w < first) before += c;
  }
  int inside = (lane < wave && first + lane < words) ? __popcll(fbits[first + lane]) : 0;  // the chunk's words in front of the wave's
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    before += __shfl_xor(before, d);
    all += __shfl_xor(all, d);
    inside += __shfl_xor(inside, d);
  }
  if (lane == 0) {
    wsum[0][wave] = before;
    wsum[1][wave] = all;
  }
  __syncthreads();
  int base = inside, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    base += wsum[0][w];
    tot += wsum[1][w];
  }
  if (blockIdx.x == 0) mc_tail(o, img, out_cap, tot);
  const int i = i0 + tid;
  if (i >= M) return;
  const u64 wbits = fbits[wmine];
  if (!((wbits >> lane) & 1ULL)) return;
  const int pos = base + __popcll(wbits & ((1ULL << lane) - 1ULL));
  if (pos >= out_cap) return;
  mc_emit(o, img, out_cap, pos, boxes + ((size_t)img * n + cand_row[cbase + i]) * 5, cand_score[cbase + i],
          cand_label[cbase + i], i);  // (keep_idx: the candidate index, ascending)
}

// finish for the score-ordered families (v3 obb_nms, v2 ml_nms_rotated): the kept rows already are in score order;
// drop the dead (too thin, v3) ones, keep the first out_cap, gather.
// Over the chip (round 5): grid = (chunks of 1024 sorted rows, images).  A workgroup counts the live rows in front of
// its chunk itself (every thread a stride of them: the kept bit, and for v3 the candidate's dead byte), then every
// thread emits its own row at count-in-front + rank inside the chunk.  (Rounds 2-4: one workgroup per image with a
// Hillis-Steele scan over per-thread serial ranges -- 18.8 us of the v3 pipeline's 94 at n = 8576.)
__global__ __launch_bounds__(1024) void mc_finish_score_chip_kernel(const float* __restrict__ boxes, int n,
                                                                    const int* __restrict__ cand_row,
                                                                    const int* __restrict__ cand_label,
                                                                    const float* __restrict__ cand_score, int cand_stride,
                                                                    const int* __restrict__ sorted_vals,
                                                                    const int* __restrict__ counts,
                                                                    const u64* __restrict__ kbits, size_t kbits_stride,
                                                                    size_t rows_stride,
                                                                    const uint8_t* __restrict__ dead, int out_cap,
                                                                    const McOut o) {
  __shared__ int wsum[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.y;
  const int M = counts[img];
  const int i0 = blockIdx.x * 1024;
  if (i0 >= M && blockIdx.x != 0) return;
  const size_t cbase = (size_t)img * cand_stride;
  kbits += img * kbits_stride;
  sorted_vals += img * rows_stride;
  if (dead) dead += img * rows_stride;
  auto live = [&](const int r) -> bool {
    return ((kbits[r >> 6] >> (r & 63)) & 1ULL) && !(dead && dead[sorted_vals[r]]);
  };
  // live rows in front of the chunk; workgroup 0 also needs all of them (the count)
  const int upto = blockIdx.x == 0 ? M : i0;
  int cnt_front = 0;
  // (measured and dropped, round 5: four rows per step with their loads requested together -- 10.1-11.1 us against 9.9)
  for (int r = tid; r < upto; r += 1024) cnt_front += live(r) ? 1 : 0;
  const int r = i0 + tid;
  const bool mine = r < M && live(r);
  const u64 mk = __ballot(mine);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt_front += __shfl_xor(cnt_front, d);
  if (lane == 0) {
    wsum[0][wave] = cnt_front;
    wsum[1][wave] = __popcll(mk);
  }
  __syncthreads();
  int front = 0, inside = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    front += wsum[0][w];
    if (w < wave) inside += wsum[1][w];
  }
  if (blockIdx.x == 0) {
    mc_tail(o, img, out_cap, front);  // (workgroup 0 counted every row: front = the image's live rows)
    front = 0;                         // ... of which none lies in front of its own chunk
  }
  if (!mine) return;
  const int pos = front + inside + __popcll(mk & ((1ULL << lane) - 1ULL));
  if (pos >= out_cap) return;
  const int cand = sorted_vals[r];
  mc_emit(o, img, out_cap, pos, boxes + ((size_t)img * n + cand_row[cbase + cand]) * 5, cand_score[cbase + cand],
          cand_label[cbase + cand], cand);
}

struct McLayout {
  int* svals;
  BoxRec* recs;
  u64* mask;
  u64* nz;
  unsigned* counter;
  unsigned* redo;
  unsigned* gqueue;
  int64_t* keep;
  int32_t* kept;
  uint8_t* flags;
  uint8_t* dead;
  uint8_t* rlab;   // label of every sorted row
  u64* kbits;      // kept rows as bits (cb words per image; zeroed with the masks)
  u64* fbits;      // kept CANDIDATES as bits (index order; the same)
  u64* gbits;      // per image and label group: its rows as bits (the same)
  float* extent;
  int* ccounts;
  PSort ps;        // the sorted-chunk form's arrays (pools beyond P_MIN_CAP)
  size_t qcap, qstride, zero_bytes;  // entries per region; entries per image
  int cb;
};

inline size_t mc_layout(int B, int cap, void* ws, McLayout* L) {
  const size_t cb = (cap + TILE - 1) / TILE, qcap = queue_entries(cap);
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* svals = take((size_t)B * cap * 4);
  char* recs = take((size_t)B * cap * sizeof(BoxRec));
  char* mask = take((size_t)B * cap * cb * 8);  // mask and nz: one fill
  char* nz = take((size_t)B * side_words((size_t)cap) * 8);  // per image: side tables
  char* kbits = take((size_t)B * cb * 8);                    // (still inside the zeroed region)
  char* fbits = take((size_t)B * cb * 8);                    // (the same)
  char* gbits = take((size_t)B * RG_GROUPS * cb * 8);        // (the same: the label groups' row bitmaps)
  char* counter = take((size_t)B * Q_CTL_WORDS * 4);
  char* gq = take((size_t)B * qcap * 4);
  char* rd = take((size_t)B * cb * cb * 4);  // redo-tile lists
  char* keep = take((size_t)B * cap * 8);
  char* kept = take((size_t)B * 4);
  char* flags = take((size_t)B * cap);
  char* dead = take((size_t)B * cap);
  char* rlab = take((size_t)B * cap);
  char* extent = take((size_t)B * 4);
  char* ccounts = take((size_t)B * 4);
  const size_t capR = ((size_t)cap + CS_N - 1) / CS_N * CS_N;
  char* p_sk = take((size_t)B * capR * 4);
  char* p_xk = take((size_t)B * capR * 4);
  char* p_xr = take((size_t)B * capR * 4);
  char* p_sl = take((size_t)B * capR * 2);
  char* p_xl = take((size_t)B * capR * 2);
  char* p_pm = take((size_t)B * capR * 2);
  char* p_rg = take((size_t)B * (capR / TILE) * 16);
  char* p_sb = take((size_t)B * capR * 16);
  if (L) {
    L->ps = PSort{(unsigned*)p_sk, (unsigned*)p_xk, (unsigned*)p_xr, (unsigned short*)p_sl, (unsigned short*)p_xl,
                  (unsigned short*)p_pm, (float4*)p_rg, (float4*)p_sb, capR, capR / TILE, nullptr};
    L->svals = (int*)svals; L->recs = (BoxRec*)recs; L->mask = (u64*)mask; L->nz = (u64*)nz;
    L->counter = (unsigned*)counter; L->gqueue = (unsigned*)gq; L->redo = (unsigned*)rd; L->keep = (int64_t*)keep;
    L->kept = (int32_t*)kept; L->flags = (uint8_t*)flags; L->dead = (uint8_t*)dead; L->rlab = (uint8_t*)rlab; L->kbits = (u64*)kbits; L->fbits = (u64*)fbits; L->gbits = (u64*)gbits; L->extent = (float*)extent; L->ccounts = (int*)ccounts;
    L->qcap = qcap / Q_NREG; L->qstride = qcap; L->zero_bytes = (size_t)(counter - mask); L->cb = (int)cb;
    const int qcap_o = g_r3_nms_qcap;  // (one read)
    if (qcap_o > 0 && (size_t)qcap_o < L->qcap) L->qcap = (size_t)qcap_o;
  }
  return off + 256;
}

inline int select_parts(int n) { return (n + SEL_T - 1) / SEL_T; }

}  // namespace

size_t r3k_mcnms_select_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 256;
  return align256((size_t)B * select_parts(n) * 4) * 2;
}

int r3k_mcnms_select(const float* boxes, const float* scores, int B, int n, int K, float score_thr,
                     int* cand_row, int* cand_label, float* cand_score, int* cand_rank, int* counts,
                     float* maxc, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (B <= 0 || n <= 0 || K <= 0 || !counts || !maxc) return -1;
  if (!boxes || !scores || !cand_row || !cand_label || !cand_score || !cand_rank || !ws) return -1;
  if (ws_bytes < r3k_mcnms_select_workspace_bytes(B, n)) return -3;
  const int parts = select_parts(n);
  int* part_cnt = (int*)ws;
  float* part_max = (float*)((char*)ws + align256((size_t)B * parts * 4));
  if (g_r3_nms_impl != 5) {
    hipLaunchKernelGGL(mc_select_kernel, dim3(parts, B), dim3(SEL_T), 0, stream, boxes, scores, n, K, score_thr, parts,
                       n * K, cand_row, cand_label, cand_score, cand_rank, counts, maxc);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  hipLaunchKernelGGL(mc_count_kernel, dim3(parts, B), dim3(SEL_T), 0, stream, boxes, scores, n, K, score_thr, parts,
                     part_cnt, part_max);
  hipLaunchKernelGGL(mc_write_kernel, dim3(parts, B), dim3(SEL_T), 0, stream, scores, n, K, score_thr, parts,
                     part_cnt, part_max, n * K, cand_row, cand_label, cand_score, cand_rank, counts, maxc);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t r3k_mcnms_workspace_bytes(int B, int cap) {
  if (B <= 0 || cap <= 0) return 256;
  return mc_layout(B, cap, nullptr, nullptr);
}

// geom 1: batched_rnms (v1, keep ascending); 3: obb_batched_nms (v3, score order, thin boxes dropped);
// 2: ml_nms_rotated (v2, label guard, score order).  maxc is used by geom 1 only.
int r3k_mcnms_run(int geom, const float* boxes, int B, int n, int K, const int* cand_row, const int* cand_label,
                  const float* cand_score, int* cand_rank, const int* counts, const float* maxc, int cap,
                  float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out, int64_t* labels_out,
                  int64_t* keep_idx_out, int32_t* counts_out, hipStream_t stream, const R3kMcPadded* padded,
                  int scale_parts) {
  if (geom < 1 || geom > 3) return -1;
  if (B <= 0 || n <= 0 || K <= 0 || cap <= 0 || out_cap <= 0 || cap >= 65536 || !(iou_thr >= 0.f)) return -1;
  if (!boxes || !cand_row || !cand_label || !cand_score || !cand_rank || !counts || !ws || !dets_out ||
      (!labels_out && !padded) || !counts_out || (geom == 1 && !maxc))
    return -1;
  if (padded && padded->img_stride < (size_t)out_cap * 7) return -1;
  McOut mo{dets_out, padded ? 7 : 6, padded ? padded->img_stride : (size_t)out_cap * 6, labels_out, keep_idx_out,
           counts_out, padded ? padded->count_f32 : nullptr, padded ? padded->count_f32_stride : 0,
           padded ? padded->overflow : nullptr, counts, cap, padded ? 1 : 0};
  if (ws_bytes < r3k_mcnms_workspace_bytes(B, cap)) return -3;
  const int S = n * K;
  McLayout L;
  mc_layout(B, cap, ws, &L);
  const size_t cbq = (size_t)L.cb;
  Batch bt{L.ccounts, (size_t)cap, (size_t)cap * cbq, side_words((size_t)cap), (size_t)Q_CTL_WORDS, L.qstride,
           (size_t)cap, cbq * cbq, (size_t)cap, L.rlab, L.gbits};
  // the rank kernel accumulates into cand_rank: zeroed here so that a caller's stale scratch cannot send
  // records out of bounds; and the counts are clamped to cap for the same reason (an image with more
  // candidates than cap is processed as its first cap candidates: the caller sizes cap from the counts,
  // or checks them afterwards and calls again)
  // (mask + side tables are adjacent and 256-byte aligned: zeroed by the begin kernel, whose grid is widened so that
  // the fill runs at memory speed)
  const bool big_pool = cap > 16384;  // (the ranking loop with 4 candidates per thread: mc_sort_prepare_kernel<., 4>)
  const int nms_impl0 = g_r3_nms_impl;  // (one read per call)
  L.ps.stamps = g_nms_stamps ? g_nms_stamps + 48 : nullptr;
  const bool use_p = (cap > P_MIN_CAP && nms_impl0 != 7) || nms_impl0 == 6;  // the sorted-chunk form
  // (its ranking kernel's 64 KB of dynamic LDS + 2 KB static are beyond the default cap: the opt-in, once per device and geometry)
  static R3DeviceOnce pp_once[3];
  const bool pp_raise = use_p && pp_once[geom - 1].first();
  const dim3 csgrid((unsigned)(L.ps.stride / CS_N) + CS_FILL_WGS, B), ppgrid((cap + PP_CANDS - 1) / PP_CANDS, B);
  // (the stream kernel of that form: a wavefront looks at up to 64 tiles of the cb x cb square; small squares take fewer
  // per wavefront so that ~4096 wavefronts share the work)
  const long long ptiles = (long long)L.cb * L.cb;
  const long long pwaves = std::max((ptiles + 63) / 64, std::min(ptiles, (long long)4096));
  const dim3 pstream_grid((unsigned)((pwaves + MASK_WAVES - 1) / MASK_WAVES), 1, B);
  const int pwc = big_pool ? RP_C * 4 : RP_C;
  const dim3 pgrid((cap + pwc - 1) / pwc, B), grid((L.cb + MASK_WAVES - 1) / MASK_WAVES, L.cb, B);
  const bool clip_fast = g_r3_clip_impl == 0;  // (one read per call -- the grid and the kernel form go together)
  const int chip_wgs = clip_fast ? 4 * r3_cu_count() : 2048;
  int dblocks = drain_blocks(L.qstride, chip_wgs);
  if (dblocks > chip_wgs / B) dblocks = chip_wgs / B > 0 ? chip_wgs / B : 1;  // B images share the chip
  const dim3 dgrid(dblocks, 1, B);
  if (geom == 3 && scale_parts <= 0)  // (reads the raw counts and clamps them itself: the clamped copy is written by the next kernel)
    hipLaunchKernelGGL(mc_hbb_extent_kernel, dim3(B), dim3(1024), 0, stream, boxes, n, cand_row, S, counts, cap, L.extent);
  (void)cand_rank;  // (scratch of the three-launch form of round 2; kept in the signature)
#define R3_MC(GEOM, LABEL, SCALE)                                                                                  \
  if (use_p) {                                                                                                     \
    hipLaunchKernelGGL((mc_chunk_sort_kernel<GEOM>), csgrid, dim3(CS_N), 0, stream, boxes, n, cand_row, cand_label, \
                       cand_score, S, counts, cap, SCALE, scale_parts, L.ps, reinterpret_cast<uint4*>(L.mask),      \
                       L.zero_bytes / 16, (int)(L.ps.stride / CS_N));                                               \
    if (pp_raise)                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mc_sort_prepare_p_kernel<GEOM, PP_CANDS>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS_BYTES);                     \
    hipLaunchKernelGGL((mc_sort_prepare_p_kernel<GEOM, PP_CANDS>), ppgrid, dim3(PP_CANDS * PP_PARTS), PP_LDS_BYTES, stream, \
                       boxes, n, cand_row,                                                                          \
                       cand_label, cand_score, S, counts, cap, L.ccounts, SCALE, L.recs, bt.recs, L.svals, L.dead,  \
                       L.rlab, L.gbits, L.counter, bt.counter, scale_parts, L.ps);                                 \
    hipLaunchKernelGGL(mc_ranges_kernel, dim3((unsigned)(L.ps.rstride + 3) / 4, B), dim3(256), 0, stream,           \
                       L.ccounts, L.ps);                                                                            \
    hipLaunchKernelGGL((nms_stream_kernel<GEOM, LABEL, true>), pstream_grid, dim3(NT), 0, stream, L.recs, 0, L.cb,  \
                       L.gqueue, (unsigned)L.qcap, L.counter, L.redo, bt, L.ps.perm, L.ps.ranges, L.ps.stride,      \
                       L.ps.rstride, (int)pwaves);                                                                  \
  } else {                                                                                                         \
  if (big_pool)                                                                                                    \
    hipLaunchKernelGGL((mc_sort_prepare_kernel<GEOM, 4>), pgrid, dim3(256), 0, stream, boxes, n, cand_row,         \
                       cand_label, cand_score, S, counts, cap, L.ccounts, SCALE, L.recs, bt.recs, L.svals, L.dead,  \
                       L.rlab, L.gbits, L.counter, bt.counter, reinterpret_cast<uint4*>(L.mask), L.zero_bytes / 16,          \
                       scale_parts);                                                                                \
  else                                                                                                             \
    hipLaunchKernelGGL((mc_sort_prepare_kernel<GEOM, 1>), pgrid, dim3(256), 0, stream, boxes, n, cand_row,         \
                       cand_label, cand_score, S, counts, cap, L.ccounts, SCALE, L.recs, bt.recs, L.svals, L.dead,  \
                       L.rlab, L.gbits, L.counter, bt.counter, reinterpret_cast<uint4*>(L.mask), L.zero_bytes / 16,          \
                       scale_parts);                                                                                \
  hipLaunchKernelGGL((nms_stream_kernel<GEOM, LABEL>), grid, dim3(NT), 0, stream, L.recs, 0, L.cb, L.gqueue,      \
                     (unsigned)L.qcap, L.counter, L.redo, bt);                                                    \
  }                                                                                                                \
  if (clip_fast)                                                                                                   \
    hipLaunchKernelGGL((nms_drain_kernel<GEOM, LABEL, true>), dgrid, dim3(256), 0, stream, L.recs, 0, L.cb,        \
                       iou_thr, L.gqueue, (unsigned)L.qcap, L.counter, L.redo, L.mask, L.nz, bt,                   \
                       use_p ? L.ps.perm : (const unsigned short*)nullptr, L.ps.stride);                           \
  else                                                                                                             \
    hipLaunchKernelGGL((nms_drain_kernel<GEOM, LABEL, false>), dgrid, dim3(256), 0, stream, L.recs, 0, L.cb,       \
                       iou_thr, L.gqueue, (unsigned)L.qcap, L.counter, L.redo, L.mask, L.nz, bt,                   \
                       use_p ? L.ps.perm : (const unsigned short*)nullptr, L.ps.stride)
  if (geom == 1) { R3_MC(1, false, maxc); }
  else if (geom == 3) { R3_MC(3, false, scale_parts > 0 ? maxc : L.extent); }
  else { R3_MC(2, true, (const float*)nullptr); }
#undef R3_MC
  counts = L.ccounts;
  {
    // one reducer workgroup per (image, label group = label mod 16) when the pool is small enough for its LDS; the
    // kernel itself falls back to one workgroup per image when the drain saw an edge between two groups
    const int nms_impl = nms_impl0;
    const int groups = (cap <= RG_MAXN && nms_impl != 2) ? RG_GROUPS : 1;
    if (cap <= WALK_MAXN && nms_impl == 4) {
      // (round 5, measured and NOT the default: one wavefront per (image, label group) walks its rows in score order --
      // 13.1 us against the round reducer's 9.7 at n = 2000, 54.6 against 16.4 at 8576, 64 against 65 on the random-weight
      // model's own pool (chains 90 deep, 10 % of the rows with > 32 suppressors): a block of 64 rows costs the walk three
      // dependent memory round trips, which the round reducer's 1024 threads overlap; profiles/r05_nms_reducer_ab.txt)
      hipLaunchKernelGGL(nms_reduce_walk_kernel, dim3(RG_GROUPS, 1, B), dim3(64), 0, stream, L.mask, L.nz, L.cb,
                         L.counter, L.kbits, cbq, L.svals, geom == 1 ? L.fbits : (u64*)nullptr, bt);
    } else {
    const size_t lds = reduce_groups_lds_bytes(cap, L.cb, groups > 1);
    // the default cap on dynamic LDS is 64 KB; the opt-in is per device: everything the CU has beyond the kernel's own
    // static LDS (read from the code object: it changes with the code)
    static R3DeviceOnce raised;
    static size_t dyn_max = 0;             // (the same on every device: one code object)
    constexpr size_t DYN_DEFAULT = 44 * 1024;  // what fits the default cap whatever the static part is (< 20 KB)
    if (lds > DYN_DEFAULT && raised.first()) {
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(nms_reduce_groups_kernel));
      dyn_max = 160 * 1024 - ((fa.sharedSizeBytes + 1023) & ~(size_t)1023);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_reduce_groups_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max);
      (void)hipGetLastError();
    }
    if (lds > DYN_DEFAULT && lds > dyn_max) return -1;  // (cap < 65536: 124 KB at most with grouping, 106 KB without)
    hipLaunchKernelGGL(nms_reduce_groups_kernel, dim3(groups, 1, B), dim3(RTHREADS), lds, stream, L.mask, L.nz, L.cb,
                       L.counter, L.kbits, cbq, L.svals, geom == 1 ? L.fbits : (u64*)nullptr, bt, g_nms_stamps);
    }
  }
  if (geom == 1)
    hipLaunchKernelGGL(mc_finish_kernel, dim3((cap + 1023) / 1024, B), dim3(1024), 0, stream, boxes, n, cand_row,
                       cand_label, cand_score, S, counts, L.fbits, cbq, out_cap, mo);
  else
    hipLaunchKernelGGL(mc_finish_score_chip_kernel, dim3((cap + 1023) / 1024, B), dim3(1024), 0, stream, boxes, n, cand_row,
                       cand_label, cand_score, S, L.svals, counts, L.kbits, cbq, (size_t)cap,
                       geom == 3 ? L.dead : (const uint8_t*)nullptr, out_cap, mo);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// batched_rnms (ops/rnms/rnms_wrapper.py:34-69) as ONE call on raw inputs: the candidate arrays of the batched
// pipeline (row = 0..n-1, label, count = n) and the wrapper's `bboxes.max()` are produced by one small kernel
// instead of five framework launches (arange, zeros, full, to(int32), max), then the B = 1 pipeline runs.
namespace {
constexpr int RB_WGS = 16;  // at most; one per 2048 rows (one workgroup: 3.7 us at n = 2000, 7.3 at 8576; five there: 5.0)
__global__ __launch_bounds__(1024) void rnms_begin_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ inds,
                                                          int n, int* __restrict__ row, int* __restrict__ lab,
                                                          int* __restrict__ cnt, float* __restrict__ part, int v3) {
  // part: v1 -- gridDim.x partial maxima of bboxes.max() (a workgroup that saw a NaN writes NaN: torch.max propagates it);
  // v3 -- as many minima, then maxima, of the circumscribed horizontal boxes (mc_hbb_extent_kernel's arithmetic over
  // rows 0 .. n-1); mc_sort_prepare_kernel reduces them (sparts = gridDim.x)
  __shared__ float s0[16], s1[16];
  __shared__ int anynan[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
  const int G = (int)gridDim.x, stride = G * 1024;
  for (int i = g * 1024 + tid; i < n; i += stride) {
    row[i] = i;
    lab[i] = inds ? (int)inds[i] : 0;
  }
  if (g == 0 && tid == 0) cnt[0] = n;
  float lo = INFINITY, hi = -INFINITY;
  int bad = 0;
  if (v3) {
    for (int i0 = g * 1024 + tid; i0 < n; i0 += 4 * stride) {
      float b[4][5];
#pragma unroll
      for (int u = 0; u < 4; u++) {  // (four boxes' loads in flight)
        const int i = min(i0 + u * stride, n - 1);
#pragma unroll
        for (int k = 0; k < 5; k++) b[u][k] = boxes[(size_t)i * 5 + k];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (i0 + u * stride >= n) continue;
        const float cs = cosf(b[u][4]), sn = sinf(b[u][4]);
        const float xb = fabsf(b[u][2] / 2 * cs) + fabsf(b[u][3] / 2 * sn);
        const float yb = fabsf(b[u][2] / 2 * sn) + fabsf(b[u][3] / 2 * cs);
        lo = fminf(lo, fminf(b[u][0] - xb, b[u][1] - yb));
        hi = fmaxf(hi, fmaxf(b[u][0] + xb, b[u][1] + yb));
      }
    }
  } else {
    hi = -3.4028235e38f;
    for (int i0 = g * 1024 + tid; i0 < 5 * n; i0 += 8 * stride) {  // torch.max over all five columns
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = boxes[min(i0 + u * stride, 5 * n - 1)];  // eight loads in flight
#pragma unroll
      for (int u = 0; u < 8; u++) {
        bad |= v[u] != v[u];
        hi = fmaxf(hi, v[u]);
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d));
    hi = fmaxf(hi, __shfl_xor(hi, d));
    bad |= __shfl_xor(bad, d);
  }
  if (lane == 0) {
    s0[wave] = lo;
    s1[wave] = hi;
    anynan[wave] = bad;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; w++) {
      lo = fminf(lo, s0[w]);
      hi = fmaxf(hi, s1[w]);
      bad |= anynan[w];
    }
    if (v3) {
      part[g] = lo;
      part[G + g] = hi;
    } else {
      part[g] = bad ? __builtin_nanf("") : hi;
    }
  }
}

struct RnmsLayout {
  int *row, *lab, *rank, *cnt;
  float* maxc;
  int64_t* labels_out;
  void* mc;
};
inline size_t rnms_layout(int n, void* ws, RnmsLayout* L) {
  const int cap = (n + 63) / 64 * 64;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return p ? p + o : nullptr; };
  char* row = take((size_t)n * 4);
  char* lab = take((size_t)n * 4);
  char* rank = take((size_t)n * 4);
  char* cnt = take(4);
  char* mx = take(4 * 2 * RB_WGS);  // (the begin kernel's partial results)
  char* lo = take((size_t)n * 8);
  char* mc = take(r3k_mcnms_workspace_bytes(1, cap));
  if (L) {
    L->row = (int*)row; L->lab = (int*)lab; L->rank = (int*)rank; L->cnt = (int*)cnt; L->maxc = (float*)mx;
    L->labels_out = (int64_t*)lo; L->mc = mc;
  }
  return off + 256;
}
}  // namespace

size_t r3k_batched_rnms_workspace_bytes(int n) { return n <= 0 ? 256 : rnms_layout(n, nullptr, nullptr); }

// geom 1: batched_rnms (keep ascending); geom 3: obb_batched_nms (nms_rotated_wrapper.py:78-98: offsets from the
// circumscribed horizontal boxes' extent, thin boxes never kept, keep in score order)
int r3k_batched_nms(int geom, const float* boxes, const float* scores, const int64_t* inds, int n, float thr, void* ws,
                    size_t ws_bytes, float* dets_out, int64_t* keep_out, int32_t* kept_out, hipStream_t stream) {
  if (geom != 1 && geom != 3) return -1;
  if (n <= 0 || n > 65472 || !boxes || !scores || !ws || !dets_out || !keep_out || !kept_out || !(thr >= 0.f)) return -1;
  if (ws_bytes < r3k_batched_rnms_workspace_bytes(n)) return -3;
  RnmsLayout L;
  rnms_layout(n, ws, &L);
  const int cap = (n + 63) / 64 * 64;
  const int bwgs = std::min(RB_WGS, (n + 2047) / 2048);
  hipLaunchKernelGGL(rnms_begin_kernel, dim3(bwgs), dim3(1024), 0, stream, boxes, inds, n, L.row, L.lab, L.cnt, L.maxc,
                     geom == 3 ? 1 : 0);
  return r3k_mcnms_run(geom, boxes, 1, n, 1, L.row, L.lab, scores, L.rank, L.cnt, L.maxc, cap, thr, n, L.mc,
                       r3k_mcnms_workspace_bytes(1, cap), dets_out, L.labels_out, keep_out, kept_out, stream, nullptr,
                       bwgs);
}


int r3k_mcnms_v1(const float* boxes, int B, int n, int K, const int* cand_row, const int* cand_label,
                 const float* cand_score, int* cand_rank, const int* counts, const float* maxc, int cap,
                 float iou_thr, int out_cap, void* ws, size_t ws_bytes, float* dets_out, int64_t* labels_out,
                 int64_t* keep_idx_out, int32_t* counts_out, hipStream_t stream) {
  return r3k_mcnms_run(1, boxes, B, n, K, cand_row, cand_label, cand_score, cand_rank, counts, maxc, cap, iou_thr,
                       out_cap, ws, ws_bytes, dets_out, labels_out, keep_idx_out, counts_out, stream, nullptr);
}

// greedy reduction of a dense upper-triangle mask (rows in score order); used by poly_nms
int r3k_nms_reduce_dense(const unsigned long long* mask, int n, int cb, const int64_t* order, int64_t* keep_out,
                         int32_t* count_out, hipStream_t stream) {
  const size_t lds = (size_t)(cb + 1) * sizeof(u64);
  if (lds > 64 * 1024) return -1;
  hipLaunchKernelGGL(nms_reduce_dense_kernel, dim3(1), dim3(1024), lds, stream, mask, n, cb, order, keep_out, count_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t r3k_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return layout(n, nullptr, nullptr);
}

int r3k_nms(int geom, const float* dets, int det_stride, const int64_t* labels,
            const int64_t* order, int n, float thr, int sort_ascending, void* ws, size_t ws_bytes,
            int64_t* keep_out, int32_t* count_out, hipStream_t stream) {
  if (n < 0 || !count_out) return -1;
  if (n == 0) {
    return r3k_zero_async(count_out, sizeof(int32_t), stream);
  }
  if (!dets || !order || !ws || !keep_out) return -1;
  if (ws_bytes < r3k_nms_workspace_bytes(n)) return -3;
  Layout L;
  layout(n, ws, &L);
  int rc;
  if (geom == 1) rc = run_nms<1, false>(dets, det_stride, nullptr, order, n, thr, L, keep_out, count_out, stream);
  else if (geom == 2 && labels) rc = run_nms<2, true>(dets, det_stride, labels, order, n, thr, L, keep_out, count_out, stream);
  else if (geom == 2) rc = run_nms<2, false>(dets, det_stride, nullptr, order, n, thr, L, keep_out, count_out, stream);
  else if (geom == 3 && labels) rc = run_nms<3, true>(dets, det_stride, labels, order, n, thr, L, keep_out, count_out, stream);
  else if (geom == 3) rc = run_nms<3, false>(dets, det_stride, nullptr, order, n, thr, L, keep_out, count_out, stream);
  else return -1;
  if (rc) return rc;
  if (sort_ascending)
    hipLaunchKernelGGL(nms_ascending_kernel, dim3(1), dim3(1024), 0, stream, n, L.flags, keep_out, count_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
