// r3_pool.hip -- the pre-NMS pool of one pyramid level for a whole batch (second half):
// what RAnchorHead._get_bboxes_single does per image and level before multiclass_nms_rotated
// (models/dense_heads/rotate_anchor_head.py:626-673; rois-as-anchors variant
// rotate_retina_refine_head.py:147-196):
//     scores = cls_score.permute(1, 2, 0).reshape(-1, C).sigmoid()
//     if nms_pre < scores.shape[0]:  topk over scores.max(dim=1) -> gather anchors / deltas / scores
//     bboxes = bbox_coder.decode(anchors, bbox_pred, max_shape=img_shape)
// In torch that is a sigmoid + max + topk (a radix sort, ~75 us per level on MI355X) + three gathers + ~10
// elementwise launches per level.  Here, per level:
//   keys   : one thread per (position, anchor): key = sigmoid(max_c logit) (= max_c sigmoid, monotone), order
//            preserving u32, read through the head output's strides (NCHW or channels_last, no copy);
//   select : the threshold's 12-bit bin from the key launch's histogram; the keys above it and the keys inside it
//            go to two per-image lists, 16 384 keys per workgroup (a flat score distribution with more than 8192
//            keys in that bin: ONE workgroup per image finds the exact k-th largest key by 4-bit radix passes whose
//            per-thread digit counts live in two byte-packed 64-bit registers);
//   emit   : rank by counting over the chip (score desc, ties by ascending row), then decode + sigmoid of the
//            winners' C logits straight into the pool arrays boxes (N, n, 5) / scores (N, n, C + 1) that
//            r3det_mcnms_select reads;
//   levels with at most nms_pre rows keep their order: one thread per row, no selection.
// Arithmetic as the torch ops it replaces: sigmoid = 1 / (1 + expf(-x)), delta2bbox_v1 with the centre clamp
// to the image (delta_xywha_rbbox_coder.py:142-211), IEEE add / mul (-ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "r3_kernels.h"

namespace {

typedef unsigned long long u64;

struct PStrides {
  long long n, c, h, w;  // in elements
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ unsigned pool_key(float f) {  // monotone: a < b  <=>  key(a) < key(b)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// row i of a level <-> (position p = i / A, anchor a = i % A), as the reference's permute + reshape orders them
constexpr int PH_BINS = 4096;  // histogram of the keys' top 12 bits (= the first three 4-bit digits of the select)

// All levels of a head in ONE set of launches (r3k_levels_pool): a launch's workgroups find their level from the
// block ranges below.  Per level and launch the kernels are short (5 - 11 us) and mostly ramp: five levels one after
// the other were 11 launches and 71 us per R3Det step, the same work in 3 launches + one memset is bounded by the
// largest level.
constexpr int PL_MAX = R3K_POOL_MAX_LEVELS;
struct PLevel {
  const float* cls;
  const float* reg;
  const float* anchors;
  PStrides sc, sr;
  int A, H, W, Lpad;
  int select;      // 1: the level is cut to its k best rows; 0: all rows in their order
  int row_offset;  // first pool row of the level
  int kblk0;       // first workgroup of the level in the key launch (select levels)
  int eblk0;       // first workgroup of the level in the emit launch
  int sel_index;   // index among the select levels (blockIdx.y of the select launch)
  unsigned* keys;
  unsigned* hist;
  unsigned long long* glist;
  unsigned long long* clist;
  int* meta;
};
struct PLevels {
  int count, C, per_image, k, kb;
  float max_ratio, clamp_x, clamp_y;
  float* boxes;
  float* scores;
  int pool_rows;
  PLevel lv[PL_MAX];
};

// keys + per-image histogram of their top 12 bits (LDS histogram per workgroup, its non-empty bins added to the
// global one: scores crowd into a few dozen bins, one global atomic per key would serialise on them).  `hist`
// zeroed by the caller.
__global__ __launch_bounds__(256) void pool_keys_kernel(const PLevels P) {
  __shared__ unsigned lh[PH_BINS];
  int li = 0;
  for (int l = 0; l < P.count; l++)
    if (P.lv[l].select && (int)blockIdx.x >= P.lv[l].kblk0) li = l;
  const PLevel& V = P.lv[li];
  const float* __restrict__ cls = V.cls;
  const PStrides sc = V.sc;
  const int A = V.A, C = P.C, H = V.H, W = V.W, Lpad = V.Lpad;
  unsigned* __restrict__ keys = V.keys;
  unsigned* __restrict__ hist = V.hist;
  const int HW = H * W, L = HW * A;
  const int n = blockIdx.y;
  const int t = ((int)blockIdx.x - V.kblk0) * 256 + threadIdx.x;
  for (int b = threadIdx.x; b < PH_BINS; b += 256) lh[b] = 0u;
  __syncthreads();
  if (t >= L) {
    if (t < Lpad) keys[(size_t)n * Lpad + t] = 0u;  // padding: below every real key (a sigmoid is positive)
  } else {
    // NCHW heads: neighbouring threads take neighbouring positions of one anchor (coalesced planes);
    // channels_last heads: neighbouring threads take neighbouring anchors of one position (contiguous logits)
    int p, a;
    if (sc.c == 1) { p = t / A; a = t - p * A; }
    else { a = t / HW; p = t - a * HW; }
    const int h = p / W, w = p - h * W;
    const float* cb = cls + n * sc.n + h * sc.h + w * sc.w + (long long)(a * C) * sc.c;
    float m = -INFINITY;
    for (int c = 0; c < C; c++) m = fmaxf(m, cb[(long long)c * sc.c]);
    const unsigned key = pool_key(sigmoidf(m));
    keys[(size_t)n * Lpad + (size_t)p * A + a] = key;
    if (key != 0u) atomicAdd(&lh[key >> 20], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < PH_BINS; b += 256) {
    const unsigned c = lh[b];
    if (c) atomicAdd(&hist[(size_t)n * PH_BINS + b], c);
  }
}

// one pool row: decode + the C class scores of level row i
__device__ __forceinline__ void pool_emit(const float* __restrict__ cls, const PStrides& sc,
                                          const float* __restrict__ reg, const PStrides& sr,
                                          const float* __restrict__ anchors, int per_image, int n, int i, int A,
                                          int C, int H, int W, float max_ratio, float clamp_x, float clamp_y,
                                          float* __restrict__ box_out, float* __restrict__ score_out) {
  const int p = i / A, a = i - p * A;
  const int h = p / W, w = p - h * W;
  const float* rb = reg + n * sr.n + h * sr.h + w * sr.w + (long long)(a * 5) * sr.c;
  const float d0 = rb[0], d1 = rb[sr.c], d2 = rb[2 * sr.c], d3 = rb[3 * sr.c], d4 = rb[4 * sr.c];
  const float* an = anchors + ((size_t)(per_image ? n : 0) * H * W * A + (size_t)i) * 5;
  const float ax = an[0], ay = an[1], aw = an[2], ah = an[3], aa = an[4];
  const float dw = fminf(fmaxf(d2, -max_ratio), max_ratio);
  const float dh = fminf(fmaxf(d3, -max_ratio), max_ratio);
  float gx = ax + aw * d0, gy = ay + ah * d1;
  if (clamp_x >= 0.f) {  // max_shape given: centres clamped to [0, W_img - 1] x [0, H_img - 1]
    gx = fminf(fmaxf(gx, 0.f), clamp_x);
    gy = fminf(fmaxf(gy, 0.f), clamp_y);
  }
  box_out[0] = gx;
  box_out[1] = gy;
  box_out[2] = aw * expf(dw);
  box_out[3] = ah * expf(dh);
  box_out[4] = aa + d4;
  const float* cb = cls + n * sc.n + h * sc.h + w * sc.w + (long long)(a * C) * sc.c;
  for (int c = 0; c < C; c++) score_out[c] = sigmoidf(cb[(long long)c * sc.c]);
  score_out[C] = 0.f;  // the background column multiclass_nms_rotated drops
}

// a level that is not cut: one thread per row, rows in their order
__device__ __forceinline__ void pool_all_body(const PLevels& P, const PLevel& V, const int blk) {
  const int L = V.H * V.W * V.A, n = blockIdx.y;
  const int i = blk * 256 + threadIdx.x;
  if (i >= L) return;
  const size_t row = (size_t)n * P.pool_rows + V.row_offset + i;
  pool_emit(V.cls, V.sc, V.reg, V.sr, V.anchors, P.per_image, n, i, V.A, P.C, V.H, V.W, P.max_ratio, P.clamp_x, P.clamp_y,
            P.boxes + row * 5, P.scores + row * (P.C + 1));
}

// Append under a predicate with ONE LDS atomic per wavefront (ballot + popcount; lane ranks by mbcnt) instead of one
// per lane: k winners appended lane by lane to one counter were k serialised same-address atomics -- most of this
// kernel's time at k = 2000 (LDS atomics retire about one lane per 3+ cycles, far slower on one address).
// Returns the slot, or -1 for lanes whose predicate is false.  Called by all active lanes of the wave together.
__device__ __forceinline__ int wave_append(int* counter, const bool pred) {
  const u64 m = __ballot(pred);
  if (m == 0) return -1;
  const int first = __ffsll((long long)m) - 1;
  int base = 0;
  if ((int)(threadIdx.x & 63) == first) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, first);
  return pred ? base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) : -1;
}

constexpr int PS_T = 1024;      // threads of the select workgroup
constexpr int PS_KMAX = 4096;   // largest nms_pre (LDS list of 8-byte entries: 32 KB)
constexpr int PS_U = 4;         // 16-byte key loads in flight per thread and step
constexpr int PS_CAND = 8192;   // (key, index) entries kept in LDS after three digits (64 KB)

// The select workgroup is alone on its CU (grid = images): what it can afford is bandwidth, not latency.  Keys are
// read as uint4, PS_U independent loads per thread and step (a first version read one key per iteration, each
// waiting for the previous: 127 us at 16 384 keys, 500 us at 147 456).
__global__ __launch_bounds__(PS_T) void pool_select_kernel(const PLevels P, u64* __restrict__ stamps = nullptr) {
  int li = 0;
  for (int l = 0; l < P.count; l++)
    if (P.lv[l].select && P.lv[l].sel_index == (int)blockIdx.y) li = l;
  const PLevel& V = P.lv[li];
  const int k = P.k, Lpad = V.Lpad;
  const unsigned* __restrict__ keys = V.keys;
  const unsigned* __restrict__ hist = V.hist;
  u64* __restrict__ glist = V.glist;
  u64* __restrict__ clist = V.clist;
  int* __restrict__ meta = V.meta;
  // (tools/probes/pool_select_probe.hip: clock stamps of workgroup 0 at the phase boundaries)
  auto stamp = [&](int i) {
    if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  // entries (key << 32) | ~index -- larger = earlier in the pool -- go to two per-image lists in the workspace:
  // `list` the keys above the threshold's 12-bit bin (all winners), `cand` the keys inside it (the first `need` of
  // them in descending order are the remaining winners).  Nothing is sorted here: the emit launch ranks by counting.
  u64* list = glist + (size_t)blockIdx.x * PS_KMAX;
  u64* cand = clist + (size_t)blockIdx.x * PS_CAND;
  __shared__ int s_ncand;
  __shared__ int wcnt[PS_T / 64][16];
  __shared__ unsigned s_prefix;
  __shared__ int s_need, s_eq_total;
  __shared__ int s_gt, s_eq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  const uint4* kn4 = reinterpret_cast<const uint4*>(keys + (size_t)n * Lpad);
  const int L4 = Lpad >> 2;
  const int steps = (L4 + PS_T * PS_U - 1) / (PS_T * PS_U);
  const int chunk = blockIdx.z;  // the split below: one step (16 384 keys) per workgroup
  if (chunk >= steps) return;

  // ---- exact k-th largest key: 8 passes over 4-bit digits, most significant first
  unsigned prefix = 0, mask = 0;
  int need = k;
  // one radix step: the per-thread digit counts `tot` -> the digit that holds the `need`-th largest key
  auto choose = [&](int (&tot)[16], const int shift) {
    // wave totals of the 16 digit counts: packed four to a 64-bit word (16-bit fields: a wave's total of one digit
    // is at most 64 x 144 keys), 24 shuffles instead of 96
    u64 pk[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
      pk[q] = (u64)(unsigned)tot[4 * q] | ((u64)(unsigned)tot[4 * q + 1] << 16) | ((u64)(unsigned)tot[4 * q + 2] << 32) |
              ((u64)(unsigned)tot[4 * q + 3] << 48);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1)
#pragma unroll
      for (int q = 0; q < 4; q++) pk[q] += __shfl_xor(pk[q], sft);
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < 16; d++) wcnt[wave][d] = (int)((pk[d >> 2] >> (16 * (d & 3))) & 0xffffULL);
    }
    __syncthreads();
    if (tid < 16) {  // digit totals in parallel (a serial walk over 16 x 16 LDS words cost 7 us per pass)
      int c = 0;
#pragma unroll
      for (int w = 0; w < PS_T / 64; w++) c += wcnt[w][tid];
      // suffix sums over the digits above this one: the digit whose range contains the `need`-th largest wins
      int above = 0;
#pragma unroll
      for (int d = 1; d < 16; d++) {
        const int o = __shfl_down(c, d, 16);
        if (tid + d < 16) above += o;
      }
      if (above < need && above + c >= need) {
        s_prefix = prefix | ((unsigned)tid << shift);
        s_need = need - above;
        s_eq_total = c;  // (after the last pass: how many keys equal the threshold)
      }
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    mask |= 15u << shift;
    __syncthreads();
    };
  // A pass over all keys costs this ONE compute unit ~20 us at 147 k keys, so: three digits from the histogram, then
  // ONE pass that sorts the keys into "above the threshold bin" / "inside it" (about 1 %) / "below".  More than
  // PS_CAND keys inside the bin (a flat score distribution): the remaining digits from global memory.
  int shift = 28;
  {
    // the first three digits come from the histogram the key kernel made (spread over the chip): bins from the top
    // down until `need` keys are covered.  Thread t owns bins 4t .. 4t+3.
    const uint4 hb = reinterpret_cast<const uint4*>(hist + (size_t)n * PH_BINS)[tid];
    const int c4[4] = {(int)hb.x, (int)hb.y, (int)hb.z, (int)hb.w};
    const int mine = c4[0] + c4[1] + c4[2] + c4[3];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wcnt[wave][0] = incl;
    __syncthreads();
    int before = incl, total = 0;
#pragma unroll
    for (int w = 0; w < PS_T / 64; w++) {
      const int o = wcnt[w][0];
      if (w < wave) before += o;
      total += o;
    }
    int above = total - before;  // keys in the bins above this thread's
#pragma unroll
    for (int b = 3; b >= 0; b--) {
      if (above < need && above + c4[b] >= need) {
        s_prefix = (unsigned)(tid * 4 + b) << 20;
        s_need = need - above;
        s_eq_total = c4[b];
      }
      above += c4[b];
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    mask = 0xfff00000u;
    shift = 16;
    __syncthreads();
  }
  stamp(1);
  const bool flat = s_eq_total > PS_CAND;  // (known from the histogram alone: every workgroup of the image agrees)
  if (flat && chunk != 0) return;
  for (; flat && shift >= 0; shift -= 4) {
    int tot[16];
#pragma unroll
    for (int d = 0; d < 16; d++) tot[d] = 0;
    u64 lo = 0, hi = 0;  // 16 byte-wide digit counters, flushed every 15 steps (15 x 16 keys < 256)
    for (int st = 0; st < steps; st++) {
      uint4 v[PS_U];
#pragma unroll
      for (int u = 0; u < PS_U; u++) {
        const int q = (st * PS_U + u) * PS_T + tid;
        v[u] = q < L4 ? kn4[q] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < PS_U; u++) {
        const unsigned kk[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned key = kk[e];
          const bool in = key != 0u && (key & mask) == prefix;
          const unsigned d = (key >> shift) & 15u;
          const u64 one = in ? (1ULL << ((d & 7u) * 8u)) : 0ULL;
          lo += (d < 8u) ? one : 0ULL;
          hi += (d < 8u) ? 0ULL : one;
        }
      }
      if ((st % 15) == 14 || st == steps - 1) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
          tot[d] += (int)((lo >> (8 * d)) & 255ULL);
          tot[8 + d] += (int)((hi >> (8 * d)) & 255ULL);
        }
        lo = hi = 0;
      }
    }
    choose(tot, shift);
  }
  if (!flat) {
    // The keys above the 12-bit prefix go to `list` (k - need of them: winners whatever the remaining digits say),
    // the keys matching it to `cand`.  No more digits and no sort: the emit launch ranks every entry by counting,
    // spread over the chip -- in one workgroup the five LDS digit passes took 18 us and the bitonic sort of the k
    // winners 27 us of a 56 us kernel (tools/probes/pool_select_probe.hip).  And the pass itself is spread over the
    // chip too: a workgroup takes ONE step of 16 384 keys, counts its entries (ballots), reserves its places in the
    // two lists with one global atomic each and writes them -- any order will do, the entries carry their index.
    // (One workgroup per image walked all keys: 52 us at RRetinaNet's 147 456 rows.)
    uint4 v[PS_U];
#pragma unroll
    for (int u = 0; u < PS_U; u++) {
      const int q = (chunk * PS_U + u) * PS_T + tid;
      v[u] = q < L4 ? kn4[q] : make_uint4(0u, 0u, 0u, 0u);
    }
    int wc = 0, wg = 0;  // this wave's entries (wave-uniform)
#pragma unroll
    for (int u = 0; u < PS_U; u++) {
      const unsigned kk[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const unsigned key = kk[e];
        const bool is_c = key != 0u && (key & mask) == prefix, is_g = !is_c && (key & mask) > prefix;
        wc += __popcll(__ballot(is_c));
        wg += __popcll(__ballot(is_g));
      }
    }
    if (lane == 0) {
      wcnt[wave][0] = wc;
      wcnt[wave][1] = wg;
    }
    __syncthreads();
    int bc = 0, bg = 0, tc = 0, tg = 0;
#pragma unroll
    for (int w = 0; w < PS_T / 64; w++) {
      const int c = wcnt[w][0], g = wcnt[w][1];
      if (w < wave) { bc += c; bg += g; }
      tc += c;
      tg += g;
    }
    if (tid == 0) {
      s_ncand = tc ? atomicAdd(&meta[n * 4 + 1], tc) : 0;
      s_gt = tg ? atomicAdd(&meta[n * 4 + 3], tg) : 0;
      if (chunk == 0) {
        meta[n * 4 + 0] = k - need;  // entries of `list` (meta[3]: the cursor that fills it)
        meta[n * 4 + 2] = need;      // winners among the entries of `cand` (meta[1]: their number)
      }
    }
    __syncthreads();
    int pc = s_ncand + bc, pg = s_gt + bg;
#pragma unroll
    for (int u = 0; u < PS_U; u++) {
      const unsigned kk[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      const int q = (chunk * PS_U + u) * PS_T + tid;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const unsigned key = kk[e];
        const u64 ent = ((u64)key << 32) | (u64)(0xffffffffu - (unsigned)(q * 4 + e));
        const bool is_c = key != 0u && (key & mask) == prefix, is_g = !is_c && (key & mask) > prefix;
        const u64 mc = __ballot(is_c), mg = __ballot(is_g);
        const int rc_ = pc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mc, 0u));
        const int rg_ = pg + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, 0u));
        if (is_c && rc_ < PS_CAND) cand[rc_] = ent;
        if (is_g && rg_ < PS_KMAX) list[rg_] = ent;
        pc += __popcll(mc);
        pg += __popcll(mg);
      }
    }
    stamp(2);
    stamp(3);
    stamp(4);
  } else {
    // more survivors than the LDS copy holds (a flat score distribution): the remaining digits were taken from
    // global memory above; collection from global memory, then the sort
    const unsigned T = prefix;  // the k-th largest key; `need` of the keys equal to it are taken (lowest indices)
    const int n_gt = k - need;
    const bool all_eq = s_eq_total == need;  // every key equal to T is a winner: no index order needed among them
    if (tid == 0) { s_gt = 0; s_eq = 0; }
    __syncthreads();
    for (int st = 0; st < steps; st++) {
      uint4 v[PS_U];
#pragma unroll
      for (int u = 0; u < PS_U; u++) {
        const int q = (st * PS_U + u) * PS_T + tid;
        v[u] = q < L4 ? kn4[q] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < PS_U; u++) {
        const unsigned kk[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const int q = (st * PS_U + u) * PS_T + tid;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned key = kk[e];
          const unsigned i = (unsigned)(q * 4 + e);
          const int pg = wave_append(&s_gt, key > T);
          if (pg >= 0) list[pg] = ((u64)key << 32) | (u64)(0xffffffffu - i);
          const int pe = wave_append(&s_eq, all_eq && key == T);
          if (pe >= 0) list[n_gt + pe] = ((u64)key << 32) | (u64)(0xffffffffu - i);
        }
      }
    }
    if (!all_eq) {
      // ties straddle the threshold: the first `need` keys equal to T in index order.  Every thread counts the
      // equal keys of a contiguous index range, an exclusive scan turns the counts into ranks.
      const unsigned* kn = keys + (size_t)n * Lpad;
      const int per = (Lpad + PS_T - 1) / PS_T;
      const int lo_i = min(tid * per, Lpad), hi_i = min(lo_i + per, Lpad);
      int ceq = 0;
      for (int i = lo_i; i < hi_i; i++) ceq += kn[i] == T;
      int incl = ceq;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      __syncthreads();
      if (lane == 63) wcnt[wave][0] = incl;
      __syncthreads();
      int base_eq = incl - ceq;
#pragma unroll
      for (int w = 0; w < PS_T / 64; w++)
        if (w < wave) base_eq += wcnt[w][0];
      for (int i = lo_i; i < hi_i && base_eq < need; i++) {
        if (kn[i] == T) {
          list[n_gt + base_eq] = ((u64)T << 32) | (u64)(0xffffffffu - (unsigned)i);
          base_eq++;
        }
      }
    }
    __syncthreads();
    stamp(4);
    if (tid == 0) {  // all k winners are in `list`
      meta[blockIdx.x * 4 + 0] = k;
      meta[blockIdx.x * 4 + 1] = 0;
      meta[blockIdx.x * 4 + 2] = 0;
    }
  }
  stamp(5);
  stamp(6);
}

// Rank by counting + decode, over the chip.  A workgroup owns PE_C = 32 entries of one list: workgroups
// 0 .. kb-1 of an image entries of `list`, the others entries of `cand`.  An entry's pool row is the number of larger
// entries of its own list (every `list` entry is above every `cand` entry; entries are distinct: they carry their
// index) -- for `cand` behind the n_gt rows of `list`, and only if that rank is below `need`.  PE_P = 8 threads share
// an entry's compares, each an eighth of every 1024-entry LDS tile (two u64 per half-wave broadcast read); k = 2000:
// 250 compares per thread.  (One thread per entry, 256 entries per workgroup: 8 workgroups per image, +20 us.)
constexpr int PE_TILE = 1024;
constexpr int PE_C = 32, PE_P = 8;

__global__ __launch_bounds__(256) void pool_emit_kernel(const PLevels P) {
  __shared__ __attribute__((aligned(16))) u64 tile[PE_TILE];
  __shared__ int partial[PE_P][PE_C];
  int li = 0;
  for (int l = 0; l < P.count; l++)
    if ((int)blockIdx.x >= P.lv[l].eblk0) li = l;
  const PLevel& V = P.lv[li];
  const int blk = (int)blockIdx.x - V.eblk0;
  if (!V.select) {
    pool_all_body(P, V, blk);
    return;
  }
  const int k = P.k, kb = P.kb;
  const int n = blockIdx.y, tid = threadIdx.x;
  const int* __restrict__ meta = V.meta;
  const int n_gt = meta[n * 4 + 0], ncand = meta[n * 4 + 1], need = meta[n * 4 + 2];
  const bool in_list = blk < kb;
  const u64* src = in_list ? V.glist + (size_t)n * PS_KMAX : V.clist + (size_t)n * PS_CAND;
  const int cnt = in_list ? n_gt : ncand;
  const int i0 = (in_list ? blk : blk - kb) * PE_C;
  if (i0 >= cnt) return;
  const int ci = tid & (PE_C - 1), part = tid / PE_C;
  const int i = i0 + ci;
  const u64 mine = i < cnt ? src[i] : ~0ULL;
  int rank = 0;
  u64 nxt[PE_TILE / 256];
#pragma unroll
  for (int u = 0; u < PE_TILE / 256; u++) nxt[u] = (u * 256 + tid) < cnt ? src[u * 256 + tid] : 0ULL;
  constexpr int SL2 = PE_TILE / PE_P / 2;  // ulonglong2 reads of a thread per tile
  const ulonglong2* t2 = reinterpret_cast<const ulonglong2*>(tile) + part * SL2;
  for (int j0 = 0; j0 < cnt; j0 += PE_TILE) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PE_TILE / 256; u++) tile[u * 256 + tid] = nxt[u];   // (0 pads: below every entry)
#pragma unroll
    for (int u = 0; u < PE_TILE / 256; u++) {
      const int j = j0 + PE_TILE + u * 256 + tid;
      nxt[u] = j < cnt ? src[j] : 0ULL;
    }
    __syncthreads();
#pragma unroll 8
    for (int q = 0; q < SL2; q++) {
      const ulonglong2 e = t2[q];
      rank += (e.x > mine) + (e.y > mine);
    }
  }
  partial[part][ci] = rank;
  __syncthreads();
  if (part != 0 || i >= cnt) return;
  rank = 0;
#pragma unroll
  for (int q = 0; q < PE_P; q++) rank += partial[q][ci];
  if (!in_list) {
    if (rank >= need) return;
    rank += n_gt;
  }
  if (rank >= k) return;  // (cannot happen for consistent lists; keeps a corrupted workspace inside the pool)
  const size_t row = (size_t)n * P.pool_rows + V.row_offset + rank;
  pool_emit(V.cls, V.sc, V.reg, V.sr, V.anchors, P.per_image, n, (int)(0xffffffffu - (unsigned)(mine & 0xffffffffULL)), V.A,
            P.C, V.H, V.W, P.max_ratio, P.clamp_x, P.clamp_y, P.boxes + row * 5, P.scores + row * (P.C + 1));
}

}  // namespace

static size_t level_ws_bytes(int N, long long L) {
  const long long Lpad = (L + 3) / 4 * 4;
  // keys | the two entry lists + 4 ints per image | histogram  (768: alignment slack of the three parts)
  return (size_t)N * Lpad * sizeof(unsigned) + (size_t)N * ((PS_KMAX + PS_CAND) * sizeof(u64) + 16) + 768 +
         (size_t)N * PH_BINS * 4;
}

size_t r3k_levels_pool_workspace_bytes(int nlevels, int N, const int* A, const int* H, const int* W, int nms_pre) {
  if (nlevels <= 0 || nlevels > PL_MAX || N <= 0 || !A || !H || !W) return 0;
  size_t total = 0;
  for (int l = 0; l < nlevels; l++) {
    if (A[l] <= 0 || H[l] <= 0 || W[l] <= 0) return 0;
    const long long L = (long long)H[l] * W[l] * A[l];
    if (nms_pre > 0 && nms_pre < L) total += level_ws_bytes(N, L);
  }
  return total;
}

size_t r3k_level_pool_workspace_bytes(int N, int A, int H, int W, int nms_pre) {
  return r3k_levels_pool_workspace_bytes(1, N, &A, &H, &W, nms_pre);
}

int r3k_levels_pool(int nlevels, const float* const* cls, const long long* cls_strides, const float* const* reg,
                    const long long* reg_strides, const float* const* anchors, int per_image, int N, const int* A, int C,
                    const int* H, const int* W, int nms_pre, float max_ratio, float clamp_x, float clamp_y, float* boxes,
                    float* scores, int pool_rows, int row_offset, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (nlevels <= 0 || nlevels > PL_MAX || N <= 0 || C <= 0 || !cls || !cls_strides || !reg || !reg_strides || !anchors ||
      !A || !H || !W || !boxes || !scores || pool_rows <= 0 || row_offset < 0)
    return -1;
  PLevels P;
  P.count = nlevels;
  P.C = C;
  P.per_image = per_image;
  P.k = nms_pre;
  P.kb = nms_pre > 0 ? (nms_pre + PE_C - 1) / PE_C : 0;
  P.max_ratio = max_ratio;
  P.clamp_x = clamp_x;
  P.clamp_y = clamp_y;
  P.boxes = boxes;
  P.scores = scores;
  P.pool_rows = pool_rows;
  int nsel = 0;
  long long kblocks = 0, eblocks = 0, off = row_offset;
  for (int l = 0; l < nlevels; l++) {
    PLevel& V = P.lv[l];
    if (A[l] <= 0 || H[l] <= 0 || W[l] <= 0 || !cls[l] || !reg[l] || !anchors[l]) return -1;
    const long long L = (long long)H[l] * W[l] * A[l];
    if (L > 0x7fffffffLL / 8) return -1;
    V.cls = cls[l];
    V.reg = reg[l];
    V.anchors = anchors[l];
    V.sc = PStrides{cls_strides[4 * l], cls_strides[4 * l + 1], cls_strides[4 * l + 2], cls_strides[4 * l + 3]};
    V.sr = PStrides{reg_strides[4 * l], reg_strides[4 * l + 1], reg_strides[4 * l + 2], reg_strides[4 * l + 3]};
    V.A = A[l];
    V.H = H[l];
    V.W = W[l];
    V.Lpad = (int)((L + 3) / 4 * 4);
    V.select = nms_pre > 0 && nms_pre < L;
    V.row_offset = (int)off;
    V.kblk0 = (int)kblocks;
    V.eblk0 = (int)eblocks;
    V.sel_index = nsel;
    V.keys = nullptr;
    V.hist = nullptr;
    V.glist = V.clist = nullptr;
    V.meta = nullptr;
    if (V.select) {
      if (nms_pre > PS_KMAX || L > 1000000) return -1;  // (the select workgroup packs per-wave digit counts in 16 bits)
      nsel++;
      kblocks += (V.Lpad + 255) / 256;
      eblocks += P.kb + PS_CAND / PE_C;
      off += nms_pre;
    } else {
      eblocks += (L + 255) / 256;
      off += L;
    }
    if (off > pool_rows || eblocks > 0x7fffffffLL) return -1;
  }
  if (nsel) {
    if (!ws || ws_bytes < r3k_levels_pool_workspace_bytes(nlevels, N, A, H, W, nms_pre)) return -3;
    if (reinterpret_cast<uintptr_t>(ws) & 15) return -1;
    // all histograms and list cursors first (one memset), then per select level: keys | list | cand
    char* p = (char*)ws;
    const size_t hist_bytes = (size_t)N * PH_BINS * 4, meta_bytes = ((size_t)N * 16 + 255) & ~(size_t)255;
    char* q = p + (size_t)nsel * (hist_bytes + meta_bytes);
    int max_steps = 1;
    for (int l = 0; l < nlevels; l++) {
      PLevel& V = P.lv[l];
      if (!V.select) continue;
      V.hist = (unsigned*)(p + (size_t)V.sel_index * hist_bytes);
      V.meta = (int*)(p + (size_t)nsel * hist_bytes + (size_t)V.sel_index * meta_bytes);
      V.keys = (unsigned*)q;
      q += ((size_t)N * V.Lpad * sizeof(unsigned) + 255) & ~(size_t)255;
      V.glist = (u64*)q;
      V.clist = V.glist + (size_t)N * PS_KMAX;
      q = (char*)(V.clist + (size_t)N * PS_CAND);
      const int steps = ((V.Lpad >> 2) + PS_T * PS_U - 1) / (PS_T * PS_U);
      if (steps > max_steps) max_steps = steps;
    }
    if (r3k_zero_async(p, (size_t)nsel * (hist_bytes + meta_bytes), stream) != 0) return -2;  // (a kernel, not a memset node: r3_kernels.h)
    hipLaunchKernelGGL(pool_keys_kernel, dim3((unsigned)kblocks, N), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(pool_select_kernel, dim3(N, nsel, max_steps), dim3(PS_T), 0, stream, P, (u64*)nullptr);
  }
  hipLaunchKernelGGL(pool_emit_kernel, dim3((unsigned)eblocks, N), dim3(256), 0, stream, P);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int r3k_level_pool(const float* cls, const long long* cls_strides, const float* reg, const long long* reg_strides,
                   const float* anchors, int per_image, int N, int A, int C, int H, int W, int nms_pre, float max_ratio,
                   float clamp_x, float clamp_y, float* boxes, float* scores, int pool_rows, int row_offset, void* ws,
                   size_t ws_bytes, hipStream_t stream) {
  if (!cls || !reg || !anchors) return -1;
  return r3k_levels_pool(1, &cls, cls_strides, &reg, reg_strides, &anchors, per_image, N, &A, C, &H, &W, nms_pre, max_ratio,
                         clamp_x, clamp_y, boxes, scores, pool_rows, row_offset, ws, ws_bytes, stream);
}
