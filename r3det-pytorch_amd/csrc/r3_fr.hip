// r3_fr.hip -- Feature Refinement sampler (rotated feature-align), forward and backward.
//
// Replaces feature_refine_forward_kernel / feature_refine_backward_kernel
// (fr/src/feature_refine_kernel.cu:112-230).  The reference runs one thread per output
// ELEMENT (n,c,h,w) and re-derives the sample point from the 20-byte box for each of the
// C channels; here the box -> (tap offsets, bilinear weights) conversion is per POSITION.
//
// Because the reference samples row <- x_ctr*scale, column <- y_ctr*scale
// (feature_refine_kernel.cu:131-132) the gather of a well-behaved box field is a TRANSPOSE
// of the plane: adjacent lanes hit adjacent rows, i.e. a different cache line per lane.
// Staging the (n,c) plane in LDS (row pitch W+1) turns that into cheap LDS reads and HBM sees
// exactly one read and one write per element (PMC: FETCH+WRITE = 1.05 x algorithmic bytes).
//
// Forward implementations (r3det_set_option("fr_impl", k); 0 = automatic choice):
//   1 generic : taps gathered from global memory (L1/L2); any H x W, points 1 or 5, no LDS.
//   2 plane   : one workgroup per tile of planes staged in LDS, taps derived in the kernel from
//               the boxes (no workspace); any plane that fits 68 KB of LDS, points 1 or 5.
//  10 cell    : (needs the caller's workspace; points = 1; 128 x 128 or 64 x 64 planes) a
//               workgroup owns G channels of one image, keeps the taps of all positions in
//               registers and streams the planes through two LDS buffers; see the kernel.
// All of them produce bit-identical outputs (tests/test_gpu_fr.py).  Variants that were tried and
// removed (skewed layouts, persistent tile walkers, 12-byte tap records, two-plane-deep
// prefetch) are described with their measurements in DESIGN.md 4.3.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

#include "r3_fr_tap.h"
#include "r3_kernels.h"
#include "r3_trig.h"

R3Option g_r3_fr_impl{0};
R3Option g_r3_fr_dbg{0};  // launch form of the forward kernels (r3_kernels.h r3_fr_dbg)
R3Option g_r3_fr_walk{8}; // strip height of the tile-pair walk (r3_fr_tap.h pair_walk; 0: row-major)
R3Option g_r3_fr_profile{0};  // 1: cell-path launches record their own start / stop events

namespace {

typedef float fr_v4 __attribute__((ext_vector_type(4)));

// Branch-free: an invalid tap has offsets 0 (always readable) and its value is discarded by
// a select, so the four reads of every tap can be issued back to back (an `if (valid)` around
// them costs one exposed LDS round trip per tap: measured 2.9 -> 4.4 TB/s-equivalent).
template <typename P>
__device__ __forceinline__ float tap_value(const Tap& t, P plane) {
  float lt = plane[t.o00], rt = plane[t.o01], lb = plane[t.o10], rb = plane[t.o11];
  float v = (t.w1 * lt + t.w2 * rt + t.w3 * lb + t.w4 * rb);
  return t.valid ? v : 0.f;
}

// ----------------------------------------------------------------------------------------
// generic kernels: block = 256 consecutive positions of one image x a slice of channels
// ----------------------------------------------------------------------------------------
constexpr int FR_BLOCK = 256;

template <int POINTS>
__global__ __launch_bounds__(FR_BLOCK) void fr_forward_generic(const float* __restrict__ feat,
                                                               const float* __restrict__ boxes,
                                                               int C, int H, int W, float scale,
                                                               int c_per_block,
                                                               float* __restrict__ out) {
  const int HW = H * W;
  const int hw = blockIdx.x * FR_BLOCK + threadIdx.x;
  const int n = blockIdx.z;
  if (hw >= HW) return;
  Tap taps[POINTS];
  make_taps<POINTS>(boxes + ((size_t)n * HW + hw) * 5, scale, H, W, W, taps);
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; c++) {
    const float* plane = feat + ((size_t)n * C + c) * HW;
    float v = plane[hw];
#pragma unroll
    for (int p = 0; p < POINTS; p++) v += tap_value(taps[p], plane);
    out[((size_t)n * C + c) * HW + hw] = v;
  }
}

template <int POINTS>
__global__ __launch_bounds__(FR_BLOCK) void fr_backward_generic(const float* __restrict__ top,
                                                                const float* __restrict__ boxes,
                                                                int C, int H, int W, float scale,
                                                                int c_per_block,
                                                                float* __restrict__ bottom) {
  const int HW = H * W;
  const int hw = blockIdx.x * FR_BLOCK + threadIdx.x;
  const int n = blockIdx.z;
  if (hw >= HW) return;
  Tap taps[POINTS];
  make_taps<POINTS>(boxes + ((size_t)n * HW + hw) * 5, scale, H, W, W, taps);
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; c++) {
    float* plane = bottom + ((size_t)n * C + c) * HW;
    float g = top[((size_t)n * C + c) * HW + hw];
    unsafeAtomicAdd(plane + hw, g);
#pragma unroll
    for (int p = 0; p < POINTS; p++) {
      const Tap& t = taps[p];
      if (t.valid) {
        unsafeAtomicAdd(plane + t.o00, g * t.w1);
        unsafeAtomicAdd(plane + t.o01, g * t.w2);
        unsafeAtomicAdd(plane + t.o10, g * t.w3);
        unsafeAtomicAdd(plane + t.o11, g * t.w4);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// lds-plane kernels (no workspace): one workgroup owns CPB consecutive (n,c) planes.
// A thread owns QUADS of 4 adjacent positions: its 4 boxes are one contiguous 80-byte read
// (5 x 16 B), the 4 results one 16-byte store; all global loads of a round are issued before
// the first use.
// ----------------------------------------------------------------------------------------
constexpr int FRP_BLOCK = 1024;
constexpr int FRP_LDS_FLOATS = 17 * 1024;  // 68 KB: a 128 x 129 plane (66 KB) fits
constexpr int FRP_STAGE_UNROLL = 4;

// global (contiguous nc planes of HW floats) -> LDS planes with row pitch W+1
// (src2: a second addend of the planes, or null -- the module's conv_5_1(conv_1_5(x)) + conv_1_1(x), summed on the way in)
template <bool VEC, bool TWO = false>
__device__ __forceinline__ void stage_planes(const float* __restrict__ src, float* lds, int nc, int HW,
                                             int W, int pitch, int psz, const int tid, const int T,
                                             const float* __restrict__ src2 = nullptr) {
  const int total = nc * HW;
  if (VEC) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const float4* t4 = reinterpret_cast<const float4*>(src2);
    const int total4 = total >> 2;
    for (int base = tid; base < total4; base += T * FRP_STAGE_UNROLL) {
      float4 v[FRP_STAGE_UNROLL], u[FRP_STAGE_UNROLL];
#pragma unroll
      for (int k = 0; k < FRP_STAGE_UNROLL; k++) {
        int i = base + k * T;
        if (i < total4) {
          v[k] = s4[i];
          if (TWO) u[k] = t4[i];
        }
      }
#pragma unroll
      for (int k = 0; k < FRP_STAGE_UNROLL; k++) {
        int i = base + k * T;
        if (i < total4) {
          int e = i << 2;
          int ch = e / HW, r = e - ch * HW;
          int y = r / W, x = r - y * W;
          float* d = lds + ch * psz + y * pitch + x;
          if (TWO) { d[0] = v[k].x + u[k].x; d[1] = v[k].y + u[k].y; d[2] = v[k].z + u[k].z; d[3] = v[k].w + u[k].w; }
          else { d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w; }
        }
      }
    }
  } else {
    for (int e = tid; e < total; e += T) {
      int ch = e / HW, r = e - ch * HW;
      int y = r / W, x = r - y * W;
      lds[ch * psz + y * pitch + x] = TWO ? src[e] + src2[e] : src[e];
    }
  }
}

// NC > 0: planes per workgroup known at compile time.  Matters for more than unrolling: gfx950
// has ONE in-order vmcnt queue for loads and stores, so "wait for the next quad's boxes" also
// waits for every store issued before them unless the boxes were requested BEFORE those stores
// and the compiler can count the younger stores (s_waitcnt vmcnt(NC) instead of vmcnt(0)).
// FUSED (VEC only): the module's tail, out = res + (P + sample(P)) with P = feat + feat2 (feature_refine_module.py:121-126)
template <int POINTS, bool VEC, int NC, bool FUSED = false>
// (tid, T: the thread's index among the T threads that work on this (image, channel group) -- the whole workgroup
// for the per-level launches, a quarter of it in the levels grid; lds_off: where their planes start)
__device__ __forceinline__ void fr_forward_plane_body(const float* __restrict__ feat, const float* __restrict__ boxes,
                                                      int C, int H, int W, float scale, int cpb,
                                                      float* __restrict__ out, const int bx, const int n, const int tid,
                                                      const int T, const int lds_off,
                                                      const float* __restrict__ feat2 = nullptr,
                                                      const float* __restrict__ res = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* lds = lds_all + lds_off;
  const int HW = H * W;
  const int pitch = W + 1;
  const int psz = H * pitch;
  const int c0 = bx * cpb;
  const int nc = NC > 0 ? NC : min(cpb, C - c0);
  const float* src = feat + ((size_t)n * C + c0) * HW;
  float* dst = out + ((size_t)n * C + c0) * HW;
  if (VEC) {
    const float4* bx4 = reinterpret_cast<const float4*>(boxes + (size_t)n * HW * 5);
    const int quads = HW >> 2;
    float4 nb[5];  // boxes of this thread's first quad: requested before staging
    if (tid < quads) {
#pragma unroll
      for (int k = 0; k < 5; k++) nb[k] = bx4[tid * 5 + k];
    }
    if (FUSED) stage_planes<true, true>(src, lds, nc, HW, W, pitch, psz, tid, T, feat2 + ((size_t)n * C + c0) * HW);
    else stage_planes<true>(src, lds, nc, HW, W, pitch, psz, tid, T);
    const float* rsd = FUSED ? res + ((size_t)n * C + c0) * HW : nullptr;
    __syncthreads();
    for (int qd = tid; qd < quads; qd += T) {
      float bq[20];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        bq[4 * k] = nb[k].x; bq[4 * k + 1] = nb[k].y; bq[4 * k + 2] = nb[k].z; bq[4 * k + 3] = nb[k].w;
      }
      const int nq = qd + T;
      if (nq < quads) {
#pragma unroll
        for (int k = 0; k < 5; k++) nb[k] = bx4[nq * 5 + k];  // next quad: before this quad's stores
      }
      const int hw0 = qd << 2;
      const int y = hw0 / W, x = hw0 - y * W;  // W % 4 == 0: the quad stays in one row
      const int self = y * pitch + x;
      Tap taps[4][POINTS];
#pragma unroll
      for (int j = 0; j < 4; j++) make_taps<POINTS>(bq + 5 * j, scale, H, W, pitch, taps[j]);
#pragma unroll
      for (int ch = 0; ch < (NC > 0 ? NC : 1); ch++) {
        for (int c2 = ch; c2 < nc; c2 += (NC > 0 ? NC : 1)) {
          const float* plane = lds + c2 * psz;
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float v = plane[self + j];
#pragma unroll
            for (int p = 0; p < POINTS; p++) v += tap_value(taps[j][p], plane);
            r[j] = v;
          }
          if (FUSED) {  // (the residual: x + (P + sample(P)), the module's last add)
            const float4 x4 = *reinterpret_cast<const float4*>(rsd + (size_t)c2 * HW + hw0);
            r[0] = x4.x + r[0]; r[1] = x4.y + r[1]; r[2] = x4.z + r[2]; r[3] = x4.w + r[3];
          }
          *reinterpret_cast<float4*>(dst + (size_t)c2 * HW + hw0) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
  } else {
    stage_planes<false>(src, lds, nc, HW, W, pitch, psz, tid, T);
    __syncthreads();
    for (int hw = tid; hw < HW; hw += T) {
      Tap taps[POINTS];
      make_taps<POINTS>(boxes + ((size_t)n * HW + hw) * 5, scale, H, W, pitch, taps);
      const int y = hw / W, x = hw - y * W;
      const int self = y * pitch + x;
      for (int ch = 0; ch < nc; ch++) {
        const float* plane = lds + ch * psz;
        float v = plane[self];
#pragma unroll
        for (int p = 0; p < POINTS; p++) v += tap_value(taps[p], plane);
        dst[(size_t)ch * HW + hw] = v;
      }
    }
  }
}

template <int POINTS, bool VEC, int NC>
__global__ __launch_bounds__(FRP_BLOCK) void fr_forward_plane(const float* __restrict__ feat,
                                                              const float* __restrict__ boxes,
                                                              int C, int H, int W, float scale,
                                                              int cpb, float* __restrict__ out) {
  fr_forward_plane_body<POINTS, VEC, NC>(feat, boxes, C, H, W, scale, cpb, out, blockIdx.x, blockIdx.y, threadIdx.x,
                                         FRP_BLOCK, 0);
}

// Several levels of a pyramid as ONE grid (points = 1, the float4 form): the coarse levels of a 1024^2 input are
// 4-6 us launches of latency each on their own.  The levels ride in the kernel arguments, a block finds its level from
// block ranges (as fr_forward_nhwc_occ_levels does); the body is the per-level kernel's.
constexpr int FRPL_MAX = 8;
constexpr int FRPL_T = 256;      // threads per (image, channel group) in the levels grid
constexpr int FRPL_BLOCK = 256;  // threads per workgroup of the levels grid
struct FrPlaneLevel {
  const float* feat;
  const float* feat2;  // FUSED grids: the second addend and the residual of the module tail
  const float* res;
  const float* boxes;
  float* out;
  float scale;
  int H, W, cpb, gx, first, N;  // gx: channel groups per image; first: the level's first block
};
struct FrPlaneLevels {
  FrPlaneLevel l[FRPL_MAX];
  int n;
  // the tap table of a level that takes the cell kernel (blocks from tfirst on; none: tfirst < 0): the small launch
  // in front of that kernel rides in this grid, which then goes first
  const float* tboxes;
  float* table;
  float tscale;
  int tN, tH, tW, tfirst;
  // (the module-tail grid carries the tables of BOTH cell levels: a second job of the same kind behind the first)
  const float* t2boxes;
  float* table2;
  float t2scale;
  int t2H, t2W, t2first;
};

// sample coordinates (row y <- x_ctr * scale, column x <- y_ctr * scale) -> the tap the cell kernel
// keeps: the clamps of bilinear_interpolate (feature_refine_kernel.cu:22-47) applied once; an
// out-of-range sample points at the zero cell (row H + 1)
__device__ __forceinline__ void cell_tap(float y, float x, int H, int W, float& ty, float& tx) {
  if (y < -1.0 || y > H || x < -1.0 || x > W) {
    y = (float)(H + 1);
    x = 0.f;
  } else {
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    if ((int)y >= H - 1) y = (float)(H - 1);
    if ((int)x >= W - 1) x = (float)(W - 1);
  }
  ty = y;
  tx = x;
}

typedef float fr_f2 __attribute__((ext_vector_type(2)));

// Round 6: a channels_last sampler launch that runs in a training step also WRITES the level's tap table -- per image
// [y: HW floats][x: HW floats], the clamped sample point of every position (cell_tap; fr_cell_table_kernel's layout) --
// as a by-product: the wave that owns position q has its box in scalar registers anyway, one lane stores two floats.
// The backward's index kernel (r3_frb.hip, TAB form) then scans 4 contiguous bytes per source instead of the 20-byte
// box records.  (The reference's backward needs no index: it scatters with atomics, feature_refine_kernel.cu:165-230.)
// (At the kernel's start, before its row loads are in flight: lanes 0 .. cnt - 1 of the wave take one position each --
// in the position loop the few temporaries of the clamps pushed the module-tail kernel past its 64 registers.)
__device__ __forceinline__ void fr_emit_tab(float* __restrict__ tabI, const float* __restrict__ bxI, const int HW,
                                            const unsigned q0, const int cnt, const int lane, const float scale,
                                            const int H, const int W) {
  if (lane < cnt) {
    const unsigned q = q0 + (unsigned)lane;
    float y_, x_;
    cell_tap(bxI[q * 5u] * scale, bxI[q * 5u + 1u] * scale, H, W, y_, x_);  // sic: row <- x_ctr, column <- y_ctr
    tabI[q] = y_;
    tabI[HW + q] = x_;
  }
}

// points = 5 (feature_refine_kernel.cu:137-151: the centre and the four corners of the box), NCHW, the cell kernel's
// plane layout -- whole plane in LDS, pitch W + 1, the last column and row staged twice so that the upper taps are
// always one step away, an invalid sample reads the zero rows behind them -- with what five points per position need:
// the sample points live in registers (two floats per point: the clamped coordinates; cell, fractions and address are
// recomputed per plane, 8 vector instructions for 4 LDS reads), so a thread keeps KQ <= 4 positions, and a plane of more
// than 4096 positions is worked on by Q workgroups that each stage the WHOLE plane (from the L2: one of them brought
// it in) and sample a Q-th of the positions.  The plane kernel this replaces for points = 5 rebuilt all taps (a sincos
// and five clamps per position) for every plane: 222 us at level 0, N = 4.
template <int POINTS, int KQ>
__global__ __launch_bounds__(1024) void fr_forward_points_kernel(const float* __restrict__ feat,
                                                                 const float* __restrict__ boxes, int C, int H, int W,
                                                                 float scale, int G, int Q, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // (H + 3) rows of pitch W + 1
  const int tid = threadIdx.x, HW = H * W, PITCH = W + 1;
  const int groups = C / G;
  const int n = blockIdx.x / (groups * Q);
  const int rem = blockIdx.x - n * groups * Q;
  const int grp = rem / Q, q = rem - grp * Q;
  float ty[KQ][POINTS], tx[KQ][POINTS];
  int own[KQ];
#pragma unroll
  for (int k = 0; k < KQ; k++) {
    const int p = (q * KQ + k) * 1024 + tid;
    own[k] = -1;
#pragma unroll
    for (int i = 0; i < POINTS; i++) ty[k][i] = (float)(H + 1), tx[k][i] = 0.f;
    if (p < HW) {
      const float* box = boxes + ((size_t)n * HW + p) * 5;
      const int y = p / W;
      own[k] = p + y;  // (y * PITCH + x)
      const float roi_y = box[0] * scale;  // sic: row <- x_ctr
      const float roi_x = box[1] * scale;  //      col <- y_ctr
      cell_tap(roi_y, roi_x, H, W, ty[k][0], tx[k][0]);
      if (POINTS > 1) {
        const float roi_w = box[2] * scale, roi_h = box[3] * scale, roi_a = box[4];
        const float w_2 = roi_w / 2, h_2 = roi_h / 2;
        float sina, cosa;
        r3_sincos(roi_a, sina, cosa);
        const float wx = cosa * w_2, wy = sina * w_2;
        const float hx = -sina * h_2, hy = cosa * h_2;
        cell_tap(roi_y + wy + hy, roi_x + wx + hx, H, W, ty[k][1], tx[k][1]);
        cell_tap(roi_y - wy + hy, roi_x - wx + hx, H, W, ty[k][2], tx[k][2]);
        cell_tap(roi_y - wy - hy, roi_x - wx - hx, H, W, ty[k][3], tx[k][3]);
        cell_tap(roi_y + wy - hy, roi_x + wx - hx, H, W, ty[k][4], tx[k][4]);
      }
    }
  }
  // Two plane buffers: the next plane's elements are requested before the current plane is sampled and written to the
  // idle buffer behind it -- one workgroup per compute unit (79 VGPRs at four positions per thread), so nothing else
  // would hide the loads; one barrier per plane.
  const int BUF = (H + 3) * PITCH;
  for (int i = tid; i < 2 * PITCH; i += 1024) lds[(H + 1) * PITCH + i] = lds[BUF + (H + 1) * PITCH + i] = 0.f;  // zero rows
  const size_t plane0 = ((size_t)n * C + (size_t)grp * G) * HW;
  constexpr int SV = 8;  // elements of HALF a plane per thread (HW <= 16 384: the launcher); 16 in flight: spills
  float sv[SV];
  auto request = [&](const float* src, const int half) {
#pragma unroll
    for (int u = 0; u < SV; u++) {
      const int i = (half * SV + u) * 1024 + tid;
      sv[u] = i < HW ? src[i] : 0.f;
    }
  };
  // (a thread's elements are 1024 apart: for the widths of a pyramid -- W divides 1024 -- the same column, rows a
  // constant step apart)
  const bool wdiv = (1024 % W) == 0;
  const int y0 = tid / W, x0 = tid - y0 * W, ystep = 1024 / W;
  auto deposit = [&](float* buf, const int half) {
#pragma unroll
    for (int u = 0; u < SV; u++) {
      const int i = (half * SV + u) * 1024 + tid;
      if (i < HW) {
        const int y = wdiv ? y0 + (half * SV + u) * ystep : i / W, x = wdiv ? x0 : i - (i / W) * W;
        buf[y * PITCH + x] = sv[u];
        if (x == W - 1) buf[y * PITCH + W] = sv[u];  // last column twice
        if (y == H - 1) {                            // last row twice (and its corner)
          buf[H * PITCH + x] = sv[u];
          if (x == W - 1) buf[H * PITCH + W] = sv[u];
        }
      }
    }
  };
  request(feat + plane0, 0);
  deposit(lds, 0);
  request(feat + plane0, 1);
  deposit(lds, 1);
  __syncthreads();
  for (int c = 0; c < G; c++) {
    const float* buf = lds + (c & 1) * BUF;
    float* dst = out + plane0 + (size_t)c * HW;
    // (the next plane in two halves: half of it requested in front of each half of this plane's positions)
    float* nbuf = lds + ((c + 1) & 1) * BUF;
    if (c + 1 < G) request(feat + plane0 + (size_t)(c + 1) * HW, 0);
#pragma unroll
    for (int k = 0; k < KQ; k++) {
      if (k == (KQ + 1) / 2 && c + 1 < G) {
        deposit(nbuf, 0);
        request(feat + plane0 + (size_t)(c + 1) * HW, 1);
      }
      if (own[k] < 0) continue;
      // (the cells of up to three points of a position requested together -- the two of a row as one 8-byte pair --
      // then the sums in the reference's order, weights in packed fp32 as in fr_forward_cell; all five at once: spills)
      float v = buf[own[k]];
#pragma unroll
      for (int i0 = 0; i0 < POINTS; i0 += 3) {
        constexpr int PB = 3;
        fr_f2 top[PB], bot[PB];
        float fy[PB], fx[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) {
          const int i = i0 + j;
          if (i < POINTS) {
            float y = ty[k][i], x = tx[k][i];
            asm volatile("" : "+v"(y), "+v"(x));  // (else cell / fractions / address of all points are hoisted: spills)
            const int yi = (int)y, xi = (int)x;
            fy[j] = __builtin_amdgcn_fractf(y), fx[j] = __builtin_amdgcn_fractf(x);
            const int a = yi * PITCH + xi;
            top[j] = fr_f2{buf[a], buf[a + 1]};
            bot[j] = fr_f2{buf[a + PITCH], buf[a + PITCH + 1]};
          }
        }
#pragma unroll
        for (int j = 0; j < PB; j++) {
          if (i0 + j < POINTS) {
            // (1.f - f == (float)(1. - (double)f) for f in [0, 1): see fr_forward_cell)
            const fr_f2 hf = {1.f - fx[j], fx[j]};
            const fr_f2 pt = ((1.f - fy[j]) * hf) * top[j];  // {w1 * lt, w2 * rt}
            const fr_f2 pb = (fy[j] * hf) * bot[j];          // {w3 * lb, w4 * rb}
            v += pt.x + pt.y + pb.x + pb.y;
          }
        }
      }
      dst[(q * KQ + k) * 1024 + tid] = v;
    }
    if (c + 1 < G) {
      if (KQ == 1) {  // (no second half of positions to put the first deposit in front of)
        deposit(nbuf, 0);
        request(feat + plane0 + (size_t)(c + 1) * HW, 1);
      }
      deposit(nbuf, 1);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void fr_cell_table_body(const float* __restrict__ boxes, int N, int H, int W, float scale,
                                                   float* __restrict__ table, const int pos) {
  const int HW = H * W;
  if (pos >= N * HW) return;
  float y, x;
  cell_tap(boxes[(size_t)pos * 5] * scale, boxes[(size_t)pos * 5 + 1] * scale, H, W, y, x);  // sic: row <- x_ctr
  const int n = pos / HW, p = pos - n * HW;
  table[(size_t)n * 2 * HW + p] = y;  // per image: [y: HW floats][x: HW floats]
  table[(size_t)n * 2 * HW + HW + p] = x;
}

template <bool FUSED = false>
__global__ __launch_bounds__(FRPL_BLOCK) void fr_forward_plane_levels(const FrPlaneLevels A, int C) {
  if (FUSED && A.t2first >= 0 && (int)blockIdx.x >= A.t2first) {
    fr_cell_table_body(A.t2boxes, A.tN, A.t2H, A.t2W, A.t2scale, A.table2, ((int)blockIdx.x - A.t2first) * FRPL_BLOCK + (int)threadIdx.x);
    return;
  }
  if (A.tfirst >= 0 && (int)blockIdx.x >= A.tfirst) {
    fr_cell_table_body(A.tboxes, A.tN, A.tH, A.tW, A.tscale, A.table, ((int)blockIdx.x - A.tfirst) * FRPL_BLOCK + (int)threadIdx.x);
    return;
  }
  int k = 0;
#pragma unroll
  for (int i = 1; i < FRPL_MAX; i++)
    if (i < A.n && (int)blockIdx.x >= A.l[i].first) k = i;
  const FrPlaneLevel& L = A.l[k];
  // A coarse level's plane is 16 ... 256 quads of positions: FRPL_T threads work on one (image, channel group), a
  // workgroup on FRPL_BLOCK / FRPL_T of them side by side.
  const int unit = ((int)blockIdx.x - L.first) * (FRPL_BLOCK / FRPL_T) + (int)(threadIdx.x / FRPL_T);
  const int tid = threadIdx.x & (FRPL_T - 1);
  const int n = unit / L.gx, bx = unit - n * L.gx;
  if (n >= L.N) {  // (a last workgroup's idle quarters keep the barrier)
    __syncthreads();
    return;
  }
  const int off = (int)(threadIdx.x / FRPL_T) * L.cpb * L.H * (L.W + 1);
  // (planes per unit at compile time where it is 1 or 2, as the per-level launches have it)
  if (L.cpb == 1)
    fr_forward_plane_body<1, true, 1, FUSED>(L.feat, L.boxes, C, L.H, L.W, L.scale, 1, L.out, bx, n, tid, FRPL_T, off, L.feat2, L.res);
  else if (L.cpb == 2)
    fr_forward_plane_body<1, true, 2, FUSED>(L.feat, L.boxes, C, L.H, L.W, L.scale, 2, L.out, bx, n, tid, FRPL_T, off, L.feat2, L.res);
  else
    fr_forward_plane_body<1, true, 0, FUSED>(L.feat, L.boxes, C, L.H, L.W, L.scale, L.cpb, L.out, bx, n, tid, FRPL_T, off, L.feat2, L.res);
}

// backward: accumulate the plane's gradient in LDS (ds_add_f32), then one coalesced
// read-modify-write (or plain write when overwrite) of bottom_grad.
template <int POINTS, bool VEC>
__global__ __launch_bounds__(FRP_BLOCK) void fr_backward_plane(const float* __restrict__ top,
                                                               const float* __restrict__ boxes,
                                                               int C, int H, int W, float scale,
                                                               int cpb, int overwrite,
                                                               float* __restrict__ bottom) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W;
  const int pitch = W + 1;
  const int psz = H * pitch;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * cpb;
  const int nc = min(cpb, C - c0);
  const float* src = top + ((size_t)n * C + c0) * HW;
  float* dst = bottom + ((size_t)n * C + c0) * HW;
  for (int i = threadIdx.x; i < nc * psz; i += FRP_BLOCK) lds[i] = 0.f;
  __syncthreads();
  if (VEC) {
    const float4* bx4 = reinterpret_cast<const float4*>(boxes + (size_t)n * HW * 5);
    const int quads = HW >> 2;
    for (int qd = threadIdx.x; qd < quads; qd += FRP_BLOCK) {
      float bq[20];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        float4 t = bx4[qd * 5 + k];
        bq[4 * k] = t.x; bq[4 * k + 1] = t.y; bq[4 * k + 2] = t.z; bq[4 * k + 3] = t.w;
      }
      const int hw0 = qd << 2;
      const int y = hw0 / W, x = hw0 - y * W;
      const int self = y * pitch + x;
      Tap taps[4][POINTS];
#pragma unroll
      for (int j = 0; j < 4; j++) make_taps<POINTS>(bq + 5 * j, scale, H, W, pitch, taps[j]);
      for (int ch = 0; ch < nc; ch++) {
        float* plane = lds + ch * psz;
        const float4 g4 = *reinterpret_cast<const float4*>(src + (size_t)ch * HW + hw0);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          atomicAdd(plane + self + j, g[j]);
#pragma unroll
          for (int p = 0; p < POINTS; p++) {
            const Tap& t = taps[j][p];
            if (t.valid) {
              atomicAdd(plane + t.o00, g[j] * t.w1);
              atomicAdd(plane + t.o01, g[j] * t.w2);
              atomicAdd(plane + t.o10, g[j] * t.w3);
              atomicAdd(plane + t.o11, g[j] * t.w4);
            }
          }
        }
      }
    }
  } else {
    for (int hw = threadIdx.x; hw < HW; hw += FRP_BLOCK) {
      Tap taps[POINTS];
      make_taps<POINTS>(boxes + ((size_t)n * HW + hw) * 5, scale, H, W, pitch, taps);
      const int y = hw / W, x = hw - y * W;
      const int self = y * pitch + x;
      for (int ch = 0; ch < nc; ch++) {
        float* plane = lds + ch * psz;
        float g = src[(size_t)ch * HW + hw];
        atomicAdd(plane + self, g);
#pragma unroll
        for (int p = 0; p < POINTS; p++) {
          const Tap& t = taps[p];
          if (t.valid) {
            atomicAdd(plane + t.o00, g * t.w1);
            atomicAdd(plane + t.o01, g * t.w2);
            atomicAdd(plane + t.o10, g * t.w3);
            atomicAdd(plane + t.o11, g * t.w4);
          }
        }
      }
    }
  }
  __syncthreads();
  const int total = nc * HW;
  if (VEC) {
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < (total >> 2); i += FRP_BLOCK) {
      int e = i << 2;
      int ch = e / HW, r = e - ch * HW;
      int y = r / W, x = r - y * W;
      const float* p = lds + ch * psz + y * pitch + x;
      float4 v = make_float4(p[0], p[1], p[2], p[3]);
      if (!overwrite) {
        float4 o = d4[i];
        v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w;
      }
      d4[i] = v;
    }
  } else {
    for (int e = threadIdx.x; e < total; e += FRP_BLOCK) {
      int ch = e / HW, r = e - ch * HW;
      int y = r / W, x = r - y * W;
      float v = lds[ch * psz + y * pitch + x];
      dst[e] = overwrite ? v : dst[e] + v;
    }
  }
}

// ----------------------------------------------------------------------------------------
// channels_last (NHWC) forward kernel.  A position's C channels are contiguous (1 KB at C = 256): one
// WAVEFRONT per output position, lane <-> 4 consecutive channels, so the identity read, each of the four
// bilinear taps and the store are single coalesced 16-byte-per-lane accesses and the (wave-uniform) tap
// geometry is computed once per position instead of once per (position, channel).  No LDS, no plane
// staging: the transpose that the reference's x / y swap makes of the gather (row <- x_ctr,
// feature_refine_kernel.cu:131-132) only changes WHICH 1 KB rows a wave reads.
// Locality: a workgroup owns an 8 x 8 tile of output positions (two rows per wave); its taps fall in a ~9 x 9 block
// of input positions (the transposed tile), so the re-reads of every input row hit L1 / L2.  Tiles are dealt to the
// eight XCDs in contiguous bands (blockIdx & 7 = XCD under round-robin dispatch): halo rows shared by
// neighbouring tiles are then fetched into ONE L2 instead of up to eight.
// FUSED (the FeatureRefineModule tail for channels_last pipelines, feature_refine_module.py:121-126): the
// sampled map is P = (a + bias_a) + (b + bias_b) with a, b the raw outputs of conv_5_1 and conv_1_1, and
// out = res + (P(id) + sample(P)) in the reference's operation order: 3 reads + 1 write per element and no
// layout switch, where the NCHW sampler needed two transposing passes (r3det_frm_mix_nchw) around it.
// ----------------------------------------------------------------------------------------
constexpr int NH_ROWS = 4;  // output rows per workgroup (one wave each)

template <int POINTS, bool FUSED, int KW>
__global__ __launch_bounds__(256) void fr_forward_nhwc(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ bias_a,
                                                       const float* __restrict__ bias_b,
                                                       const float* __restrict__ res,
                                                       const float* __restrict__ boxes, int C, int H, int W,
                                                       float scale, int tiles_x, int tiles_per_img, int T,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned t = blockIdx.x;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands of tiles
  const int n = (int)(t / (unsigned)tiles_per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)tiles_per_img);
  const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
  const int h = ty * NH_ROWS + wave;
  if (h >= H) return;
  const int HW = H * W, C4 = C >> 2;
  const size_t img = (size_t)n * HW;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int c4 = lane; c4 < C4; c4 += 64) {
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
    if (FUSED) {
      if (bias_a) ba = reinterpret_cast<const float4*>(bias_a)[c4];
      if (bias_b) bb = reinterpret_cast<const float4*>(bias_b)[c4];
    }
    // value of the sampled map at position q (a plain load, or the module's (a + bias_a) + (b + bias_b))
    auto P = [&](size_t q) -> float4 {
      float4 v = a4[q * C4 + c4];
      if (FUSED) {
        v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
        if (b) {
          float4 u = b4[q * C4 + c4];
          u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      return v;
    };
    // a wave walks KW consecutive output columns of its row
#pragma unroll
    for (int k = 0; k < KW; k++) {
      const int wq = tx * KW + k;
      const bool live = wq < W;
      const int w = live ? wq : W - 1;  // (clamped: no early exit, so that the loads of all k can be interleaved)
      const int hw = h * W + w;
      Tap taps[POINTS];
      make_taps<POINTS>(boxes + (img + hw) * 5, scale, H, W, W, taps);
      float4 v = P(img + hw);
#pragma unroll
      for (int p = 0; p < POINTS; p++) {
        const Tap& tp = taps[p];
        const float4 lt = P(img + tp.o00), rt = P(img + tp.o01), lb = P(img + tp.o10), rb = P(img + tp.o11);
        float4 s;
        s.x = tp.w1 * lt.x + tp.w2 * rt.x + tp.w3 * lb.x + tp.w4 * rb.x;
        s.y = tp.w1 * lt.y + tp.w2 * rt.y + tp.w3 * lb.y + tp.w4 * rb.y;
        s.z = tp.w1 * lt.z + tp.w2 * rt.z + tp.w3 * lb.z + tp.w4 * rb.z;
        s.w = tp.w1 * lt.w + tp.w2 * rt.w + tp.w3 * lb.w + tp.w4 * rb.w;
        if (tp.valid) { v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
      }
      if (FUSED && res) {
        const float4 r = r4[(img + hw) * C4 + c4];
        v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
      }
      if (live) o4[(img + hw) * C4 + c4] = v;
    }
  }
}

// points = 5 on channels_last memory (plain sampler): the simple kernel above reads 21 rows of 1 KB per position
// through the L2 -- 1.4 GB per launch at level 0, N = 4, which is what its 101 us are.  Here a workgroup owns an 8 x 8
// tile of positions and stages, per block of 64 channels (256 bytes per row), the rows its samples can reach: by the
// reference's row <- x_ctr / column <- y_ctr swap the TRANSPOSED tile, +- P5_R cells for the corners of the boxes
// (16 x 16 rows, 64 KB); a cell outside that region (a box larger than 2 P5_R cells) is read from memory.  The sample
// geometry of the 64 positions is computed once (one thread per position: a sincos and five clamps) into an LDS table
// of cell coordinates and weights and serves all channel blocks.  Same operations in the same order as the simple
// kernel: bit-identical.
constexpr int P5_T = 8, P5_R = 4, P5_REG = P5_T + 2 * P5_R, P5_ROWS = P5_REG * P5_REG;
struct P5Tap {
  short yl, xl, yh, xh;  // cells of the map; yl < 0: the sample is outside the map (contributes nothing)
  float w[4];
};

__global__ __launch_bounds__(1024) void fr_forward_nhwc_p5(const float* __restrict__ a, const float* __restrict__ boxes,
                                                          int C, int H, int W, float scale, int tiles_x,
                                                          int tiles_per_img, int T, float* __restrict__ out) {
  __shared__ float4 reg[P5_ROWS][16];  // the region's rows, 64 channels of them
  __shared__ P5Tap taps[P5_T * P5_T][5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned t = blockIdx.x;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands of tiles
  const int n = (int)(t / (unsigned)tiles_per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)tiles_per_img);
  const int tyi = tt / tiles_x, txi = tt - tyi * tiles_x;
  const int py0 = tyi * P5_T, px0 = txi * P5_T;      // the tile of positions
  const int ry0 = px0 - P5_R, rx0 = py0 - P5_R;      // the region of cells: rows <- columns of the tile, and back
  const int HW = H * W, C4 = C >> 2;
  const size_t img = (size_t)n * HW;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  float4* o4 = reinterpret_cast<float4*>(out);
  if (tid < P5_T * P5_T) {
    const int py = py0 + (tid >> 3), px = px0 + (tid & 7);  // (H, W multiples of 8: the launcher)
    TapYX tp[5];
    make_taps_yx<5>(boxes + (img + (size_t)py * W + px) * 5, scale, H, W, tp);
#pragma unroll
    for (int i = 0; i < 5; i++) {
      P5Tap e;
      e.yl = tp[i].valid ? (short)tp[i].yl : (short)-1;
      e.xl = (short)tp[i].xl, e.yh = (short)tp[i].yh, e.xh = (short)tp[i].xh;
      e.w[0] = tp[i].w[0], e.w[1] = tp[i].w[1], e.w[2] = tp[i].w[2], e.w[3] = tp[i].w[3];
      taps[tid][i] = e;
    }
  }
  const int grp = lane >> 4, ch = lane & 15;  // four positions of a wave at a time, 16 lanes x 16 bytes each
  for (int c0 = 0; c0 < C4; c0 += 16) {
    // staging: 256 rows x 16 float4, four per thread (16 wavefronts: a position's five points are one wavefront round;
    // with four wavefronts per workgroup and two workgroups per compute unit nothing hid the latencies: 108 us)
    {
      float4 sv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int r = u * 64 + (tid >> 4);
        const int y = ry0 + r / P5_REG, x = rx0 + r % P5_REG;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        sv[u] = in ? a4[(img + (size_t)y * W + x) * C4 + c0 + ch] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) reg[u * 64 + (tid >> 4)][ch] = sv[u];
    }
    __syncthreads();  // (also: the tap table is complete)
    // a cell's row: from the region or from memory
    auto row = [&](const int y, const int x) -> float4 {
      const int i = y - ry0, j = x - rx0;
      if ((unsigned)i < (unsigned)P5_REG && (unsigned)j < (unsigned)P5_REG) return reg[i * P5_REG + j][ch];
      return a4[(img + (size_t)y * W + x) * C4 + c0 + ch];
    };
    {
      const int pl = wave * 4 + grp;  // position inside the tile
      const int py = py0 + (pl >> 3), px = px0 + (pl & 7);
      const size_t q = (img + (size_t)py * W + px) * C4 + c0 + ch;
      float4 v = a4[q];
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const P5Tap tp = taps[pl][i];
        if (tp.yl < 0) continue;
        const float4 lt = row(tp.yl, tp.xl), rt = row(tp.yl, tp.xh), lb = row(tp.yh, tp.xl), rb = row(tp.yh, tp.xh);
        float4 sm;
        sm.x = tp.w[0] * lt.x + tp.w[1] * rt.x + tp.w[2] * lb.x + tp.w[3] * rb.x;
        sm.y = tp.w[0] * lt.y + tp.w[1] * rt.y + tp.w[2] * lb.y + tp.w[3] * rb.y;
        sm.z = tp.w[0] * lt.z + tp.w[1] * rt.z + tp.w[2] * lb.z + tp.w[3] * rb.z;
        sm.w = tp.w[0] * lt.w + tp.w[1] * rt.w + tp.w[2] * lb.w + tp.w[3] * rb.w;
        v.x += sm.x; v.y += sm.y; v.z += sm.z; v.w += sm.w;
      }
      o4[q] = v;
    }
    if (c0 + 16 < C4) __syncthreads();  // the next channel block overwrites the region
  }
}

// Software-pipelined form for points = 1 (the shipped configuration).  The simple kernel above is latency-bound:
// per position a wave waits for the box, derives the tap, waits for its 11 loads, stores (3.2 TB/s on four streams
// with one position per wave, less with more).  Here a wave walks KW consecutive columns of its row with a
// three-stage pipeline in registers: box of position i + 2 requested | tap of i + 1 derived and its 11 loads
// issued | position i consumed and stored -- so every wave always has one position's loads (11 x 1 KB) in flight.
// Measured (level 0, N = 4, C = 256, buffers rotating beyond the Infinity Cache; tools/fr_nhwc_var.py,
// gpurun PMC passes): 80-83 us whatever the form (1 / 4 / 8 / 16 columns per wave, pipelined or not, pairs of
// taps reused in registers or not): FETCH_SIZE x 2 = 335 MB against 201 MB of inputs -- conv_a and conv_b cross
// the fabric TWICE, once as the identity of their own position and once as taps of the transposed positions,
// because ~160 concurrent workgroups stream ~16 MB through a 4 MB L2 between the two uses.  335 + 67 MB written
// in 83 us = 4.85 TB/s: the same memory-side rate the NCHW kernels reach; the kernel is bound there.  Handling
// an output tile and its transpose in one workgroup (second use right after the first) did not raise the L2
// hit rate (4.05 M hits / 3.22 M misses vs 3.86 / 3.31) and cost 8 % (fewer, longer workgroups): not shipped.
struct NhwcLoads {
  float4 ia, ib, ir;     // identity: conv_a, conv_b, residual
  float4 ta[4], tb[4];   // the four taps of conv_a / conv_b
};

// PAIRED (square maps): a workgroup of 8 waves owns the 4 x 4 output tile (i, j) AND its transpose (j, i), four
// waves each, AT THE SAME TIME.  With the reference's x / y swap and a regular box field the taps of output tile
// (i, j) lie in input tile (j, i), whose lines output tile (j, i) reads as its identity -- and vice versa: the two
// uses of every conv_a / conv_b line then meet in the CU's L1 / the XCD's L2 instead of crossing the fabric twice.
template <bool FUSED, int KW, bool PAIRED>
__global__ __launch_bounds__(PAIRED ? 512 : 256) void fr_forward_nhwc_pipe(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias_a,
    const float* __restrict__ bias_b, const float* __restrict__ res, const float* __restrict__ boxes, int C, int H, int W,
    float scale, int tiles_x, int tiles_per_img, int T, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, half = threadIdx.x >> 8;
  unsigned t = blockIdx.x;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands of tiles
  const int n = (int)(t / (unsigned)tiles_per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)tiles_per_img);
  int ty, tx;
  if (PAIRED) {  // tt indexes the unordered pairs i <= j of the tiles_x x tiles_x grid
    int pj = (int)((sqrtf(8.f * (float)tt + 1.f) - 1.f) * 0.5f);
    while (pj * (pj + 1) / 2 > tt) pj--;
    while ((pj + 1) * (pj + 2) / 2 <= tt) pj++;
    const int pi = tt - pj * (pj + 1) / 2;
    if (half == 1 && pi == pj) return;  // a diagonal tile is its own transpose
    ty = half ? pj : pi;
    tx = half ? pi : pj;
  } else {
    ty = tt / tiles_x;
    tx = tt - ty * tiles_x;
  }
  const int h = ty * NH_ROWS + wave;
  if (h >= H) return;
  const int HW = H * W, C4 = C >> 2;
  const size_t img = (size_t)n * HW;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* o4 = reinterpret_cast<float4*>(out);
  const int w0 = tx * KW;
  const int cnt = min(KW, W - w0);
  const bool two = FUSED && b != nullptr, has_res = FUSED && res != nullptr;
  for (int c4 = lane; c4 < C4; c4 += 64) {
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
    if (FUSED) {
      if (bias_a) ba = reinterpret_cast<const float4*>(bias_a)[c4];
      if (bias_b) bb = reinterpret_cast<const float4*>(bias_b)[c4];
    }
    auto box_of = [&](int i, float& bx, float& by) {  // (clamped index: the loads are unconditional)
      const float* bp = boxes + (img + (size_t)h * W + w0 + min(i, cnt - 1)) * 5;
      bx = bp[0];
      by = bp[1];
    };
    auto issue = [&](int i, const Tap& tp, NhwcLoads& L) {
      const size_t q = img + (size_t)h * W + w0 + min(i, cnt - 1);
      const size_t o[4] = {img + tp.o00, img + tp.o01, img + tp.o10, img + tp.o11};
      L.ia = a4[q * C4 + c4];
      if (two) L.ib = b4[q * C4 + c4];
      if (has_res) L.ir = r4[q * C4 + c4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        L.ta[k] = a4[o[k] * C4 + c4];
        if (two) L.tb[k] = b4[o[k] * C4 + c4];
      }
    };
    auto mixv = [&](const float4& x, const float4& y) -> float4 {  // (x + bias_a) + (y + bias_b), or x alone
      float4 v = x;
      if (FUSED) {
        v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
        if (two) {
          float4 u = y;
          u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      return v;
    };
    auto consume = [&](int i, const Tap& tp, const NhwcLoads& L) {
      float4 v = mixv(L.ia, L.ib);
      const float4 lt = mixv(L.ta[0], L.tb[0]), rt = mixv(L.ta[1], L.tb[1]);
      const float4 lb = mixv(L.ta[2], L.tb[2]), rb = mixv(L.ta[3], L.tb[3]);
      float4 s;
      s.x = tp.w1 * lt.x + tp.w2 * rt.x + tp.w3 * lb.x + tp.w4 * rb.x;
      s.y = tp.w1 * lt.y + tp.w2 * rt.y + tp.w3 * lb.y + tp.w4 * rb.y;
      s.z = tp.w1 * lt.z + tp.w2 * rt.z + tp.w3 * lb.z + tp.w4 * rb.z;
      s.w = tp.w1 * lt.w + tp.w2 * rt.w + tp.w3 * lb.w + tp.w4 * rb.w;
      if (tp.valid) { v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
      if (has_res) { v.x = L.ir.x + v.x; v.y = L.ir.y + v.y; v.z = L.ir.z + v.z; v.w = L.ir.w + v.w; }
      if (i < cnt) o4[(img + (size_t)h * W + w0 + i) * C4 + c4] = v;
    };
    float bx0, by0, bx1, by1;
    box_of(0, bx0, by0);
    box_of(1, bx1, by1);
    Tap tA = make_tap(H, W, W, bx0 * scale, by0 * scale), tB;  // sic: row <- x_ctr, column <- y_ctr
    NhwcLoads LA, LB;
    issue(0, tA, LA);
#pragma unroll 1  // (a rolled loop: fully unrolled, the scheduler hoists every iteration's loads and needs 256 VGPRs)
    for (int i = 0; i < KW; i += 2) {
      // stage: box i + 2 | tap + loads of i + 1 | consume i        (buffers A / B alternate)
      float bx2, by2;
      box_of(i + 2, bx2, by2);
      tB = make_tap(H, W, W, bx1 * scale, by1 * scale);
      issue(i + 1, tB, LB);
      consume(i, tA, LA);
      box_of(i + 3, bx1, by1);
      tA = make_tap(H, W, W, bx2 * scale, by2 * scale);
      issue(i + 2, tA, LA);  // (beyond the last column: a clamped reload, never consumed)
      consume(i + 1, tB, LB);
    }
  }
}

// High-occupancy form (the shipped one).  What the measurements say (tools/probes/stream_probe.hip,
// tools/fr_nhwc_fields.py, tools/fr_nhwc_ab.py; level 0, N = 4, buffers rotating beyond the Infinity Cache):
//   * out = a + b + r alone, in this kernel's 4 x 4 tile order, streams at 5.6 TB/s (48 us) with 8 light
//     workgroups per CU; eight more cache-hitting row loads per position (the taps) make it 60 us -- the L1
//     serves 64 B per clock and CU;
//   * the software-pipelined kernels above need 156 / 240 VGPRs: 8 waves per CU, 83 / 112 us, and 72 us even when
//     every sample is out of range -- they are bound by how few loads a CU has in flight, not by the gather.
// So: no register pipeline: a wave requests the identity rows of its four positions at once and there are 16
// waves per CU; and most tap rows never pass through the L1 at all (see the kernel's first comment).
// VAR bit 0: boxes prefetched in phase 1; bit 1: non-temporal res / out; bit 2: 32-bit index arithmetic
template <bool FUSED, bool PAIRED, int VAR = 0, bool EMIT = false>  // EMIT: the launch also writes the level's tap table
__device__ __forceinline__ void fr_forward_nhwc_occ_body(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias_a,
    const float* __restrict__ bias_b, const float* __restrict__ res, const float* __restrict__ boxes, int C, int H, int W,
    float scale, int tiles_xs, int tiles_per_img, int T, float* __restrict__ out, const unsigned block,
    float* __restrict__ tab = nullptr) {
  const int tiles_x = tiles_xs & 0xfffff, strip = tiles_xs >> 20;  // (the pair walk's strip height rides in the top bits)
  // The sampled map P = (a + bias_a) + (b + bias_b) of the workgroup's own positions is shared through LDS: each
  // wave computes P for its 4 positions once (their identity term), the barrier publishes the two 4 x 4 tiles, and
  // a tap that falls inside either tile -- with transposed pairing that is 3 of 4 taps of a regular box field --
  // is one LDS read instead of two row loads through the L1 (which serves 64 B per clock: the 8 tap rows per
  // position were a third of the kernel).  Taps outside the two tiles are loaded as before.
  constexpr bool PRE = (VAR & 1) != 0, NT = (VAR & 2) != 0, LEAN = (VAR & 4) != 0;
  constexpr bool NTI = (VAR & 8) != 0;  // (LEAN) the identity rows of a tile's 2 x 2 interior are loaded non-temporally
  __shared__ float4 Ps[PAIRED ? 2 : 1][NH_ROWS * 4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  unsigned t = block;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands of tiles
  const int n = (int)(t / (unsigned)tiles_per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)tiles_per_img);
  int ty, tx, oy, ox;  // own tile, the other half's tile
  bool idle = false;   // (an idle half still takes part in the barrier)
  if (PAIRED) {
    // tt < tiles_x (tiles_x - 1) / 2: the strict pairs i < j, one half each for (i, j) and (j, i); then the
    // diagonal tiles two by two, so that every workgroup has two full halves (at N = 4, 128 x 128 that makes
    // 2048 workgroups = exactly four residency rounds of 2 per CU: no tail)
    const int off = tiles_x * (tiles_x - 1) / 2;
    if (tt < off) {
      int pi, pj;
      pair_walk(tt, tiles_x, strip, pi, pj);
      ty = half ? pj : pi;
      tx = half ? pi : pj;
      oy = tx;
      ox = ty;
    } else {
      ty = tx = 2 * (tt - off) + half;
      oy = ox = 2 * (tt - off) + (half ^ 1);
      idle = ty >= tiles_x;
    }
  } else {
    ty = tt / tiles_x;
    tx = tt - ty * tiles_x;
    oy = ty;
    ox = tx;
  }
  if (LEAN) {
    // The same kernel with 32-bit index arithmetic (per-image base pointers, unsigned byte offsets: the launcher
    // sends images of 4 GB and more through the other form) and the taps as (row, column) pairs instead of plane
    // offsets that have to be divided by W again: the first form issues ~2.3 instructions for every one here, and a
    // compute unit issues one scalar instruction per cycle for all its waves (PMC, r3_frb.hip's gather: same finding).
    const int h = ty * NH_ROWS + wave;
    idle = idle || h >= H;
    const int HW = H * W, C4 = C >> 2;
    const bool two = FUSED && b != nullptr, has_res = FUSED && res != nullptr;
    const int w0 = tx * 4, cnt = idle ? 0 : min(4, W - w0);
    const size_t imgB = (size_t)n * HW * C * 4;
    const char* aI = reinterpret_cast<const char*>(a) + imgB;
    const char* bI = two ? reinterpret_cast<const char*>(b) + imgB : aI;
    const char* rI = has_res ? reinterpret_cast<const char*>(res) + imgB : aI;
    char* oI = reinterpret_cast<char*>(out) + imgB;
    const float* bxI = boxes + (size_t)n * HW * 5;
    const unsigned rowB = (unsigned)C * 4u;
    const unsigned q0 = idle ? 0u : (unsigned)(h * W + w0);
    const int own = half * 16 + wave * 4, halfX = PAIRED ? half << 4 : 0;
    float4 (*Pf)[64] = &Ps[0][0];  // slot (half * 16 + row of the tile * 4 + column) x lane
    const int ty4 = ty * NH_ROWS, tx4 = tx * 4, oy4 = oy * NH_ROWS, ox4 = ox * 4;
    if (EMIT && tab) fr_emit_tab(tab + (size_t)n * 2 * HW, bxI, HW, q0, cnt, lane, scale, H, W);
    for (int c0 = 0; c0 < C4; c0 += 64) {  // (wave-uniform trip count: the barriers are inside)
      const bool cl = c0 + lane < C4;
      const unsigned laneB = (unsigned)(c0 + lane) * 16u;
      float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
      if (FUSED && cl) {
        if (bias_a) ba = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_a) + laneB);
        if (bias_b) bb = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_b) + laneB);
      }
      auto mixv = [&](const float4& x, const float4& y) -> float4 {  // (x + bias_a) + (y + bias_b), or x alone
        float4 v = x;
        if (FUSED) {
          v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
          if (two) {
            float4 u = y;
            u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
          }
        }
        return v;
      };
      {
        float4 ia[4], ib[4];
#pragma unroll
        for (int i = 0; i < 4; i++) ia[i] = ib[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        // The 2 x 2 interior of a tile is asked for by no other workgroup as long as a box samples within one cell
        // of its own transposed position: those rows need not stay in the L2, and a non-temporal load lands
        // sooner.  Level 0, N = 4, rotating buffers, same run: 58.1 -> 53.9 us (FETCH 124.1 -> 122.2 K).  The same
        // hint on ALL identity rows: 62.0 us (the neighbours' tap rows are gone from the L2: FETCH 133.2 K); on
        // the out-of-tile tap rows: 58.9 us alone, 55.9 us with the interior hint (FETCH 118.6 K, but no faster);
        // the residual row of position i + 1 requested before position i is worked on (63 VGPRs): +1.5 us.
        // The eight loads are ONE straight-line block per kind of wave (interior positions = a compile-time mask,
        // rows clamped into the tile: a short last tile loads a row twice): with a scalar branch per row the compiler
        // put a wait between the loads (see fr_forward_nhwc_wide).
        auto load8 = [&](auto mask_tag) {
          constexpr int M = decltype(mask_tag)::value;
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const unsigned off = (q0 + (unsigned)min(i, max(cnt - 1, 0))) * rowB + laneB;
            if ((M >> i) & 1) {
              const fr_v4 ta = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(aI + off));
              ia[i] = make_float4(ta.x, ta.y, ta.z, ta.w);
              if (two) {
                const fr_v4 tb = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(bI + off));
                ib[i] = make_float4(tb.x, tb.y, tb.z, tb.w);
              }
            } else {
              ia[i] = *reinterpret_cast<const float4*>(aI + off);
              if (two) ib[i] = *reinterpret_cast<const float4*>(bI + off);
            }
          }
        };
        if (cl && cnt > 0) {
          if (NTI && (wave == 1 || wave == 2)) load8(std::integral_constant<int, 6>{});
          else load8(std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int i = 0; i < 4; i++) Pf[own + i][lane] = mixv(ia[i], ib[i]);
      }
      __syncthreads();
      // A tap's value: from the workgroup's tiles (LDS), else two row loads.  `fetch` only REQUESTS (raw rows, not yet
      // mixed), `fin` mixes: two taps are fetched before the first is finished, so that a position whose taps lie
      // outside the tiles waits for two L2 round trips instead of four (tools/fr_offset_sweep.py).
      auto fetch = [&](const int y, const int x, float4& xa, float4& xb) -> bool {  // y, x wave-uniform, inside the map
        const int ly = y - ty4, lx = x - tx4;
        if ((unsigned)ly < (unsigned)NH_ROWS && (unsigned)lx < 4u) { xa = Pf[(half << 4) + ly * 4 + lx][lane]; return false; }
        if (PAIRED) {
          const int my = y - oy4, mx = x - ox4;
          if ((unsigned)my < (unsigned)NH_ROWS && (unsigned)mx < 4u) { xa = Pf[(((half ^ 1) << 4)) + my * 4 + mx][lane]; return false; }
        }
        const unsigned off = (unsigned)(y * W + x) * rowB + laneB;
        xa = *reinterpret_cast<const float4*>(aI + off);
        if (two) xb = *reinterpret_cast<const float4*>(bI + off);
        return true;
      };
      auto fin = [&](const bool raw, const float4& xa, const float4& xb) -> float4 { return raw ? mixv(xa, xb) : xa; };
      (void)halfX;
#pragma unroll 1
      for (int i = 0; i < cnt; i++) {
        const unsigned q = q0 + (unsigned)i;
        const float bx = bxI[q * 5u], by = bxI[q * 5u + 1u];  // (uniform addresses: scalar loads)
        if (!cl) continue;
        const unsigned off = q * rowB + laneB;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_res) {
          if (NT) {
            const fr_v4 t4 = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(rI + off));
            r = make_float4(t4.x, t4.y, t4.z, t4.w);
          } else {
            r = *reinterpret_cast<const float4*>(rI + off);
          }
        }
        const TapYX tp = make_tap_yx(H, W, bx * scale, by * scale);  // sic: row <- x_ctr, column <- y_ctr
        // (the box is wave-uniform but its float arithmetic runs in the vector unit: hand the four cell
        // coordinates back to scalar registers, so that "which tile holds this tap" below is a scalar branch and not
        // exec-mask bookkeeping in every lane)
        const int yl = __builtin_amdgcn_readfirstlane(tp.yl), xl = __builtin_amdgcn_readfirstlane(tp.xl);
        const int yh = __builtin_amdgcn_readfirstlane(tp.yh), xh = __builtin_amdgcn_readfirstlane(tp.xh);
        float4 v = Pf[own + i][lane];
        // (a sample outside the map reads nothing, as in the reference: its taps would all be cell (0, 0), one row
        // requested by every wave of the chip -- a field of such boxes ran 65 us instead of 54)
        if (__builtin_amdgcn_readfirstlane((int)tp.valid)) {
          float4 xa0, xb0 = make_float4(0.f, 0.f, 0.f, 0.f), xa1, xb1 = xb0;
          bool g0 = fetch(yl, xl, xa0, xb0), g1 = fetch(yl, xh, xa1, xb1);
          const float4 lt = fin(g0, xa0, xb0), rt = fin(g1, xa1, xb1);
          g0 = fetch(yh, xl, xa0, xb0);
          g1 = fetch(yh, xh, xa1, xb1);
          const float4 lb = fin(g0, xa0, xb0), rb = fin(g1, xa1, xb1);
          float4 sm;
          sm.x = tp.w[0] * lt.x + tp.w[1] * rt.x + tp.w[2] * lb.x + tp.w[3] * rb.x;
          sm.y = tp.w[0] * lt.y + tp.w[1] * rt.y + tp.w[2] * lb.y + tp.w[3] * rb.y;
          sm.z = tp.w[0] * lt.z + tp.w[1] * rt.z + tp.w[2] * lb.z + tp.w[3] * rb.z;
          sm.w = tp.w[0] * lt.w + tp.w[1] * rt.w + tp.w[2] * lb.w + tp.w[3] * rb.w;
          v.x += sm.x; v.y += sm.y; v.z += sm.z; v.w += sm.w;
        }
        if (has_res) { v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; }
        if (NT) {
          const fr_v4 t4 = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(t4, reinterpret_cast<fr_v4*>(oI + off));
        } else {
          *reinterpret_cast<float4*>(oI + off) = v;
        }
      }
      if (c0 + 64 < C4) __syncthreads();  // the next channel block overwrites Ps
    }
    return;
  }
  const int h = ty * NH_ROWS + wave;
  idle = idle || h >= H;
  const int HW = H * W, C4 = C >> 2;
  const size_t img = (size_t)n * HW;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* o4 = reinterpret_cast<float4*>(out);
  const bool two = FUSED && b != nullptr, has_res = FUSED && res != nullptr;
  const int w0 = tx * 4, cnt = idle ? 0 : min(4, W - w0);
  if (EMIT && tab)  // (images of 4 GB and more have fewer than 2^32 positions all the same)
    fr_emit_tab(tab + (size_t)n * 2 * HW, boxes + img * 5, HW, idle ? 0u : (unsigned)(h * W + w0), cnt, lane, scale, H, W);
  for (int c0 = 0; c0 < C4; c0 += 64) {  // (wave-uniform trip count: the barriers are inside)
    const int c4 = c0 + lane;
    const bool cl = c4 < C4;
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
    if (FUSED && cl) {
      if (bias_a) ba = reinterpret_cast<const float4*>(bias_a)[c4];
      if (bias_b) bb = reinterpret_cast<const float4*>(bias_b)[c4];
    }
    auto mixv = [&](const float4& x, const float4& y) -> float4 {  // (x + bias_a) + (y + bias_b), or x alone
      float4 v = x;
      if (FUSED) {
        v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
        if (two) {
          float4 u = y;
          u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      return v;
    };
    // phase 1: the identity streams of the wave's 4 positions and their 4 boxes, all in flight together; P to LDS
    float bxs[4], bys[4];  // (wave-uniform: scalar registers)
    {
      float4 ia[4], ib[4];
      float bxv[4], byv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const size_t q = img + (size_t)(idle ? 0 : h) * W + w0 + min(i, max(cnt - 1, 0));
        const bool on = i < cnt && cl;
        ia[i] = on ? a4[q * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        ib[i] = (on && two) ? b4[q * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        bxv[i] = boxes[q * 5];
        byv[i] = boxes[q * 5 + 1];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) Ps[half][wave * 4 + i][lane] = mixv(ia[i], ib[i]);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        bxs[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bxv[i])));
        bys[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, byv[i])));
      }
    }
    __syncthreads();
    // phase 2, one position per step: residual row requested first, the identity term back from LDS, then the taps
    auto P = [&](const int y, const int x) -> float4 {  // y, x wave-uniform, inside the map
      const int ly = y - ty * NH_ROWS, lx = x - tx * 4;
      if ((unsigned)ly < (unsigned)NH_ROWS && (unsigned)lx < 4u) return Ps[half][ly * 4 + lx][lane];
      if (PAIRED) {
        const int my = y - oy * NH_ROWS, mx = x - ox * 4;
        if ((unsigned)my < (unsigned)NH_ROWS && (unsigned)mx < 4u) return Ps[half ^ 1][my * 4 + mx][lane];
      }
      const size_t q = img + (size_t)y * W + x;
      return mixv(a4[q * C4 + c4], two ? b4[q * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f));
    };
    auto position = [&](const int i, const float bx, const float by) {
      const size_t q = img + (size_t)h * W + w0 + i;
      if (!cl) return;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_res) {
        if (NT) {
          const fr_v4 t4 = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(&r4[q * C4 + c4]));
          r = make_float4(t4.x, t4.y, t4.z, t4.w);
        } else {
          r = r4[q * C4 + c4];
        }
      }
      const Tap tp = make_tap(H, W, W, bx * scale, by * scale);  // sic: row <- x_ctr, column <- y_ctr
      float4 v = Ps[half][wave * 4 + i][lane];
      const int y0 = tp.o00 / W, x0 = tp.o00 - y0 * W, y1 = tp.o11 / W, x1 = tp.o11 - y1 * W;
      const float4 lt = P(y0, x0), rt = P(y0, x1), lb = P(y1, x0), rb = P(y1, x1);
      float4 sm;
      sm.x = tp.w1 * lt.x + tp.w2 * rt.x + tp.w3 * lb.x + tp.w4 * rb.x;
      sm.y = tp.w1 * lt.y + tp.w2 * rt.y + tp.w3 * lb.y + tp.w4 * rb.y;
      sm.z = tp.w1 * lt.z + tp.w2 * rt.z + tp.w3 * lb.z + tp.w4 * rb.z;
      sm.w = tp.w1 * lt.w + tp.w2 * rt.w + tp.w3 * lb.w + tp.w4 * rb.w;
      if (tp.valid) { v.x += sm.x; v.y += sm.y; v.z += sm.z; v.w += sm.w; }
      if (has_res) { v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; }
      if (NT) {
        const fr_v4 t4 = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t4, reinterpret_cast<fr_v4*>(&o4[q * C4 + c4]));
      } else {
        o4[q * C4 + c4] = v;
      }
    };
    if (PRE) {
      if (cnt > 0) position(0, bxs[0], bys[0]);
      if (cnt > 1) position(1, bxs[1], bys[1]);
      if (cnt > 2) position(2, bxs[2], bys[2]);
      if (cnt > 3) position(3, bxs[3], bys[3]);
    } else {
#pragma unroll 1
      for (int i = 0; i < cnt; i++) {
        const float* bp = boxes + (img + (size_t)h * W + w0 + i) * 5;
        position(i, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bp[0]))),
                 __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bp[1]))));
      }
    }
    if (c0 + 64 < C4) __syncthreads();  // the next channel block overwrites Ps
  }
}

template <bool FUSED, bool PAIRED, int VAR = 0, bool EMIT = false>
__global__ __launch_bounds__(PAIRED ? 512 : 256) void fr_forward_nhwc_occ(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias_a,
    const float* __restrict__ bias_b, const float* __restrict__ res, const float* __restrict__ boxes, int C, int H, int W,
    float scale, int tiles_xs, int tiles_per_img, int T, float* __restrict__ out, float* __restrict__ tab = nullptr) {
  fr_forward_nhwc_occ_body<FUSED, PAIRED, VAR, EMIT>(a, b, bias_a, bias_b, res, boxes, C, H, W, scale, tiles_xs, tiles_per_img,
                                                     T, out, blockIdx.x, tab);
}

// Several pyramid levels of the channels_last sampler / module tail as ONE grid (the levels in kernel arguments, a
// block finds its level from the block ranges): FeatureRefineModule.forward loops over the levels
// (fr/feature_refine_module.py:108-127) and the four coarse levels are launch-bound on their own -- 52.6 us in the
// model for a third of level 0's bytes.
constexpr int FRL_MAX = 8;
struct FrNhwcLevel {
  const float *a, *b, *res, *boxes;
  float* out;
  float* tab;  // the level's tap table to write (training) or null
  float scale;
  int H, W, tiles_xs, tiles_per_img, T, first;
};
struct FrNhwcLevels {
  FrNhwcLevel l[FRL_MAX];
  int n;
};

template <bool FUSED, bool EMIT = false>
__global__ __launch_bounds__(512) void fr_forward_nhwc_occ_levels(const FrNhwcLevels A, const float* __restrict__ bias_a,
                                                                  const float* __restrict__ bias_b, int C) {
  int lv = 0;
#pragma unroll
  for (int i = 1; i < FRL_MAX; i++)
    if (i < A.n && (int)blockIdx.x >= A.l[i].first) lv = i;
  const FrNhwcLevel& L = A.l[lv];
  fr_forward_nhwc_occ_body<FUSED, true, 14, EMIT>(L.a, L.b, bias_a, bias_b, L.res, L.boxes, C, L.H, L.W, L.scale, L.tiles_xs,
                                                  L.tiles_per_img, L.T, L.out, blockIdx.x - (unsigned)L.first, L.tab);
}

// "Wide" form of the kernel above for square maps whose side is a multiple of 8: 64 positions and 16 waves per
// workgroup instead of 32 and 8, still two residency slots' worth of LDS per position (64 KB, two workgroups per CU).
// What it buys: a region of 4 x 8 cells has 12 interior positions of 32 (an 8 x 8 one 36 of 64) against 4 of 16 for
// a 4 x 4 tile -- more identity rows that no other workgroup will ask for (non-temporal loads) -- and 30 % (53 %)
// fewer taps that leave the workgroup's regions, i.e. fewer rows fetched twice.
//   off-diagonal 8 x 8 super-blocks (I < J): a workgroup owns the 4 x 8 region A = rows 8I + 4k .. + 3, columns
//     8J .. 8J + 7 (k = 0, 1) and its TRANSPOSE, the 8 x 4 region B = rows 8J .. 8J + 7, columns 8I + 4k .. + 3 of
//     super-block (J, I): the sources of A sample B and the other way round (row <- x_ctr, column <- y_ctr);
//   diagonal super-blocks: one workgroup owns the whole 8 x 8 block, which is its own transpose.
// Slots of the LDS map: A (or the 8 x 8 block) row-major with 8 columns, then B row-major with 4 columns at slot 32.
// A wave owns 4 positions of one row.  Same arithmetic, same operation order as the other forms (bit-identical).
// Level 0, N = 4, rotating buffers, A/B inside one run (tools/fr_fwd_var_ab.py, option fr_dbg 8 / 9): as first
// written HBM traffic FETCH x 2 + WRITE 317 -> 288 MB = 1.07 x algorithmic, but 55.8 - 57.6 us against 53.9 - 55.0 for
// the 4 x 4 pairs; with the residual rows and boxes requested in front of the barrier (below) 53.4 - 55.1 us, the same
// as the pairs, at 311 MB = 1.15 x (more rows in flight turn the L2 over faster: some of the saved re-fetches return).
// EMIT: the launch also writes the level's tap table (training steps; fr_emit_tab)
template <bool FUSED, bool TWO = false, bool RES = false, bool EMIT = false>  // TWO: a second addend b; RES: a residual (both FUSED only)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void fr_forward_nhwc_wide(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias_a,
    const float* __restrict__ bias_b, const float* __restrict__ res, const float* __restrict__ boxes, int C, int H, int W,
    float scale, int S_strip, int per_img, int T, float* __restrict__ out, float* __restrict__ tab = nullptr) {
  __shared__ float4 Pw[64][64];  // slot x lane
  const int S = S_strip & 0xfffff, strip = S_strip >> 20;  // super-blocks per side; the pair walk's strip height
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned t = blockIdx.x;
  if ((T & 7) == 0) t = (t & 7u) * (unsigned)(T >> 3) + (t >> 3);  // XCD-contiguous bands
  const int n = (int)(t / (unsigned)per_img);
  const int tt = (int)(t - (unsigned)n * (unsigned)per_img);
  int y0a, x0a, ra, y0b, x0b, rb;  // region A: ra x 8 cells at (y0a, x0a); region B: rb x 4 cells at (y0b, x0b)
  const int offd = S * (S - 1);
  if (tt < offd) {
    int I, J;
    pair_walk(tt >> 1, S, strip, I, J);
    const int k4 = (tt & 1) * 4;
    y0a = 8 * I + k4; x0a = 8 * J; ra = 4;
    y0b = 8 * J; x0b = 8 * I + k4; rb = 8;
  } else {
    y0a = x0a = 8 * (tt - offd); ra = 8;
    y0b = x0b = 0; rb = 0;
  }
  // the wave's row and first column, its first slot, and which of its 4 positions are interior to their region
  const bool inB = rb != 0 && wave >= 8;
  const int wl = inB ? wave - 8 : wave;
  const int py = inB ? y0b + wl : y0a + (wl >> 1);
  const int px0 = inB ? x0b : x0a + (wl & 1) * 4;
  const int slot0 = inB ? 32 + wl * 4 : (wl >> 1) * 8 + (wl & 1) * 4;
  const int lrow = inB ? wl : (wl >> 1), nrow = inB ? 8 : ra;
  const bool row_inner = lrow >= 1 && lrow <= nrow - 2;
  // (columns: B has 4 -> positions 1, 2; A has 8 -> the left wave's 1..3, the right wave's 0..2)
  const int HW = H * W, C4 = C >> 2;
  constexpr bool two = FUSED && TWO, has_res = FUSED && RES;  // (compile-time: straight-line load blocks)
  const size_t imgB = (size_t)n * HW * C * 4;
  const char* aI = reinterpret_cast<const char*>(a) + imgB;
  const char* bI = two ? reinterpret_cast<const char*>(b) + imgB : aI;
  const char* rI = has_res ? reinterpret_cast<const char*>(res) + imgB : aI;
  char* oI = reinterpret_cast<char*>(out) + imgB;
  const float* bxI = boxes + (size_t)n * HW * 5;
  const unsigned rowB = (unsigned)C * 4u;
  const unsigned q0 = (unsigned)(py * W + px0);
  if (EMIT && tab) fr_emit_tab(tab + (size_t)n * 2 * HW, bxI, HW, q0, 4, lane, scale, H, W);
  for (int c0 = 0; c0 < C4; c0 += 64) {  // (wave-uniform trip count: the barriers are inside)
    const bool cl = c0 + lane < C4;
    const unsigned laneB = (unsigned)(c0 + lane) * 16u;
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
    if (FUSED && cl) {
      if (bias_a) ba = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_a) + laneB);
      if (bias_b) bb = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias_b) + laneB);
    }
    auto mixv = [&](const float4& x, const float4& y) -> float4 {  // (x + bias_a) + (y + bias_b), or x alone
      float4 v = x;
      if (FUSED) {
        v.x += ba.x; v.y += ba.y; v.z += ba.z; v.w += ba.w;
        if (two) {
          float4 u = y;
          u.x += bb.x; u.y += bb.y; u.z += bb.z; u.w += bb.w;
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      return v;
    };
    {
      float4 ia[4], ib[4];
#pragma unroll
      for (int i = 0; i < 4; i++) ia[i] = ib[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      // The eight identity rows in ONE straight-line block per kind of wave (which of its positions are interior to
      // the region -- non-temporal -- is a compile-time mask): with a scalar branch per row ("interior?") the compiler
      // put a wait in front of every second load and a wave had two or three rows in flight instead of eight.
      auto load8 = [&](auto mask_tag) {
        constexpr int M = decltype(mask_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned off = (q0 + (unsigned)i) * rowB + laneB;
          if ((M >> i) & 1) {
            const fr_v4 ta = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(aI + off));
            ia[i] = make_float4(ta.x, ta.y, ta.z, ta.w);
            if (two) {
              const fr_v4 tb = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(bI + off));
              ib[i] = make_float4(tb.x, tb.y, tb.z, tb.w);
            }
          } else {
            ia[i] = *reinterpret_cast<const float4*>(aI + off);
            if (two) ib[i] = *reinterpret_cast<const float4*>(bI + off);
          }
        }
      };
      if (cl) {
        if (!row_inner) load8(std::integral_constant<int, 0>{});
        else if (inB) load8(std::integral_constant<int, 6>{});            // positions 1, 2
        else if (wl & 1) load8(std::integral_constant<int, 7>{});        // the right wave of a row: 0, 1, 2
        else load8(std::integral_constant<int, 14>{});                   // the left wave: 1, 2, 3
      }
#pragma unroll
      for (int i = 0; i < 4; i++) Pw[slot0 + i][lane] = mixv(ia[i], ib[i]);
    }
    // The four residual rows and the four boxes are requested HERE, when the identity rows' registers are free again
    // (in front of the LDS writes they would be live next to them: 72 VGPRs), so that they arrive while the workgroup
    // waits at the barrier; the position loop stays rolled (unrolled, the scheduler hoists and needs 71 VGPRs), the
    // rows rotate through rq[0].  With two 16-wave workgroups per CU few other workgroups are in their load phase
    // while this one works through its positions: without this the wide form ran 55.8 - 57.6 us, with it 53.4 - 55.1
    // (the 4 x 4 pairs in the same runs: 53.2 - 54.8; they do not gain from it: 72 VGPRs, 56 - 59 us).
    float4 rq[4];
    float bxq[4], byq[4];  // (uniform addresses: scalar loads, scalar registers)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      bxq[i] = bxI[(q0 + (unsigned)i) * 5u];
      byq[i] = bxI[(q0 + (unsigned)i) * 5u + 1u];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      rq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_res && cl) {
        const fr_v4 t4 = __builtin_nontemporal_load(reinterpret_cast<const fr_v4*>(rI + ((q0 + (unsigned)i) * rowB + laneB)));
        rq[i] = make_float4(t4.x, t4.y, t4.z, t4.w);
      }
    }
    __syncthreads();
    auto P = [&](const int y, const int x) -> float4 {  // y, x wave-uniform, inside the map
      const int ly = y - y0a, lx = x - x0a;
      if ((unsigned)ly < (unsigned)ra && (unsigned)lx < 8u) return Pw[ly * 8 + lx][lane];
      const int my = y - y0b, mx = x - x0b;
      if ((unsigned)my < (unsigned)rb && (unsigned)mx < 4u) return Pw[32 + my * 4 + mx][lane];
      const unsigned off = (unsigned)(y * W + x) * rowB + laneB;
      return mixv(*reinterpret_cast<const float4*>(aI + off),
                  two ? *reinterpret_cast<const float4*>(bI + off) : make_float4(0.f, 0.f, 0.f, 0.f));
    };
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const unsigned q = q0 + (unsigned)i;
      const float bx = bxq[0], by = byq[0];
      bxq[0] = bxq[1]; bxq[1] = bxq[2]; bxq[2] = bxq[3];
      byq[0] = byq[1]; byq[1] = byq[2]; byq[2] = byq[3];
      if (!cl) continue;
      const unsigned off = q * rowB + laneB;
      const float4 r = rq[0];
      rq[0] = rq[1];
      rq[1] = rq[2];
      rq[2] = rq[3];
      const TapYX tp = make_tap_yx(H, W, bx * scale, by * scale);  // sic: row <- x_ctr, column <- y_ctr
      const int yl = __builtin_amdgcn_readfirstlane(tp.yl), xl = __builtin_amdgcn_readfirstlane(tp.xl);
      const int yh = __builtin_amdgcn_readfirstlane(tp.yh), xh = __builtin_amdgcn_readfirstlane(tp.xh);
      float4 v = Pw[slot0 + i][lane];
      if (__builtin_amdgcn_readfirstlane((int)tp.valid)) {  // (a sample outside the map reads nothing)
        // (measured, not kept: two taps requested raw before the first is mixed, so that out-of-region taps wait
        // for two L2 round trips instead of four: 48 B of scratch at the 64-VGPR cap, 89 us)
        const float4 lt = P(yl, xl), rt = P(yl, xh), lb = P(yh, xl), rbv = P(yh, xh);
        float4 sm;
        sm.x = tp.w[0] * lt.x + tp.w[1] * rt.x + tp.w[2] * lb.x + tp.w[3] * rbv.x;
        sm.y = tp.w[0] * lt.y + tp.w[1] * rt.y + tp.w[2] * lb.y + tp.w[3] * rbv.y;
        sm.z = tp.w[0] * lt.z + tp.w[1] * rt.z + tp.w[2] * lb.z + tp.w[3] * rbv.z;
        sm.w = tp.w[0] * lt.w + tp.w[1] * rt.w + tp.w[2] * lb.w + tp.w[3] * rbv.w;
        v.x += sm.x; v.y += sm.y; v.z += sm.z; v.w += sm.w;
      }
      if (has_res) { v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; }
      const fr_v4 t4 = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(t4, reinterpret_cast<fr_v4*>(oI + off));
    }
    if (c0 + 64 < C4) __syncthreads();  // the next channel block overwrites Pw
  }
}

// ----------------------------------------------------------------------------------------
// "cell" forward kernel (points = 1, W x H a compile-time power-of-two shape).
// What the plane kernel pays per (n, c) plane besides the plane itself is the per-position sample
// data (the 20-byte box, or any tap record derived from it) re-read through the CU's vector-memory
// path for every one of the C planes: +12.7 us on a 20 us copy at level 0
// (tools/probes/fr_stage_probe.hip).  Here a workgroup owns G consecutive channels of ONE image
// and keeps the taps of all H x W positions in registers for the whole launch (2 VGPR per
// position, K = H*W/1024 positions per thread); planes stream through two LDS buffers.
// With the memory side fixed the limit became INSTRUCTION ISSUE (a first version of this idea
// spent ~44 VALU/LDS instructions per position and plane: 16 positions x 4 waves per SIMD x
// 4 cycles = 6.9 us per plane and CU against a 4.8 us HBM share).  This kernel needs ~28:
//   * LDS plane layout: row pitch W + 1 where the pad word DUPLICATES the last column, plus a
//     duplicate of the last row and two all-zero rows.  The reference's clamped neighbours
//     (x_high = x_low at the right edge, y_high = y_low at the bottom,
//     feature_refine_kernel.cu:33-47) are then always at a0 + 1 / a0 + pitch, so ONE address per
//     position serves two ds_read2_b32 (immediate offsets), with no dx / dy flags; an invalid
//     sample points at the zero cell with weights (1, 0, 0, 0), so there is no validity select.
//   * taps: 8 B per position (clamped y, x; fr_cell_table_kernel).  cell = cvt_i32, fraction =
//     v_fract (exact: y - floor(y) is representable for 0 <= y < 2^23), address = one mad.
//   * a lane owns CONSECUTIVE positions of a row (4 B per lane, 256 B per wave access): its
//     sample rows step by one => LDS stride W + 1 (odd): gathers, staging writes and the identity
//     read are all bank-conflict free; their LDS addresses are tid-based with immediate offsets.
//   * weights in packed fp32: {w1, w2} = hy * {hx, lx}, {w3, w4} = ly * {hx, lx}; products and
//     the left-to-right sum keep the reference's operation order (bit-identical results).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_cell_table_kernel(const float* __restrict__ boxes, int N, int H, int W,
                                                            float scale, float* __restrict__ table) {
  fr_cell_table_body(boxes, N, H, W, scale, table, blockIdx.x * 256 + threadIdx.x);
}


// FROM_BOXES: `table` is the (N*H*W, 5) box array itself and the taps are derived in the prologue
// (no table kernel, no dependent launch: worth ~6 us per call at N = 4, where the table kernel
// plus the gap to this one cost as much as a quarter of this kernel).
// FUSED: the module's two elementwise passes around the sampler folded in (feature_refine_module.py:
// 121-126): the sampled plane is feat + feat2 (= conv_5_1(conv_1_5(x)) + conv_1_1(x)), and the result
// gets the module's residual, out = res + (plane + sample(plane)), in the reference's operation order.
// 3 reads + 1 write per element instead of (2r + 1w) + (1r + 1w) + (2r + 1w) over three launches.
template <int LOGW, int LOGH, int THREADS, bool FROM_BOXES, int FUSED = 0, bool NTP = (LOGW >= 7)>
__global__ __launch_bounds__(THREADS) void fr_forward_cell(const float* __restrict__ feat,
                                                           const float* __restrict__ table, int C, int G,
                                                           float scale, float* __restrict__ out,
                                                           const float* __restrict__ feat2,
                                                           const float* __restrict__ res) {
  constexpr int FRC_BLOCK = THREADS;
  static_assert(THREADS >= 2 * ((1 << LOGW) + 1) && THREADS >= (1 << LOGW) + (1 << LOGH) + 1, "helper threads");
  constexpr int W = 1 << LOGW, H = 1 << LOGH, HW = W * H, K = HW / FRC_BLOCK, PITCH = W + 1;
  constexpr int BUF = ((H + 3) * PITCH + 3) & ~3;
  constexpr int KSTEP = FRC_BLOCK + FRC_BLOCK / W;  // LDS distance between a thread's positions k, k + 1
  static_assert(K >= 1 && W <= FRC_BLOCK, "shape");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int groups = C / G;
  const int n = blockIdx.x / groups;
  const int c0 = (blockIdx.x - n * groups) * G;
  const size_t plane0 = (size_t)n * C + c0;
  const float* ty_g = table + (size_t)n * 2 * HW;

  float ty[K], tx[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (FROM_BOXES) {
      const float* bp = table + ((size_t)n * HW + tid + k * FRC_BLOCK) * 5;
      ty[k] = bp[0];
      tx[k] = bp[1];
    } else {
      ty[k] = ty_g[tid + k * FRC_BLOCK];
      tx[k] = ty_g[HW + tid + k * FRC_BLOCK];
    }
  }
  if (FROM_BOXES) {
#pragma unroll
    for (int k = 0; k < K; k++) cell_tap(ty[k] * scale, tx[k] * scale, H, W, ty[k], tx[k]);
  }
  const int self0 = tid + (tid >> LOGW);
  // the H + W + 1 duplicate words are staged by the first threads (one extra element each)
  const bool dup = tid < H + W + 1;
  int dsrc = 0, ddst = 0;
  if (tid < H) {
    dsrc = tid * W + W - 1;
    ddst = tid * PITCH + W;
  } else if (dup) {
    const int j = tid - H;
    dsrc = (H - 1) * W + min(j, W - 1);
    ddst = H * PITCH + j;
  }
  if (tid < 2 * PITCH) {  // the zero cell rows of both buffers, written once
    lds[(H + 1) * PITCH + tid] = 0.f;
    lds[BUF + (H + 1) * PITCH + tid] = 0.f;
  }
  // Plane loads as a ROLLING pipeline.  The CU's vector-memory queue is in order: a burst of all
  // the next plane's loads at the start of a phase blocks the stores behind it, the sampling waves
  // stall on their first store until the burst has drained, and load transfer and sampling end up
  // back to back (measured: 2.5 + 4.6 = 7.1 us per plane).  Instead element k + D of the stream is
  // requested while position k is sampled, and element k is written to the idle LDS buffer D
  // positions after its request: one load, one store and one LDS write per position, no bursts.
  constexpr int D = K > 1 ? K / 2 : 0;  // (3K/4 measured the same cold and 4 % slower warm)
  // every plane element is read once and written once by the whole launch: non-temporal at 128 x 128 (NTP; level 0,
  // N = 4, rotating buffers: 33.2 -> 32.3 us; at 64 x 64 no gain, tools/fr_nchw_nt_ab.py)
  auto LD = [](const float* p) -> float { return NTP ? __builtin_nontemporal_load(p) : *p; };
  auto ST = [](float* p, const float x) {
    if (NTP) __builtin_nontemporal_store(x, p);
    else *p = x;
  };
  float v[K], vd = 0.f;
  constexpr bool TWO = FUSED == 1, RES = FUSED != 0;  // FUSED 2: the residual only (the plane is already summed)
  float v2[K], vd2 = 0.f, vr[K];  // second addend of the plane, the residual of the output plane
  {
    const float* src = feat + (plane0 << (LOGW + LOGH));
    const float* src2 = TWO ? feat2 + (plane0 << (LOGW + LOGH)) : nullptr;
#pragma unroll
    for (int k = 0; k < K; k++) {
      v[k] = LD(&src[tid + k * FRC_BLOCK]);
      if (TWO) v2[k] = LD(&src2[tid + k * FRC_BLOCK]);
    }
    if (dup) {
      vd = LD(&src[dsrc]);
      if (TWO) vd2 = LD(&src2[dsrc]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) lds[self0 + k * KSTEP] = TWO ? v[k] + v2[k] : v[k];
    if (dup) lds[ddst] = TWO ? vd + vd2 : vd;
    src += HW;  // G >= 2 (launcher)
#pragma unroll
    for (int k = 0; k < D; k++) {
      v[k] = LD(&src[tid + k * FRC_BLOCK]);
      if (TWO) v2[k] = LD(&src2[HW + tid + k * FRC_BLOCK]);
      if (RES) vr[k] = LD(&res[(plane0 << (LOGW + LOGH)) + tid + k * FRC_BLOCK]);
    }
  }
  __syncthreads();
  // L1: plane c + 1 exists, L2: plane c + 2 exists (compile-time so that the waitcnt bookkeeping
  // of the steady-state loop sees straight-line code)
  auto phase = [&](int c, auto l1, auto l2) {
    constexpr bool L1 = decltype(l1)::value, L2 = decltype(l2)::value;
    const float* buf = lds + (c & 1) * BUF;
    float* nbuf = lds + ((c + 1) & 1) * BUF;
    const float* src1 = feat + ((plane0 + c + 1) << (LOGW + LOGH));
    const float* src21 = TWO ? feat2 + ((plane0 + c + 1) << (LOGW + LOGH)) : nullptr;
    const float* resc = RES ? res + ((plane0 + c) << (LOGW + LOGH)) : nullptr;
    float* dst = out + ((plane0 + c) << (LOGW + LOGH));
    if (L1 && dup) {
      vd = LD(&src1[dsrc]);
      if (TWO) vd2 = LD(&src21[dsrc]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (k + D < K) {
        if (L1) v[k + D] = LD(&src1[tid + (k + D) * FRC_BLOCK]);
        if (TWO && L1) v2[k + D] = LD(&src21[tid + (k + D) * FRC_BLOCK]);
        if (RES) vr[k + D] = LD(&resc[tid + (k + D) * FRC_BLOCK]);
      } else {
        if (L2) v[k + D - K] = LD(&src1[HW + tid + (k + D - K) * FRC_BLOCK]);
        if (TWO && L2) v2[k + D - K] = LD(&src21[HW + tid + (k + D - K) * FRC_BLOCK]);
        if (RES && L1) vr[k + D - K] = LD(&resc[HW + tid + (k + D - K) * FRC_BLOCK]);
      }
      float y = ty[k], x = tx[k];
      // opaque copies: without them LICM hoists cell / fractions / address of all K positions
      // out of the channel loop and spills
      asm volatile("" : "+v"(y), "+v"(x));
      const int yi = (int)y, xi = (int)x;
      const float fy = __builtin_amdgcn_fractf(y), fx = __builtin_amdgcn_fractf(x);
      const int a = (yi << LOGW) + yi + xi;  // yi * PITCH + xi without the quarter-rate v_mul_lo
      const fr_f2 top = {buf[a], buf[a + 1]};
      const fr_f2 bot = {buf[a + PITCH], buf[a + PITCH + 1]};
      const float id = buf[self0 + k * KSTEP];
      // 1.f - f == (float)(1. - (double)f) for every f in [0, 1): the double difference is exact
      // when f >= 2^-29 and both forms round to 1.0f below that, so the reference's double-typed
      // "1. - ly" (feature_refine_kernel.cu:53-54) needs no fp64 here
      const fr_f2 hf = {1.f - fx, fx};
      const fr_f2 pt = ((1.f - fy) * hf) * top;  // {w1 * v1, w2 * v2}
      const fr_f2 pb = (fy * hf) * bot;          // {w3 * v3, w4 * v4}
      const float val = (pt.x + pt.y + pb.x + pb.y);
      ST(&dst[tid + k * FRC_BLOCK], RES ? vr[k] + (id + val) : id + val);
      if (L1) nbuf[self0 + k * KSTEP] = TWO ? v[k] + v2[k] : v[k];
    }
    if (L1 && dup) nbuf[ddst] = TWO ? vd + vd2 : vd;
    __syncthreads();
  };
  int c = 0;
  for (; c + 2 < G; c++) phase(c, std::true_type{}, std::true_type{});
  phase(c, std::true_type{}, std::false_type{});
  phase(c + 1, std::false_type{}, std::false_type{});
}

// dynamic LDS above 64 KB has to be opted into once per kernel
template <typename K>
inline void allow_big_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            bytes);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Kernel-exact timing of the cell path for bench.py's roofline line: a ring of event quadruples
// (table start / stop, cell start / stop) attached to the launches themselves.
// fr_profile = 1: all four events; 2: only table start and cell stop (no event packets between the
// two kernels: the span then carries one event pair's overhead and the kernels' natural gap).
struct FrProfileSlot {
  hipEvent_t ev[4];
  int N, H, mode;
  bool used;
};
constexpr int FR_PROFILE_SLOTS = 512;
FrProfileSlot g_fr_prof[FR_PROFILE_SLOTS];
int g_fr_prof_count = 0;

inline FrProfileSlot* fr_profile_next(int N, int H) {
  if (g_fr_prof_count >= FR_PROFILE_SLOTS) return nullptr;  // ring full: later launches run untimed
  FrProfileSlot* s = &g_fr_prof[g_fr_prof_count];
  if (!s->used) {
    for (auto& e : s->ev)
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
    s->used = true;
  }
  s->N = N;
  s->H = H;
  s->mode = g_r3_fr_profile;
  g_fr_prof_count++;
  return s;
}

inline int plane_cpb(int C, int H, int W) {
  int psz = H * (W + 1);
  if (psz > FRP_LDS_FLOATS) return 0;
  int cpb = FRP_LDS_FLOATS / psz;
  if (cpb > C) cpb = C;
  return cpb < 1 ? 0 : cpb;
}

inline int cu_count() { return r3_cu_count(); }

}  // namespace

size_t r3k_fr_workspace_bytes(int N, int H, int W, int points) {
  if (N <= 0 || H <= 0 || W <= 0 || points <= 0) return 256;
  return ((size_t)N * H * W * points * 5 * sizeof(float) + 255) / 256 * 256 + 256;
}

int r3k_fr_forward(const float* feat, const float* boxes, int N, int C, int H, int W, float scale,
                   int points, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (points != 1 && points != 5) return -1;
  if (N == 0 || C == 0 || H == 0 || W == 0) return 0;
  int cpb = plane_cpb(C, H, W);
  const bool plane = g_r3_fr_impl != 1 && cpb > 0;
  // channels per workgroup for the cell kernel: a power of two, >= one workgroup per CU
  int G = 1;
  while (G * 2 <= 16 && C % (G * 2) == 0 && (size_t)N * C / (G * 2) >= (size_t)cu_count()) G *= 2;
  const bool cell_shape = (W == 128 && H == 128) || (W == 64 && H == 64);
  // cell is the default for 128 x 128 planes (level 0 of a 1024^2 input): 25.8 us at N = 4 against
  // plane 44 and a device copy of the same bytes at 20.8 (tools/probes/persist_copy_probe), and for
  // 64 x 64 planes (with the taps derived in its prologue: one launch, 12.7 us vs plane 13.7)
  const bool cell_auto = g_r3_fr_impl == 0;
  // taps from a table built by a first kernel (needs the workspace) or derived from the boxes in the
  // cell kernel's own prologue (fr_dbg 1 / 2 force one form)
  const bool have_ws = ws && ws_bytes >= r3k_fr_workspace_bytes(N, H, W, points) && aligned16(ws);
  // measured at N = 4 (tools/fr_dbg_sweep.py): 128 x 128 table 26.4 us vs boxes 27.2 (the 20-byte-strided box
  // reads make the prologue 2.5 x heavier than the 8-byte table); 64 x 64 boxes 12.7 vs table 15.6, plane 13.7
  const int dbg = r3_fr_dbg();  // (read once per call)
  const bool from_boxes = dbg == 2 || (dbg != 1 && W == 64) || !have_ws;
  if ((g_r3_fr_impl == 10 || cell_auto) && points == 1 && cell_shape && G >= 2 && aligned16(feat) && aligned16(out)) {
    float* table = reinterpret_cast<float*>(ws);
    const int total = N * H * W;
    static R3DeviceOnce once;
    if (once.first()) {
      allow_big_lds(fr_forward_cell<7, 7, 1024, false>, 160 * 1024);
      allow_big_lds(fr_forward_cell<7, 7, 1024, true>, 160 * 1024);
#ifdef R3_PROBES
      allow_big_lds(fr_forward_cell<7, 7, 1024, false, 0, false>, 160 * 1024);
      allow_big_lds(fr_forward_cell<7, 7, 1024, true, 0, false>, 160 * 1024);
#endif
    }
    const size_t lds = (size_t)2 * ((((size_t)H + 3) * (W + 1) + 3) & ~(size_t)3) * sizeof(float);
    // profiling mode (r3det_set_option("fr_profile", 1 | 2)): the launches carry their own start /
    // stop events, so the recorded durations are the kernels' and not the host's launch gaps
    FrProfileSlot* ps = g_r3_fr_profile ? fr_profile_next(N, H) : nullptr;
    const bool each = ps && ps->mode == 1 && !from_boxes;
    hipEvent_t t0 = ps ? ps->ev[0] : nullptr, t1 = each ? ps->ev[1] : nullptr;
    hipEvent_t c0 = (each || (ps && from_boxes)) ? ps->ev[from_boxes ? 0 : 2] : nullptr, c1 = ps ? ps->ev[3] : nullptr;
    if (ps) ps->mode = from_boxes ? 3 : ps->mode;  // 3: one kernel (events 0 and 3)
    const dim3 grid(N * C / G), block(1024);
    if (!from_boxes)
      hipExtLaunchKernelGGL(fr_cell_table_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t0, t1, 0, boxes, N,
                            H, W, scale, table);
    // (64 x 64 with 512- or 256-thread workgroups, several per CU, measured 1-5 % slower than 1024)
#ifdef R3_PROBES  // (fr_dbg 21: plain instead of non-temporal plane loads / stores, tools/fr_nchw_nt_ab.py)
#define R3_CELL(LW, LH, FB, SRC) \
  do { \
    if (dbg == 21) hipExtLaunchKernelGGL((fr_forward_cell<LW, LH, 1024, FB, 0, false>), grid, block, lds, stream, c0, c1, 0, feat, SRC, C, G, scale, out, (const float*)nullptr, (const float*)nullptr); \
    else hipExtLaunchKernelGGL((fr_forward_cell<LW, LH, 1024, FB>), grid, block, lds, stream, c0, c1, 0, feat, SRC, C, G, scale, out, (const float*)nullptr, (const float*)nullptr); \
  } while (0)
#else
#define R3_CELL(LW, LH, FB, SRC) \
  hipExtLaunchKernelGGL((fr_forward_cell<LW, LH, 1024, FB>), grid, block, lds, stream, c0, c1, 0, feat, SRC, C, G, scale, out, (const float*)nullptr, (const float*)nullptr)
#endif
    if (W == 128) { if (from_boxes) R3_CELL(7, 7, true, boxes); else R3_CELL(7, 7, false, table); }
    else { if (from_boxes) R3_CELL(6, 6, true, boxes); else R3_CELL(6, 6, false, table); }
#undef R3_CELL
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (plane && points == 5 && g_r3_fr_impl == 0 && (size_t)(H + 3) * (W + 1) * 8 <= 156 * 1024 && H * W <= 16384) {
    // points = 5: the sample points in registers, whole planes in LDS, Q workgroups per plane above 4096 positions
    const int HWp = H * W;
    const int Q = (HWp + 4095) / 4096;
    const int need = (HWp + Q * 1024 - 1) / (Q * 1024);  // positions per thread
    const int KQ = need <= 1 ? 1 : need <= 2 ? 2 : 4;
    int Gc = 1;
    while (Gc * 2 <= 16 && C % (Gc * 2) == 0 && (size_t)N * (C / (Gc * 2)) * Q >= (size_t)cu_count()) Gc *= 2;
    const size_t ldsb = (size_t)(H + 3) * (W + 1) * 8;  // two buffers
    static R3DeviceOnce once;
    if (once.first()) {
      allow_big_lds(fr_forward_points_kernel<5, 1>, 160 * 1024);
      allow_big_lds(fr_forward_points_kernel<5, 2>, 160 * 1024);
      allow_big_lds(fr_forward_points_kernel<5, 4>, 160 * 1024);
    }
    const dim3 pgrid((unsigned)(N * (C / Gc) * Q));
#define R3_P5(KK) hipLaunchKernelGGL((fr_forward_points_kernel<5, KK>), pgrid, dim3(1024), ldsb, stream, feat, boxes, C, H, W, scale, Gc, Q, out)
    if (KQ == 1) R3_P5(1); else if (KQ == 2) R3_P5(2); else R3_P5(4);
#undef R3_P5
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (plane) {
    // spread small levels over more workgroups: cap planes per workgroup so that the grid
    // has >= 512 workgroups when possible
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    static R3DeviceOnce once;
    if (once.first()) {
      allow_big_lds(fr_forward_plane<1, true, 1>, FRP_LDS_FLOATS * 4);
      allow_big_lds(fr_forward_plane<1, true, 2>, FRP_LDS_FLOATS * 4);
      allow_big_lds(fr_forward_plane<1, true, 0>, FRP_LDS_FLOATS * 4);
      allow_big_lds(fr_forward_plane<1, false, 0>, FRP_LDS_FLOATS * 4);
      allow_big_lds(fr_forward_plane<5, false, 0>, FRP_LDS_FLOATS * 4);
    }
    const bool vec = (W % 4 == 0) && aligned16(feat) && aligned16(boxes) && aligned16(out);
    const bool full = (C % cpb) == 0;  // every workgroup owns exactly cpb planes
#define R3_FWD(P, V, K) hipLaunchKernelGGL((fr_forward_plane<P, V, K>), grid, dim3(FRP_BLOCK), lds, stream, feat, boxes, C, H, W, scale, cpb, out)
    // points = 5 keeps 5 taps per position live: the quad form would spill, use the scalar form
    if (points == 1 && vec) {
      if (full && cpb == 1) R3_FWD(1, true, 1);
      else if (full && cpb == 2) R3_FWD(1, true, 2);
      else R3_FWD(1, true, 0);
    } else if (points == 1) {
      R3_FWD(1, false, 0);
    } else {
      R3_FWD(5, false, 0);
    }
#undef R3_FWD
  } else {
    int HW = H * W;
    int xb = (HW + FR_BLOCK - 1) / FR_BLOCK;
    // enough channel slices to fill the chip (>= ~2048 workgroups) but >= 8 channels each
    int slices = 1;
    while ((size_t)xb * N * slices < 2048 && (C / (slices * 2)) >= 8) slices *= 2;
    int cpbk = (C + slices - 1) / slices;
    dim3 grid(xb, (C + cpbk - 1) / cpbk, N);
    if (points == 1)
      hipLaunchKernelGGL(fr_forward_generic<1>, grid, dim3(FR_BLOCK), 0, stream, feat, boxes, C, H, W, scale, cpbk, out);
    else
      hipLaunchKernelGGL(fr_forward_generic<5>, grid, dim3(FR_BLOCK), 0, stream, feat, boxes, C, H, W, scale, cpbk, out);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// All levels of an NCHW pass: the levels that take the float4 plane kernel (the coarse levels of a pyramid) are ONE
// grid, every other level (the cell kernel's 128 x 128 / 64 x 64 planes, points = 5, odd shapes) its own launch as
// r3k_fr_forward would make it.  Pointer arrays are HOST arrays; ws (may be null) is carved level by level into
// r3k_fr_workspace_bytes parts.
int r3k_fr_forward_levels(int levels, const float* const* feat, const float* const* boxes, int N, int C, const int* H,
                          const int* W, const float* scales, int points, float* const* out, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
  if (levels < 0 || (levels > 0 && (!feat || !boxes || !H || !W || !scales || !out))) return -1;
  FrPlaneLevels A;
  A.n = 0;
  int blocks = 0;
  size_t lds = 0;
  bool grouped[FRPL_MAX] = {};
  for (int l = 0; l < levels && levels <= FRPL_MAX; l++) {
    if (points != 1 || g_r3_fr_impl != 0 || g_r3_frb_impl == 6 || r3_fr_dbg() != 0 || N <= 0 || C <= 0 || H[l] <= 0 || W[l] <= 0) break;
    // (64 x 64 as planes inside the grid instead of the cell kernel in front of it, measured: 48.7-49.4 against 46.0 us
    // for the five levels at N = 4, 45 against 30.7 at N = 2)
    const bool cell_shape = (W[l] == 128 && H[l] == 128) || (W[l] == 64 && H[l] == 64);
    int cpb = plane_cpb(C, H[l], W[l]);
    const bool vec = (W[l] % 4 == 0) && feat[l] && boxes[l] && out[l] && aligned16(feat[l]) && aligned16(boxes[l]) &&
                     aligned16(out[l]);
    if (cell_shape || cpb <= 0 || !vec || (FRPL_BLOCK / FRPL_T) * H[l] * (W[l] + 1) > FRP_LDS_FLOATS) continue;
    FrPlaneLevel& L = A.l[A.n++];
    L.feat = feat[l], L.boxes = boxes[l], L.out = out[l], L.scale = scales[l], L.H = H[l], L.W = W[l], L.cpb = cpb;
    L.feat2 = L.res = nullptr;
    grouped[l] = true;
  }
  // a level that takes the cell kernel with a tap table (128 x 128): its table rides in the grid
  int tl = -1;
  char* tws = nullptr;
  if (A.n >= 1 && !g_r3_fr_profile) {
    char* q = static_cast<char*>(ws);
    size_t left = ws_bytes;
    for (int l = 0; l < levels && q; l++) {
      const size_t part = r3k_fr_workspace_bytes(N, H[l], W[l], points);
      int G = 1;
      while (G * 2 <= 16 && C % (G * 2) == 0 && (size_t)N * C / (G * 2) >= (size_t)cu_count()) G *= 2;
      if (tl < 0 && W[l] == 128 && H[l] == 128 && G >= 2 && left >= part && aligned16(q) && feat[l] && boxes[l] && out[l] &&
          aligned16(feat[l]) && aligned16(out[l]) && (long long)N * H[l] * W[l] < (1LL << 30)) {
        tl = l;
        tws = q;
      }
      q += part;
      left = left >= part ? left - part : 0;
    }
  }
  if (A.n + (tl >= 0 ? 1 : 0) < 2) {  // nothing to group
    A.n = 0;
    tl = -1;
    for (int l = 0; l < FRPL_MAX; l++) grouped[l] = false;
  }
  for (int i = 0; i < A.n; i++) {
    FrPlaneLevel& L = A.l[i];
    int cpb = L.cpb;  // (what the LDS holds)
    constexpr int U = FRPL_BLOCK / FRPL_T;  // (image, channel group) units per workgroup
    while (cpb > 1 && ((size_t)N * ((C + cpb - 1) / cpb) < 512 || C % cpb || U * cpb * L.H * (L.W + 1) > FRP_LDS_FLOATS))
      cpb = (cpb + 1) / 2;  // (spread as r3k_fr_forward spreads them)
    L.cpb = cpb, L.gx = C / cpb, L.first = blocks, L.N = N;
    blocks += (L.gx * N + U - 1) / U;
    lds = std::max(lds, (size_t)U * cpb * L.H * (L.W + 1) * sizeof(float));
  }
  A.tfirst = -1, A.tboxes = nullptr, A.table = nullptr, A.tscale = 0.f, A.tN = A.tH = A.tW = 0;
  A.t2first = -1, A.t2boxes = nullptr, A.table2 = nullptr, A.t2scale = 0.f, A.t2H = A.t2W = 0;
  if (tl >= 0) {
    A.tfirst = blocks, A.tboxes = boxes[tl], A.table = reinterpret_cast<float*>(tws), A.tscale = scales[tl];
    A.tN = N, A.tH = H[tl], A.tW = W[tl];
    blocks += (N * H[tl] * W[tl] + FRPL_BLOCK - 1) / FRPL_BLOCK;
  }
  if (A.n) {  // (first: the cell kernel below reads the table)
    for (int i = A.n; i < FRPL_MAX; i++) A.l[i] = A.l[A.n - 1];
    static R3DeviceOnce once;
    if (once.first()) allow_big_lds(fr_forward_plane_levels<false>, FRP_LDS_FLOATS * 4);
    hipLaunchKernelGGL(fr_forward_plane_levels<false>, dim3(blocks), dim3(FRPL_BLOCK), lds, stream, A, C);
    if (hipGetLastError() != hipSuccess) return -2;
  }
  char* p = static_cast<char*>(ws);
  for (int l = 0; l < levels; l++) {
    const size_t part = r3k_fr_workspace_bytes(N, H[l], W[l], points);
    if (l == tl) {
      const int k = r3k_fr_forward_prepared(feat[l], nullptr, nullptr, A.table, N, C, H[l], W[l], out[l], stream);
      if (k) return k;
    } else if (!(l < FRPL_MAX && grouped[l])) {
      const int k = r3k_fr_forward(feat[l], boxes[l], N, C, H[l], W[l], scales[l], points, out[l], p,
                                   p && ws_bytes >= part ? part : 0, stream);
      if (k) return k;
    }
    if (p) {
      p += part;
      ws_bytes = ws_bytes >= part ? ws_bytes - part : 0;
    }
  }
  return 0;
}

// Split form of the cell path: the tap table of a level is built ahead of time (for instance for all
// pyramid levels before the module's convolutions run), the sampler launch is then the cell kernel
// alone -- no dependent launch in front of it.
size_t r3k_fr_table_bytes(int N, int H, int W) {
  const bool cell_shape = (W == 128 && H == 128) || (W == 64 && H == 64);
  return (cell_shape && N > 0) ? (size_t)N * H * W * 2 * sizeof(float) : 0;
}

int r3k_fr_prepare(const float* boxes, int N, int H, int W, float scale, float* table, hipStream_t stream) {
  if (!r3k_fr_table_bytes(N, H, W) || !boxes || !table || !aligned16(table)) return -1;
  const int total = N * H * W;
  hipLaunchKernelGGL(fr_cell_table_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, boxes, N, H, W, scale, table);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// feat2 / res both null: out = feat + sample(feat) (the sampler alone); both given: the module's fused form
// out = res + ((feat + feat2) + sample(feat + feat2)); res alone: out = res + (feat + sample(feat))
int r3k_fr_forward_prepared(const float* feat, const float* feat2, const float* res, const float* table, int N, int C,
                            int H, int W, float* out, hipStream_t stream) {
  if (!r3k_fr_table_bytes(N, H, W) || !feat || !table || !out || C <= 0 || (feat2 && !res)) return -1;
  int G = 1;
  while (G * 2 <= 16 && C % (G * 2) == 0 && (size_t)N * C / (G * 2) >= (size_t)cu_count()) G *= 2;
  if (G < 2 || !aligned16(feat) || !aligned16(out) || !aligned16(table)) return -1;
  if ((feat2 && !aligned16(feat2)) || (res && !aligned16(res))) return -1;
  static R3DeviceOnce once;
    if (once.first()) {
      allow_big_lds(fr_forward_cell<7, 7, 1024, false>, 160 * 1024);
      allow_big_lds(fr_forward_cell<7, 7, 1024, false, 1>, 160 * 1024);
      allow_big_lds(fr_forward_cell<7, 7, 1024, false, 2>, 160 * 1024);
    }
  const size_t lds = (size_t)2 * ((((size_t)H + 3) * (W + 1) + 3) & ~(size_t)3) * sizeof(float);
  FrProfileSlot* ps = g_r3_fr_profile ? fr_profile_next(N, H) : nullptr;
  if (ps) ps->mode = 3;
  hipEvent_t e0 = ps ? ps->ev[0] : nullptr, e1 = ps ? ps->ev[3] : nullptr;
  const dim3 grid(N * C / G), block(1024);
#define R3_PREP(LW, FU) \
  hipExtLaunchKernelGGL((fr_forward_cell<LW, LW, 1024, false, FU>), grid, block, lds, stream, e0, e1, 0, feat, table, C, G, \
                        0.f, out, feat2, res)
  if (W == 128) { if (feat2) R3_PREP(7, 1); else if (res) R3_PREP(7, 2); else R3_PREP(7, 0); }
  else { if (feat2) R3_PREP(6, 1); else if (res) R3_PREP(6, 2); else R3_PREP(6, 0); }
#undef R3_PREP
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The module tail of ALL levels of an NCHW pass, out_l = res_l + (P_l + sample(P_l)), P_l = a_l + b_l
// (feature_refine_module.py:108-127: the loop over the levels with its two elementwise passes per level), points = 1:
// ONE grid for the coarse levels (the float4 plane kernel with the two adds folded in) that also carries the tap tables
// of the cell levels (128 x 128 / 64 x 64), then one fused cell launch per cell level: 3 launches for a 1024^2 pyramid
// where the per-level form took 2 table + 2 cell launches and 3 x (add, sampler, add).  tables[l]: r3k_fr_table_bytes
// of storage for a cell level (left holding the level's tap table), ignored elsewhere.  -1: a level takes neither form
// (nothing was launched; the caller runs level by level).
int r3k_fr_module_levels(int levels, const float* const* a, const float* const* b, const float* const* res,
                         const float* const* boxes, int N, int C, const int* H, const int* W, const float* scales,
                         float* const* out, float* const* tables, hipStream_t stream) {
  if (levels < 1 || levels > FRPL_MAX || !a || !b || !res || !boxes || !H || !W || !scales || !out || !tables || N <= 0 ||
      C <= 0 || g_r3_fr_impl != 0 || r3_fr_dbg() != 0)
    return -1;
  FrPlaneLevels A;
  A.n = 0;
  int cell[2] = {-1, -1}, ncell = 0;
  int G = 1;
  while (G * 2 <= 16 && C % (G * 2) == 0 && (size_t)N * C / (G * 2) >= (size_t)cu_count()) G *= 2;
  for (int l = 0; l < levels; l++) {
    if (!a[l] || !b[l] || !res[l] || !boxes[l] || !out[l] || H[l] <= 0 || W[l] <= 0 || !aligned16(a[l]) || !aligned16(b[l]) ||
        !aligned16(res[l]) || !aligned16(out[l]) || !aligned16(boxes[l]))
      return -1;
    if (r3k_fr_table_bytes(N, H[l], W[l])) {
      if (G < 2 || ncell == 2 || !tables[l] || !aligned16(tables[l]) || (long long)N * H[l] * W[l] >= (1LL << 30)) return -1;
      cell[ncell++] = l;
      continue;
    }
    const int cpb = plane_cpb(C, H[l], W[l]);
    if (cpb <= 0 || (W[l] & 3) || (FRPL_BLOCK / FRPL_T) * H[l] * (W[l] + 1) > FRP_LDS_FLOATS) return -1;
    FrPlaneLevel& L = A.l[A.n++];
    L.feat = a[l], L.feat2 = b[l], L.res = res[l], L.boxes = boxes[l], L.out = out[l], L.scale = scales[l], L.H = H[l],
    L.W = W[l], L.cpb = cpb;
  }
  int blocks = 0;
  size_t lds = 0;
  for (int i = 0; i < A.n; i++) {
    FrPlaneLevel& L = A.l[i];
    int cpb = L.cpb;
    constexpr int U = FRPL_BLOCK / FRPL_T;
    while (cpb > 1 && ((size_t)N * ((C + cpb - 1) / cpb) < 512 || C % cpb || U * cpb * L.H * (L.W + 1) > FRP_LDS_FLOATS))
      cpb = (cpb + 1) / 2;
    if (C % cpb) return -1;
    L.cpb = cpb, L.gx = C / cpb, L.first = blocks, L.N = N;
    blocks += (L.gx * N + U - 1) / U;
    lds = std::max(lds, (size_t)U * cpb * L.H * (L.W + 1) * sizeof(float));
  }
  A.tfirst = A.t2first = -1, A.tboxes = A.t2boxes = nullptr, A.table = A.table2 = nullptr, A.tscale = A.t2scale = 0.f;
  A.tN = N, A.tH = A.tW = A.t2H = A.t2W = 0;
  if (ncell >= 1) {
    const int l = cell[0];
    A.tfirst = blocks, A.tboxes = boxes[l], A.table = tables[l], A.tscale = scales[l], A.tH = H[l], A.tW = W[l];
    blocks += (N * H[l] * W[l] + FRPL_BLOCK - 1) / FRPL_BLOCK;
  }
  if (ncell == 2) {
    const int l = cell[1];
    A.t2first = blocks, A.t2boxes = boxes[l], A.table2 = tables[l], A.t2scale = scales[l], A.t2H = H[l], A.t2W = W[l];
    blocks += (N * H[l] * W[l] + FRPL_BLOCK - 1) / FRPL_BLOCK;
  }
  if (blocks) {
    if (A.n == 0) {  // (tables only: the struct's level slots still have to be readable)
      FrPlaneLevel& L = A.l[0];
      L.feat = L.feat2 = L.res = L.boxes = nullptr, L.out = nullptr, L.scale = 0.f, L.H = L.W = L.cpb = L.gx = 1, L.N = 0;
      L.first = 0;
    }
    for (int i = std::max(A.n, 1); i < FRPL_MAX; i++) A.l[i] = A.l[std::max(A.n, 1) - 1];
    static R3DeviceOnce once;
    if (once.first()) allow_big_lds(fr_forward_plane_levels<true>, FRP_LDS_FLOATS * 4);
    hipLaunchKernelGGL(fr_forward_plane_levels<true>, dim3(blocks), dim3(FRPL_BLOCK), lds, stream, A, C);
    if (hipGetLastError() != hipSuccess) return -2;
  }
  for (int i = 0; i < ncell; i++) {
    const int l = cell[i];
    const int k = r3k_fr_forward_prepared(a[l], b[l], res[l], tables[l], N, C, H[l], W[l], out[l], stream);
    if (k) return k == -1 ? -2 : k;  // (the grid already ran: not a "nothing launched" answer)
  }
  return 0;
}

// channels_last sampler: feat / out are (N, H, W, C) contiguous.  b, biases, res null: out = feat + sample(feat)
// (r3det_feature_refine_forward on NHWC memory); otherwise the module tail
// out = res + (P + sample(P)), P = (a + bias_a) + (b + bias_b).  C % 4 == 0 and 16-byte aligned pointers.
int r3k_fr_forward_nhwc(const float* a, const float* b, const float* bias_a, const float* bias_b, const float* res,
                        const float* boxes, int N, int C, int H, int W, float scale, int points, float* out,
                        hipStream_t stream, float* tab) {
  if (!a || !boxes || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (points != 1 && points != 5)) return -1;
  if (tab && (points != 1 || !aligned16(tab))) return -1;  // (the table is the points = 1 sample point)
  if ((C & 3) || !aligned16(a) || !aligned16(out) || (b && !aligned16(b)) || (res && !aligned16(res)) ||
      (bias_a && !aligned16(bias_a)) || (bias_b && !aligned16(bias_b)))
    return -1;
  const bool fused = b || bias_a || bias_b || res;
  // points = 1: the pipelined kernel on 4 x 4 tiles (square maps: transposed tile pairs per workgroup; fr_dbg 1
  // switches the pairing off for A/B runs); points = 5: the simple kernel
  const int dbg = r3_fr_dbg();  // (read once per call)
  const bool occ = points == 1 && dbg != 2;  // fr_dbg 2 (probes builds): the register-pipelined kernel
  if (tab && !occ) return -1;                // (only the shipped kernels write the table)
  const int kw = 4, kh = NH_ROWS;
  const int tiles_x = (W + kw - 1) / kw, tiles_y = (H + kh - 1) / kh;
  const bool paired = points == 1 && tiles_x == tiles_y && dbg != 1;
  const int tpi = !paired ? tiles_x * tiles_y
                  : occ ? tiles_x * (tiles_x - 1) / 2 + (tiles_x + 1) / 2 : tiles_x * (tiles_x + 1) / 2;
  const long long T = (long long)tpi * N;
  if (T > 0x7fffffffLL) return -1;
  FrProfileSlot* ps = (g_r3_fr_profile && points == 1) ? fr_profile_next(N, H) : nullptr;
  if (ps) ps->mode = 3;
  hipEvent_t e0 = ps ? ps->ev[0] : nullptr, e1 = ps ? ps->ev[3] : nullptr;
  // square maps with a side that is a multiple of 8 take the wide form (fr_dbg 9 and the other A/B switches keep the
  // 4 x 4 tile pairs)
  // ... when that makes at least 512 workgroups (two per CU): 32 x 32 maps at N = 4 are 64 wide workgroups against
  // 144 pairs, 16.0 against 11.7 us inside the model; fr_dbg 8 forces the wide form (tests)
  if (occ && paired && H == W && (H & 7) == 0 &&
      ((dbg == 0 && (long long)(H / 8) * (H / 8) * N >= 512) || dbg == 8) &&
      (unsigned long long)H * W * C * 4ull < (1ull << 32)) {
    const int S = H / 8;
    const long long Tw = (long long)S * S * N;
    const dim3 gw((unsigned)Tw), bw(1024);
    const int s_strip = S | (g_r3_fr_walk << 20);
#define R3_WIDE(F, T2, RS) \
  do { \
    if (tab) \
      hipExtLaunchKernelGGL((fr_forward_nhwc_wide<F, T2, RS, true>), gw, bw, 0, stream, e0, e1, 0, a, b, bias_a, bias_b, res, \
                            boxes, C, H, W, scale, s_strip, S * S, (int)Tw, out, tab); \
    else \
      hipExtLaunchKernelGGL((fr_forward_nhwc_wide<F, T2, RS>), gw, bw, 0, stream, e0, e1, 0, a, b, bias_a, bias_b, res, boxes, \
                            C, H, W, scale, s_strip, S * S, (int)Tw, out, (float*)nullptr); \
  } while (0)
    if (!fused) R3_WIDE(false, false, false);
    else if (b && res) R3_WIDE(true, true, true);
    else if (b) R3_WIDE(true, true, false);
    else if (res) R3_WIDE(true, false, true);
    else R3_WIDE(true, false, false);
#undef R3_WIDE
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  const dim3 grid((unsigned)T), block(paired ? 512 : 256);
#define R3_ARGS a, b, bias_a, bias_b, res, boxes, C, H, W, scale, tiles_x, tpi, (int)T, out
  if (occ) {
    // shipped: VAR 14 = VAR 6 + non-temporal loads of the tiles' interior identity rows (fr_dbg 6: VAR 6 alone);
    // VAR 6 = 32-bit index arithmetic + non-temporal residual loads / output stores (each touched exactly
    // once: they no longer evict the a / b rows the neighbouring workgroups' halo taps are about to ask for).  Level 0,
    // N = 4, rotating buffers: 66.1 us (round-2 form, fr_dbg 3) -> 60.6 us, FETCH x 2 + WRITE 337 -> 317 MB
    // (tools/fr_nhwc_ab.py, gpurun_out/fr_fwd_pmc_*.txt); prefetching the boxes in phase 1 (VAR 1, fr_dbg 4): 67.6 us.
    // Measured and not kept (round 3, same buffers): touching the out-of-tile tap rows in phase 1 (one lane per
    // 128-byte line) 59.6 -> 63.4 us with FETCH 123.2 -> 129.4 K: the rows are fetched again anyway; requesting the
    // residual rows in phase 1 (16 more VGPRs) 60.6 us.  The pair walk in strips (r3_fr_tap.h) 60.9 -> 59.4 us at
    // unchanged FETCH.  What is left above the algorithmic bytes (+45 MB of 207 MB read) is exactly the out-of-tile
    // tap rows of about half the tile edges (fields without such taps: FETCH = 204 MB), whatever the launch order.
    const bool big = (unsigned long long)H * W * C * 4ull >= (1ull << 32);  // (32-bit byte offsets inside an image)
    const int var = (dbg == 3 || big) ? 0 : dbg == 4 ? 1 : dbg == 5 ? 2 : dbg == 6 ? 6 : 14;
#undef R3_ARGS
#define R3_ARGS a, b, bias_a, bias_b, res, boxes, C, H, W, scale, tiles_x | (g_r3_fr_walk << 20), tpi, (int)T, out, tab
#ifdef R3_PROBES  // (the variants measured on the way, tools/fr_nhwc_ab.py / fr_fwd_var_ab.py)
#define R3_OCC(F, P) \
  do { \
    if (var == 0) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 0>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else if (var == 1) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 1>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else if (var == 2) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 2>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else if (var == 6) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 6>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 14>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
  } while (0)
#else  // (the shipped form; images of 4 GB and more: the form with 64-bit offsets)
#define R3_OCC(F, P) \
  do { \
    if (var == 0) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 0>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 14>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
  } while (0)
#endif
    if (tab) {  // (training: the same kernels, also writing the level's tap table)
#define R3_OCCT(F, P) \
  do { \
    if (var == 0) hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 0, true>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
    else hipExtLaunchKernelGGL((fr_forward_nhwc_occ<F, P, 14, true>), grid, block, 0, stream, e0, e1, 0, R3_ARGS); \
  } while (0)
      if (paired) {
        if (fused) R3_OCCT(true, true); else R3_OCCT(false, true);
      } else {
        if (fused) R3_OCCT(true, false); else R3_OCCT(false, false);
      }
#undef R3_OCCT
    } else
    if (paired) {
      if (fused) R3_OCC(true, true); else R3_OCC(false, true);
    } else {
      if (fused) R3_OCC(true, false); else R3_OCC(false, false);
    }
#undef R3_OCC
#undef R3_ARGS
#define R3_ARGS a, b, bias_a, bias_b, res, boxes, C, H, W, scale, tiles_x, tpi, (int)T, out
#ifdef R3_PROBES
  } else if (points == 1) {
    if (paired) {
      if (fused) hipExtLaunchKernelGGL((fr_forward_nhwc_pipe<true, 4, true>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
      else hipExtLaunchKernelGGL((fr_forward_nhwc_pipe<false, 4, true>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
    } else {
      if (fused) hipExtLaunchKernelGGL((fr_forward_nhwc_pipe<true, 4, false>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
      else hipExtLaunchKernelGGL((fr_forward_nhwc_pipe<false, 4, false>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
    }
#endif
  } else if (!fused && dbg == 0 && (H & 7) == 0 && (W & 7) == 0 && (C & 63) == 0 && H < 32768 && W < 32768 &&
             (long long)(H / P5_T) * (W / P5_T) * N <= 0x7fffffffLL && (long long)(H / P5_T) * (W / P5_T) * N >= r3_cu_count()) {
    // points = 5, plain sampler: 8 x 8 tiles of positions with their sample region in LDS -- from one workgroup per
    // compute unit up (level 0, N = 4: 104 -> 77 us, level 1: 29.6 -> 23.3; the coarse levels, 16-64 workgroups of 4
    // channel passes each: 16-19 us against the simple kernel's 12: tools/fr_p5_nhwc_ab.py)
    const int ptx = W / P5_T, ptpi = ptx * (H / P5_T), pT = ptpi * N;
    hipLaunchKernelGGL(fr_forward_nhwc_p5, dim3((unsigned)pT), dim3(1024), 0, stream, a, boxes, C, H, W, scale, ptx, ptpi, pT,
                       out);
  } else {
    if (fused) hipExtLaunchKernelGGL((fr_forward_nhwc<5, true, 4>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
    else hipExtLaunchKernelGGL((fr_forward_nhwc<5, false, 4>), grid, block, 0, stream, e0, e1, 0, R3_ARGS);
  }
#undef R3_ARGS
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// All levels of a channels_last pass in as few launches as the kernels allow: a level that takes the wide regions
// form (level 0 of a 1024^2 input) is launched alone, the others -- 4 x 4 tile pairs -- as ONE grid.  Pointer arrays
// are HOST arrays; b / res may be null arrays or hold nulls as in r3k_fr_forward_nhwc (uniform over the levels).
int r3k_fr_forward_nhwc_levels(int levels, const float* const* a, const float* const* b, const float* bias_a,
                               const float* bias_b, const float* const* res, const float* const* boxes, int N, int C,
                               const int* H, const int* W, const float* scales, int points, float* const* out,
                               hipStream_t stream, float* const* tabs) {
  if (levels < 0 || !a || !boxes || !out || !H || !W || !scales) return -1;
  if (tabs && points != 1) return -1;
  FrNhwcLevels A;
  A.n = 0;
  int blocks = 0;
  bool grouped[FRL_MAX] = {};
  const bool has_b = b && levels > 0 && b[0], has_res = res && levels > 0 && res[0];
  const bool fused = has_b || bias_a || bias_b || has_res;
  for (int l = 0; l < levels && levels <= FRL_MAX; l++) {
    const int tiles_x = (W[l] + 3) / 4, tiles_y = (H[l] + NH_ROWS - 1) / NH_ROWS;
    const bool wide = H[l] == W[l] && (H[l] & 7) == 0 && (long long)(H[l] / 8) * (H[l] / 8) * N >= 512;
    const bool big = (unsigned long long)H[l] * W[l] * C * 4ull >= (1ull << 32);
    const bool ok = points == 1 && r3_fr_dbg() == 0 && tiles_x == tiles_y && !wide && !big && N > 0 && C > 0 && !(C & 3) &&
                    a[l] && boxes[l] && out[l] && aligned16(a[l]) && aligned16(out[l]) &&
                    (!has_b || (b[l] && aligned16(b[l]))) && (!has_res || (res[l] && aligned16(res[l]))) &&
                    (!bias_a || aligned16(bias_a)) && (!bias_b || aligned16(bias_b)) &&
                    (!tabs || !tabs[l] || aligned16(tabs[l]));
    if (!ok) continue;
    const int tpi = tiles_x * (tiles_x - 1) / 2 + (tiles_x + 1) / 2;
    FrNhwcLevel& L = A.l[A.n++];
    L.a = a[l], L.b = has_b ? b[l] : nullptr, L.res = has_res ? res[l] : nullptr, L.boxes = boxes[l], L.out = out[l];
    L.tab = tabs ? tabs[l] : nullptr;
    L.scale = scales[l], L.H = H[l], L.W = W[l], L.tiles_xs = tiles_x | (g_r3_fr_walk << 20), L.tiles_per_img = tpi;
    L.T = tpi * N, L.first = blocks;
    blocks += L.T;
    grouped[l] = true;
  }
  if (A.n < 2) {  // nothing to group
    A.n = 0;
    for (int l = 0; l < levels && l < FRL_MAX; l++) grouped[l] = false;
  }
  for (int l = 0; l < levels; l++) {
    if (l < FRL_MAX && grouped[l]) continue;
    const int k = r3k_fr_forward_nhwc(a[l], has_b ? b[l] : nullptr, bias_a, bias_b, has_res ? res[l] : nullptr, boxes[l], N, C,
                                      H[l], W[l], scales[l], points, out[l], stream, tabs ? tabs[l] : nullptr);
    if (k) return k;
  }
  if (A.n) {
    for (int i = A.n; i < FRL_MAX; i++) A.l[i] = A.l[A.n - 1];
    if (tabs) {
      if (fused)
        hipLaunchKernelGGL((fr_forward_nhwc_occ_levels<true, true>), dim3(blocks), dim3(512), 0, stream, A, bias_a, bias_b, C);
      else
        hipLaunchKernelGGL((fr_forward_nhwc_occ_levels<false, true>), dim3(blocks), dim3(512), 0, stream, A, bias_a, bias_b, C);
    } else if (fused)
      hipLaunchKernelGGL(fr_forward_nhwc_occ_levels<true>, dim3(blocks), dim3(512), 0, stream, A, bias_a, bias_b, C);
    else
      hipLaunchKernelGGL(fr_forward_nhwc_occ_levels<false>, dim3(blocks), dim3(512), 0, stream, A, bias_a, bias_b, C);
    if (hipGetLastError() != hipSuccess) return -2;
  }
  return 0;
}

// The NCHW backward is a gather over the inverse tap index of the boxes (r3_frb.hip: CSR + SELL-64, whole
// gradient planes staged in LDS, no atomics, one summation order); shapes it does not take (planes beyond the LDS,
// no workspace) fall back to the scatter kernels above.
size_t r3k_fr_backward_workspace_bytes(int N, int H, int W, int points) {
  if (points != 1 && points != 5) return 0;
  return r3k_frn_workspace_bytes(N, H, W, points);
}

// index_ready: ws already holds r3k_frn_index of these boxes; the call then fails with -1 instead of taking a
// path that would ignore it
int r3k_fr_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W,
                    float scale, int points, float* bottom_grad, int overwrite, void* ws, size_t ws_bytes,
                    int index_ready, hipStream_t stream) {
  if (points != 1 && points != 5) return -1;
  if (N == 0 || C == 0 || H == 0 || W == 0) return 0;
  const size_t need = r3k_fr_backward_workspace_bytes(N, H, W, points);
  if (g_r3_fr_impl == 0 && need && ws && ws_bytes >= need && aligned16(ws)) {
    if (!index_ready) {
      const int rc = r3k_frn_index(boxes, N, C, H, W, scale, points, ws, ws_bytes, stream);
      if (rc) return rc;
    }
    const int rc = r3k_frn_gather(top_grad, N, C, H, W, points, bottom_grad, overwrite, ws, ws_bytes, stream);
    if (rc != -1 || index_ready) return rc;
  } else if (index_ready) {
    return -1;
  }
  int cpb = plane_cpb(C, H, W);
  const bool plane = g_r3_fr_impl != 1 && cpb > 0;
  if (plane) {
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    allow_big_lds(fr_backward_plane<1, true>, FRP_LDS_FLOATS * 4);
    allow_big_lds(fr_backward_plane<1, false>, FRP_LDS_FLOATS * 4);
    allow_big_lds(fr_backward_plane<5, false>, FRP_LDS_FLOATS * 4);
    const bool vec = (W % 4 == 0) && aligned16(top_grad) && aligned16(boxes) && aligned16(bottom_grad);
#define R3_BWD(P, V) hipLaunchKernelGGL((fr_backward_plane<P, V>), grid, dim3(FRP_BLOCK), lds, stream, top_grad, boxes, C, H, W, scale, cpb, overwrite, bottom_grad)
    if (points == 1) { if (vec) R3_BWD(1, true); else R3_BWD(1, false); }
    else R3_BWD(5, false);
#undef R3_BWD
  } else {
    if (overwrite) {
      if (r3k_zero_async(bottom_grad, (size_t)N * C * H * W * sizeof(float), stream) != 0) return -2;
    }
    int HW = H * W;
    int xb = (HW + FR_BLOCK - 1) / FR_BLOCK;
    int slices = 1;
    while ((size_t)xb * N * slices < 2048 && (C / (slices * 2)) >= 8) slices *= 2;
    int cpbk = (C + slices - 1) / slices;
    dim3 grid(xb, (C + cpbk - 1) / cpbk, N);
    if (points == 1)
      hipLaunchKernelGGL(fr_backward_generic<1>, grid, dim3(FR_BLOCK), 0, stream, top_grad, boxes, C, H, W, scale, cpbk, bottom_grad);
    else
      hipLaunchKernelGGL(fr_backward_generic<5>, grid, dim3(FR_BLOCK), 0, stream, top_grad, boxes, C, H, W, scale, cpbk, bottom_grad);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Drains the profiling ring: waits for the recorded launches, writes up to `capacity` records
// {N, H, table_us, cell_us, span_us} (5 floats each) and returns their number.
int r3k_fr_profile_read(float* records, int capacity) {
  int n = 0;
  for (int i = 0; i < g_fr_prof_count; i++) {
    FrProfileSlot& s = g_fr_prof[i];
    if (hipEventSynchronize(s.ev[3]) != hipSuccess) continue;
    float t_ms = 0.f, c_ms = 0.f, all_ms = 0.f;
    if (s.mode == 1) {
      if (hipEventElapsedTime(&t_ms, s.ev[0], s.ev[1]) != hipSuccess) continue;
      if (hipEventElapsedTime(&c_ms, s.ev[2], s.ev[3]) != hipSuccess) continue;
    } else if (s.mode == 3) {  // single kernel: its own duration is the span
      if (hipEventElapsedTime(&c_ms, s.ev[0], s.ev[3]) != hipSuccess) continue;
    }
    if (hipEventElapsedTime(&all_ms, s.ev[0], s.ev[3]) != hipSuccess) continue;
    if (n < capacity && records) {
      records[5 * n + 0] = (float)s.N;
      records[5 * n + 1] = (float)s.H;
      records[5 * n + 2] = t_ms * 1e3f;
      records[5 * n + 3] = c_ms * 1e3f;
      records[5 * n + 4] = all_ms * 1e3f;  // table start -> cell stop: ONE event pair's overhead, gap included
      n++;
    }
  }
  g_fr_prof_count = 0;
  return n;
}
