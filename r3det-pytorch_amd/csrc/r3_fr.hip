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
// Forward implementations (r3det_set_option("fr_impl", k)):
//   1 generic  : taps gathered from global memory (L1/L2); any H x W, no LDS.
//   2 plane    : one workgroup per tile of planes, taps derived in the kernel (no workspace).
//   5 persist  : (default when a workspace is given) one workgroup per CU walks over its tiles
//     6 (consec) with two LDS buffers; taps come from a 20-byte-per-position table built once by
//                fr_taps_kernel.  5 = a lane owns 4 adjacent positions, 6 = lanes own consecutive
//                positions (bank-conflict-free gathers) + in-register 4x4 transposes.
// All of them produce bit-identical outputs (tests/test_gpu_fr.py).
#include <hip/hip_runtime.h>

#include "r3_kernels.h"
#include "r3_trig.h"

int g_r3_fr_impl = 0;
int g_r3_fr_dbg = 0;  // ablation bits for fr_forward_persist<.., true> (tools/fr_ablate.py)

namespace {

struct Tap {
  int o00, o01, o10, o11;  // offsets inside a plane with row pitch `pitch`
  float w1, w2, w3, w4;
  bool valid;
};

// bilinear_interpolate / _gradient coordinate logic (feature_refine_kernel.cu:16-52,67-106)
__device__ __forceinline__ Tap make_tap(int height, int width, int pitch, float y, float x) {
  Tap t;
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    t.valid = false;
    t.o00 = t.o01 = t.o10 = t.o11 = 0;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    return t;
  }
  t.valid = true;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = (float)(1. - (double)ly);
  float hx = (float)(1. - (double)lx);
  t.w1 = hy * hx;
  t.w2 = hy * lx;
  t.w3 = ly * hx;
  t.w4 = ly * lx;
  t.o00 = y_low * pitch + x_low;
  t.o01 = y_low * pitch + x_high;
  t.o10 = y_high * pitch + x_low;
  t.o11 = y_high * pitch + x_high;
  return t;
}

// sample points of one position (feature_refine_kernel.cu:125-151)
template <int POINTS>
__device__ __forceinline__ void make_taps(const float* __restrict__ box, float scale, int H, int W,
                                          int pitch, Tap* taps) {
  float roi_y = box[0] * scale;  // sic: row <- x_ctr
  float roi_x = box[1] * scale;  //      col <- y_ctr
  taps[0] = make_tap(H, W, pitch, roi_y, roi_x);
  if (POINTS > 1) {
    float roi_w = box[2] * scale;
    float roi_h = box[3] * scale;
    float roi_a = box[4];
    float w_2 = roi_w / 2, h_2 = roi_h / 2;
    float sina, cosa;
    r3_sincos(roi_a, sina, cosa);
    float wx = cosa * w_2, wy = sina * w_2;
    float hx = -sina * h_2, hy = cosa * h_2;
    taps[1] = make_tap(H, W, pitch, roi_y + wy + hy, roi_x + wx + hx);
    taps[2] = make_tap(H, W, pitch, roi_y - wy + hy, roi_x - wx + hx);
    taps[3] = make_tap(H, W, pitch, roi_y - wy - hy, roi_x - wx - hx);
    taps[4] = make_tap(H, W, pitch, roi_y + wy - hy, roi_x + wx - hx);
  }
}

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
template <bool VEC>
__device__ __forceinline__ void stage_planes(const float* __restrict__ src, float* lds, int nc, int HW,
                                             int W, int pitch, int psz) {
  const int total = nc * HW;
  if (VEC) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const int total4 = total >> 2;
    for (int base = threadIdx.x; base < total4; base += FRP_BLOCK * FRP_STAGE_UNROLL) {
      float4 v[FRP_STAGE_UNROLL];
#pragma unroll
      for (int k = 0; k < FRP_STAGE_UNROLL; k++) {
        int i = base + k * FRP_BLOCK;
        if (i < total4) v[k] = s4[i];
      }
#pragma unroll
      for (int k = 0; k < FRP_STAGE_UNROLL; k++) {
        int i = base + k * FRP_BLOCK;
        if (i < total4) {
          int e = i << 2;
          int ch = e / HW, r = e - ch * HW;
          int y = r / W, x = r - y * W;
          float* d = lds + ch * psz + y * pitch + x;
          d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < total; e += FRP_BLOCK) {
      int ch = e / HW, r = e - ch * HW;
      int y = r / W, x = r - y * W;
      lds[ch * psz + y * pitch + x] = src[e];
    }
  }
}

// NC > 0: planes per workgroup known at compile time.  Matters for more than unrolling: gfx950
// has ONE in-order vmcnt queue for loads and stores, so "wait for the next quad's boxes" also
// waits for every store issued before them unless the boxes were requested BEFORE those stores
// and the compiler can count the younger stores (s_waitcnt vmcnt(NC) instead of vmcnt(0)).
template <int POINTS, bool VEC, int NC>
__global__ __launch_bounds__(FRP_BLOCK) void fr_forward_plane(const float* __restrict__ feat,
                                                              const float* __restrict__ boxes,
                                                              int C, int H, int W, float scale,
                                                              int cpb, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W;
  const int pitch = W + 1;
  const int psz = H * pitch;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * cpb;
  const int nc = NC > 0 ? NC : min(cpb, C - c0);
  const float* src = feat + ((size_t)n * C + c0) * HW;
  float* dst = out + ((size_t)n * C + c0) * HW;
  if (VEC) {
    const float4* bx4 = reinterpret_cast<const float4*>(boxes + (size_t)n * HW * 5);
    const int quads = HW >> 2;
    float4 nb[5];  // boxes of this thread's first quad: requested before staging
    if ((int)threadIdx.x < quads) {
#pragma unroll
      for (int k = 0; k < 5; k++) nb[k] = bx4[threadIdx.x * 5 + k];
    }
    stage_planes<true>(src, lds, nc, HW, W, pitch, psz);
    __syncthreads();
    for (int qd = threadIdx.x; qd < quads; qd += FRP_BLOCK) {
      float bq[20];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        bq[4 * k] = nb[k].x; bq[4 * k + 1] = nb[k].y; bq[4 * k + 2] = nb[k].z; bq[4 * k + 3] = nb[k].w;
      }
      const int nq = qd + FRP_BLOCK;
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
          *reinterpret_cast<float4*>(dst + (size_t)c2 * HW + hw0) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
  } else {
    stage_planes<false>(src, lds, nc, HW, W, pitch, psz);
    __syncthreads();
    for (int hw = threadIdx.x; hw < HW; hw += FRP_BLOCK) {
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
// tap table (needs a caller workspace): every position is converted ONCE into a 20-byte
// record -- the same bytes a kernel would read for the box -- instead of once per channel
// plane.   record = { packed, w1, w2, w3, w4 }
//          packed = o00 | dx << 20 | dy << 21 | valid << 22   (o00 < H * (W + 1) <= 17 408)
// ----------------------------------------------------------------------------------------
template <int POINTS>
__global__ __launch_bounds__(256) void fr_taps_kernel(const float* __restrict__ boxes, int total, int H,
                                                      int W, float scale, float* __restrict__ table) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= total) return;
  const int pitch = W + 1;
  Tap taps[POINTS];
  make_taps<POINTS>(boxes + (size_t)pos * 5, scale, H, W, pitch, taps);
#pragma unroll
  for (int p = 0; p < POINTS; p++) {
    const Tap& t = taps[p];
    int packed = 0;
    if (t.valid) packed = t.o00 | ((t.o01 - t.o00) << 20) | ((t.o10 != t.o00 ? 1 : 0) << 21) | (1 << 22);
    float* d = table + ((size_t)pos * POINTS + p) * 5;
    d[0] = __int_as_float(packed);
    d[1] = t.w1; d[2] = t.w2; d[3] = t.w3; d[4] = t.w4;
  }
}

__device__ __forceinline__ Tap unpack_tap(const float* d, int pitch) {
  Tap t;
  const int packed = __float_as_int(d[0]);
  t.o00 = packed & 0xFFFFF;
  const int dx = (packed >> 20) & 1, dy = (packed >> 21) & 1;
  t.o01 = t.o00 + dx;
  t.o10 = t.o00 + (dy ? pitch : 0);
  t.o11 = t.o10 + dx;
  t.valid = (packed >> 22) & 1;
  t.w1 = d[1]; t.w2 = d[2]; t.w3 = d[3]; t.w4 = d[4];
  return t;
}

// 4 x 4 transpose inside every group of 4 adjacent lanes (register index <-> lane index),
// two DPP quad_perm exchanges.  In: lane 4q+c holds r[j] = value(j, c).  Out: lane 4q+j holds
// r[c] = value(j, c).
__device__ __forceinline__ float dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose(float (&r)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float s0 = dpp_xor1(b0 ? r[0] : r[1]);
  float s1 = dpp_xor1(b0 ? r[2] : r[3]);
  if (b0) { r[0] = s0; r[2] = s1; } else { r[1] = s0; r[3] = s1; }
  float u0 = dpp_xor2(b1 ? r[0] : r[2]);
  float u1 = dpp_xor2(b1 ? r[1] : r[3]);
  if (b1) { r[0] = u0; r[1] = u1; } else { r[2] = u0; r[3] = u1; }
}

// four positions of one lane: 4 identity reads + 16 tap reads issued together, then the math
__device__ __forceinline__ void sample4(const float* plane, const int (&self)[4], const Tap (&tp)[4],
                                        float (&r)[4], bool gather) {
  float idv[4], lt[4], rt[4], lb[4], rb[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    idv[j] = plane[self[j]];
    lt[j] = plane[tp[j].o00]; rt[j] = plane[tp[j].o01];
    lb[j] = plane[tp[j].o10]; rb[j] = plane[tp[j].o11];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float v = (tp[j].w1 * lt[j] + tp[j].w2 * rt[j] + tp[j].w3 * lb[j] + tp[j].w4 * rb[j]);
    v = (tp[j].valid && gather) ? v : 0.f;
    r[j] = idv[j] + v;
  }
}

// ----------------------------------------------------------------------------------------
// "skew" forward kernel (points = 1; planes of 4096 or 16384 floats, W a power of two): the
// default for the two large pyramid levels.  It removes the three costs the PMC / ablation
// runs found in the plane kernels above (profiles/r01_fr_forward_pmc.txt, tools/fr_ablate.py):
//   * LDS bank conflicts (72 % of LDS-active cycles): a lane owns 4 ADJACENT positions, whose
//     sample rows are 4 apart; with the usual odd pitch 4 l (W+1) = 4 l (mod 32) only hits 8
//     banks.  Here element (r, c) lives at r W + c + (r >> 2): rows 4 l + j of one column map
//     to banks c + l + const -- 32 distinct banks;
//   * the identity term is not re-read from LDS: the thread that staged float4 #i is the thread
//     that samples quad #i, so the four values are still in its registers;
//   * taps come from the 20-byte table (built once per position, not once per channel plane),
//     requested one quad ahead and BEFORE the current store; the loop is fully unrolled so the
//     compiler can wait with vmcnt(1) (one in-order queue for loads and stores on gfx950).
// Table entry for this kernel: packed = x_low | y_low << 10 | dx << 20 | dy << 21 | valid << 22.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_taps_xy_kernel(const float* __restrict__ boxes, int total, int H,
                                                         int W, float scale, float* __restrict__ table) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= total) return;
  Tap t[1];
  make_taps<1>(boxes + (size_t)pos * 5, scale, H, W, 1024, t);  // pitch 1024: o00 = y_low << 10 | x_low
  int packed = 0;
  if (t[0].valid)
    packed = t[0].o00 | ((t[0].o01 - t[0].o00) << 20) | ((t[0].o10 != t[0].o00 ? 1 : 0) << 21) | (1 << 22);
  float* d = table + (size_t)pos * 5;
  d[0] = __int_as_float(packed);
  d[1] = t[0].w1; d[2] = t[0].w2; d[3] = t[0].w3; d[4] = t[0].w4;
}

constexpr int FRS_BLOCK = 512;

template <int F4>  // float4 (= quads) per thread: plane = 4 * F4 * FRS_BLOCK floats
__global__ __launch_bounds__(FRS_BLOCK) void fr_forward_skew(const float* __restrict__ feat,
                                                             const float* __restrict__ table, int C, int logW,
                                                             int logHW, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int W = 1 << logW, HW = 1 << logHW;
  const int tid = threadIdx.x;
  const int plane = blockIdx.x;  // n * C + c
  const int n = plane / C;
  const float4* s4 = reinterpret_cast<const float4*>(feat + ((size_t)plane << logHW));
  const float4* tb4 = reinterpret_cast<const float4*>(table + ((size_t)n << logHW) * 5);
  float4* d4 = reinterpret_cast<float4*>(out + ((size_t)plane << logHW));

  float4 v[F4];
#pragma unroll
  for (int k = 0; k < F4; k++) v[k] = s4[tid + k * FRS_BLOCK];
  float4 tq[5];
#pragma unroll
  for (int k = 0; k < 5; k++) tq[k] = tb4[tid * 5 + k];
#pragma unroll
  for (int k = 0; k < F4; k++) {
    const int e = (tid + k * FRS_BLOCK) << 2;
    const int a = e + ((e >> logW) >> 2);
    lds[a] = v[k].x; lds[a + 1] = v[k].y; lds[a + 2] = v[k].z; lds[a + 3] = v[k].w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < F4; k++) {
    const int qd = tid + k * FRS_BLOCK;
    float td[20];
#pragma unroll
    for (int q = 0; q < 5; q++) {
      td[4 * q] = tq[q].x; td[4 * q + 1] = tq[q].y; td[4 * q + 2] = tq[q].z; td[4 * q + 3] = tq[q].w;
    }
    if (k + 1 < F4) {
#pragma unroll
      for (int q = 0; q < 5; q++) tq[q] = tb4[(qd + FRS_BLOCK) * 5 + q];
    }
    const float idv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
    float lt[4], rt[4], lb[4], rb[4];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int packed = __float_as_int(td[5 * j]);
      const int x = packed & 1023, y = (packed >> 10) & 1023;
      const int dx = (packed >> 20) & 1, dy = (packed >> 21) & 1;
      valid[j] = (packed >> 22) & 1;
      const int y1 = y + dy;
      const int a0 = (y << logW) + x + (y >> 2);
      const int a1 = (y1 << logW) + x + (y1 >> 2);
      lt[j] = lds[a0]; rt[j] = lds[a0 + dx];
      lb[j] = lds[a1]; rb[j] = lds[a1 + dx];
    }
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float val = (td[5 * j + 1] * lt[j] + td[5 * j + 2] * rt[j] + td[5 * j + 3] * lb[j] + td[5 * j + 4] * rb[j]);
      r[j] = idv[j] + (valid[j] ? val : 0.f);
    }
    d4[qd] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ----------------------------------------------------------------------------------------
// "chan" forward kernel (points = 1; planes of 16384 or 4096 floats, power-of-two W): the form
// the diagnosis converged on.  A plain copy THROUGH LDS with the same tiling and barrier runs at
// the device-copy rate (tools/probes/lds_copy_probe.hip: 6.7 TB/s), so staging is free; what
// made every per-plane kernel above land at ~44 us is the per-position sample data (box or tap
// record, 8-20 B) that each of the C channel planes re-reads through the CU's vector-memory
// path: 2.5 x the HBM traffic.  Here a workgroup owns G consecutive channels of one image and
// keeps the taps of ALL positions in registers (packed cell + the two fractions, 3 VGPR per
// position), so they are read once per G planes; planes are double-buffered through LDS in the
// skewed, conflict-free layout of the skew kernel.
// Table for this kernel (fr_taps3_kernel), per quad of 4 positions: 12 floats
//   [packed x4 | ly x4 | lx x4],  packed = x_low | y_low << 10 | dx << 20 | dy << 21 | valid << 22.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_taps3_kernel(const float* __restrict__ boxes, int total, int H,
                                                       int W, float scale, float* __restrict__ table) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= total) return;
  // same coordinate logic as make_tap (feature_refine_kernel.cu:22-52), keeping ly / lx
  float y = boxes[(size_t)pos * 5] * scale;      // sic: row <- x_ctr
  float x = boxes[(size_t)pos * 5 + 1] * scale;  //      col <- y_ctr
  int packed = 0;
  float ly = 0.f, lx = 0.f;
  if (!(y < -1.0 || y > H || x < -1.0 || x > W)) {
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, dy = 1, dx = 1;
    if (y_low >= H - 1) { y_low = H - 1; y = (float)y_low; dy = 0; }
    if (x_low >= W - 1) { x_low = W - 1; x = (float)x_low; dx = 0; }
    ly = y - y_low;
    lx = x - x_low;
    packed = x_low | (y_low << 10) | (dx << 20) | (dy << 21) | (1 << 22);
  }
  float* d = table + (size_t)(pos >> 2) * 12 + (pos & 3);
  d[0] = __int_as_float(packed);
  d[4] = ly;
  d[8] = lx;
}

constexpr int FRC_BLOCK = 1024;

template <int F4>  // quads per thread: plane = 4 * F4 * FRC_BLOCK floats (F4 = 4 -> 128 x 128)
__global__ __launch_bounds__(FRC_BLOCK) void fr_forward_chan(const float* __restrict__ feat,
                                                             const float* __restrict__ table, int C, int G,
                                                             int logW, int logHW, int bufsz,
                                                             float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = 1 << logHW;
  const int tid = threadIdx.x;
  const int groups = C / G;
  const int n = blockIdx.x / groups;
  const int c0 = (blockIdx.x - n * groups) * G;
  const size_t plane0 = (size_t)n * C + c0;
  const float4* tb4 = reinterpret_cast<const float4*>(table + ((size_t)n << logHW) * 3);

  // taps of this thread's F4 quads -> registers (3 per position)
  int pk[F4][4];
  float ly[F4][4], lx[F4][4];
#pragma unroll
  for (int k = 0; k < F4; k++) {
    const int qd = tid + k * FRC_BLOCK;
    const float4 a = tb4[qd * 3], b = tb4[qd * 3 + 1], c = tb4[qd * 3 + 2];
    pk[k][0] = __float_as_int(a.x); pk[k][1] = __float_as_int(a.y);
    pk[k][2] = __float_as_int(a.z); pk[k][3] = __float_as_int(a.w);
    ly[k][0] = b.x; ly[k][1] = b.y; ly[k][2] = b.z; ly[k][3] = b.w;
    lx[k][0] = c.x; lx[k][1] = c.y; lx[k][2] = c.z; lx[k][3] = c.w;
  }
  float4 v[F4];
  auto load_plane = [&](int c) {
    const float4* s4 = reinterpret_cast<const float4*>(feat + ((plane0 + c) << logHW));
#pragma unroll
    for (int k = 0; k < F4; k++) v[k] = s4[tid + k * FRC_BLOCK];
  };
  auto write_plane = [&](float* buf) {
#pragma unroll
    for (int k = 0; k < F4; k++) {
      const int e = (tid + k * FRC_BLOCK) << 2;
      const int a = e + ((e >> logW) >> 2);
      buf[a] = v[k].x; buf[a + 1] = v[k].y; buf[a + 2] = v[k].z; buf[a + 3] = v[k].w;
    }
  };
  load_plane(0);
  write_plane(lds);
  __syncthreads();
  for (int c = 0; c < G; c++) {
    const float* buf = lds + (c & 1) * bufsz;
    if (c + 1 < G) load_plane(c + 1);  // in flight (in v[]) while plane c is sampled out of LDS
    float4* d4 = reinterpret_cast<float4*>(out + ((plane0 + c) << logHW));
#pragma unroll
    for (int k = 0; k < F4; k++) {
      const int e = (tid + k * FRC_BLOCK) << 2;
      const int self = e + ((e >> logW) >> 2);
      float r[4];
      // two positions at a time: 10 LDS reads in flight, modest register footprint (the taps of
      // all F4 quads already hold 12 * F4 VGPRs)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float idv[2], lt[2], rt[2], lb[2], rb[2];
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int j = 2 * h + jj;
          int p = pk[k][j];
          // opaque copy: stops LICM from hoisting the 8 addresses + 4 weights of every position out
          // of the channel loop (16 positions x 12 values do not fit the register file: 356 B/lane
          // of scratch without this)
          asm volatile("" : "+v"(p));
          const int x = p & 1023, y = (p >> 10) & 1023;
          const int dx = (p >> 20) & 1, dy = (p >> 21) & 1;
          const int y1 = y + dy;
          const int a0 = (y << logW) + x + (y >> 2);
          const int a1 = (y1 << logW) + x + (y1 >> 2);
          idv[jj] = buf[self + j];
          lt[jj] = buf[a0]; rt[jj] = buf[a0 + dx];
          lb[jj] = buf[a1]; rb[jj] = buf[a1 + dx];
        }
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int j = 2 * h + jj;
          float fy = ly[k][j], fx = lx[k][j];
          asm volatile("" : "+v"(fy), "+v"(fx));
          // 1.f - f == (float)(1. - (double)f) for every f in [0, 1): the double difference is
          // exact when f >= 2^-29 and both forms round to 1.0f below that, so the reference's
          // double-typed "1. - ly" (feature_refine_kernel.cu:53-54) needs no fp64 here
          const float hy = 1.f - fy;
          const float hx = 1.f - fx;
          const float w1 = hy * hx, w2 = hy * fx, w3 = fy * hx, w4 = fy * fx;
          const float val = (w1 * lt[jj] + w2 * rt[jj] + w3 * lb[jj] + w4 * rb[jj]);
          r[j] = idv[jj] + (((pk[k][j] >> 22) & 1) ? val : 0.f);
        }
      }
      d4[tid + k * FRC_BLOCK] = make_float4(r[0], r[1], r[2], r[3]);
      // keep the quads sequential: interleaving all F4 of them for ILP costs > 128 VGPRs (spills)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < G) write_plane(lds + ((c + 1) & 1) * bufsz);
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------
// Persistent, double-buffered forward (points = 1, power-of-two W, H*W % 256 == 0).
// ONE workgroup per CU walks over its share of tiles (a tile = cpb planes) with two LDS
// buffers: the global loads of tile t+1 are in flight (in registers) while tile t is sampled
// out of LDS, then written to the other buffer behind a single barrier.
// DBG: ablation bits (1 no tap-table loads, 2 no gathers, 4 no stores, 8 no plane loads).
// ----------------------------------------------------------------------------------------
constexpr int FRQ_MAX_F4 = 4;  // float4 per thread per tile: tiles are <= 16 384 floats

template <bool CONSEC, bool DBG>
__global__ __launch_bounds__(FRP_BLOCK) void fr_forward_persist(const float* __restrict__ feat,
                                                                const float* __restrict__ table, int C, int H,
                                                                int logW, int logHW, int cpb, int tiles,
                                                                int bufsz, float* __restrict__ out, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int W = 1 << logW, HW = 1 << logHW;
  const int pitch = W + 1;
  const int psz = H * pitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpb = (tiles + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * tpb;
  const int t1 = min(tiles, t0 + tpb);
  if (t0 >= t1) return;
  const int tile_f4 = (cpb * HW) >> 2;  // float4 per tile (<= 4096)
  const int tiles_per_img = C / cpb;    // C % cpb == 0 is a launch precondition
  const bool no_taps = DBG && (dbg & 1), no_gather = DBG && (dbg & 2), no_store = DBG && (dbg & 4),
             no_load = DBG && (dbg & 8);

  float4 v[FRQ_MAX_F4];
  auto load_tile = [&](int t) {
    const float4* s4 = reinterpret_cast<const float4*>(feat + (size_t)t * cpb * HW);
#pragma unroll
    for (int k = 0; k < FRQ_MAX_F4; k++) {
      const int i = tid + k * FRP_BLOCK;
      if (i < tile_f4) v[k] = s4[i];
    }
  };
  auto write_tile = [&](float* buf) {
#pragma unroll
    for (int k = 0; k < FRQ_MAX_F4; k++) {
      const int i = tid + k * FRP_BLOCK;
      if (i < tile_f4) {
        const int e = i << 2;
        const int ch = e >> logHW, r = e & (HW - 1);
        float* d = buf + ch * psz + (r >> logW) * pitch + (r & (W - 1));
        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
      }
    }
  };

  load_tile(t0);
  write_tile(lds);
  __syncthreads();
  for (int t = t0; t < t1; t++) {
    const float* buf = lds + ((t - t0) & 1) * bufsz;
    float* nbuf = lds + (((t - t0) & 1) ^ 1) * bufsz;
    if (t + 1 < t1 && !no_load) load_tile(t + 1);
    const int n = t / tiles_per_img;
    const float* tb = table + (size_t)n * HW * 5;
    float* dst = out + (size_t)t * cpb * HW;
    if (CONSEC) {
      // a wavefront owns 256 consecutive positions; for j = 0..3 lane l works on position
      // 64 j + l: the 64 lanes of every LDS instruction address 64 consecutive rows (or columns)
      // of the pitch-(W+1) plane => distinct banks.  Results are transposed inside 4-lane
      // groups so that every lane still issues one 16-byte store.
      const int chunks = HW >> 8;
      constexpr int NWAVES = FRP_BLOCK / 64;
      for (int chunk = wave; chunk < chunks; chunk += NWAVES) {
        Tap tp[4];
        int self[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = (chunk << 8) + 64 * j + lane;
          self[j] = (p >> logW) * pitch + (p & (W - 1));
          if (no_taps) {
            tp[j].o00 = tp[j].o01 = tp[j].o10 = tp[j].o11 = self[j];
            tp[j].w1 = tp[j].w2 = tp[j].w3 = tp[j].w4 = 0.25f;
            tp[j].valid = true;
          } else {
            tp[j] = unpack_tap(tb + (size_t)p * 5, pitch);
          }
        }
        const int store_off = (chunk << 8) + 64 * (lane & 3) + (lane & ~3);
        for (int ch = 0; ch < cpb; ch++) {
          float r[4];
          sample4(buf + ch * psz, self, tp, r, !no_gather);
          quad_transpose(r, lane);
          if (!no_store || r[0] == 12345.678f)
            *reinterpret_cast<float4*>(dst + ((size_t)ch << logHW) + store_off) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    } else {
      const float4* tb4 = reinterpret_cast<const float4*>(tb);
      const int quads = HW >> 2;
      for (int qd = tid; qd < quads; qd += FRP_BLOCK) {
        const int hw0 = qd << 2;
        const int s0 = (hw0 >> logW) * pitch + (hw0 & (W - 1));
        const int self[4] = {s0, s0 + 1, s0 + 2, s0 + 3};
        Tap tp[4];
        if (no_taps) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            tp[j].o00 = tp[j].o01 = tp[j].o10 = tp[j].o11 = self[j];
            tp[j].w1 = tp[j].w2 = tp[j].w3 = tp[j].w4 = 0.25f;
            tp[j].valid = true;
          }
        } else {
          float td[20];
#pragma unroll
          for (int k = 0; k < 5; k++) {
            const float4 q = tb4[qd * 5 + k];
            td[4 * k] = q.x; td[4 * k + 1] = q.y; td[4 * k + 2] = q.z; td[4 * k + 3] = q.w;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) tp[j] = unpack_tap(td + 5 * j, pitch);
        }
        for (int ch = 0; ch < cpb; ch++) {
          float r[4];
          sample4(buf + ch * psz, self, tp, r, !no_gather);
          if (!no_store || r[0] == 12345.678f)
            *reinterpret_cast<float4*>(dst + ((size_t)ch << logHW) + hw0) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
    if (t + 1 < t1) write_tile(nbuf);
    __syncthreads();
  }
}

// dynamic LDS above 64 KB has to be opted into once per kernel
template <typename K>
inline void allow_big_lds(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            bytes);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int plane_cpb(int C, int H, int W) {
  int psz = H * (W + 1);
  if (psz > FRP_LDS_FLOATS) return 0;
  int cpb = FRP_LDS_FLOATS / psz;
  if (cpb > C) cpb = C;
  return cpb < 1 ? 0 : cpb;
}

inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}

inline int cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

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
  const int logW = ilog2_exact(W), logHW = ilog2_exact(H * W);
  // Measured on MI355X (tools/microbench.py fr, tools/fr_ablate.py; level 0 = 4x256x128x128):
  // plane 47 us, persist-quad 53 us, persist-consec 49 us; with HBM loads OR stores disabled the
  // persistent kernel drops to 32-34 us, with both to 25 us: a wave that stores and then waits
  // for its next loads also waits for the store acknowledgements (one in-order vmcnt queue), which
  // costs the persistent form what double buffering gains.  The plain plane kernel (workgroups
  // exit right after their stores) is therefore the default; 5 / 6 select the persistent forms.
  const bool persist = plane && (g_r3_fr_impl == 5 || g_r3_fr_impl == 6) && points == 1 && ws && logW >= 2 && logHW >= 8 &&
                       ws_bytes >= r3k_fr_workspace_bytes(N, H, W, points) && aligned16(feat) &&
                       aligned16(out) && aligned16(ws);
  // Round-1 status of the forward variants at level 0 (4x256x128x128, tools/microbench.py fr):
  // plane 43.5 us, chan 43.8 (spills), skew 45.4, persist 49-53; a copy through LDS with the same
  // tiling runs at 20 us (tools/probes/lds_copy_probe.hip).  None of the workspace variants beats
  // the workspace-free plane kernel yet, so it stays the default; 5-8 select the others.
  // ... until the chan kernel kept its taps in registers WITHOUT spilling (an opaque copy stops LICM
  // from hoisting 12 values per position out of the channel loop): 34 us at level 0 => default for
  // 128 x 128 planes when the batch gives every workgroup >= 2 channels.
  const bool skew = plane && (g_r3_fr_impl == 7 || g_r3_fr_impl == 8 || (g_r3_fr_impl == 0 && logHW == 14)) &&
                    points == 1 && ws && logW >= 2 &&
                    W <= 1024 && H <= 1024 && (logHW == 12 || logHW == 14) &&
                    ws_bytes >= r3k_fr_workspace_bytes(N, H, W, points) && aligned16(feat) && aligned16(out) &&
                    aligned16(ws);
  // channels per workgroup for the chan kernel: a power of two, >= one workgroup per CU
  int G = 1;
  while (G * 2 <= 16 && C % (G * 2) == 0 && (size_t)N * C / (G * 2) >= (size_t)cu_count()) G *= 2;
  const bool chan = skew && (g_r3_fr_impl == 8 || g_r3_fr_impl == 0) && G >= 2;
  if (chan) {
    float* table = reinterpret_cast<float*>(ws);
    const int total = N * H * W;
    hipLaunchKernelGGL(fr_taps3_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, boxes, total, H, W,
                       scale, table);
    static bool once = (allow_big_lds(fr_forward_chan<4>, 160 * 1024), allow_big_lds(fr_forward_chan<1>, 160 * 1024), true);
    (void)once;
    const int bufsz = (H * W + H / 4 + 7) & ~3;
    const size_t lds = (size_t)2 * bufsz * sizeof(float);
    if (logHW == 14)
      hipLaunchKernelGGL(fr_forward_chan<4>, dim3(N * C / G), dim3(FRC_BLOCK), lds, stream, feat, table, C, G, logW,
                         logHW, bufsz, out);
    else
      hipLaunchKernelGGL(fr_forward_chan<1>, dim3(N * C / G), dim3(FRC_BLOCK), lds, stream, feat, table, C, G, logW,
                         logHW, bufsz, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (skew && g_r3_fr_impl != 0) {
    float* table = reinterpret_cast<float*>(ws);
    const int total = N * H * W;
    hipLaunchKernelGGL(fr_taps_xy_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, boxes, total, H, W,
                       scale, table);
    static bool once = (allow_big_lds(fr_forward_skew<8>, 68 * 1024), allow_big_lds(fr_forward_skew<2>, 68 * 1024), true);
    (void)once;
    const size_t lds = ((size_t)H * W + H / 4 + 4) * sizeof(float);
    if (logHW == 14)
      hipLaunchKernelGGL(fr_forward_skew<8>, dim3(N * C), dim3(FRS_BLOCK), lds, stream, feat, table, C, logW, logHW, out);
    else
      hipLaunchKernelGGL(fr_forward_skew<2>, dim3(N * C), dim3(FRS_BLOCK), lds, stream, feat, table, C, logW, logHW, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (persist) {
    // planes per tile: a power of two, tile <= 16 384 floats, >= 512 tiles when the level allows
    int pc = 1;
    while (pc * 2 * H * W <= 16384 && pc * 2 * H * (W + 1) <= FRP_LDS_FLOATS && C % (pc * 2) == 0 &&
           (size_t)N * C / (pc * 2) >= 512)
      pc *= 2;
    float* table = reinterpret_cast<float*>(ws);
    const int total = N * H * W;
    hipLaunchKernelGGL(fr_taps_kernel<1>, dim3((total + 255) / 256), dim3(256), 0, stream, boxes, total, H, W,
                       scale, table);
    const int tiles = N * C / pc;
    const int bufsz = ((pc * H * (W + 1)) + 3) & ~3;
    const size_t lds = (size_t)2 * bufsz * sizeof(float);
    static bool once = (allow_big_lds(fr_forward_persist<false, false>, 160 * 1024),
                        allow_big_lds(fr_forward_persist<true, false>, 160 * 1024),
                        allow_big_lds(fr_forward_persist<false, true>, 160 * 1024),
                        allow_big_lds(fr_forward_persist<true, true>, 160 * 1024), true);
    (void)once;
    const int grid = tiles < cu_count() ? tiles : cu_count();
    const bool consec = g_r3_fr_impl == 6;
#define R3_PERSIST(CS, DB)                                                                                   \
  hipLaunchKernelGGL((fr_forward_persist<CS, DB>), dim3(grid), dim3(FRP_BLOCK), lds, stream, feat, table, C, \
                     H, logW, logHW, pc, tiles, bufsz, out, g_r3_fr_dbg)
    if (g_r3_fr_dbg) { if (consec) R3_PERSIST(true, true); else R3_PERSIST(false, true); }
    else { if (consec) R3_PERSIST(true, false); else R3_PERSIST(false, false); }
#undef R3_PERSIST
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (plane) {
    // spread small levels over more workgroups: cap planes per workgroup so that the grid
    // has >= 512 workgroups when possible
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    static bool once = (allow_big_lds(fr_forward_plane<1, true, 1>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_forward_plane<1, true, 2>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_forward_plane<1, true, 0>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_forward_plane<1, false, 0>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_forward_plane<5, false, 0>, FRP_LDS_FLOATS * 4), true);
    (void)once;
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

int r3k_fr_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W,
                    float scale, int points, float* bottom_grad, int overwrite,
                    hipStream_t stream) {
  if (points != 1 && points != 5) return -1;
  if (N == 0 || C == 0 || H == 0 || W == 0) return 0;
  int cpb = plane_cpb(C, H, W);
  const bool plane = g_r3_fr_impl != 1 && cpb > 0;
  if (plane) {
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    static bool once = (allow_big_lds(fr_backward_plane<1, true>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_backward_plane<1, false>, FRP_LDS_FLOATS * 4),
                        allow_big_lds(fr_backward_plane<5, false>, FRP_LDS_FLOATS * 4), true);
    (void)once;
    const bool vec = (W % 4 == 0) && aligned16(top_grad) && aligned16(boxes) && aligned16(bottom_grad);
#define R3_BWD(P, V) hipLaunchKernelGGL((fr_backward_plane<P, V>), grid, dim3(FRP_BLOCK), lds, stream, top_grad, boxes, C, H, W, scale, cpb, overwrite, bottom_grad)
    if (points == 1) { if (vec) R3_BWD(1, true); else R3_BWD(1, false); }
    else R3_BWD(5, false);
#undef R3_BWD
  } else {
    if (overwrite) {
      if (hipMemsetAsync(bottom_grad, 0, (size_t)N * C * H * W * sizeof(float), stream) != hipSuccess)
        return -2;
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
