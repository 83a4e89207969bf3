// r3_fr.hip -- Feature Refinement sampler (rotated feature-align), forward and backward.
//
// Replaces feature_refine_forward_kernel / feature_refine_backward_kernel
// (fr/src/feature_refine_kernel.cu:112-230).  The reference runs one thread per output
// ELEMENT (n,c,h,w) and re-derives the sample point from the 20-byte box for each of the
// C channels; here a thread owns one POSITION (n,h,w): it turns the box into tap offsets
// and weights once and then streams over channels.
//
// Two implementations of each direction:
//   generic   : taps gathered from global memory (L1/L2) -- any H x W.
//   lds-plane : the (n,c) plane (H x (W+1) floats, padded) is staged in LDS with coalesced
//               16-byte loads and every tap is an LDS read.  Because the reference samples
//               row <- x_ctr*scale, column <- y_ctr*scale (feature_refine_kernel.cu:131-132)
//               the gather of a well-behaved box field is a TRANSPOSE of the plane: adjacent
//               lanes hit adjacent rows, i.e. a different cache line per lane.  Through LDS
//               that costs nothing (odd row pitch => conflict-free), and HBM sees exactly one
//               read and one write per element.
#include <hip/hip_runtime.h>

#include "r3_kernels.h"
#include "r3_trig.h"

int g_r3_fr_impl = 0;

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

template <typename P>
__device__ __forceinline__ float tap_value(const Tap& t, P plane) {
  if (!t.valid) return 0.f;
  float lt = plane[t.o00], rt = plane[t.o01], lb = plane[t.o10], rb = plane[t.o11];
  return (t.w1 * lt + t.w2 * rt + t.w3 * lb + t.w4 * rb);
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
// lds-plane kernels: one workgroup owns CPB consecutive (n,c) planes, staged in LDS.
// A thread owns QUADS of 4 adjacent positions: its 4 boxes are one contiguous 80-byte read
// (5 x 16 B), the 4 results one 16-byte store; all global loads of a round are issued before
// the first use so a wave keeps >= 4 (stage) / 5 (boxes) requests in flight.
// ----------------------------------------------------------------------------------------
constexpr int FRP_BLOCK = 1024;
constexpr int FRP_LDS_FLOATS = 17 * 1024;  // 68 KB: 128 x 129 plane (66 KB) fits, 2 WGs / CU
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

template <int POINTS, bool VEC>
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
  const int nc = min(cpb, C - c0);
  const float* src = feat + ((size_t)n * C + c0) * HW;
  float* dst = out + ((size_t)n * C + c0) * HW;
  stage_planes<VEC>(src, lds, nc, HW, W, pitch, psz);
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
      const int y = hw0 / W, x = hw0 - y * W;  // W % 4 == 0: the quad stays in one row
      const int self = y * pitch + x;
      Tap taps[4][POINTS];
#pragma unroll
      for (int j = 0; j < 4; j++) make_taps<POINTS>(bq + 5 * j, scale, H, W, pitch, taps[j]);
      for (int ch = 0; ch < nc; ch++) {
        const float* plane = lds + ch * psz;
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float v = plane[self + j];
#pragma unroll
          for (int p = 0; p < POINTS; p++) v += tap_value(taps[j][p], plane);
          r[j] = v;
        }
        *reinterpret_cast<float4*>(dst + (size_t)ch * HW + hw0) = make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  } else {
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

// dynamic LDS above 64 KB has to be opted into once per kernel
template <typename K>
inline void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            FRP_LDS_FLOATS * (int)sizeof(float));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int plane_cpb(int C, int H, int W) {
  int psz = H * (W + 1);
  if (psz > FRP_LDS_FLOATS) return 0;
  int cpb = FRP_LDS_FLOATS / psz;
  if (cpb > C) cpb = C;
  // keep at least ~1024 workgroups in flight when the level is large enough
  return cpb < 1 ? 0 : cpb;
}

}  // namespace

int r3k_fr_forward(const float* feat, const float* boxes, int N, int C, int H, int W, float scale,
                   int points, float* out, hipStream_t stream) {
  if (points != 1 && points != 5) return -1;
  if (N == 0 || C == 0 || H == 0 || W == 0) return 0;
  int cpb = plane_cpb(C, H, W);
  bool plane = (g_r3_fr_impl == 2) || (g_r3_fr_impl == 0);
  if (cpb == 0) plane = false;
  if (plane) {
    // spread small levels over more workgroups: cap planes per workgroup so that the grid
    // has >= 512 workgroups when possible
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    static bool once = (allow_big_lds(fr_forward_plane<1, true>), allow_big_lds(fr_forward_plane<1, false>),
                        allow_big_lds(fr_forward_plane<5, false>), true);
    (void)once;
    const bool vec = (W % 4 == 0) && aligned16(feat) && aligned16(boxes) && aligned16(out);
#define R3_FWD(P, V) hipLaunchKernelGGL((fr_forward_plane<P, V>), grid, dim3(FRP_BLOCK), lds, stream, feat, boxes, C, H, W, scale, cpb, out)
    // points = 5 keeps 5 taps per position live: the quad form would spill, use the scalar form
    if (points == 1) { if (vec) R3_FWD(1, true); else R3_FWD(1, false); }
    else R3_FWD(5, false);
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
  bool plane = (g_r3_fr_impl == 2) || (g_r3_fr_impl == 0);
  if (cpb == 0) plane = false;
  if (plane) {
    while (cpb > 1 && (size_t)N * ((C + cpb - 1) / cpb) < 512) cpb = (cpb + 1) / 2;
    dim3 grid((C + cpb - 1) / cpb, N);
    size_t lds = (size_t)cpb * H * (W + 1) * sizeof(float);
    static bool once = (allow_big_lds(fr_backward_plane<1, true>), allow_big_lds(fr_backward_plane<1, false>),
                        allow_big_lds(fr_backward_plane<5, false>), true);
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
