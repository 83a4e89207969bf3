// r3_epilogue.hip -- convolution epilogue for the inference model around the hot path:
//   y = act(y + bias[c] (+ residual)),  in place, one pass.
// The reference's benchmark folds BatchNorm into the convolutions (tools/analysis_tools/benchmark.py:9,
// 31,88-89: mmcv.cnn.fuse_conv_bn); what is left after every MIOpen convolution is a bias add, the
// ReLU and, at the end of a ResNet bottleneck, the residual add -- two or three full passes over the
// activation as separate elementwise launches (19 % of the step's kernel time).  HBM-bound:
// 8 B per element (12 with a residual).
#include <hip/hip_runtime.h>

#include "r3_kernels.h"

namespace {

// NHWC (channels_last): channel = fastest index.  One float4 = 4 consecutive channels (C % 4 == 0).
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(float4* __restrict__ y, const float4* __restrict__ bias,
                                                            const float4* __restrict__ res, long long n4, int c4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = y[i];
    const float4 b = bias[(int)(i % c4)];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) {
      const float4 r = res[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    y[i] = v;
  }
}

// NCHW: one (n, c) plane per blockIdx.y, `inner` = H * W elements sharing one bias value.
template <bool RES, bool RELU, bool VEC>
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                            const float* __restrict__ res, int C, int inner) {
  const long long plane = blockIdx.y;
  const float b = bias[(int)(plane % C)];
  float* yp = y + plane * inner;
  const float* rp = RES ? res + plane * inner : nullptr;
  if (VEC) {
    const int n4 = inner >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 v = reinterpret_cast<float4*>(yp)[i];
      v.x += b; v.y += b; v.z += b; v.w += b;
      if (RES) {
        const float4 r = reinterpret_cast<const float4*>(rp)[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < inner; i += gridDim.x * 256) {
      float v = yp[i] + b;
      if (RES) v += rp[i];
      if (RELU) v = fmaxf(v, 0.f);
      yp[i] = v;
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// y: `outer` x C x `inner` elements with the channel in the middle (NCHW: outer = N, inner = H*W;
// channels_last: outer = N*H*W, inner = 1).  residual: same shape and layout, or null.
int r3k_bias_act(float* y, const float* bias, const float* residual, long long outer, int C, long long inner, int relu,
                 hipStream_t stream) {
  if (!y || !bias || outer < 0 || C <= 0 || inner <= 0) return -1;
  if (outer == 0) return 0;
  const long long total = outer * C * inner;
  if (inner == 1) {
    if ((C & 3) || !al16(y) || !al16(bias) || (residual && !al16(residual))) return -1;
    const long long n4 = total >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 g((unsigned)blocks), b(256);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    const float4* r4 = reinterpret_cast<const float4*>(residual);
    float4* y4 = reinterpret_cast<float4*>(y);
#define R3_NHWC(RS, RL) hipLaunchKernelGGL((bias_act_nhwc_kernel<RS, RL>), g, b, 0, stream, y4, b4, r4, n4, C >> 2)
    if (residual) { if (relu) R3_NHWC(true, true); else R3_NHWC(true, false); }
    else { if (relu) R3_NHWC(false, true); else R3_NHWC(false, false); }
#undef R3_NHWC
  } else {
    if (inner > 0x7fffffffLL || outer * C > 65535) return -1;  // (grid.y limit; callers fall back to torch)
    const bool vec = (inner % 4 == 0) && al16(y) && (!residual || al16(residual));
    const int per = vec ? (int)(inner / 4) : (int)inner;
    int bx = (per + 255) / 256;
    if (bx > 64) bx = 64;
    const dim3 g(bx, (unsigned)(outer * C)), b(256);
#define R3_NCHW(RS, RL, V) \
  hipLaunchKernelGGL((bias_act_nchw_kernel<RS, RL, V>), g, b, 0, stream, y, bias, residual, C, (int)inner)
    if (vec) {
      if (residual) { if (relu) R3_NCHW(true, true, true); else R3_NCHW(true, false, true); }
      else { if (relu) R3_NCHW(false, true, true); else R3_NCHW(false, false, true); }
    } else {
      if (residual) { if (relu) R3_NCHW(true, true, false); else R3_NCHW(true, false, false); }
      else { if (relu) R3_NCHW(false, true, false); else R3_NCHW(false, false, false); }
    }
#undef R3_NCHW
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
