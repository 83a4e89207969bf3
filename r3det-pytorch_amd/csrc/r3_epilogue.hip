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

// ReLU that propagates NaN like F.relu (fmaxf(NaN, 0) would return 0 and hide a broken activation)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

// NHWC (channels_last): channel = fastest index.  One float4 = 4 consecutive channels (C % 4 == 0).
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(float4* __restrict__ y, const float4* __restrict__ bias,
                                                            const float4* __restrict__ res, long long n4, int c4) {
  // (a thread's channels: fixed when the grid stride is a multiple of the channel count -- c4 a power of two and the
  // grid capped at 8192 x 256 -- else a 32-bit remainder per step; a 64-bit `i % c4` per element was a division loop
  // of ~40 instructions in front of every 16-byte access)
  const long long stride = (long long)gridDim.x * 256;
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool fixed = stride % c4 == 0;
  unsigned ch = (unsigned)(i0 % c4);
  const unsigned step = (unsigned)(stride % c4);
  float4 bfix = bias[ch];
  for (long long i = i0; i < n4; i += stride) {
    float4 v = y[i];
    const float4 b = fixed ? bfix : bias[ch];
    ch += step;
    if (ch >= (unsigned)c4) ch -= (unsigned)c4;
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) {
      const float4 r = res[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) {
      v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
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
        v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
      }
      reinterpret_cast<float4*>(yp)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < inner; i += gridDim.x * 256) {
      float v = yp[i] + b;
      if (RES) v += rp[i];
      if (RELU) v = relu_nan(v);
      yp[i] = v;
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// channels_last -> NCHW with the FeatureRefineModule's elementwise work in front of the sampler folded in
// (feature_refine_module.py:121-123): out[n, c, p] = (a[n, p, c] + bias_a[c]) + (b[n, p, c] + bias_b[c]) -- the
// bias adds of the two convolutions (separate launches after MIOpen), their sum and the layout switch the
// sampler needs, 2 reads + 1 write per element instead of 7 passes.  64 x 64 tile through LDS: loads run along c,
// stores along p, both 256 B per wavefront.
template <bool TWO>
__global__ __launch_bounds__(256) void mix_to_nchw_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ bias_a,
                                                          const float* __restrict__ bias_b, int C, int HW,
                                                          float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  {
    const int c = c0 + tx;
    const float ba = (bias_a && c < C) ? bias_a[c] : 0.f, bb = (TWO && bias_b && c < C) ? bias_b[c] : 0.f;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
      const int p = p0 + r;
      if (p < HW && c < C) {
        const size_t i = ((size_t)n * HW + p) * C + c;
        float v = bias_a ? a[i] + ba : a[i];
        if (TWO) v = v + (bias_b ? b[i] + bb : b[i]);
        tile[r][tx] = v;
      }
    }
  }
  __syncthreads();
  const int p = p0 + tx;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    if (p < HW && c < C) out[((size_t)n * C + c) * HW + p] = tile[tx][r];
  }
}

}  // namespace

int r3k_mix_to_nchw(const float* a, const float* b, const float* bias_a, const float* bias_b, int N, int C, int H, int W,
                    float* out, hipStream_t stream) {
  if (!a || !out || N < 0 || C <= 0 || H <= 0 || W <= 0 || (!b && bias_b)) return -1;
  if (N == 0) return 0;
  const int HW = H * W;
  const dim3 grid((HW + 63) / 64, (C + 63) / 64, N);
  if (b)
    hipLaunchKernelGGL(mix_to_nchw_kernel<true>, grid, dim3(256), 0, stream, a, b, bias_a, bias_b, C, HW, out);
  else
    hipLaunchKernelGGL(mix_to_nchw_kernel<false>, grid, dim3(256), 0, stream, a, b, bias_a, bias_b, C, HW, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

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
