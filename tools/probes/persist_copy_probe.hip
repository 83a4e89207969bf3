// Probe: what does a PERSISTENT workgroup (one per CU, looping over G planes of 64 KB) cost in
// HBM throughput compared with one-plane-per-workgroup launches?  Isolates the memory-system side
// of fr_forward_chan / _cell (no sampling math at all).
// build: hipcc --offload-arch=gfx950 -O3 -o persist_copy_probe persist_copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int PLANE_F4 = 4096;  // 64 KB

// MODE 0: load -> store            1: + barrier per plane        2: register prefetch of the next plane
// MODE 3: double-buffered through LDS like fr_forward_chan (prefetch -> regs, LDS -> store, regs -> LDS, barrier)
// MODE 4: like 3 with a two-plane-deep prefetch
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void pcopy(const float4* __restrict__ in, float4* __restrict__ out, int G,
                                                 int interleave, int nwg) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  constexpr int PER = PLANE_F4 / THREADS;
  const int tid = threadIdx.x;
  auto plane_of = [&](int c) { return interleave ? (size_t)c * nwg + blockIdx.x : (size_t)blockIdx.x * G + c; };
  float4 v[PER], w[PER];
  if (MODE <= 1) {
    for (int c = 0; c < G; c++) {
      const float4* s = in + plane_of(c) * PLANE_F4;
      float4* d = out + plane_of(c) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) v[k] = s[tid + k * THREADS];
#pragma unroll
      for (int k = 0; k < PER; k++) d[tid + k * THREADS] = v[k];
      if (MODE == 1) __syncthreads();
    }
  } else if (MODE == 2) {
    const float4* s = in + plane_of(0) * PLANE_F4;
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = s[tid + k * THREADS];
    for (int c = 0; c < G; c++) {
      if (c + 1 < G) {
        const float4* s2 = in + plane_of(c + 1) * PLANE_F4;
#pragma unroll
        for (int k = 0; k < PER; k++) w[k] = s2[tid + k * THREADS];
      }
      float4* d = out + plane_of(c) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) d[tid + k * THREADS] = v[k];
#pragma unroll
      for (int k = 0; k < PER; k++) v[k] = w[k];
    }
  } else if (MODE == 3) {
    const float4* s = in + plane_of(0) * PLANE_F4;
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = s[tid + k * THREADS];
#pragma unroll
    for (int k = 0; k < PER; k++) lds4[tid + k * THREADS] = v[k];
    __syncthreads();
    for (int c = 0; c < G; c++) {
      const float4* buf = lds4 + (c & 1) * PLANE_F4;
      if (c + 1 < G) {
        const float4* s2 = in + plane_of(c + 1) * PLANE_F4;
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = s2[tid + k * THREADS];
      }
      float4* d = out + plane_of(c) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) d[tid + k * THREADS] = buf[tid + k * THREADS];
      if (c + 1 < G) {
        float4* nb = lds4 + ((c + 1) & 1) * PLANE_F4;
#pragma unroll
        for (int k = 0; k < PER; k++) nb[tid + k * THREADS] = v[k];
      }
      __syncthreads();
    }
  } else {
    const float4* s = in + plane_of(0) * PLANE_F4;
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = s[tid + k * THREADS];
    if (G > 1) {
      const float4* s2 = in + plane_of(1) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) w[k] = s2[tid + k * THREADS];
    }
#pragma unroll
    for (int k = 0; k < PER; k++) lds4[tid + k * THREADS] = v[k];
    __syncthreads();
    for (int c = 0; c < G; c += 2) {
      if (c + 2 < G) {
        const float4* s2 = in + plane_of(c + 2) * PLANE_F4;
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = s2[tid + k * THREADS];
      }
      float4* d = out + plane_of(c) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) d[tid + k * THREADS] = lds4[tid + k * THREADS];
      if (c + 1 < G) {
#pragma unroll
        for (int k = 0; k < PER; k++) lds4[PLANE_F4 + tid + k * THREADS] = w[k];
      }
      __syncthreads();
      if (c + 1 >= G) break;
      if (c + 3 < G) {
        const float4* s2 = in + plane_of(c + 3) * PLANE_F4;
#pragma unroll
        for (int k = 0; k < PER; k++) w[k] = s2[tid + k * THREADS];
      }
      d = out + plane_of(c + 1) * PLANE_F4;
#pragma unroll
      for (int k = 0; k < PER; k++) d[tid + k * THREADS] = lds4[PLANE_F4 + tid + k * THREADS];
      if (c + 2 < G) {
#pragma unroll
        for (int k = 0; k < PER; k++) lds4[tid + k * THREADS] = v[k];
      }
      __syncthreads();
    }
  }
}

template <typename F>
float time_us(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

template <int THREADS, int MODE>
void run(const float* in, float* out, int planes, int G, int interleave, const char* name) {
  const int nwg = planes / G;
  const size_t lds = MODE >= 3 ? 2 * 65536 : 0;
  hipFuncSetAttribute((const void*)pcopy<THREADS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  float us = time_us([&] { hipLaunchKernelGGL((pcopy<THREADS, MODE>), dim3(nwg), dim3(THREADS), lds, 0, (const float4*)in, (float4*)out, G, interleave, nwg); }, 20);
  printf("planes %5d G %2d wgs %5d thr %4d %-34s %s %8.1f us %8.1f GB/s\n", planes, G, nwg, THREADS, name,
         interleave ? "interleaved" : "consecutive", us, 2.0 * planes * 65536.0 / us / 1e3);
}

int main() {
  const int maxplanes = 16 * 256;
  float *in, *out;
  CK(hipMalloc(&in, (size_t)maxplanes * 65536)); CK(hipMalloc(&out, (size_t)maxplanes * 65536));
  CK(hipMemset(in, 1, (size_t)maxplanes * 65536));
  for (int planes : {1024, 4096}) {
    for (int G : {1, 4, 16}) {
      if (planes / G < 256) continue;
      for (int il = 0; il < 2; il++) {
        run<1024, 0>(in, out, planes, G, il, "load->store");
        run<1024, 1>(in, out, planes, G, il, "load->store + barrier");
        run<1024, 2>(in, out, planes, G, il, "register prefetch");
        run<1024, 3>(in, out, planes, G, il, "LDS double buffer (chan)");
        run<1024, 4>(in, out, planes, G, il, "LDS double buffer, depth 2 (deep)");
        run<256, 0>(in, out, planes, G, il, "load->store, 256 thr");
        run<256, 2>(in, out, planes, G, il, "register prefetch, 256 thr");
        if (G == 1) break;
      }
    }
  }
  return 0;
}
