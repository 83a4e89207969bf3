// Probe: how fast can "load tile -> LDS -> barrier -> LDS -> store" run on MI355X compared with a
// plain copy, as a function of workgroup size / tile size / LDS write form?
// build: hipcc --offload-arch=gfx950 -O3 -o lds_copy_probe lds_copy_probe.hip ; run: ./lds_copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void copy_direct(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

// one tile of TILE_F4 float4 per workgroup; MODE 0: b128 LDS writes linear, 1: scalar writes with pitch 129
template <int THREADS, int TILE_F4, int MODE>
__global__ __launch_bounds__(THREADS) void copy_lds(const float4* __restrict__ in, float4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PER = TILE_F4 / THREADS;
  const float4* src = in + (size_t)blockIdx.x * TILE_F4;
  float4* dst = out + (size_t)blockIdx.x * TILE_F4;
  float4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) v[k] = src[threadIdx.x + k * THREADS];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    int i = threadIdx.x + k * THREADS;
    if (MODE == 0) {
      reinterpret_cast<float4*>(lds)[i] = v[k];
    } else {
      int e = i * 4, y = e >> 7, x = e & 127;
      float* d = lds + y * 129 + x;
      d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    int i = threadIdx.x + k * THREADS;
    float4 r;
    if (MODE == 0) {
      r = reinterpret_cast<float4*>(lds)[i];
    } else {
      int e = i * 4, y = e >> 7, x = e & 127;
      const float* d = lds + y * 129 + x;
      r = make_float4(d[0], d[1], d[2], d[3]);
    }
    dst[i] = r;
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

int main() {
  const size_t n = (size_t)4 * 256 * 128 * 128;  // floats (67 MB)
  const size_t n4 = n / 4;
  float *in, *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
  CK(hipMemset(in, 1, n * 4));
  auto report = [&](const char* name, float us) { printf("%-44s %8.1f us  %8.1f GB/s\n", name, us, 2.0 * n * 4 / us / 1e3); };
  report("direct copy, 2048 x 256 grid-stride", time_us([&] { hipLaunchKernelGGL(copy_direct, dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); }, 20));
  report("direct copy, n4/256 blocks x 256", time_us([&] { hipLaunchKernelGGL(copy_direct, dim3(n4 / 256), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); }, 20));
#define RUN(T, F4, M, LDSB, NAME)                                                                              \
  {                                                                                                            \
    hipFuncSetAttribute((const void*)copy_lds<T, F4, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    report(NAME, time_us([&] { hipLaunchKernelGGL((copy_lds<T, F4, M>), dim3(n4 / F4), dim3(T), LDSB, 0, (const float4*)in, (float4*)out); }, 20)); \
  }
  RUN(1024, 4096, 0, 65536, "lds tile 64KB, 1024 thr, b128 linear");
  RUN(1024, 4096, 1, 66048 + 64, "lds tile 64KB, 1024 thr, scalar pitch129");
  RUN(512, 4096, 0, 65536, "lds tile 64KB,  512 thr, b128 linear");
  RUN(512, 4096, 1, 66048 + 64, "lds tile 64KB,  512 thr, scalar pitch129");
  RUN(256, 4096, 0, 65536, "lds tile 64KB,  256 thr, b128 linear");
  RUN(256, 1024, 0, 16384, "lds tile 16KB,  256 thr, b128 linear");
  RUN(256, 1024, 1, 16512 + 64, "lds tile 16KB,  256 thr, scalar pitch129");
  RUN(512, 2048, 0, 32768, "lds tile 32KB,  512 thr, b128 linear");
  RUN(1024, 2048, 0, 32768, "lds tile 32KB, 1024 thr, b128 linear");
  return 0;
}
