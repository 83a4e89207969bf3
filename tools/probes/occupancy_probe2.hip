// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/occupancy_probe2 tools/probes/occupancy_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDSB, int NV, bool LB>
__device__ __forceinline__ void body(unsigned long long* st, int spin_us, const float* in, float* out) {
  __shared__ char lds[LDSB];
  if (threadIdx.x == 0) st[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = in[threadIdx.x + i * 512];
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) {
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = v[i] * 1.0001f + v[(i + 1) % NV];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) s += v[i];
  if (lds[(threadIdx.x + 1) & 63] == 77 || s == 123.f) out[threadIdx.x] = s;
}
template <int LDSB, int NV> __global__ void k_plain(unsigned long long* st, int us, const float* in, float* out) { body<LDSB, NV, false>(st, us, in, out); }
template <int LDSB, int NV> __global__ __launch_bounds__(512) void k_lb(unsigned long long* st, int us, const float* in, float* out) { body<LDSB, NV, true>(st, us, in, out); }

template <typename K>
int run(K kern, const char* what) {
  int dev_blocks = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&dev_blocks, kern, 512, 0));
  hipFuncAttributes a;
  CK(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kern)));
  const int grid = 2048;
  unsigned long long* st; float *in, *out;
  CK(hipMalloc(&st, grid * 8)); CK(hipMalloc(&in, 512 * 64 * 4)); CK(hipMalloc(&out, 4096)); CK(hipMemset(in, 0, 512 * 64 * 4));
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, st, 20, in, out); CK(hipDeviceSynchronize()); }
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), st, grid * 8, hipMemcpyDeviceToHost));
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  for (auto t : h) early += (t - t0) < 500;
  printf("%-28s regs %3d lds %6zu: runtime says %d per CU; started within 5 us: %4d = %.2f per CU\n", what, a.numRegs, a.sharedSizeBytes,
         dev_blocks, early, early / 256.0);
  return 0;
}

int main() {
  if (run(k_plain<32768, 4>, "plain, 4 values")) return 1;
  if (run(k_lb<32768, 4>, "launch_bounds, 4 values")) return 1;
  if (run(k_lb<32768, 24>, "launch_bounds, 24 values")) return 1;
  if (run(k_lb<32768, 32>, "launch_bounds, 32 values")) return 1;
  if (run(k_lb<32768, 40>, "launch_bounds, 40 values")) return 1;
  if (run(k_lb<32768, 56>, "launch_bounds, 56 values")) return 1;
  if (run(k_lb<16384, 40>, "launch_bounds, 40 v, 16 KB")) return 1;
  return 0;
}
