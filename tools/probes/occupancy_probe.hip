// Probe: how many workgroups of a given size / LDS footprint does the runtime say a compute unit holds, and how many
// actually run at once (workgroups record start stamps; those that start within 2 us of the first are resident).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/occupancy_probe tools/probes/occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDSB>
__global__ void spin(unsigned long long* st, int spin_us) {
  __shared__ char lds[LDSB];
  if (threadIdx.x == 0) st[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) {}
  if (lds[(threadIdx.x + 1) & 63] == 77) st[0] = 0;
}

template <int LDSB>
int run(int threads) {
  int dev_blocks = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&dev_blocks, spin<LDSB>, threads, 0));
  const int grid = 256 * 12;
  unsigned long long* st;
  CK(hipMalloc(&st, grid * 8));
  hipLaunchKernelGGL(spin<LDSB>, dim3(grid), dim3(threads), 0, 0, st, 20);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(spin<LDSB>, dim3(grid), dim3(threads), 0, 0, st, 20);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), st, grid * 8, hipMemcpyDeviceToHost));
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  for (auto t : h) early += (t - t0) < 500;  // 5 us
  printf("threads %4d  LDS %6d B: runtime says %d per CU; started within 5 us: %d of %d = %.2f per CU\n", threads, LDSB,
         dev_blocks, early, grid, early / 256.0);
  CK(hipFree(st));
  return 0;
}

int main() {
  if (run<32768>(512)) return 1;
  if (run<16384>(512)) return 1;
  if (run<1024>(512)) return 1;
  if (run<32768>(256)) return 1;
  if (run<1024>(256)) return 1;
  if (run<65536>(1024)) return 1;
  if (run<1024>(1024)) return 1;
  return 0;
}
