// Probe: issue rate of the compare-and-count idiom of mc_sort_prepare_kernel (v_cmp_ge_u32 + v_addc / v_cndmask)
// against plain v_add_u32, per wavefront and SIMD.  One workgroup of 256 threads per CU x 4, long loops, events.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/vcmp_rate_probe tools/probes/vcmp_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ keys, unsigned* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned tile[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = keys[i];
  __syncthreads();
  const unsigned ui = keys[1024 + threadIdx.x], uj = keys[2048 + threadIdx.x];
  unsigned c0 = 0, c1 = 0;
  const uint4* t4 = reinterpret_cast<const uint4*>(tile) + (threadIdx.x >> 3) * 8;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint4 u = t4[q];
      if (MODE == 0) {  // compare and count, one candidate
        c0 += (u.x >= ui) + (u.y >= ui) + (u.z >= ui) + (u.w >= ui);
      } else if (MODE == 1) {  // two candidates per key
        c0 += (u.x >= ui) + (u.y >= ui) + (u.z >= ui) + (u.w >= ui);
        c1 += (u.x >= uj) + (u.y >= uj) + (u.z >= uj) + (u.w >= uj);
      } else if (MODE == 2) {  // plain adds, as many vector instructions as MODE 0 should need (8)
        c0 += u.x; c1 += u.y; c0 += u.z; c1 += u.w; c0 += ui; c1 += uj; c0 ^= c1; c1 += 3;
      } else if (MODE == 3) {  // arithmetic compare: sign of the difference of 31-bit keys, no VCC
        c0 += ((ui - u.x - 1) >> 31) + ((ui - u.y - 1) >> 31) + ((ui - u.z - 1) >> 31) + ((ui - u.w - 1) >> 31);
      }
    }
    asm volatile("" : "+v"(c0), "+v"(c1));
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0 + c1;
}

template <int MODE>
int run(const char* what, const unsigned* dk, unsigned* dout, int blocks, int iters, double compares_per_iter) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dk, dout, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dk, dout, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double ns_per_iter_per_wave = ms * 1e6 / iters / waves_per_simd;
  printf("%-34s %8.1f us   %6.1f ns per 32-key step and wave-slot  (%5.2f T compares/s)\n", what, ms * 1e3, ns_per_iter_per_wave,
         compares_per_iter * 256.0 * blocks * iters / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  unsigned h[4096];
  for (int i = 0; i < 4096; i++) h[i] = (unsigned)(i * 2654435761u) >> 1;
  unsigned *dk, *dout;
  CK(hipMalloc(&dk, sizeof(h))); CK(hipMalloc(&dout, 4096 * 256 * 4));
  CK(hipMemcpy(dk, h, sizeof(h), hipMemcpyHostToDevice));
  for (int blocks : {1024, 2048, 4096}) {
    printf("blocks %d (waves per SIMD %.0f)\n", blocks, blocks * 4.0 / 1024);
    if (run<0>("cmp + count, 1 candidate", dk, dout, blocks, 2000, 32)) return 1;
    if (run<1>("cmp + count, 2 candidates", dk, dout, blocks, 2000, 64)) return 1;
    if (run<2>("8 plain vector adds per read", dk, dout, blocks, 2000, 32)) return 1;
    if (run<3>("sign of difference (31-bit keys)", dk, dout, blocks, 2000, 32)) return 1;
  }
  return 0;
}
