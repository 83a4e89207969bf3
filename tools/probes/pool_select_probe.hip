// Probe: where the time of pool_select_kernel (csrc/r3_pool.hip) goes -- s_memtime stamps of workgroup 0 at the phase
// boundaries: 0-1 histogram digits | 1-2 the pass over the keys (entries to the two lists) | the rest: meta write.  N = 4 images of L keys (argv[1], default 16384), k = 2000, detector-like score distribution.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/pool_select_probe tools/probes/pool_select_probe.hip
#include "../../r3det-pytorch_amd/csrc/r3_pool.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void hist_kernel(const unsigned* keys, int Lpad, unsigned* hist) {
  const int n = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t < Lpad && keys[(size_t)n * Lpad + t]) atomicAdd(&hist[(size_t)n * PH_BINS + (keys[(size_t)n * Lpad + t] >> 20)], 1u);
}
int main(int argc, char** argv) {
  const int N = 4, L = argc > 1 ? atoi(argv[1]) : 16384, k = 2000, Lpad = (L + 3) / 4 * 4;
  std::vector<unsigned> h((size_t)N * Lpad, 0u);
  srand(3);
  for (int n = 0; n < N; n++)
    for (int i = 0; i < L; i++) {
      float logit = -4.f + 1.5f * (float)((rand() % 20001) - 10000) / 4000.f;
      float s = 1.f / (1.f + expf(-logit));
      unsigned u; memcpy(&u, &s, 4);
      h[(size_t)n * Lpad + i] = u | 0x80000000u;
    }
  unsigned *dk, *dh; int* sel; u64* st; u64 *gl, *cl_;
  CK(hipMalloc(&dk, h.size() * 4)); CK(hipMalloc(&dh, (size_t)N * PH_BINS * 4)); CK(hipMalloc(&sel, (size_t)N * k * 4)); CK(hipMalloc(&st, 64)); CK(hipMalloc(&gl, (size_t)N * PS_KMAX * 8)); CK(hipMalloc(&cl_, (size_t)N * PS_CAND * 8));
  CK(hipMemcpy(dk, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dh, 0, (size_t)N * PH_BINS * 4));
  hipLaunchKernelGGL(hist_kernel, dim3((Lpad + 255) / 256, N), dim3(256), 0, 0, dk, Lpad, dh);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; it++) {
    CK(hipEventRecord(e0));
    PLevels P{};
    P.count = 1; P.k = k; P.kb = (k + PE_C - 1) / PE_C;
    P.lv[0].select = 1; P.lv[0].Lpad = Lpad; P.lv[0].keys = dk; P.lv[0].hist = dh; P.lv[0].glist = gl; P.lv[0].clist = cl_;
    P.lv[0].meta = sel;
    CK(hipMemset(sel, 0, (size_t)N * 16));
    hipLaunchKernelGGL(pool_select_kernel, dim3(N, 1, ((Lpad >> 2) + PS_T * PS_U - 1) / (PS_T * PS_U)), dim3(PS_T), 0, 0, P, st);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    u64 s[8]; CK(hipMemcpy(s, st, 64, hipMemcpyDeviceToHost));
    printf("L %d: launch %.1f us | hist %llu  pass %llu  digits %llu  collect %llu  sort %llu  out %llu  total %llu\n", L, ms * 1e3,
           s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[6] - s[0]);
  }
  return 0;
}
