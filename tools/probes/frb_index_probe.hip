// Probe: where the time of frb_index_sort_kernel (csrc/r3_frb.hip) goes -- s_memtime stamps of one workgroup at
// the phase boundaries (A scan of the sample rows | B taps of the reaching sources | C bitonic sort | D write),
// level 0 of a 1024^2 input (N = 4, 128 x 128), regular box field.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/frb_index_probe tools/probes/frb_index_probe.hip
#include "../../r3det-pytorch_amd/csrc/r3_frb.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int g_r3_fr_walk = 8;
int main() {
  const int N = 4, H = 128, W = 128, HW = H * W;
  std::vector<float> b((size_t)N * HW * 5);
  srand(1);
  for (int n = 0; n < N; n++)
    for (int p = 0; p < HW; p++) {
      float* q = &b[((size_t)n * HW + p) * 5];
      q[0] = (p % W) * 8.f + (rand() % 1000 - 500) * 0.006f;
      q[1] = (p / W) * 8.f + (rand() % 1000 - 500) * 0.006f;
      q[2] = 30; q[3] = 20; q[4] = -0.3f;
    }
  float* db; void* ws; u64* st;
  const size_t need = r3k_frn_workspace_bytes(N, H, W, 1);
  CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&ws, need)); CK(hipMalloc(&st, 64));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  allow_big_lds(frb_index_sort_kernel<true>, (int)sizeof(IxsLds));
  allow_big_lds(frb_index_sort_kernel<false>, (int)sizeof(IxsLds));
  const FrnLayout LN = frn_layout(ws, N, 256, H, W, 1);
  FrbSellOut so{LN.slicehdr, reinterpret_cast<int2*>(LN.sell), 4 * LN.cp, LN.cap, LN.pitch, LN.slices};
  const FrbLayout L = frb_layout(ws, N, H, W, 1);
  const int R = sort_band_rows(H, W);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 6; it++) {
    CK(hipEventRecord(e0));
    if (it & 1)
      hipLaunchKernelGGL(frb_index_sort_kernel<false>, dim3((H + R - 1) / R, N), dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R,
                         L.cellinfo, L.entries, so, st);
    else
      hipLaunchKernelGGL(frb_index_sort_kernel<true>, dim3((H + R - 1) / R, N), dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R,
                         L.cellinfo, L.entries, FrbSellOut{nullptr, nullptr, 0, 0, 0, 0}, st);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    u64 h[8]; CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
    printf("launch %.1f us | cycles(100MHz?) A %llu B %llu C %llu D %llu total %llu | entries %llu sources %llu | lds %zu B\n", ms * 1e3,
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[4] - h[0], h[5], h[6], sizeof(IxsLds));
  }
  return 0;
}
