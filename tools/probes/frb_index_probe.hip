// Probe: where the time of frb_index_sort_kernel (csrc/r3_frb.hip) goes -- s_memtime stamps of one workgroup at
// the phase boundaries (A scan of the sample rows | B taps of the reaching sources | C bitonic sort | D write),
// level 0 of a 1024^2 input (N = 4, 128 x 128), regular box field.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/frb_index_probe tools/probes/frb_index_probe.hip
#include "../../r3det-pytorch_amd/csrc/r3_frb.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
R3Option g_r3_fr_walk{8};
static void host_cell_tap(float y, float x, int H, int W, float& ty, float& tx) {
  if (y < -1.0 || y > H || x < -1.0 || x > W) { ty = (float)(H + 1); tx = 0.f; return; }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  if ((int)y >= H - 1) y = (float)(H - 1);
  if ((int)x >= W - 1) x = (float)(W - 1);
  ty = y; tx = x;
}
// usage: frb_index_probe [piles]   -- round 6: every form from the box records and from the level's tap table (TAB),
// stamps of one workgroup and a byte comparison of the two workspaces
int main(int argc, char** argv) {
  const bool piles = argc > 1;
  const int N = 4, H = 128, W = 128, HW = H * W;
  std::vector<float> b((size_t)N * HW * 5), tab((size_t)N * HW * 2);
  srand(1);
  for (int n = 0; n < N; n++)
    for (int p = 0; p < HW; p++) {
      float* q = &b[((size_t)n * HW + p) * 5];
      q[0] = (p % W) * 8.f + (rand() % 1000 - 500) * 0.006f;
      q[1] = (p / W) * 8.f + (rand() % 1000 - 500) * 0.006f;
      if (piles) {
        q[0] = ((p % W) / 4) * 32.f + 16.f + (rand() % 1000 - 500) * 0.005f;
        q[1] = ((p / W) / 4) * 32.f + 16.f + (rand() % 1000 - 500) * 0.005f;
      }
      q[2] = 30; q[3] = 20; q[4] = -0.3f;
      host_cell_tap(q[0] * 0.125f, q[1] * 0.125f, H, W, tab[(size_t)n * 2 * HW + p], tab[(size_t)n * 2 * HW + HW + p]);
    }
  float *db, *dt; void *ws, *ws2; u64* st;
  const size_t need = r3k_frn_workspace_bytes(N, H, W, 1);
  CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dt, tab.size() * 4)); CK(hipMalloc(&ws, need)); CK(hipMalloc(&ws2, need));
  CK(hipMalloc(&st, 64));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  allow_big_lds(frb_index_sort_kernel<true>, (int)sizeof(IxsLds));
  allow_big_lds(frb_index_sort_kernel<false>, (int)sizeof(IxsLds));
  allow_big_lds((frb_index_sort_kernel<true, true>), (int)sizeof(IxsLds));
  allow_big_lds((frb_index_sort_kernel<false, true>), (int)sizeof(IxsLds));
  const int R = sort_band_rows(H, W);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned char> h1(need), h2(need);
  for (int it = 0; it < 12; it++) {
    const bool sell = it & 1, tb = it & 2;
    void* w = tb ? ws2 : ws;
    CK(hipMemset(w, 0, need));
    const FrnLayout LN = frn_layout(w, N, 256, H, W, 1);
    FrbSellOut so{LN.slicehdr, reinterpret_cast<int2*>(LN.sell), 4 * LN.cp, LN.cap, LN.pitch, LN.slices};
    const FrbSellOut none{nullptr, nullptr, 0, 0, 0, 0};
    const FrbLayout L = frb_layout(w, N, H, W, 1);
    const dim3 grid((H + R - 1) / R, N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (sell && tb) hipLaunchKernelGGL((frb_index_sort_kernel<false, true>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L.cellinfo, L.entries, so, st, dt);
    else if (sell) hipLaunchKernelGGL((frb_index_sort_kernel<false, false>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L.cellinfo, L.entries, so, st, (const float*)nullptr);
    else if (tb) hipLaunchKernelGGL((frb_index_sort_kernel<true, true>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L.cellinfo, L.entries, none, st, dt);
    else hipLaunchKernelGGL((frb_index_sort_kernel<true, false>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L.cellinfo, L.entries, none, st, (const float*)nullptr);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    u64 h[8]; CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
    printf("%s %s launch %5.1f us | clocks A %5llu B %5llu C %5llu D %5llu total %6llu | entries %llu sources %llu\n", sell ? "SELL" : "CSR ",
           tb ? "TAB  " : "boxes", ms * 1e3, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[4] - h[0], h[5], h[6]);
    if (tb) {  // the same form from the boxes ran two launches ago into ws
      CK(hipMemcpy(h2.data(), ws2, need, hipMemcpyDeviceToHost));
      // (re-run the box form into ws for the comparison)
      CK(hipMemset(ws, 0, need));
      const FrnLayout LN1 = frn_layout(ws, N, 256, H, W, 1);
      FrbSellOut so1{LN1.slicehdr, reinterpret_cast<int2*>(LN1.sell), 4 * LN1.cp, LN1.cap, LN1.pitch, LN1.slices};
      const FrbLayout L1 = frb_layout(ws, N, H, W, 1);
      if (sell) hipLaunchKernelGGL((frb_index_sort_kernel<false, false>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L1.cellinfo, L1.entries, so1, st, (const float*)nullptr);
      else hipLaunchKernelGGL((frb_index_sort_kernel<true, false>), grid, dim3(IX_T), sizeof(IxsLds), 0, db, 0.125f, H, W, R, L1.cellinfo, L1.entries, none, st, (const float*)nullptr);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(), ws, need, hipMemcpyDeviceToHost));
      size_t diff = 0, first = need;
      for (size_t i = 0; i < need; i++) if (h1[i] != h2[i]) { diff++; if (first == need) first = i; }
      const size_t ci_bytes = (size_t)((char*)L1.entries - (char*)ws);
      printf("     boxes vs TAB: %zu differing bytes of %zu, first at %zu (cellinfo ends at %zu)\n", diff, need, first, ci_bytes);
      if (diff && first < ci_bytes) {
        const int2* c1 = reinterpret_cast<const int2*>(h1.data()); const int2* c2 = reinterpret_cast<const int2*>(h2.data());
        const size_t k = first / 8;
        printf("     cell %zu (image %zu, row %zu, col %zu): boxes {%d, %d} TAB {%d, %d}\n", k, k / HW, (k % HW) / W, k % W, c1[k].x, c1[k].y, c2[k].x, c2[k].y);
      }
    }
  }
  return 0;
}
