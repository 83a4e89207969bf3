// Probe: what the runtime says about the residency of the channels_last FR backward gather (512 threads, 32 KB LDS).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/frb_occ_probe tools/probes/frb_occ_probe.hip
#include "../../r3det-pytorch_amd/csrc/r3_frb.hip"
#include <cstdio>
int g_r3_fr_walk = 8;
int main() {
  int b = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, frb_gather_kernel<false, true>, 512, 0);
  printf("frb_gather_kernel<false, true>, 512 threads: %s, %d workgroups per CU\n", hipGetErrorString(e), b);
  hipFuncAttributes a;
  e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(frb_gather_kernel<false, true>));
  printf("attributes: %s  regs %d  shared %zu  local %zu  maxThreads %d\n", hipGetErrorString(e), a.numRegs, a.sharedSizeBytes,
         a.localSizeBytes, a.maxThreadsPerBlock);
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, frb_gather_kernel<false, false>, 256, 0);
  printf("frb_gather_kernel<false, false>, 256 threads: %s, %d workgroups per CU\n", hipGetErrorString(e), b);
  return 0;
}
