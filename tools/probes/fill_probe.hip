// Probe: how fast can 128 x 196416 fp32 zeros be written, as a function of the tile shape a workgroup owns?
// (The IoU stream kernel writes 16 x 1024 tiles; a plain fill of the same 100 MB is twice as fast.)
// build: hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int R = 128, N2 = 196416;

__global__ __launch_bounds__(256) void fill_linear(float4* out, size_t n4) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(0, 0, 0, 0);
}
// workgroup owns a contiguous chunk of `per` float4 (like the runtime's fill kernel)
__global__ __launch_bounds__(256) void fill_chunk(float4* out, size_t n4, int per) {
  size_t b = (size_t)blockIdx.x * per;
  for (int i = threadIdx.x; i < per && b + i < n4; i += 256) out[b + i] = make_float4(0, 0, 0, 0);
}
// tile ROWS x (256 * 4 * CG) columns; lane owns 4 adjacent columns in each of CG column groups
template <int ROWS, int CG, bool SWAP>
__global__ __launch_bounds__(256) void fill_tile(float* out, int n2) {
  const int bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
  const int row0 = by * ROWS;
#pragma unroll 1
  for (int r = 0; r < ROWS; r++) {
#pragma unroll
    for (int g = 0; g < CG; g++) {
      const int col = (bx * CG + g) * 1024 + threadIdx.x * 4;
      if (col + 4 <= n2) *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * n2 + col) = make_float4(0, 0, 0, 0);
    }
  }
}
// like fill_tile<16,1> but a wave owns 4 rows x 1024 columns (lane: 16 adjacent columns = 64 B)
__global__ __launch_bounds__(256) void fill_tile_wave_rows(float* out, int n2) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = blockIdx.y * 16 + wave * 4;
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int col = blockIdx.x * 1024 + k * 256 + lane * 4;
      if (col + 4 <= n2) *reinterpret_cast<float4*>(out + (size_t)(row0 + r) * n2 + col) = make_float4(0, 0, 0, 0);
    }
}

int main() {
  const size_t bytes = (size_t)R * N2 * 4;
  float* buf[3];
  for (int i = 0; i < 3; i++) CK(hipMalloc(&buf[i], bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ct = (N2 + 1023) / 1024;
  for (int rot = 1; rot <= 3; rot += 2) {
    auto run = [&](const char* name, auto launch) -> int {
      for (int i = 0; i < 3; i++) launch(buf[i % rot]);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int reps = 30;
      for (int i = 0; i < reps; i++) launch(buf[i % rot]);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rot=%d %-28s %7.2f us  %6.2f TB/s\n", rot, name, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e12);
      return 0;
    };
    run("hipMemsetAsync", [&](float* b) { (void)hipMemsetAsync(b, 0, bytes, 0); });
    run("linear grid 2048", [&](float* b) { fill_linear<<<2048, 256>>>((float4*)b, bytes / 16); });
    run("linear grid 8192", [&](float* b) { fill_linear<<<8192, 256>>>((float4*)b, bytes / 16); });
    run("chunk 64 KB", [&](float* b) { fill_chunk<<<(bytes / 16 + 4095) / 4096, 256>>>((float4*)b, bytes / 16, 4096); });
    run("chunk 16 KB", [&](float* b) { fill_chunk<<<(bytes / 16 + 1023) / 1024, 256>>>((float4*)b, bytes / 16, 1024); });
    run("chunk 4 KB", [&](float* b) { fill_chunk<<<(bytes / 16 + 255) / 256, 256>>>((float4*)b, bytes / 16, 256); });
    run("tile 16x1024", [&](float* b) { fill_tile<16, 1, false><<<dim3(ct, R / 16), 256>>>(b, N2); });
    run("tile 16x1024 swapped grid", [&](float* b) { fill_tile<16, 1, true><<<dim3(R / 16, ct), 256>>>(b, N2); });
    run("tile 8x1024", [&](float* b) { fill_tile<8, 1, false><<<dim3(ct, R / 8), 256>>>(b, N2); });
    run("tile 4x1024", [&](float* b) { fill_tile<4, 1, false><<<dim3(ct, R / 4), 256>>>(b, N2); });
    run("tile 1x1024", [&](float* b) { fill_tile<1, 1, false><<<dim3(ct, R), 256>>>(b, N2); });
    run("tile 32x1024", [&](float* b) { fill_tile<32, 1, false><<<dim3(ct, R / 32), 256>>>(b, N2); });
    run("tile 128x1024", [&](float* b) { fill_tile<128, 1, false><<<dim3(ct, 1), 256>>>(b, N2); });
    run("tile 4x4096", [&](float* b) { fill_tile<4, 4, false><<<dim3((ct + 3) / 4, R / 4), 256>>>(b, N2); });
    run("tile 16x4096", [&](float* b) { fill_tile<16, 4, false><<<dim3((ct + 3) / 4, R / 16), 256>>>(b, N2); });
    run("tile 2x8192", [&](float* b) { fill_tile<2, 8, false><<<dim3((ct + 7) / 8, R / 2), 256>>>(b, N2); });
    run("tile 16x1024 wave=4 rows", [&](float* b) { fill_tile_wave_rows<<<dim3(ct, R / 16), 256>>>(b, N2); });
  }
  return 0;
}
