// Probe: starting from the LDS copy (lds_copy_probe.hip: device-copy rate), add the pieces of the
// FR forward kernel one at a time to see which one costs the time.
//   V0 copy through LDS (pitch 129)                         V1 + 80-byte per-quad table loads (unused)
//   V2 + 16 transposed LDS gathers + bilinear math          V3 = V2 with conflict-free lane mapping
//   V4 = V2 without the table loads (taps synthesised)
// build: hipcc --offload-arch=gfx950 -O3 -o fr_stage_probe fr_stage_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float4* __restrict__ in, const float4* __restrict__ table,
                                              float4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PER = 4, T = 1024, W = 128, P = 129;
  const float4* src = in + (size_t)blockIdx.x * 4096;
  float4* dst = out + (size_t)blockIdx.x * 4096;
  const float4* tb = table + (size_t)(blockIdx.x >> 8) * 4096 * 5;
  float4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) v[k] = src[threadIdx.x + k * T];
  float4 tq[5];
  if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
    for (int q = 0; q < 5; q++) tq[q] = tb[threadIdx.x * 5 + q];
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    int e = (threadIdx.x + k * T) * 4, y = e >> 7, x = e & 127;
    float* d = lds + y * P + x;
    d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    int qd = threadIdx.x + k * T;
    int e = qd * 4, y = e >> 7, x = e & 127;
    float r[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
    float t0 = 0.f;
    if (MODE == 1 || MODE == 2 || MODE == 3) {
      t0 = tq[0].x + tq[1].y + tq[2].z + tq[3].w + tq[4].x;
      if (k + 1 < PER) {
#pragma unroll
        for (int q = 0; q < 5; q++) tq[q] = tb[(qd + T) * 5 + q];
      }
    }
    if (MODE >= 2) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // transposed sample cell: row <- column index of the position, column <- its row
        int px = (MODE == 3) ? ((x + j * 32 + (threadIdx.x & 31) * 0) & 127) : (x + j);
        int row = min(px, 126), col = min(y, 126);
        if (MODE == 3) { row = (threadIdx.x & 63) + 64 * (j & 1); col = min(y + (j >> 1), 126); row = min(row, 126); }
        const float* p = lds + row * P + col;
        float w = 0.25f + t0 * 1e-30f;
        r[j] += (w * p[0] + w * p[1] + w * p[P] + w * p[P + 1]);
      }
    } else if (MODE == 1) {
      r[0] += t0 * 1e-30f;
    }
    dst[qd] = make_float4(r[0], r[1], r[2], r[3]);
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
  const size_t n = (size_t)4 * 256 * 128 * 128;
  float *in, *out, *table;
  hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&table, (size_t)4 * 16384 * 20);
  hipMemset(in, 0, n * 4); hipMemset(table, 0, (size_t)4 * 16384 * 20);
#define RUN(M, NAME) { hipFuncSetAttribute((const void*)probe<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    float us = time_us([&] { hipLaunchKernelGGL(probe<M>, dim3(1024), dim3(1024), 129 * 128 * 4 + 1024, 0, (const float4*)in, (const float4*)table, (float4*)out); }, 20); \
    printf("%-64s %7.1f us  %7.1f GB/s\n", NAME, us, 2.0 * n * 4 / us / 1e3); }
  RUN(0, "V0 copy through LDS (pitch 129, 1024 thr)");
  RUN(1, "V1 + 80 B/quad table loads, prefetched before the store");
  RUN(2, "V2 + 16 transposed LDS gathers/quad (lane owns 4 adjacent: 4-way conflicts)");
  RUN(3, "V3 V2 with rows consecutive across lanes (conflict-free)");
  RUN(4, "V4 V2 without the table loads");
  return 0;
}
