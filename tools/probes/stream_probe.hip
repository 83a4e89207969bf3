// Probe: the streaming floor of the FeatureRefineModule tail: out = a + b + r (3 reads + 1 write per element)
// on N x 256 x 128 x 128 fp32 tensors, rotating over buffer sets beyond the Infinity Cache.  Variants: linear
// float4 grid-stride; one wave per 1 KB row with 4 rows in flight per wave (the sampler kernel's shape).
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void tri_linear(const float4* a, const float4* b, const float4* r, float4* o, size_t n4) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 x = a[i], y = b[i], z = r[i];
    o[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
  }
}
// UNR independent elements per thread per trip (more bytes in flight per wave)
template <int UNR>
__global__ __launch_bounds__(256) void tri_unrolled(const float4* a, const float4* b, const float4* r, float4* o, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * UNR) {
    float4 x[UNR], y[UNR], z[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) { size_t i = i0 + u * stride; if (i < n4) { x[u] = a[i]; y[u] = b[i]; z[u] = r[i]; } }
#pragma unroll
    for (int u = 0; u < UNR; u++) { size_t i = i0 + u * stride; if (i < n4) o[i] = make_float4(x[u].x + y[u].x + z[u].x, x[u].y + y[u].y + z[u].y, x[u].z + y[u].z + z[u].z, x[u].w + y[u].w + z[u].w); }
  }
}
// the sampler kernel's order: 4 x 4 tiles of 1 KB rows (C = 256), a wave walks the 4 positions of one row of the tile;
// EXTRA loads of a (cache-resident) row per position stand in for taps that hit
template <int EXTRA, bool DEP = false, bool BAND = false>
__global__ __launch_bounds__(256) void tri_tiled(const float4* a, const float4* b, const float4* r, float4* o, int H, int W,
                                                 int tiles_x, int tiles_per_img, const float* boxes = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned t = blockIdx.x;
  if (BAND) t = (t & 7u) * (gridDim.x >> 3) + (t >> 3);
  const int n = t / tiles_per_img, tt = t % tiles_per_img;
  const int ty = tt / tiles_x, tx = tt % tiles_x;
  const size_t img = (size_t)n * H * W;
  const int h = ty * 4 + wave;
  float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const size_t q = (img + (size_t)h * W + tx * 4 + i) * 64 + lane;
    int dy = 0, dx = 0;
    if (DEP) {  // the taps depend on a per-position record (two floats), as the sampler's do on the box
      const float* bp = boxes + (img + (size_t)h * W + tx * 4 + i) * 5;
      dy = __builtin_amdgcn_readfirstlane((int)(bp[0] * 0.125f)) - (tx * 4 + i);   // = 0 for the regular field
      dx = __builtin_amdgcn_readfirstlane((int)(bp[1] * 0.125f)) - h;
    }
    float4 x = a[q], y = b[q], z = r[q];
#pragma unroll
    for (int e = 0; e < EXTRA; e++) {
      const float4 t = (e & 1 ? b : a)[(img + (size_t)(tx * 4 + (e >> 1 & 1) + dy) * W + ty * 4 + (e >> 2) + dx) * 64 + lane];  // transposed tile
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    o[q] = make_float4(x.x + y.x + z.x + acc.x, x.y + y.y + z.y + acc.y, x.z + y.z + z.z + acc.z, x.w + y.w + z.w + acc.w);
  }
}
__global__ __launch_bounds__(256) void copy_linear(const float4* a, float4* o, size_t n4) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) o[i] = a[i];
}

int main() {
  const size_t n = (size_t)4 * 256 * 128 * 128, bytes = n * 4, n4 = n / 4;
  float* buf[3][4];
  for (int s = 0; s < 3; s++) for (int k = 0; k < 4; k++) { CK(hipMalloc(&buf[s][k], bytes)); CK(hipMemset(buf[s][k], 0, bytes)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double moved, auto launch) -> int {
    for (int i = 0; i < 3; i++) launch(i % 3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; i++) launch(i % 3);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.2f us  %6.2f TB/s\n", name, ms * 1000 / reps, moved / (ms / reps * 1e-3) / 1e12);
    return 0;
  };
  for (int g : {1024, 2048, 4096, 8192, 16384})  {
    char nm[64]; snprintf(nm, 64, "a+b+r linear, grid %d", g);
    run(nm, 4.0 * bytes, [&](int s) { tri_linear<<<g, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], n4); });
  }
  run("a+b+r 2 per thread, grid 2048", 4.0 * bytes, [&](int s) { tri_unrolled<2><<<2048, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], n4); });
  run("a+b+r 4 per thread, grid 2048", 4.0 * bytes, [&](int s) { tri_unrolled<4><<<2048, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], n4); });
  run("a+b+r 4 per thread, grid 1024", 4.0 * bytes, [&](int s) { tri_unrolled<4><<<1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], n4); });
  run("tiled 4x4, no extra loads", 4.0 * bytes, [&](int s) { tri_tiled<0><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024); });
  run("tiled 4x4, 4 extra loads", 4.0 * bytes, [&](int s) { tri_tiled<4><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024); });
  run("tiled 4x4, 8 extra loads", 4.0 * bytes, [&](int s) { tri_tiled<8><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024); });
  float* boxes; CK(hipMalloc(&boxes, (size_t)4 * 128 * 128 * 5 * 4));
  { float* hb = (float*)malloc((size_t)4 * 128 * 128 * 5 * 4);
    for (int n = 0; n < 4; n++) for (int h = 0; h < 128; h++) for (int w = 0; w < 128; w++) { float* q = hb + (((size_t)n * 128 + h) * 128 + w) * 5; q[0] = (w + 0.5f) * 8; q[1] = (h + 0.5f) * 8; q[2] = 30; q[3] = 10; q[4] = 0; }
    CK(hipMemcpy(boxes, hb, (size_t)4 * 128 * 128 * 5 * 4, hipMemcpyHostToDevice)); free(hb); }
  run("tiled 4x4, 8 extra, band remap", 4.0 * bytes, [&](int s) { tri_tiled<8, false, true><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024); });
  run("tiled 4x4, 8 extra, dependent", 4.0 * bytes, [&](int s) { tri_tiled<8, true, false><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024, boxes); });
  run("tiled 4x4, 8 extra, dep + band", 4.0 * bytes, [&](int s) { tri_tiled<8, true, true><<<4 * 1024, 256>>>((float4*)buf[s][0], (float4*)buf[s][1], (float4*)buf[s][2], (float4*)buf[s][3], 128, 128, 32, 1024, boxes); });
  run("copy linear, grid 4096", 2.0 * bytes, [&](int s) { copy_linear<<<4096, 256>>>((float4*)buf[s][0], (float4*)buf[s][3], n4); });
  run("copy linear, grid 16384", 2.0 * bytes, [&](int s) { copy_linear<<<16384, 256>>>((float4*)buf[s][0], (float4*)buf[s][3], n4); });
  return 0;
}
