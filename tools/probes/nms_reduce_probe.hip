// Probe: where the time of nms_reduce_groups_kernel (csrc/r3_nms.hip) goes -- s_memtime stamps of reducer workgroup
// (group 0, image 0) at the phase boundaries (label compaction | prologue loads | rounds | keep bits), on a
// clustered pool like tools/nms_prof.py's (n boxes around n / 12 objects, 15 labels), through r3k_batched_nms.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I r3det-pytorch_amd/csrc -I include -o tools/probes/nms_reduce_probe tools/probes/nms_reduce_probe.hip
#include "../../r3det-pytorch_amd/csrc/r3_nms.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
R3Option g_r3_nms_impl{0}, g_r3_nms_qcap{0}, g_r3_clip_impl{0};
int r3_cu_count() { return 256; }
namespace {
__global__ void probe_zero_kernel(unsigned* p, size_t words) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace
int r3k_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes) hipLaunchKernelGGL(probe_zero_kernel, dim3(1024), dim3(256), 0, stream, (unsigned*)p, bytes / 4);
  return 0;
}
static float urand() { return (float)(rand() % 1000003) / 1000003.f; }
static float nrand() { return sqrtf(-2.f * logf(urand() + 1e-7f)) * cosf(6.2831853f * urand()); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8576;
  const int nobj = n / 12 > 0 ? n / 12 : 1;
  srand(3);
  std::vector<float> obj(nobj * 5); std::vector<int> ocls(nobj);
  for (int o = 0; o < nobj; o++) {
    float w = 10.f + 140.f * urand(), asp = 1.f + 3.f * urand();
    obj[o * 5] = 1024.f * urand(); obj[o * 5 + 1] = 1024.f * urand(); obj[o * 5 + 2] = w; obj[o * 5 + 3] = w / asp;
    obj[o * 5 + 4] = -1.5708f * urand(); ocls[o] = rand() % 15;
  }
  std::vector<float> b(n * 5), sc(n); std::vector<long long> lab(n);
  for (int i = 0; i < n; i++) {
    const int o = rand() % nobj;
    const float m = fminf(obj[o * 5 + 2], obj[o * 5 + 3]);
    b[i * 5] = obj[o * 5] + nrand() * 0.15f * m; b[i * 5 + 1] = obj[o * 5 + 1] + nrand() * 0.15f * m;
    b[i * 5 + 2] = obj[o * 5 + 2] * expf(nrand() * 0.1f); b[i * 5 + 3] = obj[o * 5 + 3] * expf(nrand() * 0.1f);
    b[i * 5 + 4] = obj[o * 5 + 4] + nrand() * 0.05f; sc[i] = 0.05f + 0.95f * urand(); lab[i] = ocls[o];
  }
  if (argc > 2 && argv[2][0]) {  // a dumped pool (tools/cross_label_edges.py, DUMP_POOL=file): n | boxes | scores | labels, all f32
    FILE* f = fopen(argv[2], "rb");
    if (!f) { printf("cannot open %s\n", argv[2]); return 1; }
    float fn; if (fread(&fn, 4, 1, f) != 1) return 1;
    const int m = (int)fn;
    if (m != n) { printf("file holds %d boxes: pass that number as the first argument\n", m); return 1; }
    std::vector<float> lf(n);
    if (fread(b.data(), 4, (size_t)n * 5, f) != (size_t)n * 5 || fread(sc.data(), 4, n, f) != (size_t)n || fread(lf.data(), 4, n, f) != (size_t)n) return 1;
    for (int i = 0; i < n; i++) lab[i] = (long long)lf[i];
    fclose(f);
  }
  float *db, *ds, *dd; long long *dl, *dk; int* dc; void* ws; u64* st;
  const size_t wsb = r3k_batched_rnms_workspace_bytes(n);
  CK(hipMalloc(&db, n * 20)); CK(hipMalloc(&ds, n * 4)); CK(hipMalloc(&dl, n * 8)); CK(hipMalloc(&dd, n * 24)); CK(hipMalloc(&dk, n * 8));
  CK(hipMalloc(&dc, 4)); CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&st, 512)); CK(hipMemset(st, 0, 512));
  CK(hipMemcpy(db, b.data(), n * 20, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sc.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dl, lab.data(), n * 8, hipMemcpyHostToDevice));
  g_nms_stamps = st;
  { const u64 grp = argc > 3 ? (u64)atoi(argv[3]) : 0; CK(hipMemcpy(st + 15, &grp, 8, hipMemcpyHostToDevice)); }  // the stamped group
  for (int it = 0; it < 3; it++) {
    const int rc = r3k_batched_nms(1, db, ds, (const int64_t*)dl, n, 0.1f, ws, wsb, dd, (int64_t*)dk, dc, 0);
    CK(hipDeviceSynchronize());
    u64 s[64]; int kept; CK(hipMemcpy(s, st, 512, hipMemcpyDeviceToHost));
    if (it == 2) for (int r = 0; r < 8 && s[16 + 4 * r]; r++)
      printf("  round %d: stamps +%llu +%llu +%llu  (in-register form: its rows | barrier (= the slowest wavefront) + the handed-over rows' pass | -; general form: pass A | "
             "passes B, C | overflow rows + barriers; this workgroup's thread 0)\n", r, s[17 + 4 * r] > s[16 + 4 * r] ? s[17 + 4 * r] - s[16 + 4 * r] : 0,
             s[18 + 4 * r] - (s[17 + 4 * r] > s[16 + 4 * r] ? s[17 + 4 * r] : s[16 + 4 * r]), s[19 + 4 * r] - s[18 + 4 * r]); CK(hipMemcpy(&kept, dc, 4, hipMemcpyDeviceToHost));
    if (s[52] > s[48]) printf("  ranking kernel (sorted chunks), workgroup 0: record built %llu  first group of chunks %llu  the other groups %llu  records out %llu\n",
                              s[49] - s[48], s[50] - s[49], s[51] - s[50], s[52] - s[51]);
    if (s[56] > s[53]) printf("    its third group: next keys to LDS + request %llu  two searches %llu  barrier %llu\n", s[54] - s[53], s[55] - s[54], s[56] - s[55]);
    printf("n %d rc %d kept %d | cycles: compaction %llu  prologue %llu  rounds %llu (%llu rounds)  tail %llu  bits %llu  total %llu\n", n, rc,
           kept, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[7], s[4] - s[3], s[5] - s[4], s[5] - s[0]);
  }
  return 0;
}
