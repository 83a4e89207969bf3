// r3_nms.hip -- rotated NMS on gfx950: wavefront-bitmask suppression over score-sorted
// boxes with the greedy reduction done on the device.
//
// Replaces nmsr_kernel + host scan (rnms/src/rcuda/rnms_kernel.cu:229-335),
// nms_rotated_cuda_kernel + host scan (nms_rotated/src/nms_rotated_cuda.cu:13-134) and the
// ml variant (ml_nms_rotated/src/nms_rotated_cuda.cu:14-137).  Differences by design:
//   * boxes are gathered through `order` and turned into prepared records once (trig per
//     box, not per pair);
//   * only tiles with col_block >= row_block are computed (the reference computes both
//     triangles, nms_rotated_cuda.cu:23);
//   * the n x ceil(n/64) bitmask never leaves the device: one workgroup walks the rows in
//     score order (wave 0 resolves a 64-row block on its diagonal word with scalar ops, the
//     other waves OR the surviving rows into the running `removed` words held in LDS) and
//     emits the keep list and its length.  The reference copies the whole mask to the host
//     (9.2 MB at n = 8576) and scans it there.
#include <hip/hip_runtime.h>

#include "r3_geom.h"
#include "r3_kernels.h"

namespace {

typedef unsigned long long u64;

constexpr int TILE = 64;        // one wavefront = one 64-wide bitmask word
constexpr int MASK_WAVES = 4;   // tiles per workgroup of the mask kernel

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int GEOM>
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* __restrict__ dets,
                                                          int det_stride,
                                                          const int64_t* __restrict__ labels,
                                                          const int64_t* __restrict__ order, int n,
                                                          BoxRec* __restrict__ recs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t src = order[i];
  float lab = labels ? (float)labels[src] : 0.f;  // at::cat({dets, labels}) promotes to float
  BoxRec r;
  make_record<GEOM>(dets + (size_t)src * det_stride, lab, r);
  recs[i] = r;
}

// mask[row * cb + c] bit i  <=>  IoU(box_row, box_{64c+i}) > thr, for c >= row / 64.
template <int GEOM, bool LABEL>
__global__ __launch_bounds__(TILE* MASK_WAVES) void nms_mask_kernel(const BoxRec* __restrict__ recs,
                                                                    int n, int cb, float thr,
                                                                    u64* __restrict__ mask) {
  __shared__ BoxRec cols[MASK_WAVES][TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.y;
  const int cblk = blockIdx.x * MASK_WAVES + wave;
  const bool active = (cblk < cb) && (cblk >= rb);
  int col_size = 0;
  if (active) {
    col_size = min(n - cblk * TILE, TILE);
    if (lane < col_size) cols[wave][lane] = recs[cblk * TILE + lane];
  }
  __syncthreads();
  if (!active) return;
  const int row = rb * TILE + lane;
  if (row >= n) return;
  BoxRec A = recs[row];
  u64 t = 0;
  int start = (rb == cblk) ? lane + 1 : 0;
  // thr < 0 would make IoU == 0 suppress; the circle shortcut is only valid for thr >= 0
  const bool shortcut = thr >= 0.f;
  for (int i = start; i < col_size; i++) {
    const BoxRec& B = cols[wave][i];
    if (GEOM != 1 && LABEL && A.f[7] != B.f[7]) {
      if (0.f > thr) t |= 1ULL << i;
      continue;
    }
    if (shortcut && circles_apart(A.f[9], A.f[10], A.f[11], B.f[9], B.f[10], B.f[11])) continue;
    BoxRec b = B;
    float v;
    if (GEOM == 1) v = v1_pair_slow(A, b, false);
    else if (GEOM == 2) v = hull_pair_slow<true>(A, b, true);
    else v = hull_pair_slow<false>(A, b, true);
    if (v > thr) t |= 1ULL << i;
  }
  mask[(size_t)row * cb + cblk] = t;
}

__device__ __forceinline__ u64 readlane64(u64 v, int k) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffULL), k);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), k);
  return ((u64)hi << 32) | lo;
}

// Greedy scan of the bitmask (host loops rnms_kernel.cu:316-327, nms_rotated_cuda.cu:117-128)
// by ONE workgroup.  remv[] (ceil(n/64) words) lives in dynamic LDS.
__global__ __launch_bounds__(1024) void nms_reduce_kernel(const u64* __restrict__ mask, int n,
                                                          int cb, const int64_t* __restrict__ order,
                                                          int64_t* __restrict__ keep_out,
                                                          int32_t* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) u64 remv[];  // cb words + 1 (kept-bits slot)
  u64* kb_slot = remv + cb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int j = tid; j < cb; j += blockDim.x) remv[j] = 0;
  __syncthreads();
  int cnt = 0;
  for (int b = 0; b < cb; b++) {
    if (wave == 0) {
      const int row = b * TILE + lane;
      u64 diag = (row < n) ? mask[(size_t)row * cb + b] : 0ULL;
      u64 cur = readlane64(remv[b], 0);
      const int nvalid = min(TILE, n - b * TILE);
      u64 kb = 0;
      for (int k = 0; k < nvalid; k++) {
        u64 dk = readlane64(diag, k);
        if (!((cur >> k) & 1ULL)) {
          kb |= 1ULL << k;
          cur |= dk;
        }
      }
      if ((kb >> lane) & 1ULL) {
        int pos = cnt + __popcll(kb & ((1ULL << lane) - 1ULL));
        keep_out[pos] = order[row];
      }
      cnt += __popcll(kb);
      if (lane == 0) *kb_slot = kb;
    }
    __syncthreads();
    const u64 kb = *kb_slot;
    const int nwords = cb - (b + 1);
    if (nwords > 0) {
      int idx = 0;
      u64 rem = kb;
      while (rem) {
        int k = __ffsll((long long)rem) - 1;
        rem &= rem - 1;
        if ((idx % nw) == wave) {
          const u64* p = mask + (size_t)(b * TILE + k) * cb + (b + 1);
          for (int j = lane; j < nwords; j += 64) {
            u64 v = p[j];
            if (v) atomicOr(&remv[b + 1 + j], v);
          }
        }
        idx++;
      }
    }
    __syncthreads();
  }
  if (tid == 0) *count_out = cnt;
}

// rnms returns keep sorted by original index (rnms_kernel.cu:331-334): mark kept originals,
// then an ordered compaction by one workgroup.
__global__ __launch_bounds__(1024) void nms_ascending_kernel(int n, uint8_t* __restrict__ flags,
                                                             int64_t* __restrict__ keep_out,
                                                             const int32_t* __restrict__ count) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int cnt = *count;
  for (int i = tid; i < n; i += blockDim.x) flags[i] = 0;
  __syncthreads();
  for (int i = tid; i < cnt; i += blockDim.x) flags[keep_out[i]] = 1;
  __syncthreads();
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int c = 0;
  for (int i = lo; i < hi; i++) c += flags[i];
  part[tid] = c;
  __syncthreads();
  // exclusive scan over 1024 partials (Hillis-Steele in LDS)
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - c;
  for (int i = lo; i < hi; i++)
    if (flags[i]) keep_out[pos++] = i;
}

template <int GEOM, bool LABEL>
int launch_nms(const float* dets, int det_stride, const int64_t* labels, const int64_t* order, int n,
               float thr, BoxRec* recs, u64* mask, int cb, hipStream_t stream) {
  hipLaunchKernelGGL((nms_prepare_kernel<GEOM>), dim3((n + 255) / 256), dim3(256), 0, stream, dets,
                     det_stride, labels, order, n, recs);
  dim3 grid((cb + MASK_WAVES - 1) / MASK_WAVES, cb);
  hipLaunchKernelGGL((nms_mask_kernel<GEOM, LABEL>), grid, dim3(TILE * MASK_WAVES), 0, stream, recs,
                     n, cb, thr, mask);
  return 0;
}

}  // namespace

size_t r3k_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  size_t cb = (n + TILE - 1) / TILE;
  return align256((size_t)n * sizeof(BoxRec)) + align256((size_t)n * cb * sizeof(u64)) +
         align256((size_t)n) + 256;
}

int r3k_nms(int geom, const float* dets, int det_stride, const int64_t* labels,
            const int64_t* order, int n, float thr, int sort_ascending, void* ws, size_t ws_bytes,
            int64_t* keep_out, int32_t* count_out, hipStream_t stream) {
  if (n < 0 || !count_out) return -1;
  if (n == 0) {
    return hipMemsetAsync(count_out, 0, sizeof(int32_t), stream) == hipSuccess ? 0 : -2;
  }
  if (!dets || !order || !ws || !keep_out) return -1;
  if (ws_bytes < r3k_nms_workspace_bytes(n)) return -3;
  const int cb = (n + TILE - 1) / TILE;
  char* p = (char*)ws;
  BoxRec* recs = (BoxRec*)p;
  p += align256((size_t)n * sizeof(BoxRec));
  u64* mask = (u64*)p;
  p += align256((size_t)n * cb * sizeof(u64));
  uint8_t* flags = (uint8_t*)p;

  if (geom == 1) launch_nms<1, false>(dets, det_stride, nullptr, order, n, thr, recs, mask, cb, stream);
  else if (geom == 2 && labels) launch_nms<2, true>(dets, det_stride, labels, order, n, thr, recs, mask, cb, stream);
  else if (geom == 2) launch_nms<2, false>(dets, det_stride, nullptr, order, n, thr, recs, mask, cb, stream);
  else if (geom == 3 && labels) launch_nms<3, true>(dets, det_stride, labels, order, n, thr, recs, mask, cb, stream);
  else if (geom == 3) launch_nms<3, false>(dets, det_stride, nullptr, order, n, thr, recs, mask, cb, stream);
  else return -1;

  size_t lds = (size_t)(cb + 1) * sizeof(u64);
  if (lds > 64 * 1024) return -1;  // n > ~524 k boxes: not a rotated-NMS workload
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(1024), lds, stream, mask, n, cb, order,
                     keep_out, count_out);
  if (sort_ascending)
    hipLaunchKernelGGL(nms_ascending_kernel, dim3(1), dim3(1024), 0, stream, n, flags, keep_out,
                       count_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
