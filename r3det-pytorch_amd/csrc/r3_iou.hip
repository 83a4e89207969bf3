// r3_iou.hip -- rotated IoU matrix / vector kernels for gfx950.
//
// Matrix kernel (replaces mat_iou_iof_kernel rbbox_geo_kernel.cu:231-268 and
// box_iou_rotated_cuda_kernel box_iou_rotated_cuda.cu:14-63):
//   * lane <-> output column, so one wavefront writes 64 consecutive floats of an output
//     row (the reference maps threadIdx.x to the row and writes with stride n2);
//   * the row operand is staged once per workgroup as prepared records in LDS (broadcast
//     reads), the column operand lives in registers as a prepared record;
//   * per pair: circle test (~10 VALU) -> store 0; only surviving pairs clip.
// The kernel is HBM-write-bound on assignment-shaped inputs (4 B per pair).
#include <hip/hip_runtime.h>

#include "r3_geom.h"
#include "r3_kernels.h"

namespace {

constexpr int IOU_BLOCK = 256;   // columns per workgroup (4 wavefronts)
constexpr int IOU_ROWS = 128;    // row records staged in LDS per workgroup (6 KB)

template <int GEOM>
__global__ __launch_bounds__(IOU_BLOCK) void iou_mat_kernel(const float* __restrict__ b1, int n1,
                                                            const float* __restrict__ b2, int n2,
                                                            int iof, float* __restrict__ out) {
  __shared__ BoxRec rows[IOU_ROWS];
  const int col = blockIdx.x * IOU_BLOCK + threadIdx.x;
  const int row0 = blockIdx.y * IOU_ROWS;
  const int nrows = min(IOU_ROWS, n1 - row0);

  for (int r = threadIdx.x; r < nrows; r += IOU_BLOCK) {
    BoxRec rec;
    make_record<GEOM>(b1 + (size_t)(row0 + r) * 5, 0.f, rec);
    rows[r] = rec;
  }
  BoxRec mine;
  if (col < n2) make_record<GEOM>(b2 + (size_t)col * 5, 0.f, mine);
  __syncthreads();
  if (col >= n2) return;

  float* o = out + (size_t)row0 * n2 + col;
  for (int r = 0; r < nrows; r++) {
    const BoxRec& A = rows[r];
    float v;
    if (circles_apart(A.f[9], A.f[10], A.f[11], mine.f[9], mine.f[10], mine.f[11])) {
      v = 0.f;
    } else {
      BoxRec a = A;
      if (GEOM == 1) v = v1_pair_slow(a, mine, iof != 0);
      else if (GEOM == 2) v = hull_pair_slow<true>(a, mine, iof == 0);
      else v = hull_pair_slow<false>(a, mine, iof == 0);
    }
    o[(size_t)r * n2] = v;
  }
}

// vec_iou_iof_kernel (rbbox_geo_kernel.cu:271-309): out[i] = f(b1[i % n1], b2[i % n2]).
template <int GEOM>
__global__ __launch_bounds__(256) void iou_vec_kernel(const float* __restrict__ b1, int n1,
                                                      const float* __restrict__ b2, int n2,
                                                      int iof, float* __restrict__ out) {
  const int n = max(n1, n2);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
    BoxRec A, B;
    make_record<GEOM>(b1 + (size_t)(i % n1) * 5, 0.f, A);
    make_record<GEOM>(b2 + (size_t)(i % n2) * 5, 0.f, B);
    out[i] = pair_iou<GEOM, false>(A, B, iof != 0);
  }
}

}  // namespace

int r3k_iou_mat(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                hipStream_t stream) {
  if (n1 == 0 || n2 == 0) return 0;
  dim3 grid((n2 + IOU_BLOCK - 1) / IOU_BLOCK, (n1 + IOU_ROWS - 1) / IOU_ROWS);
  dim3 block(IOU_BLOCK);
  switch (geom) {
    case 1: hipLaunchKernelGGL(iou_mat_kernel<1>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 2: hipLaunchKernelGGL(iou_mat_kernel<2>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 3: hipLaunchKernelGGL(iou_mat_kernel<3>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int r3k_iou_vec(int geom, int iof, const float* b1, int n1, const float* b2, int n2, float* out,
                hipStream_t stream) {
  if (n1 == 0 || n2 == 0) return 0;
  int n = n1 > n2 ? n1 : n2;
  dim3 grid(min((n + 255) / 256, 2048)), block(256);
  switch (geom) {
    case 1: hipLaunchKernelGGL(iou_vec_kernel<1>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 2: hipLaunchKernelGGL(iou_vec_kernel<2>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    case 3: hipLaunchKernelGGL(iou_vec_kernel<3>, grid, block, 0, stream, b1, n1, b2, n2, iof, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
