// Host harness for tests/test_clip_host.py: runs csrc/r3_clip.h (the straight-line v1 clip of the drains) on the HOST
// over arrays of records, so that the algorithm is checked against the oracle without a GPU.  Test infrastructure.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "r3_clip.h"

extern "C" void clip_fast_batch(const float* recA, const float* recB, int n, int iof, float* out, uint8_t* redo) {
  for (int k = 0; k < n; k++) {
    ClipHost st;
    for (int s = 0; s < R3_CLIP_SLOTS; s++) st.px[s] = st.py[s] = 0.f;
    bool r = false;
    out[k] = v1_clip_fast(recA + (size_t)k * 9, recB + (size_t)k * 9, iof != 0, st, r);
    redo[k] = r ? 1 : 0;
  }
}

// hull records: 7 floats per box (centre, the four half products, area)
extern "C" void hull_clip_fast_batch(const float* recA, const float* recB, int n, int v2, int iou_mode, float* out,
                                     uint8_t* redo) {
  for (int k = 0; k < n; k++) {
    ClipHost st;
    for (int s = 0; s < R3_CLIP_SLOTS; s++) st.px[s] = st.py[s] = 0.f;
    bool r = false;
    out[k] = v2 ? hull_clip_fast<true>(recA + (size_t)k * 7, recB + (size_t)k * 7, iou_mode != 0, st, r)
                : hull_clip_fast<false>(recA + (size_t)k * 7, recB + (size_t)k * 7, iou_mode != 0, st, r);
    redo[k] = r ? 1 : 0;
  }
}
