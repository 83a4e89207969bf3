// harness_v1.cpp -- C-ABI driver around the REFERENCE's own CPU source
// r3det/ops/rnms/src/rcpu/rnms_cpu.cpp, which is #included from where it lies
// under /root/reference (path injected by oracle/build_ref.py as REF_RNMS_CPU).
// Nothing from the reference is copied into this repository.
//
// TEST INFRASTRUCTURE ONLY (see oracle/r3_oracle.cpp header).
//
// rbbox_geo (IoU v1) is CUDA-only in the reference, but its device functions
// (rbbox_geo_kernel.cu:43-228) have a token-identical CPU twin in
// rnms_cpu.cpp:11-221.  ref_v1_iou_mat drives those included templates in the
// order of mat_iou_iof_kernel (rbbox_geo_kernel.cu:238-266).
#include REF_RNMS_CPU

#include <cstdint>
#include <cstring>

extern "C" {

void ref_v1_iou_mat(const float* b1, int n1, int s1, const float* b2, int n2, int s2, int iof,
                    float* out) {
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++) {
      const float* rb1_p = b1 + (size_t)i * s1;
      const float* rb2_p = b2 + (size_t)j * s2;
      Point<float> v1[4], v2[4], u[64];  // 64: head-room so over-full cases do not smash the stack
      rbbox2points(rb1_p, v1);
      rbbox2points(rb2_p, v2);
      int p_cnt = 0;
      p_cnt += vertex_in_rbbox(v1, v2, u + p_cnt);
      p_cnt += vertex_in_rbbox(v2, v1, u + p_cnt);
      p_cnt += rbbox_border_intsec(v1, v2, u + p_cnt);
      float r = 0.f;
      if (p_cnt > 16) {
        r = -2.0f;  // marker: the reference overflows its 16-slot scratch here (undefined)
      } else if (p_cnt >= 3) {
        float s1a = rb1_p[2] * rb1_p[3];
        float s2a = rb2_p[2] * rb2_p[3];
        float su = area(u, p_cnt);
        su = std::min(su, s1a);
        su = std::min(su, s2a);
        su = std::max(su, 0.f);
        r = iof ? su / s1a : su / (s1a + s2a - su);
      }
      out[(size_t)i * n2 + j] = r;
    }
}

// rnms_cpu (rnms_cpu.cpp:284-293): dets (n,6) -> keep (ascending).
int ref_v1_rnms(const float* dets6, int n, float thr, int64_t* keep) {
  auto t = torch::from_blob(const_cast<float*>(dets6), {n, 6}, torch::kFloat32).clone();
  auto r = rnms_cpu(t, thr).contiguous();
  int k = (int)r.numel();
  if (k) std::memcpy(keep, r.data_ptr<int64_t>(), sizeof(int64_t) * k);
  return k;
}

}  // extern "C"
