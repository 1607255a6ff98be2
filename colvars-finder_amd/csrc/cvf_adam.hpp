// Adam update shared by the stand-alone optimiser kernel and the fused gradient-reduce kernels.
#pragma once
#include "cvf_pack.hpp"

struct AdamDev {           // device-side view of cvf_adam_args
  float* theta;
  float* m;
  float* v;
  float lr, b1, b2, eps;
  const int32_t* step;     // number t >= 1 of the current step (advanced by the gradient kernel)
  float* packed;           // MFMA fragment copy to refresh, or nullptr
  const float* lr_dev;     // device scalar overriding `lr` (a captured hipGraph then follows the host's learning rate), or nullptr
};

struct AdamScalars {
  float step_size, bc2_sqrt;
};
__device__ __forceinline__ AdamScalars adam_scalars(const AdamDev& a) {
  const int t = *a.step;
  const double bc1 = 1.0 - pow((double)a.b1, (double)t);
  const double bc2 = 1.0 - pow((double)a.b2, (double)t);
  const float lr = a.lr_dev != nullptr ? *a.lr_dev : a.lr;
  return AdamScalars{(float)((double)lr / bc1), (float)sqrt(bc2)};
}
// torch.optim.Adam (single-tensor path): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// theta -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// (m0, v0, th0: the parameter's state, which a caller may have requested long before the gradient is known)
__device__ __forceinline__ void adam_apply(const AdamDev& a, const AdamScalars& sc, const PackTab& tab, int64_t i, float g,
                                           float m0, float v0, float th0) {
  // (explicit fused multiply-adds: left to the compiler, the contraction of these expressions depends on the surrounding
  //  kernel, and the stand-alone and the fused update must agree bit for bit)
  const float mi = fmaf(g - m0, 1.0f - a.b1, m0);
  const float g2 = ((1.0f - a.b2) * g) * g;
  const float vi = fmaf(a.b2, v0, g2);
  a.m[i] = mi;
  a.v[i] = vi;
  const float denom = sqrtf(vi) / sc.bc2_sqrt + a.eps;
  const float th = fmaf(-sc.step_size, mi / denom, th0);
  a.theta[i] = th;
  if (a.packed != nullptr) pack_scatter(tab, (int)i, th, a.packed);
}
__device__ __forceinline__ void adam_apply(const AdamDev& a, const AdamScalars& sc, const PackTab& tab, int64_t i, float g) {
  adam_apply(a, sc, tab, i, g, a.m[i], a.v[i], a.theta[i]);
}
