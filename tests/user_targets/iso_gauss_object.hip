// tests/user_targets/iso_gauss.hpp behind the OBJECT ABI (include/ahmc_user_target_object.h): the isotropic Gaussian as compiled device
// code with one C symbol — for scripts/user_target_bench.py (`object` / `bitcode` rows: a linked density against the built-in family)
#include <hip/hip_runtime.h>

#include "ahmc_user_target_object.h"

extern "C" __device__ double ahmc_user_logdensity_f64(const double* /*params*/, int D, int E, const double* th, double* grad_neg, int lane, int /*d0*/, int /*G*/) {
  double ss = 0;
  for (int e = 0; e < E; ++e) {
    ss += th[e] * th[e];
    grad_neg[e] = th[e];   // padding: θ = 0 → g = 0
  }
  double part = -ss / 2;
  if (lane == 0) part -= (double)D * 1.8378770664093454835606594728112 / 2;
  return part;
}
extern "C" __device__ float ahmc_user_logdensity_f32(const float* /*params*/, int D, int E, const float* th, float* grad_neg, int lane, int /*d0*/, int /*G*/) {
  float ss = 0;
  for (int e = 0; e < E; ++e) {
    ss += th[e] * th[e];
    grad_neg[e] = th[e];
  }
  float part = -ss / 2;
  if (lane == 0) part -= (float)D * 1.8378770664093454835606594728112f / 2;
  return part;
}
