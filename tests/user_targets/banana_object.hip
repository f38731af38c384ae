// The banana density of banana.hpp behind the OBJECT ABI (include/ahmc_user_target_object.h): ONE C symbol per element type, arrays by
// pointer, no engine header but that declaration — what a user (or GPUCompiler.jl, for a Julia function) compiles WITHOUT the engine's
// sources, to a relocatable device object (`hipcc -fgpu-rdc -c`) or to raw amdgcn bitcode (`--cuda-device-only -emit-llvm`), and hands
// to build_target_plugin_from_object.  Same arithmetic, operation for operation, as banana.hpp: the chains must agree bit for bit.
#include <hip/hip_runtime.h>

#include "ahmc_user_target_object.h"

template <class T>
static __device__ __forceinline__ T banana(const T* params, int D, int E, const T* th, T* grad_neg, int d0) {
  const T a = params[0], b = params[1];
  T part = 0;
  for (int e = 0; e < E; e += 2) {
    const int d = d0 + e;
    const T x = th[e], y = th[e + 1];
    if (d + 1 < D) {
      const T u = x - a, w = y - x * x;
      part -= u * u / 2 + b * w * w;
      grad_neg[e] = u - 4 * b * w * x;
      grad_neg[e + 1] = 2 * b * w;
    } else if (d < D) {
      part -= x * x / 2;
      grad_neg[e] = x;
      grad_neg[e + 1] = 0;
    } else {
      grad_neg[e] = 0;
      grad_neg[e + 1] = 0;
    }
  }
  return part;
}

extern "C" __device__ double ahmc_user_logdensity_f64(const double* params, int D, int E, const double* theta, double* grad_neg, int /*lane*/, int d0, int /*G*/) {
  return banana<double>(params, D, E, theta, grad_neg, d0);
}
extern "C" __device__ float ahmc_user_logdensity_f32(const float* params, int D, int E, const float* theta, float* grad_neg, int /*lane*/, int d0, int /*G*/) {
  return banana<float>(params, D, E, theta, grad_neg, d0);
}
