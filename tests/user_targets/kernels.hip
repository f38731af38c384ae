// User log-densities as device KERNELS for ahmc_set_target_kernel (signature: include/ahmc_hip.h):
//     void f(const T* theta, T* lp, T* grad_neg, const int32_t* cols, int64_t n_cols, int32_t D, int64_t N, void* user)
// compiled to a code object (hipcc --genco) and bound through hipModuleLoad / hipModuleGetFunction — what a host
// language with its own GPU compiler (AMDGPU.jl) would hand over.  One wavefront per chain, 4 chains per 256-thread block.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

extern "C" __global__ __launch_bounds__(256) void iso_gauss_f64(const double* __restrict__ theta, double* __restrict__ lp, double* __restrict__ grad_neg,
                                                                 const int32_t* __restrict__ cols, int64_t n_cols, int32_t D, int64_t N, void* user) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_cols) return;
  const int64_t c = cols ? (int64_t)cols[k] : k;
  const int lane = threadIdx.x & 63;
  double ss = 0;
  for (int d = lane; d < D; d += 64) {
    const double x = theta[c * D + d];
    ss += x * x;
    grad_neg[c * D + d] = x;
  }
  ss = wave_sum(ss);
  if (lane == 0) lp[c] = -ss / 2 - D * 1.8378770664093454835606594728112 / 2;
}

// banana: ℓπ = −Σ_k [(θ_2k − a)²/2 + b (θ_2k+1 − θ_2k²)²] − (D odd: θ_{D−1}²/2); user → double[2] = (a, b) in device memory
extern "C" __global__ __launch_bounds__(256) void banana_f64(const double* __restrict__ theta, double* __restrict__ lp, double* __restrict__ grad_neg,
                                                              const int32_t* __restrict__ cols, int64_t n_cols, int32_t D, int64_t N, void* user) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_cols) return;
  const int64_t c = cols ? (int64_t)cols[k] : k;
  const int lane = threadIdx.x & 63;
  const double a = static_cast<const double*>(user)[0], b = static_cast<const double*>(user)[1];
  double part = 0;
  for (int d = 2 * lane; d < D; d += 128) {
    const double x = theta[c * D + d];
    if (d + 1 < D) {
      const double y = theta[c * D + d + 1];
      const double u = x - a, w = y - x * x;
      part -= u * u / 2 + b * w * w;
      grad_neg[c * D + d] = u - 4 * b * w * x;
      grad_neg[c * D + d + 1] = 2 * b * w;
    } else {
      part -= x * x / 2;
      grad_neg[c * D + d] = x;
    }
  }
  part = wave_sum(part);
  if (lane == 0) lp[c] = part;
}
