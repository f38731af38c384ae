// leaf_exp (ahmc_device.hpp) vs the device library's exp, bit for bit, on 2^24 arguments + the special cases.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I advancedhmc.jl_amd/csrc scripts/probe/exp_bits.hip -o scripts/probe/exp_bits.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "ahmc_device.hpp"

__global__ void k(const double* x, unsigned long long* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = ahmc::leaf_exp(x[i]), b = exp(x[i]);
  if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b)) atomicAdd(bad, 1ULL);
}

int main() {
  const int n = 1 << 24;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (s >> 11) * (1.0 / 9007199254740992.0);
    h[i] = (i & 1) ? -1100.0 + 1830.0 * u : -40.0 + 42.0 * u;  // the whole range / where leaf weights live
  }
  const double sp[] = {0.0, -0.0, 1.0, -1.0, 709.782712893384, 709.8, 1024.0, 1024.5, -745.2, -1075.0, -1075.5, 1e308, -1e308,
                       __builtin_inf(), -__builtin_inf(), __builtin_nan("")};
  std::memcpy(h.data(), sp, sizeof sp);
  double* d; unsigned long long* bad; unsigned long long hb = 0;
  hipMalloc(&d, n * sizeof(double)); hipMalloc(&bad, 8);
  hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice); hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, bad, n);
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("leaf_exp vs exp: %llu of %d arguments differ\n", hb, n);
  return hb != 0;
}
