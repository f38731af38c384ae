// probe: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (independent accumulators, no memory traffic)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  double* d; (void)hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(d, 10);
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 1024 * NACC * (double)iters * blocks * 4;
  printf("NACC=%d blocks=%d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per wave\n", NACC, blocks, ms, flops / ms / 1e9, ms * 1e6 / ((double)iters * NACC));
  (void)hipFree(d);
}
int main() {
  run<1>(256, 20000); run<2>(256, 20000); run<4>(256, 10000); run<8>(256, 5000);
  run<4>(512, 10000); run<4>(1024, 10000); run<4>(2048, 5000);
  return 0;
}
