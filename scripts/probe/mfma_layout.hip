// probe: register layout of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(double* out, float* outf) {
  int l = threadIdx.x;
  // A[i][k] = 100*i + k  (i = l%16, k = l/16) ; B[k][j] = (k == 0) * (j == 3 ? 1 : 0) -> C[i][3] = A[i][0] = 100 i
  double a = 100.0 * (l % 16) + (l / 16);
  double b = ((l / 16) == 0 && (l % 16) == 3) ? 1.0 : 0.0;
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
  v4f cf = {0, 0, 0, 0};
  cf = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, cf, 0, 0, 0);
  for (int v = 0; v < 4; ++v) outf[l * 4 + v] = cf[v];
}
int main() {
  double* d; float* f;
  hipMalloc(&d, 256 * 8); hipMalloc(&f, 256 * 4);
  k<<<1, 64>>>(d, f);
  double h[256]; float hf[256];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hf, f, sizeof hf, hipMemcpyDeviceToHost);
  // nonzero entries: lane, v, value -> row i = value / 100, column j should be 3
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) if (h[l * 4 + v] != 0 || (l % 16 == 3)) printf("f64 lane %d v %d = %g\n", l, v, h[l * 4 + v]);
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) if (hf[l * 4 + v] != 0 || (l % 16 == 3)) printf("f32 lane %d v %d = %g\n", l, v, hf[l * 4 + v]);
  return 0;
}
