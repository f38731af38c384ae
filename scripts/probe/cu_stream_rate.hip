// probe: what ONE workgroup of 8 waves per CU (k_dense_epoch's shape: the 16 accumulators take half the register file) moves through
// its CU's memory pipeline — 16 bytes per lane, whole 4 KB rows of chains whose pool regions lie 460 KB apart, as the kernel's epilogue
// does — as loads, as stores, as non-temporal stores, and as the epilogue's mix (3 loads + 5 stores per row set).  256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int ROW = 512;                          // doubles per vector (4 KB)
constexpr size_t CHAIN_STRIDE = 23 * 5 * ROW;     // the point pool of a chain: 23 points of 5 vectors
template <int MODE>                               // 0 loads, 1 stores, 2 non-temporal stores, 3 the epilogue's mix (3 loads, 5 nt stores)
__global__ __launch_bounds__(512) void k(double* pool, int chains_per_wg, int iters, double* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  d2 acc = {0, 0};
  for (int it = 0; it < iters; ++it) {
    for (int u = 0; u < chains_per_wg / 8; ++u) {  // each wave: whole chains (two per pass here, as 32 lanes x 16 B x 4 pieces per row)
      const size_t c = (size_t)blockIdx.x * chains_per_wg + (size_t)u * 8 + w;
      double* base = pool + c * CHAIN_STRIDE + (size_t)((it * 7 + u) % 23) * 5 * ROW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d2* p0 = reinterpret_cast<d2*>(base + (i * 64 + lane) * 2);
        if (MODE == 0) { acc += p0[0]; acc += p0[ROW / 2]; acc += p0[ROW]; acc += p0[3 * ROW / 2]; acc += p0[2 * ROW]; }
        if (MODE == 1) { const d2 v = {(double)it, (double)lane}; p0[0] = v; p0[ROW / 2] = v; p0[ROW] = v; p0[3 * ROW / 2] = v; p0[2 * ROW] = v; }
        if (MODE == 2) { const d2 v = {(double)it, (double)lane}; for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(v, p0 + q * ROW / 2); }
        if (MODE == 3) {
          d2 a = p0[0], b = p0[ROW / 2], c2 = p0[ROW];
          a += b; b += c2;
          double* spec = base + 5 * ROW;  // the next point of the chain
          d2* s0 = reinterpret_cast<d2*>(spec + (i * 64 + lane) * 2);
          __builtin_nontemporal_store(a, p0 + ROW / 2); __builtin_nontemporal_store(b, p0 + 3 * ROW / 2);
          __builtin_nontemporal_store(a, s0); __builtin_nontemporal_store(b, s0 + ROW / 2); __builtin_nontemporal_store(c2, s0 + ROW);
          acc += a;
        }
      }
    }
  }
  if (acc[0] == 12345.678) sink[0] = acc[1];
}
template <int MODE>
void run(const char* what, double* pool, double* sink, int vecs_per_row_set) {
  const int wgs = 256, cpw = 32, iters = 64;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<wgs, 512>>>(pool, cpw, 2, sink);
  (void)hipEventRecord(e0);
  k<MODE><<<wgs, 512>>>(pool, cpw, iters, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * cpw * iters * vecs_per_row_set * ROW * 8;
  printf("%-28s %7.3f ms  %6.2f TB/s chip  %6.1f GB/s per CU\n", what, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
}
int main() {
  const size_t n = (size_t)256 * 32 * CHAIN_STRIDE + 16 * ROW;
  double *pool, *sink; (void)hipMalloc(&pool, n * 8); (void)hipMalloc(&sink, 64); (void)hipMemset(pool, 0, n * 8);
  run<0>("loads (5 vectors)", pool, sink, 5);
  run<1>("stores (5 vectors)", pool, sink, 5);
  run<2>("non-temporal stores (5)", pool, sink, 5);
  run<3>("epilogue mix (3 ld + 5 nt st)", pool, sink, 8);
  return 0;
}
