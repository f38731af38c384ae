// probe: sustained VALU issue rate on gfx950, PER INSTRUCTION CLASS (what the VALU-issue roofline of k_nuts is priced against).
//
// MI355X_MICROARCH.md ("Wave scheduling"): a CU has 4 SIMD-32 units, a wave64 VALU instruction issues over 2 cycles — but
// f64 arithmetic runs at 78.6 TFLOP/s = 256 CUs × 4 SIMDs × 2.4 GHz × 64 lanes × 2 flop / 4 cycles, i.e. one v_fma_f64 per
// 4 cycles per SIMD.  k_nuts is a MIX (f64 arithmetic, 32-bit DPP moves, permlane swaps, Philox integer multiplies,
// compares / selects), so its roof is the mix-weighted one: Σ_class n_class · cycles_class per leapfrog.
//
// Each kernel issues one instruction class from 8 independent register chains (dependent only on itself every 8th
// instruction), no memory traffic, W waves per SIMD (W blocks of 256 threads per CU, pinned by the LDS allocation).
// Reported per class and W: cycles per wave-instruction per SIMD from s_memtime (clock-independent) and
// G wave-instr/s per chip from HIP events (what the bench's roofline divides by).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/valu_rate.hip -o scripts/probe/valu_rate.bin && scripts/probe/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

enum Op {
  FMA_F64, ADD_F64, MUL_F64, MAX_F64, LDEXP_F64, RNDNE_F64, RCP_F64, CMP_F64, CVT_I32_F64, CVT_F64_I32,
  FMA_F32, ADD_F32, PK_FMA_F32,
  ADD_U32, XOR_B32, LSHL_B32, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32, LSHL_B64, LSHL_ADD_U64, ADD_CO_U32,
  MOV_B32, MOV_B64, CNDMASK_B32, DPP_ROR, DPP_QUAD, DPP_MIRROR, DPP_BCAST, PERMLANE16_SWAP, PERMLANE32_SWAP, READFIRSTLANE,
  DS_SWIZZLE, S_MOV,
  N_OPS
};
static const char* NAMES[N_OPS] = {
  "v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_ldexp_f64", "v_rndne_f64", "v_rcp_f64", "v_cmp_lt_f64", "v_cvt_i32_f64", "v_cvt_f64_i32",
  "v_fma_f32", "v_add_f32", "v_pk_fma_f32",
  "v_add_u32", "v_xor_b32", "v_lshlrev_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_lshl_add_u64", "v_add_co_u32",
  "v_mov_b32", "v_mov_b64", "v_cndmask_b32", "v_mov_b32_dpp row_ror:4", "v_mov_b32_dpp quad_perm", "v_mov_b32_dpp row_mirror", "v_mov_b32_dpp row_newbcast:0",
  "v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_readfirstlane_b32",
  "ds_swizzle_b32 (LDS pipe)", "s_mov_b32 (scalar pipe)"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(unsigned long long* out, int iters) {
  extern __shared__ char lds_pin[];  // (occupancy pin only)
  double d[8];
  float f[8];
  unsigned u[8];
  unsigned long long q[8];
  for (int i = 0; i < 8; ++i) {
    d[i] = 1.0 + threadIdx.x * 1e-9 + i;
    f[i] = 1.0f + threadIdx.x * 1e-6f + i;
    u[i] = threadIdx.x * 2654435761u + i;
    q[i] = ((unsigned long long)u[i] << 20) + i;
  }
  const double db = 0.999999, dc = 1e-9;
  const float fb = 0.9999f, fc = 1e-6f;
  const unsigned ub = 0x9E3779B9u;
  unsigned sink_s = 0;
  const unsigned long long cmask = 0x5555555555555555ull ^ (unsigned long long)iters;  // (select mask in an SGPR pair: no vcc hazard nops)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if constexpr (OP == FMA_F64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD_F64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
        REP8(X)
#undef X
      } else if constexpr (OP == MUL_F64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
        REP8(X)
#undef X
      } else if constexpr (OP == MAX_F64) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
        REP8(X)
#undef X
      } else if constexpr (OP == LDEXP_F64) {
#define X(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i] & 1));
        REP8(X)
#undef X
      } else if constexpr (OP == RNDNE_F64) {
#define X(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == RCP_F64) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == CMP_F64) {
#define X(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (OP == CVT_I32_F64) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == CVT_F64_I32) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD_F32) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fc));
        REP8(X)
#undef X
      } else if constexpr (OP == PK_FMA_F32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
        REP8(X)
#undef X
      } else if constexpr (OP == XOR_B32) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
        REP8(X)
#undef X
      } else if constexpr (OP == LSHL_B32) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
        REP8(X)
#undef X
      } else if constexpr (OP == MUL_HI_U32) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
        REP8(X)
#undef X
      } else if constexpr (OP == MAD_U64_U32) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(ub) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (OP == LSHL_B64) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == LSHL_ADD_U64) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD_CO_U32) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(ub) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (OP == MOV_B32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if constexpr (OP == MOV_B64) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if constexpr (OP == CNDMASK_B32) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "s"(cmask));
        REP8(X)
#undef X
      } else if constexpr (OP == DPP_ROR) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == DPP_QUAD) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == DPP_MIRROR) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == DPP_BCAST) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == PERMLANE16_SWAP) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(f[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == PERMLANE32_SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(f[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == READFIRSTLANE) {
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sink_s) : "v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == DS_SWIZZLE) {
#define X(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n s_waitcnt lgkmcnt(0)" : "+v"(u[i]));
        REP8(X)
#undef X
      } else if constexpr (OP == S_MOV) {
#define X(i) asm volatile("s_mov_b32 %0, %0" : "+s"(sink_s));
        REP8(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  unsigned long long x = sink_s;
  for (int i = 0; i < 8; ++i) { s += d[i] + f[i]; x += u[i] + q[i]; }
  if (s == 12345.678 && x == 42) out[0] = 1;  // (keeps the chains alive; never true)
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { double cyc_per_inst_simd; double ginstr; double ms; };

template <int OP>
Res run(int W, int iters, int n_cu) {
  const int blocks = n_cu * W;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * (1 + blocks * 4));
  (void)hipMemset(d, 0, sizeof(unsigned long long) * (1 + blocks * 4));
  const size_t lds = (size_t)(160 * 1024) / W - 1024;  // W blocks fill a CU's LDS: W waves on each of the 4 SIMDs
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rate<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), lds, 0, d, 16);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), lds, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * (1 + blocks * 4));
  (void)hipMemcpy(h, d, sizeof(unsigned long long) * (1 + blocks * 4), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < blocks * 4; ++i) mean += (double)h[1 + i];
  mean /= blocks * 4;
  free(h);
  (void)hipFree(d);
  const double n_inst = (double)iters * 64;  // per wave
  Res r;
  r.cyc_per_inst_simd = mean / (n_inst * W);  // W waves share the SIMD for `mean` cycles and issue W·n_inst instructions
  r.ginstr = n_inst * blocks * 4 / (ms * 1e-3) / 1e9;
  r.ms = ms;
  return r;
}

template <int OP>
void all(int n_cu, int iters, FILE* js, bool& first) {
  printf("%-32s", NAMES[OP]);
  if (js) fprintf(js, "%s\n  \"%s\": {", first ? "" : ",", NAMES[OP]);
  first = false;
  const int Ws[4] = {1, 2, 4, 8};
  for (int k = 0; k < 4; ++k) {
    Res r = run<OP>(Ws[k], iters, n_cu);
    printf("  W=%d %6.2f cyc %7.1f G/s", Ws[k], r.cyc_per_inst_simd, r.ginstr);
    if (js) fprintf(js, "%s\"W%d\": {\"cycles_per_wave_instr_per_simd\": %.4f, \"gwave_instr_per_s_chip\": %.2f}", k ? ", " : "", Ws[k], r.cyc_per_inst_simd, r.ginstr);
  }
  printf("\n");
  if (js) fprintf(js, "}");
  if constexpr (OP + 1 < N_OPS) all<OP + 1>(n_cu, iters, js, first);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  printf("%s: %d CUs, clockRate %d kHz; cycles = s_memtime ticks per wave-instruction per SIMD; G/s = 1e9 wave-instr/s over the chip (HIP events)\n",
         p.name, n_cu, p.clockRate);
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\"device\": \"%s\", \"n_cu\": %d, \"clock_khz\": %d, \"classes\": {", p.name, n_cu, p.clockRate);
  bool first = true;
  all<0>(n_cu, 4000, js, first);
  if (js) { fprintf(js, "\n}}\n"); fclose(js); }
  return 0;
}
