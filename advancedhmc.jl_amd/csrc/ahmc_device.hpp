// ahmc_device.hpp — device-side building blocks of the chain-batched HMC/NUTS engine (gfx950).
//
// Thread mapping: a chain is owned by a *group* of G consecutive lanes of one
// wave64 (G ∈ {4,8,16,32,64}); lane l of the group holds the E contiguous elements
// d = l*E .. l*E+E-1 of every (D,) vector of its chain in registers.  G*E >= D; padded elements
// are kept at exactly 0 (momentum/gradient) so they never contribute to a reduction.
// All per-chain scalars (energies, weights, RNG draws, control flow) are computed redundantly
// and bit-identically by every lane of the group, so control flow is uniform inside a group.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ahmc {

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG.  Stream layout is the specification shared with the oracle:
//   counter = (chain, iteration, purpose, slot), key = (seed_lo, seed_hi)
// ------------------------------------------------------------------------------------------------
enum : uint32_t { RNG_MOMENTUM = 0, RNG_TRANSITION = 1, RNG_JITTER = 2, RNG_FINDEPS = 3 };
constexpr uint32_t COUPLED_CHAIN = 0xFFFFFFFFu;

struct Philox4 {
  uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniform in the open interval (0,1)
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  uint64_t bits = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
  return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

struct Rng {
  uint32_t k0, k1, chain, iter;
  __device__ __forceinline__ Philox4 raw(uint32_t purpose, uint32_t slot) const {
    return philox4x32_10(chain, iter, purpose, slot, k0, k1);
  }
  __device__ __forceinline__ double uniform(uint32_t purpose, uint32_t slot) const {
    Philox4 p = raw(purpose, slot);
    return u53(p.v[0], p.v[1]);
  }
  __device__ __forceinline__ bool boolean(uint32_t purpose, uint32_t slot) const {
    return (raw(purpose, slot).v[0] >> 31) != 0;
  }
  __device__ __forceinline__ double randexp(uint32_t purpose, uint32_t slot) const {
    return -log(uniform(purpose, slot));
  }
  // Box–Muller pair `pair` → (z0, z1) = standard normals of elements 2*pair, 2*pair+1
  __device__ __forceinline__ void normal_pair(uint32_t purpose, uint32_t pair, double& z0, double& z1) const {
    Philox4 p = raw(purpose, pair);
    double u1 = u53(p.v[0], p.v[1]), u2 = u53(p.v[2], p.v[3]);
    double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(6.283185307179586476925286766559 * u2, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
  }
};

// E standard normals for the elements d0 .. d0+E-1 owned by this lane
template <class T, int E>
__device__ __forceinline__ void normals(const Rng& rng, uint32_t purpose, int d0, T (&z)[E]) {
  if constexpr (E == 1) {
    double a, b;
    rng.normal_pair(purpose, (uint32_t)d0 >> 1, a, b);
    z[0] = (T)((d0 & 1) ? b : a);
  } else {
    static_assert(E % 2 == 0, "E must be 1 or even");
#pragma unroll
    for (int e = 0; e < E; e += 2) {
      double a, b;
      rng.normal_pair(purpose, (uint32_t)(d0 + e) >> 1, a, b);
      z[e] = (T)a;
      z[e + 1] = (T)b;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// numerics helpers that must follow Julia semantics (NaN-propagating min/max; LogExpFunctions)
// ------------------------------------------------------------------------------------------------
template <class T> struct Lim;
template <> struct Lim<float> {
  static __device__ __forceinline__ float inf() { return __int_as_float(0x7f800000); }
  static __device__ __forceinline__ float nan() { return __int_as_float(0x7fc00000); }
};
template <> struct Lim<double> {
  static __device__ __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000LL); }
  static __device__ __forceinline__ double nan() { return __longlong_as_double(0x7ff8000000000000LL); }
};

template <class T> __device__ __forceinline__ T jl_min(T a, T b) { return (a != a || b != b) ? Lim<T>::nan() : (a < b ? a : b); }
template <class T> __device__ __forceinline__ T jl_max(T a, T b) { return (a != a || b != b) ? Lim<T>::nan() : (a > b ? a : b); }
template <class T> __device__ __forceinline__ bool is_finite(T v) { return isfinite(v); }
// PhasePoint constructor: non-finite ℓπ/ℓκ values become -Inf (src/hamiltonian.jl:95-104)
template <class T> __device__ __forceinline__ T sanitize(T v) { return is_finite(v) ? v : -Lim<T>::inf(); }
// LogExpFunctions.logaddexp (src/trajectory.jl:192,198)
template <class T> __device__ __forceinline__ T logaddexp(T x, T y) {
  T d = (x == y) ? T(0) : fabs(x - y);
  return jl_max(x, y) + log1p(exp(-d));
}
// exp for the leaf weight of the linear-domain NUTS kernels, one call per leapfrog.  Float64: the ROCm device library's
// own algorithm (2^n range reduction with a two-part ln 2, degree-11 minimax Horner, ldexp; same constants, same
// operation order, so the same bits as `exp`), with the Horner steps issued as three-address v_fma_f64.  The
// compiler's version of that loop re-materialises each coefficient in front of a two-address v_fmac_f64 — 9 extra
// VALU instructions per leaf of a kernel that is VALU-issue bound.
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double leaf_exp(double x) {
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  const double dn = __builtin_rint(x * C(0x3ff71547652b82feULL));                 // x·log2(e)
  double t = __builtin_fma(dn, C(0xbfe62e42fefa39efULL), x);                      // − n·ln2 (high part)
  t = __builtin_fma(dn, C(0xbc7abc9e3b39803fULL), t);                             // − n·ln2 (low part)
  double q = __builtin_fma(t, C(0x3e5ade156a5dcb37ULL), C(0x3e928af3fca7ab0cULL));
  q = fma3(t, q, C(0x3ec71dee623fde64ULL));
  q = fma3(t, q, C(0x3efa01997c89e6b0ULL));
  q = fma3(t, q, C(0x3f2a01a014761f6eULL));
  q = fma3(t, q, C(0x3f56c16c1852b7b0ULL));
  q = fma3(t, q, C(0x3f81111111122322ULL));
  q = fma3(t, q, C(0x3fa55555555502a1ULL));
  q = fma3(t, q, C(0x3fc5555555555511ULL));
  q = fma3(t, q, C(0x3fe000000000000bULL));
  q = __builtin_fma(t, q, 1.0);
  q = __builtin_fma(t, q, 1.0);
  double z = __builtin_ldexp(q, (int)dn);
  z = x > 1024.0 ? Lim<double>::inf() : z;
  z = x < -1075.0 ? 0.0 : z;
  return z;
}
__device__ __forceinline__ float leaf_exp(float x) { return exp(x); }

// The weight of a NUTS leaf in the linear-domain kernels: W = exp(ℓw) for ℓw <= LINW_LIMIT (a larger ℓw flags the chain
// for the log-domain redo pass and its W is never used), so the overflow branch of exp is dead; ℓw = -Inf (a divergent
// leaf) must give exactly 0.  AHMC_LEAF_EXP = 0: leaf_exp (the device library's algorithm, 11 Horner steps);
// 1: the same without the overflow select (3 VALU less, same bits where it matters);
// 2: table-assisted — x = (64 n + j)·ln2/64 + t, |t| <= ln2/128: exp(x) = 2^n · T[j] · (1 + (e^t − 1)) with a 64-entry
//    table of 2^(j/64) (a scalar load when the chain owns the wave) and a degree-5 polynomial: ≈9 VALU less than the
//    Horner-11 form; ≤ 1.01 ulp against the exact value (checked on 2·10^5 arguments), i.e. last-ulp differences
//    against the device library's exp().
#ifndef AHMC_LEAF_EXP
#define AHMC_LEAF_EXP 2   // measured on cfg2 (sampling phase, in-kernel leapfrog/s): 1: 2.654e9, 2: 2.718e9
#endif
// 2^(j/64), j = 0..63, correctly rounded (generated with 60-digit decimal arithmetic)
static __device__ const unsigned long long EXP2_64_BITS[64] = {
    0x3ff0000000000000ULL, 0x3ff02c9a3e778061ULL, 0x3ff059b0d3158574ULL, 0x3ff0874518759bc8ULL,
    0x3ff0b5586cf9890fULL, 0x3ff0e3ec32d3d1a2ULL, 0x3ff11301d0125b51ULL, 0x3ff1429aaea92de0ULL,
    0x3ff172b83c7d517bULL, 0x3ff1a35beb6fcb75ULL, 0x3ff1d4873168b9aaULL, 0x3ff2063b88628cd6ULL,
    0x3ff2387a6e756238ULL, 0x3ff26b4565e27cddULL, 0x3ff29e9df51fdee1ULL, 0x3ff2d285a6e4030bULL,
    0x3ff306fe0a31b715ULL, 0x3ff33c08b26416ffULL, 0x3ff371a7373aa9cbULL, 0x3ff3a7db34e59ff7ULL,
    0x3ff3dea64c123422ULL, 0x3ff4160a21f72e2aULL, 0x3ff44e086061892dULL, 0x3ff486a2b5c13cd0ULL,
    0x3ff4bfdad5362a27ULL, 0x3ff4f9b2769d2ca7ULL, 0x3ff5342b569d4f82ULL, 0x3ff56f4736b527daULL,
    0x3ff5ab07dd485429ULL, 0x3ff5e76f15ad2148ULL, 0x3ff6247eb03a5585ULL, 0x3ff6623882552225ULL,
    0x3ff6a09e667f3bcdULL, 0x3ff6dfb23c651a2fULL, 0x3ff71f75e8ec5f74ULL, 0x3ff75feb564267c9ULL,
    0x3ff7a11473eb0187ULL, 0x3ff7e2f336cf4e62ULL, 0x3ff82589994cce13ULL, 0x3ff868d99b4492edULL,
    0x3ff8ace5422aa0dbULL, 0x3ff8f1ae99157736ULL, 0x3ff93737b0cdc5e5ULL, 0x3ff97d829fde4e50ULL,
    0x3ff9c49182a3f090ULL, 0x3ffa0c667b5de565ULL, 0x3ffa5503b23e255dULL, 0x3ffa9e6b5579fdbfULL,
    0x3ffae89f995ad3adULL, 0x3ffb33a2b84f15fbULL, 0x3ffb7f76f2fb5e47ULL, 0x3ffbcc1e904bc1d2ULL,
    0x3ffc199bdd85529cULL, 0x3ffc67f12e57d14bULL, 0x3ffcb720dcef9069ULL, 0x3ffd072d4a07897cULL,
    0x3ffd5818dcfba487ULL, 0x3ffda9e603db3285ULL, 0x3ffdfc97337b9b5fULL, 0x3ffe502ee78b3ff6ULL,
    0x3ffea4afa2a490daULL, 0x3ffefa1bee615a27ULL, 0x3fff50765b6e4540ULL, 0x3fffa7c1819e90d8ULL};
#ifndef AHMC_LEAF_EXP_NARROW_TABLE
// Chains that SHARE a wave (G < 64): the table entry is a per-lane GLOBAL load in the middle of the leaf's chain of dependent stages — and
// its s_waitcnt vmcnt(0) also waits for every slot store still in flight.  Round 5, stage stamps of a measurement build
// (profiles/r5_leaf_latency_cfg3.json): the weight stage took 1 112 of a lone wave's 2 905 cycles per leaf step on cfg3 (1 830 of 4 023 at
// the bench's occupancy) against 310 of 1 571 on cfg2, where the entry is a scalar load.  0: those kernels evaluate the Horner-11 form
// (no memory access: 9 more VALU, ≈ 700 fewer cycles); 1: the table for every geometry (rounds 3–4).
#define AHMC_LEAF_EXP_NARROW_TABLE 0
#endif
#ifndef AHMC_LEAF_MICRO
#define AHMC_LEAF_MICRO 1   // round 5's two instruction cuts in the leaf (no underflow select in the weight's exp; ΔH_max by a scalar branch on the direction); 0: round 4's forms, for A/B runs
#endif
template <bool UNIFORM>
__device__ __forceinline__ double leaf_weight_exp(double x) {
  constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
#if AHMC_LEAF_EXP == 2
  if constexpr (UNIFORM || AHMC_LEAF_EXP_NARROW_TABLE) {
  // (clamped first: ℓw = −Inf is a routine input — every divergent leaf — and a float-to-int conversion of ±Inf is undefined
  // behaviour, however the hardware's v_cvt_i32_f64 saturates; the final select still returns 0 below −1075)
  const double xc = x < -1100.0 ? -1100.0 : x;
  const double dk = __builtin_rint(xc * 92.33248261689366);                // 64 / ln 2
  double t = __builtin_fma(dk, -0x1.62e42fef00000p-7, xc);                  // − k·ln2/64: high part (20 trailing zero bits: exact product)
  t = __builtin_fma(dk, -0x1.473de6af278edp-40, t);                         // low part
  int k = (int)dk;
  if constexpr (UNIFORM) k = __builtin_amdgcn_readfirstlane(k);             // a chain owns the wave: scalar table load
  const double tj = C(EXP2_64_BITS[k & 63]);
  double q = __builtin_fma(t, 8.3333333333333332e-03, 4.1666666666666664e-02);  // 1/120, 1/24
  q = fma3(t, q, 1.6666666666666666e-01);
  q = fma3(t, q, 0.5);
  q = fma3(t, q, 1.0);
  q = q * t;                                                                // e^t − 1
  // (no select for the underflow: x is clamped at −1100 above — ℓw = −Inf included — and from −1075 down ldexp's exponent is ≤ −1551, below
  // the smallest subnormal: the result IS 0 there; a NaN x fails the clamp's compare and propagates.  Round 5: three VALU per leaf fewer.)
#if AHMC_LEAF_MICRO
  return __builtin_ldexp(__builtin_fma(tj, q, tj), k >> 6);
#else
  double z = __builtin_ldexp(__builtin_fma(tj, q, tj), k >> 6);
  z = x < -1075.0 ? 0.0 : z;
  return z;
#endif
  }
#endif
  {
  const double xh = x < -1100.0 ? -1100.0 : x;                                    // (as above: −Inf is a routine input; ldexp underflows to 0 for it)
  const double dn = __builtin_rint(xh * C(0x3ff71547652b82feULL));                // x·log2(e)
  double t = __builtin_fma(dn, C(0xbfe62e42fefa39efULL), xh);                     // − n·ln2 (high part)
  t = __builtin_fma(dn, C(0xbc7abc9e3b39803fULL), t);                             // − n·ln2 (low part)
  double q = __builtin_fma(t, C(0x3e5ade156a5dcb37ULL), C(0x3e928af3fca7ab0cULL));
  q = fma3(t, q, C(0x3ec71dee623fde64ULL));
  q = fma3(t, q, C(0x3efa01997c89e6b0ULL));
  q = fma3(t, q, C(0x3f2a01a014761f6eULL));
  q = fma3(t, q, C(0x3f56c16c1852b7b0ULL));
  q = fma3(t, q, C(0x3f81111111122322ULL));
  q = fma3(t, q, C(0x3fa55555555502a1ULL));
  q = fma3(t, q, C(0x3fc5555555555511ULL));
  q = fma3(t, q, C(0x3fe000000000000bULL));
  q = __builtin_fma(t, q, 1.0);
  q = __builtin_fma(t, q, 1.0);
  double z = __builtin_ldexp(q, (int)dn);
#if AHMC_LEAF_EXP == 0
  z = x > 1024.0 ? Lim<double>::inf() : z;
#endif
#if !AHMC_LEAF_MICRO
  z = x < -1075.0 ? 0.0 : z;
#endif
  return z;   // (the underflow needs no select either: clamped at −1100, ldexp's exponent −1587 gives 0)
  }
}
template <bool UNIFORM>
__device__ __forceinline__ float leaf_weight_exp(float x) { return exp(x); }

// exp(min(0, ℓw)) for the log-domain kernels, through leaf_weight_exp: min(1, W) of the linear-domain kernels, bit for bit
// (ℓw > 0 gives exactly 1 there as well: W >= 1).  NaN propagates (Julia's min): the compare fails and exp(NaN) = NaN.
template <class T, bool UNIFORM>
__device__ __forceinline__ T alpha_from_logweight(T lw) {
  const T w = leaf_weight_exp<UNIFORM>(lw > T(0) ? T(0) : lw);
  return w >= T(1) ? T(1) : w;
}

template <class T> __device__ __forceinline__ T maxabs(T a, T b) { return fabs(a) > fabs(b) ? a : b; }  // :526

// ------------------------------------------------------------------------------------------------
// cross-lane primitives inside a group of G lanes.  Butterfly all-reduce built from DPP
// (quad_perm / row_half_mirror / row_mirror), ds_swizzle (xor 16) and v_permlane32_swap (xor 32):
// no LDS traffic, no address VGPRs.  Every lane of the group ends with the same bits.
// ------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
  // every control used here reads a valid lane for every lane, so `old` is never taken: mov_dpp
  // (old = undef, bound_ctrl) saves the two v_mov that initialise `old` per exchanged f64
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(dpp_i32<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_i32<CTRL>(hi), dpp_i32<CTRL>(lo));
}
// partner = lane ^ 16 inside each half-wave (ds_swizzle bit-mode: and=0x1f, or=0, xor=0x10)
__device__ __forceinline__ float swz16(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ double swz16(double v) {
  int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
  int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
  return __hiloint2double(hi, lo);
}
// own + value of lane ^ 32.  v_permlane32_swap swaps lanes [63:32] of its first operand with
// lanes [31:0] of its second; with both = v the returned pair is {own, partner} for lanes < 32
// and {partner, own} for lanes >= 32, so p[0] + p[1] is the same sum, bit for bit, in both.
// own + value of lane ^ 16: v_permlane16_swap swaps the odd 16-lane rows of its first operand with
// the even rows of its second, so with both = v the pair holds {row 2k, row 2k+1} in every lane of
// both rows — a pure VALU exchange (ds_swizzle would go through the LDS pipeline).
__device__ __forceinline__ float xor16_sum(float v) {
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(p[0]) + __int_as_float(p[1]);
}
__device__ __forceinline__ double xor16_sum(double v) {
  auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(p[0]) + __int_as_float(p[1]);
}
__device__ __forceinline__ double xor32_sum(double v) {
  auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// The same pair all-reduce on the MATRIX pipe for a chain that owns a whole wave (G = 64).  The trajectory kernels are
// VALU-issue bound (≈95 % VALU busy) while the MFMA pipe idles; v_mfma_*_16x16x4 sums over its k index,
// which the operand layout (probed on gfx950, ahmc_dense.hpp) maps to the four 16-lane row groups of the wave:
//     a: A[i = lane%16][k = lane/16]     b: B[k = lane/16][j = lane%16]
//     f64 acc[v]: D[i = lane/16 + 4v][j]     f32 acc[v]: D[i = 4·(lane/16) + v][j]
//   1. D = A·1 with A = a: D[i][·] = Σ_k a[lane i + 16k]; every lane then adds its 4 accumulator rows: y = the sum of a
//      over a quarter of the lanes (which quarter depends on the dtype's row mapping; the four quarters tile the wave)
//   2. the same for b
//   3. D = A′·1 with A′[i][k] = (row i belongs to the "a" half ? ya : yb) of row group k: Σ_k over the four quarters —
//      the a-rows of D hold Σa, the b-rows Σb, and every lane owns one row of each (acc[0] and acc[2] / acc[1]).
// 3 MFMA + 6 adds + 2 selects instead of 30 VALU; all lanes receive the same bits (every output element goes through
// the same FMA chain on the same inputs).  The summation ORDER differs from the DPP butterfly, i.e. last-ulp
// differences against it — inside the 1e-9 parity tolerance against the oracle, which sums in index order anyway.
#ifndef AHMC_MFMA_REDUCE
#define AHMC_MFMA_REDUCE 0
#endif
template <class T> struct MfmaRed;
template <> struct MfmaRed<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static constexpr unsigned SEL = 8u;  // rows i >= 8 (acc[2], acc[3]) carry b
  static constexpr int RA = 0, RB = 2;
};
template <> struct MfmaRed<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static constexpr unsigned SEL = 1u;  // odd rows (acc[1], acc[3]) carry b
  static constexpr int RA = 0, RB = 1;
};
template <class T>
__device__ __forceinline__ void wave64_allsum2_mfma(T& a, T& b) {
  using M = MfmaRed<T>;
  const typename M::acc_t z = {T(0), T(0), T(0), T(0)};
  const T one = T(1);
  const typename M::acc_t da = M::mma(a, one, z);
  const typename M::acc_t db = M::mma(b, one, z);
  const T ya = (da[0] + da[1]) + (da[2] + da[3]);
  const T yb = (db[0] + db[1]) + (db[2] + db[3]);
  const T sel = (threadIdx.x & M::SEL) ? yb : ya;
  const typename M::acc_t d = M::mma(sel, one, z);
  a = d[M::RA];
  b = d[M::RB];
}

// all-reduce of a PAIR of values across groups of 16/32/64 lanes in ~half the instructions of two
// butterflies.  The first exchange is transposed — even lanes collect `a`, odd lanes collect `b`
// — so every later stage moves one value instead of two; rotations inside the 16-lane row keep
// the lane parity, the row pairs/halves are joined as before, and lanes 0 / 1 of the row are
// broadcast back (row_newbcast).  Every lane ends with the bits of those two lanes.
//   f64: 7 + 3 + 3 + 3 (+5 +5) + 4 = 30 VALU for G = 64, against 2 x 22.
// The exchanges of a butterfly on the LDS pipe instead of the VALU: ds_swizzle_b32 (any lane pattern inside 32 lanes)
// and ds_bpermute_b32 (lane ^ 32) move the data, the VALU only adds.  An exchange costs 2 DS instructions instead of 2
// (DPP) or 4 (permlane: 2 copies + 2 swaps) VALU instructions; the LDS pipe has its own issue port and is almost idle in
// these kernels (≈17 of ≈900 wave-cycles per leapfrog), the VALU port is the bottleneck.  The price is latency
// (≈60-100 cycles per exchange against 8): it pays only while the other waves of the SIMD fill the gap.
// AHMC_DS_REDUCE: bit 0 = the xor-16 / xor-32 stages, bit 1 = also the four in-row stages, bit 2 = the final broadcast.
#ifndef AHMC_DS_REDUCE
#define AHMC_DS_REDUCE 0
#endif
template <int PATTERN> __device__ __forceinline__ float ds_swz(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN));
}
template <int PATTERN> __device__ __forceinline__ double ds_swz(double v) {
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN), __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN));
}
__device__ __forceinline__ float ds_xor32(float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)((__lane_id() ^ 32u) << 2), __float_as_int(v)));
}
__device__ __forceinline__ double ds_xor32(double v) {
  const int addr = (int)((__lane_id() ^ 32u) << 2);
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
// bit-mode swizzle patterns: offset = (xor << 10) | (or << 5) | and; lane' = ((lane & and) | or) ^ xor within 32 lanes
constexpr int SWZ_XOR1 = (1 << 10) | 0x1F, SWZ_XOR2 = (2 << 10) | 0x1F, SWZ_XOR4 = (4 << 10) | 0x1F, SWZ_XOR8 = (8 << 10) | 0x1F, SWZ_XOR16 = (16 << 10) | 0x1F;

template <int G, class T>
__device__ __forceinline__ void wave_allsum2(T& a, T& b) {
  static_assert(G == 16 || G == 32 || G == 64, "pair reduction needs whole 16-lane rows");
  if constexpr (AHMC_MFMA_REDUCE && G == 64) {
    wave64_allsum2_mfma(a, b);
    return;
  }
  if constexpr ((AHMC_DS_REDUCE & 2) != 0) {
    // every exchange on the LDS pipe: transposed first stage (even lanes collect a, odd lanes b), xor butterflies after
    const bool odd = (threadIdx.x & 1u) != 0;
    const T keep = odd ? b : a, send = odd ? a : b;
    T x = keep + ds_swz<SWZ_XOR1>(send);
    x += ds_swz<SWZ_XOR2>(x);
    x += ds_swz<SWZ_XOR4>(x);
    x += ds_swz<SWZ_XOR8>(x);
    if constexpr (G >= 32) x += ds_swz<SWZ_XOR16>(x);
    if constexpr (G >= 64) x += ds_xor32(x);
    const T y = ds_swz<SWZ_XOR1>(x);  // the partner lane holds the other sum
    a = odd ? y : x;
    b = odd ? x : y;
    return;
  }
  const bool odd = (threadIdx.x & 1u) != 0;
  const T keep = odd ? b : a, send = odd ? a : b;
  T x = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);             // quad_perm [2,3,0,1]
  x += dpp_mov<0x124>(x);            // row_ror:4
  x += dpp_mov<0x128>(x);            // row_ror:8
  if constexpr ((AHMC_DS_REDUCE & 1) != 0) {
    if constexpr (G >= 32) x += ds_swz<SWZ_XOR16>(x);
    if constexpr (G >= 64) x += ds_xor32(x);
  } else {
    if constexpr (G >= 32) x = xor16_sum(x);
    if constexpr (G >= 64) x = xor32_sum(x);
  }
  a = dpp_mov<0x150>(x);  // row_newbcast:0
  b = dpp_mov<0x151>(x);  // row_newbcast:1
}

// the same for FOUR values: two transposed exchanges (lane & 1, then lane & 2) leave one value per lane, lanes 0..3 of
// the row are broadcast back.  14 + 7 + 3 + 3 (+5 +5) + 8 = 45 VALU for G = 64 against 2 x 30, and the additions each
// value goes through are those of wave_allsum2 (a, b as there; c, d as a second pair), so the results are bit-identical.
template <int G, class T>
__device__ __forceinline__ void wave_allsum4(T& a, T& b, T& c, T& d) {
  static_assert(G == 16 || G == 32 || G == 64, "quad reduction needs whole 16-lane rows");
  const bool odd = (threadIdx.x & 1u) != 0, up = (threadIdx.x & 2u) != 0;
  const T x = (odd ? b : a) + dpp_mov<0xB1>(odd ? a : b);  // quad_perm [1,0,3,2]
  const T y = (odd ? d : c) + dpp_mov<0xB1>(odd ? c : d);
  T z = (up ? y : x) + dpp_mov<0x4E>(up ? x : y);          // quad_perm [2,3,0,1]
  z += dpp_mov<0x124>(z);                                   // row_ror:4
  z += dpp_mov<0x128>(z);                                   // row_ror:8
  if constexpr (G >= 32) z = xor16_sum(z);
  if constexpr (G >= 64) z = xor32_sum(z);
  a = dpp_mov<0x150>(z);  // row_newbcast:0..3
  b = dpp_mov<0x151>(z);
  c = dpp_mov<0x152>(z);
  d = dpp_mov<0x153>(z);
}

// all-reduce (sum) of K independent values across the G lanes of each group
template <int G, class T, int K>
__device__ __forceinline__ void wave_allsum(T (&v)[K]) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "bad group size");
  if constexpr (G >= 16 && K == 4) {
    wave_allsum4<G>(v[0], v[1], v[2], v[3]);
    return;
  }
  if constexpr (G >= 16 && K % 2 == 0) {
#pragma unroll
    for (int k = 0; k < K; k += 2) wave_allsum2<G>(v[k], v[k + 1]);
    return;
  }
  if constexpr (G >= 2) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_mov<0xB1>(v[k]);  // quad_perm [1,0,3,2]
  }
  if constexpr (G >= 4) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x4E>(v[k]);  // quad_perm [2,3,0,1]
  }
  if constexpr (G >= 8) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x141>(v[k]);  // row_half_mirror
  }
  if constexpr (G >= 16) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x140>(v[k]);  // row_mirror
  }
  if constexpr (G >= 32) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr ((AHMC_DS_REDUCE & 1) != 0) v[k] += ds_swz<SWZ_XOR16>(v[k]); else v[k] = xor16_sum(v[k]);
    }
  }
  if constexpr (G >= 64) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr ((AHMC_DS_REDUCE & 1) != 0) v[k] += ds_xor32(v[k]); else v[k] = xor32_sum(v[k]);
    }
  }
}
// Exchange buffer of the multi-wave groups (G = 128/256/512: one chain per workgroup of G/64 waves,
// D up to 4096).  All waves of such a group execute identical control flow (every decision is taken
// on values that are bit-identical in all lanes), so the barriers below are always reached together.
__device__ __forceinline__ double* xwave_buf() {
  __shared__ __attribute__((aligned(16))) double buf[8 * 8];
  return buf;
}

// all-reduce (sum) of K values across the G lanes of each group; G > 64 = G/64 whole waves
template <int G, class T, int K>
__device__ __forceinline__ void group_allsum(T (&v)[K]) {
  if constexpr (G <= 64) {
    wave_allsum<G>(v);
  } else {
    static_assert(G == 128 || G == 256 || G == 512, "bad multi-wave group size");
    static_assert(K <= 8, "at most 8 values per reduction");
    constexpr int NW = G / 64;
    wave_allsum<64>(v);
    double* b = xwave_buf();
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) b[w * 8 + k] = (double)v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      T s = 0;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += (T)b[ww * 8 + k];  // fixed order: same bits in every wave
      v[k] = s;
    }
    __syncthreads();
  }
}
template <int G, class T>
__device__ __forceinline__ T group_sum1(T x) {
  T v[1] = {x};
  group_allsum<G>(v);
  return v[0];
}

// The same sum with ONE barrier, for a call site whose next use of ITS OWN buffer is always preceded by another barrier of the
// workgroup (then every wave has read the buffer before it is written again).  Round 4: the hierarchical-Gaussian leaf of a
// multi-wave chain took eight barriers (two broadcasts, the target's two sums, ℓπ / ℓκ: two each); with an exchange buffer per
// site — the broadcast's next write follows the sums' barrier, the sums' follows the next leaf's broadcast, the energies' follows
// both — it takes three.  Same additions in the same order as group_allsum: the bits do not change.
__device__ __forceinline__ double* xwave_buf_h() {
  __shared__ __attribute__((aligned(16))) double buf[2];
  return buf;
}
__device__ __forceinline__ double* xwave_buf_s() {
  __shared__ __attribute__((aligned(16))) double buf[8 * 2];
  return buf;
}
__device__ __forceinline__ double* xwave_buf_f() {
  __shared__ __attribute__((aligned(16))) double buf[8 * 4];
  return buf;
}
// Round 6: μ and log τ of the NEXT leaf's position, written by the chain's first lane just before the barrier of THIS leaf's energy
// exchange (hier_publish_next) and read at the top of the next leaf's target_eval (use_pre) — the broadcast exchange of its own and its
// barrier are gone for every leaf but the first of a doubling.  Lifetime: read before the next leaf's sums' barrier, written again after it.
__device__ __forceinline__ double* xwave_buf_p() {
  __shared__ __attribute__((aligned(16))) double buf[2];
  return buf;
}
template <int G, class T, int K>
__device__ __forceinline__ void group_allsum_once(T (&v)[K], double* b) {
  static_assert(G == 128 || G == 256 || G == 512, "multi-wave groups only");
  constexpr int NW = G / 64;
  wave_allsum<64>(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) b[w * K + k] = (double)v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    T s = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += (T)b[ww * K + k];  // fixed order: same bits in every wave
    v[k] = s;
  }
}
// the all-reduce that ends a leapfrog (ℓπ, ℓκ and what rides with them): one barrier where the target's own exchanges separate its uses
template <int G, int TK, class T, int K>
__device__ __forceinline__ void leapfrog_allsum(T (&v)[K]) {
  if constexpr (G > 64 && TK == 3) {
    static_assert(K <= 4, "xwave_buf_f holds four values per wave");
    group_allsum_once<G>(v, xwave_buf_f());
  } else {
    group_allsum<G>(v);
  }
}

// broadcast the value held by lane `src` (0..G-1) of the group to the whole group
template <int G>
__device__ __forceinline__ int group_bcast_i32(int v, int src) {
  int base = (int)(__lane_id() & ~(unsigned)((G > 64 ? 64 : G) - 1));
  return __builtin_amdgcn_ds_bpermute((base + src) << 2, v);
}
template <int G, class T>
__device__ __forceinline__ T xwave_bcast(T v, int src) {
  double* b = xwave_buf();
  if ((int)threadIdx.x == src) b[0] = (double)v;
  __syncthreads();
  T r = (T)b[0];
  __syncthreads();
  return r;
}
template <int G> __device__ __forceinline__ float group_bcast(float v, int src) {
  if constexpr (G > 64) return xwave_bcast<G>(v, src);
  else return __int_as_float(group_bcast_i32<G>(__float_as_int(v), src));
}
template <int G> __device__ __forceinline__ double group_bcast(double v, int src) {
  if constexpr (G > 64) return xwave_bcast<G>(v, src);
  else return __hiloint2double(group_bcast_i32<G>(__double2hiint(v), src), group_bcast_i32<G>(__double2loint(v), src));
}

// ------------------------------------------------------------------------------------------------
// built-in log-density families (include/ahmc_hip.h AHMC_TARGET_*).  `grad` receives the
// reference's cached gradient -∇ℓπ (src/hamiltonian.jl:45-48).  The return value is this lane's
// PARTIAL of ℓπ: the caller sums it over the group together with the kinetic partial, so the
// common Gaussian case costs one butterfly per leapfrog step.
// ------------------------------------------------------------------------------------------------
template <class T>
struct TargetP {
  int kind;
  int D;
  const T* params;
};

#define AHMC_LOG2PI 1.8378770664093454835606594728112

#ifdef AHMC_USER_TARGET_HEADER
// A target plugin: the user's log-density as a device function, compiled into every trajectory kernel (TK = 4).  The header
// defines, in namespace ahmc_user (contract and helpers: include/ahmc_user_target.h):
//   template <class T, int G, int E>
//   __device__ T logdensity(const T* params, int D, const T (&theta)[E], T (&grad_neg)[E], int lane, int d0);
}  // namespace ahmc
#include AHMC_USER_TARGET_HEADER
namespace ahmc {
#endif

// use_pre (multi-wave hierarchical target only): μ, log τ of this position were published by the previous leaf (xwave_buf_p)
template <class T, int G, int E, int TK>
__device__ __forceinline__ T target_eval(const TargetP<T>& tp, const T (&th)[E], T (&grad)[E], int lane, int d0, bool use_pre = false) {
  const T log2pi = (T)AHMC_LOG2PI;
  const int D = tp.D;
  T part = 0;
  {  // the target family is a compile-time parameter: a 4-way run-time switch costs 36 VGPRs in k_nuts
    if constexpr (TK == 0) {  // AHMC_TARGET_ISO_GAUSS (test/common.jl:40-44, m = 0, s = 1)
      // padded elements (d >= D) hold θ = 0 in every kernel, so they need no mask: Σθ² and g = θ are
      // already right, and the constant −D/2·log 2π goes in once (lane 0's partial)
      T ss = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ss += th[e] * th[e];
        grad[e] = th[e];
      }
      part = -ss / 2;
      if (lane == 0) part -= (T)D * log2pi / 2;
    }
    if constexpr (TK == 1) {  // AHMC_TARGET_DIAG_GAUSS: params = m[D], s[D]
#pragma unroll
      for (int e = 0; e < E; ++e) {
        // branch-free on purpose: padded elements (d >= D) read the last valid parameters and are zeroed by a select at the end.
        // (With `ok ? load : 0` the loads, the log and the divisions sat in exec-masked blocks — which is where the register
        // allocator then parked spills of values the other lanes need too: isa_check.py, round 4.)
        const bool ok = d0 + e < D;
        const int d = ok ? d0 + e : D - 1;
        const T m = tp.params[d];
        const T s = tp.params[D + d];
        const T diff = m - th[e];
        const T s2 = s * s;
        const T val = -(log2pi + 2 * log(s) + diff * diff / s2) / 2;
        const T gv = -(diff / s2);
        part += ok ? val : T(0);
        grad[e] = ok ? gv : T(0);
      }

    }
    if constexpr (TK == 2) {  // AHMC_TARGET_FUNNEL (research/notebooks/geweke_test.ipynb cell 4)
      T y = group_bcast<G>(th[0], 0);
      T ssp = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int d = d0 + e;
        ssp += (d >= 1 && d < D) ? th[e] * th[e] : T(0);
      }
      T ss = group_sum1<G>(ssp);
      T ey = exp(-y);
      T nm1 = (T)(D - 1);
      T total = -(log2pi + 2 * log(T(3)) + y * y / 9) / 2 - nm1 * (log2pi + y) / 2 - ss * ey / 2;
      part = (lane == 0) ? total : T(0);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int d = d0 + e;
        T gv = (d == 0) ? -(-y / 9 - nm1 / 2 + ss * ey / 2) : th[e] * ey;
        grad[e] = (d < D) ? gv : T(0);
      }

    }
#ifdef AHMC_USER_TARGET_HEADER
    if constexpr (TK == 4) part = ::ahmc_user::logdensity<T, G, E>(tp.params, D, th, grad, lane, d0);  // target plugin
#endif
    if constexpr (TK == 3) {  // AHMC_TARGET_HIER_GAUSS: θ = (μ, log τ, x...)
      T mu, lt;
      if constexpr (G > 64) {  // (E >= 4 here: both leading elements sit in the chain's first lane; one exchange, one barrier)
        // Invariants of the one-barrier exchanges (xwave_buf_h / _s / _f): ONE chain per workgroup of G threads — threadIdx.x == 0 is the
        // chain's first lane and wave w = threadIdx.x >> 6 (the launch plan of every multi-wave geometry, ahmc_kernels.hpp: group_grid);
        // θ[0] and θ[1] in that lane; and every leapfrog_allsum<G,3> follows a target_eval, whose barriers separate two uses of a buffer.
        static_assert(E >= 2, "multi-wave hierarchical target: mu and log tau must both sit in the chain's first lane (E >= 2)");
        if (use_pre) {   // (chain-uniform: every wave of the workgroup takes the same branch)
          const double* hp = xwave_buf_p();
          mu = (T)hp[0];
          lt = (T)hp[1];
        } else {
          double* hb = xwave_buf_h();
          if (threadIdx.x == 0) {
            hb[0] = (double)th[0];
            hb[1] = (double)th[E >= 2 ? 1 : 0];
          }
          __syncthreads();
          mu = (T)hb[0];
          lt = (T)hb[1];
        }
      } else {
        mu = group_bcast<G>(th[0], 0);
        lt = (E >= 2) ? group_bcast<G>(th[E >= 2 ? 1 : 0], 0) : group_bcast<G>(th[0], 1);
      }
      T s[2] = {0, 0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int d = d0 + e;
        T df = th[e] - mu;
        bool ok = d >= 2 && d < D;
        s[0] += ok ? df : T(0);
        s[1] += ok ? df * df : T(0);
      }
      if constexpr (G > 64) group_allsum_once<G>(s, xwave_buf_s());
      else group_allsum<G>(s);
      T itau2 = exp(-2 * lt);
      T n = (T)(D - 2);
      T total = -(log2pi + mu * mu) / 2 - (log2pi + lt * lt) / 2 - n * (log2pi + 2 * lt) / 2 - s[1] * itau2 / 2;
      part = (lane == 0) ? total : T(0);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int d = d0 + e;
        T gv;
        if (d == 0) gv = -(-mu + s[0] * itau2);
        else if (d == 1) gv = -(-lt - n + s[1] * itau2);
        else gv = (th[e] - mu) * itau2;
        grad[e] = (d < D) ? gv : T(0);
      }

    }
#ifdef AHMC_USER_TARGET_HEADER
    constexpr int TK_MAX = 4;
#else
    constexpr int TK_MAX = 3;
#endif
    if constexpr (TK < 0 || TK > TK_MAX) {
#pragma unroll
      for (int e = 0; e < E; ++e) grad[e] = Lim<T>::nan();
      part = Lim<T>::nan();
    }
  }
  return part;
}

// ------------------------------------------------------------------------------------------------
// per-chain phase point in registers and one leapfrog step (src/integrator.jl:233-243)
// ------------------------------------------------------------------------------------------------
template <class T, int E>
struct Point {
  T th[E], r[E], g[E];  // θ, r, -∇ℓπ(θ)
  T lp, lk;             // ℓπ(θ), ℓκ = -K(r)  (sanitised)
};

template <class T, int E>
__device__ __forceinline__ void copy_vec(T (&dst)[E], const T (&src)[E]) {
#pragma unroll
  for (int e = 0; e < E; ++e) dst[e] = src[e];
}

template <class T> struct LeapfrogP {
  int kind;       // AHMC_INTEGRATOR_*
  T sqrt_alpha;   // TemperedLeapfrog: sqrt(α)
};

// temper(lf, r, (i, is_half), n_steps) (src/integrator.jl:198-209)
template <class T, int E>
__device__ __forceinline__ void temper(const LeapfrogP<T>& lf, T (&r)[E], int64_t i, bool is_half, int64_t n_steps) {
  if (lf.kind != 2) return;
  int64_t i_temper = 2 * (i - 1) + 1 + (is_half ? 0 : 1);
  if (i_temper <= n_steps) {
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = r[e] * lf.sqrt_alpha;
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = r[e] / lf.sqrt_alpha;
  }
}

// kinetic partial: Σ_e M⁻¹ r² over this lane's elements (Unit: minv = 1)
template <class T, int E>
__device__ __forceinline__ T kinetic_partial(const T (&r)[E], const T (&minv)[E]) {
  T s = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) s += (r[e] * r[e]) * minv[e];
  return s;
}

// Multi-wave hierarchical target: the chain's first lane publishes θ′[0], θ′[1] of the leapfrog that would FOLLOW this one from the point
// just completed with the same signed step — r½ = r − ϵ/2·g, θ′ = θ + ϵ·(M⁻¹·r½), the very expressions (and contractions) of the next
// leapfrog_step, so the bits are the ones its own lane 0 will compute — into xwave_buf_p, before the barrier of the energy exchange.
template <class T, int G, int E, int TK>
__device__ __forceinline__ void hier_publish_next(const Point<T, E>& z, const T (&minv)[E], T eps) {
  if constexpr (G > 64 && TK == 3 && E >= 2) {
    if (threadIdx.x == 0) {
      const T eh = eps / 2;
      double* hp = xwave_buf_p();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const T rh = z.r[k] - eh * z.g[k];
        const T tn = z.th[k] + eps * (minv[k] * rh);
        hp[k] = (double)tn;
      }
    }
  }
}

// One leapfrog step of step i of n (tempering indices) with signed step size eps:
//   r -= ϵ/2 g ; θ += ϵ M⁻¹ r ; (ℓπ, g) = ∂H∂θ(θ) ; r -= ϵ/2 g ; ℓκ = -½ rᵀM⁻¹r ; sanitise
// TEMPER = false compiles the tempering out (the default NUTS kernels: the if-converted r·√α / r/√α
// select costs 6 VALU per leaf even when unused; TemperedLeapfrog runs the general instantiation)
template <class T, int G, int E, int TK, bool TEMPER = true>
__device__ __forceinline__ void leapfrog_step(Point<T, E>& z, const T (&minv)[E], T eps, const TargetP<T>& tp,
                                              const LeapfrogP<T>& lf, int lane, int d0, int64_t i, int64_t n, bool use_pre = false) {
  if constexpr (TEMPER) temper(lf, z.r, i, true, n);
  const T eh = eps / 2;
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
#pragma unroll
  for (int e = 0; e < E; ++e) z.th[e] = z.th[e] + eps * (minv[e] * z.r[e]);
  T red[2];
  red[0] = target_eval<T, G, E, TK>(tp, z.th, z.g, lane, d0, !TEMPER && use_pre);
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
  if constexpr (TEMPER) temper(lf, z.r, i, false, n);
  red[1] = kinetic_partial(z.r, minv);
  if constexpr (!TEMPER) hier_publish_next<T, G, E, TK>(z, minv, eps);
  leapfrog_allsum<G, TK>(red);
  z.lp = sanitize(red[0]);
  z.lk = sanitize(-red[1] / 2);
}

// The leaf of a NUTS tree needs ONE number from the new point: neg_energy(z′) = ℓπ + ℓκ (src/trajectory.jl:641-643) —
// the weight, the divergence test and ΔH all derive from it; ℓπ and ℓκ themselves are not read again until the
// candidate's caches are rebuilt at the end of the transition.  So the lane partials are combined BEFORE the all-reduce
// (one value, 22 VALU, instead of the pair's 34 incl. its selects) and the PhasePoint sanitation (non-finite ℓπ or ℓκ →
// −Inf, src/hamiltonian.jl:95-104) is applied to the sum: sanitize(ℓπ) + sanitize(ℓκ) is −Inf exactly when either is
// non-finite, and then ℓπ + ℓκ is non-finite too (the one exception, two finite values of ≈1e308 that overflow when
// added, does not occur for a log-density).  Returns neg_energy; z.lp / z.lk are NOT updated.
template <class T, int G, int E, int TK>
__device__ __forceinline__ T leapfrog_step_ne(Point<T, E>& z, const T (&minv)[E], T eps, const TargetP<T>& tp, int lane, int d0) {
  const T eh = eps / 2;
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
#pragma unroll
  for (int e = 0; e < E; ++e) z.th[e] = z.th[e] + eps * (minv[e] * z.r[e]);
  const T part = target_eval<T, G, E, TK>(tp, z.th, z.g, lane, d0);
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
  const T kin = kinetic_partial(z.r, minv);
  T sv[1] = {part - kin / 2};
  leapfrog_allsum<G, TK>(sv);
  const T s = sv[0];
  return is_finite(s) ? s : -Lim<T>::inf();
}

// leapfrog_step (untempered) that sums two more caller-supplied partials of the UPDATED point in the same all-reduce
// as ℓπ and the kinetic energy: the NUTS kernels pass the two U-turn dot products of the merge that follows the
// leaf, which saves one of the two reductions of such a leaf (15 of 60 VALU; a barrier pair in the multi-wave groups).
// Every value takes exactly the additions it would take in a reduction of its own, so the bits do not change.
template <class T, int G, int E, int TK, class F>
__device__ __forceinline__ void leapfrog_step_plus2(Point<T, E>& z, const T (&minv)[E], T eps, const TargetP<T>& tp, int lane, int d0,
                                                    T (&extra)[2], F&& partials, bool use_pre = false) {
  const T eh = eps / 2;
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
#pragma unroll
  for (int e = 0; e < E; ++e) z.th[e] = z.th[e] + eps * (minv[e] * z.r[e]);
  T red[4];
  red[0] = target_eval<T, G, E, TK>(tp, z.th, z.g, lane, d0, use_pre);
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
  red[1] = kinetic_partial(z.r, minv);
  partials(red[2], red[3]);
  hier_publish_next<T, G, E, TK>(z, minv, eps);
  leapfrog_allsum<G, TK>(red);
  z.lp = sanitize(red[0]);
  z.lk = sanitize(-red[1] / 2);
  extra[0] = red[2];
  extra[1] = red[3];
}

// phasepoint(h, θ, r): fill the caches at the current (θ, r) (src/hamiltonian.jl:115-119)
template <class T, int G, int E, int TK>
__device__ __forceinline__ void fill_caches(Point<T, E>& z, const T (&minv)[E], const TargetP<T>& tp, int lane, int d0) {
  T red[2];
  red[0] = target_eval<T, G, E, TK>(tp, z.th, z.g, lane, d0);
  red[1] = kinetic_partial(z.r, minv);
  leapfrog_allsum<G, TK>(red);
  z.lp = sanitize(red[0]);
  z.lk = sanitize(-red[1] / 2);
}

// vector load/store of one chain's slice: element d0+e of chain c lives at base[off + d0 + e].
// Full, 16-byte aligned lanes use dwordx4/dwordx2 accesses (CH elements per access).
template <class T, int E>
__device__ __forceinline__ void load_vec(T (&v)[E], const T* __restrict__ base, int64_t off, int d0, int D, T pad) {
  constexpr int CH = (E * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : E;
  const T* ptr = base + off + d0;
  if (CH > 1 && d0 + E <= D && (reinterpret_cast<uintptr_t>(ptr) % (CH * sizeof(T)) == 0)) {
    using V = T __attribute__((ext_vector_type(CH)));
#pragma unroll
    for (int e = 0; e < E; e += CH) {
      V t = *reinterpret_cast<const V*>(ptr + e);
#pragma unroll
      for (int k = 0; k < CH; ++k) v[e + k] = t[k];
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (d0 + e < D) ? ptr[e] : pad;
  }
}
template <class T, int E>
__device__ __forceinline__ void store_vec(const T (&v)[E], T* __restrict__ base, int64_t off, int d0, int D) {
  constexpr int CH = (E * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : E;
  T* ptr = base + off + d0;
  if (CH > 1 && d0 + E <= D && (reinterpret_cast<uintptr_t>(ptr) % (CH * sizeof(T)) == 0)) {
    using V = T __attribute__((ext_vector_type(CH)));
#pragma unroll
    for (int e = 0; e < E; e += CH) {
      V t;
#pragma unroll
      for (int k = 0; k < CH; ++k) t[k] = v[e + k];
      *reinterpret_cast<V*>(ptr + e) = t;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (d0 + e < D) ptr[e] = v[e];
  }
}

}  // namespace ahmc
