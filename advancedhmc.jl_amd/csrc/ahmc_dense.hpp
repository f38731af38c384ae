// ahmc_dense.hpp — the step-synchronous engine for DenseEuclideanMetric and the dense Gaussian
// target (src/hamiltonian.jl:60-68,179-184, src/metric.jl:89-120,311-320).
//
// ∂H∂r = M⁻¹ r and ∇ℓπ = −Pθ couple all D elements of a chain, so a chain can no longer live in the
// registers of one lane group.  What is shared instead is the D×D matrix: all chains multiply by
// the same M⁻¹ / P, i.e. a GEMM (D×D)·(D×N) on the f64/f32 MFMA units.  The engine therefore keeps
// the whole tree state of every chain in HBM and advances ALL chains by one leapfrog per "global
// step":
//     r −= ϵ/2 g → V = M⁻¹R (MFMA) → θ += ϵV → (ℓπ, g) (MFMA for the dense target) → r −= ϵ/2 g
//     → V = M⁻¹R, ℓκ = −½ r·v (MFMA) → k_d_tree: one NUTS leaf + its merges per chain
// k_d_tree is the iterative build_tree of ahmc_nuts.hpp turned into a resumable state machine (one
// wave per chain, state in DChain + vector slots).  Chains run asynchronously through a batch of
// transitions: a chain that ends a transition starts its next one in the same call (its fresh
// momentum r = U⁻¹z and v = M⁻¹r were produced for the whole batch by two GEMMs up front), so the
// only idle time is at the end of the batch.
#pragma once

#include "ahmc_kernels.hpp"
#include "ahmc_nuts.hpp"

namespace ahmc {

// ------------------------------------------------------------------------------------------------
// Y (D,N) = A (D,D) · X (D,N).  A symmetric (M⁻¹, P) or general (U⁻¹) column-major; X, Y column-major
// (a chain per column).  64×64 output tile per 256-thread workgroup, 32×32 per wave as 2×2 MFMA
// 16x16x4 tiles; K stepped by 16 through LDS with register-staged prefetch.
// MFMA 16x16x4 operand layout (probed on gfx950, scripts/probe/mfma_layout.hip):
//   a: A[i = lane%16][k = lane/16]   b: B[k = lane/16][j = lane%16]
//   f64 acc[v]: C[i = lane/16 + 4v][j = lane%16]     f32 acc[v]: C[i = 4*(lane/16) + v][j = lane%16]
// ------------------------------------------------------------------------------------------------
template <class T> struct Mfma;
template <> struct Mfma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int v) { return (lane >> 4) + 4 * v; }
};
template <> struct Mfma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int v) { return 4 * (lane >> 4) + v; }
};

#ifndef AHMC_GB_PAD
#define AHMC_GB_PAD 16
#endif
#ifndef AHMC_GB_P
#define AHMC_GB_P 2
#endif
constexpr int GB_M = 64, GB_N = 64, GB_K = 16, GB_PAD = AHMC_GB_PAD;  // row stride 80 doubles: rows k, k+1 land on disjoint LDS banks
constexpr int GB_P = AHMC_GB_P;
// Measured (D = 512, f64, µs per GEMM at N = 512 / 2048 / 4096 / 8192 columns): P=1 43/45/62/103, P=2 38/40/58/99,
// P=4 38/41/60/110; the padding makes no difference.  A lone wave issues one f64 16x16x4 MFMA per ≈61 ns and the
// whole chip sustains 48.8 TFLOP/s on independent MFMAs (scripts/probe/mfma_rate.hip; spec 78.6): at N = 8192
// this kernel runs at 43.5 TFLOP/s = 89 % of that, and a single workgroup's 32 k-steps × 16 MFMAs ≈ 38 µs is
// the floor for small N.  // software pipeline depth: tiles t+1 .. t+P are in flight (registers) while tile t is multiplied

// Pipeline: LDS holds tile t (buffer t&1); registers hold tiles t+1 … t+P-1 as they arrive and the
// loads of tile t+P are issued before tile t is multiplied, so a lone workgroup on a CU (the tail of a
// NUTS batch, when few chains are still running) still covers the ≈1-2 µs load latency with
// P·16 MFMAs ≈ 1.7 µs of matrix work.  One barrier per tile.
// Point pool (k_d_tree2): a column's vector may live in the chain's pool point ptidx[col] instead of a fixed (D,N) array — then
// its address is base + col·xcs + ptidx[col]·xps (X) / col·ycs + ptidx[col]·yps (Y, Y2): the pool is chain-major, a chain's points
// are contiguous.  Without ptidx (or with xps / yps = 0 and xcs / ycs = D): the plain (D,N) array.
template <class T>
__global__ __launch_bounds__(256) void k_dgemm(const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y, int D, int64_t N,
                                               const int* __restrict__ idx,  // idx: optional list of the N columns (chains) to process
                                               const T* __restrict__ A2, T* __restrict__ Y2,  // optional second product Y2 = A2·X in the same launch
                                               const int* __restrict__ ptidx, int64_t xps, int64_t yps, int64_t xcs, int64_t ycs) {
  __shared__ T As[2][GB_K][GB_M + GB_PAD];
  __shared__ T Bs[2][GB_K][GB_N + GB_PAD];
  using M = Mfma<T>;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
  // the row blocks of ONE column block are given the same id mod 8 — they read the same X tile from
  // the same L2 instead of eight L2s each fetching it from the Infinity Cache.  (Launched as a 1-D grid
  // of row_blocks × 8·⌈col_blocks/8⌉ workgroups.)
  const int nrb1 = (D + GB_M - 1) / GB_M;
  const int nrb = A2 ? 2 * nrb1 : nrb1;  // two products: the row blocks of A2 follow those of A (same X tile, same L2)
  const unsigned lin = blockIdx.x;
  const unsigned xcd = lin & 7u, slot = lin >> 3;
  const int64_t cb = (int64_t)(slot / nrb) * 8 + xcd;
  int rblk = (int)(slot % nrb);
  if (rblk >= nrb1) {
    rblk -= nrb1;
    A = A2;
    Y = Y2;
  }
  const int m0 = rblk * GB_M;
  const int64_t n0 = cb * GB_N;
  if (n0 >= N) return;
  const int wm = (w & 1) * 32, wn = (w >> 1) * 32;
  typename M::acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = typename M::acc_t{0, 0, 0, 0};
  T ra[GB_P][4], rb[GB_P][4];
  const int ai = (tid & 31) * 2, ak = tid >> 5;  // A tile: rows ai, ai+1 of k-rows ak and ak+8
  const int bn = tid >> 2, bk = (tid & 3) * 4;   // X tile: column bn, k-rows bk..bk+3
  const int64_t bcol = n0 + bn < N ? (idx ? (int64_t)idx[n0 + bn] : n0 + bn) : -1;
  const bool arow0 = m0 + ai < D, arow1 = m0 + ai + 1 < D;
  const T* Ap = A + (m0 + ai);
  const T* Xp = X + (bcol >= 0 ? bcol : 0) * xcs + ((ptidx && bcol >= 0) ? (int64_t)ptidx[bcol] * xps : 0);
  auto load_tile = [&](int k0, T (&a)[4], T (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = k0 + ak + 8 * q;
      a[2 * q + 0] = (k < D && arow0) ? Ap[(int64_t)k * D] : T(0);
      a[2 * q + 1] = (k < D && arow1) ? Ap[(int64_t)k * D + 1] : T(0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + bk + e;
      b[e] = (bcol >= 0 && k < D) ? Xp[k] : T(0);
    }
  };
  auto store_tile = [&](int buf, const T (&a)[4], const T (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) As[buf][ak + 8 * q][ai + e] = a[2 * q + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) Bs[buf][bk + e][bn] = b[e];
  };
  const int nk = (D + GB_K - 1) / GB_K;
  const int nk_round = (nk + GB_P - 1) / GB_P * GB_P;  // tiles past D are all zero: harmless to multiply
#pragma unroll
  for (int s = 0; s < GB_P; ++s) load_tile(s * GB_K, ra[s], rb[s]);
  store_tile(0, ra[0], rb[0]);
  __syncthreads();
  for (int kt = 0; kt < nk_round; kt += GB_P) {
#pragma unroll
    for (int s = 0; s < GB_P; ++s) {
      const int t = kt + s;  // tile t sits in LDS buffer s & 1 (GB_P is even); registers s are free again
      load_tile((t + GB_P) * GB_K, ra[s], rb[s]);
      const int buf = (GB_P % 2 == 0) ? (s & 1) : (t & 1);
#pragma unroll
      for (int ks = 0; ks < GB_K / 4; ++ks) {
        const int kq = ks * 4 + (lane >> 4), l16 = lane & 15;
        const T a0 = As[buf][kq][wm + l16], a1 = As[buf][kq][wm + 16 + l16];
        const T b0 = Bs[buf][kq][wn + l16], b1 = Bs[buf][kq][wn + 16 + l16];
        acc[0][0] = M::mma(a0, b0, acc[0][0]);
        acc[0][1] = M::mma(a0, b1, acc[0][1]);
        acc[1][0] = M::mma(a1, b0, acc[1][0]);
        acc[1][1] = M::mma(a1, b1, acc[1][1]);
      }
      // tile t+1 → the other buffer (last read while multiplying tile t-1, i.e. before the previous barrier)
      store_tile(buf ^ 1, ra[(s + 1) % GB_P], rb[(s + 1) % GB_P]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int64_t j = n0 + wn + tj * 16 + (lane & 15);
      const int64_t col = j < N ? (idx ? (int64_t)idx[j] : j) : -1;
      const int64_t yoff = col >= 0 ? col * ycs + (ptidx ? (int64_t)ptidx[col] * yps : 0) : 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = m0 + wm + ti * 16 + M::row(lane, v);
        if (row < D && col >= 0) Y[row + yoff] = acc[ti][tj][v];
      }
    }
}

// The same product for FEW columns (the tail of a NUTS batch): 64×16 output tile per workgroup, one 16×16
// MFMA tile per wave, so four times as many workgroups share the columns and the serial MFMA chain of a
// wave is 4× shorter (≈10 µs instead of ≈38 µs at D = 512).
template <class T>
__global__ __launch_bounds__(256) void k_dgemm_small(const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y, int D, int64_t N,
                                                     const int* __restrict__ idx, const T* __restrict__ A2, T* __restrict__ Y2,
                                                     const int* __restrict__ ptidx, int64_t xps, int64_t yps, int64_t xcs, int64_t ycs) {
  constexpr int BN = 16;
  __shared__ T As[2][GB_K][GB_M + GB_PAD];
  __shared__ T Bs[2][GB_K][BN + 4];
  using M = Mfma<T>;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nrb1 = (D + GB_M - 1) / GB_M;
  int rblk = blockIdx.x;
  if (rblk >= nrb1) {  // second product of the launch
    rblk -= nrb1;
    A = A2;
    Y = Y2;
  }
  const int m0 = rblk * GB_M;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  typename M::acc_t acc = typename M::acc_t{0, 0, 0, 0};
  T ra[2][4], rb[2];
  const int ai = (tid & 31) * 2, ak = tid >> 5;
  const int bn = tid >> 4, bk = tid & 15;  // X tile: column bn, k-row bk
  const int64_t bcol = n0 + bn < N ? (idx ? (int64_t)idx[n0 + bn] : n0 + bn) : -1;
  const bool arow0 = m0 + ai < D, arow1 = m0 + ai + 1 < D;
  const T* Ap = A + (m0 + ai);
  const T* Xp = X + (bcol >= 0 ? bcol : 0) * xcs + ((ptidx && bcol >= 0) ? (int64_t)ptidx[bcol] * xps : 0);
  auto load_tile = [&](int k0, T (&a)[4], T& b) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = k0 + ak + 8 * q;
      a[2 * q + 0] = (k < D && arow0) ? Ap[(int64_t)k * D] : T(0);
      a[2 * q + 1] = (k < D && arow1) ? Ap[(int64_t)k * D + 1] : T(0);
    }
    b = (bcol >= 0 && k0 + bk < D) ? Xp[k0 + bk] : T(0);
  };
  auto store_tile = [&](int buf, const T (&a)[4], T b) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) As[buf][ak + 8 * q][ai + e] = a[2 * q + e];
    Bs[buf][bk][bn] = b;
  };
  const int nk = (D + GB_K - 1) / GB_K;
  const int nk_round = (nk + 1) / 2 * 2;
  load_tile(0, ra[0], rb[0]);
  load_tile(GB_K, ra[1], rb[1]);
  store_tile(0, ra[0], rb[0]);
  __syncthreads();
  for (int kt = 0; kt < nk_round; kt += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      load_tile((kt + s + 2) * GB_K, ra[s], rb[s]);
#pragma unroll
      for (int ks = 0; ks < GB_K / 4; ++ks) {
        const int kq = ks * 4 + (lane >> 4), l16 = lane & 15;
        acc = M::mma(As[s][kq][w * 16 + l16], Bs[s][kq][l16], acc);
      }
      store_tile(s ^ 1, ra[s ^ 1], rb[s ^ 1]);
      __syncthreads();
    }
  }
  const int64_t j = n0 + (lane & 15);
  const int64_t col = j < N ? (idx ? (int64_t)idx[j] : j) : -1;
  const int64_t yoff = col >= 0 ? col * ycs + (ptidx ? (int64_t)ptidx[col] * yps : 0) : 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = m0 + w * 16 + M::row(lane, v);
    if (row < D && col >= 0) Y[row + yoff] = acc[v];
  }
}

// out[c] = sanitize(scale · Σ_d a[d,c] b[d,c]); one wave per chain
template <class T>
__global__ __launch_bounds__(256) void k_d_coldot(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, T scale, int D, int64_t N,
                                                  const int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const int64_t c = idx ? idx[j] : j;
  T s[2] = {0, 0};
  for (int d = lane; d < D; d += 64) s[0] += a[c * D + d] * b[c * D + d];
  wave_allsum2<64>(s[0], s[1]);
  if (lane == 0) out[c] = sanitize(scale * s[0]);
}

// First half of a leapfrog for every listed chain (src/integrator.jl:231-237):
//   r ← r − ϵ/2 g ;  v ← M⁻¹r   (dense: v ← v − ϵ/2 w with w = M⁻¹g carried along, so no GEMM) ;  θ ← θ + ϵ v
template <class T>
__global__ __launch_bounds__(256) void k_d_pre(T* __restrict__ th, T* __restrict__ r, const T* __restrict__ g, T* __restrict__ v, const T* __restrict__ w,
                                               const T* __restrict__ minv, int per_chain, const T* __restrict__ es, int D, int64_t N,
                                               const int* __restrict__ list) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * N) return;
  if (list) idx = (int64_t)list[idx / D] * D + idx % D;
  const T e = es[idx / D];
  if (e == T(0)) return;
  const T rh = r[idx] - e / 2 * g[idx];
  T vh;
  if (w) vh = v[idx] - e / 2 * w[idx];
  else vh = minv ? minv[per_chain ? idx : idx % D] * rh : rh;
  r[idx] = rh;
  v[idx] = vh;
  th[idx] = th[idx] + e * vh;
}
// Second half, one wave per chain (src/integrator.jl:238-243): r ← r − ϵ/2 g′ ; v likewise ; ℓκ = −½ r·v ;
// dense target: ℓπ = −½ θ·g′ (g′ = Pθ from the GEMM)
template <class T>
__global__ __launch_bounds__(256) void k_d_post(const T* __restrict__ th, T* __restrict__ r, const T* __restrict__ g, T* __restrict__ v,
                                                const T* __restrict__ w, const T* __restrict__ minv, int per_chain, const T* __restrict__ es,
                                                T* __restrict__ lp, T* __restrict__ lk, int dense_target, int D, int64_t N,
                                                const int* __restrict__ list) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const int64_t c = list ? list[j] : j;
  const T e = es[c];
  T s[2] = {0, 0};
  for (int d = lane; d < D; d += 64) {
    const int64_t i = c * D + d;
    const T gd = g[i];
    T rn = r[i], vn;
    if (e != T(0)) rn = rn - e / 2 * gd;
    if (w) vn = e != T(0) ? v[i] - e / 2 * w[i] : v[i];
    else vn = minv ? minv[per_chain ? i : d] * rn : rn;
    if (e != T(0) || !w) { r[i] = rn; v[i] = vn; }
    s[0] += rn * vn;
    s[1] += th[i] * gd;
  }
  wave_allsum2<64>(s[0], s[1]);
  if (lane == 0) {
    lk[c] = sanitize(-s[0] / 2);
    if (dense_target) lp[c] = sanitize(-s[1] / 2);
  }
}
// v = M⁻¹ ⊙ r for Unit / Diag metrics used together with the dense target (minv may be null)
template <class T>
__global__ __launch_bounds__(256) void k_d_vdiag(const T* __restrict__ r, const T* __restrict__ minv, int per_chain, T* __restrict__ v, int D, int64_t N,
                                                 const int* __restrict__ list) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * N) return;
  if (list) idx = (int64_t)list[idx / D] * D + idx % D;
  v[idx] = minv ? minv[per_chain ? idx : idx % D] * r[idx] : r[idx];
}
// r = z ./ sqrtM⁻¹ (Diag) or r = z (Unit), n vectors of the batch at once
template <class T>
__global__ __launch_bounds__(256) void k_d_rdiag(const T* __restrict__ z, const T* __restrict__ sq, int per_chain, T* __restrict__ r, int D, int64_t N, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t j = idx % ((int64_t)D * N);
  r[idx] = sq ? z[idx] / sq[per_chain ? j : j % D] : z[idx];
}
template <class T>
__global__ __launch_bounds__(256) void k_d_set(T* __restrict__ out, const T* __restrict__ in, T scale, int64_t n) {  // out = scale·in (in may be null → scale)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = in ? scale * in[idx] : scale;
}
// jitter(rng, lf) (src/integrator.jl:140-156) → ϵ_cur
template <class T>
__global__ __launch_bounds__(256) void k_d_jitter(KP<T> p) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.N) return;
  Rng rng = make_rng(p, c);
  p.eps_cur()[c] = chain_eps(p, rng, c);
}
// es[c] = 0 once the chain's point is non-finite: step() stops at the first non-finite point
// (src/integrator.jl:248-255), per chain
template <class T>
__global__ __launch_bounds__(256) void k_d_freeze(const T* __restrict__ lp, const T* __restrict__ lk, T* __restrict__ es, int64_t N) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  if (!(isfinite(lp[c]) && isfinite(lk[c]))) es[c] = T(0);
}

// ------------------------------------------------------------------------------------------------
// Per-chain tree state of the dense engine
// ------------------------------------------------------------------------------------------------
constexpr int DN_MAXLEV = 16;  // pending levels = max_depth − 1
enum { DPH_IDLE = 0, DPH_START = 1, DPH_RUN = 2, DPH_WARM = 3 };  // WARM: one motionless step that computes W = M⁻¹g at the start point
// vector slots: T[slot][N][D]
enum {
  DS_CUR_V = 0, DS_CUR_W,  // v = M⁻¹r and w = M⁻¹g of the moving edge (w: dense metric only)
  DS_OTH_W,
  DS_OTH_TH, DS_OTH_R, DS_OTH_G, DS_OTH_V,
  DS_TREE_RHO,
  DS_CAND_TH, DS_CAND_R, DS_CAND_G,
  DS_SUB_RHO,
  DS_START_TH, DS_START_R, DS_START_G,  // static HMC: the start point (for rejected proposals)
  DS_FIXED,
  DS_PER_LEVEL = 5  // ρ, v_first, candidate θ, r, g
};

template <class T>
struct DChain {
  T H0, eps, w_tree, sa_tree, dh_tree, lu;
  T cand_lp, cand_lk, sub_lp, sub_lk;  // energies of the tree-level / current-subtree candidates
  T w_c, sa_c, dh_c;
  T pw[DN_MAXLEV], psa[DN_MAXLEV], pdh[DN_MAXLEV], plp[DN_MAXLEV], plk[DN_MAXLEV];
  int32_t pna[DN_MAXLEV];
  int32_t phase, it, jw, leaf, v, cur_is_left, na_tree, na_c, depth, numerical;
  uint32_t k;  // sequential draws consumed in this transition
};

template <class T>
struct DP {  // dense-engine arguments (beside KP)
  T* W;            // vector slots
  DChain<T>* S;    // per-chain state
  T* es;           // (N,) signed step of the next global step (0 = idle)
  const T* RB;     // (n_trans, N, D) fresh momenta of the batch
  const T* VB;     // (n_trans, N, D) M⁻¹ · RB
  int n_trans;
  int* n_active;   // device counter: chains that have not finished the batch
  const int* list; // chains still running (compacted every few steps), or null = all
  int64_t n_list;
  int dense_metric;
};

// stream compaction of the running chains: out = { c in in[0..n) : phase(c) != idle }
template <class T>
__global__ __launch_bounds__(256) void k_d_compact(const DChain<T>* __restrict__ S, const int* __restrict__ in, int64_t n, int* __restrict__ out,
                                                   int* __restrict__ count) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int c = in ? in[j] : (int)j;
  if (S[c].phase != DPH_IDLE) out[atomicAdd(count, 1)] = c;
}

template <class T>
__global__ __launch_bounds__(256) void k_d_iota(int* __restrict__ out, int start, int64_t n) {  // out[j] = start + j
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = start + (int)j;
}

template <class T>
__device__ __forceinline__ T* dslot(const DP<T>& q, const KP<T>& p, int slot, int64_t c) {
  return q.W + ((int64_t)slot * p.N + c) * p.D;
}
template <class T>
__device__ __forceinline__ void vcopy(T* __restrict__ dst, const T* __restrict__ src, int D, int lane) {
  for (int d = lane; d < D; d += 64) dst[d] = src[d];
}

// One call = for every chain that is running: account for the leapfrog that has just completed
// (leaf, merges, end of subtree, end of doubling, end of transition, start of the next transition)
// and publish the signed step of its next leapfrog.  MultinomialTS / SliceTS with
// GeneralisedNoUTurn; log-domain weights as the reference (src/trajectory.jl:144-206,626-742).
// Threads per chain in the tree kernel (template parameter DT).  Measured on cfg4 (D = 512): 128 -> 2.21e7, 256 -> 2.25e7,
// 512 -> 2.14e7 leapfrog/s (one wave per chain: 2.01e7).  Round 2: the thread count follows D — a chain of D <= 128 gets
// ONE wave (the workgroup barriers become wave-local, no idle waves), D <= 256 two: the step-synchronous engine also
// serves user log-densities of any dimension (ahmc_ext_*), where a request at D = 128 paid 127 µs for a kernel shaped
// for D = 512.
constexpr int DT_THREADS = 256;
inline int dt_threads_for(int64_t D) {
  static const int ov = getenv("AHMC_DENSE_DT") ? atoi(getenv("AHMC_DENSE_DT")) : 0;  // experiments: 64 / 128 / 256 threads per chain
  if (ov == 64 || ov == 128 || ov == 256) return ov;
  // round 3, cfg4 (D = 512, two pipelines, point-pool kernel), whole loop: 256 threads 2.54e7, 128 threads 2.67e7, 64 threads 2.55e7
  return D <= 128 ? 64 : (D <= 512 ? 128 : DT_THREADS);
}
// all-reduce of a pair over the DT_THREADS threads of the workgroup (every decision of d_tree_advance is taken on
// such sums or on per-chain scalars, so all threads follow the same control flow and reach the barriers together)
template <int DT, class T>
__device__ __forceinline__ void block_allsum2(T& a, T& b) {
  if constexpr (DT <= 64) {  // a chain inside one wave (DT = 16 / 32: several chains per wave, k_dense_epoch): no cross-wave exchange, no barrier
    wave_allsum2<DT>(a, b);
  } else {
    __shared__ double xb[DT / 64][2];
    wave_allsum2<64>(a, b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { xb[w][0] = (double)a; xb[w][1] = (double)b; }
    __syncthreads();
    T sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < DT / 64; ++k) { sa += (T)xb[k][0]; sb += (T)xb[k][1]; }  // fixed order: same bits in every wave
    __syncthreads();
    a = sa;
    b = sb;
  }
}

// One pass over NV vectors of a chain whose DC elements (compile time) are spread over a group of DT lanes, lane l holding the
// pairs (i·DT + l)·2, +1: the loads of a chunk of up to 8 pairs per lane (4 with five vectors) are ALL issued before the first element is used (and
// before any store of the pass, which the compiler could not prove not to alias), so a pass costs one memory round trip per
// chunk — the loops over a run-time D take one per element.  f(d, x): the element pair at d, x[v] its values in vector v.
template <class T, int DT, int DC, int NV, int CHCAP = 8, class F>
__device__ __forceinline__ void dn_vec_pass(int lane, const T* const (&src)[NV], F&& f) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  // CHCAP: pairs per lane and vector in flight at most (8: ≤ 128 registers of operands with four vectors; k_dense_epoch2 compiled for
  // four waves per SIMD passes 4).  Round 6: NP need not be a multiple of the chunk — the last chunk is shorter (D = 384: 12 = 8 + 4);
  // a lane still meets its elements in ascending order, so the sums have the bits of any other chunking.
  constexpr int NP = DC / (2 * DT), CHM = NV >= 5 ? CHCAP / 2 : CHCAP, CH = NP < CHM ? NP : CHM;
  static_assert(NP >= 1 && DC % (2 * DT) == 0, "dn_vec_pass: DC must be a multiple of 2·DT");
#pragma unroll
  for (int c0 = 0; c0 < NP; c0 += CH) {
    T2 val[CH][NV];
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (c0 + i < NP) {
#pragma unroll
        for (int v = 0; v < NV; ++v) val[i][v] = *reinterpret_cast<const T2*>(src[v] + ((c0 + i) * DT + lane) * 2);
      }
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (c0 + i < NP) f(((c0 + i) * DT + lane) * 2, val[i]);
  }
}

// Returns the signed step of the chain's next leapfrog (0 = motionless step or idle).  lp_in / lk_in: ℓπ, ℓκ of the
// point the leapfrog that has just completed arrived at (registers, uniform across the wave).
// CRIT: the termination criterion (AHMC_TC_*).  1 = GeneralisedNoUTurn (:566-570), the default and the only one the
// kernel k_d_tree is built for; 0 = ClassicNoUTurn (:551-557): the "ρ" slots hold θ of the first-built leaf instead
// of a momentum sum; 2 = StrictGeneralisedNoUTurn (:579-617): three more vectors per pending level (r of the
// first-built leaf, r and v of the last-built one) and r, v of the edge the current subtree grows from.
template <int CRIT>
struct DLevel {
  static constexpr int STRIDE = CRIT == 2 ? DS_PER_LEVEL + 3 : DS_PER_LEVEL;  // vector slots per pending level
};
template <class T, int CRIT = 1, int DT = DT_THREADS>
__device__ __forceinline__ T d_tree_advance(const KP<T>& p, const DP<T>& q, int64_t c, int lane /* thread of the chain's workgroup, 0 .. DT-1 */, T lp_in, T lk_in) {
  constexpr int DT_THREADS = DT;  // (shadows the default: every loop below strides by the workgroup's size)
  constexpr int LS = DLevel<CRIT>::STRIDE;
  DChain<T>& S = q.S[c];
  const int D = p.D;
  const bool slice = p.sampler == 2;
  T* th = p.th() + c * D;
  T* r = p.r() + c * D;
  T* g = p.g() + c * D;
  T* V = dslot(q, p, DS_CUR_V, c);
  Rng rng = make_rng(p, c);
  DrawStream ds;
  auto resume_draws = [&](uint32_t it, uint32_t k) {
    rng.iter = p.iteration + it;
    ds.resume(rng, k);
  };
  // scalars are uniform across the wave: every lane computes them; lane 0 writes them back
  int phase = S.phase, it = S.it;
  if (phase == DPH_WARM) {
    // the motionless step has produced W = M⁻¹g at the start point: remember it for the other edge
    // too and let the first leapfrog go
    const T* Wc = dslot(q, p, DS_CUR_W, c);
    T* o_w = dslot(q, p, DS_OTH_W, c);
    for (int d = lane; d < D; d += DT_THREADS) o_w[d] = Wc[d];
    const T e_first = S.v < 0 ? -S.eps : S.eps;
    __syncthreads();  // (every thread has read the phase before thread 0 changes it)
    if (lane == 0) S.phase = DPH_RUN;
    return e_first;
  }
  bool start = phase == DPH_START;
  T lp_start = lp_in;  // ℓπ(θ) of the point the next transition starts from
  if (!start) {
    resume_draws((uint32_t)it, S.k);
    const T H0 = S.H0, eps = S.eps;
    const int v = S.v, jw = S.jw;
    int leaf = S.leaf;
    const uint32_t nleaf = 1u << jw;
    const T lp = lp_in, lk = lk_in;
    // ---- leaf (:638-647) ----
    const T ne = lp + lk;
    const T dH = -ne - H0;
    T sa_c = exp(jl_min(T(0), -dH)), dh_c = dH, w_c;
    int na_c = 1;
    bool sub_term;
    if (slice) {
      w_c = (S.lu <= ne) ? T(1) : T(0);
      sub_term = !(S.lu < p.delta_max + ne);
    } else {
      w_c = H0 + ne;
      sub_term = !(-H0 < p.delta_max + ne);
    }
    bool numerical = S.numerical != 0 || sub_term;
    const int cur_is_left_in = S.cur_is_left;
    // every thread has now read the chain's scalars of this call: only from here on may thread 0 update them
    // (an odd leaf parks without any reduction, i.e. without any other barrier in between)
    __syncthreads();
    // The subtree being assembled lives only inside this call, so its vectors are VIEWS: a fresh leaf's
    // ρ, v_first and candidate are the moving edge itself; after a merge ρ is in the SUB_RHO slot and
    // v_first / the candidate may point into a pending level.  Data is copied only when it must
    // outlive the call (park, tree-level candidate).
    T* s_rho = dslot(q, p, DS_SUB_RHO, c);
    const T* rho_v = CRIT == 0 ? th : r;  // Classic: θ of the subtree's first-built leaf
    const T* vf_v = V;
    const T* rf_v = r;  // Strict: r of the subtree's first-built leaf
    const T *cth_v = th, *cr_v = r, *cg_v = g;
    T sub_lp = lp, sub_lk = lk;
    // ---- merges: one per trailing zero bit of `leaf` (:649-673) ----
    const int nm = __builtin_ctz((uint32_t)leaf);
    int merged = 0;
    for (int lvl = 0; lvl < nm && !sub_term; ++lvl) {
      T* p_rho = dslot(q, p, DS_FIXED + LS * lvl + 0, c);
      T* p_vf = dslot(q, p, DS_FIXED + LS * lvl + 1, c);
      const T w_p = S.pw[lvl];
      bool keep_first;
      T w_new;
      if (slice) {
        w_new = w_p + w_c;
        keep_first = w_new * (T)ds.uniform() < w_p;
      } else {
        w_new = logaddexp(w_p, w_c);
        keep_first = w_new < w_p + (T)ds.randexp();
      }
      if (keep_first) {
        cth_v = dslot(q, p, DS_FIXED + LS * lvl + 2, c);
        cr_v = dslot(q, p, DS_FIXED + LS * lvl + 3, c);
        cg_v = dslot(q, p, DS_FIXED + LS * lvl + 4, c);
        sub_lp = S.plp[lvl];
        sub_lk = S.plk[lvl];
      }
      w_c = w_new;
      sa_c = S.psa[lvl] + sa_c;
      na_c = S.pna[lvl] + na_c;
      const T dh_p = S.pdh[lvl];
      dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
      if constexpr (CRIT == 1) {
        // ρ = ρ_first + ρ_second; generalised_uturn_criterion with v = M⁻¹r at the two ends (:566-570,619-621)
        T dots[2] = {0, 0};
        for (int d = lane; d < D; d += DT_THREADS) {
          const T rho = p_rho[d] + rho_v[d];
          dots[0] += rho * p_vf[d];
          dots[1] += rho * V[d];
          s_rho[d] = rho;
        }
        rho_v = s_rho;
        vf_v = p_vf;
        block_allsum2<DT>(dots[0], dots[1]);
        sub_term = (dots[0] <= 0) || (dots[1] <= 0);
      } else if constexpr (CRIT == 0) {
        // ClassicNoUTurn (:551-557): ends = the pending half's first-built leaf (θ in the ρ slot, v_first) and the
        // current leaf; Δθ = θ_right − θ_left; terminated if Δθ·M⁻¹(−r_left) >= 0 or −Δθ·M⁻¹r_right >= 0
        T dots[2] = {0, 0};
        for (int d = lane; d < D; d += DT_THREADS) {
          const T thl = v > 0 ? p_rho[d] : th[d], thr = v > 0 ? th[d] : p_rho[d];
          const T vl = v > 0 ? p_vf[d] : V[d], vr = v > 0 ? V[d] : p_vf[d];
          const T dth = thr - thl;
          dots[0] += dth * (-vl);
          dots[1] += (-dth) * vr;
        }
        rho_v = p_rho;  // the merged subtree's first-built leaf is the pending half's
        vf_v = p_vf;
        block_allsum2<DT>(dots[0], dots[1]);
        sub_term = (dots[0] >= 0) || (dots[1] >= 0);
      } else {
        // StrictGeneralisedNoUTurn (:579-617).  F = the pending (first-built) half, S = the half just completed:
        //   (ρ_F + ρ_S ; ends F.first, S.last)   (ρ_F + r_S.first ; ends F.first, S.first)   (r_F.last + ρ_S ; ends F.last, S.last)
        const T* p_rf = dslot(q, p, DS_FIXED + LS * lvl + 5, c);
        const T* p_rl = dslot(q, p, DS_FIXED + LS * lvl + 6, c);
        const T* p_vl = dslot(q, p, DS_FIXED + LS * lvl + 7, c);
        T dots[6] = {0, 0, 0, 0, 0, 0};
        for (int d = lane; d < D; d += DT_THREADS) {
          const T rho = p_rho[d] + rho_v[d];
          const T rho2 = p_rho[d] + rf_v[d];
          const T rho3 = p_rl[d] + rho_v[d];
          dots[0] += rho * p_vf[d];
          dots[1] += rho * V[d];
          dots[2] += rho2 * p_vf[d];
          dots[3] += rho2 * vf_v[d];
          dots[4] += rho3 * p_vl[d];
          dots[5] += rho3 * V[d];
          s_rho[d] = rho;
        }
        rho_v = s_rho;
        vf_v = p_vf;
        rf_v = p_rf;
        block_allsum2<DT>(dots[0], dots[1]);
        block_allsum2<DT>(dots[2], dots[3]);
        block_allsum2<DT>(dots[4], dots[5]);
        sub_term = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
      }
      merged = lvl + 1;
    }
    bool subtree_over = true;
    if (sub_term) {
      // enclosing unfinished subtrees still absorb the statistics of their first halves (:666)
      const uint32_t pend = (((uint32_t)leaf - 1u) >> merged) << merged;
      for (int qq = merged; (pend >> qq) != 0u; ++qq) {
        if ((pend >> qq) & 1u) {
          sa_c = S.psa[qq] + sa_c;
          na_c = S.pna[qq] + na_c;
          const T dh_p = S.pdh[qq];
          dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
        }
      }
    } else if ((uint32_t)leaf < nleaf) {
      // park the finished level-nm subtree until its sibling is built
      T* p_rho = dslot(q, p, DS_FIXED + LS * nm + 0, c);
      T* p_vf = dslot(q, p, DS_FIXED + LS * nm + 1, c);
      T* p_cth = dslot(q, p, DS_FIXED + LS * nm + 2, c);
      T* p_cr = dslot(q, p, DS_FIXED + LS * nm + 3, c);
      T* p_cg = dslot(q, p, DS_FIXED + LS * nm + 4, c);
      for (int d = lane; d < D; d += DT_THREADS) {
        p_rho[d] = rho_v[d];
        p_vf[d] = vf_v[d];
        p_cth[d] = cth_v[d];
        p_cr[d] = cr_v[d];
        p_cg[d] = cg_v[d];
      }
      if constexpr (CRIT == 2) {  // r of the first-built leaf, r and v of the last-built one (the current leaf)
        T* p_rf = dslot(q, p, DS_FIXED + LS * nm + 5, c);
        T* p_rl = dslot(q, p, DS_FIXED + LS * nm + 6, c);
        T* p_vl = dslot(q, p, DS_FIXED + LS * nm + 7, c);
        for (int d = lane; d < D; d += DT_THREADS) {
          p_rf[d] = rf_v[d];
          p_rl[d] = r[d];
          p_vl[d] = V[d];
        }
      }
      if (lane == 0) {
        S.pw[nm] = w_c;
        S.psa[nm] = sa_c;
        S.pdh[nm] = dh_c;
        S.pna[nm] = na_c;
        S.plp[nm] = sub_lp;
        S.plk[nm] = sub_lk;
        S.leaf = leaf + 1;
        S.numerical = numerical ? 1 : 0;
        S.k = ds.k;
      }
      subtree_over = false;  // next leapfrog: same edge, same direction (es unchanged)
    }
    if (!subtree_over) return v > 0 ? eps : -eps;

    // ---- top level of the doubling loop (:708-722) ----
    T w_tree = S.w_tree, sa_tree = S.sa_tree, dh_tree = S.dh_tree;
    int na_tree = S.na_tree, depth = S.depth;
    T cand_lp = S.cand_lp, cand_lk = S.cand_lk;
    if (!sub_term) {
      ++depth;
      bool acc;  // mh_accept(rng, sampler, sampler′): biased progressive sampling (:202-206)
      if (slice) acc = w_tree * (T)ds.uniform() < w_c;
      else acc = w_tree < w_c + (T)ds.randexp();
      if (acc) {
        T* c_th = dslot(q, p, DS_CAND_TH, c);
        T* c_r = dslot(q, p, DS_CAND_R, c);
        T* c_g = dslot(q, p, DS_CAND_G, c);
        for (int d = lane; d < D; d += DT_THREADS) {
          c_th[d] = cth_v[d];
          c_r[d] = cr_v[d];
          c_g[d] = cg_v[d];
        }
        cand_lp = sub_lp;
        cand_lk = sub_lk;
      }
    }
    sa_tree = sa_tree + sa_c;
    na_tree = na_tree + na_c;
    dh_tree = v < 0 ? maxabs(dh_c, dh_tree) : maxabs(dh_tree, dh_c);
    w_tree = slice ? w_tree + w_c : logaddexp(w_tree, w_c);
    // isterminated on the whole tree; its edges are `cur` and the dormant one
    bool turn;
    if constexpr (CRIT == 1) {
      T* t_rho = dslot(q, p, DS_TREE_RHO, c);
      const T* o_v = dslot(q, p, DS_OTH_V, c);
      T dots[2] = {0, 0};
      for (int d = lane; d < D; d += DT_THREADS) {
        const T rho = t_rho[d] + rho_v[d];
        dots[0] += rho * V[d];
        dots[1] += rho * o_v[d];
        t_rho[d] = rho;
      }
      block_allsum2<DT>(dots[0], dots[1]);
      turn = (dots[0] <= 0) || (dots[1] <= 0);
    } else if constexpr (CRIT == 0) {
      const T* o_th = dslot(q, p, DS_OTH_TH, c);
      const T* o_v = dslot(q, p, DS_OTH_V, c);
      const bool cl = cur_is_left_in != 0;
      T dots[2] = {0, 0};
      for (int d = lane; d < D; d += DT_THREADS) {
        const T thl = cl ? th[d] : o_th[d], thr = cl ? o_th[d] : th[d];
        const T vl = cl ? V[d] : o_v[d], vr = cl ? o_v[d] : V[d];
        const T dth = thr - thl;
        dots[0] += dth * (-vl);
        dots[1] += (-dth) * vr;
      }
      block_allsum2<DT>(dots[0], dots[1]);
      turn = (dots[0] >= 0) || (dots[1] >= 0);
    } else {
      // strict at the top: (ρ_tree + ρ_sub ; ends current edge, other edge), (ρ_tree + r_sub.first ; ends other edge,
      // sub.first) and (r_start + ρ_sub ; ends start edge, current edge); start edge = the one the subtree grew from
      T* t_rho = dslot(q, p, DS_TREE_RHO, c);
      const T* o_v = dslot(q, p, DS_OTH_V, c);
      const T* rs = dslot(q, p, DS_START_R, c);
      const T* vs = dslot(q, p, DS_START_G, c);
      T dots[6] = {0, 0, 0, 0, 0, 0};
      for (int d = lane; d < D; d += DT_THREADS) {
        const T rho = t_rho[d] + rho_v[d];
        const T rho2 = t_rho[d] + rf_v[d];
        const T rho3 = rs[d] + rho_v[d];
        dots[0] += rho * V[d];
        dots[1] += rho * o_v[d];
        dots[2] += rho2 * o_v[d];
        dots[3] += rho2 * vf_v[d];
        dots[4] += rho3 * vs[d];
        dots[5] += rho3 * V[d];
        t_rho[d] = rho;
      }
      block_allsum2<DT>(dots[0], dots[1]);
      block_allsum2<DT>(dots[2], dots[3]);
      block_allsum2<DT>(dots[4], dots[5]);
      turn = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
    }
    const bool done = sub_term || turn || (jw + 1 >= p.max_depth);
    if (!done) {
      // ---- next doubling: direction (:693), edge selection ----
      const bool vleft = ds.boolean();
      const bool cur_is_left = cur_is_left_in != 0;
      if (vleft != cur_is_left) {  // continue from the other edge: swap the two edge points
        T* o_th = dslot(q, p, DS_OTH_TH, c);
        T* o_r = dslot(q, p, DS_OTH_R, c);
        T* o_g = dslot(q, p, DS_OTH_G, c);
        T* o_v = dslot(q, p, DS_OTH_V, c);
        T* o_w = dslot(q, p, DS_OTH_W, c);
        T* Wc = dslot(q, p, DS_CUR_W, c);
        for (int d = lane; d < D; d += DT_THREADS) {
          T t;
          t = o_th[d]; o_th[d] = th[d]; th[d] = t;
          t = o_r[d]; o_r[d] = r[d]; r[d] = t;
          t = o_g[d]; o_g[d] = g[d]; g[d] = t;
          t = o_v[d]; o_v[d] = V[d]; V[d] = t;
          if (q.dense_metric) { t = o_w[d]; o_w[d] = Wc[d]; Wc[d] = t; }
        }
      }
      if constexpr (CRIT == 2) {  // r, v of the edge the next subtree grows from (each thread re-reads its own elements)
        T* rs = dslot(q, p, DS_START_R, c);
        T* vs = dslot(q, p, DS_START_G, c);
        for (int d = lane; d < D; d += DT_THREADS) {
          rs[d] = r[d];
          vs[d] = V[d];
        }
      }
      if (lane == 0) {
        S.w_tree = w_tree; S.sa_tree = sa_tree; S.dh_tree = dh_tree; S.na_tree = na_tree; S.depth = depth;
        S.cand_lp = cand_lp; S.cand_lk = cand_lk;
        S.numerical = numerical ? 1 : 0;
        S.cur_is_left = vleft ? 1 : 0;
        S.v = vleft ? -1 : 1;
        S.jw = jw + 1;
        S.leaf = 1;
        S.k = ds.k;
      }
      return vleft ? -eps : eps;
    }
    // ---- Transition(zcand, stats) (:725-741) ----
    {
      const T* c_th = dslot(q, p, DS_CAND_TH, c);
      const T* c_r = dslot(q, p, DS_CAND_R, c);
      const T* c_g = dslot(q, p, DS_CAND_G, c);
      T* s1 = p.acc_sum() + c * D;
      T* s2 = p.acc_sumsq() + c * D;
      T* so = p.samples_out ? p.samples_out + ((int64_t)it * p.N + c) * D : nullptr;
      for (int d = lane; d < D; d += DT_THREADS) {
        const T t = c_th[d];
        th[d] = t;
        r[d] = c_r[d];
        g[d] = c_g[d];
        if (p.accum) { s1[d] += t; s2[d] += t * t; }
        if (so) so[d] = t;
      }
      if (lane == 0) {
        const T H = -(cand_lp + cand_lk);
        p.lp()[c] = cand_lp;
        p.lk()[c] = cand_lk;
        p.eps_cur()[c] = eps;
        p.st_nsteps()[c] = na_tree;
        p.st_accept()[c] = 1;
        p.st_accrate()[c] = sa_tree / (T)na_tree;
        p.st_logdens()[c] = cand_lp;
        p.st_H()[c] = H;
        p.st_Herr()[c] = H - H0;
        p.st_maxHerr()[c] = dh_tree;
        p.st_depth()[c] = depth;
        p.st_numerr()[c] = numerical ? 1 : 0;
        if (p.accum) {
          p.acc_nsteps()[c] += na_tree;
          p.acc_ndiv()[c] += numerical ? 1 : 0;
          accumulate_energy(p, c, H);
        }
      }
      ++it;
      if (it >= q.n_trans) {
        if (lane == 0) {
          S.phase = DPH_IDLE;
          S.it = it;
          atomicSub(q.n_active, 1);
        }
        return T(0);
      }
      start = true;
      lp_start = cand_lp;  // (lane 0 has just stored it; the other lanes must not re-read it)
    }
  }
  // ---- start of transition `it` (src/sampler.jl:54-57, src/trajectory.jl:677-690): the state is the
  // previous candidate (θ, g, ℓπ already in place); fresh momentum and v = M⁻¹r from the batch ----
  {
    resume_draws((uint32_t)it, 0u);
    const T eps = chain_eps(p, rng, c);
    const T* rb = q.RB + ((int64_t)it * p.N + c) * D;
    const T* vb = q.VB + ((int64_t)it * p.N + c) * D;
    T* o_th = dslot(q, p, DS_OTH_TH, c);
    T* o_r = dslot(q, p, DS_OTH_R, c);
    T* o_g = dslot(q, p, DS_OTH_G, c);
    T* o_v = dslot(q, p, DS_OTH_V, c);
    T* t_rho = dslot(q, p, DS_TREE_RHO, c);
    T* c_th = dslot(q, p, DS_CAND_TH, c);
    T* c_r = dslot(q, p, DS_CAND_R, c);
    T* c_g = dslot(q, p, DS_CAND_G, c);
    T dots[2] = {0, 0};
    for (int d = lane; d < D; d += DT_THREADS) {
      const T rd = rb[d], vd = vb[d], td = th[d], gd = g[d];
      dots[0] += rd * vd;
      r[d] = rd;
      V[d] = vd;
      o_th[d] = td; o_r[d] = rd; o_g[d] = gd; o_v[d] = vd;
      t_rho[d] = rd;
      c_th[d] = td; c_r[d] = rd; c_g[d] = gd;
      if constexpr (CRIT == 2) {
        dslot(q, p, DS_START_R, c)[d] = rd;
        dslot(q, p, DS_START_G, c)[d] = vd;
      }
    }
    block_allsum2<DT>(dots[0], dots[1]);
    const T lp = lp_start;
    const T lk = sanitize(-dots[0] / 2);
    const T H0 = -(lp + lk);
    T lu = 0, w_tree;
    if (slice) {
      lu = -H0 - (T)ds.randexp();  // SliceTS(rng, z0) (:144-145)
      w_tree = 1;
    } else {
      w_tree = 0;  // MultinomialTS(rng, z0): ℓw = 0 (:155)
    }
    const bool vleft = ds.boolean();
    if (lane == 0) {
      p.lk()[c] = lk;
      S.H0 = H0; S.eps = eps; S.lu = lu; S.w_tree = w_tree; S.sa_tree = 0; S.dh_tree = 0; S.na_tree = 0;
      S.cand_lp = lp; S.cand_lk = lk;
      S.depth = 0; S.numerical = 0; S.jw = 0; S.leaf = 1;
      S.cur_is_left = vleft ? 1 : 0;
      S.v = vleft ? -1 : 1;
      S.k = ds.k;
      S.it = it;
      S.phase = q.dense_metric ? DPH_WARM : DPH_RUN;
    }
    return q.dense_metric ? T(0) : (vleft ? -eps : eps);
  }
}

// One global step of the NUTS batch for every listed chain, fused (one wave per chain):
//   second half of the leapfrog that the GEMMs have just served (k_d_post) → d_tree_advance → first half of the
//   next leapfrog (k_d_pre).  `do_post` = 0 for the very first call of a batch (no leapfrog in flight yet).
// One chain per workgroup of 256 threads (2 elements per thread at D = 512): the kernel is a chain of dependent
// memory round trips, so more threads per chain = fewer trips (one wave per chain: 75 µs per call, this: see DESIGN).
template <class T, int DT = DT_THREADS>
__global__ __launch_bounds__(DT) void k_d_tree(KP<T> p, DP<T> q, const T* __restrict__ minv, int per_chain, int dense_target, int do_post) {
  constexpr int DT_THREADS = DT;
  const int lane = threadIdx.x;  // one chain per workgroup of DT threads
  const int64_t j = blockIdx.x;
  if (j >= q.n_list) return;
  const int64_t c = q.list ? q.list[j] : j;
  if (q.S[c].phase == DPH_IDLE) return;
  const int D = p.D;
  T* th = p.th() + c * D;
  T* r = p.r() + c * D;
  T* g = p.g() + c * D;
  T* V = dslot(q, p, DS_CUR_V, c);
  T* W = q.dense_metric ? dslot(q, p, DS_CUR_W, c) : nullptr;
  T lp = p.lp()[c], lk = p.lk()[c];
  if (do_post) {
    const T e = q.es[c];
    T s[2] = {0, 0};
    for (int d = lane; d < D; d += DT_THREADS) {
      const T gd = g[d];
      T rn = r[d], vn;
      if (e != T(0)) rn = rn - e / 2 * gd;
      if (W) vn = e != T(0) ? V[d] - e / 2 * W[d] : V[d];
      else vn = minv ? minv[per_chain ? c * D + d : d] * rn : rn;
      if (e != T(0) || !W) { r[d] = rn; V[d] = vn; }
      s[0] += rn * vn;
      s[1] += th[d] * gd;
    }
    block_allsum2<DT>(s[0], s[1]);
    lk = sanitize(-s[0] / 2);
    if (dense_target) lp = sanitize(-s[1] / 2);
    if (lane == 0) {
      p.lk()[c] = lk;
      if (dense_target) p.lp()[c] = lp;
    }
  }
  const T e = d_tree_advance<T, 1, DT>(p, q, c, lane, lp, lk);
  if (lane == 0) q.es[c] = e;
  if (e != T(0)) {  // first half of the next leapfrog (src/integrator.jl:231-237)
    for (int d = lane; d < D; d += DT_THREADS) {
      const T rh = r[d] - e / 2 * g[d];
      const T vh = W ? V[d] - e / 2 * W[d] : (minv ? minv[per_chain ? c * D + d : d] * rh : rh);
      r[d] = rh;
      V[d] = vh;
      th[d] = th[d] + e * vh;
    }
  }
}

// the same global step with ClassicNoUTurn (CRIT = 0) / StrictGeneralisedNoUTurn (CRIT = 2) and / or TemperedLeapfrog
// (TEMPER): k_d_tree's body around d_tree_advance<T, CRIT> (kept as a second kernel so that the code of the default
// one does not move).  A NUTS leaf is step(lf, h, z, 1): temper multiplies r by √α before the first half-step and
// divides it by √α after the second (src/integrator.jl:198-209 with n_steps = 1); v = M⁻¹r goes with it.
template <class T, int CRIT, bool TEMPER, int DT = DT_THREADS>
__global__ __launch_bounds__(DT) void k_d_tree_crit(KP<T> p, DP<T> q, const T* __restrict__ minv, int per_chain, int dense_target, int do_post) {
  constexpr int DT_THREADS = DT;
  const int lane = threadIdx.x;  // one chain per workgroup of DT threads
  const int64_t j = blockIdx.x;
  if (j >= q.n_list) return;
  const int64_t c = q.list ? q.list[j] : j;
  if (q.S[c].phase == DPH_IDLE) return;
  const int D = p.D;
  T* th = p.th() + c * D;
  T* r = p.r() + c * D;
  T* g = p.g() + c * D;
  T* V = dslot(q, p, DS_CUR_V, c);
  T* W = q.dense_metric ? dslot(q, p, DS_CUR_W, c) : nullptr;
  T lp = p.lp()[c], lk = p.lk()[c];
  if (do_post) {
    const T e = q.es[c];
    T s[2] = {0, 0};
    for (int d = lane; d < D; d += DT_THREADS) {
      const T gd = g[d];
      T rn = r[d], vn;
      if (e != T(0)) rn = rn - e / 2 * gd;
      if constexpr (TEMPER) {
        if (e != T(0)) rn = rn / p.lf.sqrt_alpha;
      }
      if (W) {
        vn = e != T(0) ? V[d] - e / 2 * W[d] : V[d];
        if constexpr (TEMPER) {
          if (e != T(0)) vn = vn / p.lf.sqrt_alpha;
        }
      } else {
        vn = minv ? minv[per_chain ? c * D + d : d] * rn : rn;
      }
      if (e != T(0) || !W) { r[d] = rn; V[d] = vn; }
      s[0] += rn * vn;
      s[1] += th[d] * gd;
    }
    block_allsum2<DT>(s[0], s[1]);
    lk = sanitize(-s[0] / 2);
    if (dense_target) lp = sanitize(-s[1] / 2);
    if (lane == 0) {
      p.lk()[c] = lk;
      if (dense_target) p.lp()[c] = lp;
    }
  }
  const T e = d_tree_advance<T, CRIT, DT>(p, q, c, lane, lp, lk);
  if (lane == 0) q.es[c] = e;
  if (e != T(0)) {  // first half of the next leapfrog (src/integrator.jl:231-237)
    for (int d = lane; d < D; d += DT_THREADS) {
      T r0 = r[d], v0 = W ? V[d] : T(0);
      if constexpr (TEMPER) {
        r0 = r0 * p.lf.sqrt_alpha;
        v0 = v0 * p.lf.sqrt_alpha;
      }
      const T rh = r0 - e / 2 * g[d];
      const T vh = W ? v0 - e / 2 * W[d] : (minv ? minv[per_chain ? c * D + d : d] * rh : rh);
      r[d] = rh;
      V[d] = vh;
      th[d] = th[d] + e * vh;
    }
  }
}

// reset the chain states for a batch of n_trans transitions
template <class T>
__global__ __launch_bounds__(256) void k_d_tree_reset(DChain<T>* S, T* es, int* n_active, int64_t N) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) *n_active = (int)N;
  if (c >= N) return;
  S[c].phase = DPH_START;
  S[c].it = 0;
  es[c] = T(0);
}

// ================================================================================================
// The NUTS loop on a POINT POOL (round 3): k_d_tree2.
//
// k_d_tree keeps "the current point" in fixed (D,N) arrays and COPIES whatever must outlive a leapfrog: every leaf that does
// not end its subtree parks five vectors (ρ, v_first, candidate θ / r / g), an accepted subtree copies its candidate, a
// change of direction swaps five vectors, a new transition copies the state into eight slots — ≈ 20 D-vectors of HBM traffic
// per chain-step where the two half-steps need 8, and at cfg4 that kernel was as long as the GEMM it alternates with
// (profiles/r2_cfg4_top_kernels.txt: 40 % of device time).
//
// Here a phase point is an immutable record in a per-chain pool, P[chain][pt][θ r g v w][D], and everything the tree remembers
// is an INDEX: the other edge, the tree-level candidate, per pending level the first-built leaf (→ its v for the U-turn test
// and, at level 0, its r = the one-leaf subtree's ρ) and the level's candidate.  The first half-step of a leapfrog reads its
// start point and writes a FRESH point (θ′, r½, v½) — no more traffic than updating in place —, the GEMM reads θ′ and
// writes g′, w′ into that point through a per-column offset (k_dgemm: ptidx), the second half-step completes it in place.
// Parking, accepting and changing direction move no vector at all; only the ρ of merged subtrees (sums, not points) are
// vectors of their own, written straight into the slot of the level they will be parked at.  A new transition starts on
// the candidate's own point (fresh r, v written over it; its g and w = M⁻¹g are still valid, so the motionless warm-up step
// of k_d_tree is needed for the first transition of a batch only).  A free point is any index no holder names
// (≤ 2·max_depth + 2 live at once; found from a bit mask, no reference counts).
//
// GeneralisedNoUTurn, untempered leapfrogs, MultinomialTS / SliceTS — the default NUTS; the other criteria and the
// TemperedLeapfrog keep k_d_tree_crit, ask / tell (ahmc_ext_*) keeps k_d_tree (its positions must sit in ctx->th).
// Same draws in the same order, same arithmetic per element as k_d_tree: chains are bit-identical to it.
// ================================================================================================
enum { PV_TH = 0, PV_R = 1, PV_G = 2, PV_V = 3, PV_W = 4, PV_COUNT = 5 };
enum { PR_TREE = 0, PR_SUB0 = 1, PR_SUB1 = 2, PR_LEVEL0 = 2 };  // ρ vectors: whole tree, two scratch, level l >= 1 at PR_LEVEL0 + l

template <class T>
struct DChain2 {
  T H0, eps, w_tree, sa_tree, dh_tree, lu;
  T cand_lp, cand_lk;
  T pw[DN_MAXLEV], psa[DN_MAXLEV], pdh[DN_MAXLEV], plp[DN_MAXLEV], plk[DN_MAXLEV];
  int32_t pna[DN_MAXLEV];
  int32_t phase, it, jw, leaf, v, cur_is_left, na_tree, depth, numerical;
  uint32_t k;
  int8_t p_first[DN_MAXLEV], p_cand[DN_MAXLEV];  // per pending level: point of its first-built leaf, point of its candidate
  int8_t p_last[DN_MAXLEV];                      // … and of its last-built leaf (StrictGeneralisedNoUTurn: the inner ends of a merge, :597-615)
  int8_t cur, oth, cand, edge;                   // the leaf in flight (moving edge), the other edge, the tree-level candidate; the edge the
};                                               // doubling in progress grew from (Strict: the old tree's inner end at the top-level merge)

static_assert(offsetof(DChain2<double>, p_first) % 8 == 0 && offsetof(DChain2<double>, p_cand) == offsetof(DChain2<double>, p_first) + DN_MAXLEV &&
                  offsetof(DChain2<double>, p_last) == offsetof(DChain2<double>, p_first) + 2 * DN_MAXLEV && DN_MAXLEV == 16,
              "d_tree_advance2 reads p_first / p_cand / p_last as six 8-byte words");

// the scalars of a chain every call needs: read at the top of k_d_tree2, together with the energies and the step, so that the
// tree bookkeeping does not start with a memory round trip of its own after the second half-step's reduction
template <class T>
struct DHot {
  T H0, eps, lu;
  int32_t phase, it, jw, leaf, v, cur_is_left, numerical;
  uint32_t k;
  int cur, oth, cand, edge;
};

template <class T>
struct DP2 {
  T* P;             // point pool  [N][n_pt][PV_COUNT][D]   (chain-major: a chain's points are one contiguous region — with the
  T* R;             // ρ vectors   [N][n_rho][D]             point-major layout every chain touched pages all over the pool)
  DChain2<T>* S;
  int* ptcur;       // (N,) pool point of the leapfrog in flight: what the GEMM reads θ′ from and writes g′, w′ to
  T* es;
  const T* RB;
  const T* VB;
  int n_trans;
  int n_pt, n_rho;
  int* n_active;
  const int* list;
  int64_t n_list;
  int dense_metric;
  int staged;       // 1: the target is not the dense Gaussian — θ′ goes to ctx->th as well and (ℓπ, g′) come back in ctx->lp / ctx->g
  // StepSizeAdaptor inside the kernel (warm-up in batches): adapt!(h, κ, adaptor, i, n_adapts, z, α) of src/sampler.jl:72-90 for a
  // NesterovDualAveraging (stepsize.jl:178-210) needs only the chain's own α, so a chain adapts its ϵ at the end of each of its
  // transitions and starts the next one at once — no barrier of all chains per transition (k_nuts MODE 3 does the same)
  int adapt_ss;
  int64_t i0, n_adapts;  // iterations done before this batch; the adaptor stops after iteration n_adapts (finalize!)
  T delta, gamma, t0, kappa;
  int32_t* da_m;
  T *da_eps, *da_mu, *da_xbar, *da_Hbar;
  const T* da_tab;
  int lazy_gw;      // 1: a point's g′, w′ are on record only where a leapfrog can START from it (k_dense_epoch); a transition then begins with the motionless step
  unsigned long long* prof;  // k_dense_epoch built with -DAHMC_EPOCH_PROF: Σ cycles in the products / the epilogue / the trees, Σ steps
};

template <class T>
__device__ __forceinline__ T* ppt(const DP2<T>& q, const KP<T>& p, int pt, int vec, int64_t c) {
  return q.P + ((c * q.n_pt + pt) * PV_COUNT + vec) * p.D;
}
template <class T>
__device__ __forceinline__ T* prho(const DP2<T>& q, const KP<T>& p, int slot, int64_t c) {
  return q.R + (c * q.n_rho + slot) * p.D;
}

template <class T>
__global__ __launch_bounds__(256) void k_d_compact2(const DChain2<T>* __restrict__ S, const int* __restrict__ in, int64_t n, int* __restrict__ out,
                                                    int* __restrict__ count) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int c = in ? in[j] : (int)j;
  if (S[c].phase != DPH_IDLE) out[atomicAdd(count, 1)] = c;
}

template <class T>
__global__ __launch_bounds__(256) void k_d_tree2_reset(DChain2<T>* S, T* es, int* ptcur, int* n_active, int64_t N) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) *n_active = (int)N;
  if (c >= N) return;
  S[c].phase = DPH_START;
  S[c].it = 0;
  S[c].cur = S[c].oth = S[c].cand = S[c].edge = 0;
  es[c] = T(0);
  ptcur[c] = 0;
}

// The tree bookkeeping of one completed leapfrog (the leaf is the pool point S.cur, its energies lp_in / lk_in).  Returns the
// signed step of the chain's next leapfrog (0 = idle, or the motionless warm-up step) and, in `src`, the pool point that
// leapfrog starts from; `used` = bit mask of the points that must survive it.
template <bool WV>
__device__ __forceinline__ void dn_chain_barrier() {
  if constexpr (WV) {
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// WV: the chain is served by ONE wave of a larger workgroup (k_dense_epoch): the barriers that order thread 0's updates of the chain's
// scalars against the other threads' reads become wave-local fences
// DC > 0: D at compile time — the vector passes go through dn_vec_pass (other element order: the sums differ in the last bits
// from the run-time loops')
// VCH: dn_vec_pass's chunk cap (registers of operands in flight)
// CRIT (round 6): the termination criterion, AHMC_TC_* — 1 GeneralisedNoUTurn (:566-570); 0 ClassicNoUTurn (:551-557): no ρ at all, the
// test needs θ and v at the two ends of a (sub)tree, which are pool points already; 2 StrictGeneralisedNoUTurn (:579-617): two more
// index holders — per pending level its LAST-built leaf (p_last), per doubling the edge it grew from (S.edge) — and at every merge the two
// extra checks (ρ_F + r_S.first ; ends F.first, S.first) and (r_F.last + ρ_S ; ends F.last, S.last) for F = the first-built half, S = the
// second.  Before round 6 these two criteria ran on the copying kernel k_d_tree_crit only.
template <class T, int DT, bool WV = false, int DC = 0, int VCH = 8, int CRIT = 1>
__device__ __forceinline__ T d_tree_advance2(const KP<T>& p, const DP2<T>& q, int64_t c, int lane, T lp_in, T lk_in, const DHot<T>& hot, int& src,
                                             uint64_t& used, bool& rewritten /* the point `src` was given a fresh momentum in this call */) {
  rewritten = false;
  DChain2<T>& S = q.S[c];
  const int D = p.D;
  const bool slice = p.sampler == 2;
  // DC > 0: the points of the pending levels (p_first, p_cand: 32 bytes) come in with the first round trip, so a merge does not
  // start with a dependent load of its own
  constexpr int NPFW = CRIT == 2 ? 6 : 4;
  unsigned long long pfw[NPFW] = {};
  if constexpr (DC > 0) {
    const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(S.p_first);
#pragma unroll
    for (int i = 0; i < NPFW; ++i) pfw[i] = pp[i];
  }
  auto lvl_last = [&](int l) -> int {   // (Strict only)
    if constexpr (CRIT != 2) return 0;
    else if constexpr (DC > 0) return (int)(int8_t)((l < 8 ? pfw[NPFW - 2] : pfw[NPFW - 1]) >> (8 * (l & 7)));
    else return S.p_last[l];
  };
  auto lvl_first = [&](int l) -> int {
    if constexpr (DC > 0) return (int)(int8_t)((l < 8 ? pfw[0] : pfw[1]) >> (8 * (l & 7)));
    else return S.p_first[l];
  };
  auto lvl_cand = [&](int l) -> int {
    if constexpr (DC > 0) return (int)(int8_t)((l < 8 ? pfw[2] : pfw[3]) >> (8 * (l & 7)));
    else return S.p_cand[l];
  };
  Rng rng = make_rng(p, c);
  DrawStream ds;
  auto resume_draws = [&](uint32_t it, uint32_t k) {
    rng.iter = p.iteration + it;
    ds.resume(rng, k);
  };
  auto bit = [](int pt) { return (uint64_t)1 << pt; };
  int phase = hot.phase, it = hot.it;
  const int cur = hot.cur;
  if (phase == DPH_WARM) {
    // the motionless step has produced w = M⁻¹g at the start point of the batch's first transition: let the first leapfrog go
    const T e_first = hot.v < 0 ? -hot.eps : hot.eps;
    dn_chain_barrier<WV>();  // (every thread has read the phase before thread 0 changes it)
    if (lane == 0) S.phase = DPH_RUN;
    src = cur;
    used = bit(cur);  // cur = oth = cand = the start point
    return e_first;
  }
  bool start = phase == DPH_START;
  T lp_start = lp_in;
  int start_pt = cur;
  if (!start) {
    resume_draws((uint32_t)it, hot.k);
    const T H0 = hot.H0, eps = hot.eps;
    const int v = hot.v, jw = hot.jw;
    const int leaf = hot.leaf;
    const uint32_t nleaf = 1u << jw;
    const int oth = hot.oth;
    int cand_tree = hot.cand;
    const T lp = lp_in, lk = lk_in;
    // ---- leaf (:638-647) ----
    const T ne = lp + lk;
    const T dH = -ne - H0;
    T sa_c = exp(jl_min(T(0), -dH)), dh_c = dH, w_c;
    int na_c = 1;
    bool sub_term;
    if (slice) {
      w_c = (hot.lu <= ne) ? T(1) : T(0);
      sub_term = !(hot.lu < p.delta_max + ne);
    } else {
      w_c = H0 + ne;
      sub_term = !(-H0 < p.delta_max + ne);
    }
    bool numerical = hot.numerical != 0 || sub_term;
    const int cur_is_left_in = hot.cur_is_left;
    // the subtree being assembled: its first-built leaf, its candidate (pool points) and its ρ (a view: the leaf's own r
    // until the first merge, then a ρ vector)
    const T* Vc = ppt(q, p, cur, PV_V, c);
    const T* rho_v = ppt(q, p, cur, PV_R, c);
    int first_c = cur, cand_c = cur;
    T sub_lp = lp, sub_lk = lk;
    // ---- merges: one per trailing zero bit of `leaf` (:649-673) ----
    const int nm = __builtin_ctz((uint32_t)leaf);
    const bool will_park = (uint32_t)leaf < nleaf;
    int merged = 0;
    for (int lvl = 0; lvl < nm && !sub_term; ++lvl) {
      const int pf = lvl_first(lvl);
      const T* p_rho = lvl == 0 ? ppt(q, p, pf, PV_R, c) : prho(q, p, PR_LEVEL0 + lvl, c);
      const T* p_vf = ppt(q, p, pf, PV_V, c);
      // the merged subtree has level lvl + 1: the last merge of a subtree that will be parked writes its ρ straight into
      // that level's slot (free: a first half is completing there), the others into the two scratch vectors in turn
      T* out = (lvl == nm - 1 && will_park) ? prho(q, p, PR_LEVEL0 + nm, c) : prho(q, p, (lvl & 1) ? PR_SUB1 : PR_SUB0, c);
      const T w_p = S.pw[lvl];
      bool keep_first;
      T w_new;
      if (slice) {
        w_new = w_p + w_c;
        keep_first = w_new * (T)ds.uniform() < w_p;
      } else {
        w_new = logaddexp(w_p, w_c);
        keep_first = w_new < w_p + (T)ds.randexp();
      }
      if (keep_first) {
        cand_c = lvl_cand(lvl);
        sub_lp = S.plp[lvl];
        sub_lk = S.plk[lvl];
      }
      w_c = w_new;
      sa_c = S.psa[lvl] + sa_c;
      na_c = S.pna[lvl] + na_c;
      const T dh_p = S.pdh[lvl];
      dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
      if constexpr (CRIT == 1) {
      // ρ = ρ_first + ρ_second; generalised_uturn_criterion with v = M⁻¹r at the two ends (:566-570,619-621)
      T dots[2] = {0, 0};
      if constexpr (DC > 0) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const T* const srcs[4] = {p_rho, rho_v, p_vf, Vc};
        dn_vec_pass<T, DT, DC, 4, VCH>(lane, srcs, [&](int d, const T2 (&x)[4]) {
          const T2 rho = x[0] + x[1];
          dots[0] += rho[0] * x[2][0] + rho[1] * x[2][1];
          dots[1] += rho[0] * x[3][0] + rho[1] * x[3][1];
          *reinterpret_cast<T2*>(out + d) = rho;
        });
      } else {
        for (int d = lane; d < D; d += DT) {
          const T rho = p_rho[d] + rho_v[d];
          dots[0] += rho * p_vf[d];
          dots[1] += rho * Vc[d];
          out[d] = rho;
        }
      }
      rho_v = out;
      first_c = pf;
      block_allsum2<DT>(dots[0], dots[1]);
      sub_term = (dots[0] <= 0) || (dots[1] <= 0);
      } else if constexpr (CRIT == 0) {
        // ClassicNoUTurn (:551-557): ends = the pending half's first-built leaf and the current leaf, left / right by the direction;
        // Δθ = θ_right − θ_left; terminated if Δθ·M⁻¹(−r_left) >= 0 or −Δθ·M⁻¹r_right >= 0 (the products as the reference forms them)
        const T* th_f = ppt(q, p, pf, PV_TH, c);
        const T* th_c = ppt(q, p, cur, PV_TH, c);
        T dots[2] = {0, 0};
        auto el = [&](T tf, T tc, T vf, T vc) {
          const T thl = v > 0 ? tf : tc, thr = v > 0 ? tc : tf, vl = v > 0 ? vf : vc, vr = v > 0 ? vc : vf;
          const T dth = thr - thl;
          dots[0] += dth * (-vl);
          dots[1] += (-dth) * vr;
        };
        if constexpr (DC > 0) {
          typedef T T2 __attribute__((ext_vector_type(2)));
          const T* const srcs[4] = {th_f, th_c, p_vf, Vc};
          dn_vec_pass<T, DT, DC, 4, VCH>(lane, srcs, [&](int, const T2 (&x)[4]) {
            el(x[0][0], x[1][0], x[2][0], x[3][0]);
            el(x[0][1], x[1][1], x[2][1], x[3][1]);
          });
        } else {
          for (int d = lane; d < D; d += DT) el(th_f[d], th_c[d], p_vf[d], Vc[d]);
        }
        first_c = pf;  // the merged subtree's first-built leaf is the pending half's
        block_allsum2<DT>(dots[0], dots[1]);
        sub_term = (dots[0] >= 0) || (dots[1] >= 0);
      } else {
        // StrictGeneralisedNoUTurn (:579-617).  F = the pending (first-built) half — first leaf pf, last leaf pl, ρ_F —, S = the half just
        // completed — first leaf first_c, last leaf cur, ρ_S:
        //   (ρ_F + ρ_S ; ends F.first, S.last)   (ρ_F + r_S.first ; ends F.first, S.first)   (r_F.last + ρ_S ; ends F.last, S.last)
        const int pl = lvl_last(lvl);
        const T* r_sf = ppt(q, p, first_c, PV_R, c);
        const T* v_sf = ppt(q, p, first_c, PV_V, c);
        const T* r_fl = ppt(q, p, pl, PV_R, c);
        const T* v_fl = ppt(q, p, pl, PV_V, c);
        T dots[6] = {0, 0, 0, 0, 0, 0};
        auto el = [&](T rf, T rs, T rsf, T rfl, T vff, T vsl, T vsf, T vfl) -> T {
          const T rho = rf + rs, rho2 = rf + rsf, rho3 = rfl + rs;
          dots[0] += rho * vff;
          dots[1] += rho * vsl;
          dots[2] += rho2 * vff;
          dots[3] += rho2 * vsf;
          dots[4] += rho3 * vfl;
          dots[5] += rho3 * vsl;
          return rho;
        };
        if constexpr (DC > 0) {
          typedef T T2 __attribute__((ext_vector_type(2)));
          const T* const srcs[8] = {p_rho, rho_v, r_sf, r_fl, p_vf, Vc, v_sf, v_fl};
          dn_vec_pass<T, DT, DC, 8, VCH>(lane, srcs, [&](int d, const T2 (&x)[8]) {
            T2 rho;
            rho[0] = el(x[0][0], x[1][0], x[2][0], x[3][0], x[4][0], x[5][0], x[6][0], x[7][0]);
            rho[1] = el(x[0][1], x[1][1], x[2][1], x[3][1], x[4][1], x[5][1], x[6][1], x[7][1]);
            *reinterpret_cast<T2*>(out + d) = rho;
          });
        } else {
          for (int d = lane; d < D; d += DT) out[d] = el(p_rho[d], rho_v[d], r_sf[d], r_fl[d], p_vf[d], Vc[d], v_sf[d], v_fl[d]);
        }
        rho_v = out;
        first_c = pf;
        block_allsum2<DT>(dots[0], dots[1]);
        block_allsum2<DT>(dots[2], dots[3]);
        block_allsum2<DT>(dots[4], dots[5]);
        sub_term = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
      }
      merged = lvl + 1;
    }
    bool subtree_over = true;
    if (sub_term) {
      // enclosing unfinished subtrees still absorb the statistics of their first halves (:666)
      const uint32_t pend = (((uint32_t)leaf - 1u) >> merged) << merged;
      for (int qq = merged; (pend >> qq) != 0u; ++qq) {
        if ((pend >> qq) & 1u) {
          sa_c = S.psa[qq] + sa_c;
          na_c = S.pna[qq] + na_c;
          const T dh_p = S.pdh[qq];
          dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
        }
      }
    } else if (will_park) {
      // park the finished level-nm subtree until its sibling is built: indices and scalars only (its ρ is the first leaf's r
      // at level 0 and already sits in the level's slot otherwise)
      uint64_t u = bit(cur) | bit(oth) | bit(cand_tree) | bit(first_c) | bit(cand_c);
      if constexpr (CRIT == 2) u |= bit(hot.edge);   // (cur is the parked subtree's last-built leaf: already in the mask)
      for (int l = 0; l < DN_MAXLEV; ++l)
        if (l != nm && (((uint32_t)leaf >> l) & 1u)) {
          u |= bit(lvl_first(l)) | bit(lvl_cand(l));
          if constexpr (CRIT == 2) u |= bit(lvl_last(l));
        }
      dn_chain_barrier<WV>();  // (all threads have read p_first / p_cand of the pending levels)
      if (lane == 0) {
        S.p_first[nm] = (int8_t)first_c;
        S.p_cand[nm] = (int8_t)cand_c;
        if constexpr (CRIT == 2) S.p_last[nm] = (int8_t)cur;
        S.pw[nm] = w_c;
        S.psa[nm] = sa_c;
        S.pdh[nm] = dh_c;
        S.pna[nm] = na_c;
        S.plp[nm] = sub_lp;
        S.plk[nm] = sub_lk;
        S.leaf = leaf + 1;
        S.numerical = numerical ? 1 : 0;
        S.k = ds.k;
      }
      subtree_over = false;  // next leapfrog: same edge, same direction
      src = cur;
      used = u;
    }
    if (!subtree_over) return v > 0 ? eps : -eps;

    // ---- top level of the doubling loop (:708-722) ----
    T w_tree = S.w_tree, sa_tree = S.sa_tree, dh_tree = S.dh_tree;
    int na_tree = S.na_tree, depth = S.depth;
    T cand_lp = S.cand_lp, cand_lk = S.cand_lk;
    if (!sub_term) {
      ++depth;
      bool acc;  // mh_accept(rng, sampler, sampler′): biased progressive sampling (:202-206)
      if (slice) acc = w_tree * (T)ds.uniform() < w_c;
      else acc = w_tree < w_c + (T)ds.randexp();
      if (acc) {
        cand_tree = cand_c;
        cand_lp = sub_lp;
        cand_lk = sub_lk;
      }
    }
    sa_tree = sa_tree + sa_c;
    na_tree = na_tree + na_c;
    dh_tree = v < 0 ? maxabs(dh_c, dh_tree) : maxabs(dh_tree, dh_c);
    w_tree = slice ? w_tree + w_c : logaddexp(w_tree, w_c);
    // isterminated on the whole tree; its edges are `cur` and the other one
    bool turn;
    if constexpr (CRIT == 0) {
      // ClassicNoUTurn on the whole tree: ends `oth` and `cur`, left / right by the direction of this doubling
      const T* th_o = ppt(q, p, oth, PV_TH, c);
      const T* th_c = ppt(q, p, cur, PV_TH, c);
      const T* o_v = ppt(q, p, oth, PV_V, c);
      T dots[2] = {0, 0};
      auto el = [&](T to, T tc, T vo, T vc) {
        const T thl = v > 0 ? to : tc, thr = v > 0 ? tc : to, vl = v > 0 ? vo : vc, vr = v > 0 ? vc : vo;
        const T dth = thr - thl;
        dots[0] += dth * (-vl);
        dots[1] += (-dth) * vr;
      };
      if constexpr (DC > 0) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const T* const srcs[4] = {th_o, th_c, o_v, Vc};
        dn_vec_pass<T, DT, DC, 4, VCH>(lane, srcs, [&](int, const T2 (&x)[4]) {
          el(x[0][0], x[1][0], x[2][0], x[3][0]);
          el(x[0][1], x[1][1], x[2][1], x[3][1]);
        });
      } else {
        for (int d = lane; d < D; d += DT) el(th_o[d], th_c[d], o_v[d], Vc[d]);
      }
      block_allsum2<DT>(dots[0], dots[1]);
      turn = (dots[0] >= 0) || (dots[1] >= 0);
    } else if constexpr (CRIT == 2) {
      // Strict on the whole tree: F = the old tree (first = `oth`, last = the edge this doubling grew from, ρ_F = the tree's ρ), S = the new
      // subtree (first = first_c, last = `cur`, ρ_S)
      T* t_rho = prho(q, p, PR_TREE, c);
      const T* o_v = ppt(q, p, oth, PV_V, c);
      const int pe = hot.edge;
      const T* r_sf = ppt(q, p, first_c, PV_R, c);
      const T* v_sf = ppt(q, p, first_c, PV_V, c);
      const T* r_fl = ppt(q, p, pe, PV_R, c);
      const T* v_fl = ppt(q, p, pe, PV_V, c);
      T dots[6] = {0, 0, 0, 0, 0, 0};
      auto el = [&](T rf, T rs, T rsf, T rfl, T vff, T vsl, T vsf, T vfl) -> T {
        const T rho = rf + rs, rho2 = rf + rsf, rho3 = rfl + rs;
        dots[0] += rho * vsl;   // (the order of the generalised test below: the moving edge first)
        dots[1] += rho * vff;
        dots[2] += rho2 * vff;
        dots[3] += rho2 * vsf;
        dots[4] += rho3 * vfl;
        dots[5] += rho3 * vsl;
        return rho;
      };
      if constexpr (DC > 0) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const T* const srcs[8] = {t_rho, rho_v, r_sf, r_fl, o_v, Vc, v_sf, v_fl};
        dn_vec_pass<T, DT, DC, 8, VCH>(lane, srcs, [&](int d, const T2 (&x)[8]) {
          T2 rho;
          rho[0] = el(x[0][0], x[1][0], x[2][0], x[3][0], x[4][0], x[5][0], x[6][0], x[7][0]);
          rho[1] = el(x[0][1], x[1][1], x[2][1], x[3][1], x[4][1], x[5][1], x[6][1], x[7][1]);
          *reinterpret_cast<T2*>(t_rho + d) = rho;
        });
      } else {
        for (int d = lane; d < D; d += DT) t_rho[d] = el(t_rho[d], rho_v[d], r_sf[d], r_fl[d], o_v[d], Vc[d], v_sf[d], v_fl[d]);
      }
      block_allsum2<DT>(dots[0], dots[1]);
      block_allsum2<DT>(dots[2], dots[3]);
      block_allsum2<DT>(dots[4], dots[5]);
      turn = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
    } else
    {
      T* t_rho = prho(q, p, PR_TREE, c);
      const T* o_v = ppt(q, p, oth, PV_V, c);
      T dots[2] = {0, 0};
      if constexpr (DC > 0) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const T* const srcs[4] = {t_rho, rho_v, Vc, o_v};
        dn_vec_pass<T, DT, DC, 4, VCH>(lane, srcs, [&](int d, const T2 (&x)[4]) {
          const T2 rho = x[0] + x[1];
          dots[0] += rho[0] * x[2][0] + rho[1] * x[2][1];
          dots[1] += rho[0] * x[3][0] + rho[1] * x[3][1];
          *reinterpret_cast<T2*>(t_rho + d) = rho;
        });
      } else {
        for (int d = lane; d < D; d += DT) {
          const T rho = t_rho[d] + rho_v[d];
          dots[0] += rho * Vc[d];
          dots[1] += rho * o_v[d];
          t_rho[d] = rho;
        }
      }
      block_allsum2<DT>(dots[0], dots[1]);
      turn = (dots[0] <= 0) || (dots[1] <= 0);
    }
    const bool done = sub_term || turn || (jw + 1 >= p.max_depth);
    if (!done) {
      // ---- next doubling: direction (:693), edge selection — a change of direction swaps two indices ----
      const bool vleft = ds.boolean();
      const bool cur_is_left = cur_is_left_in != 0;
      const bool swap = vleft != cur_is_left;
      const int new_oth = swap ? cur : oth;
      src = swap ? oth : cur;
      used = bit(src) | bit(new_oth) | bit(cand_tree);
      if (lane == 0) {
        S.w_tree = w_tree; S.sa_tree = sa_tree; S.dh_tree = dh_tree; S.na_tree = na_tree; S.depth = depth;
        S.cand_lp = cand_lp; S.cand_lk = cand_lk;
        S.cand = (int8_t)cand_tree;
        S.oth = (int8_t)new_oth;
        S.edge = (int8_t)src;   // (Strict: the tree's inner end at this doubling's top-level merge; it stays in `used` while the doubling runs)
        S.numerical = numerical ? 1 : 0;
        S.cur_is_left = vleft ? 1 : 0;
        S.v = vleft ? -1 : 1;
        S.jw = jw + 1;
        S.leaf = 1;
        S.k = ds.k;
      }
      return vleft ? -eps : eps;
    }
    // ---- Transition(zcand, stats) (:725-741) ----
    {
      const T* c_th = ppt(q, p, cand_tree, PV_TH, c);
      T* s1 = p.acc_sum() + c * D;
      T* s2 = p.acc_sumsq() + c * D;
      T* so = p.samples_out ? p.samples_out + ((int64_t)it * p.N + c) * D : nullptr;
      const bool last = it + 1 >= q.n_trans;
      if (p.accum || so || last) {
        const T* c_r = ppt(q, p, cand_tree, PV_R, c);
        const T* c_g = ppt(q, p, cand_tree, PV_G, c);
        T* th = p.th() + c * D;
        T* r = p.r() + c * D;
        T* g = p.g() + c * D;
        if constexpr (DC > 0) {
          typedef T T2 __attribute__((ext_vector_type(2)));
          const T* const srcs[5] = {c_th, c_r, c_g, s1, s2};
          dn_vec_pass<T, DT, DC, 5, VCH>(lane, srcs, [&](int d, const T2 (&x)[5]) {
            const T2 t = x[0];
            if (p.accum) {
              *reinterpret_cast<T2*>(s1 + d) = x[3] + t;
              *reinterpret_cast<T2*>(s2 + d) = x[4] + t * t;
            }
            if (so) *reinterpret_cast<T2*>(so + d) = t;
            if (last) {
              *reinterpret_cast<T2*>(th + d) = t;
              *reinterpret_cast<T2*>(r + d) = x[1];
              *reinterpret_cast<T2*>(g + d) = x[2];
            }
          });
        } else
        for (int d = lane; d < D; d += DT) {
          const T t = c_th[d];
          if (p.accum) { s1[d] += t; s2[d] += t * t; }
          if (so) so[d] = t;
          if (last) {  // the state the context holds after the batch (ahmc_get_phasepoint, the next call's start point)
            th[d] = t;
            r[d] = c_r[d];
            g[d] = c_g[d];
          }
        }
      }
      if (lane == 0) {
        const T H = -(cand_lp + cand_lk);
        p.lp()[c] = cand_lp;
        p.lk()[c] = cand_lk;
        p.eps_cur()[c] = eps;
        p.st_nsteps()[c] = na_tree;
        p.st_accept()[c] = 1;
        p.st_accrate()[c] = sa_tree / (T)na_tree;
        p.st_logdens()[c] = cand_lp;
        p.st_H()[c] = H;
        p.st_Herr()[c] = H - H0;
        p.st_maxHerr()[c] = dh_tree;
        p.st_depth()[c] = depth;
        p.st_numerr()[c] = numerical ? 1 : 0;
        if (p.accum) {
          p.acc_nsteps()[c] += na_tree;
          p.acc_ndiv()[c] += numerical ? 1 : 0;
          accumulate_energy(p, c, H);
        }
        if (q.adapt_ss) {  // this chain's adapt! (the same arithmetic as k_adapt_da: a batched warm-up == the per-iteration one)
          const int64_t i = q.i0 + it + 1;
          if (i <= q.n_adapts) {
            DAState<T> das{q.da_m[c], q.da_eps[c], q.da_mu[c], q.da_xbar[c], q.da_Hbar[c]};
            da_step(das, sa_tree / (T)na_tree, q.delta, q.gamma, q.t0, q.kappa, q.da_tab);
            if (i == q.n_adapts) das.eps = exp(das.xbar);  // finalize! (stepsize.jl:55-62)
            q.da_m[c] = das.m; q.da_eps[c] = das.eps; q.da_mu[c] = das.mu; q.da_xbar[c] = das.xbar; q.da_Hbar[c] = das.Hbar;
            p.eps_nom()[c] = das.eps;                      // update(κ, adaptor): nominal step size ← getϵ
          }
        }
      }
      ++it;
      if (it >= q.n_trans) {
        if (lane == 0) {
          S.phase = DPH_IDLE;
          S.it = it;
          atomicSub(q.n_active, 1);
        }
        src = cand_tree;
        used = bit(cand_tree);
        return T(0);
      }
      if (q.adapt_ss) {  // the next transition's ϵ is read from memory by every thread (chain_eps): thread 0's store must be visible
        __threadfence_block();
        dn_chain_barrier<WV>();
      }
      start = true;
      lp_start = cand_lp;
      start_pt = cand_tree;  // the next transition starts ON the candidate's point
    }
  }
  // ---- start of transition `it` (src/sampler.jl:54-57, src/trajectory.jl:677-690) on the point start_pt: θ, g (and w) are
  // the candidate's (first transition of a batch: copied in from the context), fresh r and v = M⁻¹r from the batch ----
  {
    resume_draws((uint32_t)it, 0u);
    const T eps = chain_eps(p, rng, c);
    const T* rb = q.RB + ((int64_t)it * p.N + c) * D;
    const T* vb = q.VB + ((int64_t)it * p.N + c) * D;
    T* s_r = ppt(q, p, start_pt, PV_R, c);
    T* s_v = ppt(q, p, start_pt, PV_V, c);
    T* t_rho = prho(q, p, PR_TREE, c);
    const bool first_of_batch = phase == DPH_START;
    T* s_th = ppt(q, p, start_pt, PV_TH, c);
    T* s_g = ppt(q, p, start_pt, PV_G, c);
    const T* th = p.th() + c * D;
    const T* g = p.g() + c * D;
    T dots[2] = {0, 0};
    if constexpr (DC > 0) {
      typedef T T2 __attribute__((ext_vector_type(2)));
      const T* const srcs[4] = {rb, vb, th, g};
      dn_vec_pass<T, DT, DC, 4, VCH>(lane, srcs, [&](int d, const T2 (&x)[4]) {
        dots[0] += x[0][0] * x[1][0] + x[0][1] * x[1][1];
        *reinterpret_cast<T2*>(s_r + d) = x[0];
        *reinterpret_cast<T2*>(s_v + d) = x[1];
        *reinterpret_cast<T2*>(t_rho + d) = x[0];
        if (first_of_batch) {
          *reinterpret_cast<T2*>(s_th + d) = x[2];
          *reinterpret_cast<T2*>(s_g + d) = x[3];
        }
      });
    } else
    for (int d = lane; d < D; d += DT) {
      const T rd = rb[d], vd = vb[d];
      dots[0] += rd * vd;
      s_r[d] = rd;
      s_v[d] = vd;
      t_rho[d] = rd;
      if (first_of_batch) {
        s_th[d] = th[d];
        s_g[d] = g[d];
      }
    }
    block_allsum2<DT>(dots[0], dots[1]);
    const T lp = lp_start;
    const T lk = sanitize(-dots[0] / 2);
    const T H0 = -(lp + lk);
    T lu = 0, w_tree;
    if (slice) {
      lu = -H0 - (T)ds.randexp();  // SliceTS(rng, z0) (:144-145)
      w_tree = 1;
    } else {
      w_tree = 0;  // MultinomialTS(rng, z0): ℓw = 0 (:155)
    }
    const bool vleft = ds.boolean();
    const bool warm = q.dense_metric && (first_of_batch || q.lazy_gw);  // w = M⁻¹g of the start point is not known yet (lazy_gw: the candidate's record may lack g, w)
    if (lane == 0) {
      p.lk()[c] = lk;
      S.H0 = H0; S.eps = eps; S.lu = lu; S.w_tree = w_tree; S.sa_tree = 0; S.dh_tree = 0; S.na_tree = 0;
      S.cand_lp = lp; S.cand_lk = lk;
      S.depth = 0; S.numerical = 0; S.jw = 0; S.leaf = 1;
      S.cur_is_left = vleft ? 1 : 0;
      S.v = vleft ? -1 : 1;
      S.k = ds.k;
      S.it = it;
      S.cur = S.oth = S.cand = S.edge = (int8_t)start_pt;
      S.phase = warm ? DPH_WARM : DPH_RUN;
    }
    src = start_pt;
    used = bit(start_pt);
    rewritten = true;
    return warm ? T(0) : (vleft ? -eps : eps);
  }
}

// One global step for every listed chain: second half of the leapfrog in flight (completing its pool point) → d_tree_advance2
// → first half of the next leapfrog INTO A FRESH POINT.  One chain per workgroup of DT threads.
// The kernel is a chain of dependent memory round trips (scalars → the point's vectors → [merge operands] → stores); the
// values of the completed point stay in registers (NE elements per thread) for the first half-step that usually follows from
// the same point, so that half-step does not read them again.
template <class T, int DT, int CRIT = 1>
__global__ __launch_bounds__(DT) void k_d_tree2(KP<T> p, DP2<T> q, const T* __restrict__ minv, int per_chain, int dense_target, int do_post) {
  constexpr int NE = 4;  // elements of a vector a thread keeps in registers (D <= NE·DT: every default thread count)
  const int lane = threadIdx.x;
  const int64_t j = blockIdx.x;
  if (j >= q.n_list) return;
  const int64_t c = q.list ? q.list[j] : j;
  DChain2<T>& S = q.S[c];
  if (S.phase == DPH_IDLE) return;
  const int D = p.D;
  T lp = p.lp()[c], lk = p.lk()[c];
  DHot<T> hot;
  hot.H0 = S.H0; hot.eps = S.eps; hot.lu = S.lu;
  hot.phase = S.phase; hot.it = S.it; hot.jw = S.jw; hot.leaf = S.leaf; hot.v = S.v; hot.cur_is_left = S.cur_is_left; hot.numerical = S.numerical;
  hot.k = S.k;
  hot.cur = S.cur; hot.oth = S.oth; hot.cand = S.cand; hot.edge = S.edge;
  const int cur = hot.cur;
  const bool in_regs = do_post && D <= NE * DT;
  T k_th[NE], k_r[NE], k_g[NE], k_v[NE], k_w[NE];
  if (do_post) {
    T* TH = ppt(q, p, cur, PV_TH, c);
    T* R = ppt(q, p, cur, PV_R, c);
    T* G = ppt(q, p, cur, PV_G, c);
    T* V = ppt(q, p, cur, PV_V, c);
    const T* W = q.dense_metric ? ppt(q, p, cur, PV_W, c) : nullptr;
    const T* g_in = q.staged ? p.g() + c * D : G;  // staged: the target kernel left g′ in the context's array
    const T e = q.es[c];
    // TemperedLeapfrog (src/integrator.jl:198-209), round 6: every NUTS leaf is step(lf, h, z, 1) — r·√α before its first half-step (below),
    // r/√α after its second; v = M⁻¹r is carried by the same recurrence and scales with r
    const bool tmp = p.lf.kind == 2;
    const T sqa = tmp ? p.lf.sqrt_alpha : T(1);
    T s[2] = {0, 0};
    auto second_half = [&](int d, T& o_th, T& o_r, T& o_g, T& o_v, T& o_w) {
      const T gd = g_in[d], td = TH[d], wd = W ? W[d] : T(0);
      T rn = R[d], vn;
      if (e != T(0)) { rn = rn - e / 2 * gd; if (tmp) rn = rn / sqa; }
      if (W) { vn = e != T(0) ? V[d] - e / 2 * wd : V[d]; if (tmp && e != T(0)) vn = vn / sqa; }
      else vn = minv ? minv[per_chain ? c * D + d : d] * rn : rn;
      if (e != T(0) || !W) { R[d] = rn; V[d] = vn; }
      if (q.staged) G[d] = gd;
      s[0] += rn * vn;
      s[1] += td * gd;
      o_th = td; o_r = rn; o_g = gd; o_v = vn; o_w = wd;
    };
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int d = lane + i * DT;
        if (d < D) second_half(d, k_th[i], k_r[i], k_g[i], k_v[i], k_w[i]);
      }
    } else {
      T t0, t1, t2, t3, t4;
      for (int d = lane; d < D; d += DT) second_half(d, t0, t1, t2, t3, t4);
    }
    block_allsum2<DT>(s[0], s[1]);
    lk = sanitize(-s[0] / 2);
    if (dense_target) lp = sanitize(-s[1] / 2);
    else lp = sanitize(lp);  // (a user kernel's ℓπ: PhasePoint's non-finite → −Inf, src/hamiltonian.jl:95-104 — here instead of in a launch of its own)
    if (lane == 0) {
      p.lk()[c] = lk;
      p.lp()[c] = lp;
    }
  }
  int src = cur;
  uint64_t used = 0;
  bool rewritten = false;
  __syncthreads();  // (every thread holds the chain's scalars: from here on thread 0 may update them)
  const T e = d_tree_advance2<T, DT, false, 0, 8, CRIT>(p, q, c, lane, lp, lk, hot, src, used, rewritten);
  if (lane == 0) q.es[c] = e;
  if (e != T(0)) {  // first half of the next leapfrog (src/integrator.jl:231-237): src → a fresh point
    const int dst = __builtin_ctzll(~used);
    T* dTH = ppt(q, p, dst, PV_TH, c);
    T* dR = ppt(q, p, dst, PV_R, c);
    T* dV = ppt(q, p, dst, PV_V, c);
    T* th_st = q.staged ? p.th() + c * D : nullptr;
    const bool dm = q.dense_metric != 0;
    const bool tmp1 = p.lf.kind == 2;
    const T sqa1 = tmp1 ? p.lf.sqrt_alpha : T(1);
    auto first_half = [&](int d, T td, T rd, T gd, T vd, T wd) {
      const T rh = (tmp1 ? rd * sqa1 : rd) - e / 2 * gd;
      const T vh = dm ? (tmp1 ? vd * sqa1 : vd) - e / 2 * wd : (minv ? minv[per_chain ? c * D + d : d] * rh : rh);
      const T tn = td + e * vh;
      dR[d] = rh;
      dV[d] = vh;
      dTH[d] = tn;
      if (th_st) th_st[d] = tn;
    };
    if (in_regs && src == cur && !rewritten) {  // the usual case inside a subtree: the point just completed, still in registers
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int d = lane + i * DT;
        if (d < D) first_half(d, k_th[i], k_r[i], k_g[i], k_v[i], k_w[i]);
      }
    } else {
      const T* sTH = ppt(q, p, src, PV_TH, c);
      const T* sR = ppt(q, p, src, PV_R, c);
      const T* sG = ppt(q, p, src, PV_G, c);
      const T* sV = ppt(q, p, src, PV_V, c);
      const T* sW = dm ? ppt(q, p, src, PV_W, c) : nullptr;
      for (int d = lane; d < D; d += DT) first_half(d, sTH[d], sR[d], sG[d], sV[d], sW ? sW[d] : T(0));
    }
    if (lane == 0) {
      S.cur = (int8_t)dst;
      q.ptcur[c] = dst;
    }
  } else if (lane == 0) {
    q.ptcur[c] = src;  // motionless / idle: the GEMM (if any) serves the point the chain sits on
  }
}

// ================================================================================================
// k_dense_epoch (round 4): CHAIN-COMPLETE workgroups.  The alternation k_dgemm → k_d_tree2 makes every chain's step wait for a
// chip-wide launch twice, and run side by side the two kernels share every CU.  Here a workgroup of 8 waves OWNS 32
// chains for a whole epoch of global steps: per step it computes g′ = Pθ′ and w′ = (M⁻¹P)θ′ for ITS chains — all D rows of both
// products, wave w the rows [w·D/8, (w+1)·D/8) — with the matrices streamed from L2 in MFMA-fragment order (k_dense_swizzle: one
// 16-byte load per lane feeds two MFMAs; no LDS staging, no barrier in the k loop; θ′ comes straight from the chains' pool points),
// completes the leapfrog in the EPILOGUE (second half-step on the accumulators, transposed through the wave's LDS tile into whole cache
// lines; r·v and θ′·g′ summed over the workgroup in a fixed order; the first half-step of the NEXT leapfrog written speculatively
// into a point no holder can name), and then advances the trees of its chains (d_tree_advance2 with 16 lanes per chain: a wave's
// four chains at once, each under its own exec mask), which adopt the speculative point or take the half-step themselves.
// No other workgroup touches these chains during the epoch, so the only synchronisation is the workgroup's own barrier; workgroups
// drift apart, and one's memory-bound phases run beside its neighbours' MFMA loops.
// The state between two steps is k_d_tree2's (DChain2, the point pool, es, ptcur): an epoch can end after any step and the
// step-synchronous kernels can carry on (they do, for the tail of a batch when few chains are left) — with one difference,
// DP2::lazy_gw: g′, w′ of a point are on record only where a leapfrog can START from it, and every transition begins with the
// motionless step that recomputes them at its start point (what bounds the epilogue is its stores: 38.1 → 42.1 TFLOP/s on cfg4).
// Accumulation over k is in the order of k_dgemm (k ascending, four at a time): g′, w′ have its bits, and the chains take the
// decisions of the step-synchronous kernels (tests/test_gpu_parity.py: test_dense_epoch_kernel_equals_step_synchronous_kernels).
// ================================================================================================
constexpr int DE_WAVES = 8, DE_NCT = 2, DE_CHAINS = 16 * DE_NCT;
#ifndef AHMC_EPOCH_NT
#define AHMC_EPOCH_NT 1  // the epilogue's stores are non-temporal (cfg4 42.5 -> 42.9 TFLOP/s); 0: plain stores
#endif
#if AHMC_EPOCH_NT
#define DE_STORE(ptr, val) __builtin_nontemporal_store(val, reinterpret_cast<T2*>(ptr))
#else
#define DE_STORE(ptr, val) (*reinterpret_cast<T2*>(ptr) = (val))
#endif
#ifndef AHMC_EPOCH_NB
#define AHMC_EPOCH_NB 8  // (cfg4: 4 → 8 chains per batch of the epilogue 35.7 → 36.1 TFLOP/s)
#endif

// matrices in fragment order: out[(((kk·8 + w)·RT + j)·64 + lane)·2 + e] = A_m[row][4kk + lane/16], fragment f = 2j + e of wave w:
// m = f / RT (0: A0 → g′, 1: A1 → w′), row tile t = f % RT, row = w·16RT + 4RT·(i mod 4) + 4t + i/4 for the MFMA row i = lane mod 16 —
// the accumulator of lane (q, n) then holds rows w·16RT + 4RT·q + 4t + v of chain n: 4·RT contiguous elements per lane over (t, v)
template <class T>
__global__ __launch_bounds__(256) void k_dense_swizzle(const T* __restrict__ A0, const T* __restrict__ A1, T* __restrict__ out, int D, int RT) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2 * (int64_t)D * D) return;
  const int e = (int)(idx & 1), l = (int)((idx >> 1) & 63);
  int64_t rest = idx >> 7;
  const int j = (int)(rest % RT);
  rest /= RT;
  const int w = (int)(rest % DE_WAVES), kk = (int)(rest / DE_WAVES);
  const int f = 2 * j + e, m = f / RT, t = f % RT, i = l & 15;
  const int row = w * 16 * RT + 4 * RT * (i & 3) + 4 * t + (i >> 2), k = 4 * kk + (l >> 4);
  out[idx] = (m ? A1 : A0)[row + (int64_t)k * D];
}

template <class T>
struct DEMeta {  // what the epilogue needs of a chain of the workgroup (per wave: each wave keeps its own copy, no barrier)
  long long c;
  T e;
  int cur, act;  // act bit 0: the chain is running; bit 1: g′, w′ of this point go on record
};

template <class T, int RT>
__global__ __launch_bounds__(64 * DE_WAVES) void k_dense_epoch(KP<T> p, DP2<T> q, const T* __restrict__ Asw, int max_steps) {
  using M = Mfma<T>;
  typedef T T2 __attribute__((ext_vector_type(2)));
  static_assert(RT == 2 || RT == 4, "k_dense_epoch: D = 256 or 512");
  constexpr int D = 128 * RT, NK = D / 4, NF = 2 * RT;
  constexpr int RW = 16 * RT;   // rows of a wave
  constexpr int LPC = RW / 2;   // lanes per chain in the coalesced pass of the epilogue (16 bytes per lane)
  constexpr int CPI = 64 / LPC; // chains per instruction there
  constexpr int TP = RW + 2;    // pitch of a transposition tile (doubles): 16-byte rows, conflict-free columns
  constexpr int GL = 16, CPW = DE_CHAINS / DE_WAVES;  // tree phase: lanes per chain, chains per wave (all at once)
  static_assert(CPW * GL == 64, "tree phase: the chains of a wave fill it");
  __shared__ T tile[DE_WAVES][2][16][TP];
  __shared__ DEMeta<T> meta[DE_WAVES][DE_CHAINS];
  __shared__ double red[DE_WAVES][DE_CHAINS][2];
  __shared__ int any_active[2];
  __shared__ int spec_pt[DE_CHAINS];  // per chain: the pool point its speculative half-step goes to (−1: none), set by the tree phase
  const int64_t j0 = (int64_t)blockIdx.x * DE_CHAINS;
  if (j0 >= q.n_list) return;
  const int64_t c_safe = q.list ? (int64_t)q.list[j0] : j0;
  if (threadIdx.x == 0) any_active[0] = any_active[1] = 0;
  if (threadIdx.x < DE_CHAINS) spec_pt[threadIdx.x] = -1;
  __syncthreads();
  constexpr size_t AKS = (size_t)DE_WAVES * RT * 64;  // T2 elements per k-step
#ifdef AHMC_EPOCH_PROF
  unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc_ = __builtin_readcyclecounter();
#define AHMC_EPOCH_TICK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pc_; pc_ = n_; }
#else
#define AHMC_EPOCH_TICK(i)
#endif
  for (int step = 0; step < max_steps; ++step) {
    // Nothing a lane holds may live across the tree phase below: its four chain groups run under their own exec masks, and a value
    // the register allocator spills INSIDE such a region is stored for the active lanes only and reloaded for all (isa_check.py).
    // So every per-lane quantity is derived anew, per step and again per phase, from a thread index the compiler cannot trace back.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, w = tid >> 6, qd = lane >> 4, n16 = lane & 15;
    const bool tmp = p.lf.kind == 2;                      // TemperedLeapfrog (src/integrator.jl:198-209)
    const T sqa = tmp ? p.lf.sqrt_alpha : T(1);            // √α (not `sa`: the tree phase has a sum of that name)
    const T2* Ap = reinterpret_cast<const T2*>(Asw) + (size_t)w * RT * 64 + lane;
    // ---- the columns of this step: the point each chain's leapfrog in flight sits on ----
    const T* Bp[DE_NCT];
#pragma unroll
    for (int ct = 0; ct < DE_NCT; ++ct) {
      const int64_t jc = j0 + 16 * ct + n16;
      const int64_t colc = jc < q.n_list ? (q.list ? (int64_t)q.list[jc] : jc) : -1;
      const int64_t cc = colc >= 0 ? colc : c_safe;
      const int cur = q.ptcur[cc];
      const DChain2<T>& Sc = q.S[cc];
      int act = (colc >= 0 && Sc.phase != DPH_IDLE) ? 1 : 0;
      const T ec = act ? q.es[cc] : T(0);
      // g′ and w′ are read again only by a leapfrog that STARTS from this point: the motionless step's point, the last leaf of a doubling
      // (the tree's new edge), and — for its g — the candidate of the batch's last transition; everything else goes on from the
      // speculative half-step (a chain without one keeps the record: bit 1 is also set in the epilogue when there is none)
      if (!q.lazy_gw || ec == T(0) || Sc.leaf == (1 << Sc.jw) || Sc.it + 1 >= q.n_trans) act |= 2 * act;
      Bp[ct] = ppt(q, p, cur, PV_TH, cc) + qd;
      if (qd == 0) meta[w][16 * ct + n16] = DEMeta<T>{(long long)cc, ec, cur, act};
    }
    // ---- g′ = Pθ′, w′ = (M⁻¹P)θ′: 2·RT row tiles × 2 column tiles per wave, operands two k-steps ahead in registers ----
    typename M::acc_t acc[NF][DE_NCT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int ct = 0; ct < DE_NCT; ++ct) acc[f][ct] = typename M::acc_t{0, 0, 0, 0};
    AHMC_EPOCH_TICK(0)
    {
      T2 a[3][RT];
      T b[3][DE_NCT];
      auto load = [&](int kk, T2 (&aa)[RT], T (&bb)[DE_NCT]) {
        kk = kk < NK ? kk : NK - 1;
#pragma unroll
        for (int j = 0; j < RT; ++j) aa[j] = Ap[(size_t)kk * AKS + (size_t)j * 64];
#pragma unroll
        for (int ct = 0; ct < DE_NCT; ++ct) bb[ct] = Bp[ct][4 * kk];
      };
      auto mult = [&](const T2 (&aa)[RT], const T (&bb)[DE_NCT]) {
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ct = 0; ct < DE_NCT; ++ct) acc[2 * j + e][ct] = M::mma(aa[j][e], bb[ct], acc[2 * j + e][ct]);
      };
      load(0, a[0], b[0]);
      load(1, a[1], b[1]);
      for (int kk = 0; kk < NK; kk += 3) {
        load(kk + 2, a[2], b[2]);
        mult(a[0], b[0]);
        if (kk + 1 < NK) {
          load(kk + 3, a[0], b[0]);
          mult(a[1], b[1]);
        }
        if (kk + 2 < NK) {
          load(kk + 4, a[1], b[1]);
          mult(a[2], b[2]);
        }
      }
    }
    AHMC_EPOCH_TICK(1)
    // ---- epilogue: second half of the leapfrog (src/integrator.jl:243-250).  The accumulators hold 4·RT consecutive rows of ONE
    // chain per lane and sixteen chains per instruction; through the wave's own LDS tile they become 16 bytes per lane with LPC
    // consecutive lanes per chain, so every access below covers whole cache lines ----
#pragma unroll
    for (int ct = 0; ct < DE_NCT; ++ct) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int r0 = 4 * RT * qd + 4 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<T2*>(&tile[w][0][n16][r0 + 2 * h]) = T2{acc[t][ct][2 * h], acc[t][ct][2 * h + 1]};
          *reinterpret_cast<T2*>(&tile[w][1][n16][r0 + 2 * h]) = T2{acc[RT + t][ct][2 * h], acc[RT + t][ct][2 * h + 1]};
        }
      }
      __builtin_amdgcn_wave_barrier();
      constexpr int NIT = 16 / CPI, NB = NIT < AHMC_EPOCH_NB ? NIT : AHMC_EPOCH_NB;  // chains in batches of NB per half-wave: their loads are all in flight before the first is used
      const int pos = lane % LPC, dd = w * RW + 2 * pos;
#pragma unroll
      for (int ib = 0; ib < NIT; ib += NB) {
      T2 r2[NB], v2[NB], th2[NB];
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const DEMeta<T> m = meta[w][16 * ct + (ib + it) * CPI + lane / LPC];
        th2[it] = r2[it] = v2[it] = T2{0, 0};  // (defined on every path: a value left undefined for the idle chains' lanes is carried around the step loop)
        if (m.act) {
          th2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_TH, (int64_t)m.c) + dd);
          r2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_R, (int64_t)m.c) + dd);
          v2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_V, (int64_t)m.c) + dd);
        }
      }
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int nn = (ib + it) * CPI + lane / LPC;
        const DEMeta<T> m = meta[w][16 * ct + nn];
        T s0 = 0, s1 = 0;
        if (m.act) {
          const T e = m.e;
          const T2 g2 = *reinterpret_cast<const T2*>(&tile[w][0][nn][2 * pos]), w2 = *reinterpret_cast<const T2*>(&tile[w][1][nn][2 * pos]);
          T2 rr = r2[it], vv = v2[it];
          if (e != T(0)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              rr[h] = rr[h] - e / 2 * g2[h];
              vv[h] = vv[h] - e / 2 * w2[h];
              if (tmp) { rr[h] = rr[h] / sqa; vv[h] = vv[h] / sqa; }   // TemperedLeapfrog: r/√α after the second half-step (k_d_tree2)
            }
            DE_STORE(ppt(q, p, m.cur, PV_R, (int64_t)m.c) + dd, rr);
            DE_STORE(ppt(q, p, m.cur, PV_V, (int64_t)m.c) + dd, vv);
            const int sp = spec_pt[16 * ct + nn];
            if (sp >= 0) {  // the first half of the NEXT leapfrog if the tree goes on from this point with this step (the usual case), into a point no
              T2 rh, vh, tn;  // holder can name whatever the tree decides; the tree phase adopts it or takes the half-step itself
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                rh[h] = (tmp ? rr[h] * sqa : rr[h]) - e / 2 * g2[h];
                vh[h] = (tmp ? vv[h] * sqa : vv[h]) - e / 2 * w2[h];
                tn[h] = th2[it][h] + e * vh[h];
              }
              DE_STORE(ppt(q, p, sp, PV_R, (int64_t)m.c) + dd, rh);
              DE_STORE(ppt(q, p, sp, PV_V, (int64_t)m.c) + dd, vh);
              DE_STORE(ppt(q, p, sp, PV_TH, (int64_t)m.c) + dd, tn);
            }
          }
          if ((m.act & 2) || e == T(0) || spec_pt[16 * ct + nn] < 0) {
            DE_STORE(ppt(q, p, m.cur, PV_G, (int64_t)m.c) + dd, g2);
            DE_STORE(ppt(q, p, m.cur, PV_W, (int64_t)m.c) + dd, w2);
          }
          s0 = rr[0] * vv[0] + rr[1] * vv[1];
          s1 = th2[it][0] * g2[0] + th2[it][1] * g2[1];
        }
        wave_allsum2<LPC>(s0, s1);
        if (pos == 0) { red[w][16 * ct + nn][0] = (double)s0; red[w][16 * ct + nn][1] = (double)s1; }
      }
      }
      __builtin_amdgcn_wave_barrier();
    }
    AHMC_EPOCH_TICK(2)
    __syncthreads();  // (A) the points are complete, the partial sums are in LDS
    AHMC_EPOCH_TICK(3)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    if (tid2 == 0) any_active[(step + 1) & 1] = 0;
    // ---- trees: the wave serves its CPW chains AT ONCE, GL lanes each (k_d_tree2 from the reduction on; the groups of a wave follow
    // their own control flow — every exchange of d_tree_advance2 stays inside a 16-lane row) ----
    {
      const int lane2 = tid2 & 63, w2 = tid2 >> 6;
      const int grp = lane2 / GL, l16 = lane2 % GL;
      const int slot = w2 * CPW + grp;
      const int64_t jj = j0 + slot;
      bool chain_active = false;
      if (jj < q.n_list) {
        const int64_t c = q.list ? (int64_t)q.list[jj] : jj;
        DChain2<T>& S = q.S[c];
        if (S.phase != DPH_IDLE) {
          T sa = 0, sb = 0;
#pragma unroll
          for (int k = 0; k < DE_WAVES; ++k) { sa += (T)red[k][slot][0]; sb += (T)red[k][slot][1]; }
          const T lk = sanitize(-sa / 2), lp = sanitize(-sb / 2);
          DHot<T> hot;
          hot.H0 = S.H0; hot.eps = S.eps; hot.lu = S.lu;
          hot.phase = S.phase; hot.it = S.it; hot.jw = S.jw; hot.leaf = S.leaf; hot.v = S.v; hot.cur_is_left = S.cur_is_left; hot.numerical = S.numerical;
          hot.k = S.k;
          hot.cur = S.cur; hot.oth = S.oth; hot.cand = S.cand; hot.edge = S.edge;
          if (l16 == 0) {
            p.lk()[c] = lk;
            p.lp()[c] = lp;
          }
          int src = hot.cur;
          uint64_t used = 0;
          bool rewritten = false;
          dn_chain_barrier<true>();
          const T e = d_tree_advance2<T, GL, true, D>(p, q, c, l16, lp, lk, hot, src, used, rewritten);
          if (l16 == 0) q.es[c] = e;
          if (e != T(0)) {  // first half of the next leapfrog (src/integrator.jl:231-237): src → a fresh point
            const DEMeta<T> m = meta[w2][slot];
            const int sp = spec_pt[slot];
            // the epilogue has already taken it if the leapfrog goes on from the point just completed with the same signed step
            const bool hit = sp >= 0 && m.e != T(0) && src == hot.cur && e == m.e && !rewritten;
            const int dst = hit ? sp : __builtin_ctzll(~used);
            T* dTH = ppt(q, p, dst, PV_TH, c);
            T* dR = ppt(q, p, dst, PV_R, c);
            T* dV = ppt(q, p, dst, PV_V, c);
            dn_chain_barrier<true>();  // (the start of a transition has just written r, v of `src`)
            const T* const srcs[5] = {ppt(q, p, src, PV_TH, c), ppt(q, p, src, PV_R, c), ppt(q, p, src, PV_G, c), ppt(q, p, src, PV_V, c), ppt(q, p, src, PV_W, c)};
            if (l16 == 0) spec_pt[slot] = __builtin_ctzll(~(used | ((uint64_t)1 << dst)));
            if (!hit) dn_vec_pass<T, GL, D, 5>(l16, srcs, [&](int d, const T2 (&x)[5]) {
              T2 rh, vh, tn;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                rh[h] = (tmp ? x[1][h] * sqa : x[1][h]) - e / 2 * x[2][h];
                vh[h] = (tmp ? x[3][h] * sqa : x[3][h]) - e / 2 * x[4][h];
                tn[h] = x[0][h] + e * vh[h];
              }
              *reinterpret_cast<T2*>(dR + d) = rh;
              *reinterpret_cast<T2*>(dV + d) = vh;
              *reinterpret_cast<T2*>(dTH + d) = tn;
            });
            if (l16 == 0) {
              S.cur = (int8_t)dst;
              q.ptcur[c] = dst;
            }
            chain_active = true;
          } else {
            if (l16 == 0) {
              q.ptcur[c] = src;  // motionless / idle: the products serve the point the chain sits on
              spec_pt[slot] = -1;
            }
            dn_chain_barrier<true>();        // (thread 0 may just have set the phase)
            if (S.phase != DPH_IDLE) chain_active = true;
          }
        }
      }
      if (chain_active && l16 == 0) any_active[step & 1] = 1;
    }
    AHMC_EPOCH_TICK(4)
    __syncthreads();  // (B) every chain's next point, step and phase are visible to the whole workgroup
    AHMC_EPOCH_TICK(5)
#ifdef AHMC_EPOCH_PROF
    pt_[7] += 1;
#endif
    if (!any_active[step & 1]) break;
  }
#ifdef AHMC_EPOCH_PROF
  if (threadIdx.x == 0 && q.prof)
    for (int i = 0; i < 8; ++i) atomicAdd(q.prof + i, pt_[i]);
#endif
}

// ================================================================================================
// k_dense_epoch2 (round 6): the chain-complete kernel for ANY D = 64·NW (NW = 4 … 16 waves per workgroup), Float64 and Float32,
// with one or two MFMA column tiles (16 or 32 chains) per workgroup.
//
// Why.  k_dense_epoch multiplies 60 % of a workgroup-step: its 8 waves hold 16 accumulators each (256 VGPRs: one workgroup, two waves
// per SIMD, per CU), and while they are in the epilogue / the trees / the barriers nothing issues an MFMA.  profiles/r5_mfma_clock.jsonl
// says what fills the pipe: TWO multiplying waves per SIMD with 8 accumulators each.  So: a workgroup of NW waves owns ONE column tile —
// 16 chains, wave w the rows [64w, 64w + 64) of both matrices = 8 accumulators —, fits 128 registers (WPE = 4 waves per SIMD) and
// 8.4 KB of LDS per wave, and TWO such workgroups share a CU: one's memory phases run beside the other's products, with two
// multiplying waves per SIMD either way.  The price is the matrices streamed from L2 once per 16 chain-steps instead of once per 32.
//
// What changed against k_dense_epoch, beside the shape:
//  * the transposition tile holds ONE matrix at a time (tile[NW][16][66]): pass G turns g′ into r ← r½ − ϵ/2·g′ (and the speculative
//    next r½), pass W turns w′ into v (and v½, θ″); r stays in registers between the two for r·v, θ is read again in pass W;
//  * the tree phase serves 4 chains per wave and round: waves 0 … (chains / 4 − 1) of a round, the others go to the barrier;
//  * the fragment order of the matrices follows the element type (k_dense_swizzle2): an f64 accumulator of lane (q, n) holds MFMA rows
//    q + 4v, an f32 one rows 4q + v — either way the lane ends up with 16 consecutive elements of chain n.
// Same products in the same order as k_dgemm, the same per-element arithmetic as k_dense_epoch: chains take the decisions of the
// step-synchronous kernels (tests/test_gpu_parity.py: test_dense_epoch_kernel_equals_step_synchronous_kernels, all shapes).
// ================================================================================================
constexpr int DE2_RT = 4, DE2_RW = 16 * DE2_RT;

// matrices in fragment order: out[(((kk·NW + w)·RT + j)·64 + lane)·2 + e], fragment f = 2j + e of wave w: m = f / RT (0: A0 → g′, 1: A1 → w′),
// row tile t = f % RT; MFMA row i = lane mod 16 ↦ matrix row w·16RT + 4RT·iq + 4t + iv with (iq, iv) = (i mod 4, i / 4) for f64 and
// (i / 4, i mod 4) for f32 (Mfma<T>::row): the accumulators of lane (q, n) are rows w·16RT + 4RT·q + 4t + v of chain n
template <class T>
__global__ __launch_bounds__(256) void k_dense_swizzle2(const T* __restrict__ A0, const T* __restrict__ A1, T* __restrict__ out, int D, int NW) {
  constexpr int RT = DE2_RT;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2 * (int64_t)D * D) return;
  const int e = (int)(idx & 1), l = (int)((idx >> 1) & 63);
  int64_t rest = idx >> 7;
  const int j = (int)(rest % RT);
  rest /= RT;
  const int w = (int)(rest % NW), kk = (int)(rest / NW);
  const int f = 2 * j + e, m = f / RT, t = f % RT, i = l & 15;
  const int iq = sizeof(T) == 8 ? (i & 3) : (i >> 2), iv = sizeof(T) == 8 ? (i >> 2) : (i & 3);
  const int row = w * 16 * RT + 4 * RT * iq + 4 * t + iv, k = 4 * kk + (l >> 4);
  out[idx] = (m ? A1 : A0)[row + (int64_t)k * D];
}

template <class T, int NW, int NCT, int WPE, int CRIT = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_dense_epoch2(KP<T> p, DP2<T> q, const T* __restrict__ Asw, int max_steps) {
  using M = Mfma<T>;
  typedef T T2 __attribute__((ext_vector_type(2)));
  constexpr int RT = DE2_RT, RW = DE2_RW, D = RW * NW, NK = D / 4, NF = 2 * RT;
  constexpr int CH = 16 * NCT;  // chains of the workgroup
  constexpr int LPC = RW / 2;   // lanes per chain in the coalesced passes of the epilogue (one element pair per lane)
  constexpr int CPI = 64 / LPC; // chains per instruction there
  constexpr int TP = RW + 2;    // pitch of the transposition tile: conflict-free columns
  constexpr int GL = 16;        // tree phase: lanes per chain, four chains per wave at once
  constexpr int VCH = (WPE >= 4 || CRIT != 1) ? 4 : 8;  // pairs per lane and vector in flight in the tree phase's vector passes
  constexpr int NIT = 16 / CPI, NB = WPE >= 4 ? 2 : (NIT < AHMC_EPOCH_NB ? NIT : AHMC_EPOCH_NB);  // epilogue: chains per half-wave whose loads are in flight together
  static_assert(NW >= 4 && NW <= 16 && (NCT == 1 || NCT == 2), "k_dense_epoch2: D = 256 … 1024 in steps of 64");
  __shared__ T tile[NW][16][TP];
  __shared__ DEMeta<T> meta[NW][CH];
  __shared__ double red[NW][CH][2];
  __shared__ int any_active[2];
  __shared__ int spec_pt[CH];  // per chain: the pool point its speculative half-step goes to (−1: none), set by the tree phase
  const int64_t j0 = (int64_t)blockIdx.x * CH;
  if (j0 >= q.n_list) return;
  const int64_t c_safe = q.list ? (int64_t)q.list[j0] : j0;
  if (threadIdx.x == 0) any_active[0] = any_active[1] = 0;
  if (threadIdx.x < CH) spec_pt[threadIdx.x] = -1;
  __syncthreads();
  constexpr size_t AKS = (size_t)NW * RT * 64;  // T2 elements per k-step
#ifdef AHMC_EPOCH_PROF
  unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc_ = __builtin_readcyclecounter();
#endif
  for (int step = 0; step < max_steps; ++step) {
    // (as in k_dense_epoch: every per-lane quantity is derived anew per step and per phase from a thread index the compiler cannot
    // trace, so that nothing a lane holds lives across the divergent tree phase)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, w = tid >> 6, qd = lane >> 4, n16 = lane & 15;
    const bool tmp = p.lf.kind == 2;                      // TemperedLeapfrog (src/integrator.jl:198-209)
    const T sqa = tmp ? p.lf.sqrt_alpha : T(1);            // √α (not `sa`: the tree phase has a sum of that name)
    const T2* Ap = reinterpret_cast<const T2*>(Asw) + (size_t)w * RT * 64 + lane;
    // ---- the columns of this step: the point each chain's leapfrog in flight sits on ----
    const T* Bp[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int64_t jc = j0 + 16 * ct + n16;
      const int64_t colc = jc < q.n_list ? (q.list ? (int64_t)q.list[jc] : jc) : -1;
      const int64_t cc = colc >= 0 ? colc : c_safe;
      const int cur = q.ptcur[cc];
      const DChain2<T>& Sc = q.S[cc];
      int act = (colc >= 0 && Sc.phase != DPH_IDLE) ? 1 : 0;
      const T ec = act ? q.es[cc] : T(0);
      if (!q.lazy_gw || ec == T(0) || Sc.leaf == (1 << Sc.jw) || Sc.it + 1 >= q.n_trans) act |= 2 * act;   // (g′, w′ go on record: k_dense_epoch)
      Bp[ct] = ppt(q, p, cur, PV_TH, cc) + qd;
      if (qd == 0) meta[w][16 * ct + n16] = DEMeta<T>{(long long)cc, ec, cur, act};
    }
    // ---- g′ = Pθ′, w′ = (M⁻¹P)θ′: 2·RT row tiles × NCT column tiles per wave, operands two k-steps ahead in registers ----
    typename M::acc_t acc[NF][NCT];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[f][ct] = typename M::acc_t{0, 0, 0, 0};
    AHMC_EPOCH_TICK(0)
    {
      T2 a[3][RT];
      T b[3][NCT];
      auto load = [&](int kk, T2 (&aa)[RT], T (&bb)[NCT]) {
        kk = kk < NK ? kk : NK - 1;
#pragma unroll
        for (int j = 0; j < RT; ++j) aa[j] = Ap[(size_t)kk * AKS + (size_t)j * 64];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bb[ct] = Bp[ct][4 * kk];
      };
      auto mult = [&](const T2 (&aa)[RT], const T (&bb)[NCT]) {
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[2 * j + e][ct] = M::mma(aa[j][e], bb[ct], acc[2 * j + e][ct]);
      };
      load(0, a[0], b[0]);
      load(1, a[1], b[1]);
      for (int kk = 0; kk < NK; kk += 3) {
        load(kk + 2, a[2], b[2]);
        mult(a[0], b[0]);
        if (kk + 1 < NK) {
          load(kk + 3, a[0], b[0]);
          mult(a[1], b[1]);
        }
        if (kk + 2 < NK) {
          load(kk + 4, a[1], b[1]);
          mult(a[2], b[2]);
        }
      }
    }
    AHMC_EPOCH_TICK(1)
    // ---- epilogue: second half of the leapfrog (src/integrator.jl:243-250), one matrix at a time through the wave's tile.  The
    // accumulators hold 4·RT consecutive rows of ONE chain per lane and sixteen chains per instruction; read back they are one
    // element pair per lane with LPC consecutive lanes per chain, so every global access below covers whole cache lines ----
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int pos = lane % LPC, dd = w * RW + 2 * pos;
      T2 rkeep[NIT];
      T s1keep[NIT];
      // -- pass G: r ← r½ − ϵ/2·g′ (+ the speculative next r½, the record of g′), θ′·g′ --
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int r0 = 4 * RT * qd + 4 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<T2*>(&tile[w][n16][r0 + 2 * h]) = T2{acc[t][ct][2 * h], acc[t][ct][2 * h + 1]};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ib = 0; ib < NIT; ib += NB) {
        T2 r2[NB], th2[NB];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const DEMeta<T> m = meta[w][16 * ct + (ib + it) * CPI + lane / LPC];
          th2[it] = r2[it] = T2{0, 0};  // (defined on every path: a value left undefined for the idle chains' lanes is carried around the step loop)
          if (m.act) {
            th2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_TH, (int64_t)m.c) + dd);
            r2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_R, (int64_t)m.c) + dd);
          }
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const int nn = (ib + it) * CPI + lane / LPC;
          const DEMeta<T> m = meta[w][16 * ct + nn];
          T2 rr = r2[it];
          T s1 = 0;
          if (m.act) {
            const T e = m.e;
            const T2 g2 = *reinterpret_cast<const T2*>(&tile[w][nn][2 * pos]);
            const int sp = spec_pt[16 * ct + nn];
            if (e != T(0)) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                rr[h] = rr[h] - e / 2 * g2[h];
                if (tmp) rr[h] = rr[h] / sqa;   // TemperedLeapfrog: r/√α after the second half-step (k_d_tree2)
              }
              DE_STORE(ppt(q, p, m.cur, PV_R, (int64_t)m.c) + dd, rr);
              if (sp >= 0) {  // the first half of the NEXT leapfrog if the tree goes on from this point with this step (k_dense_epoch)
                T2 rh;
#pragma unroll
                for (int h = 0; h < 2; ++h) rh[h] = (tmp ? rr[h] * sqa : rr[h]) - e / 2 * g2[h];
                DE_STORE(ppt(q, p, sp, PV_R, (int64_t)m.c) + dd, rh);
              }
            }
            if ((m.act & 2) || e == T(0) || sp < 0) DE_STORE(ppt(q, p, m.cur, PV_G, (int64_t)m.c) + dd, g2);
            s1 = th2[it][0] * g2[0] + th2[it][1] * g2[1];
          }
          rkeep[ib + it] = rr;
          s1keep[ib + it] = s1;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // -- pass W: v ← v½ − ϵ/2·w′ (+ the speculative v½ and θ″ = θ′ + ϵ·v½, the record of w′), r·v --
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int r0 = 4 * RT * qd + 4 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<T2*>(&tile[w][n16][r0 + 2 * h]) = T2{acc[RT + t][ct][2 * h], acc[RT + t][ct][2 * h + 1]};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ib = 0; ib < NIT; ib += NB) {
        T2 v2[NB], th2[NB];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const DEMeta<T> m = meta[w][16 * ct + (ib + it) * CPI + lane / LPC];
          th2[it] = v2[it] = T2{0, 0};
          if (m.act) {
            v2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_V, (int64_t)m.c) + dd);
            if (m.e != T(0) && spec_pt[16 * ct + (ib + it) * CPI + lane / LPC] >= 0) th2[it] = *reinterpret_cast<const T2*>(ppt(q, p, m.cur, PV_TH, (int64_t)m.c) + dd);
          }
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const int nn = (ib + it) * CPI + lane / LPC;
          const DEMeta<T> m = meta[w][16 * ct + nn];
          T s0 = 0, s1 = s1keep[ib + it];
          if (m.act) {
            const T e = m.e;
            const T2 w2 = *reinterpret_cast<const T2*>(&tile[w][nn][2 * pos]);
            const int sp = spec_pt[16 * ct + nn];
            T2 vv = v2[it];
            if (e != T(0)) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                vv[h] = vv[h] - e / 2 * w2[h];
                if (tmp) vv[h] = vv[h] / sqa;
              }
              DE_STORE(ppt(q, p, m.cur, PV_V, (int64_t)m.c) + dd, vv);
              if (sp >= 0) {
                T2 vh, tn;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  vh[h] = (tmp ? vv[h] * sqa : vv[h]) - e / 2 * w2[h];
                  tn[h] = th2[it][h] + e * vh[h];
                }
                DE_STORE(ppt(q, p, sp, PV_V, (int64_t)m.c) + dd, vh);
                DE_STORE(ppt(q, p, sp, PV_TH, (int64_t)m.c) + dd, tn);
              }
            }
            if ((m.act & 2) || e == T(0) || sp < 0) DE_STORE(ppt(q, p, m.cur, PV_W, (int64_t)m.c) + dd, w2);
            const T2 rr = rkeep[ib + it];
            s0 = rr[0] * vv[0] + rr[1] * vv[1];
          }
          wave_allsum2<LPC>(s0, s1);
          if (pos == 0) { red[w][16 * ct + nn][0] = (double)s0; red[w][16 * ct + nn][1] = (double)s1; }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    AHMC_EPOCH_TICK(2)
    __syncthreads();  // (A) the points are complete, the partial sums are in LDS
    AHMC_EPOCH_TICK(3)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    if (tid2 == 0) any_active[(step + 1) & 1] = 0;
    // ---- trees: a wave serves four chains AT ONCE, GL lanes each (k_d_tree2 from the reduction on; the groups of a wave follow their own
    // control flow — every exchange of d_tree_advance2 stays inside a 16-lane row); round rd: chains 4·NW·rd … of the workgroup ----
#pragma unroll 1
    for (int base = 0; base < CH; base += 4 * NW) {
      const int lane2 = tid2 & 63, w2 = tid2 >> 6;
      // (whether a wave has chains in this round is wave-uniform: said so, it is a scalar branch instead of one more exec-mask region for the
      // register allocator to park a spill in — the 12- and 16-wave shapes were refused by isa_check.py for exactly that)
      // (NW >= 12 only: the shapes with fewer waves were measured and scanned in the form without it, and every change to this phase moves
      // the register allocator's spills somewhere else)
      constexpr bool WIDE = NW >= 12;
      if constexpr (WIDE) {
        if (base + 4 * __builtin_amdgcn_readfirstlane(w2) >= CH) continue;
      }
      const int grp = lane2 / GL, l16_in = lane2 % GL;
      const int slot_in = base + w2 * 4 + grp;
      const int64_t jj = j0 + slot_in;
      bool chain_active = false;
      if ((WIDE || slot_in < CH) && jj < q.n_list) {
        const int64_t c_in = q.list ? (int64_t)q.list[jj] : jj;
        const DChain2<T>& S_in = q.S[c_in];
        if (S_in.phase != DPH_IDLE) {
          T sa = 0, sb = 0;
#pragma unroll
          for (int k = 0; k < NW; ++k) { sa += (T)red[k][slot_in][0]; sb += (T)red[k][slot_in][1]; }
          const T lk = sanitize(-sa / 2), lp = sanitize(-sb / 2);
          DHot<T> hot;
          hot.H0 = S_in.H0; hot.eps = S_in.eps; hot.lu = S_in.lu;
          hot.phase = S_in.phase; hot.it = S_in.it; hot.jw = S_in.jw; hot.leaf = S_in.leaf; hot.v = S_in.v; hot.cur_is_left = S_in.cur_is_left; hot.numerical = S_in.numerical;
          hot.k = S_in.k;
          hot.cur = S_in.cur; hot.oth = S_in.oth; hot.cand = S_in.cand; hot.edge = S_in.edge;
          if (l16_in == 0) {
            p.lk()[c_in] = lk;
            p.lp()[c_in] = lp;
          }
          int src = hot.cur;
          uint64_t used = 0;
          bool rewritten = false;
          dn_chain_barrier<true>();
          const int cur_in = hot.cur;
          const T e = d_tree_advance2<T, GL, true, D, VCH, CRIT>(p, q, c_in, l16_in, lp, lk, hot, src, used, rewritten);
          // Everything a lane needs from here on is derived AGAIN from a thread index the compiler cannot trace (as at the top of every
          // phase): the chain index and the address of its record do not live across the tree bookkeeping — where the register allocator
          // spilled them under the inner exec mask and reloaded them outside it (the 16-wave shapes were refused by isa_check.py).
          int tid3 = threadIdx.x;
          if constexpr (WIDE) asm volatile("" : "+v"(tid3));
          const int w3 = WIDE ? tid3 >> 6 : w2, l16 = WIDE ? (tid3 & 63) % GL : l16_in, slot = WIDE ? base + w3 * 4 + (tid3 & 63) / GL : slot_in;
          const int64_t jj3 = j0 + slot;
          const int64_t c = WIDE ? (q.list ? (int64_t)q.list[jj3] : jj3) : c_in;
          DChain2<T>& S = q.S[c];
          if (l16 == 0) q.es[c] = e;
          if (e != T(0)) {  // first half of the next leapfrog (src/integrator.jl:231-237): src → a fresh point
            const DEMeta<T> m = meta[w3][slot];
            const int sp = spec_pt[slot];
            // the epilogue has already taken it if the leapfrog goes on from the point just completed with the same signed step
            const bool hit = sp >= 0 && m.e != T(0) && src == cur_in && e == m.e && !rewritten;
            const int dst = hit ? sp : __builtin_ctzll(~used);
            T* dTH = ppt(q, p, dst, PV_TH, c);
            T* dR = ppt(q, p, dst, PV_R, c);
            T* dV = ppt(q, p, dst, PV_V, c);
            dn_chain_barrier<true>();  // (the start of a transition has just written r, v of `src`)
            const T* const srcs[5] = {ppt(q, p, src, PV_TH, c), ppt(q, p, src, PV_R, c), ppt(q, p, src, PV_G, c), ppt(q, p, src, PV_V, c), ppt(q, p, src, PV_W, c)};
            if (l16 == 0) spec_pt[slot] = __builtin_ctzll(~(used | ((uint64_t)1 << dst)));
            if (!hit) dn_vec_pass<T, GL, D, 5, VCH>(l16, srcs, [&](int d, const T2 (&x)[5]) {
              T2 rh, vh, tn;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                rh[h] = (tmp ? x[1][h] * sqa : x[1][h]) - e / 2 * x[2][h];
                vh[h] = (tmp ? x[3][h] * sqa : x[3][h]) - e / 2 * x[4][h];
                tn[h] = x[0][h] + e * vh[h];
              }
              *reinterpret_cast<T2*>(dR + d) = rh;
              *reinterpret_cast<T2*>(dV + d) = vh;
              *reinterpret_cast<T2*>(dTH + d) = tn;
            });
            if (l16 == 0) {
              S.cur = (int8_t)dst;
              q.ptcur[c] = dst;
            }
            chain_active = true;
          } else {
            if (l16 == 0) {
              q.ptcur[c] = src;  // motionless / idle: the products serve the point the chain sits on
              spec_pt[slot] = -1;
            }
            dn_chain_barrier<true>();        // (thread 0 may just have set the phase)
            if (S.phase != DPH_IDLE) chain_active = true;
          }
        }
      }
      if (chain_active && (WIDE || l16_in == 0)) any_active[step & 1] = 1;   // (WIDE: every lane of an active chain writes the same 1)
    }
    AHMC_EPOCH_TICK(4)
    __syncthreads();  // (B) every chain's next point, step and phase are visible to the whole workgroup
    AHMC_EPOCH_TICK(5)
#ifdef AHMC_EPOCH_PROF
    pt_[7] += 1;
#endif
    if (!any_active[step & 1]) break;
  }
#ifdef AHMC_EPOCH_PROF
  if (threadIdx.x == 0 && q.prof)
    for (int i = 0; i < 8; ++i) atomicAdd(q.prof + i, pt_[i]);
#endif
}

// ------------------------------------------------------------------------------------------------
// static HMC, EndPointTS (src/trajectory.jl:271-340, :855-880): begin = jitter + keep the start point;
// end = MH accept (H′ < H + Exp(1)), rejected chains revert, momentum flip, statistics.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_d_hmc_begin(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  const int D = p.D;
  Rng rng = make_rng(p, c);
  const T eps = chain_eps(p, rng, c);
  const T* th = p.th() + c * D;
  const T* r = p.r() + c * D;
  const T* g = p.g() + c * D;
  T* s_th = dslot(q, p, DS_START_TH, c);
  T* s_r = dslot(q, p, DS_START_R, c);
  T* s_g = dslot(q, p, DS_START_G, c);
  for (int d = lane; d < D; d += 64) {
    s_th[d] = th[d];
    s_r[d] = r[d];
    s_g[d] = g[d];
  }
  if (lane == 0) {
    DChain<T>& S = q.S[c];
    S.H0 = -(p.lp()[c] + p.lk()[c]);
    S.cand_lp = p.lp()[c];
    S.cand_lk = p.lk()[c];
    S.eps = eps;
    p.eps_cur()[c] = eps;
    q.es[c] = eps;
  }
}
template <class T>
__global__ __launch_bounds__(256) void k_d_hmc_end(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  const int D = p.D;
  DChain<T>& S = q.S[c];
  Rng rng = make_rng(p, c);
  const T H0 = S.H0;
  const T lp1 = p.lp()[c], lk1 = p.lk()[c];
  const T H1 = -(lp1 + lk1);
  const T e = (T)rng.randexp(RNG_TRANSITION, 0);
  const bool accept = H1 < H0 + e;                       // mh_accept_ratio (:855-880)
  const T alpha = jl_min(T(1), exp(H0 - H1));
  T* th = p.th() + c * D;
  T* r = p.r() + c * D;
  T* g = p.g() + c * D;
  const T* s_th = dslot(q, p, DS_START_TH, c);
  const T* s_r = dslot(q, p, DS_START_R, c);
  const T* s_g = dslot(q, p, DS_START_G, c);
  T* s1 = p.acc_sum() + c * D;
  T* s2 = p.acc_sumsq() + c * D;
  for (int d = lane; d < D; d += 64) {
    T t = th[d], rr = r[d], gg = g[d];
    if (!accept) { t = s_th[d]; rr = s_r[d]; gg = s_g[d]; }  // accept_phasepoint! (:312-332)
    th[d] = t;
    r[d] = -rr;  // z = PhasePoint(z.θ, -z.r, ...) (:283)
    g[d] = gg;
    if (p.accum) { s1[d] += t; s2[d] += t * t; }
  }
  if (lane == 0) {
    const T lp = accept ? lp1 : S.cand_lp, lk = accept ? lk1 : S.cand_lk;
    const T H = -(lp + lk);
    p.lp()[c] = lp;
    p.lk()[c] = lk;
    p.st_nsteps()[c] = (int32_t)p.L;
    p.st_accept()[c] = accept ? 1 : 0;
    p.st_accrate()[c] = alpha;
    p.st_logdens()[c] = lp;
    p.st_H()[c] = H;
    p.st_Herr()[c] = H - H0;
    p.st_maxHerr()[c] = 0;
    p.st_depth()[c] = 0;
    p.st_numerr()[c] = (isfinite(lp1) && isfinite(lk1)) ? 0 : 1;
    if (p.accum) {
      p.acc_nsteps()[c] += p.L;
      p.acc_ndiv()[c] += (isfinite(lp1) && isfinite(lk1)) ? 0 : 1;
      accumulate_energy(p, c, H);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// find_good_stepsize (src/trajectory.jl:768-837) as a per-chain state machine.  DChain fields reused:
// H0 = H(z0), eps / lu = the bracket (ϵ, ϵ′), w_tree = the ϵ being evaluated, phase = 0 first evaluation,
// 1 doubling/halving until the acceptance ratio crosses ½ (Q3: A is evaluated at ϵ, not ϵ′), 2 bisection to
// (¼, ¾), 3 done; v = too_high; it = iteration counter.  z0 is kept in the OTH slots.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_d_fe_save(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  const int D = p.D;
  vcopy(dslot(q, p, DS_START_TH, c), p.th() + c * D, D, lane);
  vcopy(dslot(q, p, DS_START_R, c), p.r() + c * D, D, lane);
  vcopy(dslot(q, p, DS_START_G, c), p.g() + c * D, D, lane);
  if (lane == 0) {
    q.S[c].cand_lp = p.lp()[c];
    q.S[c].cand_lk = p.lk()[c];
  }
}
template <class T>
__global__ __launch_bounds__(256) void k_d_fe_begin(KP<T> p, DP<T> q, int* n_active) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c == 0 && lane == 0) *n_active = (int)p.N;
  if (c >= p.N) return;
  const int D = p.D;
  vcopy(dslot(q, p, DS_OTH_TH, c), p.th() + c * D, D, lane);
  vcopy(dslot(q, p, DS_OTH_R, c), p.r() + c * D, D, lane);
  vcopy(dslot(q, p, DS_OTH_G, c), p.g() + c * D, D, lane);
  vcopy(dslot(q, p, DS_OTH_V, c), dslot(q, p, DS_CUR_V, c), D, lane);
  if (q.dense_metric) vcopy(dslot(q, p, DS_OTH_W, c), dslot(q, p, DS_CUR_W, c), D, lane);
  if (lane == 0) {
    DChain<T>& S = q.S[c];
    S.H0 = -(p.lp()[c] + p.lk()[c]);
    S.sub_lp = p.lp()[c];  // ℓπ, ℓκ of z0 (restored with it)
    S.sub_lk = p.lk()[c];
    S.eps = p.init_eps;
    S.lu = p.init_eps;
    S.w_tree = p.init_eps;
    S.phase = 0;
    S.it = 0;
    S.v = 0;
    q.es[c] = p.init_eps;
  }
}
template <class T>
__global__ __launch_bounds__(256) void k_d_fe_iter(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  DChain<T>& S = q.S[c];
  int phase = S.phase;
  if (phase == 3) return;
  const int D = p.D;
  const T log_a_min = 2 * log(T(0.5)), log_a_cross = log(T(0.5)), log_a_max = log(T(0.75));
  const T dH = S.H0 - (-(p.lp()[c] + p.lk()[c]));  // H − A(ϵ_eval)
  T eps = S.eps, epsp = S.lu;
  int it = S.it;
  bool too_high = S.v != 0;
  auto to_bisection = [&]() {
    const T lo = jl_min(eps, epsp), hi = jl_max(eps, epsp);
    eps = lo;
    epsp = hi;
    it = 0;
    phase = p.max_iters > 0 ? 2 : 3;
  };
  if (phase == 0) {
    too_high = dH > log_a_cross;
    if (p.max_iters > 0) phase = 1;
    else to_bisection();
  } else if (phase == 1) {
    epsp = too_high ? 2 * eps : eps / 2;
    if (too_high != (dH > log_a_cross)) {
      to_bisection();
    } else {
      eps = epsp;
      if (++it >= p.max_iters) to_bisection();
    }
  } else {  // bisection: the evaluation was at mid = S.w_tree
    const T mid = S.w_tree;
    bool done = false;
    if (dH > log_a_max) eps = mid;
    else if (dH < log_a_min) epsp = mid;
    else { eps = mid; done = true; }
    if (done || ++it >= p.max_iters) phase = 3;
  }
  const T eval = phase == 2 ? eps / 2 + epsp / 2 : eps;
  // rewind to z0 for the next evaluation
  vcopy(p.th() + c * D, dslot(q, p, DS_OTH_TH, c), D, lane);
  vcopy(p.r() + c * D, dslot(q, p, DS_OTH_R, c), D, lane);
  vcopy(p.g() + c * D, dslot(q, p, DS_OTH_G, c), D, lane);
  vcopy(dslot(q, p, DS_CUR_V, c), dslot(q, p, DS_OTH_V, c), D, lane);
  if (q.dense_metric) vcopy(dslot(q, p, DS_CUR_W, c), dslot(q, p, DS_OTH_W, c), D, lane);
  if (lane == 0) {
    p.lp()[c] = S.sub_lp;
    p.lk()[c] = S.sub_lk;
    S.eps = eps;
    S.lu = epsp;
    S.w_tree = eval;
    S.it = it;
    S.v = too_high ? 1 : 0;
    S.phase = phase;
    q.es[c] = phase == 3 ? T(0) : eval;
    if (phase == 3) atomicSub(q.n_active, 1);
  }
}
template <class T>
__global__ __launch_bounds__(256) void k_d_fe_end(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  const int D = p.D;
  vcopy(p.th() + c * D, dslot(q, p, DS_START_TH, c), D, lane);
  vcopy(p.r() + c * D, dslot(q, p, DS_START_R, c), D, lane);
  vcopy(p.g() + c * D, dslot(q, p, DS_START_G, c), D, lane);
  if (lane == 0) {
    p.lp()[c] = q.S[c].cand_lp;
    p.lk()[c] = q.S[c].cand_lk;
    p.eps_cur()[c] = q.S[c].eps;
    q.es[c] = T(0);
  }
}

// ------------------------------------------------------------------------------------------------
// WelfordCov behind the shared DenseEuclideanMetric (src/adaptation/massmatrix.jl:283-340).  The oracle pushes the
// N chains' positions one after another; here one adapt! folds the whole batch in at once (Chan et al.):
//   m_b = mean_c x_c,  S_b = Σ_c (x_c − m_b)(x_c − m_b)ᵀ  (a rank-N update on the MFMA units),
//   δ = m_b − μ,  M += S_b + δδᵀ·n·N/(n+N),  μ += δ·N/(n+N),  n += N.
// ------------------------------------------------------------------------------------------------
constexpr int COV_SLICES = 64;
template <class T>
__global__ __launch_bounds__(256) void k_d_colsum_partial(const T* __restrict__ X, T* __restrict__ partial, int D, int64_t N) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int sl = blockIdx.y;
  if (d >= D) return;
  const int64_t per = (N + COV_SLICES - 1) / COV_SLICES;
  const int64_t n0 = sl * per, n1 = n0 + per < N ? n0 + per : N;
  T s = 0;
  for (int64_t n = n0; n < n1; ++n) s += X[d + n * D];
  partial[(int64_t)sl * D + d] = s;
}
template <class T>
__global__ __launch_bounds__(256) void k_d_colsum_final(const T* __restrict__ partial, T* __restrict__ mean, int D, int64_t N) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  T s = 0;
  for (int sl = 0; sl < COV_SLICES; ++sl) s += partial[(int64_t)sl * D + d];  // fixed order
  mean[d] = s / (T)N;
}
// S (D,D) = Σ_n (x_n − m)(x_n − m)ᵀ: 64×64 tile per workgroup, K = the N chains, 2×2 MFMA tiles per wave
template <class T>
__global__ __launch_bounds__(256) void k_dsyrk(const T* __restrict__ X, const T* __restrict__ mean, T* __restrict__ S, int D, int64_t N) {
  __shared__ T As[GB_K][GB_M + GB_PAD];
  __shared__ T Bs[GB_K][GB_N + GB_PAD];
  using M = Mfma<T>;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;
  const int wm = (w & 1) * 32, wn = (w >> 1) * 32;
  typename M::acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = typename M::acc_t{0, 0, 0, 0};
  const int ai = (tid & 31) * 2, ak = tid >> 5;  // rows ai, ai+1 of chains ak and ak+8 (both operand tiles)
  T ma[2], mb[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    ma[e] = m0 + ai + e < D ? mean[m0 + ai + e] : T(0);
    mb[e] = n0 + ai + e < D ? mean[n0 + ai + e] : T(0);
  }
  for (int64_t k0 = 0; k0 < N; k0 += GB_K) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t n = k0 + ak + 8 * q;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ia = m0 + ai + e, ib = n0 + ai + e;
        As[ak + 8 * q][ai + e] = (n < N && ia < D) ? X[ia + n * D] - ma[e] : T(0);
        Bs[ak + 8 * q][ai + e] = (n < N && ib < D) ? X[ib + n * D] - mb[e] : T(0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < GB_K / 4; ++ks) {
      const int kq = ks * 4 + (lane >> 4), l16 = lane & 15;
      const T a0 = As[kq][wm + l16], a1 = As[kq][wm + 16 + l16];
      const T b0 = Bs[kq][wn + l16], b1 = Bs[kq][wn + 16 + l16];
      acc[0][0] = M::mma(a0, b0, acc[0][0]);
      acc[0][1] = M::mma(a0, b1, acc[0][1]);
      acc[1][0] = M::mma(a1, b0, acc[1][0]);
      acc[1][1] = M::mma(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + wn + tj * 16 + (lane & 15);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = m0 + wm + ti * 16 + M::row(lane, v);
        if (row < D && col < D) S[row + (int64_t)col * D] = acc[ti][tj][v];
      }
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_d_cov_combine(T* __restrict__ Mc, const T* __restrict__ mu, const T* __restrict__ mb, const T* __restrict__ Sb,
                                                       T n, T nb, int D) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * D) return;
  const int i = (int)(idx % D), j = (int)(idx / D);
  const T di = mb[i] - mu[i], dj = mb[j] - mu[j];
  Mc[idx] = Mc[idx] + Sb[idx] + di * dj * (n * nb / (n + nb));
}
template <class T>
__global__ __launch_bounds__(256) void k_d_cov_mean(T* __restrict__ mu, const T* __restrict__ mb, T n, T nb, int D) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < D) mu[d] = mu[d] + (mb[d] - mu[d]) * (nb / (n + nb));
}
// get_estimation(wc) (:333-340): n/((n+5)(n−1))·M + 10⁻³·5/(n+5)·I
template <class T>
__global__ __launch_bounds__(256) void k_d_cov_estimate(const T* __restrict__ Mc, T* __restrict__ cov, T n, int D) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * D) return;
  const int i = (int)(idx % D), j = (int)(idx / D);
  cov[idx] = n / ((n + 5) * (n - 1)) * Mc[idx] + (i == j ? T(1e-3) * (5 / (n + 5)) : T(0));
}

}  // namespace ahmc
