// ahmc_kernels.hpp — the chain-batched kernels (gfx950).  One group of G lanes per chain, all
// per-chain state in registers, one launch per whole transition.
#pragma once

#include "ahmc_device.hpp"
#include "ahmc_hip.h"

namespace ahmc {

// Everything a kernel needs, passed by value.  Arrays are (D,N) column-major / (N,).
// The context keeps its per-chain arrays in four slabs so that the kernel arguments stay small
// (a fat kernarg struct exhausts the 102 SGPRs and pushes every uniform value into VGPRs):
//   vbase: T[5][D*N]  θ, r, -∇ℓπ, Σθ, Σθ²          tbase: T[14][N]  ℓπ, ℓκ, ϵ_nom, ϵ_cur, stats…, energy sums (EBFMI)
//   ibase: int32[4][N] n_steps, is_accept, depth, numerical_error   lbase: int64[2][N] Σn_steps, Σdiv
template <class T>
struct KP {
  int D;
  int64_t N;
  T* vbase;
  T* tbase;
  int32_t* ibase;
  long long* lbase;
  __device__ __forceinline__ T* vec(int k) const { return vbase + (int64_t)k * D * N; }
  __device__ __forceinline__ T* sca(int k) const { return tbase + (int64_t)k * N; }
  __device__ __forceinline__ T* th() const { return vec(0); }
  __device__ __forceinline__ T* r() const { return vec(1); }
  __device__ __forceinline__ T* g() const { return vec(2); }
  __device__ __forceinline__ T* acc_sum() const { return vec(3); }
  __device__ __forceinline__ T* acc_sumsq() const { return vec(4); }
  __device__ __forceinline__ T* lp() const { return sca(0); }
  __device__ __forceinline__ T* lk() const { return sca(1); }
  __device__ __forceinline__ T* eps_nom() const { return sca(2); }
  __device__ __forceinline__ T* eps_cur() const { return sca(3); }
  __device__ __forceinline__ T* st_accrate() const { return sca(4); }
  __device__ __forceinline__ T* st_logdens() const { return sca(5); }
  __device__ __forceinline__ T* st_H() const { return sca(6); }
  __device__ __forceinline__ T* st_Herr() const { return sca(7); }
  __device__ __forceinline__ T* st_maxHerr() const { return sca(8); }
  // running sums over the energies of the kept transitions (EBFMI, src/diagnosis.jl:1-3): n, E_prev, Σ(E_i − E_{i−1})²,
  // Welford mean and M2 of E
  __device__ __forceinline__ T* acc_energy() const { return sca(9); }
  __device__ __forceinline__ int32_t* st_nsteps() const { return ibase; }
  __device__ __forceinline__ int32_t* st_accept() const { return ibase + N; }
  __device__ __forceinline__ int32_t* st_depth() const { return ibase + 2 * N; }
  __device__ __forceinline__ int32_t* st_numerr() const { return ibase + 3 * N; }
  __device__ __forceinline__ long long* acc_nsteps() const { return lbase; }
  __device__ __forceinline__ long long* acc_ndiv() const { return lbase + N; }
  // metric (Unit: minv == nullptr)
  const T* minv;
  const T* sqrt_minv;
  int minv_per_chain;
  // integrator
  LeapfrogP<T> lf;
  T jitter;
  TargetP<T> tp;
  // rng
  uint32_t k0, k1, chain_offset, chain_stride, iteration;
  int accum;  // accumulate Σn_steps, Σθ, Σθ² for this (kept) transition
  int no_lk;  // k_fill_caches: leave ℓκ alone (dense metric: the dense engine computes it)
  T refresh_alpha;
  // NUTS
  int max_depth;
  T delta_max;
  int criterion, sampler;
  T* scratch;           // vector slots that do not fit in LDS
  const void* adaptk;   // k_nuts MODE 3: AdaptK<T> in device memory (in-kernel adaptation), else null
  const int* order;     // k_nuts: chain handled by group slot i (longest expected trees first), or null = identity
  unsigned int n_chunks;
  int n_lds_levels;     // number of vector slots held in LDS (hottest first)
  int32_t* redo;        // per-chain flag: linear-domain weights came near overflow → redo in log domain
  int redo_only;        // 1: process only the flagged chains
  int n_trans;          // transitions per launch of k_nuts
  const T* znorm;       // (n_trans, D, N) standard normals of the momentum draws (k_normals)
  T* samples_out;       // optional (n_trans, D, N) device buffer receiving θ after every transition
  // static HMC
  int64_t L;
  T* hmc_H;  // (L+1, N) energies for MultinomialTS
  // misc
  int64_t n_steps;  // k_leapfrog
  T init_eps;
  int max_iters;
};
// LDS / scratch layout of k_nuts (ahmc_nuts.hpp), needed by the host launch plan as well
constexpr int NUTS_NSC = 3;      // T scalars per pending level: w, Σα, ΔH_max
constexpr int NUTS_NSI = 2;      // int scalars per pending level: nα, candidate leaf index
constexpr int NUTS_NAT = 5;      // per chain, T: the adaptor's state while a warm-up batch runs (nominal ϵ, DAState ϵ, μ, x̄, H̄)
constexpr int NUTS_NAI = 3;      // per chain, int: DAState m, Welford count, the re-integration checkpoint (position << 1 | triple)
constexpr int NUTS_DORMANT = 7;  // vector slots OTH_TH, OTH_R, OTH_G, TREE_A, Z0_R, Z0_G, START_R (strict)
#ifndef AHMC_CKPT
#define AHMC_CKPT 1              // k_nuts re-integrates to the candidate from the EDGE its subtree grew from instead of from z0 (ahmc_nuts.hpp); 0: from z0 (rounds 1–4)
#endif
constexpr int NUTS_CKPT = AHMC_CKPT ? 6 : 0;   // two (θ, r, g) triples behind the dormant slots, always in global scratch
// The layout as ONE number: the host's launch plan (ahmc_api.hip, plan_nuts) sizes the scratch the kernels index, and the two may
// come from different compilations (the instantiation units of a variant build with other -D flags; a target plugin built on
// another machine).  Every launch table carries the word of ITS translation unit; plan_nuts refuses a table whose word is not the host's.
constexpr int64_t nuts_scratch_layout() {
  return (int64_t)NUTS_NSC | (int64_t)NUTS_NSI << 8 | (int64_t)NUTS_NAT << 16 | (int64_t)NUTS_NAI << 24 | (int64_t)NUTS_DORMANT << 32 | (int64_t)NUTS_CKPT << 40;
}

template <class T, int G, int E>
struct Geo {
  static constexpr int CPW = 64 / G;  // chains per wave
  static constexpr int DP = G * E;    // padded dimension
};

template <class T>
__device__ __forceinline__ Rng make_rng(const KP<T>& p, int64_t c) {
  Rng g;
  g.k0 = p.k0;
  g.k1 = p.k1;
  g.chain = p.chain_offset + p.chain_stride * (uint32_t)c;
  g.iter = p.iteration;
  return g;
}

template <class T, int E>
__device__ __forceinline__ void load_minv(const KP<T>& p, int64_t c, int d0, T (&minv)[E]) {
  if (p.minv) {
    load_vec<T, E>(minv, p.minv, p.minv_per_chain ? c * p.D : 0, d0, p.D, T(1));
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) minv[e] = T(1);
  }
}

// jitter(rng, lf) (src/integrator.jl:140-156): ϵ = ϵ0 (1 + jitter (2u − 1))
template <class T>
__device__ __forceinline__ T chain_eps(const KP<T>& p, const Rng& rng, int64_t c) {
  T e0 = p.eps_nom()[c];
  if (p.lf.kind == 1) {
    T u = (T)rng.uniform(RNG_JITTER, 0);
    return e0 * (1 + p.jitter * (2 * u - 1));
  }
  return e0;
}

template <class T>
__device__ __forceinline__ T chain_eps_from(const KP<T>& p, const Rng& rng, T e0) {  // the same with ϵ0 in a register
  if (p.lf.kind == 1) {
    T u = (T)rng.uniform(RNG_JITTER, 0);
    return e0 * (1 + p.jitter * (2 * u - 1));
  }
  return e0;
}

// rand_momentum + refresh (src/metric.jl:290-309, src/hamiltonian.jl:213-254); fills z.r only
template <class T, int E>
__device__ __forceinline__ void draw_momentum(const KP<T>& p, const Rng& rng, uint32_t purpose, int64_t c, int d0,
                                              T (&r)[E], T alpha) {
  T z[E];
  normals<T, E>(rng, purpose, d0, z);
  if (p.minv) {
    T sq[E];
    load_vec<T, E>(sq, p.sqrt_minv, p.minv_per_chain ? c * p.D : 0, d0, p.D, T(1));
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = z[e] / sq[e];  // r ./= sqrtM⁻¹
  }
  if (alpha != T(0)) {  // PartialMomentumRefreshment: α r + sqrt(1-α²) ξ
    T s = sqrt(1 - alpha * alpha);
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = alpha * r[e] + s * z[e];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) r[e] = (d0 + e < p.D) ? z[e] : T(0);
}

// the same refresh with the standard normals read from memory (written by k_normals)
template <class T, int E>
__device__ __forceinline__ void momentum_from_normals(const KP<T>& p, const T* __restrict__ zsrc, int64_t c, int d0,
                                                      T (&r)[E]) {
  T z[E];
  load_vec<T, E>(z, zsrc, c * p.D, d0, p.D, T(0));
  if (p.minv) {
    T sq[E];
    load_vec<T, E>(sq, p.sqrt_minv, p.minv_per_chain ? c * p.D : 0, d0, p.D, T(1));
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = z[e] / sq[e];  // r ./= sqrtM⁻¹
  }
  if (p.refresh_alpha != T(0)) {  // PartialMomentumRefreshment: α r + sqrt(1-α²) ξ, r = the stored momentum
    T ro[E];
    load_vec<T, E>(ro, p.r(), c * p.D, d0, p.D, T(0));
    const T s = sqrt(1 - p.refresh_alpha * p.refresh_alpha);
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = p.refresh_alpha * ro[e] + s * z[e];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) r[e] = (d0 + e < p.D) ? z[e] : T(0);
}

template <class T, int E>
__device__ __forceinline__ void store_point(const KP<T>& p, int64_t c, int d0, int lane, const Point<T, E>& z) {
  store_vec<T, E>(z.th, p.th(), c * p.D, d0, p.D);
  store_vec<T, E>(z.r, p.r(), c * p.D, d0, p.D);
  store_vec<T, E>(z.g, p.g(), c * p.D, d0, p.D);
  if (lane == 0) {
    p.lp()[c] = z.lp;
    p.lk()[c] = z.lk;
  }
}

// one lane per chain: fold the energy H of a kept transition into the chain's running sums
template <class T>
__device__ __forceinline__ void accumulate_energy(const KP<T>& p, int64_t c, T H) {
  T* ea = p.acc_energy();
  const T n0 = ea[c], prev = ea[p.N + c];
  const T n = n0 + 1;
  if (n0 > 0) { const T d = H - prev; ea[2 * p.N + c] += d * d; }
  const T mean = ea[3 * p.N + c], delta = H - mean;
  const T mean2 = mean + delta / n;
  ea[3 * p.N + c] = mean2;
  ea[4 * p.N + c] += delta * (H - mean2);
  ea[c] = n;
  ea[p.N + c] = H;
}

template <class T, int E>
__device__ __forceinline__ void accumulate(const KP<T>& p, int64_t c, int d0, int lane, const T (&th)[E], int n_steps,
                                           int numerr, T H) {
  if (!p.accum) return;
  T s1[E], s2[E];
  load_vec<T, E>(s1, p.acc_sum(), c * p.D, d0, p.D, T(0));
  load_vec<T, E>(s2, p.acc_sumsq(), c * p.D, d0, p.D, T(0));
#pragma unroll
  for (int e = 0; e < E; ++e) {
    s1[e] += th[e];
    s2[e] += th[e] * th[e];
  }
  store_vec<T, E>(s1, p.acc_sum(), c * p.D, d0, p.D);
  store_vec<T, E>(s2, p.acc_sumsq(), c * p.D, d0, p.D);
  if (lane == 0) {
    p.acc_nsteps()[c] += n_steps;
    p.acc_ndiv()[c] += numerr;
    accumulate_energy(p, c, H);
  }
}

#define AHMC_GEOMETRY()                                   \
  const int lane64 = threadIdx.x & 63;                    \
  const int lane = (int)(threadIdx.x & (G - 1));          \
  const int d0 = lane * E;                                \
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; \
  const bool active = c < p.N;                            \
  (void)lane64

// ------------------------------------------------------------------------------------------------
// phasepoint(h, θ, r) for arrays already in the context (ahmc_set_position)
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E, int TK>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_fill_caches(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.th, p.th(), c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.r, p.r(), c * p.D, d0, p.D, T(0));
  fill_caches<T, G, E, TK>(z, minv, p.tp, lane, d0);
  store_vec<T, E>(z.g, p.g(), c * p.D, d0, p.D);
  if (lane == 0) {
    p.lp()[c] = z.lp;
    if (!p.no_lk) p.lk()[c] = z.lk;
  }
}

// kinetic cache only (ahmc_set_phasepoint with caller-supplied ℓπ, -∇ℓπ; ahmc_lf_post)
template <class T, int G, int E>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_kinetic(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  T minv[E], r[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(r, p.r(), c * p.D, d0, p.D, T(0));
  T s = group_sum1<G>(kinetic_partial(r, minv));
  if (lane == 0) {
    p.lk()[c] = sanitize(-s / 2);
    p.lp()[c] = sanitize(p.lp()[c]);
  }
}

// refresh(rng, refreshment, h, z)
template <class T, int G, int E, int TK>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_refresh(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.r, p.r(), c * p.D, d0, p.D, T(0));  // α r + sqrt(1-α²) ξ needs the old momentum
  load_vec<T, E>(z.th, p.th(), c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  const T eps = chain_eps(p, rng, c);  // jitter(rng, lf) (src/integrator.jl:140-156)
  draw_momentum<T, E>(p, rng, RNG_MOMENTUM, c, d0, z.r, p.refresh_alpha);
  if (lane == 0) p.eps_cur()[c] = eps;
  fill_caches<T, G, E, TK>(z, minv, p.tp, lane, d0);
  store_point<T, E>(p, c, d0, lane, z);
}

// ------------------------------------------------------------------------------------------------
// step(lf, h, z, n_steps) fused: src/integrator.jl:216-265.  Each chain stops after its own
// first non-finite point (the reference's scalar-chain behaviour; Q1 in DESIGN.md).
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E, int TK>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_leapfrog(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.th, p.th(), c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.r, p.r(), c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.g, p.g(), c * p.D, d0, p.D, T(0));
  z.lp = p.lp()[c];
  z.lk = p.lk()[c];
  const int64_t n = p.n_steps < 0 ? -p.n_steps : p.n_steps;
  const T eps = p.n_steps > 0 ? p.eps_nom()[c] : -p.eps_nom()[c];
  bool alive = true;
  for (int64_t i = 1; i <= n; ++i) {
    if (alive) {
      leapfrog_step<T, G, E, TK>(z, minv, eps, p.tp, p.lf, lane, d0, i, n);
      alive = is_finite(z.lp) && is_finite(z.lk);
    }
  }
  store_point<T, E>(p, c, d0, lane, z);
}

// split step around an external gradient: first half (src/integrator.jl:231-236)
template <class T>
__global__ __launch_bounds__(256) void k_lf_pre(KP<T> p, int fwd, int64_t i, int64_t n) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.N * p.D) return;
  int64_t c = idx / p.D;
  int d = (int)(idx - c * p.D);
  T eps = fwd ? p.eps_nom()[c] : -p.eps_nom()[c];
  T r = p.r()[idx];
  if (p.lf.kind == 2) {
    int64_t it = 2 * (i - 1) + 1;
    r = (it <= n) ? r * p.lf.sqrt_alpha : r / p.lf.sqrt_alpha;
  }
  r = r - eps / 2 * p.g()[idx];
  T mi = p.minv ? p.minv[p.minv_per_chain ? idx : d] : T(1);
  p.th()[idx] = p.th()[idx] + eps * (mi * r);
  p.r()[idx] = r;
}

// second half (src/integrator.jl:238-243): g ← caller's -∇ℓπ, r -= ϵ/2 g, temper; k_kinetic follows
template <class T>
__global__ __launch_bounds__(256) void k_lf_post(KP<T> p, int fwd, int64_t i, int64_t n, const T* gneg) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.N * p.D) return;
  int64_t c = idx / p.D;
  T eps = fwd ? p.eps_nom()[c] : -p.eps_nom()[c];
  T g = gneg[idx];
  T r = p.r()[idx] - eps / 2 * g;
  if (p.lf.kind == 2) {
    int64_t it = 2 * (i - 1) + 2;
    r = (it <= n) ? r * p.lf.sqrt_alpha : r / p.lf.sqrt_alpha;
  }
  p.g()[idx] = g;
  p.r()[idx] = r;
}

// ------------------------------------------------------------------------------------------------
// Static HMC transition, one launch: jitter → refresh → L leapfrogs → MH → revert → flip → stats
// (src/sampler.jl:48-58, src/trajectory.jl:271-340, :855-880, :303-332).
// MultinomialTS (:369-390): pass 1 integrates backwards and forwards recording only the L+1
// energies, the categorical index is drawn exactly as randcat does (src/utilities.jl:51-59),
// pass 2 re-integrates to the selected point (bitwise the same arithmetic) — no (L+1)·D store.
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E, int TK>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_hmc(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z0, z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z0.th, p.th(), c * p.D, d0, p.D, T(0));
  if (p.refresh_alpha != T(0)) load_vec<T, E>(z0.r, p.r(), c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  const T eps = chain_eps(p, rng, c);
  draw_momentum<T, E>(p, rng, RNG_MOMENTUM, c, d0, z0.r, p.refresh_alpha);
  fill_caches<T, G, E, TK>(z0, minv, p.tp, lane, d0);
  const T H0 = -(z0.lp + z0.lk);
  const int64_t L = p.L;
  bool is_accept;
  T alpha;
  T Hprop;
  if (p.sampler == 0) {  // EndPointTS
    z = z0;
    bool alive = true;
    for (int64_t i = 1; i <= L; ++i) {
      if (alive) {
        leapfrog_step<T, G, E, TK>(z, minv, eps, p.tp, p.lf, lane, d0, i, L);
        alive = is_finite(z.lp) && is_finite(z.lk);
      }
    }
    Hprop = -(z.lp + z.lk);
    is_accept = Hprop < H0 + (T)rng.randexp(RNG_TRANSITION, 0);  // mh_accept_ratio
    alpha = jl_min(T(1), exp(H0 - Hprop));
  } else {  // MultinomialTS
    // rand_coupled(rng, 0:n_steps): ONE draw shared by all chains (Q4)
    Rng shared = rng;
    shared.chain = COUPLED_CHAIN;
    double uc = shared.uniform(RNG_TRANSITION, 0);
    int64_t n_fwd = (int64_t)floor(uc * (double)(L + 1));
    if (n_fwd > L) n_fwd = L;
    const int64_t n_bwd = L - n_fwd;
    T* Hs = p.hmc_H + c * (L + 1);  // index k: position k - n_bwd relative to the start point
    int64_t got_bwd = 0, got_fwd = 0;
    // pass 1: energies only
    z = z0;
    bool alive = true;
    for (int64_t i = 1; i <= n_bwd; ++i) {
      if (alive) {
        leapfrog_step<T, G, E, TK>(z, minv, -eps, p.tp, p.lf, lane, d0, i, n_bwd);
        got_bwd = i;
        Hs[n_bwd - i] = -(z.lp + z.lk);  // every lane of the group writes the same value and
                                         // later reads back only what it wrote itself
        alive = is_finite(z.lp) && is_finite(z.lk);
      }
    }
    z = z0;
    alive = true;
    for (int64_t i = 1; i <= n_fwd; ++i) {
      if (alive) {
        leapfrog_step<T, G, E, TK>(z, minv, eps, p.tp, p.lf, lane, d0, i, n_fwd);
        got_fwd = i;
        Hs[n_bwd + i] = -(z.lp + z.lk);
        alive = is_finite(z.lp) && is_finite(z.lk);
      }
    }
    Hs[n_bwd] = H0;
    // zs = vcat(reverse(zs_bwd)..., z, zs_fwd...): indices n_bwd-got_bwd .. n_bwd+got_fwd
    const int64_t lo = n_bwd - got_bwd, hi = n_bwd + got_fwd;
    T mx = -Lim<T>::inf();
    for (int64_t k = lo; k <= hi; ++k) mx = jl_max(mx, -Hs[k]);
    T se = 0, sa = 0;
    for (int64_t k = lo; k <= hi; ++k) {
      T Hk = Hs[k];
      se += exp(-Hk - mx);
      sa += exp(jl_min(T(0), -(Hk - H0)));
    }
    const T lse = mx + log(se);
    const T u = (T)rng.uniform(RNG_TRANSITION, 0);
    T cum = 0;
    int64_t idx = lo;
    while (cum < u && idx <= hi) {
      cum += exp(-Hs[idx] - lse);
      ++idx;
    }
    int64_t sel = idx - 1;  // position in lo..hi (max(i, 1) of randcat)
    if (sel < lo) sel = lo;
    alpha = sa / (T)(hi - lo + 1);
    is_accept = true;
    // pass 2: re-integrate to the selected point
    z = z0;
    const int64_t steps = sel >= n_bwd ? sel - n_bwd : n_bwd - sel;
    const T e2 = sel >= n_bwd ? eps : -eps;
    const int64_t ntot = sel >= n_bwd ? n_fwd : n_bwd;
    for (int64_t i = 1; i <= steps; ++i) leapfrog_step<T, G, E, TK>(z, minv, e2, p.tp, p.lf, lane, d0, i, ntot);
    Hprop = -(z.lp + z.lk);
  }
  // accept_phasepoint! + momentum flip (src/trajectory.jl:281-283)
  if (!is_accept) z = z0;
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = -z.r[e];
  store_point<T, E>(p, c, d0, lane, z);
  const T H = -(z.lp + z.lk);
  const int numerr = is_finite(Hprop) ? 0 : 1;
  if (lane == 0) {
    p.eps_cur()[c] = eps;
    p.st_nsteps()[c] = (int32_t)L;
    p.st_accept()[c] = is_accept ? 1 : 0;
    p.st_accrate()[c] = alpha;
    p.st_logdens()[c] = z.lp;
    p.st_H()[c] = H;
    p.st_Herr()[c] = H - H0;
    p.st_maxHerr()[c] = 0;
    p.st_depth()[c] = 0;
    p.st_numerr()[c] = numerr;
  }
  accumulate<T, E>(p, c, d0, lane, z.th, (int)L, numerr, H);
}

// ------------------------------------------------------------------------------------------------
// find_good_stepsize for every chain (src/trajectory.jl:753-837), incl. the Q3 quirk
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E, int TK>
__global__ __launch_bounds__(G > 256 ? G : 256) void k_find_eps(KP<T> p, T* eps_out) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z0;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z0.th, p.th(), c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  draw_momentum<T, E>(p, rng, RNG_FINDEPS, c, d0, z0.r, T(0));
  fill_caches<T, G, E, TK>(z0, minv, p.tp, lane, d0);
  const T H = -(z0.lp + z0.lk);
  LeapfrogP<T> plain;
  plain.kind = 0;
  plain.sqrt_alpha = 1;
  auto A = [&](T e) {
    Point<T, E> z = z0;
    leapfrog_step<T, G, E, TK>(z, minv, e, p.tp, plain, lane, d0, 1, 1);
    return -(z.lp + z.lk);
  };
  const T log_a_min = 2 * log(T(0.5)), log_a_cross = log(T(0.5)), log_a_max = log(T(0.75));
  T eps = p.init_eps, epsp = p.init_eps;
  T dH = H - A(eps);
  const bool too_high = dH > log_a_cross;
  for (int it = 0; it < p.max_iters; ++it) {
    epsp = too_high ? 2 * eps : eps / 2;
    dH = H - A(eps);  // Q3: evaluated at ϵ, not ϵ′ (src/trajectory.jl:799-800)
    if (too_high != (dH > log_a_cross)) break;
    eps = epsp;
  }
  T lo = jl_min(eps, epsp), hi = jl_max(eps, epsp);
  eps = lo;
  epsp = hi;
  for (int it = 0; it < p.max_iters; ++it) {
    T mid = eps / 2 + epsp / 2;
    dH = H - A(mid);
    if (dH > log_a_max) eps = mid;
    else if (dH < log_a_min) epsp = mid;
    else {
      eps = mid;
      break;
    }
  }
  if (lane == 0) eps_out[c] = eps;
}

// ------------------------------------------------------------------------------------------------
// adaptation (src/adaptation/*.jl), element-wise over chains / (D,N)
// ------------------------------------------------------------------------------------------------
// The adaptation arithmetic, shared by the stand-alone kernels below and by the in-kernel adaptation of
// k_nuts MODE 3 (ahmc_nuts.hpp).  Floating-point contraction is off inside, so both compile to the same
// operations and a batched run reproduces the per-iteration one bit for bit.
template <class T>
struct DAState {
  int32_t m;
  T eps, mu, xbar, Hbar;
};
// The two transcendental functions of the iteration count in adapt_stepsize! — √m and m^(−κ) — tabulated ON THE DEVICE by the same
// calls that da_step would make (k_da_table), for m < DA_TAB_M: every chain of every wave would otherwise evaluate an f64 `pow`
// and an f64 `sqrt` per transition on a number that is the same for all of them (≈ 250 of the ≈ 640 VALU instructions per
// transition that the warm-up instantiation of k_nuts issues beyond the sampling one, and its widest register peak).  Same bits:
// the table holds what the functions return.
constexpr int DA_TAB_M = 4096;
// NesterovDualAveraging's constants (src/adaptation/stepsize.jl:168-172) — one definition: the table is built for THIS κ
constexpr double DA_GAMMA = 0.05, DA_T0 = 10.0, DA_KAPPA = 0.75;
// Round 4: also the two DIVISIONS whose operands are the same for every chain, η_H = 1 / (m + t0) and √m / γ (an f64 division is
// a quarter-rate reciprocal and a Newton iteration, ≈ 70 issue cycles each, once per chain and transition): four entries per m.
template <class T>
__global__ void k_da_table(T* __restrict__ tab, T kappa, T gamma, T t0) {
#pragma clang fp contract(off)
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= DA_TAB_M) return;
  const T sq = sqrt((T)m);
  tab[4 * m] = sq;
  tab[4 * m + 1] = m > 0 ? pow((T)m, -kappa) : T(0);
  tab[4 * m + 2] = T(1) / ((T)m + t0);
  tab[4 * m + 3] = sq / gamma;
}
// adapt_stepsize! (src/adaptation/stepsize.jl:178-210)
template <class T>
__device__ __forceinline__ void da_step(DAState<T>& s, T alpha, T delta, T gamma, T t0, T kappa, const T* __restrict__ tab = nullptr) {
#pragma clang fp contract(off)
  const int32_t m = s.m + 1;
  const bool tabulated = tab != nullptr && m < DA_TAB_M && m >= 0;  // (the table was built for THESE γ, t0, κ: DA_GAMMA, DA_T0, DA_KAPPA)
  const T eta_H = tabulated ? tab[4 * m + 2] : T(1) / ((T)m + t0);
  const T Hbar = (T(1) - eta_H) * s.Hbar + eta_H * (delta - jl_min(T(1), alpha));
  const T sqrt_m_over_gamma = tabulated ? tab[4 * m + 3] : sqrt((T)m) / gamma;
  const T x = s.mu - Hbar * sqrt_m_over_gamma;
  const T eta_x = tabulated ? tab[4 * m + 1] : pow((T)m, -kappa);
  const T xbar = (T(1) - eta_x) * s.xbar + eta_x * x;
  const T eps = exp(x);
  if (is_finite(eps)) {  // otherwise the previous (m, ϵ, x̄, H̄) are kept (:199-203)
    s.m = m;
    s.eps = eps;
    s.xbar = xbar;
    s.Hbar = Hbar;
  }
}
template <class T>
__device__ __forceinline__ void da_reset(DAState<T>& s) {  // reset!(das) (:40-53)
  s.m = 0;
  s.mu = log(10 * s.eps);
  s.xbar = 0;
  s.Hbar = 0;
}
// push!(wv, s) (src/adaptation/massmatrix.jl:141-149), n = the count after this push
template <class T>
__device__ __forceinline__ void welford_push(T& mu, T& M, T x, T n) {
#pragma clang fp contract(off)
  const T delta = x - mu;
  mu = mu + delta / n;
  M = M + delta * delta * ((n - 1) / n);
}
// get_estimation(wv) (:152-157)
template <class T>
__device__ __forceinline__ T welford_estimate(T M, T n) {
#pragma clang fp contract(off)
  return n / ((n + 5) * (n - 1)) * M + T(1e-3) * (5 / (n + 5));
}

// In-kernel adaptation (k_nuts MODE 3): everything adapt!(h, κ, adaptor, i, n_adapts, z, α) needs for a batch of
// consecutive warm-up transitions.  Step sizes and (Diag, per-chain) mass matrices are per chain and the Stan
// window schedule depends on the iteration index only, so no chain ever needs another chain's data.
template <class T>
struct AdaptK {
  int kind, has_ss, has_mm, nutpie;
  int pooled;        // AHMC_VAR_POOLED: the kernel only pushes; update / reset are the host's (pooled_update) at the batch end
  int64_t i0;        // adaptation iterations done before this launch
  int64_t n_adapts;
  int64_t stan_i0;   // StanHMCAdaptor state.i before this launch
  int64_t window_start, window_end;
  int n_splits;
  int64_t splits[24];
  int64_t wv_n0, wv_nmin;
  T delta, gamma, t0, kappa;
  int32_t* da_m;
  T *da_eps, *da_mu, *da_xbar, *da_Hbar;
  T *wv_mu, *wv_M, *wv_var, *wg_mu, *wg_M;
  T *minv, *sqrt_minv, *eps_nom;
  const T* da_tab;   // √m, m^(−κ) for m < DA_TAB_M (k_da_table), or null
};

// stream-ordered upload of a small argument block (the previous launch may still be reading *dst)
template <class S>
__global__ void k_put(S v, S* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

template <class T>
struct AdaptP {
  int64_t N, DN;
  // dual averaging (stepsize.jl:178-210)
  int do_da, da_reset, da_finalize;
  T delta, gamma, t0, kappa;
  int32_t* da_m;
  T *da_eps, *da_mu, *da_xbar, *da_Hbar;
  const T* alpha;  // acceptance_rate of the last transition
  const T* da_tab; // √m, m^(−κ) for m < DA_TAB_M (k_da_table), or null
  T* eps_nom;      // update(κ, adaptor): nominal step size ← getϵ
  // Welford variance (massmatrix.jl:141-157)
  int do_push, do_update, wv_reset;
  T wv_n;  // count AFTER this push
  const T* th;
  T *wv_mu, *wv_M, *wv_var;
  T *minv, *sqrt_minv;  // update(h, adaptor): M⁻¹ ← var, sqrtM⁻¹ recomputed (metric.jl:61-63)
  // NutpieVar (massmatrix.jl:160-250): a second Welford estimator on z.ℓπ.gradient
  int nutpie;
  const T* gr;
  T *wg_mu, *wg_M;
};

template <class T>
__global__ __launch_bounds__(256) void k_adapt_da(AdaptP<T> a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  DAState<T> st{a.da_m[i], a.da_eps[i], a.da_mu[i], a.da_xbar[i], a.da_Hbar[i]};
  if (a.do_da) da_step(st, a.alpha[i], a.delta, a.gamma, a.t0, a.kappa, a.da_tab);
  if (a.da_reset) da_reset(st);
  if (a.da_finalize) st.eps = exp(st.xbar);  // finalize! (:55-62)
  a.da_m[i] = st.m; a.da_eps[i] = st.eps; a.da_mu[i] = st.mu; a.da_xbar[i] = st.xbar; a.da_Hbar[i] = st.Hbar;
  a.eps_nom[i] = a.da_eps[i];
}

template <class T>
__global__ __launch_bounds__(256) void k_adapt_wv(AdaptP<T> a) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.DN) return;
  T mu = a.wv_mu[k], M = a.wv_M[k];
  T mug = 0, Mg = 0;
  if (a.nutpie) { mug = a.wg_mu[k]; Mg = a.wg_M[k]; }
  if (a.do_push) {
    const T n = a.wv_n;
    welford_push(mu, M, a.th[k], n);
    if (a.nutpie) welford_push(mug, Mg, a.gr[k], n);  // push!(nv, z) (:238-243)
  }
  if (a.do_update) {  // get_estimation (:152-157), only when n >= n_min (host decides)
    const T n = a.wv_n;
    T var = welford_estimate(M, n);
    if (a.nutpie) var = sqrt(var / welford_estimate(Mg, n));  // sqrt.(est(θ) ./ est(∇)) (:246-250)
    a.wv_var[k] = var;
    a.minv[k] = var;
    a.sqrt_minv[k] = sqrt(var);
  }
  if (a.wv_reset) {
    mu = 0;
    M = 0;
    mug = 0;
    Mg = 0;
  }
  a.wv_mu[k] = mu;
  a.wv_M[k] = M;
  if (a.nutpie) { a.wg_mu[k] = mug; a.wg_M[k] = Mg; }
}

// standard normals of the momentum draws of `n_trans` consecutive transitions: element d of chain c
// at transition kt = Box–Muller half (d & 1) of Philox block (chain, iteration+kt, MOMENTUM, d >> 1)
template <class T>
__global__ __launch_bounds__(256) void k_normals(KP<T> p, T* __restrict__ out, int n_trans, uint32_t purpose) {
  const int64_t pairs_per_chain = (p.D + 1) / 2;
  const int64_t total = pairs_per_chain * p.N * n_trans;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t kt = i / (pairs_per_chain * p.N);
    const int64_t rem = i - kt * pairs_per_chain * p.N;
    const int64_t c = rem / pairs_per_chain;
    const int pair = (int)(rem - c * pairs_per_chain);
    Rng rng = make_rng(p, c);
    rng.iter = p.iteration + (uint32_t)kt;
    double a, b;
    rng.normal_pair(purpose, (uint32_t)pair, a, b);
    T* dst = out + (kt * p.N + c) * p.D + 2 * pair;
    dst[0] = (T)a;
    if (2 * pair + 1 < p.D) dst[1] = (T)b;
  }
}

// ------------------------------------------------------------------------------------------------
// Dispatch order of k_nuts (LPT: chains with the most expected work first).  A counting sort on 16-bit keys,
// entirely on the stream: key = bin index, smaller bin = earlier.  by_work = 0: ascending step size (the
// 8 exponent + 8 top mantissa bits of float(ϵ)); by_work = n > 0: descending Σ n_steps over the last n kept transitions.
// Chains inside one bin land in arbitrary order (atomics) — the order only schedules, results do not depend on it.
// ------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ unsigned order_key(const T* eps, const long long* work, int by_work, int64_t i) {
  if (by_work) {  // by_work = number of transitions the counts cover: key = mean leapfrogs per transition, in 1/16ths
    long long w = work[i] * 16 / by_work;
    return 65535u - (unsigned)(w < 0 ? 0 : (w > 65535 ? 65535 : w));
  }
  return (__float_as_uint((float)eps[i]) >> 15) & 0xFFFFu;
}
template <class T>
__global__ __launch_bounds__(256) void k_order_hist(const T* eps, const long long* work, int by_work, unsigned* hist, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) atomicAdd(&hist[order_key(eps, work, by_work, i)], 1u);
}
template <class U>
__global__ __launch_bounds__(1024) void k_order_scan(U* hist) {  // exclusive prefix sum of 65536 bins, one block
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  unsigned s = 0;
  for (int k = 0; k < 64; ++k) s += hist[t * 64 + k];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    unsigned v = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned base = t ? part[t - 1] : 0u;
  for (int k = 0; k < 64; ++k) {
    unsigned h = hist[t * 64 + k];
    hist[t * 64 + k] = base;
    base += h;
  }
}
template <class T>
__global__ __launch_bounds__(256) void k_order_scatter(const T* eps, const long long* work, int by_work, unsigned* offs, int* order, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) order[atomicAdd(&offs[order_key(eps, work, by_work, i)], 1u)] = (int)i;
}

template <class T>
__global__ __launch_bounds__(256) void k_sqrt(const T* in, T* out, int64_t n) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = sqrt(in[k]);
}

template <class T>
__global__ __launch_bounds__(256) void k_bcast_cols(const T* in, T* out, int64_t D, int64_t N) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < D * N) out[k] = in[k % D];
}

template <class T>
__global__ __launch_bounds__(256) void k_fill(T* out, T v, int64_t n) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = v;
}

}  // namespace ahmc
