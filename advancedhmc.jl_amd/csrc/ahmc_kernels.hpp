// ahmc_kernels.hpp — the chain-batched kernels (gfx950).  One group of G lanes per chain, all
// per-chain state in registers, one launch per whole transition.
#pragma once

#include "ahmc_device.hpp"

namespace ahmc {

// Everything a kernel needs, passed by value.  Arrays are (D,N) column-major / (N,).
template <class T>
struct KP {
  int D;
  int64_t N;
  // phase point
  T *th, *r, *g, *lp, *lk;
  // metric (Unit: minv == nullptr)
  const T* minv;
  const T* sqrt_minv;
  int minv_per_chain;
  // integrator
  const T* eps_nom;
  T* eps_cur;
  LeapfrogP<T> lf;
  T jitter;
  TargetP<T> tp;
  // rng
  uint32_t k0, k1, chain_offset, chain_stride, iteration;
  // stats of the last transition
  int32_t *st_nsteps, *st_accept, *st_depth, *st_numerr;
  T *st_accrate, *st_logdens, *st_H, *st_Herr, *st_maxHerr;
  // accumulators over kept transitions
  int accum;
  long long *acc_nsteps, *acc_ndiv;
  T *acc_sum, *acc_sumsq;
  T refresh_alpha;
  // NUTS
  int max_depth;
  T delta_max;
  int criterion, sampler;
  T* scratch;           // pending-subtree vectors
  unsigned int* queue;  // work queue head
  unsigned int n_chunks;
  // static HMC
  int64_t L;
  T* hmc_H;  // (L+1, N) energies for MultinomialTS
  // misc
  int64_t n_steps;  // k_leapfrog
  T init_eps;
  int max_iters;
};

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
  T e0 = p.eps_nom[c];
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

template <class T, int E>
__device__ __forceinline__ void store_point(const KP<T>& p, int64_t c, int d0, int lane, const Point<T, E>& z) {
  store_vec<T, E>(z.th, p.th, c * p.D, d0, p.D);
  store_vec<T, E>(z.r, p.r, c * p.D, d0, p.D);
  store_vec<T, E>(z.g, p.g, c * p.D, d0, p.D);
  if (lane == 0) {
    p.lp[c] = z.lp;
    p.lk[c] = z.lk;
  }
}

template <class T, int E>
__device__ __forceinline__ void accumulate(const KP<T>& p, int64_t c, int d0, int lane, const T (&th)[E], int n_steps,
                                           int numerr) {
  if (!p.accum) return;
  T s1[E], s2[E];
  load_vec<T, E>(s1, p.acc_sum, c * p.D, d0, p.D, T(0));
  load_vec<T, E>(s2, p.acc_sumsq, c * p.D, d0, p.D, T(0));
#pragma unroll
  for (int e = 0; e < E; ++e) {
    s1[e] += th[e];
    s2[e] += th[e] * th[e];
  }
  store_vec<T, E>(s1, p.acc_sum, c * p.D, d0, p.D);
  store_vec<T, E>(s2, p.acc_sumsq, c * p.D, d0, p.D);
  if (lane == 0) {
    p.acc_nsteps[c] += n_steps;
    p.acc_ndiv[c] += numerr;
  }
}

#define AHMC_GEOMETRY()                                   \
  const int lane64 = threadIdx.x & 63;                    \
  const int lane = lane64 & (G - 1);                      \
  const int d0 = lane * E;                                \
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; \
  const bool active = c < p.N;                            \
  (void)lane64

// ------------------------------------------------------------------------------------------------
// phasepoint(h, θ, r) for arrays already in the context (ahmc_set_position)
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_fill_caches(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.th, p.th, c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.r, p.r, c * p.D, d0, p.D, T(0));
  fill_caches<T, G, E>(z, minv, p.tp, lane, d0);
  store_vec<T, E>(z.g, p.g, c * p.D, d0, p.D);
  if (lane == 0) {
    p.lp[c] = z.lp;
    p.lk[c] = z.lk;
  }
}

// kinetic cache only (ahmc_set_phasepoint with caller-supplied ℓπ, -∇ℓπ; ahmc_lf_post)
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_kinetic(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  T minv[E], r[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(r, p.r, c * p.D, d0, p.D, T(0));
  T s = group_sum1<G>(kinetic_partial(r, minv));
  if (lane == 0) {
    p.lk[c] = sanitize(-s / 2);
    p.lp[c] = sanitize(p.lp[c]);
  }
}

// refresh(rng, refreshment, h, z)
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_refresh(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.th, p.th, c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.r, p.r, c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  draw_momentum<T, E>(p, rng, RNG_MOMENTUM, c, d0, z.r, p.refresh_alpha);
  fill_caches<T, G, E>(z, minv, p.tp, lane, d0);
  store_point<T, E>(p, c, d0, lane, z);
}

// ------------------------------------------------------------------------------------------------
// step(lf, h, z, n_steps) fused: src/integrator.jl:216-265.  Each chain stops after its own
// first non-finite point (the reference's scalar-chain behaviour; Q1 in DESIGN.md).
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_leapfrog(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z.th, p.th, c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.r, p.r, c * p.D, d0, p.D, T(0));
  load_vec<T, E>(z.g, p.g, c * p.D, d0, p.D, T(0));
  z.lp = p.lp[c];
  z.lk = p.lk[c];
  const int64_t n = p.n_steps < 0 ? -p.n_steps : p.n_steps;
  const T eps = p.n_steps > 0 ? p.eps_nom[c] : -p.eps_nom[c];
  bool alive = true;
  for (int64_t i = 1; i <= n; ++i) {
    if (alive) {
      leapfrog_step<T, G, E>(z, minv, eps, p.tp, p.lf, lane, d0, i, n);
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
  T eps = fwd ? p.eps_nom[c] : -p.eps_nom[c];
  T r = p.r[idx];
  if (p.lf.kind == 2) {
    int64_t it = 2 * (i - 1) + 1;
    r = (it <= n) ? r * p.lf.sqrt_alpha : r / p.lf.sqrt_alpha;
  }
  r = r - eps / 2 * p.g[idx];
  T mi = p.minv ? p.minv[p.minv_per_chain ? idx : d] : T(1);
  p.th[idx] = p.th[idx] + eps * (mi * r);
  p.r[idx] = r;
}

// second half (src/integrator.jl:238-243): g ← caller's -∇ℓπ, r -= ϵ/2 g, temper; k_kinetic follows
template <class T>
__global__ __launch_bounds__(256) void k_lf_post(KP<T> p, int fwd, int64_t i, int64_t n, const T* gneg) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.N * p.D) return;
  int64_t c = idx / p.D;
  T eps = fwd ? p.eps_nom[c] : -p.eps_nom[c];
  T g = gneg[idx];
  T r = p.r[idx] - eps / 2 * g;
  if (p.lf.kind == 2) {
    int64_t it = 2 * (i - 1) + 2;
    r = (it <= n) ? r * p.lf.sqrt_alpha : r / p.lf.sqrt_alpha;
  }
  p.g[idx] = g;
  p.r[idx] = r;
}

// ------------------------------------------------------------------------------------------------
// Static HMC transition, one launch: jitter → refresh → L leapfrogs → MH → revert → flip → stats
// (src/sampler.jl:48-58, src/trajectory.jl:271-340, :855-880, :303-332).
// MultinomialTS (:369-390): pass 1 integrates backwards and forwards recording only the L+1
// energies, the categorical index is drawn exactly as randcat does (src/utilities.jl:51-59),
// pass 2 re-integrates to the selected point (bitwise the same arithmetic) — no (L+1)·D store.
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_hmc(KP<T> p) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z0, z;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z0.th, p.th, c * p.D, d0, p.D, T(0));
  if (p.refresh_alpha != T(0)) load_vec<T, E>(z0.r, p.r, c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  const T eps = chain_eps(p, rng, c);
  draw_momentum<T, E>(p, rng, RNG_MOMENTUM, c, d0, z0.r, p.refresh_alpha);
  fill_caches<T, G, E>(z0, minv, p.tp, lane, d0);
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
        leapfrog_step<T, G, E>(z, minv, eps, p.tp, p.lf, lane, d0, i, L);
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
        leapfrog_step<T, G, E>(z, minv, -eps, p.tp, p.lf, lane, d0, i, n_bwd);
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
        leapfrog_step<T, G, E>(z, minv, eps, p.tp, p.lf, lane, d0, i, n_fwd);
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
    for (int64_t i = 1; i <= steps; ++i) leapfrog_step<T, G, E>(z, minv, e2, p.tp, p.lf, lane, d0, i, ntot);
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
    p.eps_cur[c] = eps;
    p.st_nsteps[c] = (int32_t)L;
    p.st_accept[c] = is_accept ? 1 : 0;
    p.st_accrate[c] = alpha;
    p.st_logdens[c] = z.lp;
    p.st_H[c] = H;
    p.st_Herr[c] = H - H0;
    p.st_maxHerr[c] = 0;
    p.st_depth[c] = 0;
    p.st_numerr[c] = numerr;
  }
  accumulate<T, E>(p, c, d0, lane, z.th, (int)L, numerr);
}

// ------------------------------------------------------------------------------------------------
// NUTS transition (src/trajectory.jl:626-742), iterative form of build_tree:
// leaves are visited in integration order; after leaf i one merge is done per trailing zero bit
// of i, lowest level first — the same merges, in the same order, with the same RNG draws as the
// recursion.  A chain = one group of G lanes; a wave holds 64/G chains and pulls its next chunk
// of chains from a global work queue (persistent waves, so the pending-subtree scratch is indexed
// by wave slot and stays cache-resident).
//   pending subtree of level ℓ: vectors {A = ρ (generalised) or θ of its first-built leaf
//   (classic), RF = r of its first-built leaf, CT/CR = θ/r of its candidate} in global scratch,
//   scalars {w = ℓw or n, Σα, nα, ΔH_max, candidate ℓπ, ℓκ} in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int NUTS_NV = 4;  // A, RF, CT, CR
constexpr int NUTS_NS = 5;  // T-typed scalars per level: w, sa, dh, clp, clk  (+ int na)

template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_nuts(KP<T> p) {
  constexpr int CPW = Geo<T, G, E>::CPW;
  constexpr int DP = Geo<T, G, E>::DP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wib = threadIdx.x >> 6;
  const int lane64 = threadIdx.x & 63;
  const int lane = lane64 & (G - 1);
  const int gi = lane64 / G;
  const int d0 = lane * E;
  const int NLEV = p.max_depth > 1 ? p.max_depth - 1 : 1;
  // LDS: per wave [NUTS_NS][NLEV][CPW] of T, then [NLEV][CPW] of int
  T* sT = reinterpret_cast<T*>(smem) + (size_t)wib * NUTS_NS * NLEV * CPW;
  int* sI = reinterpret_cast<int*>(reinterpret_cast<T*>(smem) + (size_t)(blockDim.x >> 6) * NUTS_NS * NLEV * CPW) +
            (size_t)wib * NLEV * CPW;
#define S_T(which, lvl) sT[((which) * NLEV + (lvl)) * CPW + gi]
#define S_NA(lvl) sI[(lvl) * CPW + gi]
  const int64_t wave_slot = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
  T* scr = p.scratch + ((wave_slot * CPW + gi) * (int64_t)NLEV) * NUTS_NV * DP + d0;
#define SCR(lvl, v) (scr + ((int64_t)(lvl) * NUTS_NV + (v)) * DP)
  const bool classic = p.criterion == 0;
  const bool slice = p.sampler == 2;

  for (;;) {
    unsigned int chunk = 0;
    if (lane64 == 0) chunk = atomicAdd(p.queue, 1u);
    chunk = (unsigned int)__builtin_amdgcn_readfirstlane((int)chunk);
    if (chunk >= p.n_chunks) break;
    const int64_t c = (int64_t)chunk * CPW + gi;
    const bool active = c < p.N;
    const int64_t cc = active ? c : 0;  // inactive groups shadow chain 0 without writing

    Point<T, E> cur, oth;
    T minv[E];
    load_minv<T, E>(p, cc, d0, minv);
    load_vec<T, E>(cur.th, p.th, cc * p.D, d0, p.D, T(0));
    if (p.refresh_alpha != T(0)) load_vec<T, E>(cur.r, p.r, cc * p.D, d0, p.D, T(0));
    Rng rng = make_rng(p, cc);
    const T eps = chain_eps(p, rng, cc);
    draw_momentum<T, E>(p, rng, RNG_MOMENTUM, cc, d0, cur.r, p.refresh_alpha);
    fill_caches<T, G, E>(cur, minv, p.tp, lane, d0);
    oth = cur;
    const T H0 = -(cur.lp + cur.lk);
    uint32_t draw = 0;

    // whole-tree state (BinaryTree + sampler + candidate, src/trajectory.jl:682-689)
    T A_tree[E], ct_tree[E], cr_tree[E];
    copy_vec(A_tree, cur.r);  // ρ = r0 (TurnStatistic, :461-463); unused for classic
    copy_vec(ct_tree, cur.th);
    copy_vec(cr_tree, cur.r);
    T clp_tree = cur.lp, clk_tree = cur.lk;
    T w_tree, sa_tree = 0, dh_tree = 0, lu = 0;
    int na_tree = 0;
    if (slice) {
      lu = -H0 - (T)rng.randexp(RNG_TRANSITION, draw++);  // SliceTS(rng, z0) (:144-145)
      w_tree = 1;
    } else {
      w_tree = 0;  // MultinomialTS(rng, z0): ℓw = 0 (:155)
    }
    bool cur_is_left = false;  // which edge `cur` currently holds
    bool numerical = false;
    int j = 0;
    bool done = !active;

    // subtree-in-progress
    T A_c[E], RF_c[E], ct_c[E], cr_c[E];
    T w_c = 0, sa_c = 0, dh_c = 0, clp_c = 0, clk_c = 0;
    int na_c = 0;
    uint32_t leaf = 0, nleaf = 0;
    int v = 1;

    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
      if (!done) {
        if (leaf == 0) {  // start doubling j (:691-707)
          const bool vleft = rng.boolean(RNG_TRANSITION, draw++);
          v = vleft ? -1 : 1;
          if (vleft != cur_is_left) {  // make `cur` the edge that is extended
            Point<T, E> t = cur;
            cur = oth;
            oth = t;
            cur_is_left = vleft;
          }
          nleaf = 1u << j;
        }
        // ---- leaf: one leapfrog step in direction v (:638-647) ----
        leapfrog_step<T, G, E>(cur, minv, v > 0 ? eps : -eps, p.tp, p.lf, lane, d0, 1, 1);
        ++leaf;
        const T ne = cur.lp + cur.lk;  // neg_energy(z′)
        const T Hp = -ne;
        const T dH = Hp - H0;
        sa_c = exp(jl_min(T(0), -dH));
        na_c = 1;
        dh_c = dH;
        clp_c = cur.lp;
        clk_c = cur.lk;
        bool sub_term;
        if (slice) {
          w_c = (lu <= ne) ? T(1) : T(0);
          sub_term = !(lu < p.delta_max + ne);  // Termination(::SliceTS, ...) (:500-502)
        } else {
          w_c = H0 + ne;
          sub_term = !(-H0 < p.delta_max + ne);  // Termination(::MultinomialTS, ...) (:503-507)
        }
        numerical = numerical || sub_term;
        if (classic) copy_vec(A_c, cur.th); else copy_vec(A_c, cur.r);
        copy_vec(RF_c, cur.r);
        copy_vec(ct_c, cur.th);
        copy_vec(cr_c, cur.r);
        // ---- merges: one per trailing zero bit of `leaf` (:649-673) ----
        int lvl = 0;
        if (!sub_term) {
          for (; ((leaf >> lvl) & 1u) == 0u; ++lvl) {
            T A_p[E], RF_p[E];
            load_vec<T, E>(A_p, SCR(lvl, 0), 0, 0, E, T(0));
            load_vec<T, E>(RF_p, SCR(lvl, 1), 0, 0, E, T(0));
            const T w_p = S_T(0, lvl);
            // combine(rng, sampler′, sampler′′): `first` = the half built first (:178-195)
            bool keep_first;
            T w_new;
            if (slice) {
              w_new = w_p + w_c;
              keep_first = w_new * (T)rng.uniform(RNG_TRANSITION, draw++) < w_p;
            } else {
              w_new = logaddexp(w_p, w_c);
              keep_first = w_new < w_p + (T)rng.randexp(RNG_TRANSITION, draw++);
            }
            if (keep_first) {
              load_vec<T, E>(ct_c, SCR(lvl, 2), 0, 0, E, T(0));
              load_vec<T, E>(cr_c, SCR(lvl, 3), 0, 0, E, T(0));
              clp_c = S_T(3, lvl);
              clk_c = S_T(4, lvl);
            }
            w_c = w_new;
            // combine(treeleft, treeright) (:533-542); position order matters for maxabs only
            sa_c = S_T(1, lvl) + sa_c;
            na_c = S_NA(lvl) + na_c;
            const T dh_p = S_T(2, lvl);
            dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
            // isterminated(tc, h, tree′) on the merged subtree (:551-570)
            T dots[2] = {0, 0};
            if (classic) {
              // ends: first-built leaf (A_p, RF_p) and the current leaf; Δθ = θ_right − θ_left
#pragma unroll
              for (int e = 0; e < E; ++e) {
                T thl = v > 0 ? A_p[e] : cur.th[e], thr = v > 0 ? cur.th[e] : A_p[e];
                T rl = v > 0 ? RF_p[e] : cur.r[e], rr = v > 0 ? cur.r[e] : RF_p[e];
                T dth = thr - thl;
                dots[0] += dth * (minv[e] * (-rl));
                dots[1] += (-dth) * (minv[e] * rr);
                A_c[e] = A_p[e];
              }
              group_allsum<G>(dots);
              sub_term = (dots[0] >= 0) || (dots[1] >= 0);
            } else {
#pragma unroll
              for (int e = 0; e < E; ++e) {
                A_c[e] = A_p[e] + A_c[e];  // ρ = ρ_left + ρ_right
                dots[0] += A_c[e] * (minv[e] * RF_p[e]);
                dots[1] += A_c[e] * (minv[e] * cur.r[e]);
              }
              group_allsum<G>(dots);
              sub_term = (dots[0] <= 0) || (dots[1] <= 0);  // generalised_uturn_criterion (:619-621)
            }
            copy_vec(RF_c, RF_p);
            if (sub_term) {
              ++lvl;
              break;
            }
          }
        }
        if (sub_term) {
          // enclosing unfinished subtrees still absorb the statistics of their first halves
          // (tree′ = combine(treeleft, treeright) at every level that is a second half, :666)
          const uint32_t pend = (leaf - 1u) >> lvl << lvl;
          for (int q = lvl; (pend >> q) != 0u; ++q) {
            if ((pend >> q) & 1u) {
              sa_c = S_T(1, q) + sa_c;
              na_c = S_NA(q) + na_c;
              const T dh_p = S_T(2, q);
              dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
            }
          }
        } else if (leaf < nleaf) {
          // park the finished level-`lvl` subtree until its sibling is built
          store_vec<T, E>(A_c, SCR(lvl, 0), 0, 0, E);
          store_vec<T, E>(RF_c, SCR(lvl, 1), 0, 0, E);
          store_vec<T, E>(ct_c, SCR(lvl, 2), 0, 0, E);
          store_vec<T, E>(cr_c, SCR(lvl, 3), 0, 0, E);
          S_T(0, lvl) = w_c;
          S_T(1, lvl) = sa_c;
          S_T(2, lvl) = dh_c;
          S_T(3, lvl) = clp_c;
          S_T(4, lvl) = clk_c;
          S_NA(lvl) = na_c;
        }
        if (sub_term || leaf == nleaf) {
          // ---- top level of the doubling loop (:708-722) ----
          if (!sub_term) {
            ++j;
            bool acc;  // mh_accept(rng, sampler, sampler′): biased progressive sampling (:202-206)
            if (slice) acc = w_tree * (T)rng.uniform(RNG_TRANSITION, draw++) < w_c;
            else acc = w_tree < w_c + (T)rng.randexp(RNG_TRANSITION, draw++);
            if (acc) {
              copy_vec(ct_tree, ct_c);
              copy_vec(cr_tree, cr_c);
              clp_tree = clp_c;
              clk_tree = clk_c;
            }
          }
          sa_tree = sa_tree + sa_c;
          na_tree = na_tree + na_c;
          dh_tree = v < 0 ? maxabs(dh_c, dh_tree) : maxabs(dh_tree, dh_c);
          w_tree = slice ? w_tree + w_c : logaddexp(w_tree, w_c);
          // isterminated(tc, h, tree) on the whole tree; its edges are `cur` and `oth`
          T dots[2] = {0, 0};
          bool turn;
          if (classic) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              T thl = cur_is_left ? cur.th[e] : oth.th[e], thr = cur_is_left ? oth.th[e] : cur.th[e];
              T rl = cur_is_left ? cur.r[e] : oth.r[e], rr = cur_is_left ? oth.r[e] : cur.r[e];
              T dth = thr - thl;
              dots[0] += dth * (minv[e] * (-rl));
              dots[1] += (-dth) * (minv[e] * rr);
            }
            group_allsum<G>(dots);
            turn = (dots[0] >= 0) || (dots[1] >= 0);
          } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              A_tree[e] = A_tree[e] + A_c[e];
              dots[0] += A_tree[e] * (minv[e] * cur.r[e]);
              dots[1] += A_tree[e] * (minv[e] * oth.r[e]);
            }
            group_allsum<G>(dots);
            turn = (dots[0] <= 0) || (dots[1] <= 0);
          }
          leaf = 0;
          if (sub_term || turn || j >= p.max_depth) done = true;
        }
      }
    }

    // ---- Transition(zcand, stats) (:725-741) ----
    if (active) {
      Point<T, E> zc;
      copy_vec(zc.th, ct_tree);
      copy_vec(zc.r, cr_tree);
      zc.lp = clp_tree;
      zc.lk = clk_tree;
      (void)target_eval<T, G, E>(p.tp, zc.th, zc.g, lane, d0);  // cached -∇ℓπ of the candidate
      store_point<T, E>(p, c, d0, lane, zc);
      const T H = -(clp_tree + clk_tree);
      if (lane == 0) {
        p.eps_cur[c] = eps;
        p.st_nsteps[c] = na_tree;
        p.st_accept[c] = 1;
        p.st_accrate[c] = sa_tree / (T)na_tree;
        p.st_logdens[c] = clp_tree;
        p.st_H[c] = H;
        p.st_Herr[c] = H - H0;
        p.st_maxHerr[c] = dh_tree;
        p.st_depth[c] = j;
        p.st_numerr[c] = numerical ? 1 : 0;
      }
      accumulate<T, E>(p, c, d0, lane, zc.th, na_tree, numerical ? 1 : 0);
    }
  }
#undef S_T
#undef S_NA
#undef SCR
}

// ------------------------------------------------------------------------------------------------
// find_good_stepsize for every chain (src/trajectory.jl:753-837), incl. the Q3 quirk
// ------------------------------------------------------------------------------------------------
template <class T, int G, int E>
__global__ __launch_bounds__(256) void k_find_eps(KP<T> p, T* eps_out) {
  AHMC_GEOMETRY();
  if (!active) return;
  Point<T, E> z0;
  T minv[E];
  load_minv<T, E>(p, c, d0, minv);
  load_vec<T, E>(z0.th, p.th, c * p.D, d0, p.D, T(0));
  Rng rng = make_rng(p, c);
  draw_momentum<T, E>(p, rng, RNG_FINDEPS, c, d0, z0.r, T(0));
  fill_caches<T, G, E>(z0, minv, p.tp, lane, d0);
  const T H = -(z0.lp + z0.lk);
  LeapfrogP<T> plain;
  plain.kind = 0;
  plain.sqrt_alpha = 1;
  auto A = [&](T e) {
    Point<T, E> z = z0;
    leapfrog_step<T, G, E>(z, minv, e, p.tp, plain, lane, d0, 1, 1);
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
template <class T>
struct AdaptP {
  int64_t N, DN;
  // dual averaging (stepsize.jl:178-210)
  int do_da, da_reset, da_finalize;
  T delta, gamma, t0, kappa;
  int32_t* da_m;
  T *da_eps, *da_mu, *da_xbar, *da_Hbar;
  const T* alpha;  // acceptance_rate of the last transition
  T* eps_nom;      // update(κ, adaptor): nominal step size ← getϵ
  // Welford variance (massmatrix.jl:141-157)
  int do_push, do_update, wv_reset;
  T wv_n;  // count AFTER this push
  const T* th;
  T *wv_mu, *wv_M, *wv_var;
  T *minv, *sqrt_minv;  // update(h, adaptor): M⁻¹ ← var, sqrtM⁻¹ recomputed (metric.jl:61-63)
};

template <class T>
__global__ __launch_bounds__(256) void k_adapt_da(AdaptP<T> a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  if (a.do_da) {
    int32_t m = a.da_m[i] + 1;
    T eta_H = T(1) / ((T)m + a.t0);
    T Hbar = (T(1) - eta_H) * a.da_Hbar[i] + eta_H * (a.delta - jl_min(T(1), a.alpha[i]));
    T x = a.da_mu[i] - Hbar * (sqrt((T)m) / a.gamma);
    T eta_x = pow((T)m, -a.kappa);
    T xbar = (T(1) - eta_x) * a.da_xbar[i] + eta_x * x;
    T eps = exp(x);
    if (is_finite(eps)) {  // otherwise the previous (m, ϵ, x̄, H̄) are kept (:199-203)
      a.da_m[i] = m;
      a.da_eps[i] = eps;
      a.da_xbar[i] = xbar;
      a.da_Hbar[i] = Hbar;
    }
  }
  if (a.da_reset) {  // reset!(das) (:40-53)
    a.da_m[i] = 0;
    a.da_mu[i] = log(10 * a.da_eps[i]);
    a.da_xbar[i] = 0;
    a.da_Hbar[i] = 0;
  }
  if (a.da_finalize) a.da_eps[i] = exp(a.da_xbar[i]);  // finalize! (:55-62)
  a.eps_nom[i] = a.da_eps[i];
}

template <class T>
__global__ __launch_bounds__(256) void k_adapt_wv(AdaptP<T> a) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.DN) return;
  T mu = a.wv_mu[k], M = a.wv_M[k];
  if (a.do_push) {
    const T n = a.wv_n;
    T delta = a.th[k] - mu;
    mu = mu + delta / n;
    M = M + delta * delta * ((n - 1) / n);
  }
  if (a.do_update) {  // get_estimation (:152-157), only when n >= n_min (host decides)
    const T n = a.wv_n;
    T var = n / ((n + 5) * (n - 1)) * M + T(1e-3) * (5 / (n + 5));
    a.wv_var[k] = var;
    a.minv[k] = var;
    a.sqrt_minv[k] = sqrt(var);
  }
  if (a.wv_reset) {
    mu = 0;
    M = 0;
  }
  a.wv_mu[k] = mu;
  a.wv_M[k] = M;
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
