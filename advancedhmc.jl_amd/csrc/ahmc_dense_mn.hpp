// ahmc_dense_mn.hpp — static HMC with MultinomialTS (src/trajectory.jl:369-390, randcat src/utilities.jl:51-59) on
// the step-synchronous engine (dense metric / dense target / external target).  Same scheme as the fused k_hmc
// (ahmc_kernels.hpp): pass 1 integrates backwards and forwards from the start point and records only the energies
// (hmc_H, (L+1) per chain; a chain stops at its own first non-finite point, which is still recorded, as
// `step(...; full_trajectory = true)` does, src/integrator.jl:248-255); the categorical index is drawn exactly as
// randcat does; pass 2 re-integrates every chain to its selected point (the same arithmetic, so the same bits).
// The number of forward steps is ONE draw shared by all chains (rand_coupled, src/trajectory.jl:373, quirk Q4).
//
// DChain fields reused: H0, eps, cand_lp / cand_lk (the start point's, set by k_d_hmc_begin), na_c = backward
// points recorded, na_tree = forward points recorded, it = leapfrogs left in pass 2, sa_tree = mean acceptance.
#pragma once

namespace ahmc {

// per-chain counters ← 0; thread 0 also publishes n_fwd = rand_coupled(rng, 0:L) and clears the pass-2 maximum
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_init(KP<T> p, DP<T> q, int* __restrict__ out /* [0] = n_fwd, [1] = max pass-2 steps */) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) {
    Rng shared = make_rng(p, 0);
    shared.chain = COUPLED_CHAIN;
    const double uc = shared.uniform(RNG_TRANSITION, 0);
    int64_t n_fwd = (int64_t)floor(uc * (double)(p.L + 1));
    if (n_fwd > p.L) n_fwd = p.L;
    out[0] = (int)n_fwd;
    out[1] = 0;
  }
  if (c >= p.N) return;
  q.S[c].na_c = 0;
  q.S[c].na_tree = 0;
}

// after leapfrog i of a pass-1 direction: record the energy of the point every still-moving chain arrived at; a chain
// whose point is non-finite stops there (es = 0)
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_rec(KP<T> p, DP<T> q, int64_t i, int dir, int64_t n_bwd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.N) return;
  if (q.es[c] == T(0)) return;
  const T lp = p.lp()[c], lk = p.lk()[c];
  T* Hs = p.hmc_H + c * (p.L + 1);  // index k: position k − n_bwd relative to the start point
  Hs[dir < 0 ? n_bwd - i : n_bwd + i] = -(lp + lk);
  if (dir < 0) q.S[c].na_c = (int32_t)i; else q.S[c].na_tree = (int32_t)i;
  if (!(isfinite(lp) && isfinite(lk))) q.es[c] = T(0);
}

// back to the start point (θ, r, -∇ℓπ, ℓπ; v, w and ℓκ are recomputed by the host exactly as they were made), every
// chain moving again with signed step `sign`·ϵ
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_restore(KP<T> p, DP<T> q, T sign) {
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
    q.es[c] = sign * q.S[c].eps;
  }
}

// randcat over the recorded energies (the arithmetic of k_hmc) → pass-2 plan of the chain
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_select(KP<T> p, DP<T> q, int64_t n_bwd, int* __restrict__ max_steps) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.N) return;
  DChain<T>& S = q.S[c];
  Rng rng = make_rng(p, c);
  T* Hs = p.hmc_H + c * (p.L + 1);
  const T H0 = S.H0;
  Hs[n_bwd] = H0;
  // zs = vcat(reverse(zs_bwd)..., z, zs_fwd...): indices n_bwd − got_bwd .. n_bwd + got_fwd
  const int64_t lo = n_bwd - S.na_c, hi = n_bwd + S.na_tree;
  T mx = -Lim<T>::inf();
  for (int64_t k = lo; k <= hi; ++k) mx = jl_max(mx, -Hs[k]);
  T se = 0, sa = 0;
  for (int64_t k = lo; k <= hi; ++k) {
    const T Hk = Hs[k];
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
  const int64_t steps = sel >= n_bwd ? sel - n_bwd : n_bwd - sel;
  S.sa_tree = sa / (T)(hi - lo + 1);
  S.it = (int32_t)steps;
  q.es[c] = steps > 0 ? (sel >= n_bwd ? S.eps : -S.eps) : T(0);
  if (steps > 0) atomicMax(max_steps, (int)steps);
}

// after a pass-2 leapfrog: one step less to go; a chain that has arrived stops
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_count(KP<T> p, DP<T> q) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.N) return;
  if (q.es[c] == T(0)) return;
  const int left = q.S[c].it - 1;
  q.S[c].it = left;
  if (left <= 0) q.es[c] = T(0);
}

// Transition(z, stats) (src/trajectory.jl:281-298): the selected point is always accepted; momentum flip; statistics
template <class T>
__global__ __launch_bounds__(256) void k_d_mn_end(KP<T> p, DP<T> q) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= p.N) return;
  const int D = p.D;
  DChain<T>& S = q.S[c];
  const T lp = p.lp()[c], lk = p.lk()[c];
  const T H = -(lp + lk);
  T* th = p.th() + c * D;
  T* r = p.r() + c * D;
  T* s1 = p.acc_sum() + c * D;
  T* s2 = p.acc_sumsq() + c * D;
  for (int d = lane; d < D; d += 64) {
    const T t = th[d];
    r[d] = -r[d];  // z = PhasePoint(z.θ, -z.r, ...) (:283)
    if (p.accum) { s1[d] += t; s2[d] += t * t; }
  }
  if (lane == 0) {
    const int numerr = isfinite(H) ? 0 : 1;
    p.st_nsteps()[c] = (int32_t)p.L;
    p.st_accept()[c] = 1;
    p.st_accrate()[c] = S.sa_tree;
    p.st_logdens()[c] = lp;
    p.st_H()[c] = H;
    p.st_Herr()[c] = H - S.H0;
    p.st_maxHerr()[c] = 0;
    p.st_depth()[c] = 0;
    p.st_numerr()[c] = numerr;
    q.es[c] = T(0);
    if (p.accum) {
      p.acc_nsteps()[c] += p.L;
      p.acc_ndiv()[c] += numerr;
      accumulate_energy(p, c, H);
    }
  }
}

// PartialMomentumRefreshment(α) (src/hamiltonian.jl:243-254): out = α·r + sqrt(1 − α²)·ξ, ξ = the fresh momentum
// draw (rand_momentum) — element-wise over `total` elements; out may alias r or ξ
template <class T>
__global__ __launch_bounds__(256) void k_d_partial(T* out, const T* r, const T* xi, T alpha, T s, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) out[idx] = alpha * r[idx] + s * xi[idx];
}

// temper(lf, r, (i, is_half), n_steps) (src/integrator.jl:198-209) for the chains that are moving: r ← r·√α while
// 2(i−1)+1+[second half] <= n_steps, r ← r/√α afterwards; v = M⁻¹r scales with it, and after the second half so does
// the kinetic energy the step has just stored (ℓκ ∝ r·v).  n_steps may differ with the direction (pass 2 of the
// multinomial sampler re-integrates backward chains through n_bwd and forward chains through n_fwd steps).
template <class T>
__global__ __launch_bounds__(256) void k_d_temper(T* __restrict__ r, T* __restrict__ v, T* __restrict__ lk, const T* __restrict__ es, T sqrt_alpha,
                                                  int64_t i, int second_half, int64_t n_pos, int64_t n_neg, int D, int64_t N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * N) return;
  const int64_t c = idx / D;
  const T e = es[c];
  if (e == T(0)) return;
  const int64_t i_temper = 2 * (i - 1) + 1 + (second_half ? 1 : 0);
  const bool up = i_temper <= (e > T(0) ? n_pos : n_neg);
  r[idx] = up ? r[idx] * sqrt_alpha : r[idx] / sqrt_alpha;
  v[idx] = up ? v[idx] * sqrt_alpha : v[idx] / sqrt_alpha;
  if (second_half && idx == c * D) {
    const T f = up ? sqrt_alpha : T(1) / sqrt_alpha;
    lk[c] = sanitize(lk[c] * f * f);
  }
}

}  // namespace ahmc
