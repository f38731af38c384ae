// ahmc_dense_mn_host.hpp — host side of static HMC with MultinomialTS on the step-synchronous engine (kernels and
// scheme: ahmc_dense_mn.hpp).  A small state machine (c->mn) advanced once per completed leapfrog, so that the same
// code serves the dense engine's own loop (dn_hmc_multinomial: the gradient is a GEMM / a built-in family kernel)
// and the ask / tell protocol (ahmc_ext_host.hpp: the gradient is the caller's).
#pragma once

// first half of a leapfrog of every chain (k_d_pre; chains with es = 0 do not move)
template <class T>
int dn_pre_all(Ctx<T>* c) {
  const bool dm = c->metric_kind == AHMC_METRIC_DENSE;
  T* V = c->dn_W + (size_t)DS_CUR_V * c->D * c->N;
  T* W = dm ? c->dn_W + (size_t)DS_CUR_W * c->D * c->N : nullptr;
  const T* minv = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
  hipLaunchKernelGGL((k_d_pre<T>), dim3(dn_grid_elems(c)), dim3(256), 0, c->stream, c->th, c->r, c->g, V, W, minv, c->minv_per_chain ? 1 : 0, c->dn_es,
                     (int)c->D, c->N, (const int*)nullptr);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}
// second half once g′ (and, unless dense_target, ℓπ) of the new positions are in place: w′ = M⁻¹g′ for the dense
// metric, then k_d_post
template <class T>
int dn_post_all(Ctx<T>* c, bool dense_target) {
  const bool dm = c->metric_kind == AHMC_METRIC_DENSE;
  T* V = c->dn_W + (size_t)DS_CUR_V * c->D * c->N;
  T* W = dm ? c->dn_W + (size_t)DS_CUR_W * c->D * c->N : nullptr;
  const T* minv = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
  if (dm) {
    int rc = dn_gemm(c, c->dn_minv, c->g, W, c->N);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((k_d_post<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, c->th, c->r, c->g, V, W, minv, c->minv_per_chain ? 1 : 0, c->dn_es,
                     c->lp, c->lk, dense_target ? 1 : 0, (int)c->D, c->N, (const int*)nullptr);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

enum { MN_NONE = 0, MN_BWD = 1, MN_FWD = 2, MN_REINT = 3, MN_DONE = 4 };

// TemperedLeapfrog: the temper call of the leapfrog the state machine is about to start (first half) or has just
// finished (second half).  Pass 1 integrates all chains n_bwd steps backwards, then n_fwd forwards; pass 2 takes a
// chain through the steps of its own direction again (step(lf, h, z, n; fwd), src/trajectory.jl:374-376).
template <class T>
int mn_temper(Ctx<T>* c, bool second_half) {
  const MnRun& m = c->mn;
  const int64_t i = second_half ? m.i : m.i + 1;  // m.i counts completed leapfrogs of the current pass
  return dn_temper(c, i, second_half, m.n_fwd, m.n_bwd);
}
// first half-step of the next leapfrog of the current pass
template <class T>
int mn_pre(Ctx<T>* c) {
  int rc = mn_temper(c, false);
  if (rc) return rc;
  return dn_pre_all(c);
}

template <class T>
KP<T> mn_kp(Ctx<T>* c) {
  KP<T> p = make_kp(c);
  p.L = c->mn.L;
  p.accum = c->mn.accum ? 1 : 0;
  return p;
}

template <class T>
int mn_end(Ctx<T>* c) {
  KP<T> p = mn_kp(c);
  DP<T> q = make_dp(c);
  hipLaunchKernelGGL((k_d_mn_end<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  HIPCHK(hipGetLastError());
  c->mn.phase = MN_DONE;
  return AHMC_OK;
}

// every chain back at the start point, moving with sign·ϵ: θ, r, -∇ℓπ, ℓπ from the START slots; v = M⁻¹r, ℓκ and
// w = M⁻¹g by the same launches that made them at the start of the transition (same bits)
template <class T>
int mn_restore(Ctx<T>* c, T sign) {
  KP<T> p = mn_kp(c);
  DP<T> q = make_dp(c);
  hipLaunchKernelGGL((k_d_mn_restore<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q, sign);
  HIPCHK(hipGetLastError());
  int rc = dn_velocity(c);
  if (rc) return rc;
  return dn_prepare_w(c);
}

// pass 1 is complete: draw every chain's point, plan pass 2
template <class T>
int mn_select(Ctx<T>* c) {
  MnRun& m = c->mn;
  if (m.L > 0) {
    int rc = mn_restore(c, T(1));
    if (rc) return rc;
  }
  KP<T> p = mn_kp(c);
  DP<T> q = make_dp(c);
  hipLaunchKernelGGL((k_d_mn_select<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, p, q, m.n_bwd, c->dn_active + 1);
  HIPCHK(hipGetLastError());
  int left = 0;
  HIPCHK(hipMemcpyAsync(&left, c->dn_active + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  m.phase = MN_REINT;
  m.left = left;
  m.i = 0;
  if (left > 0) return mn_pre(c);
  return mn_end(c);
}

// Called after k_d_hmc_begin (jitter, start point saved, H0, es = +ϵ).  Returns with the first half of the first
// leapfrog done (c->mn.phase in BWD / FWD / REINT) or with the transition complete (MN_DONE).
template <class T>
int mn_begin(Ctx<T>* c, int64_t L, bool accum) {
  MnRun& m = c->mn;
  m = MnRun();
  m.L = L;
  m.accum = accum;
  const size_t need = (size_t)(L + 1) * (size_t)c->N;
  if (need > c->hmc_H_elems) {
    if (c->hmc_H) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->hmc_H)); }
    c->hmc_H = nullptr;
    c->hmc_H_elems = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->hmc_H), need * sizeof(T)));
    c->hmc_H_elems = need;
  }
  KP<T> p = mn_kp(c);
  DP<T> q = make_dp(c);
  hipLaunchKernelGGL((k_d_mn_init<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, p, q, c->dn_active);
  HIPCHK(hipGetLastError());
  int h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, c->dn_active, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  m.n_fwd = h[0];
  m.n_bwd = L - m.n_fwd;
  m.i = 0;
  if (m.n_bwd > 0) {
    m.phase = MN_BWD;
    hipLaunchKernelGGL((k_d_set<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->dn_es, c->eps_cur, T(-1), c->N);
    HIPCHK(hipGetLastError());
    return mn_pre(c);
  }
  if (m.n_fwd > 0) {
    m.phase = MN_FWD;  // (es = +ϵ from k_d_hmc_begin)
    return mn_pre(c);
  }
  return mn_select(c);
}

// the second half of a leapfrog has just completed (k_d_post): account for it and start the next one, if any
template <class T>
int mn_after_step(Ctx<T>* c) {
  MnRun& m = c->mn;
  KP<T> p = mn_kp(c);
  DP<T> q = make_dp(c);
  const dim3 gridN((unsigned)((c->N + 255) / 256));
  if (m.phase == MN_BWD || m.phase == MN_FWD) {
    const bool bwd = m.phase == MN_BWD;
    m.i += 1;
    {
      int rc = mn_temper(c, true);
      if (rc) return rc;
    }
    hipLaunchKernelGGL((k_d_mn_rec<T>), gridN, dim3(256), 0, c->stream, p, q, m.i, bwd ? -1 : 1, m.n_bwd);
    HIPCHK(hipGetLastError());
    if (m.i < (bwd ? m.n_bwd : m.n_fwd)) return mn_pre(c);
    if (bwd && m.n_fwd > 0) {
      int rc = mn_restore(c, T(1));
      if (rc) return rc;
      m.phase = MN_FWD;
      m.i = 0;
      return mn_pre(c);
    }
    return mn_select(c);
  }
  if (m.phase == MN_REINT) {
    m.i += 1;
    {
      int rc = mn_temper(c, true);
      if (rc) return rc;
    }
    hipLaunchKernelGGL((k_d_mn_count<T>), gridN, dim3(256), 0, c->stream, p, q);
    HIPCHK(hipGetLastError());
    m.left -= 1;
    if (m.left > 0) return mn_pre(c);
    return mn_end(c);
  }
  return fail(c, AHMC_ERR_STATE, "multinomial HMC: no leapfrog in flight");
}

// the dense engine's own loop: gradient by GEMM (dense target) or by the built-in family's kernel
template <class T>
int dn_hmc_multinomial(Ctx<T>* c, int64_t L, bool accum) {
  int rc = mn_begin(c, L, accum);
  if (rc) return rc;
  const bool dt = c->target_kind == AHMC_TARGET_DENSE_GAUSS;
  while (c->mn.phase == MN_BWD || c->mn.phase == MN_FWD || c->mn.phase == MN_REINT) {
    rc = dt ? dn_gemm(c, c->tparams, c->th, c->g, c->N) : dn_other_target(c, nullptr, c->N);  // g′ (and ℓπ for the built-in families)
    if (rc) return rc;
    rc = dn_post_all(c, dt);
    if (rc) return rc;
    rc = mn_after_step(c);
    if (rc) return rc;
  }
  return AHMC_OK;
}
