// ahmc_multi_host.hpp — what comes AFTER the trajectory path on the device:
//   * the final gather of a sharded run over RCCL / xGMI (ahmc_comm_*, ahmc_gather_moments, ahmc_gather_state),
//   * checkpoint / resume of the adaptor (ahmc_get_adaptor_state / ahmc_set_adaptor_state),
//   * the variance estimator pooled over chains and GPUs (AHMC_VAR_POOLED),
//   * EBFMI and ESS reductions on the device (ahmc_ebfmi, ahmc_ess).
// Included by ahmc_api.hip after the engine's own host code.
#pragma once

// (<dlfcn.h> and <rccl/rccl.h> are included at the top of ahmc_api.hip, outside its anonymous namespace)

// ---- RCCL, resolved at run time from the copy already in the process -----------------------------------------
// The communicator a host hands to ahmc_set_comm was made by SOME librccl (Julia's binding, torch's bundled copy,
// /opt/rocm's): its entry points must come from that same copy, so nothing is linked — the symbols are looked up in
// the library already mapped (RTLD_NOLOAD), and only a process without any RCCL loads /opt/rocm's.
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string where, err;
  bool ok = false;
};

inline RcclApi& rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    const char* env = getenv("AHMC_RCCL_LIB");
    if (env) { h = dlopen(env, RTLD_NOW | RTLD_GLOBAL); a.where = env; }
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (h) a.where = std::string(name) + " (already loaded)";
    }
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) a.where = name;
    }
    if (!h) { a.err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return a; }
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) a.err += std::string(" missing ") + n; return p; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.err.empty();
    return a;
  }();
  return api;
}

#define NCCLCHK(expr)                                                                                             \
  do {                                                                                                            \
    ncclResult_t _r = (expr);                                                                                     \
    if (_r != ncclSuccess) {                                                                                      \
      c->err = std::string(#expr) + ": " + (rccl_api().GetErrorString ? rccl_api().GetErrorString(_r) : "RCCL error"); \
      return AHMC_ERR_RUNTIME;                                                                                    \
    }                                                                                                             \
  } while (0)

template <class T>
int comm_release(Ctx<T>* c) {
  if (c->comm && c->comm_owned) {
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(c->comm));
  }
  c->comm = nullptr;
  c->comm_owned = false;
  c->comm_ranks = 1;
  c->comm_rank = 0;
  return AHMC_OK;
}

// device scratch of `n` doubles for the reductions below
template <class T>
int red_buf(Ctx<T>* c, size_t n) {
  if (n <= c->red_elems) return AHMC_OK;
  if (c->red) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->red)); c->red = nullptr; c->red_elems = 0; }
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->red), n * sizeof(double)));
  c->red_elems = n;
  return AHMC_OK;
}

// What the communicator itself says about the world (ahmc_comm_info): Σ 1 and Σ N over the ranks (one all-reduce, sum) and the
// largest / smallest N (one all-reduce of (N, −N), max).  Run when a communicator is attached: it proves every rank joined, and
// tells ahmc_gather_state (equal counts only) and the pooled estimator what they are dealing with before any data moves.
template <class T>
int comm_probe(Ctx<T>* c) {
  c->comm_seen = 1; c->comm_chains_total = c->comm_chains_min = c->comm_chains_max = c->N;
  if (!c->comm) return AHMC_OK;   // (a one-rank communicator is probed too: the same two collectives, so the path runs on one GPU)
  int rc = red_buf(c, 4);
  if (rc) return rc;
  double v[4] = {1.0, (double)c->N, (double)c->N, -(double)c->N};
  HIPCHK(hipMemcpyAsync(c->red, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
  NCCLCHK(rccl_api().AllReduce(c->red, c->red, 2, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
  NCCLCHK(rccl_api().AllReduce(c->red + 2, c->red + 2, 2, ncclDouble, ncclMax, static_cast<ncclComm_t>(c->comm), c->stream));
  HIPCHK(hipMemcpyAsync(v, c->red, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->comm_seen = (int64_t)(v[0] + 0.5);
  c->comm_chains_total = (int64_t)(v[1] + 0.5);
  c->comm_chains_max = (int64_t)(v[2] + 0.5);
  c->comm_chains_min = (int64_t)(-v[3] + 0.5);
  if (c->comm_seen != c->comm_ranks)
    return fail(c, AHMC_ERR_RUNTIME, "communicator: " + std::to_string(c->comm_seen) + " rank(s) answered the all-reduce, " + std::to_string(c->comm_ranks) + " were announced");
  return AHMC_OK;
}

template <class T>
int comm_init(Ctx<T>* c, const void* id, int n_ranks, int rank) {
  if (!id) return fail(c, AHMC_ERR_ARGUMENT, "comm_init: id is NULL");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, AHMC_ERR_ARGUMENT, "comm_init: rank outside [0, n_ranks)");
  RcclApi& api = rccl_api();
  if (!api.ok) return fail(c, AHMC_ERR_RUNTIME, "comm_init: " + api.err);
  int rc = comm_release(c);
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  NCCLCHK(api.CommInitRank(&comm, n_ranks, uid, rank));
  c->comm = comm;
  c->comm_owned = true;
  c->comm_ranks = n_ranks;
  c->comm_rank = rank;
  return comm_probe(c);
}

template <class T>
int set_comm(Ctx<T>* c, void* comm, int n_ranks, int rank) {
  if (comm && (n_ranks < 1 || rank < 0 || rank >= n_ranks)) return fail(c, AHMC_ERR_ARGUMENT, "set_comm: rank outside [0, n_ranks)");
  if (comm && !rccl_api().ok) return fail(c, AHMC_ERR_RUNTIME, "set_comm: " + rccl_api().err);
  int rc = comm_release(c);
  if (rc) return rc;
  c->comm = comm;
  c->comm_ranks = comm ? n_ranks : 1;
  c->comm_rank = comm ? rank : 0;
  return comm_probe(c);
}

// ---- moments_reduce (SURVEY §7 kernel list): per-dimension Σ_c Σθ, Σ_c Σθ² and the counters, in double ----------
// One workgroup per dimension block: thread t owns dimension d = blockIdx.x*64 + (t & 63) and strides the chains by
// 4 (t >> 6); a wave reads 64 consecutive dimensions of a chain (512 B, coalesced).  out = [Σθ (D) | Σθ² (D) | Σn_steps, Σn_div]
template <class T>
__global__ __launch_bounds__(256) void k_moments_reduce(const T* __restrict__ s1, const T* __restrict__ s2, const long long* __restrict__ nst,
                                                        const long long* __restrict__ ndiv, int D, int64_t N, double* __restrict__ out) {
  __shared__ double sh[2][4][64];
  const int dl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dl;
  double a = 0, b = 0;
  if (d < D)
    for (int64_t ch = q; ch < N; ch += 4) {
      a += (double)s1[ch * D + d];
      b += (double)s2[ch * D + d];
    }
  sh[0][q][dl] = a;
  sh[1][q][dl] = b;
  __syncthreads();
  if (q == 0 && d < D) {
    out[d] = (sh[0][0][dl] + sh[0][1][dl]) + (sh[0][2][dl] + sh[0][3][dl]);
    out[D + d] = (sh[1][0][dl] + sh[1][1][dl]) + (sh[1][2][dl] + sh[1][3][dl]);
  }
  if (blockIdx.x == 0) {  // the two counters: 256 threads over N, then a fixed-order sum
    __shared__ double cn[2][256];
    double x = 0, y = 0;
    for (int64_t ch = threadIdx.x; ch < N; ch += 256) { x += (double)nst[ch]; y += (double)ndiv[ch]; }
    cn[0][threadIdx.x] = x;
    cn[1][threadIdx.x] = y;
    __syncthreads();
    if (threadIdx.x < 2) {
      double s = 0;
      for (int k = 0; k < 256; ++k) s += cn[threadIdx.x][k];
      out[2 * D + threadIdx.x] = s;
    }
  }
}

// Pooled moments of the kept draws of ALL ranks: mean / var per dimension (host doubles), Σ n_steps, Σ divergences, draws.
// One all-reduce of 2·D + 3 doubles ("ncclReduce of per-dimension first/second moments, 2·D·8 B").
template <class T>
int gather_moments(Ctx<T>* c, double* mean, double* var, int64_t* n_draws, int64_t* total_n_steps, int64_t* n_divergent) {
  const int D = (int)c->D;
  int rc = red_buf(c, (size_t)(2 * D + 3));
  if (rc) return rc;
  hipLaunchKernelGGL((k_moments_reduce<T>), dim3((unsigned)((D + 63) / 64)), dim3(256), 0, c->stream, c->acc_sum, c->acc_sumsq, c->acc_nsteps, c->acc_ndiv,
                     D, c->N, c->red);
  HIPCHK(hipGetLastError());
  const double draws = (double)c->acc_ntrans * (double)c->N;
  HIPCHK(hipMemcpyAsync(c->red + 2 * D + 2, &draws, sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->comm)   // (a one-rank communicator runs the collective too: the single GPU of the test box exercises this call)
    NCCLCHK(rccl_api().AllReduce(c->red, c->red, (size_t)(2 * D + 3), ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
  std::vector<double> h((size_t)(2 * D + 3));
  HIPCHK(hipMemcpyAsync(h.data(), c->red, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const double n = h[(size_t)(2 * D + 2)];
  for (int d = 0; d < D; ++d) {
    const double m = n > 0 ? h[(size_t)d] / n : 0.0;
    if (mean) mean[d] = m;
    if (var) var[d] = n > 0 ? h[(size_t)(D + d)] / n - m * m : 0.0;
  }
  if (n_draws) *n_draws = (int64_t)(n + 0.5);
  if (total_n_steps) *total_n_steps = (int64_t)(h[(size_t)(2 * D)] + 0.5);
  if (n_divergent) *n_divergent = (int64_t)(h[(size_t)(2 * D + 1)] + 0.5);
  return AHMC_OK;
}

// all-gather of the positions: theta_all (device) receives (D, N, n_ranks) — rank r's chains in block r.  Every rank
// must hold the same N (ncclAllGather moves equal counts); 512 MiB per GPU at cfg5.
template <class T>
int gather_state(Ctx<T>* c, void* theta_all) {
  if (!theta_all) return fail(c, AHMC_ERR_ARGUMENT, "gather_state: theta_all is NULL");
  const size_t n = (size_t)c->D * (size_t)c->N;
  if (c->comm) {
    // every rank must contribute the same count (ncclAllGather with unequal counts hangs or corrupts): the largest and the
    // smallest N of the communicator were measured when it was attached (comm_probe)
    if (c->comm_chains_min != c->comm_chains_max)
      return fail(c, AHMC_ERR_ARGUMENT, "gather_state: the ranks hold different numbers of chains (" + std::to_string(c->comm_chains_min) + " … " +
                                            std::to_string(c->comm_chains_max) + "; all ranks must hold the same N)");
    NCCLCHK(rccl_api().AllGather(c->th, theta_all, n, sizeof(T) == 8 ? ncclDouble : ncclFloat, static_cast<ncclComm_t>(c->comm), c->stream));
  } else {
    HIPCHK(hipMemcpyAsync(theta_all, c->th, n * sizeof(T), hipMemcpyDefault, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return AHMC_OK;
}

// ---- pooled variance estimator (AHMC_VAR_POOLED) --------------------------------------------------------------
// Every chain runs its own WelfordVar exactly as in matrix mode (massmatrix.jl:141-150 on (D,N)); an update pools
// them into ONE (D,) estimate.  Chains hold equal counts n, so the merge of Chan et al. is
//     μ = mean_c μ_c,   M = Σ_c M_c + n Σ_c (μ_c − μ)²,   n_tot = n·N
// first over the chains of this GPU (kernel below: partition p = [N_p, μ_p, M_p] per dimension), then over the ranks
// (one all-gather of 2·D + 1 doubles, merged in rank order so every rank gets the same bits), then
//     var = n_tot / ((n_tot + 5)(n_tot − 1)) · M + 10⁻³ · 5 / (n_tot + 5)          (get_estimation, :152-157)
// written to the shared (D,) M⁻¹ and √M⁻¹.
template <class T>
__global__ __launch_bounds__(256) void k_pool_var(const T* __restrict__ mu, const T* __restrict__ M, int D, int64_t N, double n, double* __restrict__ out) {
  __shared__ double sh[3][4][64];
  const int dl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + dl;
  double s_mu = 0, s_M = 0;
  if (d < D)
    for (int64_t ch = q; ch < N; ch += 4) {
      s_mu += (double)mu[ch * D + d];
      s_M += (double)M[ch * D + d];
    }
  sh[0][q][dl] = s_mu;
  sh[1][q][dl] = s_M;
  __syncthreads();
  const double mean = ((sh[0][0][dl] + sh[0][1][dl]) + (sh[0][2][dl] + sh[0][3][dl])) / (double)N;
  double s_d = 0;
  if (d < D)
    for (int64_t ch = q; ch < N; ch += 4) {
      const double df = (double)mu[ch * D + d] - mean;
      s_d += df * df;
    }
  sh[2][q][dl] = s_d;
  __syncthreads();
  if (q == 0 && d < D) {
    out[d] = mean;
    out[D + d] = ((sh[1][0][dl] + sh[1][1][dl]) + (sh[1][2][dl] + sh[1][3][dl])) + n * ((sh[2][0][dl] + sh[2][1][dl]) + (sh[2][2][dl] + sh[2][3][dl]));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * D] = n * (double)N;
}

// parts = n_parts × [μ (D) | M (D) | count]; merged in order, estimate written to minv / sqrt_minv / var (all (D,))
template <class T>
__global__ __launch_bounds__(256) void k_pool_finish(const double* __restrict__ parts, int n_parts, int D, T* __restrict__ minv, T* __restrict__ sqrt_minv,
                                                     T* __restrict__ var_out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const int stride = 2 * D + 1;
  double na = parts[2 * D], mu = parts[d], M = parts[D + d];
  for (int p = 1; p < n_parts; ++p) {
    const double nb = parts[p * stride + 2 * D], mub = parts[p * stride + d], Mb = parts[p * stride + D + d];
    const double delta = mub - mu, nt = na + nb;
    mu = mu + delta * (nb / nt);
    M = M + Mb + delta * delta * (na * nb / nt);
    na = nt;
  }
  const double est = na / ((na + 5) * (na - 1)) * M + 1e-3 * (5 / (na + 5));
  minv[d] = (T)est;
  sqrt_minv[d] = (T)sqrt((T)est);
  var_out[d] = (T)est;
}

template <class T>
int pooled_update(Ctx<T>* c) {
  const int D = (int)c->D;
  const int R = c->comm ? c->comm_ranks : 1;
  const size_t part = (size_t)(2 * D + 1);
  int rc = red_buf(c, part * (size_t)(R + 1));
  if (rc) return rc;
  double* mine = c->red + part * (size_t)R;  // (separate from the gathered block: ncclAllGather out of place)
  hipLaunchKernelGGL((k_pool_var<T>), dim3((unsigned)((D + 63) / 64)), dim3(256), 0, c->stream, c->wv_mu, c->wv_M, D, c->N, (double)c->wv_n, mine);
  HIPCHK(hipGetLastError());
  const double* parts = mine;
  if (c->comm) {
    NCCLCHK(rccl_api().AllGather(mine, c->red, part, ncclDouble, static_cast<ncclComm_t>(c->comm), c->stream));
    parts = c->red;
  }
  hipLaunchKernelGGL((k_pool_finish<T>), dim3((unsigned)((D + 255) / 256)), dim3(256), 0, c->stream, parts, R, D, c->minv, c->sqrt_minv, c->wv_var);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// ---- adaptor state: checkpoint / resume ----------------------------------------------------------------------
template <class T>
int get_adaptor_state(Ctx<T>* c, ahmc_adaptor_state* s, void* da_out, void* wv_out) {
  if (!s) return fail(c, AHMC_ERR_ARGUMENT, "get_adaptor_state: state is NULL");
  const bool has_mm = c->adapt_kind != AHMC_ADAPT_NONE && c->adapt_kind != AHMC_ADAPT_STEPSIZE;
  if (has_mm && c->metric_kind == AHMC_METRIC_DENSE)
    return fail(c, AHMC_ERR_UNSUPPORTED, "get_adaptor_state: the WelfordCov state of a DenseEuclideanMetric adaptor does not round-trip yet");
  memset(s, 0, sizeof(*s));
  s->kind = c->adapt_kind;
  s->var_estimator = c->var_estimator;
  s->init_buffer = c->stan_init; s->term_buffer = c->stan_term; s->window_size = c->stan_window;
  s->adapting = c->adapting ? 1 : 0;
  s->delta = c->da_delta;
  s->stan_i = c->stan_i;
  s->n_adapts = c->windows_n_adapts;
  s->wv_n = c->wv_n;
  s->iteration = (int64_t)c->iteration;
  s->n_welford = (has_mm && c->metric_kind == AHMC_METRIC_DIAG && c->wv_mu) ? (c->var_estimator == AHMC_VAR_NUTPIE ? 5 : 3) : 0;
  s->has_da = (c->adapt_kind != AHMC_ADAPT_NONE && c->adapt_kind != AHMC_ADAPT_MASSMATRIX && c->da_m) ? 1 : 0;
  const size_t N = (size_t)c->N, DN = (size_t)c->D * N;
  if (da_out && s->has_da) {  // (5, N) T: m, ϵ, μ, x̄, H̄  (DAState, stepsize.jl:13-23)
    T* o = static_cast<T*>(da_out);
    std::vector<int32_t> m(N);
    HIPCHK(hipMemcpyAsync(m.data(), c->da_m, sizeof(int32_t) * N, hipMemcpyDeviceToHost, c->stream));
    std::vector<T> mt(N);
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < N; ++i) mt[i] = (T)m[i];
    HIPCHK(hipMemcpyAsync(o, mt.data(), sizeof(T) * N, hipMemcpyDefault, c->stream));
    const T* src[4] = {c->da_eps, c->da_mu, c->da_xbar, c->da_Hbar};
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpyAsync(o + (size_t)(k + 1) * N, src[k], sizeof(T) * N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // (mt is a local)
  }
  if (wv_out && s->n_welford) {  // (3 | 5, D, N) T: μ, M, var [, μ_g, M_g]  (WelfordVar, massmatrix.jl:84-101; NutpieVar :172-190)
    T* o = static_cast<T*>(wv_out);
    const T* src[5] = {c->wv_mu, c->wv_M, c->wv_var, c->wg_mu, c->wg_M};
    for (int k = 0; k < s->n_welford; ++k) HIPCHK(hipMemcpyAsync(o + (size_t)k * DN, src[k], sizeof(T) * DN, hipMemcpyDefault, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return AHMC_OK;
}

template <class T>
int adaptor_init(Ctx<T>* c, int kind, double delta, int ib, int tb, int ws);

template <class T>
int set_adaptor_state(Ctx<T>* c, const ahmc_adaptor_state* s, const void* da_in, const void* wv_in) {
  if (!s) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: state is NULL");
  if (s->kind < AHMC_ADAPT_NONE || s->kind > AHMC_ADAPT_STAN) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: unknown adaptor kind");
  if (s->has_da && !da_in) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: the state has a dual-averaging part but da is NULL");
  if (s->n_welford && !wv_in) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: the state has a variance estimator but welford is NULL");
  // the constructor path allocates and shapes everything (incl. the promotion of a shared M⁻¹ to per-chain); the saved
  // values then overwrite what it initialised.  The metric and the step sizes are the caller's to restore first
  // (ahmc_set_metric / ahmc_set_stepsize), as they are fields of h and κ, not of the adaptor (src/abstractmcmc.jl:11-27).
  if (s->var_estimator != AHMC_VAR_WELFORD && s->var_estimator != AHMC_VAR_NUTPIE && s->var_estimator != AHMC_VAR_POOLED)
    return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: unknown variance estimator");
  if (s->iteration < 0 || s->stan_i < 0 || s->wv_n < 0 || s->n_adapts < 0) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: negative counter");
  c->var_estimator = s->var_estimator;
  int rc = adaptor_init(c, s->kind, s->delta, s->init_buffer, s->term_buffer, s->window_size);
  if (rc) return rc;
  c->adapting = s->adapting != 0;
  c->stan_i = s->stan_i;
  c->wv_n = s->wv_n;
  c->iteration = (uint64_t)s->iteration;
  c->windows_n_adapts = s->n_adapts;
  if (s->kind == AHMC_ADAPT_STAN && s->n_adapts > 0) c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, s->n_adapts);
  const size_t N = (size_t)c->N, DN = (size_t)c->D * N;
  if (s->has_da) {
    if (!c->da_m) return fail(c, AHMC_ERR_STATE, "set_adaptor_state: this adaptor kind has no dual averaging");
    const T* in = static_cast<const T*>(da_in);
    std::vector<T> mt(N);
    HIPCHK(hipMemcpy(mt.data(), in, sizeof(T) * N, hipMemcpyDefault));
    std::vector<int32_t> m(N);
    for (size_t i = 0; i < N; ++i) m[i] = (int32_t)mt[i];
    HIPCHK(hipMemcpyAsync(c->da_m, m.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice, c->stream));
    T* dst[4] = {c->da_eps, c->da_mu, c->da_xbar, c->da_Hbar};
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpyAsync(dst[k], in + (size_t)(k + 1) * N, sizeof(T) * N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (s->n_welford) {
    if (!c->wv_mu) return fail(c, AHMC_ERR_STATE, "set_adaptor_state: a variance estimator needs a DiagEuclideanMetric (set the metric first)");
    if (s->n_welford == 5 && !c->wg_mu) return fail(c, AHMC_ERR_STATE, "set_adaptor_state: NutpieVar state for a WelfordVar adaptor");
    const T* in = static_cast<const T*>(wv_in);
    T* dst[5] = {c->wv_mu, c->wv_M, c->wv_var, c->wg_mu, c->wg_M};
    for (int k = 0; k < s->n_welford; ++k) HIPCHK(hipMemcpyAsync(dst[k], in + (size_t)k * DN, sizeof(T) * DN, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->order_valid = false;
  return AHMC_OK;
}

// ---- EBFMI (src/diagnosis.jl:1-3) from running sums of the energies of the kept transitions ----------------------
// k_nuts / k_hmc leave the last transition's energy in st_H; k_energy_accum folds it into per-chain running sums after
// every per-iteration transition (and the fused kernels do the same themselves, see accumulate_energy):
//   n, E_prev, Σ (E_i − E_{i−1})², Welford (mean, M2) of E.   EBFMI = [Σd² / (n − 1)] / [M2 / (n − 1)]
template <class T>
__global__ __launch_bounds__(256) void k_ebfmi(const T* __restrict__ ea, int64_t N, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const T n = ea[i], sd2 = ea[2 * N + i], M2 = ea[4 * N + i];
  out[i] = n >= 2 ? (sd2 / (n - 1)) / (M2 / (n - 1)) : Lim<T>::nan();
}

// ---- ESS on the device: Geyer's initial monotone sequence on the direct autocovariances ---------------------------
// draws (D, N, K) as ahmc_sample writes them; one thread per (d, c) series.  γ_t by direct sums (a NUTS chain on a
// well-conditioned target stops after a handful of lags; the loop is capped at K/2 pairs), pairs P_t = ρ_2t + ρ_2t+1
// truncated at the first non-positive one and made non-increasing; τ = −1 + 2 Σ P_t; ESS = K / τ  (the estimator of
// advancedhmc.jl_amd/diagnostics.py — the reference computes no ESS itself, MCMCChains.jl does: parity unpinned).
template <class T>
__global__ __launch_bounds__(256) void k_ess(const T* __restrict__ draws, int64_t DN, int64_t K, T* __restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= DN) return;
  double mean = 0, lo = (double)draws[s], hi = lo;
  for (int64_t k = 0; k < K; ++k) {
    const double x = (double)draws[k * DN + s];
    mean += x;
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
  mean /= (double)K;
  // a series that never moved has no autocorrelation to estimate: K.  (Decided on the values, not on γ₀ > 0: Σx / K of K identical
  // values can be an ulp off x, and the "variance" of 1e-32 that leaves gave such a series an ESS of ≈ 1 or K by rounding luck.)
  if (!(hi > lo)) { out[s] = (T)K; return; }
  auto gamma = [&](int64_t t) {
    double g = 0;
    for (int64_t k = 0; k + t < K; ++k) g += ((double)draws[k * DN + s] - mean) * ((double)draws[(k + t) * DN + s] - mean);
    return g / (double)K;
  };
  const double g0 = gamma(0);
  if (!(g0 > 0)) { out[s] = (T)K; return; }
  double tau = -1, prev = 1e300;
  for (int64_t t = 0; 2 * t + 1 < K; ++t) {
    double P = (gamma(2 * t) + gamma(2 * t + 1)) / g0;
    if (!(P > 0)) break;
    P = P < prev ? P : prev;
    prev = P;
    tau += 2 * P;
  }
  if (tau < 1.0 / (double)K) tau = 1.0 / (double)K;
  out[s] = (T)((double)K / tau);
}
