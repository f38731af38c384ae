// ahmc_api.hip — host side of the C ABI (include/ahmc_hip.h) for the HIP engine (gfx950).
// Owns the context (device buffers, stream), validates arguments the way the reference's Julia
// methods do, and enqueues the kernels of ahmc_kernels.hpp.  No CPU compute path exists here:
// every numerical result comes from a kernel.
#include "ahmc_hip.h"
#include "ahmc_inst.hpp"
#include "ahmc_dense.hpp"
#include "ahmc_dense_mn.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the entry points are resolved at run time (ahmc_multi_host.hpp)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

using namespace ahmc;

namespace {

thread_local std::string g_create_err;

struct CtxBase {
  virtual ~CtxBase() {}
  std::string err;
  int dtype = AHMC_F64;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
};

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      c->err = std::string(#expr) + ": " + hipGetErrorString(_e);                         \
      return AHMC_ERR_RUNTIME;                                                            \
    }                                                                                     \
  } while (0)

struct StanWindows {
  int64_t window_start = 0, window_end = 0;
  std::vector<int64_t> splits;
};

// initialize!(::StanHMCAdaptorState, ...) (src/adaptation/stan_adaptor.jl:13-50) — host integers
StanWindows stan_windows(int64_t init_buffer, int64_t term_buffer, int64_t window_size, int64_t n_adapts) {
  StanWindows w;
  w.window_start = init_buffer + 1;
  w.window_end = n_adapts - term_buffer;
  int64_t next_window = init_buffer + window_size;
  while (next_window <= w.window_end) {
    int64_t next_window_boundary = next_window + 2 * window_size;
    if (next_window_boundary > w.window_end) next_window = w.window_end;
    w.splits.push_back(next_window);
    window_size *= 2;
    next_window += window_size;
  }
  if (!w.splits.empty() && w.splits.back() == n_adapts) w.splits.pop_back();
  return w;
}

// static HMC with MultinomialTS on the step-synchronous engine: the transition in progress (ahmc_dense_mn_host.hpp)
struct MnRun {
  int phase = 0;  // MN_NONE / BWD / FWD / REINT / DONE
  int64_t L = 0, n_bwd = 0, n_fwd = 0, i = 0, left = 0;
  bool accum = false;
};

// an ahmc_ext_* run in progress (external target, ask / tell: ahmc_ext_host.hpp)
enum { EXT_IDLE = 0, EXT_NUTS = 1, EXT_HMC = 2, EXT_FINDEPS = 3 };
struct ExtRun {
  int mode = EXT_IDLE;
  ahmc_kernel_cfg cfg{};
  int n_trans = 0;
  int it = 0;            // static HMC: transition in progress
  int64_t L = 0, l = 0;  // static HMC: leapfrogs per transition / completed in this one
  const int* list = nullptr;  // device list of the chains the engine waits for (null = all N)
  int64_t n_list = 0;
  int pp = 0;  // which half of dn_list the next compaction writes
  int64_t steps = 0, max_steps = 0;
  double fe_init = 0;
  int fe_max = 0, fe_it = 0, fe_total = 0;
};

void comm_destroy_raw(void* comm);  // ncclCommDestroy through the run-time binding of ahmc_multi_host.hpp

template <class T>
struct Ctx : CtxBase {
  int64_t D = 0, N = 0;
  int G = 0, E = 0;  // thread geometry chosen for D
  // phase point
  T *vbase = nullptr, *tbase = nullptr;  // slabs; the per-field pointers are aliases into them
  int32_t* ibase = nullptr;
  long long* lbase = nullptr;
  T *th = nullptr, *r = nullptr, *g = nullptr, *lp = nullptr, *lk = nullptr;
  bool have_point = false;
  // target
  int target_kind = AHMC_TARGET_ISO_GAUSS;
  T* tparams = nullptr;
  // AHMC_TARGET_PLUGIN: the user's log-density compiled into the trajectory kernels (a dlopen'ed Inst<T, 4>)
  void* plugin_dl = nullptr;
  const TargetOps<T>* plugin_ops = nullptr;
  // AHMC_TARGET_KERNEL: the user's log-density as a device kernel the step-synchronous engine launches itself
  int uk_kind = -1;           // AHMC_KERNEL_HIP_FUNCTION / AHMC_KERNEL_HIP_SYMBOL
  void* uk_handle = nullptr;
  int uk_block = 256, uk_cpb = 1;
  void* uk_user = nullptr;
  // metric
  int metric_kind = AHMC_METRIC_UNIT;
  bool minv_per_chain = false;
  T *minv = nullptr, *sqrt_minv = nullptr;  // capacity D*N
  int64_t minv_n = 0;
  // integrator
  int integ_kind = AHMC_INTEGRATOR_LEAPFROG;
  double integ_param = 0;
  T *eps_nom = nullptr, *eps_cur = nullptr;
  bool eps_scalar = true;
  double eps_scalar_value = 0.1;
  // rng
  uint64_t seed = 0, chain_offset = 0, chain_stride = 1, iteration = 0;
  // stats
  int32_t *st_nsteps = nullptr, *st_accept = nullptr, *st_depth = nullptr, *st_numerr = nullptr;
  T *st_accrate = nullptr, *st_logdens = nullptr, *st_H = nullptr, *st_Herr = nullptr, *st_maxHerr = nullptr;
  // accumulators
  long long *acc_nsteps = nullptr, *acc_ndiv = nullptr;
  long long *work_prev = nullptr, *work_last = nullptr;  // Σ n_steps before / of the last k_nuts launch (AHMC_NUTS_ORDER_REFRESH)
  T *acc_sum = nullptr, *acc_sumsq = nullptr;
  T* acc_energy = nullptr;  // (5, N): n, E_prev, Σ(ΔE)², mean(E), M2(E) over the kept transitions (EBFMI)
  int64_t acc_ntrans = 0;
  // NUTS scratch
  T* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* order = nullptr;       // chain order for k_nuts (ascending step size), valid while order_valid
  bool order_valid = false, order_from_work = false;
  // Launch length of the sampling phase, found by measurement (sample_from_impl): phase 0 nothing measured yet, 1 the start
  // length is measured, 2 going down, 3 going up, 4 settled.  Reset whenever the step sizes change (the trees change with them).
  // A length is timed over a GROUP of launches (>= 64 transitions, one host synchronisation at each end).
  struct DrawSched {
    int phase = 0; int64_t len = 0, best_len = 0; double best_thr = 0;
    bool primed = false;            // one unmeasured launch has put the dispatch order on measured work
    int64_t g_len = 0; int g_left = 0; std::chrono::steady_clock::time_point g_t0;  // the group being timed
  } sched;
  long long* work_grp = nullptr;  // the accumulators' Σ n_steps per chain at the start of the group being timed
  long long* work_sum = nullptr;  // Σ over chains of the group's work (one device word)
  unsigned* order_hist = nullptr;
  AdaptK<T>* adaptk_dev = nullptr;  // k_nuts MODE 3 arguments
  bool adapting = false;            // an adaptor is initialised and has not seen its last iteration yet
  T* znorm = nullptr;  // standard normals of the momentum draws of one k_nuts launch
  size_t znorm_elems = 0;
  int64_t znorm_cap_trans = 0;  // > 0: the device could not hold the normals of a full batch; launches are capped at this many transitions
  size_t znorm_cap_fail_bytes = 0;  // the smallest allocation that failed when the cap was set: the cap holds while the device cannot offer twice that
  // Round 6: the normals of the NEXT launch are generated on a stream of their own while this launch's k_nuts runs (its tail leaves the chip
  // half empty; k_normals was 3.8 % of cfg2's device time in series with it).  `znorm2` receives them; a launch that finds its normals there
  // swaps the two buffers.  The values are a pure function of (seed, chain, iteration, element): which stream made them cannot show.
  T* znorm2 = nullptr;
  size_t znorm2_elems = 0;
  hipStream_t stream_norm = nullptr;
  hipEvent_t ev_norm_ready = nullptr, ev_z2_free = nullptr;
  bool z2_has_reader = false;       // ev_z2_free marks the end of the last k_nuts that read what is now znorm2
  struct NormPre { bool valid = false; uint64_t iter = 0, k0 = 0, k1 = 0, chain_offset = 0, chain_stride = 0; int64_t n = 0; } npre;
  // (ABI v6) ahmc_set_ref_compat: the reference's matrix-mode early exit (Q1) for ahmc_leapfrog and static EndPointTS transitions
  bool ref_compat = false;
  T* compat_save = nullptr;         // θ, r, g (3·D·N) + the scalar slab (14·N): the state a dry run must give back
  int* compat_flag = nullptr;
  int64_t norm_hint = 0;            // set by the sampling loop before a launch: transitions of the launch that will follow it (0: unknown)
  int norm_prefetch = -1;           // -1 undecided, 0 off (AHMC_NORMALS_PREFETCH=0 or no memory for the second buffer), 1 on
  int64_t norm_prefetch_hits = 0;
  int32_t* redo = nullptr;  // per-chain "redo in the log domain" flags of the NUTS fast pass
  int nuts_blocks = 0;
  // static multinomial
  T* hmc_H = nullptr;
  size_t hmc_H_elems = 0;
  // adaptation
  int adapt_kind = AHMC_ADAPT_NONE;
  double da_delta = 0.8;
  int32_t* da_m = nullptr;
  T *da_eps = nullptr, *da_mu = nullptr, *da_xbar = nullptr, *da_Hbar = nullptr;
  T* da_tab = nullptr;  // √m, m^(−κ) of the dual averaging, tabulated on the device (k_da_table)
  int64_t wv_n = 0, wv_nmin = 10;
  T *wv_mu = nullptr, *wv_M = nullptr, *wv_var = nullptr;
  int var_estimator = AHMC_VAR_WELFORD;            // WelfordVar or NutpieVar (massmatrix.jl:160-250)
  T *wg_mu = nullptr, *wg_M = nullptr, *ext_g = nullptr;  // NutpieVar: gradient estimator state, staging for a caller's gradient
  T *ext_th = nullptr, *ext_alpha = nullptr;  // staging for ahmc_adapt(θ, α) arguments
  int stan_init = 75, stan_term = 50, stan_window = 25;
  int64_t stan_i = 0;
  StanWindows windows;
  int64_t windows_n_adapts = 0;  // the n_adapts `windows` was made for (0 = not yet): part of the adaptor's checkpoint
  // multi-GPU (ahmc_multi_host.hpp): the RCCL communicator of the final gather / the pooled variance estimator
  void* comm = nullptr;
  bool comm_owned = false;
  int comm_ranks = 1, comm_rank = 0;
  int64_t comm_seen = 1, comm_chains_total = 0, comm_chains_min = 0, comm_chains_max = 0;  // what the communicator's own all-reduces said (comm_probe)
  double* red = nullptr;  // device scratch of the cross-chain reductions
  size_t red_elems = 0;
  int n_cu = 256;
  int64_t nuts_launches = 0;  // launches of the dominant NUTS kernel (MODE 0), for bench.py's per-launch roofline
  // HIP events around each of those launches (on this context's stream): bench.py prices the
  // roofline on the kernel's own duration, the quantity rocprofv3's kernel trace reports
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool, ev_pending, ev_pending_warm;
  double nuts_kernel_ns = 0;
  int64_t nuts_warm_launches = 0;  // the same for the warm-up instantiation (MODE 3: adapt! inside the kernel)
  double nuts_warm_kernel_ns = 0;
  // step-synchronous dense engine (ahmc_dense.hpp): M⁻¹, U⁻¹ (D,D); vector slots; per-chain tree state
  T *dn_minv = nullptr, *dn_uinv = nullptr, *dn_W = nullptr, *dn_es = nullptr, *dn_RB = nullptr, *dn_VB = nullptr;
  DChain<T>* dn_S = nullptr;
  // the point pool of the NUTS loop (k_d_tree2): pool, ρ vectors, per-chain state, the point each chain's leapfrog is in
  T *dn_P = nullptr, *dn_R = nullptr;
  DChain2<T>* dn_S2 = nullptr;
  int* dn_ptcur = nullptr;
  int dn_npt = 0, dn_nrho = 0;
  int64_t dn_gemm_big = 0, dn_gemm_small = 0;  // launches of the 64×64-tile / 64×16-tile GEMM (introspection for the tests)
  int dn_last_pipelines = 0, dn_last_pool = 0;
  int* dn_active = nullptr;
  int* dn_list = nullptr;
  size_t dn_slots = 0, dn_batch_elems = 0;
  int64_t dn_global_steps = 0, dn_chain_steps = 0;
  hipStream_t stream2 = nullptr;  // second pipeline of the dense NUTS loop (dn_nuts_transition)
  hipStream_t stream_x[2] = {nullptr, nullptr};  // third / fourth pipeline (AHMC_DENSE_PIPES)
  hipEvent_t ev_join_x[2] = {nullptr, nullptr};
  hipEvent_t ev_split = nullptr, ev_join = nullptr;
  hipEvent_t ev_gemm[2] = {nullptr, nullptr}, ev_tree[2] = {nullptr, nullptr};  // AHMC_DENSE_SPLIT=2: GEMM stream <-> tree stream hand-over per chain half
  // WelfordCov of the shared dense metric: μ (D) [+ batch mean + column-sum partials], M, batch scatter, estimate
  T *wc_mu = nullptr, *wc_M = nullptr, *wc_S = nullptr, *wc_cov = nullptr;
  T* dn_C = nullptr;   // M⁻¹·P (dense metric + dense target), see dn_refresh_fused
  T* dn_Asw = nullptr; // P and M⁻¹·P in MFMA-fragment order (k_dense_swizzle), refreshed at the start of every batch
  int64_t dn_epoch_launches = 0;
  bool dn_fused_ok = false;
  // ahmc_sample(samples_out = host buffer): two device stages; the D2H copy of one batch's draws runs on copy_stream
  // while k_nuts fills the other stage
  T* stage[2] = {nullptr, nullptr};
  size_t stage_elems[2] = {0, 0};
  hipStream_t copy_stream = nullptr;
  hipEvent_t stage_ready[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};
  int64_t wc_n = 0;
  // external target, ask / tell (ahmc_ext_host.hpp): the run in progress; staging for a caller's host (ℓπ, -∇ℓπ)
  ExtRun ext;
  MnRun mn;
  T *ext_gstage = nullptr, *ext_lpstage = nullptr;

  ~Ctx() override {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm && comm_owned) comm_destroy_raw(comm);
    if (red) (void)hipFree(red);
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
    if (stream_norm) { (void)hipStreamSynchronize(stream_norm); (void)hipStreamDestroy(stream_norm); }
    for (hipEvent_t e : {ev_norm_ready, ev_z2_free})
      if (e) (void)hipEventDestroy(e);
    if (znorm2) (void)hipFree(znorm2);
    if (compat_save) (void)hipFree(compat_save);
    if (compat_flag) (void)hipFree(compat_flag);
    for (int k = 0; k < 2; ++k) {
      if (stream_x[k]) { (void)hipStreamSynchronize(stream_x[k]); (void)hipStreamDestroy(stream_x[k]); }
      if (ev_join_x[k]) (void)hipEventDestroy(ev_join_x[k]);
    }
    for (hipEvent_t e : {ev_split, ev_join, ev_gemm[0], ev_gemm[1], ev_tree[0], ev_tree[1]})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {stage_ready[0], stage_ready[1], stage_free[0], stage_free[1]})
      if (e) (void)hipEventDestroy(e);
    void* bufs[] = {vbase, tbase, ibase, lbase, tparams, minv, sqrt_minv, scratch, order, order_hist, adaptk_dev, hmc_H, da_m, da_eps, da_mu, da_xbar,
                    da_Hbar, wv_mu, wv_M, wv_var, ext_th, ext_alpha, redo, znorm, dn_minv, dn_uinv, dn_W, dn_es, dn_RB, dn_VB,
                    dn_S, dn_active, dn_list, wg_mu, wg_M, ext_g, wc_mu, wc_M, wc_S, wc_cov, stage[0], stage[1], dn_C, ext_gstage, ext_lpstage,
                    dn_P, dn_R, dn_S2, dn_ptcur, dn_Asw, da_tab, work_prev, work_last, work_sum, work_grp};
    for (void* b : bufs)
      if (b) (void)hipFree(b);
    for (auto* v : {&ev_pool, &ev_pending, &ev_pending_warm})
      for (auto& e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    if (plugin_dl) (void)dlclose(plugin_dl);
  }
};

template <class T>
int fail(Ctx<T>* c, int code, const std::string& msg) {
  c->err = msg;
  return code;
}

template <class T, class U>
int dev_alloc(Ctx<T>* c, U** ptr, size_t n) {
  HIPCHK(hipMalloc(reinterpret_cast<void**>(ptr), n * sizeof(U)));
  return AHMC_OK;
}

// (G, E) for a given D; the compiled geometries are listed in ahmc_inst.hpp (AHMC_GEOMETRIES).
// The per-chain scalar work of NUTS costs one wave instruction no matter how many chains share the
// wave, so the table packs several chains per wave; E = 4 is the measured optimum for Float64.
inline bool pick_geometry(int64_t D, int& G, int& E) {
  if (D <= 4) { G = 4; E = 1; }
  else if (D <= 8) { G = 4; E = 2; }
  else if (D <= 16) { G = 8; E = 2; }
  else if (D <= 32) { G = 16; E = 2; }
  else if (D <= 64) { G = 32; E = 2; }
  else if (D <= 128) { G = 64; E = 2; }  // measured on cfg2 (f64, batched): (64,2) 1.41e9 at 3 waves/SIMD, (32,4) 1.33e9, (16,8) 0.90e9
  else if (D <= 256) { G = 64; E = 4; }
  else if (D <= 512) { G = 64; E = 8; }
  // multi-wave chains: one chain per workgroup of G/64 waves, reductions cross waves through LDS
  // measured (hier Gaussian, f64, leapfrog/s; E = 8 kernels capped at 256 VGPRs = 2 waves/SIMD, spilling
  // to scratch — at 1 wave/SIMD they ran at about half these rates): D=512 (64,8) 5.5e8 / (128,4) 3.1e8;
  // D=1024 (128,8) 2.4e8 / (256,4) 1.5e8; D=2048 (256,8) 1.08e8 / (512,4) 5.2e7
  else if (D <= 1024) { G = 128; E = 8; }
  else if (D <= 2048) { G = 256; E = 8; }
  else if (D <= 4096) { G = 512; E = 8; }
  else return false;
  const char* ov = getenv("AHMC_GEOMETRY");  // "G,E" override for experiments
  if (ov) {
    int g = 0, e = 0;
    if (sscanf(ov, "%d,%d", &g, &e) == 2 && (int64_t)g * e >= D) {
#define AHMC_GEO_OK(gg, ee) if (g == gg && e == ee) { G = g; E = e; }
      AHMC_GEOMETRIES(AHMC_GEO_OK)
#undef AHMC_GEO_OK
    }
  }
  return true;
}


template <class T>
KP<T> make_kp(Ctx<T>* c) {
  KP<T> p;
  memset(&p, 0, sizeof(p));
  p.D = (int)c->D;
  p.N = c->N;
  p.vbase = c->vbase; p.tbase = c->tbase; p.ibase = c->ibase; p.lbase = c->lbase;
  p.minv = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
  p.sqrt_minv = c->metric_kind == AHMC_METRIC_DIAG ? c->sqrt_minv : nullptr;
  p.minv_per_chain = c->minv_per_chain ? 1 : 0;
  p.lf.kind = c->integ_kind;
  p.lf.sqrt_alpha = c->integ_kind == AHMC_INTEGRATOR_TEMPERED ? (T)std::sqrt((T)c->integ_param) : T(1);
  p.jitter = (T)c->integ_param;
  p.tp.kind = c->target_kind;
  p.tp.D = (int)c->D;
  p.tp.params = c->tparams;
  p.k0 = (uint32_t)c->seed;
  p.k1 = (uint32_t)(c->seed >> 32);
  p.chain_offset = (uint32_t)c->chain_offset;
  p.chain_stride = (uint32_t)c->chain_stride;
  p.iteration = (uint32_t)c->iteration;
  p.scratch = c->scratch;
  p.order = nullptr;
  p.hmc_H = c->hmc_H;
  return p;
}

template <class T>
unsigned group_grid(Ctx<T>* c) {  // blocks of 256 threads covering N groups of G lanes (G > 64: one group per block)
  if (c->G > 64) return (unsigned)c->N;
  int64_t threads = c->N * c->G;
  return (unsigned)((threads + 255) / 256);
}

// the launch table of the context's log-density family: a built-in one, or the plugin's
template <class T>
const TargetOps<T>* ops_for(const Ctx<T>* c) {
  static const TargetOps<T> builtin[AHMC_N_TARGETS] = {make_target_ops<T, 0>(), make_target_ops<T, 1>(), make_target_ops<T, 2>(), make_target_ops<T, 3>()};
  if (c->target_kind == AHMC_TARGET_PLUGIN) return c->plugin_ops;
  if (c->target_kind >= 0 && c->target_kind < AHMC_N_TARGETS) return &builtin[c->target_kind];
  return nullptr;
}

template <class T>
int check_builtin(Ctx<T>* c, const char* what) {
  if (c->target_kind == AHMC_TARGET_PLUGIN && !c->plugin_ops) return fail(c, AHMC_ERR_STATE, std::string(what) + ": no target plugin is bound");
  if (c->target_kind == AHMC_TARGET_EXTERNAL)
    return fail(c, AHMC_ERR_STATE, std::string(what) + " needs a built-in target; with AHMC_TARGET_EXTERNAL the caller evaluates the log-density: "
                                   "ahmc_ext_* (whole transitions, find_good_stepsize) or ahmc_lf_pre / ahmc_lf_post (single leapfrogs)");
  if (dense_engine(c))
    return fail(c, AHMC_ERR_UNSUPPORTED, std::string(what) + " is not implemented for DenseEuclideanMetric / AHMC_TARGET_DENSE_GAUSS");
  return AHMC_OK;
}

template <class T>
int launch_fill_caches_builtin(Ctx<T>* c) {
  KP<T> p = make_kp(c);
  p.no_lk = c->metric_kind == AHMC_METRIC_DENSE ? 1 : 0;  // ℓκ = −½ rᵀM⁻¹r comes from the dense engine
  if (const TargetOps<T>* o = ops_for(c)) o->fill_caches(c->G, c->E, group_grid(c), c->stream, p);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

#include "ahmc_dense_host.hpp"
#include "ahmc_dense_mn_host.hpp"
#include "ahmc_ext_host.hpp"
#include "ahmc_multi_host.hpp"

void comm_destroy_raw(void* comm) {
  if (comm && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(comm));
}

template <class T>
int launch_fill_caches(Ctx<T>* c) {
  return dense_engine(c) ? dn_fill_caches(c) : launch_fill_caches_builtin(c);
}

template <class T>
int launch_kinetic(Ctx<T>* c) {
  if (dense_engine(c)) {
    int rc = dn_ensure(c, 2);
    return rc ? rc : dn_velocity(c);
  }
  KP<T> p = make_kp(c);
  with_geometry(c->G, c->E, [&](auto g, auto e) {
    hipLaunchKernelGGL((k_kinetic<T, decltype(g)::value, decltype(e)::value>), dim3(group_grid(c)), dim3(decltype(g)::value > 64 ? decltype(g)::value : 256), 0,
                       c->stream, p);
  });
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

template <class T>
int set_metric(Ctx<T>* c, int kind, const T* minv, int64_t n) {
  if (kind != AHMC_METRIC_DENSE) c->dn_fused_ok = false;
  if (kind == AHMC_METRIC_UNIT) {
    c->metric_kind = kind;
    c->minv_per_chain = false;
    c->minv_n = 0;
    return AHMC_OK;
  }
  if (kind == AHMC_METRIC_DENSE) {
    if (!minv) return fail(c, AHMC_ERR_ARGUMENT, "set_metric: M⁻¹ pointer is NULL");
    if (n != c->D * c->D) return fail(c, AHMC_ERR_ARGUMENT, "AxesMismatch: dense M⁻¹ must have D*D elements");
    return dn_set_metric(c, minv);
  }
  if (kind != AHMC_METRIC_DIAG) return fail(c, AHMC_ERR_ARGUMENT, "set_metric: unknown metric kind");
  if (!minv) return fail(c, AHMC_ERR_ARGUMENT, "set_metric: M⁻¹ pointer is NULL");
  if (n != c->D && n != c->D * c->N)
    return fail(c, AHMC_ERR_ARGUMENT, "AxesMismatch: diagonal M⁻¹ must have D or D*N elements");
  if (!c->minv) {
    int rc = dev_alloc(c, &c->minv, (size_t)(c->D * c->N));
    if (rc) return rc;
    rc = dev_alloc(c, &c->sqrt_minv, (size_t)(c->D * c->N));
    if (rc) return rc;
  }
  if (minv != c->minv) HIPCHK(hipMemcpyAsync(c->minv, minv, sizeof(T) * n, hipMemcpyDefault, c->stream));
  hipLaunchKernelGGL((k_sqrt<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->minv, c->sqrt_minv, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));  // the source may be a pageable host buffer
  c->metric_kind = kind;
  c->minv_per_chain = (n == c->D * c->N) && c->N != 1;
  c->minv_n = n;
  return AHMC_OK;
}

// Launch plan of k_nuts for this context: one chunk of 64/G chains per single-wave workgroup.
//   occupancy (waves/CU) comes from the register budget; the LDS left per wave then decides how
//   many vector slots (hottest first: pending levels 0,1,2,…, then the dormant ones) live in LDS;
//   the rest go to a per-wave region of global scratch.
template <class T, int MODE>
int plan_nuts(Ctx<T>* c, int max_depth, int criterion, int& blocks, int& wpb, size_t& smem, int& n_lds_slots) {
  const int CPW = c->G >= 64 ? 1 : 64 / c->G;
  const int NW = c->G > 64 ? c->G / 64 : 1;  // waves per chain (multi-wave groups: one chain per workgroup)
  const int NLEV = max_depth > 1 ? max_depth - 1 : 1;
  const int n_slots = (criterion == AHMC_TC_STRICT ? 3 : 2) * NLEV + NUTS_DORMANT + NUTS_CKPT;
  const size_t slot_bytes = (size_t)64 * c->E * sizeof(T);
  const size_t scalar_bytes = (size_t)(NUTS_NSC * NLEV + NUTS_NAT) * CPW * sizeof(T) + (size_t)(NUTS_NSI * NLEV + NUTS_NAI) * CPW * sizeof(int);
  const int64_t n_chunks = (c->N + CPW - 1) / CPW;
  int occ = 0;  // single-wave workgroups per CU
  const TargetOps<T>* o = ops_for(c);
  if (!o) return fail(c, AHMC_ERR_STATE, "k_nuts: the context's target has no fused kernels");
  // the scratch sized here is indexed by kernels of another translation unit (or of a plugin): same layout constants, or no launch
  if (!o->scratch_layout || o->scratch_layout() != nuts_scratch_layout())
    return fail(c, AHMC_ERR_STATE, "k_nuts: the kernels of this target were compiled with another scratch layout (AHMC_CKPT / NUTS_* constants) than "
                                   "the host side of this library: rebuild the library (or the target plugin) as a whole");
  occ = o->nuts_occupancy(c->G, c->E, MODE, scalar_bytes * NW);
  occ *= NW;
  if (occ < 1) occ = 4;
  if (occ > 32) occ = 32;
  const char* ov = getenv("AHMC_NUTS_WAVES_PER_CU");
  if (ov && atoi(ov) > 0) occ = atoi(ov);
  const size_t lds_per_cu = 160 * 1024;
  size_t per_wave = lds_per_cu / (size_t)occ;
  // (a gfx950 workgroup may hold more than 64 KB of LDS — the rate probe runs with 159 KB —: a multi-wave chain's workgroup takes
  // its full share, 80 KB at two workgroups per CU = five 16 KB slots at D = 2 048 instead of four.  AHMC_NUTS_LDS_WG_KB caps it.)
  static const size_t wg_cap = (size_t)(getenv("AHMC_NUTS_LDS_WG_KB") ? atoi(getenv("AHMC_NUTS_LDS_WG_KB")) : 160) * 1024;
  if (per_wave > wg_cap / (size_t)NW) per_wave = wg_cap / (size_t)NW;
  const size_t static_lds = 128 + (NW > 1 ? 1024 / (size_t)NW : 0);  // per wave: the exchange buffers of the reductions (multi-wave chains: one per call site, ahmc_device.hpp)
  per_wave = per_wave > scalar_bytes + static_lds ? per_wave - scalar_bytes - static_lds : 0;
  n_lds_slots = (int)std::min<size_t>((size_t)(n_slots - NUTS_CKPT), per_wave / slot_bytes);   // (the checkpoint triples are cold and addressed per lane: always global)
  const char* ovs = getenv("AHMC_NUTS_LDS_SLOTS");
  if (ovs) n_lds_slots = std::max(0, std::min(n_slots - NUTS_CKPT, atoi(ovs)));
  smem = (size_t)n_lds_slots * slot_bytes + scalar_bytes;
  static const bool dbg = getenv("AHMC_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[ahmc] k_nuts<%s,%d,%d,mode=%d>: occupancy %d waves/CU, %d/%d vector slots in LDS, %zu B LDS/wave, %lld waves\n", sizeof(T) == 8 ? "f64" : "f32", c->G, c->E, MODE, occ, n_lds_slots, n_slots, smem, (long long)n_chunks);
  o->nuts_set_smem(c->G, c->E, MODE, smem);
  // waves per workgroup (AHMC_NUTS_WPB): measured on cfg2, leapfrog/s for 1 / 2 / 4 waves per workgroup =
  // 8.4e8 / 7.0e8 / 5.6e8 — a workgroup holds its LDS and registers until its slowest wave ends
  static const int wpb_env = getenv("AHMC_NUTS_WPB") ? atoi(getenv("AHMC_NUTS_WPB")) : 1;
  wpb = NW > 1 ? NW : std::max(1, std::min(4, wpb_env));
  blocks = NW > 1 ? (int)n_chunks : (int)((n_chunks + wpb - 1) / wpb);
  smem *= (size_t)wpb;
  o->nuts_set_smem(c->G, c->E, MODE, smem);
  size_t need = (size_t)blocks * wpb * (size_t)(n_slots - n_lds_slots) * slot_bytes + 256;
  if (need > c->scratch_bytes) {
    if (c->scratch) {
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipFree(c->scratch));
    }
    c->scratch = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->scratch), need));
    c->scratch_bytes = need;
  }
  return AHMC_OK;
}

template <class T, int MODE>
int launch_nuts(Ctx<T>* c, KP<T> p, int max_depth) {
  const int criterion = p.criterion;
  int blocks = 0, n_lds_slots = 0, wpb = 1;
  size_t smem = 0;
  int rc = plan_nuts<T, MODE>(c, max_depth, criterion, blocks, wpb, smem, n_lds_slots);
  if (rc) return rc;
  p.scratch = c->scratch;
  p.n_lds_levels = n_lds_slots;
#if AHMC_WAVE_TIMELINE
  // measurement build only: one record of eight 64-bit words per wave of the MODE 0 / MODE 3 pass, written to
  // $AHMC_WAVE_TIMELINE.<launch>.bin after the launch (synchronises)
  static int tl_launch = 0;
  unsigned long long* tl_buf = nullptr;
  const char* tl_path = getenv("AHMC_WAVE_TIMELINE_OUT");
  const size_t tl_words = (size_t)p.n_chunks * AHMC_TL_WORDS;
  p.hmc_H = nullptr;
  if (tl_path && (MODE == 0 || MODE == 3)) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&tl_buf), tl_words * 8));
    HIPCHK(hipMemsetAsync(tl_buf, 0, tl_words * 8, c->stream));
    p.hmc_H = reinterpret_cast<T*>(tl_buf);
  }
#endif
  if (const TargetOps<T>* o = ops_for(c)) o->nuts(c->G, c->E, MODE, (unsigned)blocks, wpb, smem, c->stream, p);
  HIPCHK(hipGetLastError());
#if AHMC_WAVE_TIMELINE
  if (tl_buf) {
    std::vector<unsigned long long> h(tl_words);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(h.data(), tl_buf, tl_words * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(tl_buf));
    const std::string fn = std::string(tl_path) + "." + std::to_string(tl_launch++) + ".mode" + std::to_string(MODE) + ".bin";
    if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(h.data(), 8, tl_words, f); fclose(f); }
  }
#endif
  return AHMC_OK;
}

__global__ __launch_bounds__(256) void k_work_since(const long long* __restrict__ acc, const long long* __restrict__ prev, long long* __restrict__ out,
                                                    int64_t N) {  // Σ n_steps a chain has done since `prev` was taken
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = acc[i] - prev[i];
}

__global__ __launch_bounds__(256) void k_work_sum(const long long* __restrict__ acc, const long long* __restrict__ prev, long long* __restrict__ out, int64_t N) {
  long long s = 0;  // Σ over chains of the n_steps done since `prev` was taken
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) s += acc[i] - prev[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)s);
}

// chain dispatch order of k_nuts (see k_order_* in ahmc_kernels.hpp): asynchronous on the stream
template <class T>
int build_order(Ctx<T>* c, int by_work, const long long* work = nullptr) {
  if (!work) work = c->acc_nsteps;
  if (!c->order_hist) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->order_hist), 65536 * sizeof(unsigned)));
  HIPCHK(hipMemsetAsync(c->order_hist, 0, 65536 * sizeof(unsigned), c->stream));
  const unsigned grid = (unsigned)((c->N + 255) / 256);
  hipLaunchKernelGGL((k_order_hist<T>), dim3(grid), dim3(256), 0, c->stream, c->eps_nom, work, by_work, c->order_hist, c->N);
  hipLaunchKernelGGL((k_order_scan<unsigned>), dim3(1), dim3(1024), 0, c->stream, c->order_hist);
  hipLaunchKernelGGL((k_order_scatter<T>), dim3(grid), dim3(256), 0, c->stream, c->eps_nom, work, by_work, c->order_hist, c->order, c->N);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// fold the finished launch timings into nuts_kernel_ns (synchronises the stream)
template <class T>
int flush_nuts_events(Ctx<T>* c) {
  if (c->ev_pending.empty() && c->ev_pending_warm.empty()) return AHMC_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int warm = 0; warm < 2; ++warm) {
    auto& pend = warm ? c->ev_pending_warm : c->ev_pending;
    for (auto& e : pend) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
      (warm ? c->nuts_warm_kernel_ns : c->nuts_kernel_ns) += (double)ms * 1e6;
      c->ev_pool.push_back(e);
    }
    pend.clear();
  }
  return AHMC_OK;
}

// a pair of events around one launch of the dominant kernel (on the context's stream): begin() before, end() after
template <class T>
int nuts_event_begin(Ctx<T>* c, std::pair<hipEvent_t, hipEvent_t>& ev) {
  if (c->ev_pending.size() + c->ev_pending_warm.size() >= 1024) { int rc = flush_nuts_events(c); if (rc) return rc; }
  if (!c->ev_pool.empty()) { ev = c->ev_pool.back(); c->ev_pool.pop_back(); }
  else { HIPCHK(hipEventCreate(&ev.first)); HIPCHK(hipEventCreate(&ev.second)); }
  HIPCHK(hipEventRecord(ev.first, c->stream));
  return AHMC_OK;
}

template <class T>
int nuts_transition(Ctx<T>* c, int max_depth, double delta_max, int criterion, int sampler, double refresh_alpha,
                    bool accum, int n_trans = 1, T* samples_dev = nullptr, const AdaptK<T>* adapt_host = nullptr) {
  if (!c->have_point) return fail(c, AHMC_ERR_STATE, "transition before set_position");
  if (dense_engine(c)) {
    if (sampler != AHMC_TS_MULTINOMIAL && sampler != AHMC_TS_SLICE)
      return fail(c, AHMC_ERR_ARGUMENT, "NUTS supports MultinomialTS and SliceTS");
    if (max_depth < 1) return fail(c, AHMC_ERR_ARGUMENT, "max_depth must be >= 1");
    return dn_nuts_transition(c, max_depth, delta_max, criterion, sampler, refresh_alpha, accum, n_trans, samples_dev);
  }
  int rc = check_builtin(c, "nuts_transition");
  if (rc) return rc;
  if (sampler != AHMC_TS_MULTINOMIAL && sampler != AHMC_TS_SLICE)
    return fail(c, AHMC_ERR_ARGUMENT, "NUTS supports MultinomialTS and SliceTS");
  if (criterion != AHMC_TC_CLASSIC && criterion != AHMC_TC_GENERALISED && criterion != AHMC_TC_STRICT)
    return fail(c, AHMC_ERR_ARGUMENT, "unknown termination criterion");
  if (max_depth < 1) return fail(c, AHMC_ERR_ARGUMENT, "max_depth must be >= 1");
  if (max_depth > 24) return fail(c, AHMC_ERR_UNSUPPORTED, "nuts_transition: the fused kernels support max_depth <= 24 (16.7 M leapfrogs per transition)");
  KP<T> p = make_kp(c);
  p.max_depth = max_depth;
  p.delta_max = (T)delta_max;
  p.criterion = criterion;
  p.sampler = sampler;
  p.refresh_alpha = (T)refresh_alpha;
  p.accum = accum ? 1 : 0;
  const int CPW = c->G >= 64 ? 1 : 64 / c->G;
  p.n_chunks = (unsigned int)((c->N + CPW - 1) / CPW);
  p.redo = c->redo;
  static const bool no_linw = getenv("AHMC_NUTS_LOGW") != nullptr;
  // standard normals of the n_trans momentum refreshes (rand_momentum, src/metric.jl:290-309): already made beside the launch before
  // (prefetch, below), or made now
  const int64_t hint = c->norm_hint;
  c->norm_hint = 0;
  auto normals_grid = [&](int64_t n) {
    const int64_t pairs = ((c->D + 1) / 2) * c->N * n;
    return (unsigned)std::min<int64_t>((pairs + 255) / 256, (int64_t)c->n_cu * 32);
  };
  {
    const size_t need = (size_t)n_trans * (size_t)c->D * (size_t)c->N;
    auto& np = c->npre;
    const bool hit = np.valid && np.iter == c->iteration && np.n >= n_trans && np.k0 == (uint64_t)p.k0 && np.k1 == (uint64_t)p.k1 &&
                     np.chain_offset == (uint64_t)p.chain_offset && np.chain_stride == (uint64_t)p.chain_stride && c->znorm2_elems >= need;
    np.valid = false;   // (used or stale: either way the buffer's content is spent)
    if (hit) {
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_norm_ready, 0));
      std::swap(c->znorm, c->znorm2);
      std::swap(c->znorm_elems, c->znorm2_elems);
      HIPCHK(hipEventRecord(c->ev_z2_free, c->stream));   // everything that read the buffer that is now znorm2 lies before this point
      c->z2_has_reader = true;
      c->norm_prefetch_hits += 1;
    } else {
      if (need > c->znorm_elems) {
        if (c->znorm) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->znorm)); }
        c->znorm = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->znorm), need * sizeof(T)));
        c->znorm_elems = need;
      }
      hipLaunchKernelGGL((k_normals<T>), dim3(normals_grid(n_trans)), dim3(256), 0, c->stream, p, c->znorm, n_trans, (uint32_t)RNG_MOMENTUM);
      HIPCHK(hipGetLastError());
    }
  }
  p.n_trans = n_trans;
  p.znorm = c->znorm;
  p.samples_out = samples_dev;
  const bool no_order = getenv("AHMC_NUTS_NO_ORDER") != nullptr;  // (read per call: the tests toggle it)
  static const bool order_adapt = getenv("AHMC_NUTS_ORDER_ADAPT") ? atoi(getenv("AHMC_NUTS_ORDER_ADAPT")) != 0 : false;  // measured: no gain (a per-transition launch lasts as long as its longest tree)
  if ((n_trans > 1 || order_adapt) && !c->eps_scalar && !no_order) {
    // sampling phase: dispatch the chains in ascending step size (longest expected trees first); the
    // permutation is rebuilt only when the step sizes have changed (host argsort of N floats)
    if (!c->order_valid) {
      int rc2 = build_order(c, 0);
      if (rc2) return rc2;
      c->order_valid = true;
      c->order_from_work = false;
    }
    p.order = c->order;
  }
  if (adapt_host) {
    // warm-up batch: MODE 3 = the log-domain kernel + adapt! after every transition, inside the kernel
    if (!c->adaptk_dev) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->adaptk_dev), sizeof(AdaptK<T>)));
    hipLaunchKernelGGL((k_put<AdaptK<T>>), dim3(1), dim3(64), 0, c->stream, *adapt_host, c->adaptk_dev);
    HIPCHK(hipGetLastError());
    p.adaptk = c->adaptk_dev;
    // dispatch by the step sizes at the start of the batch (they move inside it, but slowly)
    p.order = nullptr;
    if (!c->eps_scalar && !no_order) {
      int rc2 = build_order(c, 0);
      if (rc2) return rc2;
      p.order = c->order;
    }
    if (!no_linw) {  // linear-domain pass, then the log-domain redo pass for the chains it flagged (as MODE 0 → 1)
      p.redo_only = 0;
      std::pair<hipEvent_t, hipEvent_t> ev;
      rc = nuts_event_begin(c, ev);
      if (rc) return rc;
      rc = launch_nuts<T, 3>(c, p, max_depth);
      HIPCHK(hipEventRecord(ev.second, c->stream));
      c->ev_pending_warm.push_back(ev);
      if (rc) return rc;
      c->nuts_warm_launches += 1;
      p.redo_only = 1;
      rc = launch_nuts<T, 4>(c, p, max_depth);
    } else {
      p.redo_only = 0;
      rc = launch_nuts<T, 4>(c, p, max_depth);
    }
  } else if (sampler == AHMC_TS_MULTINOMIAL && criterion == AHMC_TC_GENERALISED && c->integ_kind != AHMC_INTEGRATOR_TEMPERED) {
    if (!no_linw) {
      // fast pass: multinomial weights in the linear domain; chains that came near overflow are
      // flagged and redone, from the same counter-based RNG stream, by the log-domain kernel
      p.redo_only = 0;
      std::pair<hipEvent_t, hipEvent_t> ev;
      rc = nuts_event_begin(c, ev);
      if (rc) return rc;
      rc = launch_nuts<T, 0>(c, p, max_depth);
      HIPCHK(hipEventRecord(ev.second, c->stream));
      c->ev_pending.push_back(ev);
      if (rc) return rc;
      c->nuts_launches += 1;
      p.redo_only = 1;
      rc = launch_nuts<T, 1>(c, p, max_depth);
    } else {
      p.redo_only = 0;
      rc = launch_nuts<T, 1>(c, p, max_depth);
    }
  } else {  // SliceTS, Classic / Strict criteria, TemperedLeapfrog: the general instantiation
    p.redo_only = 0;
    rc = launch_nuts<T, 2>(c, p, max_depth);
  }
  if (rc) return rc;
  c->iteration += (uint64_t)n_trans;
  // ---- the normals of the launch that follows, beside this one ----
  // Measured (profiles/r6_experiments.md r6n): cfg3's 4-transition launches gain 6 % in the sampling phase (3.20 -> 3.40e9: the 0.1 ms of
  // k_normals and its launch gap no longer sit between two 6 ms launches); cfg2's 256-transition launches lose 0.4 % (a 5.7 ms k_normals beside
  // a VALU-bound k_nuts takes what it gives) and cfg5's 32 are unchanged — so only launches whose normals are at most 2 GiB are prefetched.
  const size_t prefetch_max_bytes = getenv("AHMC_NORMALS_PREFETCH_MAX_MB") ? (size_t)atoll(getenv("AHMC_NORMALS_PREFETCH_MAX_MB")) << 20 : (size_t)2 << 30;
  if (hint > 0 && refresh_alpha == 0 && (size_t)hint * (size_t)c->D * (size_t)c->N * sizeof(T) <= prefetch_max_bytes) {
    if (c->norm_prefetch < 0) c->norm_prefetch = (getenv("AHMC_NORMALS_PREFETCH") && atoi(getenv("AHMC_NORMALS_PREFETCH")) == 0) ? 0 : 1;
    const size_t need2 = (size_t)hint * (size_t)c->D * (size_t)c->N;
    if (c->norm_prefetch == 1 && need2 > c->znorm2_elems) {
      // a second buffer only where the device has room to spare (the draws of a run, another context): 2x its size must be free
      size_t free_b = 0, total_b = 0;
      const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + c->znorm2_elems * sizeof(T) >= 2 * need2 * sizeof(T);
      (void)hipGetLastError();
      if (room) {
        if (c->znorm2) { HIPCHK(hipStreamSynchronize(c->stream)); if (c->stream_norm) HIPCHK(hipStreamSynchronize(c->stream_norm)); HIPCHK(hipFree(c->znorm2)); c->z2_has_reader = false; }
        c->znorm2 = nullptr;
        c->znorm2_elems = 0;
        if (hipMalloc(reinterpret_cast<void**>(&c->znorm2), need2 * sizeof(T)) == hipSuccess) c->znorm2_elems = need2;
        else { (void)hipGetLastError(); c->znorm2 = nullptr; c->norm_prefetch = 0; }
      } else if (!c->znorm2) {
        c->norm_prefetch = 0;
      }
    }
    if (c->norm_prefetch == 1 && c->znorm2 && need2 <= c->znorm2_elems) {
      if (!c->stream_norm) {
        HIPCHK(hipStreamCreateWithFlags(&c->stream_norm, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_norm_ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_z2_free, hipEventDisableTiming));
      }
      if (c->z2_has_reader) HIPCHK(hipStreamWaitEvent(c->stream_norm, c->ev_z2_free, 0));
      KP<T> p2 = p;
      p2.iteration = (uint32_t)c->iteration;   // (make_kp's field: the launch that follows starts here)
      hipLaunchKernelGGL((k_normals<T>), dim3(normals_grid(hint)), dim3(256), 0, c->stream_norm, p2, c->znorm2, (int)hint, (uint32_t)RNG_MOMENTUM);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(c->ev_norm_ready, c->stream_norm));
      c->npre.valid = true;
      c->npre.iter = c->iteration; c->npre.n = hint;
      c->npre.k0 = (uint64_t)p.k0; c->npre.k1 = (uint64_t)p.k1; c->npre.chain_offset = (uint64_t)p.chain_offset; c->npre.chain_stride = (uint64_t)p.chain_stride;
    }
  }
  return AHMC_OK;
}

// An out-of-memory cap on the launch length (reserve_normals) holds only while the device cannot offer twice the allocation that
// failed — counting the buffer the context holds as free; after that the full request is tried again.
template <class T>
static bool normals_cap_holds(const Ctx<T>* c, size_t free_b) {
  if (c->znorm_cap_trans <= 0) return false;
  return (free_b + c->znorm_elems * sizeof(T)) / 2 < c->znorm_cap_fail_bytes;
}

template <class T>
int64_t nuts_batch(Ctx<T>* c) {
  // transitions per launch of k_nuts (sampling AND fused warm-up): more = less tree-size tail per launch (round 1, cfg2:
  // 8/16/32/64 -> 1.35/1.39/1.42/1.43e9 leapfrog/s); the pre-generated momentum normals take batch * D * N elements — the
  // caps and what they were measured against are at the two return statements below
  const int batch_env = getenv("AHMC_NUTS_BATCH") ? atoi(getenv("AHMC_NUTS_BATCH")) : 0;  // (read per call: the tests toggle it)
  if (batch_env > 0) return batch_env;
  if (dense_engine(c)) {
    // dense engine: chains run asynchronously through the batch and only its end has idle chains, so longer is better.  Three
    // (batch, D, N) arrays (normals, momenta, M⁻¹·momenta): up to 1 024 transitions under min(128 GiB, a third of what is free
    // on THIS device now — several contexts on one GPU, a user log-density at large D·N, a part with less than 288 GB; the
    // buffers this context already holds count as free).  History: round 1 64 under 8 GiB, round 2 256 under 48 GiB, round 3
    // 1 024 under 128 GiB (with the step-size adaptation inside the tree kernel the warm-up runs in batches as well, and the idle
    // tail of a batch is the same number of global steps however long the batch is).
    size_t free_b = 0, total_b = 0;
    size_t budget = 128ull << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t held = c->dn_batch_elems * 2 * sizeof(T) + c->znorm_elems * sizeof(T);
      budget = std::min<size_t>(budget, (free_b + held) / 3);
    }
    (void)hipGetLastError();
    const int64_t capd = (int64_t)budget / (int64_t)(3 * sizeof(T) * c->D * c->N);
    return std::max<int64_t>(1, std::min<int64_t>(1024, capd));
  }
  // Round 2: 32 -> 128 under an 8 GiB cap.  A launch cannot end before its slowest chain, and in the warm-up (step sizes
  // still moving, the dual averaging restarting at every window end) a few chains build 10-30x the mean tree for a
  // while: over 31 transitions they set the launch time (44 ms measured against 28 ms of work); over 125 they average
  // out.  cfg2 warm-up 1.84e9 -> 2.00e9 leapfrog/s (batch 64: 1.94e9), sampling phase +1 %.
  // Round 3: 128 -> 256 under 16 GiB (cfg2 whole loop, two runs each: 125 per launch 2.39–2.42e9, 250 2.433e9, 500 2.445e9,
  // 1 000 2.450e9 — but 64 GiB of normals cost a second to allocate and first-touch).
  // cfg3 (D = 32: 16 MiB of normals per transition; heavy-tailed funnel trees): 256 per launch 1.51e9, 512 1.63e9 — the count is
  // bounded by the bytes only (cfg2: 256, cfg3: 1 024, cfg5: 32 — where 50 / 100 per launch measured slower: 6.44 / 6.30e7
  // against 6.58e7, the dispatch order is re-sorted by measured work between launches).
  // Round 4: the 16 GiB are also bounded by what is free on THIS device now (half of it; the buffer this context already holds
  // counts as free) — several contexts on one GPU or a part with less than 288 GB get shorter launches instead of an
  // out-of-memory error — and by what reserve_normals could actually get (znorm_cap_trans).
  size_t budget = 16ull << 30, free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<size_t>(budget, (free_b + c->znorm_elems * sizeof(T)) / 2);
  (void)hipGetLastError();
  int64_t cap = (int64_t)budget / (int64_t)(sizeof(T) * c->D * c->N);
  if (normals_cap_holds(c, free_b)) cap = std::min<int64_t>(cap, c->znorm_cap_trans);
  return std::max<int64_t>(1, std::min<int64_t>(1024, cap));
}

// The momentum normals of a launch of `n_trans` transitions: make sure the buffer holds them, BEFORE the launches of a run
// (a 16 GiB hipMalloc inside the first launch of a timed run was seen to take up to a second).  Out of memory is not an
// error: the launch length is halved until the buffer fits, and the context remembers the cap.
template <class T>
int reserve_normals(Ctx<T>* c, int64_t n_trans) {
  if (n_trans < 1) n_trans = 1;
  const int64_t requested = n_trans;  // (before the cap: what decides whether a success lifts it)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  (void)hipGetLastError();
  if (normals_cap_holds(c, free_b)) n_trans = std::min<int64_t>(n_trans, c->znorm_cap_trans);
  size_t need = (size_t)n_trans * (size_t)c->D * (size_t)c->N;
  if (need <= c->znorm_elems) return AHMC_OK;
  if (c->znorm) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->znorm)); }
  c->znorm = nullptr;
  c->znorm_elems = 0;
  for (;;) {
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->znorm), need * sizeof(T));
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    c->znorm = nullptr;
    if (e != hipErrorOutOfMemory || n_trans <= 1) return fail(c, AHMC_ERR_RUNTIME, std::string("momentum normals: hipMalloc: ") + hipGetErrorString(e));
    c->znorm_cap_fail_bytes = need * sizeof(T);
    n_trans = (n_trans + 1) / 2;
    c->znorm_cap_trans = n_trans;
    need = (size_t)n_trans * (size_t)c->D * (size_t)c->N;
  }
  c->znorm_elems = need;
  // The cap is what the device could hold THEN.  Once memory has been freed (normals_cap_holds: twice the failed allocation is
  // on offer) the request is no longer clamped, and getting all of it lifts the cap for good.
  if (c->znorm_cap_trans > 0 && n_trans == requested && requested > c->znorm_cap_trans) {
    c->znorm_cap_trans = 0;
    c->znorm_cap_fail_bytes = 0;
  }
  return AHMC_OK;
}

// nsteps(τ) for FixedIntegrationTime(λ) (src/trajectory.jl:241-243): max(1, floor(λ / nominal step size)).  Needs ONE
// step size: a scalar one, or the single chain's own (which adaptation keeps on the device)
template <class T>
int resolve_integration_time(Ctx<T>* c, double lambda, int64_t& L) {
  double eps = (double)(T)c->eps_scalar_value;
  if (!c->eps_scalar) {
    if (c->N != 1) return fail(c, AHMC_ERR_ARGUMENT, "FixedIntegrationTime needs a scalar step size (src/trajectory.jl:241-243)");
    T e1;
    HIPCHK(hipMemcpyAsync(&e1, c->eps_nom, sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    eps = (double)e1;
  }
  const int64_t n = (int64_t)std::floor(lambda / eps);
  L = n < 1 ? 1 : n;
  return AHMC_OK;
}

// ---- (ABI v6) the reference's matrix-mode early exit, opt-in (src/integrator.jl:252-258) ----
template <class T>
__global__ __launch_bounds__(256) void k_any_nonfinite(const T* __restrict__ lp, const T* __restrict__ lk, int64_t N, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // (ℓπ, ℓκ are stored sanitised: a non-finite value is −Inf — `isfinite(z)` of src/hamiltonian.jl:141-142 is false for it)
  if (i < N && !(is_finite(lp[i]) && is_finite(lk[i]))) atomicOr(flag, 1);
}
// one leapfrog step of every chain at a time (k_leapfrog with n = ±1), stopping ALL chains after the first step that left any chain
// non-finite; `done` = the steps taken.  One launch and one 4-byte read-back per step: a comparison mode.
template <class T>
int compat_step_loop(Ctx<T>* c, int64_t n_abs, bool fwd, int64_t& done) {
  if (c->integ_kind == AHMC_INTEGRATOR_TEMPERED)
    return fail(c, AHMC_ERR_UNSUPPORTED, "ahmc_set_ref_compat: the coupled early exit is not implemented for TemperedLeapfrog (the step index of a launch of one step)");
  if (!c->compat_flag) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->compat_flag), sizeof(int)));
  const TargetOps<T>* o = ops_for(c);
  if (!o) return fail(c, AHMC_ERR_STATE, "ahmc_set_ref_compat: the context's target has no fused kernels");
  done = 0;
  for (int64_t s = 1; s <= n_abs; ++s) {
    KP<T> p = make_kp(c);
    p.n_steps = fwd ? 1 : -1;
    o->leapfrog(c->G, c->E, group_grid(c), c->stream, p);
    HIPCHK(hipMemsetAsync(c->compat_flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL((k_any_nonfinite<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->lp, c->lk, c->N, c->compat_flag);
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->compat_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    done = s;
    if (flag) break;
  }
  return AHMC_OK;
}
// static EndPointTS transition in compat mode: the step after which the reference's loop would have ended for everybody, found by a DRY RUN
// of the transition's integration (fresh momentum from the transition's own counter-based stream, one step at a time) on a copy of the state
template <class T>
int compat_hmc_stop_step(Ctx<T>* c, int64_t L, double refresh_alpha, int64_t& L_eff) {
  const size_t DN = (size_t)c->D * (size_t)c->N, nv = 3 * DN, ns = 14 * (size_t)c->N;
  if (!c->compat_save) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->compat_save), (nv + ns) * sizeof(T)));
  HIPCHK(hipMemcpyAsync(c->compat_save, c->vbase, nv * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->compat_save + nv, c->tbase, ns * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  const TargetOps<T>* o = ops_for(c);
  if (!o) return fail(c, AHMC_ERR_STATE, "ahmc_set_ref_compat: the context's target has no fused kernels");
  KP<T> p = make_kp(c);
  p.refresh_alpha = (T)refresh_alpha;
  o->refresh(c->G, c->E, group_grid(c), c->stream, p);   // the momentum k_hmc will draw (same stream, same counters) and the caches
  HIPCHK(hipGetLastError());
  int64_t done = 0;
  int rc = compat_step_loop(c, L, true, done);
  L_eff = done > 0 ? done : L;
  HIPCHK(hipMemcpyAsync(c->vbase, c->compat_save, nv * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->tbase, c->compat_save + nv, ns * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  return rc;
}

template <class T>
int hmc_transition(Ctx<T>* c, int64_t L, double lambda, int sampler, double refresh_alpha, bool accum) {
  if (!c->have_point) return fail(c, AHMC_ERR_STATE, "transition before set_position");
  int rc = dense_engine(c) ? AHMC_OK : check_builtin(c, "hmc_transition");
  if (rc) return rc;
  if (sampler != AHMC_TS_ENDPOINT && sampler != AHMC_TS_MULTINOMIAL)
    return fail(c, AHMC_ERR_ARGUMENT, "static HMC supports EndPointTS and MultinomialTS");
  if (lambda > 0) {  // nsteps(τ) for FixedIntegrationTime (src/trajectory.jl:241-243)
    int rc2 = resolve_integration_time(c, lambda, L);
    if (rc2) return rc2;
  }
  if (L < 0) L = -L;
  if (dense_engine(c)) {
    if (c->ref_compat && sampler == AHMC_TS_ENDPOINT) return fail(c, AHMC_ERR_UNSUPPORTED, "ahmc_set_ref_compat is not implemented on the dense engine");
    return dn_hmc_transition(c, L, sampler, refresh_alpha, accum);
  }
  if (sampler == AHMC_TS_MULTINOMIAL) {
    size_t need = (size_t)(L + 1) * (size_t)c->N;
    if (need > c->hmc_H_elems) {
      if (c->hmc_H) HIPCHK(hipFree(c->hmc_H));
      c->hmc_H = nullptr;
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->hmc_H), need * sizeof(T)));
      c->hmc_H_elems = need;
    }
  }
  int64_t L_eff = L;
  if (c->ref_compat && sampler == AHMC_TS_ENDPOINT) {   // Q1, opt-in: every chain integrates as far as the reference's coupled loop would
    int rc3 = compat_hmc_stop_step(c, L, refresh_alpha, L_eff);
    if (rc3) return rc3;
  }
  KP<T> p = make_kp(c);
  p.L = L_eff;
  p.sampler = sampler;
  p.refresh_alpha = (T)refresh_alpha;
  p.accum = accum ? 1 : 0;
  if (const TargetOps<T>* o = ops_for(c)) o->hmc(c->G, c->E, group_grid(c), c->stream, p);
  HIPCHK(hipGetLastError());
  if (L_eff != L) {  // (the statistic is the trajectory's nominal length, src/trajectory.jl:288)
    static_assert(sizeof(int32_t) == 4, "n_steps is a 32-bit statistic");
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->ibase), (int)L, (size_t)c->N, c->stream));   // (KP::st_nsteps() = ibase[0 … N))
  }
  c->iteration += 1;
  return AHMC_OK;
}

template <class T>
int reset_accum(Ctx<T>* c) {
  HIPCHK(hipMemsetAsync(c->acc_nsteps, 0, sizeof(long long) * c->N, c->stream));
  HIPCHK(hipMemsetAsync(c->acc_ndiv, 0, sizeof(long long) * c->N, c->stream));
  HIPCHK(hipMemsetAsync(c->acc_sum, 0, sizeof(T) * c->D * c->N, c->stream));
  HIPCHK(hipMemsetAsync(c->acc_sumsq, 0, sizeof(T) * c->D * c->N, c->stream));
  HIPCHK(hipMemsetAsync(c->acc_energy, 0, sizeof(T) * 5 * c->N, c->stream));
  c->acc_ntrans = 0;
  return AHMC_OK;
}

template <class T>
int adaptor_init(Ctx<T>* c, int kind, double delta, int ib, int tb, int ws) {
  c->adapt_kind = kind;
  c->adapting = kind != AHMC_ADAPT_NONE;
  c->da_delta = delta;
  c->stan_init = ib; c->stan_term = tb; c->stan_window = ws;
  c->stan_i = 0;
  c->windows_n_adapts = 0;
  if (kind == AHMC_ADAPT_NONE) return AHMC_OK;
  int rc;
  if (!c->da_m) {
    if ((rc = dev_alloc(c, &c->da_m, (size_t)c->N))) return rc;
    if ((rc = dev_alloc(c, &c->da_eps, (size_t)c->N))) return rc;
    if ((rc = dev_alloc(c, &c->da_mu, (size_t)c->N))) return rc;
    if ((rc = dev_alloc(c, &c->da_xbar, (size_t)c->N))) return rc;
    if ((rc = dev_alloc(c, &c->da_Hbar, (size_t)c->N))) return rc;
    if ((rc = dev_alloc(c, &c->da_tab, (size_t)4 * DA_TAB_M))) return rc;
    hipLaunchKernelGGL((k_da_table<T>), dim3(DA_TAB_M / 256), dim3(256), 0, c->stream, c->da_tab, T(DA_KAPPA), T(DA_GAMMA), T(DA_T0));
    HIPCHK(hipGetLastError());
  }
  // NesterovDualAveraging(δ, ϵ) → DAState(ϵ) (src/adaptation/stepsize.jl:25-33): a reset with ϵ = nominal ϵ
  HIPCHK(hipMemcpyAsync(c->da_eps, c->eps_nom, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
  {
    AdaptP<T> a;
    memset(&a, 0, sizeof(a));
    a.N = c->N;
    a.da_reset = 1;
    a.da_m = c->da_m; a.da_eps = c->da_eps; a.da_mu = c->da_mu; a.da_xbar = c->da_xbar; a.da_Hbar = c->da_Hbar;
    a.eps_nom = c->eps_nom;
    hipLaunchKernelGGL((k_adapt_da<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
  }
  if (c->metric_kind == AHMC_METRIC_DENSE && kind != AHMC_ADAPT_STEPSIZE) {  // WelfordCov (src/AdvancedHMC.jl:116-118)
    if ((rc = dn_cov_init(c))) return rc;
  }
  // WelfordVar{T}(size(metric); var = copy(M⁻¹)) per chain (src/AdvancedHMC.jl:113-115)
  if (c->metric_kind == AHMC_METRIC_DIAG && kind != AHMC_ADAPT_STEPSIZE) {
    const int64_t DN = c->D * c->N;
    if (!c->wv_mu) {
      if ((rc = dev_alloc(c, &c->wv_mu, (size_t)DN))) return rc;
      if ((rc = dev_alloc(c, &c->wv_M, (size_t)DN))) return rc;
      if ((rc = dev_alloc(c, &c->wv_var, (size_t)DN))) return rc;
    }
    if (c->var_estimator == AHMC_VAR_POOLED) {
      // one (D,) M⁻¹ for all chains, estimated from all of them (and from all GPUs when a communicator is set)
      if (c->minv_per_chain) return fail(c, AHMC_ERR_ARGUMENT, "adaptor_init: AHMC_VAR_POOLED adapts ONE shared (D,) M⁻¹: set a (D,) DiagEuclideanMetric");
      HIPCHK(hipMemcpyAsync(c->wv_var, c->minv, sizeof(T) * c->D, hipMemcpyDeviceToDevice, c->stream));
    } else if (!c->minv_per_chain && c->N != 1) {  // promote a shared (D,) M⁻¹ to per-chain (D,N)
      hipLaunchKernelGGL((k_bcast_cols<T>), dim3((unsigned)((DN + 255) / 256)), dim3(256), 0, c->stream, c->minv,
                         c->wv_var, c->D, c->N);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(c->minv, c->wv_var, sizeof(T) * DN, hipMemcpyDeviceToDevice, c->stream));
      hipLaunchKernelGGL((k_sqrt<T>), dim3((unsigned)((DN + 255) / 256)), dim3(256), 0, c->stream, c->minv, c->sqrt_minv, DN);
      HIPCHK(hipGetLastError());
      c->minv_per_chain = true;
      c->minv_n = DN;
    } else {
      HIPCHK(hipMemcpyAsync(c->wv_var, c->minv, sizeof(T) * DN, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipMemsetAsync(c->wv_mu, 0, sizeof(T) * DN, c->stream));
    HIPCHK(hipMemsetAsync(c->wv_M, 0, sizeof(T) * DN, c->stream));
    if (c->var_estimator == AHMC_VAR_NUTPIE) {
      if (!c->wg_mu) {
        if ((rc = dev_alloc(c, &c->wg_mu, (size_t)DN))) return rc;
        if ((rc = dev_alloc(c, &c->wg_M, (size_t)DN))) return rc;
      }
      HIPCHK(hipMemsetAsync(c->wg_mu, 0, sizeof(T) * DN, c->stream));
      HIPCHK(hipMemsetAsync(c->wg_M, 0, sizeof(T) * DN, c->stream));
    }
    c->wv_n = 0;
  }
  return AHMC_OK;
}

// adapt!(h, κ, adaptor, i, n_adapts, z, α) + update(h/κ, adaptor) (src/sampler.jl:72-90, :3-22)
template <class T>
int adapt(Ctx<T>* c, int64_t i, int64_t n_adapts, const T* th_ext = nullptr, const T* alpha_ext = nullptr, const T* g_ext = nullptr) {
  if (c->adapt_kind == AHMC_ADAPT_NONE || i > n_adapts) return AHMC_OK;
  if (i == n_adapts) c->adapting = false;
  const bool has_ss = c->adapt_kind != AHMC_ADAPT_MASSMATRIX;
  const bool has_mm = c->adapt_kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DIAG;
  const bool has_cov = c->adapt_kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DENSE;  // WelfordCov
  bool do_push = false, do_update = false, wv_reset = false, da_reset = false;
  if (c->adapt_kind == AHMC_ADAPT_STAN) {
    if (i == 1 || c->windows_n_adapts != n_adapts) {  // initialize! (also after a resume that did not carry the schedule)
      c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, n_adapts);
      c->windows_n_adapts = n_adapts;
    }
    c->stan_i += 1;  // adapt!(tp::StanHMCAdaptor, ...) (stan_adaptor.jl:137-159)
    const bool in_window = c->stan_i >= c->windows.window_start && c->stan_i <= c->windows.window_end;
    const bool window_end = std::find(c->windows.splits.begin(), c->windows.splits.end(), c->stan_i) != c->windows.splits.end();
    if (in_window && (has_mm || has_cov)) {
      do_push = true;
      do_update = window_end;
    }
    if (window_end) {
      da_reset = true;
      wv_reset = has_mm || has_cov;
    }
  } else if (has_mm || has_cov) {
    do_push = true;
    do_update = true;
  }
  AdaptP<T> a;
  memset(&a, 0, sizeof(a));
  a.N = c->N;
  a.DN = c->D * c->N;
  if (has_ss) {
    a.do_da = 1;
    a.da_reset = da_reset ? 1 : 0;
    a.da_finalize = (i == n_adapts) ? 1 : 0;
    a.delta = (T)c->da_delta; a.gamma = T(DA_GAMMA); a.t0 = T(DA_T0); a.kappa = T(DA_KAPPA);  // stepsize.jl:168-172
    a.da_m = c->da_m; a.da_eps = c->da_eps; a.da_mu = c->da_mu; a.da_xbar = c->da_xbar; a.da_Hbar = c->da_Hbar;
    a.alpha = c->st_accrate;
    a.da_tab = c->da_tab;
    if (alpha_ext) {  // caller-supplied α (host or device): stage it on the stream
      if (!c->ext_alpha) { int rc2 = dev_alloc(c, &c->ext_alpha, (size_t)c->N); if (rc2) return rc2; }
      HIPCHK(hipMemcpyAsync(c->ext_alpha, alpha_ext, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
      a.alpha = c->ext_alpha;
    }
    a.eps_nom = c->eps_nom;
    hipLaunchKernelGGL((k_adapt_da<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    c->eps_scalar = false;
    c->order_valid = false; c->sched = {};
  }
  if (has_cov && (do_push || wv_reset)) {
    const T* th = c->th;
    if (th_ext) {  // caller-supplied θ
      if (!c->ext_th) { int rc2 = dev_alloc(c, &c->ext_th, (size_t)a.DN); if (rc2) return rc2; }
      HIPCHK(hipMemcpyAsync(c->ext_th, th_ext, sizeof(T) * a.DN, hipMemcpyDefault, c->stream));
      th = c->ext_th;
    }
    int rc2 = AHMC_OK;
    if (do_push) rc2 = dn_cov_push(c, th);
    if (!rc2 && do_update) rc2 = dn_cov_update(c);
    if (!rc2 && wv_reset) rc2 = dn_cov_init(c);
    if (rc2) return rc2;
  }
  if (has_mm && (do_push || wv_reset)) {
    if (do_push) c->wv_n += 1;
    const bool pooled = c->var_estimator == AHMC_VAR_POOLED;
    a.do_push = do_push ? 1 : 0;
    a.do_update = (!pooled && do_update && c->wv_n >= c->wv_nmin) ? 1 : 0;  // update!(ve) (massmatrix.jl:60-62)
    a.wv_reset = (!pooled && wv_reset) ? 1 : 0;
    a.wv_n = (T)c->wv_n;
    a.th = c->th;
    if (th_ext) {  // caller-supplied θ
      if (!c->ext_th) { int rc2 = dev_alloc(c, &c->ext_th, (size_t)a.DN); if (rc2) return rc2; }
      HIPCHK(hipMemcpyAsync(c->ext_th, th_ext, sizeof(T) * a.DN, hipMemcpyDefault, c->stream));
      a.th = c->ext_th;
    }
    a.wv_mu = c->wv_mu; a.wv_M = c->wv_M; a.wv_var = c->wv_var;
    a.minv = c->minv; a.sqrt_minv = c->sqrt_minv;
    if (c->var_estimator == AHMC_VAR_NUTPIE) {
      if (th_ext && !g_ext)  // massmatrix.jl:234-236
        return fail(c, AHMC_ERR_ARGUMENT, "`NutpieVar` adaptation requires position and gradient information!");
      a.nutpie = 1;
      a.gr = c->g;
      if (g_ext) {
        if (!c->ext_g) { int rc2 = dev_alloc(c, &c->ext_g, (size_t)a.DN); if (rc2) return rc2; }
        HIPCHK(hipMemcpyAsync(c->ext_g, g_ext, sizeof(T) * a.DN, hipMemcpyDefault, c->stream));
        a.gr = c->ext_g;
      }
      a.wg_mu = c->wg_mu; a.wg_M = c->wg_M;
    }
    hipLaunchKernelGGL((k_adapt_wv<T>), dim3((unsigned)((a.DN + 255) / 256)), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    if (pooled) {  // the per-chain estimators are pooled into the shared (D,) M⁻¹; reset them afterwards
      if (do_update && c->wv_n >= c->wv_nmin) { int rc2 = pooled_update(c); if (rc2) return rc2; }
      if (wv_reset) {
        HIPCHK(hipMemsetAsync(c->wv_mu, 0, sizeof(T) * a.DN, c->stream));
        HIPCHK(hipMemsetAsync(c->wv_M, 0, sizeof(T) * a.DN, c->stream));
      }
    }
    if (wv_reset) c->wv_n = 0;
  }
  return AHMC_OK;
}

// A batch of k warm-up transitions i .. i+k-1 with the adaptor's adapt! done inside the kernel (k_nuts MODE 3); the host
// only mirrors the bookkeeping that is the same for every chain (Stan window counter, Welford count).
// Stage `slot` of ahmc_sample's host-output path, at least `need` elements, not in use by a copy any more
// (device-side wait: the next k_nuts that writes it is ordered after the D2H copy that reads it).
template <class T>
int stage_acquire(Ctx<T>* c, int slot, size_t need) {
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int s = 0; s < 2; ++s) {
      HIPCHK(hipEventCreateWithFlags(&c->stage_ready[s], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->stage_free[s], hipEventDisableTiming));
    }
  }
  if (need > c->stage_elems[slot]) {
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->stage[slot]) HIPCHK(hipFree(c->stage[slot]));
    c->stage[slot] = nullptr;
    c->stage_elems[slot] = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->stage[slot]), need * sizeof(T)));
    c->stage_elems[slot] = need;
    c->stage_busy[slot] = false;
  } else if (c->stage_busy[slot]) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->stage_free[slot], 0));
    c->stage_busy[slot] = false;
  }
  return AHMC_OK;
}

// the chain-independent part of adapt!'s state after one more adapting transition (what the kernel's `schedule` mirrors)
template <class T>
int advance_adapt_host(Ctx<T>* c, bool has_mm, bool pooled) {
  if (c->adapt_kind == AHMC_ADAPT_STAN) {
    c->stan_i += 1;
    const bool in_window = c->stan_i >= c->windows.window_start && c->stan_i <= c->windows.window_end;
    const bool window_end = std::find(c->windows.splits.begin(), c->windows.splits.end(), c->stan_i) != c->windows.splits.end();
    if (in_window && has_mm) c->wv_n += 1;
    if (window_end && has_mm) {
      if (pooled) {  // (ahmc_sample ends a pooled batch at every split, so this is the batch's last transition)
        if (in_window && c->wv_n >= c->wv_nmin) { int rc2 = pooled_update(c); if (rc2) return rc2; }
        HIPCHK(hipMemsetAsync(c->wv_mu, 0, sizeof(T) * c->D * c->N, c->stream));
        HIPCHK(hipMemsetAsync(c->wv_M, 0, sizeof(T) * c->D * c->N, c->stream));
      }
      c->wv_n = 0;
    }
  } else if (has_mm) {
    c->wv_n += 1;
  }
  return AHMC_OK;
}

template <class T>
AdaptK<T> make_adaptk(Ctx<T>* c, int64_t i, int64_t n_adapts, bool has_ss, bool has_mm, bool pooled) {
  AdaptK<T> a;
  memset(&a, 0, sizeof(a));
  a.kind = c->adapt_kind;
  a.has_ss = has_ss ? 1 : 0;
  a.has_mm = has_mm ? 1 : 0;
  a.nutpie = c->var_estimator == AHMC_VAR_NUTPIE ? 1 : 0;
  a.pooled = pooled ? 1 : 0;
  a.i0 = i - 1;
  a.n_adapts = n_adapts;
  a.stan_i0 = c->stan_i;
  a.window_start = c->windows.window_start;
  a.window_end = c->windows.window_end;
  a.n_splits = (int)std::min<size_t>(c->windows.splits.size(), 24);
  for (int s2 = 0; s2 < a.n_splits; ++s2) a.splits[s2] = c->windows.splits[(size_t)s2];
  a.wv_n0 = c->wv_n;
  a.wv_nmin = c->wv_nmin;
  a.delta = (T)c->da_delta; a.gamma = T(DA_GAMMA); a.t0 = T(DA_T0); a.kappa = T(DA_KAPPA);  // stepsize.jl:168-172
  a.da_m = c->da_m; a.da_eps = c->da_eps; a.da_mu = c->da_mu; a.da_xbar = c->da_xbar; a.da_Hbar = c->da_Hbar;
  a.wv_mu = c->wv_mu; a.wv_M = c->wv_M; a.wv_var = c->wv_var; a.wg_mu = c->wg_mu; a.wg_M = c->wg_M;
  a.minv = c->minv; a.sqrt_minv = c->sqrt_minv; a.eps_nom = c->eps_nom;
  a.da_tab = c->da_tab;
  return a;
}

template <class T>
int nuts_adapt_batch(Ctx<T>* c, const ahmc_kernel_cfg* cfg, int k, int64_t i, int64_t n_adapts, bool accum, T* samples_dev) {
  const bool has_ss = c->adapt_kind != AHMC_ADAPT_MASSMATRIX;
  const bool has_mm = c->adapt_kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DIAG;
  if (c->adapt_kind == AHMC_ADAPT_STAN && (i == 1 || c->windows_n_adapts != n_adapts)) {  // initialize!
    c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, n_adapts);
    c->windows_n_adapts = n_adapts;
  }
  const bool pooled = has_mm && c->var_estimator == AHMC_VAR_POOLED;
  const AdaptK<T> a = make_adaptk(c, i, n_adapts, has_ss, has_mm, pooled);
  int rc = nuts_transition(c, cfg->max_depth, cfg->delta_max, cfg->criterion, cfg->sampler, cfg->refresh_alpha, accum, k, samples_dev, &a);
  if (rc) return rc;
  for (int kt = 0; kt < k; ++kt) {  // the chain-independent part of adapt!'s state
    rc = advance_adapt_host(c, has_mm, pooled);
    if (rc) return rc;
  }
  if (has_ss) { c->eps_scalar = false; c->order_valid = false; c->sched = {}; }
  if (i + k - 1 >= n_adapts) c->adapting = false;
  return AHMC_OK;
}

}  // namespace

// roctx ranges around every C-ABI call: with AHMC_ROCTX=1 the library loads roctx (rocprofiler-sdk's,
// else roctracer's) and each entry point shows up by name in `rocprofv3 --marker-trace` timelines next to the kernels
// it enqueued.  Off by default: one predictable branch per call.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    const char* on = getenv("AHMC_ROCTX");
    if (!on || atoi(on) == 0) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
inline RoctxApi& roctx_api() { static RoctxApi a; return a; }
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name); }
  ~RoctxRange() { if (on) roctx_api().pop(); }
};

#define FOR_CTX(ctx, ...)                                                                              \
  do {                                                                                                 \
    CtxBase* _b = reinterpret_cast<CtxBase*>(ctx);                                                     \
    if (!_b) return AHMC_ERR_ARGUMENT;                                                                 \
    RoctxRange _roctx(__func__);                                                                       \
    (void)hipSetDevice(_b->device);                                                                    \
    if (_b->dtype == AHMC_F32) { using T = float; auto* c = static_cast<Ctx<T>*>(_b); __VA_ARGS__ }    \
    else { using T = double; auto* c = static_cast<Ctx<T>*>(_b); __VA_ARGS__ }                         \
  } while (0)

// entry points that change the context: refused while an ahmc_ext_* run is in progress
#define FOR_CTX_MUT(ctx, ...)                                                                                        \
  FOR_CTX(ctx, {                                                                                                     \
    if (c->ext.mode != EXT_IDLE)                                                                                     \
      return fail(c, AHMC_ERR_STATE, std::string(__func__) + ": an ahmc_ext_* run is in progress (finish it or call ahmc_ext_cancel)"); \
    __VA_ARGS__                                                                                                      \
  })

template <class T>
static int32_t create_impl(int32_t device, int32_t dtype, int64_t D, int64_t N, void* stream, ahmc_ctx** out) {
  auto* c = new Ctx<T>();
  c->dtype = dtype;
  c->device = device;
  c->D = D;
  c->N = N;
  auto bail = [&](const std::string& m, int code) {
    g_create_err = m;
    delete c;
    return code;
  };
  if (!pick_geometry(D, c->G, c->E))
    return bail("ahmc_create: D > 4096 has no HIP kernel geometry", AHMC_ERR_UNSUPPORTED);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(e), AHMC_ERR_RUNTIME);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return bail(std::string("hipStreamCreate: ") + hipGetErrorString(e), AHMC_ERR_RUNTIME);
    c->own_stream = true;
  }
  const size_t DN = (size_t)D * N, n = (size_t)N;
  bool ok = true;
  auto A = [&](auto** p, size_t cnt) {
    if (ok && hipMalloc(reinterpret_cast<void**>(p), cnt * sizeof(**p)) != hipSuccess) ok = false;
  };
  // four slabs (see KP in ahmc_kernels.hpp); the per-field pointers below are aliases into them
  A(&c->vbase, 5 * DN); A(&c->tbase, 14 * n); A(&c->ibase, 4 * n); A(&c->lbase, 2 * n);
  if (ok) {
    c->th = c->vbase; c->r = c->vbase + DN; c->g = c->vbase + 2 * DN; c->acc_sum = c->vbase + 3 * DN; c->acc_sumsq = c->vbase + 4 * DN;
    c->lp = c->tbase; c->lk = c->tbase + n; c->eps_nom = c->tbase + 2 * n; c->eps_cur = c->tbase + 3 * n;
    c->st_accrate = c->tbase + 4 * n; c->st_logdens = c->tbase + 5 * n; c->st_H = c->tbase + 6 * n; c->st_Herr = c->tbase + 7 * n;
    c->st_maxHerr = c->tbase + 8 * n;
    c->acc_energy = c->tbase + 9 * n;
    c->st_nsteps = c->ibase; c->st_accept = c->ibase + n; c->st_depth = c->ibase + 2 * n; c->st_numerr = c->ibase + 3 * n;
    c->acc_nsteps = c->lbase; c->acc_ndiv = c->lbase + n;
  }
  A(&c->order, n);
  A(&c->redo, n);
  if (!ok) return bail("ahmc_create: hipMalloc failed (out of device memory?)", AHMC_ERR_RUNTIME);
  (void)hipMemsetAsync(c->th, 0, sizeof(T) * DN, c->stream);
  (void)hipMemsetAsync(c->r, 0, sizeof(T) * DN, c->stream);
  (void)hipMemsetAsync(c->g, 0, sizeof(T) * DN, c->stream);
  (void)hipMemsetAsync(c->st_nsteps, 0, sizeof(int32_t) * n, c->stream);
  (void)hipMemsetAsync(c->redo, 0, sizeof(int32_t) * n, c->stream);
  (void)hipMemsetAsync(c->st_accept, 0, sizeof(int32_t) * n, c->stream);
  (void)hipMemsetAsync(c->st_depth, 0, sizeof(int32_t) * n, c->stream);
  (void)hipMemsetAsync(c->st_numerr, 0, sizeof(int32_t) * n, c->stream);
  T* fl[] = {c->st_accrate, c->st_logdens, c->st_H, c->st_Herr, c->st_maxHerr, c->lp, c->lk};
  for (T* f : fl) (void)hipMemsetAsync(f, 0, sizeof(T) * n, c->stream);
  hipLaunchKernelGGL((k_fill<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->eps_nom, T(0.1), (int64_t)n);
  hipLaunchKernelGGL((k_fill<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->eps_cur, T(0.1), (int64_t)n);
  (void)hipMemsetAsync(c->acc_nsteps, 0, sizeof(long long) * n, c->stream);
  (void)hipMemsetAsync(c->acc_ndiv, 0, sizeof(long long) * n, c->stream);
  (void)hipMemsetAsync(c->acc_sum, 0, sizeof(T) * DN, c->stream);
  (void)hipMemsetAsync(c->acc_sumsq, 0, sizeof(T) * DN, c->stream);
  e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return bail(std::string("ahmc_create: ") + hipGetErrorString(e), AHMC_ERR_RUNTIME);
  *out = reinterpret_cast<ahmc_ctx*>(static_cast<CtxBase*>(c));
  return AHMC_OK;
}

extern "C" {

int32_t ahmc_abi_version(void) { return AHMC_ABI_VERSION; }
const char* ahmc_backend(void) { return "hip:gfx950"; }

int32_t ahmc_create(int32_t device, int32_t dtype, int64_t D, int64_t N, void* stream, ahmc_ctx** out) {
  if (!out) { g_create_err = "ahmc_create: out is NULL"; return AHMC_ERR_ARGUMENT; }
  if (D < 1 || N < 1) { g_create_err = "ahmc_create: D and N must be >= 1"; return AHMC_ERR_ARGUMENT; }
  if (dtype == AHMC_F32) return create_impl<float>(device, dtype, D, N, stream, out);
  if (dtype == AHMC_F64) return create_impl<double>(device, dtype, D, N, stream, out);
  g_create_err = "ahmc_create: dtype must be AHMC_F32 or AHMC_F64";
  return AHMC_ERR_ARGUMENT;
}

int32_t ahmc_destroy(ahmc_ctx* ctx) {
  delete reinterpret_cast<CtxBase*>(ctx);
  return AHMC_OK;
}

const char* ahmc_last_error(const ahmc_ctx* ctx) {
  if (!ctx) return g_create_err.c_str();
  return reinterpret_cast<const CtxBase*>(ctx)->err.c_str();
}

int32_t ahmc_sync(ahmc_ctx* ctx) {
  FOR_CTX(ctx, { HIPCHK(hipStreamSynchronize(c->stream)); return AHMC_OK; });
}

void* ahmc_stream(ahmc_ctx* ctx) {
  CtxBase* b = reinterpret_cast<CtxBase*>(ctx);
  return b ? reinterpret_cast<void*>(b->stream) : nullptr;
}

int32_t ahmc_set_target(ahmc_ctx* ctx, int32_t kind, const void* params, int64_t n_params) {
  FOR_CTX_MUT(ctx, {
    int64_t need = 0;
    if (kind == AHMC_TARGET_PLUGIN || kind == AHMC_TARGET_KERNEL)
      return fail(c, AHMC_ERR_ARGUMENT, "set_target: use ahmc_set_target_plugin / ahmc_set_target_kernel");
    switch (kind) {
      case AHMC_TARGET_ISO_GAUSS: case AHMC_TARGET_FUNNEL: case AHMC_TARGET_HIER_GAUSS: case AHMC_TARGET_EXTERNAL: need = 0; break;
      case AHMC_TARGET_DIAG_GAUSS: need = 2 * c->D; break;
      case AHMC_TARGET_DENSE_GAUSS: need = c->D * c->D; break;
      default: return fail(c, AHMC_ERR_ARGUMENT, "set_target: unknown target kind");
    }
    if (kind == AHMC_TARGET_FUNNEL && c->D < 2) return fail(c, AHMC_ERR_ARGUMENT, "funnel needs D >= 2");
    if (kind == AHMC_TARGET_HIER_GAUSS && c->D < 3) return fail(c, AHMC_ERR_ARGUMENT, "hier_gauss needs D >= 3");
    if (n_params != need || (need > 0 && !params)) return fail(c, AHMC_ERR_ARGUMENT, "set_target: wrong parameter count for this family");
    // the new parameters go into a buffer of their own and replace the old ones only once they are on the device: a failure
    // on the way leaves the context with its previous target intact
    T* np_dev = nullptr;
    if (need > 0) {
      int rc = dev_alloc(c, &np_dev, (size_t)need);
      if (rc) return rc;
      if (hipMemcpyAsync(np_dev, params, sizeof(T) * need, hipMemcpyDefault, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(np_dev);
        return fail(c, AHMC_ERR_RUNTIME, std::string("set_target: copying the parameters failed: ") + hipGetErrorString(hipGetLastError()));
      }
    } else {
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (c->tparams) (void)hipFree(c->tparams);
    c->tparams = np_dev;
    c->target_kind = kind;
    c->have_point = false;
    c->order_valid = false; c->sched = {};   // (a dispatch order and a launch length measured on another density say nothing about this one)
    return dn_refresh_fused(c);
  });
}

#ifndef AHMC_KERNEL_SOURCES_DIGEST
#define AHMC_KERNEL_SOURCES_DIGEST ""
#endif
int32_t ahmc_set_target_plugin(ahmc_ctx* ctx, const char* plugin_so, const void* params, int64_t n_params) {
  FOR_CTX_MUT(ctx, {
    if (!plugin_so) return fail(c, AHMC_ERR_ARGUMENT, "set_target_plugin: path is NULL");
    if (n_params < 0 || (n_params > 0 && !params)) return fail(c, AHMC_ERR_ARGUMENT, "set_target_plugin: bad parameter array");
    void* dl = dlopen(plugin_so, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return fail(c, AHMC_ERR_ARGUMENT, std::string("set_target_plugin: cannot load ") + plugin_so + ": " + dlerror());
    const TargetPluginDesc* d = static_cast<const TargetPluginDesc*>(dlsym(dl, "ahmc_target_plugin_v1"));
    auto reject = [&](const std::string& why) {
      dlclose(dl);
      return fail(c, AHMC_ERR_ARGUMENT, std::string("set_target_plugin: ") + plugin_so + ": " + why);
    };
    if (!d) return reject("no symbol ahmc_target_plugin_v1 (not a target plugin)");
    if (d->plugin_abi != AHMC_PLUGIN_ABI || d->struct_bytes != (int32_t)sizeof(TargetPluginDesc) || d->kp_bytes != (int32_t)sizeof(KP<T>))
      return reject("built against another version of the engine's kernel sources: rebuild it (build_target_plugin)");
    // the kernel sources themselves (round 5): the scratch layout of k_nuts is a contract between the launch plan here and the kernels there that no
    // struct size shows — a plugin compiled from other sources than this library is refused (an empty digest on either side: not checked)
    if (d->sources_digest && d->sources_digest[0] && AHMC_KERNEL_SOURCES_DIGEST[0] && strcmp(d->sources_digest, AHMC_KERNEL_SOURCES_DIGEST) != 0)
      return reject(std::string("compiled from other kernel sources (") + std::string(d->sources_digest).substr(0, 12) + "…) than this library (" +
                    std::string(AHMC_KERNEL_SOURCES_DIGEST).substr(0, 12) + "…): rebuild it (build_target_plugin)");
    if (d->dtype != (sizeof(T) == 4 ? AHMC_F32 : AHMC_F64)) return reject("built for the other element type");
    {  // (checked whatever the digests say — an empty one skips that check: -D flags can change the layout without changing a source)
      const TargetOps<T>* po = static_cast<const TargetOps<T>*>(d->ops);
      if (!po || !po->scratch_layout || po->scratch_layout() != nuts_scratch_layout())
        return reject("its kernels index another k_nuts scratch layout (AHMC_CKPT / NUTS_* constants) than this library allocates: rebuild it");
    }
    if (d->G != c->G || d->E != c->E)
      return reject("built for thread geometry (" + std::to_string(d->G) + "," + std::to_string(d->E) + "), the context uses (" + std::to_string(c->G) + "," +
                    std::to_string(c->E) + ")");
    if (d->n_params >= 0 && d->n_params != n_params) return reject("expects " + std::to_string(d->n_params) + " parameters");
    // (a HIP error from here on must not leave the library loaded: the checks close it before they return)
    auto hip_ok = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return true;
      c->err = std::string(what) + ": " + hipGetErrorString(e);
      dlclose(dl);
      return false;
    };
    if (!hip_ok(hipStreamSynchronize(c->stream), "set_target_plugin: hipStreamSynchronize")) return AHMC_ERR_RUNTIME;
    T* np_dev = nullptr;  // (committed below, after everything that can fail: the previous target stays whole until then)
    if (n_params > 0) {
      int rc = dev_alloc(c, &np_dev, (size_t)n_params);
      if (rc) { dlclose(dl); return rc; }
      if (!hip_ok(hipMemcpyAsync(np_dev, params, sizeof(T) * n_params, hipMemcpyDefault, c->stream), "set_target_plugin: hipMemcpyAsync") ||
          !hip_ok(hipStreamSynchronize(c->stream), "set_target_plugin: hipStreamSynchronize")) {
        (void)hipFree(np_dev);
        return AHMC_ERR_RUNTIME;
      }
    }
    if (c->tparams) (void)hipFree(c->tparams);
    c->tparams = np_dev;
    if (c->plugin_dl) dlclose(c->plugin_dl);  // (the stream is idle: nothing of the previous plugin's code is running)
    c->plugin_dl = dl;
    c->plugin_ops = static_cast<const TargetOps<T>*>(d->ops);
    c->target_kind = AHMC_TARGET_PLUGIN;
    c->have_point = false;
    c->order_valid = false; c->sched = {};
    return dn_refresh_fused(c);
  });
}

int32_t ahmc_set_target_kernel(ahmc_ctx* ctx, int32_t handle_kind, void* handle, int32_t block_threads, int32_t chains_per_block, void* user) {
  FOR_CTX_MUT(ctx, {
    if (handle_kind == AHMC_KERNEL_HOST) return fail(c, AHMC_ERR_UNSUPPORTED, "set_target_kernel: a host function is the CPU checker's form; the HIP engine takes device kernels");
    if (handle_kind != AHMC_KERNEL_HIP_FUNCTION && handle_kind != AHMC_KERNEL_HIP_SYMBOL) return fail(c, AHMC_ERR_ARGUMENT, "set_target_kernel: unknown handle kind");
    if (!handle) return fail(c, AHMC_ERR_ARGUMENT, "set_target_kernel: handle is NULL");
    if (block_threads < 1 || block_threads > 1024 || chains_per_block < 1) return fail(c, AHMC_ERR_ARGUMENT, "set_target_kernel: block_threads in 1..1024, chains_per_block >= 1");
    if (c->tparams) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->tparams)); c->tparams = nullptr; }
    c->uk_kind = handle_kind;
    c->uk_handle = handle;
    c->uk_block = block_threads;
    c->uk_cpb = chains_per_block;
    c->uk_user = user;
    c->target_kind = AHMC_TARGET_KERNEL;
    c->have_point = false;
    c->order_valid = false; c->sched = {};
    return dn_refresh_fused(c);
  });
}

int32_t ahmc_set_metric(ahmc_ctx* ctx, int32_t kind, const void* Minv, int64_t n) {
  FOR_CTX_MUT(ctx, {
    c->order_valid = false; c->sched = {};   // (tree sizes follow the metric: the measured order and launch length are another sampler's)
    return set_metric(c, kind, static_cast<const T*>(Minv), n);
  });
}

int32_t ahmc_get_metric(ahmc_ctx* ctx, void* out, int64_t n) {
  FOR_CTX(ctx, {
    if (c->metric_kind == AHMC_METRIC_UNIT) return fail(c, AHMC_ERR_ARGUMENT, "get_metric: unit metric has no array");
    if (n != c->minv_n) return fail(c, AHMC_ERR_ARGUMENT, "get_metric: size mismatch");
    HIPCHK(hipMemcpyAsync(out, c->metric_kind == AHMC_METRIC_DENSE ? c->dn_minv : c->minv, sizeof(T) * n, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_set_stepsize(ahmc_ctx* ctx, const void* eps, int64_t n) {
  FOR_CTX_MUT(ctx, {
    if (!eps || (n != 1 && n != c->N)) return fail(c, AHMC_ERR_ARGUMENT, "set_stepsize: need 1 or N step sizes");
    if (n == 1) {
      T v;
      HIPCHK(hipMemcpy(&v, eps, sizeof(T), hipMemcpyDefault));
      hipLaunchKernelGGL((k_fill<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->eps_nom, v, c->N);
      HIPCHK(hipGetLastError());
      c->eps_scalar_value = (double)v;
    } else {
      HIPCHK(hipMemcpyAsync(c->eps_nom, eps, sizeof(T) * n, hipMemcpyDefault, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->eps_cur, c->eps_nom, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->eps_scalar = (n == 1);
    c->order_valid = false; c->sched = {};
    return AHMC_OK;
  });
}

int32_t ahmc_get_stepsize(ahmc_ctx* ctx, void* out) {
  FOR_CTX(ctx, {
    HIPCHK(hipMemcpyAsync(out, c->eps_nom, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_set_integrator(ahmc_ctx* ctx, int32_t kind, double param) {
  FOR_CTX_MUT(ctx, {
    if (kind < AHMC_INTEGRATOR_LEAPFROG || kind > AHMC_INTEGRATOR_TEMPERED) return fail(c, AHMC_ERR_ARGUMENT, "set_integrator: unknown kind");
    c->integ_kind = kind;
    c->integ_param = param;
    return AHMC_OK;
  });
}

int32_t ahmc_seed(ahmc_ctx* ctx, uint64_t seed, uint64_t chain_offset, uint64_t chain_stride, uint64_t iteration) {
  FOR_CTX_MUT(ctx, {
    c->seed = seed; c->chain_offset = chain_offset; c->chain_stride = chain_stride; c->iteration = iteration;
    return AHMC_OK;
  });
}

int32_t ahmc_set_position(ahmc_ctx* ctx, const void* theta, const void* r) {
  FOR_CTX_MUT(ctx, {
    if (!theta) return fail(c, AHMC_ERR_ARGUMENT, "set_position: theta is NULL");
    int rc = dense_engine(c) ? dn_check(c, "set_position", 0) : check_builtin(c, "set_position");
    if (rc) return rc;
    const size_t nb = sizeof(T) * c->D * c->N;
    HIPCHK(hipMemcpyAsync(c->th, theta, nb, hipMemcpyDefault, c->stream));
    if (r) HIPCHK(hipMemcpyAsync(c->r, r, nb, hipMemcpyDefault, c->stream));
    else HIPCHK(hipMemsetAsync(c->r, 0, nb, c->stream));
    rc = launch_fill_caches(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_point = true;
    return AHMC_OK;
  });
}

int32_t ahmc_set_phasepoint(ahmc_ctx* ctx, const void* theta, const void* r, const void* lp, const void* grad) {
  FOR_CTX_MUT(ctx, {
    if (!theta || !r || !lp || !grad) return fail(c, AHMC_ERR_ARGUMENT, "set_phasepoint: NULL argument");
    const size_t nb = sizeof(T) * c->D * c->N;
    HIPCHK(hipMemcpyAsync(c->th, theta, nb, hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(c->r, r, nb, hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(c->g, grad, nb, hipMemcpyDefault, c->stream));
    HIPCHK(hipMemcpyAsync(c->lp, lp, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    int rc = launch_kinetic(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_point = true;
    return AHMC_OK;
  });
}

int32_t ahmc_get_phasepoint(ahmc_ctx* ctx, void* theta, void* r, void* lp, void* grad, void* lk) {
  FOR_CTX(ctx, {
    const size_t nb = sizeof(T) * c->D * c->N;
    if (theta) HIPCHK(hipMemcpyAsync(theta, c->th, nb, hipMemcpyDefault, c->stream));
    if (r) HIPCHK(hipMemcpyAsync(r, c->r, nb, hipMemcpyDefault, c->stream));
    if (grad) HIPCHK(hipMemcpyAsync(grad, c->g, nb, hipMemcpyDefault, c->stream));
    if (lp) HIPCHK(hipMemcpyAsync(lp, c->lp, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    if (lk) HIPCHK(hipMemcpyAsync(lk, c->lk, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_refresh_momentum(ahmc_ctx* ctx, double alpha) {
  FOR_CTX_MUT(ctx, {
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "refresh before set_position");
    if (dense_engine(c)) return dn_refresh(c, alpha);
    int rc = check_builtin(c, "refresh_momentum");
    if (rc) return rc;
    KP<T> p = make_kp(c);
    p.refresh_alpha = (T)alpha;
    if (const TargetOps<T>* o = ops_for(c)) o->refresh(c->G, c->E, group_grid(c), c->stream, p);
    HIPCHK(hipGetLastError());
    return AHMC_OK;
  });
}

int32_t ahmc_leapfrog(ahmc_ctx* ctx, int64_t n_steps) {
  FOR_CTX_MUT(ctx, {
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "leapfrog before set_position");
    if (dense_engine(c)) {
      if (c->ref_compat) return fail(c, AHMC_ERR_UNSUPPORTED, "ahmc_set_ref_compat is not implemented on the dense engine");
      return dn_leapfrog(c, n_steps);
    }
    int rc = check_builtin(c, "leapfrog");
    if (rc) return rc;
    if (c->ref_compat) {   // Q1, opt-in: all chains stop after the first step that left any chain non-finite
      int64_t done = 0;
      return compat_step_loop(c, n_steps < 0 ? -n_steps : n_steps, n_steps > 0, done);
    }
    KP<T> p = make_kp(c);
    p.n_steps = n_steps;
    if (const TargetOps<T>* o = ops_for(c)) o->leapfrog(c->G, c->E, group_grid(c), c->stream, p);
    HIPCHK(hipGetLastError());
    return AHMC_OK;
  });
}

int32_t ahmc_lf_pre(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps) {
  FOR_CTX_MUT(ctx, {
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "lf_pre before set_phasepoint");
    if (c->metric_kind == AHMC_METRIC_DENSE) return fail(c, AHMC_ERR_UNSUPPORTED, "lf_pre/lf_post are not implemented for DenseEuclideanMetric");
    KP<T> p = make_kp(c);
    const int64_t DN = c->D * c->N;
    hipLaunchKernelGGL((k_lf_pre<T>), dim3((unsigned)((DN + 255) / 256)), dim3(256), 0, c->stream, p, (int)fwd, i, n_steps);
    HIPCHK(hipGetLastError());
    return AHMC_OK;
  });
}

int32_t ahmc_lf_post(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps, const void* lp, const void* grad_neg) {
  FOR_CTX_MUT(ctx, {
    if (!lp || !grad_neg) return fail(c, AHMC_ERR_ARGUMENT, "lf_post: NULL argument");
    const int64_t DN = c->D * c->N;
    // the caller's arrays may live on the host or on the device: host arrays go through the context's persistent
    // staging buffers (shared with ahmc_ext_advance; no allocation per step), and the call then returns only after
    // the stream has consumed them — the caller may reuse (or free) its host arrays as soon as this returns
    auto on_device = [&](const void* ptr) {
      hipPointerAttribute_t at;
      const bool dev = hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice;
      (void)hipGetLastError();
      return dev;
    };
    const bool g_dev = on_device(grad_neg), lp_dev = on_device(lp);
    if ((!g_dev || !lp_dev) && !c->ext_gstage) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->ext_gstage), sizeof(T) * DN));
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->ext_lpstage), sizeof(T) * c->N));
    }
    const T* gsrc = static_cast<const T*>(grad_neg);
    const T* lsrc = static_cast<const T*>(lp);
    if (!g_dev) {
      HIPCHK(hipMemcpyAsync(c->ext_gstage, grad_neg, sizeof(T) * DN, hipMemcpyDefault, c->stream));
      gsrc = c->ext_gstage;
    }
    if (!lp_dev) {
      HIPCHK(hipMemcpyAsync(c->ext_lpstage, lp, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
      lsrc = c->ext_lpstage;
    }
    HIPCHK(hipMemcpyAsync(c->lp, lsrc, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
    KP<T> p = make_kp(c);
    hipLaunchKernelGGL((k_lf_post<T>), dim3((unsigned)((DN + 255) / 256)), dim3(256), 0, c->stream, p, (int)fwd, i, n_steps, gsrc);
    HIPCHK(hipGetLastError());
    int rc = launch_kinetic(c);
    if (!g_dev || !lp_dev) HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
  });
}

void* ahmc_theta_ptr(ahmc_ctx* ctx) {
  CtxBase* b = reinterpret_cast<CtxBase*>(ctx);
  if (!b) return nullptr;
  if (b->dtype == AHMC_F32) return static_cast<Ctx<float>*>(b)->th;
  return static_cast<Ctx<double>*>(b)->th;
}

int32_t ahmc_hmc_transition(ahmc_ctx* ctx, int64_t L, double lambda, int32_t sampler) {
  FOR_CTX_MUT(ctx, { return hmc_transition(c, L, lambda, sampler, 0.0, false); });
}

int32_t ahmc_nuts_transition(ahmc_ctx* ctx, int32_t max_depth, double delta_max, int32_t criterion, int32_t sampler) {
  FOR_CTX_MUT(ctx, { return nuts_transition(c, max_depth, delta_max, criterion, sampler, 0.0, false); });
}

int32_t ahmc_get_stat(ahmc_ctx* ctx, int32_t field, void* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "get_stat: out is NULL");
    const void* src = nullptr;
    size_t nb = sizeof(T) * c->N;
    switch (field) {
      case AHMC_STAT_N_STEPS: src = c->st_nsteps; nb = sizeof(int32_t) * c->N; break;
      case AHMC_STAT_IS_ACCEPT: src = c->st_accept; nb = sizeof(int32_t) * c->N; break;
      case AHMC_STAT_ACCEPTANCE_RATE: src = c->st_accrate; break;
      case AHMC_STAT_LOG_DENSITY: src = c->st_logdens; break;
      case AHMC_STAT_HAMILTONIAN_ENERGY: src = c->st_H; break;
      case AHMC_STAT_HAMILTONIAN_ENERGY_ERROR: src = c->st_Herr; break;
      case AHMC_STAT_MAX_HAMILTONIAN_ENERGY_ERROR: src = c->st_maxHerr; break;
      case AHMC_STAT_TREE_DEPTH: src = c->st_depth; nb = sizeof(int32_t) * c->N; break;
      case AHMC_STAT_NUMERICAL_ERROR: src = c->st_numerr; nb = sizeof(int32_t) * c->N; break;
      case AHMC_STAT_STEP_SIZE: src = c->eps_cur; break;
      case AHMC_STAT_NOM_STEP_SIZE: src = c->eps_nom; break;
      default: return fail(c, AHMC_ERR_ARGUMENT, "get_stat: unknown field");
    }
    HIPCHK(hipMemcpyAsync(out, src, nb, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_find_good_stepsize(ahmc_ctx* ctx, double initial_step_size, int32_t max_n_iters) {
  FOR_CTX_MUT(ctx, {
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "find_good_stepsize before set_position");
    if (dense_engine(c)) return dn_find_eps(c, initial_step_size, max_n_iters);
    int rc = check_builtin(c, "find_good_stepsize");
    if (rc) return rc;
    KP<T> p = make_kp(c);
    p.init_eps = (T)initial_step_size;
    p.max_iters = max_n_iters;
    T* out = c->eps_cur;
    if (const TargetOps<T>* o = ops_for(c)) o->find_eps(c->G, c->E, group_grid(c), c->stream, p, out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->eps_nom, c->eps_cur, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
    c->eps_scalar = false;
    c->order_valid = false; c->sched = {};
    return AHMC_OK;
  });
}

int32_t ahmc_adaptor_init(ahmc_ctx* ctx, int32_t kind, double delta, int32_t init_buffer, int32_t term_buffer, int32_t window_size) {
  FOR_CTX_MUT(ctx, {
    if (kind < AHMC_ADAPT_NONE || kind > AHMC_ADAPT_STAN) return fail(c, AHMC_ERR_ARGUMENT, "adaptor_init: unknown adaptor kind");
    int rc = adaptor_init(c, kind, delta, init_buffer, term_buffer, window_size);
    if (rc) return rc;
    return AHMC_OK;
  });
}

int32_t ahmc_adapt_point(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta, const void* grad, const void* alpha) {
  FOR_CTX_MUT(ctx, { return adapt(c, i, n_adapts, static_cast<const T*>(theta), static_cast<const T*>(alpha), static_cast<const T*>(grad)); });
}

int32_t ahmc_set_var_estimator(ahmc_ctx* ctx, int32_t est) {
  FOR_CTX_MUT(ctx, {
    if (est != AHMC_VAR_WELFORD && est != AHMC_VAR_NUTPIE && est != AHMC_VAR_POOLED) return fail(c, AHMC_ERR_ARGUMENT, "set_var_estimator: unknown estimator");
    c->var_estimator = est;
    return AHMC_OK;
  });
}

int32_t ahmc_adapt(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta, const void* alpha) {
  FOR_CTX_MUT(ctx, { return adapt(c, i, n_adapts, static_cast<const T*>(theta), static_cast<const T*>(alpha)); });
}

int32_t ahmc_stan_windows(int32_t init_buffer, int32_t term_buffer, int32_t window_size, int64_t n_adapts,
                          int64_t* window_start, int64_t* window_end, int64_t* splits, int32_t cap, int32_t* n_splits) {
  StanWindows w = stan_windows(init_buffer, term_buffer, window_size, n_adapts);
  if (window_start) *window_start = w.window_start;
  if (window_end) *window_end = w.window_end;
  if (n_splits) *n_splits = (int32_t)w.splits.size();
  if (splits)
    for (int32_t k = 0; k < cap && k < (int32_t)w.splits.size(); ++k) splits[k] = w.splits[k];
  return AHMC_OK;
}

// (one internal body, reached without the PLT: a process may hold the HIP engine AND the CPU checker, both exporting
// these names — a call from one exported function to another could bind to the other library's)
static int32_t sample_from_impl(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                                void* samples_out);

int32_t ahmc_sample_reserve(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples) {
  FOR_CTX_MUT(ctx, {
    if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "sample_reserve: cfg is NULL");
    if (n_samples < 1 || !cfg->nuts || dense_engine(c) || c->target_kind == AHMC_TARGET_EXTERNAL) return AHMC_OK;  // nothing to reserve ahead of time
    return reserve_normals(c, std::min<int64_t>(nuts_batch(c), n_samples));
  });
}

int32_t ahmc_set_ref_compat(ahmc_ctx* ctx, int32_t on) {
  FOR_CTX_MUT(ctx, {
    c->ref_compat = on != 0;
    return AHMC_OK;
  });
}

int32_t ahmc_sample(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup, void* samples_out) {
  return sample_from_impl(ctx, cfg, 1, n_samples, n_adapts, drop_warmup, samples_out);
}

int32_t ahmc_sample_from(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                         void* samples_out) {
  return sample_from_impl(ctx, cfg, i_first, n_samples, n_adapts, drop_warmup, samples_out);
}

static int32_t sample_from_impl(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                                void* samples_out) {
  FOR_CTX_MUT(ctx, {
    if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "sample: cfg is NULL");
    if (i_first < 1) return fail(c, AHMC_ERR_ARGUMENT, "sample_from: i_first must be >= 1");
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "sample before set_position");
    if (drop_warmup && c->adapt_kind == AHMC_ADAPT_NONE)
      return fail(c, AHMC_ERR_ARGUMENT, "Cannot drop warmup samples if there is no adaptation phase.");  // src/sampler.jl:172
    // a resumed run must continue where the restored state stopped: a Stan adaptor counts its own calls (state.i,
    // stan_adaptor.jl:137-159), so while it is adapting the absolute iteration is known and a mismatch is an error rather
    // than a silently wrong window schedule
    if (i_first > 1 && c->adapt_kind == AHMC_ADAPT_STAN && c->adapting && i_first <= n_adapts && c->stan_i != i_first - 1)
      return fail(c, AHMC_ERR_STATE, "sample_from: i_first = " + std::to_string(i_first) + " but the adaptor has seen " + std::to_string(c->stan_i) +
                                         " iterations (restore the checkpoint taken after iteration i_first - 1: ahmc_set_adaptor_state)");
    c->sched.g_left = 0;  // a timed group of launches never spans two calls (the host time between them would be in its interval)
    T* so = static_cast<T*>(samples_out);
    // the accumulators are reset at the first kept transition — unless the run is being RESUMED beyond it (ahmc_sample_from):
    // then they continue (a checkpoint carries them: ahmc_get/set_accum_state)
    bool reset_done = i_first > (drop_warmup ? n_adapts + 1 : 1);
    const size_t nb = sizeof(T) * c->D * c->N;
    // can k_nuts write the kept draws itself?  (device buffer, or none requested)
    bool so_on_device = false;
    if (so) {
      hipPointerAttribute_t at;
      so_on_device = hipPointerGetAttributes(&at, so) == hipSuccess && at.type == hipMemoryTypeDevice;
      (void)hipGetLastError();
    }
    int64_t batch = nuts_batch(c);
    if (cfg->nuts && !dense_engine(c) && c->target_kind != AHMC_TARGET_EXTERNAL && n_samples >= i_first) {
      // the normals of this call's longest launch (lazily, bounded by the call's own length; ahmc_sample_reserve does it ahead of time)
      int rc0 = reserve_normals(c, std::min<int64_t>(batch, n_samples - i_first + 1));
      if (rc0) return rc0;
      if (c->znorm_cap_trans > 0) batch = std::min<int64_t>(batch, c->znorm_cap_trans);
    }
    // Dispatch order by measured work: a better predictor of a chain's tree sizes than its step size is what it
    // actually did — Σ n_steps per chain of the previous sampling call (still in the accumulators here) or of
    // this call's first batch (below).  Counting sort on the stream, no host synchronisation.
    // (Round 3, measured and NOT taken: using the counts of the call before even when the ϵ order has been invalidated — i.e. the
    // warm-up's Σ n_steps for the first launch of the draws, and then for all of them —: cfg2 draws 2.97e9 -> 2.41e9, cfg3
    // 1.70e9 -> 1.58e9 (one launch: a clean comparison; cfg2's figure also contains that its later launches no longer switched to
    // the first launch's counts).  What a chain did while it was still adapting predicts its sampling work worse than its final ϵ.)
    auto order_by_work = [&](bool refresh) -> int {
      if (!cfg->nuts || dense_engine(c) || !c->order_valid || (c->order_from_work && !refresh) || c->acc_ntrans < 4) return AHMC_OK;
      int rc2 = build_order(c, (int)std::min<int64_t>(c->acc_ntrans, 1 << 20));
      if (!rc2) c->order_from_work = true;
      return rc2;
    };
    {
      int rc2 = order_by_work(true);  // the previous call's counts are the freshest estimate there is
      if (rc2) return rc2;
    }
    int64_t n_staged = 0;
    T* pend_dst = nullptr;
    int pend_slot = 0;
    size_t pend_bytes = 0;
    auto flush_pending = [&]() -> int {
      if (!pend_dst) return AHMC_OK;
      HIPCHK(hipStreamWaitEvent(c->copy_stream, c->stage_ready[pend_slot], 0));
      HIPCHK(hipMemcpyAsync(pend_dst, c->stage[pend_slot], pend_bytes, hipMemcpyDeviceToHost, c->copy_stream));
      HIPCHK(hipEventRecord(c->stage_free[pend_slot], c->copy_stream));
      c->stage_busy[pend_slot] = true;
      pend_dst = nullptr;
      return AHMC_OK;
    };
    for (int64_t i = i_first; i <= n_samples;) {  // src/sampler.jl:182-228
      const bool keep = !drop_warmup || i > n_adapts;
      if (keep && !reset_done) {
        int rc0 = reset_accum(c);
        if (rc0) return rc0;
        reset_done = true;
      }
      const bool adapting = c->adapt_kind != AHMC_ADAPT_NONE && i <= n_adapts;
      if (cfg->nuts && !adapting && keep) {
        // chains are independent and nothing is adapted any more: run a batch of transitions per
        // launch (no per-transition barrier; see the note on tree-size tails in ahmc_nuts.hpp)
        // (split the remaining transitions evenly: 50 = 13+13+12+12, not 16+16+16+2 — a short
        // last batch would pay the whole tree-size tail for two transitions)
        // Launch length of the sampling phase (round 4).  Two things pull in opposite directions: a launch cannot end before its
        // slowest wave (the tail is paid once per launch: long launches), and the dispatch order — heaviest chains first, lockstep
        // neighbours with similar trees — is only as good as the prediction of a chain's work, which on heavy-tailed targets is its
        // work in the launch just finished and fades within tens of transitions (short launches).  Measured whole sampling phase,
        // every launch ordered by the work of the one before it: cfg3 (funnel) 250 / 62 / 16 / 8 / 4 per launch 1.83 / 2.06 / 2.42 /
        // 2.56 / 2.73e9 leapfrog/s (one launch of 1 000 ordered by step size: 1.69e9); cfg2 (iso Gaussian) 250 / 64 / 16 / 8 2.94 /
        // 2.89 / 2.80 / 2.58e9.  Neither the imbalance nor the launch-to-launch correlation at one length separates the two cases
        // ahead of time, so the engine MEASURES: starting from 32 it times groups of launches (>= 64 transitions: leapfrogs of the
        // group ÷ wall time, the stream synchronised at both ends — only while it searches), halves while that gains > 2 %; if the
        // first halving does not, the tails decide and it takes the longest launch unless that loses > 1.5 % (then one doubling
        // at a time from the start length).  Then it stays at the best length, asynchronous again (cfg3 settles at 4, cfg2 at 256).  The
        // length is kept until the step sizes change.  AHMC_NUTS_DRAW_BATCH=n fixes it; AHMC_NUTS_SCHED=0 = one length for all
        // (AHMC_INFO_NUTS_BATCH), as before round 4.  The chains do not depend on any of it (tests/test_pipeline_parity.py).
        const int64_t draw_batch_env = getenv("AHMC_NUTS_DRAW_BATCH") ? atoll(getenv("AHMC_NUTS_DRAW_BATCH")) : 0;
        const int sched_env = getenv("AHMC_NUTS_SCHED") ? atoi(getenv("AHMC_NUTS_SCHED")) : 1;
        const char* orf = getenv("AHMC_NUTS_ORDER_REFRESH");
        // the dispatch order of every launch from the work of the launch BEFORE it alone (default since round 4; =0: from the run's totals)
        const bool order_refresh = (orf ? atoi(orf) != 0 : true) && !dense_engine(c) && !c->eps_scalar && !getenv("AHMC_NUTS_NO_ORDER");
        const int64_t left = n_samples - i + 1;
        constexpr int64_t SCHED_MIN = 4, SCHED_START = 32, SCHED_GROUP = 64;
        int64_t k;
        bool probing = false;   // this launch belongs to a group that is being timed
        auto& sc = c->sched;
        if (draw_batch_env <= 0 && sched_env != 0 && order_refresh && sc.phase != 4 && batch >= 2 * SCHED_MIN) {
          if (sc.g_left > 0 && sc.g_len > left) sc.g_left = 0;   // (never a launch longer than what is left; a group an earlier call left
                                                                   // unfinished was dropped at this call's entry)
          if (sc.g_left > 0) {                       // inside a group
            k = sc.g_len; probing = true;
          } else if (!sc.primed && !c->order_from_work && left >= 4 * SCHED_START) {
            k = 2 * SCHED_MIN; sc.primed = true;     // (untimed: the first launch of a run is still ordered by step size)
          } else {
            const int64_t L = sc.phase == 0 ? std::min<int64_t>(SCHED_START, batch)
                            : sc.phase == 3 ? batch                     // shorter did not pay: the tails decide, so the longest launch next
                            : sc.phase == 5 ? std::min<int64_t>(sc.len * 2, batch)   // … and only if THAT loses, up one doubling at a time
                            : std::max<int64_t>(sc.len / 2, SCHED_MIN);   // phase 1: the shorter neighbour first; phase 2: further down
            const int64_t n_g = std::max<int64_t>(1, SCHED_GROUP / L);
            if (left >= L * n_g + L) {               // (worth timing, and something left to use the answer on)
              sc.primed = true;
              sc.g_len = L; sc.g_left = (int)n_g;
              k = L; probing = true;
              if (!c->work_grp) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->work_grp), sizeof(long long) * (size_t)c->N));
              if (!c->work_sum) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->work_sum), sizeof(long long)));
              HIPCHK(hipMemcpyAsync(c->work_grp, c->acc_nsteps, sizeof(long long) * (size_t)c->N, hipMemcpyDeviceToDevice, c->stream));
              HIPCHK(hipStreamSynchronize(c->stream));  // (the clock starts on an empty stream)
              sc.g_t0 = std::chrono::steady_clock::now();
            } else {
              const int64_t dbatch = sc.best_len > 0 ? sc.best_len : batch;   // too little left to learn from: the best length known
              const int64_t nb_left = (left + dbatch - 1) / dbatch;
              k = (left + nb_left - 1) / nb_left;
            }
          }
        } else {
          const int64_t dbatch = draw_batch_env > 0 ? draw_batch_env : (sc.phase == 4 && sched_env != 0 && order_refresh ? sc.best_len : batch);
          // (split the remaining transitions evenly: 50 = 13+13+12+12, not 16+16+16+2 — a short last batch would pay the whole
          // tree-size tail for two transitions)
          const int64_t nb_left = (left + dbatch - 1) / dbatch;
          k = (left + nb_left - 1) / nb_left;
        }
        // AHMC_NUTS_FIRST_BATCH=n (experiments; default off): a short first launch while the dispatch order is still the one by
        // step size, so that everything after it is scheduled by measured work (order_by_work below)
        const int first_batch = getenv("AHMC_NUTS_FIRST_BATCH") ? atoi(getenv("AHMC_NUTS_FIRST_BATCH")) : 0;
        if (first_batch >= 4 && !c->order_from_work && !c->eps_scalar && left > 2 * (int64_t)first_batch && !probing) k = std::min<int64_t>(k, first_batch);
        const int64_t j = i - (drop_warmup ? n_adapts : 0);
        T* dst = so ? so + (size_t)(j - 1) * c->D * c->N : nullptr;
        T* dev_dst = dst;
        const bool via_stage = so && !so_on_device;
        const int slot = (int)(n_staged & 1);
        if (via_stage) {  // host buffer: the kernel writes the batch's draws into a device stage
          int rc1 = stage_acquire(c, slot, (size_t)k * c->D * c->N);
          if (rc1) return rc1;
          dev_dst = c->stage[slot];
        }
        if (order_refresh) {
          if (!c->work_prev) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->work_prev), sizeof(long long) * (size_t)c->N));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->work_last), sizeof(long long) * (size_t)c->N));
          }
          HIPCHK(hipMemcpyAsync(c->work_prev, c->acc_nsteps, sizeof(long long) * (size_t)c->N, hipMemcpyDeviceToDevice, c->stream));
        }
        // (the launch after this one: the same length unless a timed group ends here or the run does — its normals are made beside this one)
        c->norm_hint = (left > k && !(probing && sc.g_left == 1)) ? std::min<int64_t>(k, left - k) : 0;
        int rc = nuts_transition(c, cfg->max_depth, cfg->delta_max, cfg->criterion, cfg->sampler, cfg->refresh_alpha, true,
                                 (int)k, dev_dst);
        if (rc) return rc;
        if (order_refresh && k >= 2) {
          hipLaunchKernelGGL(k_work_since, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->acc_nsteps, c->work_prev, c->work_last, (int64_t)c->N);
          HIPCHK(hipGetLastError());
          rc = build_order(c, (int)k, c->work_last);
          if (rc) return rc;
          c->order_valid = true;
          c->order_from_work = true;
        }
        if (probing && --sc.g_left == 0) {
          // leapfrogs of the group ÷ its wall time (everything it needed: normals, both passes, the re-sorts)
          HIPCHK(hipMemsetAsync(c->work_sum, 0, sizeof(long long), c->stream));
          hipLaunchKernelGGL(k_work_sum, dim3(64), dim3(256), 0, c->stream, c->acc_nsteps, c->work_grp, c->work_sum, (int64_t)c->N);
          HIPCHK(hipGetLastError());
          long long w = 0;
          HIPCHK(hipMemcpyAsync(&w, c->work_sum, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(hipStreamSynchronize(c->stream));
          const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - sc.g_t0).count();
          const double thr = dt > 0 ? (double)w / dt : 0.0;
          static const bool dbg_s = getenv("AHMC_DEBUG") != nullptr;
          if (dbg_s) fprintf(stderr, "[ahmc] sched: phase %d, %lld transitions per launch: %.4e leapfrog/s (best so far %lld: %.4e)\n", sc.phase, (long long)k, thr, (long long)sc.best_len, sc.best_thr);
          if (sc.phase == 0) { sc.best_len = sc.len = k; sc.best_thr = thr; sc.phase = k / 2 >= SCHED_MIN ? 1 : 3; }
          else if (sc.phase == 1 || sc.phase == 2) {  // tried the shorter neighbour of the best
            if (thr > sc.best_thr * 1.02) {
              sc.best_len = sc.len = k; sc.best_thr = thr;
              sc.phase = k / 2 >= SCHED_MIN ? 2 : 4;
            } else if (sc.phase == 1) { sc.len = sc.best_len; sc.phase = sc.best_len * 2 <= batch ? 3 : 4; }  // shorter does not pay: look the other way
            else sc.phase = 4;
          } else if (sc.phase == 3) {                 // tried the longest launch
            if (thr >= sc.best_thr * 0.985) { sc.best_len = sc.len = k; sc.best_thr = std::max(sc.best_thr, thr); sc.phase = 4; }
            else sc.phase = sc.best_len * 4 <= batch ? 5 : 4;   // (something between the start length and the longest is left to try)
          } else if (sc.phase == 5) {                 // tried the longer neighbour
            if (thr >= sc.best_thr * 0.985) {
              sc.best_len = sc.len = k; sc.best_thr = std::max(sc.best_thr, thr);
              sc.phase = k * 4 <= batch ? 5 : 4;
            } else sc.phase = 4;
          }
          if (dbg_s && sc.phase == 4) fprintf(stderr, "[ahmc] sched: settled at %lld transitions per launch\n", (long long)sc.best_len);
        }
        if (via_stage) {
          HIPCHK(hipEventRecord(c->stage_ready[slot], c->stream));
          // the PREVIOUS batch's draws go to the host while this batch computes (a copy to pageable memory blocks the
          // calling thread, so it is issued after this batch's launch, not before)
          rc = flush_pending();
          if (rc) return rc;
          pend_dst = dst; pend_slot = slot; pend_bytes = nb * (size_t)k;
          ++n_staged;
        }
        c->acc_ntrans += k;
        i += k;
        rc = order_by_work(false);  // (first batch of a fresh chain set: from now on schedule by measured work)
        if (rc) return rc;
        continue;
      }
      static const bool fused_adapt = getenv("AHMC_ADAPT_FUSED") ? atoi(getenv("AHMC_ADAPT_FUSED")) != 0 : true;
      if (adapting && fused_adapt && cfg->nuts && cfg->sampler == AHMC_TS_MULTINOMIAL && cfg->criterion == AHMC_TC_GENERALISED &&
          !dense_engine(c) && c->integ_kind != AHMC_INTEGRATOR_TEMPERED && c->target_kind != AHMC_TARGET_EXTERNAL &&
          (!so || !keep || so_on_device) &&
          !(c->var_estimator == AHMC_VAR_POOLED && c->adapt_kind != AHMC_ADAPT_STAN && c->adapt_kind != AHMC_ADAPT_STEPSIZE)) {
        // warm-up in batches too: adapt! runs inside the kernel (k_nuts MODE 3), no per-transition launch
        // (round 4, measured and dropped: the warm-up in launches of 8 / 32 / 64 transitions, each ordered by the work of the one before
        // it — cfg3 1.98 / 2.00e9 against 1.98e9 for one launch ordered by step size, cfg2 2.27e9 against 2.39e9: while the step
        // sizes still move a launch's work does not predict the next one's any better than ϵ does, and every launch pays its tail.
        // Nor does a PILOT: the first 50 / 100 / 200 transitions as a launch of their own and the rest ordered by the work measured
        // in it — cfg3 warm-up 1.99 / 1.96 / 1.90e9 against 2.05e9, cfg2 2.54e9 against 2.57e9.  The one launch's wave timeline
        // (profiles/r4_cfg3_wave_timeline_warmup_launch.json): fill 0.65, the longest wave 0.48 of the launch)
        const int64_t left = std::min(n_adapts, n_samples) - i + 1, nb_left = (left + batch - 1) / batch;  // (a run may end mid-warm-up)
        int64_t k = (left + nb_left - 1) / nb_left;
        if (c->var_estimator == AHMC_VAR_POOLED && c->adapt_kind == AHMC_ADAPT_STAN && c->metric_kind == AHMC_METRIC_DIAG) {
          // the pooled estimator couples the chains at the window ends: a batch stops there (one reduction, and with a
          // communicator one all-gather, per window — not per transition)
          if (i == 1 || c->windows_n_adapts != n_adapts) {
            c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, n_adapts);
            c->windows_n_adapts = n_adapts;
          }
          for (int64_t sp : c->windows.splits) {
            const int64_t stan_at = c->stan_i + 1;  // StanHMCAdaptor.state.i of transition i
            if (sp >= stan_at) { k = std::min<int64_t>(k, sp - stan_at + 1); break; }
          }
        }
        const int64_t j = i - (drop_warmup ? n_adapts : 0);
        T* dst = (so && keep) ? so + (size_t)(j - 1) * c->D * c->N : nullptr;
        // (round 4, measured and dropped: the warm-up as 2 / 4 interleaved groups of chains, each a sequence of launches of 8 … 125
        // transitions on its own stream, so that the slots one group's launch leaves empty at its end would be filled by the others'
        // — cfg3 warm-up 1.81 / 1.10e9 (launches of 32) against 1.94e9 for the one launch, cfg2 2.31–2.38e9 against 2.59e9: the
        // queues do not interleave at workgroup granularity, a group's launch only under-fills the chip)
        {
          const int64_t after = std::min(n_adapts, n_samples) - (i + k) + 1;   // adapting transitions left after this launch
          c->norm_hint = after > 0 ? std::min<int64_t>(k, after) : 0;
        }
        int rc = nuts_adapt_batch(c, cfg, (int)k, i, n_adapts, keep, dst);
        if (rc) return rc;
        if (keep) c->acc_ntrans += k;
        i += k;
        continue;
      }
      const int dense_pool_env = getenv("AHMC_DENSE_POOL") ? atoi(getenv("AHMC_DENSE_POOL")) : 1;  // (per call, as dn_nuts_transition reads it)
      if (adapting && fused_adapt && dense_pool_env != 0 && cfg->nuts && dense_engine(c) && c->adapt_kind == AHMC_ADAPT_STEPSIZE &&
          (cfg->sampler == AHMC_TS_MULTINOMIAL || cfg->sampler == AHMC_TS_SLICE) &&
          cfg->refresh_alpha == 0 && c->target_kind != AHMC_TARGET_EXTERNAL &&
          (!so || !keep || so_on_device)) {
        // dense engine, StepSizeAdaptor: the warm-up in batches too — every chain adapts its own ϵ at the end of each of its
        // transitions inside the tree kernel and goes on, instead of all chains waiting for the longest tree of every transition
        const int64_t left = std::min(n_adapts, n_samples) - i + 1, nb_left = (left + batch - 1) / batch;
        const int64_t k = (left + nb_left - 1) / nb_left;
        const int64_t j = i - (drop_warmup ? n_adapts : 0);
        T* dst = (so && keep) ? so + (size_t)(j - 1) * c->D * c->N : nullptr;
        int rc = dn_nuts_transition(c, cfg->max_depth, cfg->delta_max, cfg->criterion, cfg->sampler, cfg->refresh_alpha, keep, (int)k, dst, i - 1, n_adapts);
        if (rc) return rc;
        c->eps_scalar = false;
        if (i + k - 1 >= n_adapts) c->adapting = false;
        if (keep) c->acc_ntrans += k;
        i += k;
        continue;
      }
      int rc = cfg->nuts ? nuts_transition(c, cfg->max_depth, cfg->delta_max, cfg->criterion, cfg->sampler, cfg->refresh_alpha, keep)
                         : hmc_transition(c, cfg->L, cfg->lambda, cfg->sampler, cfg->refresh_alpha, keep);
      if (rc) return rc;
      rc = adapt(c, i, n_adapts);
      if (rc) return rc;
      if (keep) {
        c->acc_ntrans += 1;
        if (so) {
          int64_t j = i - (drop_warmup ? n_adapts : 0);
          HIPCHK(hipMemcpyAsync(so + (size_t)(j - 1) * c->D * c->N, c->th, nb, hipMemcpyDefault, c->stream));
        }
      }
      ++i;
    }
    {  // last staged batch; the context's stream then waits for the copies, so ahmc_sync covers them
      int rc = flush_pending();
      if (rc) return rc;
      for (int s = 0; s < 2; ++s)
        if (c->stage_busy[s]) {
          HIPCHK(hipStreamWaitEvent(c->stream, c->stage_free[s], 0));
          c->stage_busy[s] = false;
        }
    }
    return AHMC_OK;
  });
}

// ---- external target: ask / tell (ahmc_ext_host.hpp) ----
int32_t ahmc_ext_begin(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int32_t n_trans) {
  FOR_CTX(ctx, { return ext_begin(c, cfg, (int)n_trans); });
}

int32_t ahmc_ext_find_good_stepsize_begin(ahmc_ctx* ctx, double initial_step_size, int32_t max_n_iters) {
  FOR_CTX(ctx, { return ext_find_eps_begin(c, initial_step_size, (int)max_n_iters); });
}

int32_t ahmc_ext_pending(ahmc_ctx* ctx, int64_t* n_pending, int32_t* chains_out, void* theta_out) {
  FOR_CTX(ctx, { return ext_pending(c, n_pending, chains_out, theta_out); });
}

int32_t ahmc_ext_advance(ahmc_ctx* ctx, const void* lp, const void* grad_neg) {
  FOR_CTX(ctx, { return ext_advance(c, lp, grad_neg); });
}

int32_t ahmc_ext_cancel(ahmc_ctx* ctx) {
  FOR_CTX(ctx, {
    HIPCHK(hipStreamSynchronize(c->stream));
    c->ext = ExtRun();
    return AHMC_OK;
  });
}

int32_t ahmc_get_info(ahmc_ctx* ctx, int32_t what, int64_t* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "get_info: out is NULL");
    switch (what) {
      case AHMC_INFO_GROUP_LANES: *out = c->G; break;
      case AHMC_INFO_ELEMS_PER_LANE: *out = c->E; break;
      case AHMC_INFO_NUTS_LAUNCHES: *out = c->nuts_launches; break;
      case AHMC_INFO_NUTS_BATCH: *out = nuts_batch(c); break;
      case AHMC_INFO_ITERATION: *out = (int64_t)c->iteration; break;
      case AHMC_INFO_NUTS_KERNEL_NS: {
        int rc = flush_nuts_events(c);
        if (rc) return rc;
        *out = (int64_t)c->nuts_kernel_ns;
        break;
      }
      case AHMC_INFO_NUTS_WARM_LAUNCHES: *out = c->nuts_warm_launches; break;
      case AHMC_INFO_NUTS_WARM_KERNEL_NS: {
        int rc = flush_nuts_events(c);
        if (rc) return rc;
        *out = (int64_t)c->nuts_warm_kernel_ns;
        break;
      }
      case AHMC_INFO_DENSE_GEMM_LAUNCHES: *out = c->dn_gemm_big; break;
      case AHMC_INFO_DENSE_GEMM_SMALL_LAUNCHES: *out = c->dn_gemm_small; break;
      case AHMC_INFO_DENSE_PIPELINES: *out = c->dn_last_pipelines; break;
      case AHMC_INFO_DENSE_POOL: *out = c->dn_last_pool; break;
      case AHMC_INFO_DENSE_EPOCH_LAUNCHES: *out = c->dn_epoch_launches; break;
      case AHMC_INFO_NUTS_DRAW_BATCH: *out = c->sched.phase == 4 ? c->sched.best_len : 0; break;
      case AHMC_INFO_STEPSIZE_SCALAR: *out = c->eps_scalar ? 1 : 0; break;
      default: return fail(c, AHMC_ERR_ARGUMENT, "get_info: unknown key");
    }
    return AHMC_OK;
  });
}

int32_t ahmc_get_accum(ahmc_ctx* ctx, int64_t* total_n_steps, int64_t* n_transitions, int64_t* n_divergent, void* sum_theta, void* sumsq_theta) {
  FOR_CTX(ctx, {
    std::vector<long long> a((size_t)c->N), b((size_t)c->N);
    HIPCHK(hipMemcpyAsync(a.data(), c->acc_nsteps, sizeof(long long) * c->N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(b.data(), c->acc_ndiv, sizeof(long long) * c->N, hipMemcpyDeviceToHost, c->stream));
    const size_t nb = sizeof(T) * c->D * c->N;
    if (sum_theta) HIPCHK(hipMemcpyAsync(sum_theta, c->acc_sum, nb, hipMemcpyDefault, c->stream));
    if (sumsq_theta) HIPCHK(hipMemcpyAsync(sumsq_theta, c->acc_sumsq, nb, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long s = 0, d = 0;
    for (int64_t i = 0; i < c->N; ++i) { s += a[(size_t)i]; d += b[(size_t)i]; }
    if (total_n_steps) *total_n_steps = s;
    if (n_transitions) *n_transitions = c->acc_ntrans;
    if (n_divergent) *n_divergent = d;
    return AHMC_OK;
  });
}

int32_t ahmc_reset_accum(ahmc_ctx* ctx) {
  FOR_CTX_MUT(ctx, { return reset_accum(c); });
}

int32_t ahmc_get_accum_state(ahmc_ctx* ctx, int64_t* n_transitions, int64_t* n_steps, int64_t* n_divergent, void* sum_theta, void* sumsq_theta,
                             void* energy_sums) {
  FOR_CTX(ctx, {
    static_assert(sizeof(long long) == sizeof(int64_t), "accumulators are 64-bit");
    const size_t nb = sizeof(T) * c->D * c->N;
    if (n_transitions) *n_transitions = c->acc_ntrans;
    if (n_steps) HIPCHK(hipMemcpyAsync(n_steps, c->acc_nsteps, sizeof(int64_t) * c->N, hipMemcpyDefault, c->stream));
    if (n_divergent) HIPCHK(hipMemcpyAsync(n_divergent, c->acc_ndiv, sizeof(int64_t) * c->N, hipMemcpyDefault, c->stream));
    if (sum_theta) HIPCHK(hipMemcpyAsync(sum_theta, c->acc_sum, nb, hipMemcpyDefault, c->stream));
    if (sumsq_theta) HIPCHK(hipMemcpyAsync(sumsq_theta, c->acc_sumsq, nb, hipMemcpyDefault, c->stream));
    if (energy_sums) HIPCHK(hipMemcpyAsync(energy_sums, c->acc_energy, sizeof(T) * 5 * c->N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_set_accum_state(ahmc_ctx* ctx, int64_t n_transitions, const int64_t* n_steps, const int64_t* n_divergent, const void* sum_theta,
                             const void* sumsq_theta, const void* energy_sums) {
  FOR_CTX_MUT(ctx, {
    if (n_transitions < 0) return fail(c, AHMC_ERR_ARGUMENT, "set_accum_state: n_transitions < 0");
    const size_t nb = sizeof(T) * c->D * c->N;
    if (n_steps) HIPCHK(hipMemcpyAsync(c->acc_nsteps, n_steps, sizeof(int64_t) * c->N, hipMemcpyDefault, c->stream));
    if (n_divergent) HIPCHK(hipMemcpyAsync(c->acc_ndiv, n_divergent, sizeof(int64_t) * c->N, hipMemcpyDefault, c->stream));
    if (sum_theta) HIPCHK(hipMemcpyAsync(c->acc_sum, sum_theta, nb, hipMemcpyDefault, c->stream));
    if (sumsq_theta) HIPCHK(hipMemcpyAsync(c->acc_sumsq, sumsq_theta, nb, hipMemcpyDefault, c->stream));
    if (energy_sums) HIPCHK(hipMemcpyAsync(c->acc_energy, energy_sums, sizeof(T) * 5 * c->N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // (the sources may be pageable host buffers)
    c->acc_ntrans = n_transitions;
    return AHMC_OK;
  });
}

// ---- adaptor checkpoint, multi-GPU gather, device diagnostics (ahmc_multi_host.hpp) ----
int32_t ahmc_get_adaptor_state(ahmc_ctx* ctx, ahmc_adaptor_state* state, void* da, void* welford) {
  FOR_CTX(ctx, { return get_adaptor_state(c, state, da, welford); });
}

int32_t ahmc_set_adaptor_state(ahmc_ctx* ctx, const ahmc_adaptor_state* state, const void* da, const void* welford) {
  FOR_CTX_MUT(ctx, { return set_adaptor_state(c, state, da, welford); });
}

int32_t ahmc_comm_unique_id(void* id_out) {
  if (!id_out || !rccl_api().ok) {
    g_create_err = !id_out ? "ahmc_comm_unique_id: id_out is NULL" : "ahmc_comm_unique_id: " + rccl_api().err;
    return !id_out ? AHMC_ERR_ARGUMENT : AHMC_ERR_RUNTIME;
  }
  ncclUniqueId uid;
  if (rccl_api().GetUniqueId(&uid) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return AHMC_ERR_RUNTIME; }
  static_assert(sizeof(uid) == AHMC_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &uid, sizeof(uid));
  return AHMC_OK;
}

int32_t ahmc_comm_init(ahmc_ctx* ctx, const void* id, int32_t n_ranks, int32_t rank) {
  FOR_CTX_MUT(ctx, { return comm_init(c, id, (int)n_ranks, (int)rank); });
}

int32_t ahmc_set_comm(ahmc_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank) {
  FOR_CTX_MUT(ctx, { return set_comm(c, nccl_comm, (int)n_ranks, (int)rank); });
}

int32_t ahmc_comm_info(ahmc_ctx* ctx, int64_t* ranks_seen, int64_t* chains_total, int64_t* chains_min, int64_t* chains_max) {
  FOR_CTX(ctx, {
    const bool multi = c->comm != nullptr;   // (what comm_probe measured; without a communicator: a world of one)
    if (ranks_seen) *ranks_seen = multi ? c->comm_seen : 1;
    if (chains_total) *chains_total = multi ? c->comm_chains_total : c->N;
    if (chains_min) *chains_min = multi ? c->comm_chains_min : c->N;
    if (chains_max) *chains_max = multi ? c->comm_chains_max : c->N;
    return AHMC_OK;
  });
}

int32_t ahmc_gather_moments(ahmc_ctx* ctx, double* mean, double* var, int64_t* n_draws, int64_t* total_n_steps, int64_t* n_divergent) {
  FOR_CTX(ctx, { return gather_moments(c, mean, var, n_draws, total_n_steps, n_divergent); });
}

int32_t ahmc_gather_state(ahmc_ctx* ctx, void* theta_all) {
  FOR_CTX(ctx, { return gather_state(c, theta_all); });
}

int32_t ahmc_ebfmi(ahmc_ctx* ctx, void* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "ebfmi: out is NULL");
    int rc = red_buf(c, (size_t)c->N);  // (T <= double: N doubles hold N elements of T)
    if (rc) return rc;
    T* tmp = reinterpret_cast<T*>(c->red);
    hipLaunchKernelGGL((k_ebfmi<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->acc_energy, c->N, tmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, tmp, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

int32_t ahmc_ess(ahmc_ctx* ctx, const void* draws, int64_t n_draws, void* out) {
  FOR_CTX(ctx, {
    if (!draws || !out) return fail(c, AHMC_ERR_ARGUMENT, "ess: NULL argument");
    if (n_draws < 4) return fail(c, AHMC_ERR_ARGUMENT, "ess: at least 4 draws per chain");
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, draws) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (!dev) return fail(c, AHMC_ERR_ARGUMENT, "ess: draws must be the device buffer ahmc_sample filled");
    const int64_t DN = c->D * c->N;
    const bool out_dev = hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    T* dst = static_cast<T*>(out);
    if (!out_dev) {
      int rc = red_buf(c, (size_t)DN);
      if (rc) return rc;
      dst = reinterpret_cast<T*>(c->red);
    }
    hipLaunchKernelGGL((k_ess<T>), dim3((unsigned)((DN + 255) / 256)), dim3(256), 0, c->stream, static_cast<const T*>(draws), DN, n_draws, dst);
    HIPCHK(hipGetLastError());
    if (!out_dev) HIPCHK(hipMemcpyAsync(out, dst, sizeof(T) * DN, hipMemcpyDefault, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AHMC_OK;
  });
}

}  // extern "C"
