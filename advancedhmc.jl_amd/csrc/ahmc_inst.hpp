// ahmc_inst.hpp — launch table of the target-dependent kernels.
//
// The built-in log-density family is a COMPILE-TIME parameter of every kernel that evaluates it
// (a 4-way run-time switch costs 36 VGPRs in k_nuts), so each (element type, family) pair gets its
// own translation unit (ahmc_inst.hip compiled 8 times, in parallel, by build.py) that explicitly
// instantiates Inst<T, TK>; the host API (ahmc_api.hip) only sees these declarations.
#pragma once

#include "ahmc_kernels.hpp"

namespace ahmc {

// Thread geometries (G lanes per chain, E elements per lane) with compiled kernels.
#if defined(AHMC_PLUGIN_G) && defined(AHMC_PLUGIN_E)
// a target plugin (ahmc_set_target_plugin) is compiled for the ONE geometry of the context it serves
#define AHMC_GEOMETRIES(X) X(AHMC_PLUGIN_G, AHMC_PLUGIN_E)
#else
#ifndef AHMC_GEOMETRIES_EXTRA
#define AHMC_GEOMETRIES_EXTRA(X)  // (experiments: -D'AHMC_GEOMETRIES_EXTRA(X)=X(8,4)' adds geometries to a variant build)
#endif
#define AHMC_GEOMETRIES(X) \
  X(4, 1) X(8, 1) X(16, 1) X(32, 1) X(64, 1) X(4, 2) X(8, 2) X(16, 2) X(32, 2) X(64, 2) X(32, 4) X(64, 4) X(64, 8) \
  X(128, 4) X(256, 4) X(512, 4) X(128, 8) X(256, 8) X(512, 8) AHMC_GEOMETRIES_EXTRA(X)
#endif

// call f(std::integral_constant<int,G>{}, std::integral_constant<int,E>{}) for a run-time geometry
template <class F>
inline void with_geometry(int G, int E, F&& f) {
#define AHMC_GEO_CASE(g, e) \
  if (G == g && E == e) { f(std::integral_constant<int, g>{}, std::integral_constant<int, e>{}); return; }
  AHMC_GEOMETRIES(AHMC_GEO_CASE)
#undef AHMC_GEO_CASE
}

constexpr int AHMC_N_TARGETS = 4;  // iso, diag, funnel, hier (AHMC_TARGET_* 0..3)

template <class T, int TK>
struct Inst {
  static void fill_caches(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  static void refresh(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  static void leapfrog(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  static void hmc(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  static void find_eps(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p, T* eps_out);
  static int nuts_occupancy(int G, int E, int mode, size_t smem);  // single-wave workgroups per CU
  static void nuts_set_smem(int G, int E, int mode, size_t smem);
  static void nuts(int G, int E, int mode, unsigned grid, int waves_per_block, size_t smem, hipStream_t s, const KP<T>& p);
  static int64_t scratch_layout();  // nuts_scratch_layout() as THIS unit's kernels were compiled (both parts)
};

// The NUTS kernels come in two PARTS that are compiled with different optimiser settings (round 4, build.py): part B — the
// warm-up instantiations (MODE 3 / 4, adapt! inside the kernel) of every geometry and every instantiation of the multi-wave
// geometries (G > 64) — is built with the machine-level loop-invariant code motion off.  Those kernels are at their register cap,
// and what LICM hoists out of their transition loop (the f64 constants of exp / log in the adaptor's arithmetic, addresses) it then
// has to spill and reload in every transition: k_nuts<double,64,2,3,0> 240 → 48 B of scratch per lane, <double,256,8,3,3> 596 →
// 380; cfg5 9.6e7 → 1.12e8 leapfrog/s, cfg2's warm-up +1.6 %.  The sampling kernels of the single-wave geometries lose by it
// (cfg2 draws −1.5 %, cfg3 −4 %: their hot loop re-materialises what was hoisted), so they stay in part A.
constexpr bool nuts_in_part_b(int G, int mode) { return mode >= 3 || G > 64; }
template <class T, int TK>
struct InstB {
  static int nuts_occupancy(int G, int E, int mode, size_t smem);
  static void nuts_set_smem(int G, int E, int mode, size_t smem);
  static void nuts(int G, int E, int mode, unsigned grid, int waves_per_block, size_t smem, hipStream_t s, const KP<T>& p);
  static int64_t scratch_layout();
};

// The launch table of one Inst<T, TK> as plain function pointers: what the host API calls.  The four built-in families
// fill it from the instantiations linked into the library; a user log-density compiled INTO the trajectory kernels
// (TK = AHMC_TK_PLUGIN, `ahmc_set_target_plugin`) fills it from the plugin .so's own Inst<T, 4> behind dlopen.
template <class T>
struct TargetOps {
  void (*fill_caches)(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  void (*refresh)(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  void (*leapfrog)(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  void (*hmc)(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p);
  void (*find_eps)(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p, T* eps_out);
  int (*nuts_occupancy)(int G, int E, int mode, size_t smem);
  void (*nuts_set_smem)(int G, int E, int mode, size_t smem);
  void (*nuts)(int G, int E, int mode, unsigned grid, int waves_per_block, size_t smem, hipStream_t s, const KP<T>& p);
  int64_t (*scratch_layout)();   // (AHMC_PLUGIN_ABI 2) the k_nuts scratch layout these kernels index: −1 if the unit's two parts disagree
};
template <class T, int TK>
inline TargetOps<T> make_target_ops() {
  return TargetOps<T>{&Inst<T, TK>::fill_caches, &Inst<T, TK>::refresh, &Inst<T, TK>::leapfrog, &Inst<T, TK>::hmc, &Inst<T, TK>::find_eps,
                      &Inst<T, TK>::nuts_occupancy, &Inst<T, TK>::nuts_set_smem, &Inst<T, TK>::nuts, &Inst<T, TK>::scratch_layout};
}

constexpr int AHMC_PLUGIN_ABI = 2;  // bump when KP<T>, TargetP<T> or TargetOps<T> change meaning without changing size
constexpr int AHMC_TK_PLUGIN = 4;  // the TK of a plugin's instantiation (not an AHMC_TARGET_* code: the API kind is AHMC_TARGET_PLUGIN)
// what a plugin .so exports under the C name `ahmc_target_plugin_v1` (built by advancedhmc.jl_amd/build.py: build_target_plugin)
struct TargetPluginDesc {
  int32_t plugin_abi;     // AHMC_PLUGIN_ABI of the headers it was compiled against
  int32_t struct_bytes;   // sizeof(TargetPluginDesc): layout check
  int32_t kp_bytes;       // sizeof(KP<T>): the plugin and the engine must agree on the kernel-argument struct
  int32_t dtype;          // AHMC_F32 = 0 / AHMC_F64 = 1 (ahmc_hip.h)
  int32_t G, E;           // the one thread geometry it was compiled for
  int64_t n_params;       // parameters its log-density reads (tp.params), −1 = any
  const void* ops;        // const TargetOps<T>*
  const char* sources_digest;  // kernel-source digest of the engine headers it was compiled against (informational; the build helper keys its cache on it)
};

#define AHMC_DECLARE_INST(T) \
  extern template struct Inst<T, 0>; extern template struct Inst<T, 1>; \
  extern template struct Inst<T, 2>; extern template struct Inst<T, 3>;

#ifndef AHMC_INST_T  // the host API only links against the instantiations
AHMC_DECLARE_INST(float)
AHMC_DECLARE_INST(double)
#endif

// run f(std::integral_constant<int,TK>{}) for a run-time target kind 0..3
template <class F>
inline void with_target(int kind, F&& f) {
  switch (kind) {
    case 0: f(std::integral_constant<int, 0>{}); break;
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    default: break;
  }
}

}  // namespace ahmc
