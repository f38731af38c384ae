// ahmc_inst.hip — one translation unit per (AHMC_INST_T, AHMC_INST_TK): the kernels of that element
// type and log-density family for every thread geometry.  Built 8 times by build.py.
#include <type_traits>

#include "ahmc_inst.hpp"
#include "ahmc_nuts.hpp"

#ifndef AHMC_INST_T
#error "compile with -DAHMC_INST_T=float|double -DAHMC_INST_TK=0..3 (4 = a target plugin: -DAHMC_USER_TARGET_HEADER=... -DAHMC_PLUGIN_G= -DAHMC_PLUGIN_E=)"
#endif

namespace ahmc {

#define GEO_LAUNCH(KERNEL, ...)                                                                       \
  with_geometry(G, E, [&](auto g, auto e) {                                                            \
    hipLaunchKernelGGL((KERNEL<T, decltype(g)::value, decltype(e)::value, TK>), dim3(grid), dim3(decltype(g)::value > 64 ? decltype(g)::value : 256), 0, s, __VA_ARGS__); \
  })

template <class T, int TK>
void Inst<T, TK>::fill_caches(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p) { GEO_LAUNCH(k_fill_caches, p); }
template <class T, int TK>
void Inst<T, TK>::refresh(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p) { GEO_LAUNCH(k_refresh, p); }
template <class T, int TK>
void Inst<T, TK>::leapfrog(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p) { GEO_LAUNCH(k_leapfrog, p); }
template <class T, int TK>
void Inst<T, TK>::hmc(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p) { GEO_LAUNCH(k_hmc, p); }
template <class T, int TK>
void Inst<T, TK>::find_eps(int G, int E, unsigned grid, hipStream_t s, const KP<T>& p, T* eps_out) {
  GEO_LAUNCH(k_find_eps, p, eps_out);
}

// k_nuts of one (geometry, mode): the three things the host does with it.  PART_B selects which half of the instantiations this
// function body may name (nuts_in_part_b): the other half is compiled in the other translation unit.
template <class T, int TK, bool PART_B, class F>
static void with_nuts_kernel(int G, int E, int mode, F&& f) {
  with_geometry(G, E, [&](auto g, auto e) {
    constexpr int GG = decltype(g)::value, EE = decltype(e)::value;
    auto one = [&](auto m) {
      constexpr int M = decltype(m)::value;
      if constexpr (nuts_in_part_b(GG, M) == PART_B) f(reinterpret_cast<const void*>(&k_nuts<T, GG, EE, M, TK>), std::integral_constant<int, GG>{});
    };
    if (mode == 0) one(std::integral_constant<int, 0>{});
    else if (mode == 1) one(std::integral_constant<int, 1>{});
    else if (mode == 3) one(std::integral_constant<int, 3>{});
    else if (mode == 4) one(std::integral_constant<int, 4>{});
    else one(std::integral_constant<int, 2>{});
  });
}
template <class T, int TK, bool PART_B>
static int nuts_occupancy_impl(int G, int E, int mode, size_t smem) {
  int occ = 0;
  hipError_t err = hipErrorInvalidValue;
  with_nuts_kernel<T, TK, PART_B>(G, E, mode, [&](const void* f, auto gg) {
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, f, (decltype(gg)::value > 64 ? decltype(gg)::value : 64), smem);
  });
  return err == hipSuccess ? occ : 0;
}
template <class T, int TK, bool PART_B>
static void nuts_set_smem_impl(int G, int E, int mode, size_t smem) {
  with_nuts_kernel<T, TK, PART_B>(G, E, mode, [&](const void* f, auto) { (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  (void)hipGetLastError();
}
template <class T, int TK, bool PART_B>
static void nuts_impl(int G, int E, int mode, unsigned grid, int wpb, size_t smem, hipStream_t s, const KP<T>& p) {
  with_nuts_kernel<T, TK, PART_B>(G, E, mode, [&](const void* f, auto) {
    KP<T> arg = p;
    void* args[] = {&arg};
    (void)hipLaunchKernel(f, dim3(grid), dim3(64 * wpb), args, smem, s);
  });
}

#if !defined(AHMC_INST_PART) || AHMC_INST_PART == 0
template <class T, int TK>
int Inst<T, TK>::nuts_occupancy(int G, int E, int mode, size_t smem) {
  return nuts_in_part_b(G, mode) ? InstB<T, TK>::nuts_occupancy(G, E, mode, smem) : nuts_occupancy_impl<T, TK, false>(G, E, mode, smem);
}
template <class T, int TK>
void Inst<T, TK>::nuts_set_smem(int G, int E, int mode, size_t smem) {
  if (nuts_in_part_b(G, mode)) InstB<T, TK>::nuts_set_smem(G, E, mode, smem);
  else nuts_set_smem_impl<T, TK, false>(G, E, mode, smem);
}
template <class T, int TK>
void Inst<T, TK>::nuts(int G, int E, int mode, unsigned grid, int wpb, size_t smem, hipStream_t s, const KP<T>& p) {
  if (nuts_in_part_b(G, mode)) InstB<T, TK>::nuts(G, E, mode, grid, wpb, smem, s, p);
  else nuts_impl<T, TK, false>(G, E, mode, grid, wpb, smem, s, p);
}
template <class T, int TK>
int64_t Inst<T, TK>::scratch_layout() {
  const int64_t b = InstB<T, TK>::scratch_layout();
  return b == nuts_scratch_layout() ? b : -1;
}
#endif
#if !defined(AHMC_INST_PART) || AHMC_INST_PART == 1
template <class T, int TK>
int InstB<T, TK>::nuts_occupancy(int G, int E, int mode, size_t smem) { return nuts_occupancy_impl<T, TK, true>(G, E, mode, smem); }
template <class T, int TK>
void InstB<T, TK>::nuts_set_smem(int G, int E, int mode, size_t smem) { nuts_set_smem_impl<T, TK, true>(G, E, mode, smem); }
template <class T, int TK>
void InstB<T, TK>::nuts(int G, int E, int mode, unsigned grid, int wpb, size_t smem, hipStream_t s, const KP<T>& p) {
  nuts_impl<T, TK, true>(G, E, mode, grid, wpb, smem, s, p);
}
template <class T, int TK>
int64_t InstB<T, TK>::scratch_layout() { return nuts_scratch_layout(); }
template struct InstB<AHMC_INST_T, AHMC_INST_TK>;
#endif

#if !defined(AHMC_INST_PART) || AHMC_INST_PART == 0
#if defined(AHMC_INST_PART)
extern template struct InstB<AHMC_INST_T, AHMC_INST_TK>;   // (the other translation unit of this (T, TK))
#endif
template struct Inst<AHMC_INST_T, AHMC_INST_TK>;
#endif

#if AHMC_INST_TK == 4
// ---- target plugin: the user's log-density (AHMC_USER_TARGET_HEADER, see include/ahmc_user_target.h) compiled into every
// trajectory kernel of ONE element type and ONE thread geometry; the engine binds this table with dlopen ----
#ifndef AHMC_PLUGIN_NPARAMS
#define AHMC_PLUGIN_NPARAMS -1
#endif
#ifndef AHMC_SOURCES_DIGEST
#define AHMC_SOURCES_DIGEST ""
#endif
static const TargetOps<AHMC_INST_T> plugin_ops = make_target_ops<AHMC_INST_T, 4>();
}  // namespace ahmc
extern "C" __attribute__((visibility("default"))) const ahmc::TargetPluginDesc ahmc_target_plugin_v1 = {
    ahmc::AHMC_PLUGIN_ABI, (int32_t)sizeof(ahmc::TargetPluginDesc), (int32_t)sizeof(ahmc::KP<AHMC_INST_T>), (int32_t)(sizeof(AHMC_INST_T) == 4 ? 0 : 1),
    AHMC_PLUGIN_G, AHMC_PLUGIN_E, (int64_t)(AHMC_PLUGIN_NPARAMS), &ahmc::plugin_ops, AHMC_SOURCES_DIGEST};
namespace ahmc {
#endif

}  // namespace ahmc
