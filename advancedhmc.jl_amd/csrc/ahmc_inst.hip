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

template <class T, int TK>
int Inst<T, TK>::nuts_occupancy(int G, int E, int mode, size_t smem) {
  int occ = 0;
  hipError_t err = hipErrorInvalidValue;
  with_geometry(G, E, [&](auto g, auto e) {
    constexpr int GG = decltype(g)::value, EE = decltype(e)::value;
    if (mode == 0) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_nuts<T, GG, EE, 0, TK>, (GG > 64 ? GG : 64), smem);
    else if (mode == 1) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_nuts<T, GG, EE, 1, TK>, (GG > 64 ? GG : 64), smem);
    else if (mode == 3) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_nuts<T, GG, EE, 3, TK>, (GG > 64 ? GG : 64), smem);
    else if (mode == 4) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_nuts<T, GG, EE, 4, TK>, (GG > 64 ? GG : 64), smem);
    else err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_nuts<T, GG, EE, 2, TK>, (GG > 64 ? GG : 64), smem);
  });
  return err == hipSuccess ? occ : 0;
}

template <class T, int TK>
void Inst<T, TK>::nuts_set_smem(int G, int E, int mode, size_t smem) {
  with_geometry(G, E, [&](auto g, auto e) {
    constexpr int GG = decltype(g)::value, EE = decltype(e)::value;
    const void* f = mode == 0   ? reinterpret_cast<const void*>(&k_nuts<T, GG, EE, 0, TK>)
                    : mode == 1 ? reinterpret_cast<const void*>(&k_nuts<T, GG, EE, 1, TK>)
                    : mode == 3 ? reinterpret_cast<const void*>(&k_nuts<T, GG, EE, 3, TK>)
                    : mode == 4 ? reinterpret_cast<const void*>(&k_nuts<T, GG, EE, 4, TK>)
                                : reinterpret_cast<const void*>(&k_nuts<T, GG, EE, 2, TK>);
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  (void)hipGetLastError();
}

template <class T, int TK>
void Inst<T, TK>::nuts(int G, int E, int mode, unsigned grid, int wpb, size_t smem, hipStream_t s, const KP<T>& p) {
  with_geometry(G, E, [&](auto g, auto e) {
    constexpr int GG = decltype(g)::value, EE = decltype(e)::value;
    if (mode == 0) hipLaunchKernelGGL((k_nuts<T, GG, EE, 0, TK>), dim3(grid), dim3(64 * wpb), smem, s, p);
    else if (mode == 1) hipLaunchKernelGGL((k_nuts<T, GG, EE, 1, TK>), dim3(grid), dim3(64 * wpb), smem, s, p);
    else if (mode == 3) hipLaunchKernelGGL((k_nuts<T, GG, EE, 3, TK>), dim3(grid), dim3(64 * wpb), smem, s, p);
    else if (mode == 4) hipLaunchKernelGGL((k_nuts<T, GG, EE, 4, TK>), dim3(grid), dim3(64 * wpb), smem, s, p);
    else hipLaunchKernelGGL((k_nuts<T, GG, EE, 2, TK>), dim3(grid), dim3(64 * wpb), smem, s, p);
  });
}

template struct Inst<AHMC_INST_T, AHMC_INST_TK>;

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
