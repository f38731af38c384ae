// ahmc_nuts.hpp — the NUTS transition kernel (src/trajectory.jl:626-742), gfx950.
//
// Iterative form of build_tree: leaves are visited in integration order; after
// leaf i one merge is done per trailing zero bit of i, lowest level first — the same merges, in
// the same order, with the same RNG draws as the reference's recursion.
//
// Mapping.  A chain = a group of G lanes (E elements per lane); a wave carries
// 64/G chains in LOCKSTEP: every live chain of the wave is at the same (doubling j, leaf i), so
// loop counters, merge levels and slot addresses are wave-uniform (scalar) and only "is this
// chain still alive" is a per-lane predicate.  Waves are persistent and pull chunks of 64/G
// chains from a global work queue.
//
// What lives where:
//   registers : the moving edge of the tree (θ, r, -∇ℓπ), M⁻¹, and the turn statistic of the
//               subtree being merged (A = ρ or θ_first; RF = r of its first-built leaf)
//   vector slots (LDS for the hottest, global scratch for the rest; 16-byte-chunk-interleaved so
//               every wave access is 64 consecutive 16-B pieces): the pending subtree of every
//               level (A, RF), the dormant edge of the tree (θ, r, g), the whole-tree ρ, and
//               (r0, g0) of the start point
//   LDS scalars: per pending level {w = ℓw or n, Σα, nα, ΔH_max, candidate leaf index}
// The multinomial/slice candidate is carried as a LEAF INDEX (signed distance from the start
// point along the trajectory), not as a phase point: the selected point is re-integrated from the
// start at the end (vector work only, no reductions, no RNG), which halves the pending state and
// removes every candidate copy from the merge path.
#pragma once
#ifndef AHMC_SCALAR_ANY
#define AHMC_SCALAR_ANY 1      // loop-control predicates of a wave-owning chain are tested directly instead of through a ballot (0: ballot)
#endif

#ifndef AHMC_ADAPT_REREAD
#define AHMC_ADAPT_REREAD 1   // the warm-up kernels re-read the adaptor's argument block in every epilogue instead of carrying it (below)
#endif
#ifndef AHMC_RUNNING_STATS
// 1: Σα, nα and ΔH_max of the subtree a doubling builds are RUNNING values over its leaves in build order — the statistics the
// reference carries through every `combine` (src/trajectory.jl:533-542) are a sum, a count and a maximum of |·| over the same
// leaves, so only the order of the additions changes (acceptance_rate in the last bit; no decision depends on it): no per-level
// Σα / nα / ΔH_max in LDS, nothing to fold in at a merge, at a park or when a subtree ends early.  0: through every merge.
#define AHMC_RUNNING_STATS 1
#endif

// AHMC_LEAF_PROF (measurement build only, scripts/leaf_latency.py; implies the per-wave timeline record): where a leaf step's cycles go.
// Every wave reads the shader-clock counter (s_memtime) at the stage boundaries of the leaf loop — fenced by scheduling barriers, so no
// instruction moves across a stamp — and sums the differences per stage: [0] leapfrog vector work (half-steps, density and its
// gradient), [1] the energy all-reduce, [2] leaf weight / exp, divergence test, running statistics, [3] merges (operand loads, dot
// products, their all-reduce, RNG, selects), [4] parking a finished subtree + loop control, [5] the top level of a doubling (direction,
// edge swap, whole-tree U-turn test), [6] transition prologue, [7] epilogue (re-integration to the candidate, stores, adapt!); counts:
// [8] leaf steps, [9] merges, [10] doublings, [11] transitions.  A stamp costs an s_memtime + s_waitcnt lgkmcnt(0): the build is for
// attribution, not for throughput numbers.
#ifndef AHMC_LEAF_PROF
#define AHMC_LEAF_PROF 0
#endif
#if AHMC_LEAF_PROF
#undef AHMC_WAVE_TIMELINE
#define AHMC_WAVE_TIMELINE 1
#define AHMC_TL_WORDS 24
// (32-bit sums, kept scalar through readfirstlane: a launch stays far below 2^32 cycles per stage and wave)
#define AHMC_LP_TICK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned n_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__builtin_readcyclecounter()); __builtin_amdgcn_sched_barrier(0); lp_t[i] += n_ - lp_c; lp_c = n_; }
#define AHMC_LP_COUNT(i, n) { lp_t[i] += (unsigned)(n); }
#else
#define AHMC_LP_TICK(i)
#define AHMC_LP_COUNT(i, n)
#endif
#ifndef AHMC_HIER_PRE
#define AHMC_HIER_PRE 1   // multi-wave hierarchical target: μ, log τ of the next leaf ride on this leaf's energy exchange (0: round 5's broadcast per leaf)
#endif
#ifndef AHMC_CKPT_G64
#define AHMC_CKPT_G64 0   // 1: the checkpoint also where one chain fills one wave (experiment)
#endif
#ifndef AHMC_TL_WORDS
#define AHMC_TL_WORDS 8
#endif

#include <type_traits>

#include "ahmc_kernels.hpp"

namespace ahmc {

constexpr double LINW_LIMIT = 600.0;  // exp(600)·2^10 leaves is still far from the Float64 overflow; Float32 uses 60
enum { SL_OTH_TH = 0, SL_OTH_R = 1, SL_OTH_G = 2, SL_TREE_A = 3, SL_Z0_R = 4, SL_Z0_G = 5, SL_START_R = 6 };

template <class T, int E>
struct Chunking {
  static constexpr int CH = (E * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : E;  // elements per 16-B piece
  static constexpr int NCH = E / CH;
};

// Vector slots of one wave.  Slot s, piece c, lane l lives at ((s*NCH + c)*64 + l)*CH elements.
// Addresses are always formed as (wave-uniform slot base) + (32-bit lane offset) so that global
// accesses use the saddr+voffset form and no per-lane 64-bit pointer has to stay live.
// Round 5: the two homes of a slot are pointers in their OWN address spaces (LDS / global).  As two generic pointers the
// compiler merged `slot < n_lds ? lds : glb` into a select of addresses and ONE flat instruction per piece — 64-bit
// address arithmetic per access, the LDS slots reached through the flat path (longer latency, counted on vmcnt AND lgkmcnt,
// so a wait for one slot also waited for every scratch store in flight).  Now the choice is a scalar branch between a
// ds_read / ds_write with an immediate offset and a global access in the saddr form.
#ifndef AHMC_SLOTS_ADDRSPACE
#define AHMC_SLOTS_ADDRSPACE 1   // 0: generic pointers (rounds 1–4)
#endif
#if AHMC_SLOTS_ADDRSPACE
#define AHMC_AS_LDS __attribute__((address_space(3)))
#define AHMC_AS_GLB __attribute__((address_space(1)))
#else
#define AHMC_AS_LDS
#define AHMC_AS_GLB
#endif
template <class T, int E>
struct Slots {
  AHMC_AS_LDS T* lds;   // first n_lds slots
  AHMC_AS_GLB T* glb;   // the remaining ones
  int n_lds;
  unsigned lane_off;  // lane64 * CH (elements)

#define AHMC_SLOT_PUT(AS)                                                                                   \
  static __device__ __forceinline__ void put(AS T* __restrict__ sb, unsigned off, const T (&v)[E]) {         \
    constexpr int CH = Chunking<T, E>::CH, NCH = Chunking<T, E>::NCH;                                       \
    if constexpr (CH > 1) {                                                                                 \
      using V = T __attribute__((ext_vector_type(CH)));                                                     \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                     \
        V t;                                                                                                \
        _Pragma("unroll") for (int k = 0; k < CH; ++k) t[k] = v[c * CH + k];                                \
        *(AS V*)(&sb[off + (unsigned)(c * 64 * CH)]) = t;                                                    \
      }                                                                                                     \
    } else {                                                                                                \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) sb[off + (unsigned)(c * 64)] = v[c];                  \
    }                                                                                                       \
  }                                                                                                         \
  static __device__ __forceinline__ void get(const AS T* __restrict__ sb, unsigned off, T (&v)[E]) {         \
    constexpr int CH = Chunking<T, E>::CH, NCH = Chunking<T, E>::NCH;                                       \
    if constexpr (CH > 1) {                                                                                 \
      using V = T __attribute__((ext_vector_type(CH)));                                                     \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                     \
        V t = *(const AS V*)(&sb[off + (unsigned)(c * 64 * CH)]);                                            \
        _Pragma("unroll") for (int k = 0; k < CH; ++k) v[c * CH + k] = t[k];                                \
      }                                                                                                     \
    } else {                                                                                                \
      _Pragma("unroll") for (int c = 0; c < NCH; ++c) v[c] = sb[off + (unsigned)(c * 64)];                  \
    }                                                                                                       \
  }
#if AHMC_SLOTS_ADDRSPACE
  AHMC_SLOT_PUT(AHMC_AS_LDS)
  AHMC_SLOT_PUT(AHMC_AS_GLB)
#else
  AHMC_SLOT_PUT()
#endif
#undef AHMC_SLOT_PUT
  __device__ __forceinline__ void store(int slot, const T (&v)[E]) const {
    constexpr int SE = Chunking<T, E>::NCH * 64 * Chunking<T, E>::CH;
    if (slot < n_lds) put(lds + slot * SE, lane_off, v);
    else put(glb + (size_t)(slot - n_lds) * SE, lane_off, v);
  }
  __device__ __forceinline__ void load(int slot, T (&v)[E]) const {
    constexpr int SE = Chunking<T, E>::NCH * 64 * Chunking<T, E>::CH;
    if (slot < n_lds) get(lds + slot * SE, lane_off, v);
    else get(glb + (size_t)(slot - n_lds) * SE, lane_off, v);
  }
  // a slot that is NEVER in LDS (the launch plan keeps the checkpoint triples behind n_lds), `extra` more elements into it PER LANE:
  // chains that share a wave may address different triples
  __device__ __forceinline__ void store_cold(int slot, unsigned extra, const T (&v)[E]) const {
    constexpr int SE = Chunking<T, E>::NCH * 64 * Chunking<T, E>::CH;
    put(glb + (size_t)(slot - n_lds) * SE, lane_off + extra, v);
  }
  __device__ __forceinline__ void load_cold(int slot, unsigned extra, T (&v)[E]) const {
    constexpr int SE = Chunking<T, E>::NCH * 64 * Chunking<T, E>::CH;
    get(glb + (size_t)(slot - n_lds) * SE, lane_off + extra, v);
  }
};

// the vector half of a leapfrog step (no energies): used to re-integrate to the candidate
template <class T, int G, int E, int TK, bool TEMPER = true>
__device__ __forceinline__ void leapfrog_core(Point<T, E>& z, const T (&minv)[E], T eps, const TargetP<T>& tp,
                                              const LeapfrogP<T>& lf, int lane, int d0) {
  if constexpr (TEMPER) temper(lf, z.r, 1, true, 1);
  const T eh = eps / 2;
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
#pragma unroll
  for (int e = 0; e < E; ++e) z.th[e] = z.th[e] + eps * (minv[e] * z.r[e]);
  (void)target_eval<T, G, E, TK>(tp, z.th, z.g, lane, d0);
#pragma unroll
  for (int e = 0; e < E; ++e) z.r[e] = z.r[e] - eh * z.g[e];
  if constexpr (TEMPER) temper(lf, z.r, 1, false, 1);
}

// Sequential scalar draws of one NUTS transition (direction bits, multinomial / slice uniforms, Exp(1)):
// draw k = 32-bit word k & 3 of Philox block k >> 2; uniform = (w + ½)·2⁻³² ∈ (0,1), boolean = top bit.
// One Philox block (≈75 VALU) serves four draws; 32 bits of resolution are ample for tree sampling
// (the momentum normals, jitter and the static-HMC draws keep 53-bit uniforms).
// WIDE (a chain owns whole wavefronts, G >= 64): the four 16-lane rows of the wave run Philox on four consecutive
// blocks at once, so one Philox evaluation (≈75 VALU) serves SIXTEEN draws; a draw is then one v_readlane from the
// row that holds its block.  Same blocks, same words, same draw order as the narrow form — only who computes what.
template <bool WIDE>
struct DrawStreamT {
  Rng rng;
  uint32_t k;
  Philox4 blk;
  __device__ __forceinline__ void init(const Rng& r) { rng = r; k = 0; }
  __device__ __forceinline__ void resume(const Rng& r, uint32_t k0) {  // continue a stream at draw k0
    rng = r;
    k = k0;
    if constexpr (WIDE) {
      if (k & 15u) blk = rng.raw(RNG_TRANSITION, ((k >> 4) << 2) + ((threadIdx.x & 63u) >> 4));
    } else {
      if (k & 3u) blk = rng.raw(RNG_TRANSITION, k >> 2);
    }
  }
  __device__ __forceinline__ uint32_t word() {
    if constexpr (WIDE) {
      if ((k & 15u) == 0u) blk = rng.raw(RNG_TRANSITION, (k >> 2) + ((threadIdx.x & 63u) >> 4));
    } else {
      if ((k & 3u) == 0u) blk = rng.raw(RNG_TRANSITION, k >> 2);
    }
    const uint32_t lo = (k & 1u) ? blk.v[1] : blk.v[0], hi = (k & 1u) ? blk.v[3] : blk.v[2];
    uint32_t w = (k & 2u) ? hi : lo;
    if constexpr (WIDE) {
      const int row = __builtin_amdgcn_readfirstlane((int)((k >> 2) & 3u));  // k is the same in every lane of the chain
      w = (uint32_t)__builtin_amdgcn_readlane((int)w, row * 16);
    }
    ++k;
    return w;
  }
  __device__ __forceinline__ double uniform() { return ((double)word() + 0.5) * 2.3283064365386962890625e-10; }
  __device__ __forceinline__ bool boolean() { return (word() >> 31) != 0; }
  __device__ __forceinline__ double randexp() { return -log(uniform()); }
};
typedef DrawStreamT<false> DrawStream;

// MODE 0: multinomial + generalised, linear-domain weights (the default fast path)
// MODE 1: multinomial + generalised, log-domain weights (redo pass for chains flagged by MODE 0)
// MODE 3 / 4: MODE 0 / 1 plus the adaptor's adapt! after every transition, inside the kernel — the warm-up phase in
//         batches (the host path launches per transition because adapt! needs every transition's α; step sizes
//         and per-chain mass matrices never need another chain's data, so the kernel can do it itself).  MODE 4 is
//         the redo pass of MODE 3, as MODE 1 is of MODE 0: a chain that bails out of the linear domain at
//         transition kt saves its adaptation state and is resumed there.
// MODE 2: any sampler / criterion chosen at run time, log-domain weights (the run-time variants
//         cost registers: with them in the fast kernel (32,4) loses a wave per SIMD)
// Minimum waves per SIMD (register cap): MEASURED, not derived — the compiler left to itself takes 168 / 231 /
// 330 registers for E = 2 / 4 / 8 (3 / 2 / 1 waves); capping at 128 / 168 / 256 with a few spills to scratch
// is faster every time: cfg2 (64,2) 1.68e9 -> 1.79e9 (5 waves: 1.22e9), hier D=256 (64,4) 7.1e8 -> 1.0e9,
// D=512 (64,8) 3.1e8 -> 5.5e8 (3 waves: 2.3e8), D=2048 (256,8) 5.8e7 -> 1.08e8.
template <class T, int G, int E, int MODE, int TK>
// (round 3: the warm-up instantiations capped for 3 waves per SIMD instead of 4 — 168 VGPRs, no spills, 12 LDS slots — run at 2.24e9
// in-kernel against 2.37e9: occupancy matters more than the spills.)
__global__ __launch_bounds__((G > 256 ? G : 256), (E >= 16 ? 1 : (MODE == 2 ? (E <= 2 ? 3 : 2) : (E <= 2 ? 4 : (E <= 4 ? 3 : 2))))) void k_nuts(KP<T> p) {
  constexpr int CPW = G >= 64 ? 1 : 64 / G;  // chains per wave (G > 64: one chain per workgroup of G/64 waves)
  // A chain that owns whole waves makes every per-chain predicate wave-uniform; saying so (a ballot is uniform by
  // construction) turns the exec-mask save/restore of divergent branches into scalar branches and keeps the
  // predicates in SGPRs instead of VGPR 0/1 values.
#define AHMC_UNI(b) (CPW == 1 ? (__builtin_amdgcn_ballot_w64(b) != 0) : (b))
  // any lane of the wave?  For a chain that owns the wave the predicate is already wave-uniform (built from AHMC_UNI values):
  // testing it directly is a scalar branch, a ballot of it would re-materialise the lane mask (v_cndmask + v_cmp).
  // Measured (round 3, cfg2, A/B on one box, two runs each): whole loop 2.36e9 -> 2.43e9, warm-up 1.97e9 -> 2.06e9.
#if AHMC_SCALAR_ANY
#define AHMC_ANY(b) (CPW == 1 ? (bool)(b) : (__builtin_amdgcn_ballot_w64(b) != 0))
#else
#define AHMC_ANY(b) (__builtin_amdgcn_ballot_w64(b) != 0)
#endif
  constexpr int NCH = Chunking<T, E>::NCH, CH = Chunking<T, E>::CH;
  constexpr int SLOT_ELEMS = NCH * 64 * CH;  // elements per vector slot (= 64 * E)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwaves = blockDim.x >> 6;
  const int wib = threadIdx.x >> 6;
  const int lane64 = threadIdx.x & 63;
  int lane = (int)(threadIdx.x & (G - 1));
  int gi = G >= 64 ? 0 : lane64 / G;
  int d0 = lane * E;
  const int NLEV = p.max_depth > 1 ? p.max_depth - 1 : 1;  // pending levels 0 .. NLEV-1
  constexpr bool LINW = MODE == 0 || MODE == 3;
  constexpr bool GENERAL = MODE == 2;
  // Fast kernels of the multi-wave groups: the first merge's U-turn dot products are reduced together with the leaf's
  // energies (one barrier pair less per such leaf; cfg5 +7 %).  Measured SLOWER for one wave per chain (cfg2 2.12e9 ->
  // 1.87e9 although 216 -> 212 VALU per leapfrog), so G <= 64 keeps two reductions.
  constexpr bool FUSE_M0 = !GENERAL && G > 64;
  constexpr bool ADAPT = MODE >= 3;  // MODE 0 / 1 + adapt!(…) after every transition, inside the kernel (AdaptK)
  const bool strict = GENERAL && p.criterion == 2;
  const int NV = strict ? 3 : 2;  // vectors per pending level: A, RF (, RL)
  const int n_slots = NV * NLEV + NUTS_DORMANT + NUTS_CKPT;
  const int CK0 = NV * NLEV + NUTS_DORMANT;   // first checkpoint slot (physical = logical: behind every other slot of either numbering)
  const int n_lds_slots = p.n_lds_levels;  // (re-used field) number of vector slots held in LDS
  // LDS carve-up: [nwaves][n_lds_slots][SLOT_ELEMS] T | [nwaves][NSC][NLEV][CPW] T | [nwaves][NAT][CPW] T |
  //                [nwaves][NSI][NLEV][CPW] int | [nwaves][NAI][CPW] int
  T* lds_vec = reinterpret_cast<T*>(smem) + (size_t)wib * n_lds_slots * SLOT_ELEMS;
  T* sT = reinterpret_cast<T*>(smem) + (size_t)nwaves * n_lds_slots * SLOT_ELEMS + (size_t)wib * NUTS_NSC * NLEV * CPW;
  T* sAT = reinterpret_cast<T*>(smem) + (size_t)nwaves * (n_lds_slots * SLOT_ELEMS + NUTS_NSC * NLEV * CPW) + (size_t)wib * NUTS_NAT * CPW;
  int* sIbase = reinterpret_cast<int*>(reinterpret_cast<T*>(smem) + (size_t)nwaves * (n_lds_slots * SLOT_ELEMS + (NUTS_NSC * NLEV + NUTS_NAT) * CPW));
  int* sI = sIbase + (size_t)wib * NUTS_NSI * NLEV * CPW;
  int* sAI = sIbase + (size_t)nwaves * NUTS_NSI * NLEV * CPW + (size_t)wib * NUTS_NAI * CPW;
  // the adaptor's per-chain state of a warm-up batch lives in LDS, not in registers: it is touched once per transition,
  // and ~12 VGPRs live across the whole tree loop were the difference between the warm-up instantiation's 388 B/lane of
  // scratch and the sampling one's 180 (profiles/r2_*: 70 % VALU-busy against 90 %, twice the HBM traffic)
#define A_EPSNOM sAT[0 * CPW + gi]
#define A_DAEPS sAT[1 * CPW + gi]
#define A_DAMU sAT[2 * CPW + gi]
#define A_DAXBAR sAT[3 * CPW + gi]
#define A_DAHBAR sAT[4 * CPW + gi]
#define A_DAM sAI[0 * CPW + gi]
#define A_WVN sAI[1 * CPW + gi]
#define A_CHK sAI[2 * CPW + gi]   // the checkpoint of the re-integration: touched at the top of a doubling only, so it lives in LDS, not in a register
#define S_W(lvl) sT[((0) * NLEV + (lvl)) * CPW + gi]
#define S_SA(lvl) sT[((1) * NLEV + (lvl)) * CPW + gi]
#define S_DH(lvl) sT[((2) * NLEV + (lvl)) * CPW + gi]
#define S_NA(lvl) sI[((0) * NLEV + (lvl)) * CPW + gi]
#define S_CK(lvl) sI[((1) * NLEV + (lvl)) * CPW + gi]
  const int64_t wave_slot = (int64_t)blockIdx.x * nwaves + wib;
  Slots<T, E> sl;
  sl.lds = (AHMC_AS_LDS T*)lds_vec;
  sl.glb = (AHMC_AS_GLB T*)(p.scratch + wave_slot * (int64_t)(n_slots - n_lds_slots) * SLOT_ELEMS);
  sl.n_lds = n_lds_slots;
  sl.lane_off = (unsigned)lane64 * CH;
  const int DORM = NV * NLEV;  // first dormant slot (logical numbering: levels first, then the dormant vectors)
  // Physical slot order = order of use (the first n_lds_slots live in LDS, the rest in global scratch).  Round 3, from the
  // counters: a wave spends ≈ 16 000 quad-cycles per transition in s_waitcnt — ≈ 15 dependent memory round trips, six of them
  // the loads of the whole-tree ρ and of the other edge's r from GLOBAL scratch at the top of every doubling, because the
  // nine LDS slots went to the pending levels 0–3 and level 4.  In the fast kernels a one-leaf subtree's ρ and first-built r are
  // the same vector (the leaf's r), so level 0 needs ONE slot, and the nine become: level 0, levels 1–3 (A, RF), whole-tree ρ,
  // other edge's r — every slot the tree touches once per doubling or more often is in LDS.
  //   fast kernels : L0 | L1.A L1.RF | L2.A L2.RF | L3.A L3.RF | TREE_A | OTH_R | L4.A L4.RF … | OTH_TH OTH_G Z0_R Z0_G START_R
  //   GENERAL (run-time sampler / criterion; Classic keeps θ_first in A, Strict a third vector per level): the logical order
  const int NL_HOT = NLEV < 4 ? NLEV : 4;
  const int HOT_END = 2 * NL_HOT - 1;                                  // first slot after the hot levels
  const int COLD_DORM = HOT_END + 2 + 2 * (NLEV > 4 ? NLEV - 4 : 0);   // first slot of the remaining dormant vectors
  auto LA = [&](int l) -> int {  // level l: ρ (Classic: θ) of the pending first half
    if constexpr (GENERAL) return NV * l;
    else return l == 0 ? 0 : (l < NL_HOT ? 2 * l - 1 : HOT_END + 2 + 2 * (l - 4));
  };
  auto LRF = [&](int l) -> int {  // level l: r of the pending first half's first-built leaf (level 0, fast kernels: the same slot as A)
    if constexpr (GENERAL) return NV * l + 1;
    else return l == 0 ? 0 : (l < NL_HOT ? 2 * l : HOT_END + 3 + 2 * (l - 4));
  };
  auto DS = [&](int k) -> int {  // dormant vector k (SL_*)
    if constexpr (GENERAL) return DORM + k;
    else return k == SL_TREE_A ? HOT_END : (k == SL_OTH_R ? HOT_END + 1
                : COLD_DORM + (k == SL_OTH_TH ? 0 : k == SL_OTH_G ? 1 : k == SL_Z0_R ? 2 : k == SL_Z0_G ? 3 : 4));
  };
  const bool classic = GENERAL && p.criterion == 0;
  const bool slice = GENERAL && p.sampler == 2;

  // One chunk of 64/G chains per wave, one wave per workgroup: the hardware dispatcher is the work
  // queue.  (A persistent per-wave loop over chunks was measured first: it makes every prologue and
  // epilogue value loop-invariant, the compiler hoists them all and the kernel needs 230+ VGPRs.)
  //
  // The wave runs p.n_trans consecutive transitions of its chains before it exits.  Tree sizes
  // are heavy-tailed (cfg2 after adaptation: mean 31 leaves, but every transition has a chain with
  // 700-1000 leaves, a strictly serial ~2.3 ms); with one transition per launch the whole GPU waits
  // for that chain each time.  Chains are independent, so batching transitions removes the
  // per-transition barrier and the tail is paid once per launch.
  const unsigned int chunk = G > 64 ? blockIdx.x : blockIdx.x * nwaves + wib;
  if (chunk >= p.n_chunks) return;
  // Dispatch order: workgroups start in blockIdx order, so slot i takes chain order[i] — sorted by step size,
  // smallest ϵ (longest trees) first — and the stragglers of a launch are the cheap chains (LPT scheduling);
  // lockstep neighbours (G < 64) then also have similar trees.
  const int64_t slot = (int64_t)chunk * CPW + gi;
  const int64_t c = slot < p.N ? (p.order ? (int64_t)p.order[slot] : slot) : p.N;
  int64_t cc = c < p.N ? c : 0;  // out-of-range groups shadow chain 0 and never write
  // redo pass: only the chains flagged by the linear-domain pass, from the transition they bailed at
  const int kt0 = p.redo_only ? p.redo[cc] - 1 : 0;
  bool active = c < p.N && kt0 >= 0;
  if (!AHMC_ANY(active)) return;  // (redo pass: the common case)
  T minv[E];
  load_minv<T, E>(p, cc, d0, minv);
  T th_cur[E];  // the chain's position, carried in registers from one transition to the next
  load_vec<T, E>(th_cur, p.th(), cc * p.D, d0, p.D, T(0));
  // in-kernel adaptation state (MODE 3): dual averaging of this chain, its nominal step size, the Welford count
  const AdaptK<T>* ak0 = ADAPT ? static_cast<const AdaptK<T>*>(p.adaptk) : nullptr;
  const AdaptK<T>* ak = ak0;

  // what adapt! does at batch-local transition kt2 (the same for every chain): push / update / reset of the variance
  // estimator and reset of the dual averaging — `adapt` in ahmc_api.hip, stan_adaptor.jl:137-159
  auto schedule = [&](int kt2, bool& do_push, bool& do_update, bool& wv_reset, bool& dareset) {
    do_push = do_update = wv_reset = dareset = false;
    if (ak->kind == AHMC_ADAPT_STAN) {
      const int64_t si = ak->stan_i0 + kt2 + 1;
      const bool in_window = si >= ak->window_start && si <= ak->window_end;
      bool window_end = false;
      for (int k2 = 0; k2 < ak->n_splits; ++k2) window_end = window_end || ak->splits[k2] == si;
      if (in_window && ak->has_mm) { do_push = true; do_update = window_end; }
      if (window_end) { dareset = true; wv_reset = ak->has_mm != 0; }
    } else if (ak->has_mm) {
      do_push = true;
      do_update = true;
    }
    if (ak->pooled) { do_update = false; wv_reset = false; }  // one (D,) estimate over all chains: pooled on the host side
  };
  auto save_adapt_state = [&]() {  // dual averaging + nominal step size of this chain (the Welford vectors are always in memory)
    if (ak->has_ss && lane == 0) {
      const T e = A_DAEPS;
      ak->da_m[cc] = A_DAM; ak->da_eps[cc] = e; ak->da_mu[cc] = A_DAMU; ak->da_xbar[cc] = A_DAXBAR; ak->da_Hbar[cc] = A_DAHBAR;
      ak->eps_nom[cc] = e;
    }
  };
  if constexpr (ADAPT) {
    A_EPSNOM = p.eps_nom()[cc];
    if (ak->has_ss) { A_DAM = ak->da_m[cc]; A_DAEPS = ak->da_eps[cc]; A_DAMU = ak->da_mu[cc]; A_DAXBAR = ak->da_xbar[cc]; A_DAHBAR = ak->da_Hbar[cc]; }
    int wv_n = (int)ak->wv_n0;
    for (int kt2 = 0; kt2 < kt0; ++kt2) {  // redo pass: the Welford count at the transition this chain resumes at
      bool a1, a2, a3, a4;
      schedule(kt2, a1, a2, a3, a4);
      if (a1) wv_n += 1;
      if (a3) wv_n = 0;
    }
    A_WVN = wv_n;
  }

#if AHMC_WAVE_TIMELINE
  // measurement build only (scripts/wave_timeline.py): when did this wave run, how many leaf steps did it make, how many of its
  // chains were still building in each of them, how many re-integration steps
  const unsigned long long tl_t0 = wall_clock64();
  unsigned long long tl_steps = 0, tl_alive = 0, tl_re = 0, tl_trans = 0;
#endif
#if AHMC_LEAF_PROF
  unsigned lp_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned lp_c = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__builtin_readcyclecounter());
#endif
  for (int kt = 0; kt < p.n_trans; ++kt) {
    // Make the per-lane indices opaque once per transition: otherwise every address and every
    // constant derived from them is loop-invariant w.r.t. this loop, gets hoisted out of it and
    // stays live through the whole tree (measured: +50 VGPRs, one wave per SIMD less).
    asm volatile("" : "+v"(cc), "+v"(d0), "+v"(lane), "+v"(gi), "+v"(sl.lane_off));
    const bool on = AHMC_UNI(active && kt >= kt0);
    if (!AHMC_ANY(on)) continue;
    // ---- transition prologue (src/sampler.jl:54-57): jitter, fresh momentum, caches.  The standard
    // normals come from k_normals (same Philox stream): keeping the f64 Box–Muller out of this
    // kernel saves ~25 VGPRs at its register peak ----
    Point<T, E> cur;
    copy_vec(cur.th, th_cur);
    Rng rng = make_rng(p, cc);
    rng.iter = p.iteration + (uint32_t)kt;
    // nominal ϵ of this transition: the adaptor's current one (LDS) while adapting, else the context's (re-read per
    // transition: two VGPRs fewer across the tree loop than carrying it)
    T eps_nom_t;
    if constexpr (ADAPT) eps_nom_t = A_EPSNOM; else eps_nom_t = p.eps_nom()[cc];
    const T eps = chain_eps_from(p, rng, eps_nom_t);
    momentum_from_normals<T, E>(p, p.znorm + (int64_t)kt * p.D * p.N, cc, d0, cur.r);
    fill_caches<T, G, E, TK>(cur, minv, p.tp, lane, d0);
    // (θ0 of this transition — what the epilogue's re-integration and the redo pass start from — is already in p.th(): the
    // launch's start point at kt = 0, the previous transition's store_point after it.  Round 4: the redundant `if (on && kt > 0)`
    // store that stood here was also where the register allocator parked the spills of the chain index and of the lane's first
    // dimension in two instantiations — under the block's narrowed exec mask, skipped altogether at kt = 0: a memory fault on
    // the MI355X, now also what isa_check's second pass looks for.)
    const T H0 = -(cur.lp + cur.lk);
    DrawStreamT<(G >= 64)> ds;
    ds.init(rng);
    // both edges of the one-leaf tree are z0 (src/trajectory.jl:682-685)
    sl.store(DS(SL_OTH_TH), cur.th);
    sl.store(DS(SL_OTH_R), cur.r);
    sl.store(DS(SL_OTH_G), cur.g);
    sl.store(DS(SL_Z0_R), cur.r);
    sl.store(DS(SL_Z0_G), cur.g);
    if (!classic) sl.store(DS(SL_TREE_A), cur.r);  // ρ = r0 (TurnStatistic, :461-463)
    T w_tree, sa_tree = 0, dh_tree = 0, lu = 0;
    int na_tree = 0, ck_tree = 0;
    if (slice) {
      lu = -H0 - (T)ds.randexp();  // SliceTS(rng, z0) (:144-145)
      w_tree = 1;
    } else {
      w_tree = LINW ? T(1) : T(0);  // MultinomialTS(rng, z0): ℓw = 0 (:155); LINW carries W = exp(ℓw)
    }
    bool cur_is_left = false;
    int pos_cur = 0, pos_oth = 0;  // leaf index (signed distance from z0) of the two edges
#if AHMC_CKPT
    // Checkpoint of the re-integration (round 5).  The candidate is carried as a leaf index and re-integrated at the end — from z0 that
    // is ≈ 0.5 leapfrogs per leaf step (0.37 / 0.51 / 0.48 measured on cfg2 / cfg3 / cfg5: 9 / 14 / 18 % of a leaf step's cycles).  The state
    // of the EDGE a doubling grows from is in registers when the doubling starts, and every leaf of that doubling lies beyond it: saved
    // there (three stores to cold global slots, off the chain of dependent stages), it is where the replay starts if the tree's
    // candidate ends up in that subtree — the same leapfrogs from the same state, so the same bits, and half as many of them.
    // chk = (position << 1) | triple of the COMMITTED checkpoint (position 0: none, replay from z0); a doubling saves into the other triple.
    // (not in the Float32 warm-up kernels: with it the register allocator parks a spill under a narrowed exec mask in k_nuts<float,32,4,3,·> and
    // <float,16|32,2,3,·>, which isa_check refuses; the replay there starts at z0 as before — same results either way)
#ifndef AHMC_CKPT_MIN_JW
#define AHMC_CKPT_MIN_JW 2
#endif
    constexpr int CKPT_MIN_JW = AHMC_CKPT_MIN_JW;   // subtrees of 4 leaves and more
    // … and not where ONE chain fills ONE wave (G = 64: cfg2's (64,2), and (64,4), (64,8)): those kernels are bound by VALU issue at their
    // register caps, the replay is ≈ 3 % of their instructions, and the checkpoint's extra live value costs them spilled registers — measured
    // with / without: (64,2) 2.851e9 / 2.881e9, (64,4) at D = 256 1.587e9 / 1.603e9, (64,8) at D = 512 7.78e8 / 8.08e8 leapfrog/s.  The chains
    // that share a wave (cfg3 +7 %) and the multi-wave chains (cfg5 +6.7 %) are bound by the latency of their dependent stages, where half
    // the replay is half the time (profiles/r5_experiments.md r5i, r5j).
    constexpr bool CKPT = !(sizeof(T) == 4 && ADAPT) && (G != 64 || AHMC_CKPT_G64);
    if constexpr (CKPT) A_CHK = 0;
#else
    constexpr int CKPT_MIN_JW = 2;
    constexpr bool CKPT = false;
#endif
    bool numerical = false;
    bool redo = false;  // LINW only: a weight came too close to overflow
    int depth = 0;
    bool done = !on;

    // (round 4, measured and dropped: a wave raising its issue priority — s_setprio 3 — from doubling 5 / 7 / 9 of a tree to the end of
    // the transition, so that the deep trees every launch waits for would step faster: cfg3 2.47 / 2.52 / 2.56e9 against 2.49e9
    // without, cfg2 and cfg5 unchanged — within the run-to-run spread.  A lone deep tree is bound by the latency of its own chain of
    // dependent reductions, not by the issue slots it shares.)
    AHMC_LP_TICK(6)
    for (int jw = 0; jw < p.max_depth; ++jw) {  // doubling loop (:691-723), wave-uniform
      if (!AHMC_ANY(!done)) break;
      AHMC_LP_COUNT(10, 1)
      // ---- direction (:693) and edge selection ----
      bool vleft = false;
      if (!done) vleft = AHMC_UNI(ds.boolean());
      const int v = vleft ? -1 : 1;
      const bool need_swap = AHMC_UNI(!done && (vleft != cur_is_left));
      if (AHMC_ANY(need_swap)) {
        if (need_swap) {
          if (jw > 0) {  // at jw == 0 both edges are z0
            Point<T, E> t;
            sl.load(DS(SL_OTH_TH), t.th);
            sl.load(DS(SL_OTH_R), t.r);
            sl.load(DS(SL_OTH_G), t.g);
            sl.store(DS(SL_OTH_TH), cur.th);
            sl.store(DS(SL_OTH_R), cur.r);
            sl.store(DS(SL_OTH_G), cur.g);
            copy_vec(cur.th, t.th);
            copy_vec(cur.r, t.r);
            copy_vec(cur.g, t.g);
          }
          int ti = pos_cur; pos_cur = pos_oth; pos_oth = ti;
          cur_is_left = vleft;
        }
      }
      if (strict && AHMC_ANY(!done)) {
        if (!done) sl.store(DS(SL_START_R), cur.r);  // r of the edge the subtree grows from
      }
      if (CKPT && jw >= CKPT_MIN_JW) {   // (wave-uniform)
        // A chain that owns its wave saves only an edge that is not z0 itself; chains that SHARE a wave all store (a finished chain, or an
        // edge at z0, writes a triple nobody will read): no exec-masked block here for the register allocator to park a spill in
        // (isa_check refused the masked form in the f32 warm-up kernels)
        bool ck_save = true;
        if constexpr (CPW == 1) ck_save = AHMC_UNI(!done && pos_cur != 0);
        if (ck_save) {
          const unsigned xo = (unsigned)(((A_CHK & 1) ^ 1) * 3 * SLOT_ELEMS);
          sl.store_cold(CK0 + 0, xo, cur.th);
          sl.store_cold(CK0 + 1, xo, cur.r);
          sl.store_cold(CK0 + 2, xo, cur.g);
        }
      }
      // ---- build the subtree of 2^jw leaves (:626-675) ----
      const uint32_t nleaf = 1u << jw;
      bool alive = !done;      // still adding leaves to this subtree
      bool sub_term = false;
      T A_c[E], RF_c[E];
      T w_c = 0, sa_c = 0, dh_c = 0;
      int na_c = 0, ck_c = 0;
      AHMC_LP_TICK(5)
      for (uint32_t leaf = 1; leaf <= nleaf; ++leaf) {
        if (!AHMC_ANY(alive)) break;
        AHMC_LP_COUNT(8, 1)
#if AHMC_WAVE_TIMELINE
        tl_steps += 1;
        tl_alive += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(alive && lane == 0));
#endif
        int merged = 0;
        // merges after this leaf: one per trailing zero bit of `leaf` (:649-673); the count is wave-uniform
        const int nm = __builtin_ctz(leaf);
        T dots_m0[2] = {0, 0}, RF_m0[E];  // fast kernels: U-turn dot products / first-built r of the FIRST merge
        if (alive) {
          // leaf: one leapfrog step in direction v (:638-647)
          T ne_leaf = 0;
          if constexpr (GENERAL) {
            leapfrog_step<T, G, E, TK, true>(cur, minv, v > 0 ? eps : -eps, p.tp, p.lf, lane, d0, 1, 1);
          } else if (FUSE_M0 && nm > 0) {
            // a merge follows: its two dot products ride in the leaf's all-reduce (they only need the new r)
            leapfrog_step_plus2<T, G, E, TK>(cur, minv, v > 0 ? eps : -eps, p.tp, lane, d0, dots_m0, [&](T& s0, T& s1) {
              T A_p[E];
              sl.load(LA(0), A_p);  // level 0: one slot (ρ = first-built r = the leaf's r)
              copy_vec(RF_m0, A_p);
              s0 = 0;
              s1 = 0;
#pragma unroll
              for (int e = 0; e < E; ++e) {
                A_c[e] = A_p[e] + cur.r[e];  // ρ = ρ_left + ρ_right
                s0 += A_c[e] * (minv[e] * RF_m0[e]);
                s1 += A_c[e] * (minv[e] * cur.r[e]);
              }
            }, AHMC_HIER_PRE && leaf > 1);
          } else if constexpr (G > 64) {
            // multi-wave chains keep the (ℓπ, ℓκ) pair reduction.  Round 2: with the single-value form the (128,8) instantiation
            // returned wrong candidates — a register-allocation artefact (a spill stored under the lane-0 mask of the cross-wave
            // exchange; the build now scans for it).  Round 3 measured the single-value leaf with an exchange that has
            // no narrowed block (every lane stores): parity-green, but cfg5 5.92e7 -> 4.70e7 leapfrog/s (the all-lane store alone:
            // 5.36e7) — more scratch traffic in the leaf loop than the barrier pair it saves.  Not taken.
            // (round 6: from the second leaf of a doubling on, μ and log τ of the new position were published by the leaf before — no
            // broadcast exchange of their own)
            leapfrog_step<T, G, E, TK, false>(cur, minv, v > 0 ? eps : -eps, p.tp, p.lf, lane, d0, 1, 1, AHMC_HIER_PRE && leaf > 1);
            ne_leaf = cur.lp + cur.lk;
          } else {
#if AHMC_LEAF_PROF
            {
              const T e_ = v > 0 ? eps : -eps, eh_ = e_ / 2;
#pragma unroll
              for (int e = 0; e < E; ++e) cur.r[e] = cur.r[e] - eh_ * cur.g[e];
#pragma unroll
              for (int e = 0; e < E; ++e) cur.th[e] = cur.th[e] + e_ * (minv[e] * cur.r[e]);
              const T part_ = target_eval<T, G, E, TK>(p.tp, cur.th, cur.g, lane, d0);
#pragma unroll
              for (int e = 0; e < E; ++e) cur.r[e] = cur.r[e] - eh_ * cur.g[e];
              const T kin_ = kinetic_partial(cur.r, minv);
              T sv_[1] = {part_ - kin_ / 2};
              asm volatile("" : "+v"(sv_[0]));   // the partial is complete before the stamp
              AHMC_LP_TICK(0)
              leapfrog_allsum<G, TK>(sv_);
              asm volatile("" : "+v"(sv_[0]));
              AHMC_LP_TICK(1)
              ne_leaf = is_finite(sv_[0]) ? sv_[0] : -Lim<T>::inf();
            }
#else
            ne_leaf = leapfrog_step_ne<T, G, E, TK>(cur, minv, v > 0 ? eps : -eps, p.tp, lane, d0);
#endif
          }
#if AHMC_LEAF_PROF
          if (GENERAL || G > 64) AHMC_LP_TICK(0)   // (the other leaf forms: the whole leapfrog incl. its reduction under [0])
#endif
          pos_cur += v;
          const T ne = (GENERAL || (FUSE_M0 && nm > 0)) ? cur.lp + cur.lk : ne_leaf;  // neg_energy(z′)
          const T dH = -ne - H0;
          // α′ = exp(min(0, −ΔH)) (:641-643).  Evaluated by the SAME function as the linear-domain kernels' min(1, W) — which is
          // this value bit for bit there — so that a chain the log-domain redo pass carries through the rest of a launch keeps the
          // acceptance rates, and with them the dual averaging, of the launch lengths that never hand it over (round 4: a warm-up of
          // 60 transitions in one launch left the per-iteration path on every chain that had met −ΔH > 600 once — the device
          // library's exp and the table form differ in the last bit now and then, and dual averaging doubles that every iteration)
          T sa_leaf = 0;
          if constexpr (!LINW) sa_leaf = alpha_from_logweight<T, (CPW == 1)>(H0 + ne);
          ck_c = pos_cur;
          if (slice) {
            w_c = (lu <= ne) ? T(1) : T(0);
            sub_term = AHMC_UNI(!(lu < p.delta_max + ne));  // Termination(::SliceTS, ...) (:500-502)
          } else {
            if constexpr (LINW) {
              // multinomial weights in the linear domain: W = exp(ℓw), ℓw = H0 - H′ = -ΔH.  Then
              // α′ = exp(min(0, -ΔH)) = min(1, W) bit for bit, subtree weights add, and the
              // progressive-sampling tests ℓw < ℓw₁ + Exp(1) become u·W < W₁ — one exp per leaf
              // instead of exp + log1p + log per merge.  Valid while no weight can overflow; a
              // chain that meets -ΔH > 600 is flagged and redone by the log-domain kernel.
              const T lw = H0 + ne;
              w_c = leaf_weight_exp<(CPW == 1)>(lw);
              sa_leaf = w_c >= T(1) ? T(1) : w_c;  // exp(min(0, ℓw)) = min(1, W), NaN-propagating like Julia's min
              redo = redo || AHMC_UNI(lw > (sizeof(T) == 4 ? T(60) : T(LINW_LIMIT)));
            } else {
              w_c = H0 + ne;
            }
            sub_term = AHMC_UNI(!(-H0 < p.delta_max + ne));  // Termination(::MultinomialTS, ...) (:503-507)
          }
          numerical = numerical || sub_term;
#if AHMC_RUNNING_STATS
          sa_c = sa_c + sa_leaf;                                     // over the leaves of this doubling, in build order
          na_c = na_c + 1;
          // v > 0 ? maxabs(dh_c, dH) : maxabs(dH, dh_c) (maxabs keeps the right-hand one on a tie, :526).  As a select between the two calls
          // the compiler evaluated both (two compares, six v_cndmask per leaf: round 5's census of the leaf loop); a chain that owns its wave
          // has a wave-uniform direction, so it is a scalar branch around ONE of them (the empty asm keeps it a branch).
          if constexpr (CPW == 1 && AHMC_LEAF_MICRO) {
            if (v > 0) { dh_c = maxabs(dh_c, dH); asm volatile(""); }
            else { dh_c = maxabs(dH, dh_c); asm volatile(""); }
          } else {
            dh_c = v > 0 ? maxabs(dh_c, dH) : maxabs(dH, dh_c);
          }
#else
          sa_c = sa_leaf;
          na_c = 1;
          dh_c = dH;
#endif
          // a one-leaf subtree has ρ = r (Classic: θ) and first-built r = r of this leaf.  The general kernel copies
          // them into A_c / RF_c here; the fast kernels read cur.r in their place until the first merge (below)
          if constexpr (GENERAL) {
            if (classic) copy_vec(A_c, cur.th); else copy_vec(A_c, cur.r);
            copy_vec(RF_c, cur.r);
          }
        }
        AHMC_LP_TICK(2)
        // A_in / RF_in: ρ (Classic: θ of the first-built leaf) and first-built r of the half just completed;
        // pre: the first merge of a fast kernel — A_c, the dot products and RF_p were made with the leaf (above)
        auto merge_level = [&](const int lvl, const T (&A_in)[E], const T (&RF_in)[E], auto pre) {
          {
            T A_p[E], RF_p[E];
            if constexpr (!pre.value) {
              sl.load(LA(lvl), A_p);
              if (!GENERAL && lvl == 0) copy_vec(RF_p, A_p); else sl.load(LRF(lvl), RF_p);
            }
            const T w_p = S_W(lvl);
            // combine(rng, sampler′, sampler′′): `first` = the half built first (:178-195)
            bool keep_first;
            T w_new;
            if (slice) {
              w_new = w_p + w_c;
              keep_first = AHMC_UNI(w_new * (T)ds.uniform() < w_p);
            } else {
              if constexpr (LINW) w_new = w_p + w_c; else w_new = logaddexp(w_p, w_c);
              if constexpr (LINW) keep_first = AHMC_UNI((T)ds.uniform() * w_new < w_p); else keep_first = AHMC_UNI(w_new < w_p + (T)ds.randexp());
            }
            if (keep_first) ck_c = S_CK(lvl);
            w_c = w_new;
            // combine(treeleft, treeright) (:533-542); position order matters for maxabs only
#if !AHMC_RUNNING_STATS
            sa_c = S_SA(lvl) + sa_c;
            na_c = S_NA(lvl) + na_c;
            const T dh_p = S_DH(lvl);
            dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
#endif
            // isterminated(tc, h, tree′, tleft, tright) on the merged subtree (:551-617)
            if (classic) {
              // ends: first-built leaf (A_p, RF_p) and the current leaf; Δθ = θ_right − θ_left
              T dots[2] = {0, 0};
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
            } else if (!strict) {
              T dots[2] = {dots_m0[0], dots_m0[1]};
              if constexpr (!pre.value) {
                dots[0] = 0;
                dots[1] = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                  A_c[e] = A_p[e] + A_in[e];  // ρ = ρ_left + ρ_right
                  dots[0] += A_c[e] * (minv[e] * RF_p[e]);
                  dots[1] += A_c[e] * (minv[e] * cur.r[e]);
                }
                // (round 4, measured and dropped: the signs of the two sums from a single-precision all-reduce — the lane partials made in
                // double, converted, four values in one float reduction with DPP-fused adds, 27 instructions instead of 34 — with
                // the double reduction only inside an error band of 2⁻¹⁹ Σ(|p₀| + |p₁|): decisions identical, every GPU test green,
                // and not faster: cfg2 draws 3.228e9 against 3.243e9 in-kernel, cfg3 2.41e9 against 2.54e9 whole loop.  The chain of
                // dependent stages is as long as before and the DPP hazards leave wait states where the double version has work.)
                group_allsum<G>(dots);
              }
              sub_term = AHMC_UNI(dots[0] <= 0) || AHMC_UNI(dots[1] <= 0);  // generalised_uturn_criterion (:619-621)
            } else {
              // StrictGeneralisedNoUTurn (:579-617).  F = the pending (first-built) half, S = the half
              // just completed; in built order the two extra checks are symmetric in the direction:
              //   (ρ_F + r_S.first ; ends F.first, S.first)   and   (r_F.last + ρ_S ; ends F.last, S.last)
              T RL_p[E];
              sl.load(NV * lvl + 2, RL_p);
              T dots[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const T rho = A_p[e] + A_in[e];
                const T rho2 = A_p[e] + RF_in[e];
                const T rho3 = RL_p[e] + A_in[e];
                dots[0] += rho * (minv[e] * RF_p[e]);
                dots[1] += rho * (minv[e] * cur.r[e]);
                dots[2] += rho2 * (minv[e] * RF_p[e]);
                dots[3] += rho2 * (minv[e] * RF_in[e]);
                dots[4] += rho3 * (minv[e] * RL_p[e]);
                dots[5] += rho3 * (minv[e] * cur.r[e]);
                A_c[e] = rho;
              }
              group_allsum<G>(dots);
              sub_term = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
            }
            if constexpr (pre.value) copy_vec(RF_c, RF_m0); else copy_vec(RF_c, RF_p);
            merged = lvl + 1;
            AHMC_LP_COUNT(9, 1)
          }
        };
        if constexpr (GENERAL) {
          for (int lvl = 0; lvl < nm; ++lvl) {
            const bool m = AHMC_UNI(alive && !sub_term);
            if (!AHMC_ANY(m)) break;
            if (m) merge_level(lvl, A_c, RF_c, std::false_type{});
          }
        } else if (nm > 0) {
          // first merge peeled: the completed half is the single leaf `cur` (its vector work was done with the leaf);
          // later merges carry A_c / RF_c
          // (round 4, measured and dropped: TWO merge levels per all-reduce — the ρ of level l + 1 is a vector sum, known without
          // the reduction, so the four dot products of a leaf's first two merges can be reduced together (45 VALU and one chain of
          // dependent stages instead of 2 × 30 and two), the scalar work then run in the reference's order.  Bit-identical, and
          // slower: the four vectors live across the reduction push k_nuts<double,64,2,0,0> from 12 to 128 B of scratch per
          // lane, inside the leaf loop — cfg2 2.89e9 → 2.41e9, cfg3 2.49e9 → 2.25e9.)
          bool m = AHMC_UNI(alive && !sub_term);
          if (AHMC_ANY(m)) {
            if (m) merge_level(0, cur.r, cur.r, std::bool_constant<FUSE_M0>{});
            for (int lvl = 1; lvl < nm; ++lvl) {
              m = AHMC_UNI(alive && !sub_term);
              if (!AHMC_ANY(m)) break;
              if (m) merge_level(lvl, A_c, RF_c, std::false_type{});
            }
          }
        }
        AHMC_LP_TICK(3)
        if (alive && sub_term) {
          // enclosing unfinished subtrees still absorb the statistics of their first halves
          // (tree′ = combine(treeleft, treeright) at every level that is a second half, :666)
#if !AHMC_RUNNING_STATS
          const uint32_t pend = ((leaf - 1u) >> merged) << merged;
          for (int q = merged; (pend >> q) != 0u; ++q) {
            if ((pend >> q) & 1u) {
              sa_c = S_SA(q) + sa_c;
              na_c = S_NA(q) + na_c;
              const T dh_p = S_DH(q);
              dh_c = v > 0 ? maxabs(dh_p, dh_c) : maxabs(dh_c, dh_p);
            }
          }
#endif
          alive = false;
        } else if (alive && leaf < nleaf) {
          // park the finished level-nm subtree until its sibling is built
          if (GENERAL || nm > 0) {
            sl.store(LA(nm), A_c);
            sl.store(LRF(nm), RF_c);
          } else {
            sl.store(LA(0), cur.r);  // (level 0 of a fast kernel: ρ and first-built r are this one vector)
          }
          if (strict) sl.store(NV * nm + 2, cur.r);  // r of its last-built leaf
          S_W(nm) = w_c;
#if !AHMC_RUNNING_STATS
          S_SA(nm) = sa_c;
          S_DH(nm) = dh_c;
          S_NA(nm) = na_c;
#endif
          S_CK(nm) = ck_c;
        }
        AHMC_LP_TICK(4)
      }
      // ---- top level of the doubling loop (:708-722) ----
      if constexpr (!GENERAL) {
        if (jw == 0) {  // the one-leaf subtree of the first doubling never went through a merge
          copy_vec(A_c, cur.r);
          copy_vec(RF_c, cur.r);
        }
      }
      if (!done) {
        if (!sub_term) {
          ++depth;
          bool acc;  // mh_accept(rng, sampler, sampler′): biased progressive sampling (:202-206)
          if (slice) acc = w_tree * (T)ds.uniform() < w_c;
          else if constexpr (LINW) acc = AHMC_UNI((T)ds.uniform() * w_tree < w_c);
          else acc = AHMC_UNI(w_tree < w_c + (T)ds.randexp());
          if (acc) {
            ck_tree = ck_c;
            if constexpr (CKPT) {
              const int start = pos_cur - v * (int)nleaf;   // the edge this (complete) subtree grew from
              int chk = A_CHK;
              if (jw >= CKPT_MIN_JW && start != 0) chk = (start << 1) | ((chk & 1) ^ 1);   // saved above: commit it
              else if (((chk >> 1) > 0) != (v > 0) || (chk >> 1) == 0) chk = chk & 1;        // an older checkpoint on the other side: none
              // (an older one on the same side lies between z0 and the new candidate: still a valid, if distant, start)
              A_CHK = chk;
            }
          }
        }
        sa_tree = sa_tree + sa_c;
        na_tree = na_tree + na_c;
        dh_tree = v < 0 ? maxabs(dh_c, dh_tree) : maxabs(dh_tree, dh_c);
        if constexpr (LINW) w_tree = w_tree + w_c; else w_tree = slice ? w_tree + w_c : logaddexp(w_tree, w_c);
        // isterminated(tc, h, tree, tleft, tright) on the whole tree; its edges are `cur` and the dormant one
        bool turn;
        T oth_r[E];
        sl.load(DS(SL_OTH_R), oth_r);
        // (round 3: requesting the other edge's θ and g — global scratch — here, in flight during the reduction below, for the change
        // of direction that follows half of the doublings, was measured: 2.645e9 against 2.666e9 whole loop; not taken)
        if (classic) {
          T oth_th[E];
          sl.load(DS(SL_OTH_TH), oth_th);
          T dots[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < E; ++e) {
            T thl = cur_is_left ? cur.th[e] : oth_th[e], thr = cur_is_left ? oth_th[e] : cur.th[e];
            T rl = cur_is_left ? cur.r[e] : oth_r[e], rr = cur_is_left ? oth_r[e] : cur.r[e];
            T dth = thr - thl;
            dots[0] += dth * (minv[e] * (-rl));
            dots[1] += (-dth) * (minv[e] * rr);
          }
          group_allsum<G>(dots);
          turn = (dots[0] >= 0) || (dots[1] >= 0);
        } else {
          T A_tree[E];
          sl.load(DS(SL_TREE_A), A_tree);
          if (!strict) {
            T dots[2] = {0, 0};
#pragma unroll
            for (int e = 0; e < E; ++e) {
              A_tree[e] = A_tree[e] + A_c[e];
              dots[0] += A_tree[e] * (minv[e] * cur.r[e]);
              dots[1] += A_tree[e] * (minv[e] * oth_r[e]);
            }
            group_allsum<G>(dots);
            turn = AHMC_UNI(dots[0] <= 0) || AHMC_UNI(dots[1] <= 0);
          } else {
            // strict at the top: (ρ_tree + r_sub.first ; ends other edge, sub.first) and
            //                    (r_start + ρ_sub ; ends start edge, current edge)
            T rs[E];
            sl.load(DS(SL_START_R), rs);
            T dots[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const T rho = A_tree[e] + A_c[e];
              const T rho2 = A_tree[e] + RF_c[e];
              const T rho3 = rs[e] + A_c[e];
              dots[0] += rho * (minv[e] * cur.r[e]);
              dots[1] += rho * (minv[e] * oth_r[e]);
              dots[2] += rho2 * (minv[e] * oth_r[e]);
              dots[3] += rho2 * (minv[e] * RF_c[e]);
              dots[4] += rho3 * (minv[e] * rs[e]);
              dots[5] += rho3 * (minv[e] * cur.r[e]);
              A_tree[e] = rho;
            }
            group_allsum<G>(dots);
            turn = (dots[0] <= 0) || (dots[1] <= 0) || (dots[2] <= 0) || (dots[3] <= 0) || (dots[4] <= 0) || (dots[5] <= 0);
          }
          sl.store(DS(SL_TREE_A), A_tree);
        }
        if (sub_term || turn) done = true;
      }
      AHMC_LP_TICK(5)
    }

    // ---- Transition(zcand, stats) (:725-741): re-integrate from z0 to the candidate leaf ----
    {
      // `ce` is the chain index made opaque to the optimiser: otherwise every array address of the
      // prologue is kept alive (2 VGPRs each) across the whole tree loop just to be reused here
      int64_t ce = cc;
      asm volatile("" : "+v"(ce));
      Point<T, E> zc;
      int chk = 0;
      if constexpr (CKPT) chk = A_CHK;
      const int cp = chk >> 1;
      const bool from_ck = CKPT && AHMC_UNI(on && cp != 0);
      if constexpr (!CKPT) {
        load_vec<T, E>(zc.th, p.th(), ce * p.D, d0, p.D, T(0));
        sl.load(DS(SL_Z0_R), zc.r);
        sl.load(DS(SL_Z0_G), zc.g);
      } else if constexpr (CPW == 1) {   // (wave-uniform: a scalar branch)
        if (from_ck) {
          const unsigned xo = (unsigned)((chk & 1) * 3 * SLOT_ELEMS);
          sl.load_cold(CK0 + 0, xo, zc.th);
          sl.load_cold(CK0 + 1, xo, zc.r);
          sl.load_cold(CK0 + 2, xo, zc.g);
        } else {
          load_vec<T, E>(zc.th, p.th(), ce * p.D, d0, p.D, T(0));
          sl.load(DS(SL_Z0_R), zc.r);
          sl.load(DS(SL_Z0_G), zc.g);
        }
      } else {   // chains that share a wave: both starts are read and the chain's own is selected — no exec-masked block (see the save)
        Point<T, E> zk;
        const unsigned xo = (unsigned)((chk & 1) * 3 * SLOT_ELEMS);
        load_vec<T, E>(zc.th, p.th(), ce * p.D, d0, p.D, T(0));
        sl.load(DS(SL_Z0_R), zc.r);
        sl.load(DS(SL_Z0_G), zc.g);
        sl.load_cold(CK0 + 0, xo, zk.th);
        sl.load_cold(CK0 + 1, xo, zk.r);
        sl.load_cold(CK0 + 2, xo, zk.g);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          zc.th[e] = from_ck ? zk.th[e] : zc.th[e];
          zc.r[e] = from_ck ? zk.r[e] : zc.r[e];
          zc.g[e] = from_ck ? zk.g[e] : zc.g[e];
        }
      }
      const int dist = from_ck ? ck_tree - cp : ck_tree;
      const int steps = on ? (dist < 0 ? -dist : dist) : 0;
      const T es = ck_tree < 0 ? -eps : eps;
      for (int s = 0;; ++s) {
        const bool go = AHMC_UNI(s < steps);
        if (!AHMC_ANY(go)) break;
#if AHMC_WAVE_TIMELINE
        tl_re += 1;
#endif
        if (go) leapfrog_core<T, G, E, TK, GENERAL>(zc, minv, es, p.tp, p.lf, lane, d0);
      }
      if (on && redo) {
        // stop here: the log-domain kernel resumes this chain at transition kt (its θ0 is in p.th)
        if (lane == 0) p.redo[ce] = kt + 1;
        if constexpr (ADAPT) save_adapt_state();  // as of before this transition: the redo pass resumes from it
        active = false;
      } else if (on) {
        if (p.redo_only && lane == 0 && kt == p.n_trans - 1) p.redo[ce] = 0;
        fill_caches<T, G, E, TK>(zc, minv, p.tp, lane, d0);  // ℓπ, -∇ℓπ, ℓκ of the candidate
        store_point<T, E>(p, ce, d0, lane, zc);
        const T H = -(zc.lp + zc.lk);
        if (lane == 0) {
          p.eps_cur()[ce] = eps;
          p.st_nsteps()[ce] = na_tree;
          p.st_accept()[ce] = 1;
          p.st_accrate()[ce] = sa_tree / (T)na_tree;
          p.st_logdens()[ce] = zc.lp;
          p.st_H()[ce] = H;
          p.st_Herr()[ce] = H - H0;
          p.st_maxHerr()[ce] = dh_tree;
          p.st_depth()[ce] = depth;
          p.st_numerr()[ce] = numerical ? 1 : 0;
        }
        accumulate<T, E>(p, ce, d0, lane, zc.th, na_tree, numerical ? 1 : 0, H);
        if (p.samples_out) store_vec<T, E>(zc.th, p.samples_out + (int64_t)kt * p.D * p.N, ce * p.D, d0, p.D);
        copy_vec(th_cur, zc.th);
        if constexpr (ADAPT) {
#if AHMC_ADAPT_REREAD
          // The adaptor's argument block is read HERE, once per transition, through a pointer the optimiser cannot see through: its
          // fourteen array pointers and its schedule are loop-invariant, were hoisted out of the transition loop and — with the
          // scalar registers full — lived in per-lane registers that were spilled to scratch and reloaded in every epilogue.
          // (Measured: the scratch footprint did not change — what is spilled there turned out to be the f64 constants of exp / log,
          // which the build now keeps inside the loop by compiling these kernels without machine LICM, build.py PART_B_FLAGS.  Kept:
          // without it the allocator parks a spill under a narrowed exec mask in k_nuts<float,32,4,4,1>, which isa_check refuses.)
          const AdaptK<T>* ak = ak0;
          asm volatile("" : "+s"(ak));
#endif
          // adapt!(h, κ, adaptor, i, n_adapts, z, α) + update(h/κ, adaptor) (src/sampler.jl:72-90, :3-22) for this chain —
          // the same decisions and the same arithmetic as the host path (`adapt` in ahmc_api.hip, k_adapt_da / k_adapt_wv)
          const int64_t i = ak->i0 + kt + 1;
          bool do_push, do_update, wv_reset, dareset;
          schedule(kt, do_push, do_update, wv_reset, dareset);
          if (ak->has_ss) {
            DAState<T> das{A_DAM, A_DAEPS, A_DAMU, A_DAXBAR, A_DAHBAR};
            da_step(das, sa_tree / (T)na_tree, ak->delta, ak->gamma, ak->t0, ak->kappa, ak->da_tab);
            if (dareset) da_reset(das);
            if (i == ak->n_adapts) das.eps = exp(das.xbar);  // finalize! (stepsize.jl:55-62)
            A_DAM = das.m; A_DAEPS = das.eps; A_DAMU = das.mu; A_DAXBAR = das.xbar; A_DAHBAR = das.Hbar;
            A_EPSNOM = das.eps;                              // update(κ, adaptor): nominal step size ← getϵ
          }
          if (ak->has_mm && (do_push || wv_reset)) {
            int wv_n = A_WVN;
            if (do_push) wv_n += 1;
            const T n = (T)wv_n;
            T mu[E], M2[E], mug[E], Mg[E];
            load_vec<T, E>(mu, ak->wv_mu, ce * p.D, d0, p.D, T(0));
            load_vec<T, E>(M2, ak->wv_M, ce * p.D, d0, p.D, T(0));
            if (ak->nutpie) {
              load_vec<T, E>(mug, ak->wg_mu, ce * p.D, d0, p.D, T(0));
              load_vec<T, E>(Mg, ak->wg_M, ce * p.D, d0, p.D, T(0));
            }
            // (round 3: requesting these, the accumulators and the energy sums at the top of the epilogue — one round trip with
            // the start point instead of three behind the re-integration — was measured: the warm-up instantiation went from 89
            // to 217 spilled VGPRs and from 2.26e9 to 1.45e9 leapfrog/s; not taken)
            if (do_push) {
#pragma unroll
              for (int e = 0; e < E; ++e) {
                welford_push(mu[e], M2[e], zc.th[e], n);
                if (ak->nutpie) welford_push(mug[e], Mg[e], zc.g[e], n);
              }
            }
            if (do_update && wv_n >= ak->wv_nmin) {  // update!(ve): M⁻¹ ← estimate, √M⁻¹ recomputed (metric.jl:61-63)
              T var[E], sq[E];
#pragma unroll
              for (int e = 0; e < E; ++e) {
                T v2 = welford_estimate(M2[e], n);
                if (ak->nutpie) v2 = sqrt(v2 / welford_estimate(Mg[e], n));
                var[e] = v2;
                sq[e] = sqrt(v2);
                if (d0 + e < p.D) minv[e] = v2;
              }
              store_vec<T, E>(var, ak->wv_var, ce * p.D, d0, p.D);
              store_vec<T, E>(var, ak->minv, ce * p.D, d0, p.D);
              store_vec<T, E>(sq, ak->sqrt_minv, ce * p.D, d0, p.D);
            }
            if (wv_reset) {
#pragma unroll
              for (int e = 0; e < E; ++e) { mu[e] = 0; M2[e] = 0; mug[e] = 0; Mg[e] = 0; }
              wv_n = 0;
            }
            A_WVN = wv_n;
            store_vec<T, E>(mu, ak->wv_mu, ce * p.D, d0, p.D);
            store_vec<T, E>(M2, ak->wv_M, ce * p.D, d0, p.D);
            if (ak->nutpie) {
              store_vec<T, E>(mug, ak->wg_mu, ce * p.D, d0, p.D);
              store_vec<T, E>(Mg, ak->wg_M, ce * p.D, d0, p.D);
            }
          }
        }
      }
    }
#if AHMC_WAVE_TIMELINE
    tl_trans += 1;
#endif
    AHMC_LP_TICK(7)
    AHMC_LP_COUNT(11, 1)
  }  // transitions of this launch
#if AHMC_WAVE_TIMELINE
  if (p.hmc_H && lane64 == 0) {  // (the static-HMC energy buffer is not used by NUTS: the measurement build borrows the field)
    // (the chunk index recomputed here: carried from the top it is one more value live across the whole kernel, and the register
    // allocator parked it in scratch under the `chunk < n_chunks` mask — harmless there, but isa_check refuses the pattern)
    unsigned int tid_tl = threadIdx.x;
    asm volatile("" : "+v"(tid_tl));   // (opaque: otherwise the expression is the one at the top and its value is carried after all)
    const unsigned int chunk_tl = G > 64 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + (tid_tl >> 6);
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(p.hmc_H) + (size_t)chunk_tl * AHMC_TL_WORDS;
#if AHMC_LEAF_PROF
    for (int i = 0; i < 12; ++i) tl[8 + i] = (unsigned long long)lp_t[i];
#endif
    tl[0] = tl_t0; tl[1] = wall_clock64(); tl[2] = tl_steps; tl[3] = tl_alive; tl[4] = tl_re; tl[5] = tl_trans;
    tl[6] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    tl[7] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
  }
#endif
  if constexpr (ADAPT) {
    if (active) save_adapt_state();
  }
#undef S_W
#undef S_SA
#undef S_DH
#undef S_NA
#undef S_CK
#undef A_EPSNOM
#undef A_DAEPS
#undef A_DAMU
#undef A_DAXBAR
#undef A_DAHBAR
#undef A_DAM
#undef A_WVN
#undef A_CHK
#undef AHMC_UNI
}

}  // namespace ahmc
