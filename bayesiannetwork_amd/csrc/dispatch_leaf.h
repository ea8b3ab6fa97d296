// Launching the leaf-family kernels: which instantiation serves a launch, its geometry and the
// bookkeeping of the loop state.  Included by the translation units that instantiate these kernels
// (leaf_f32.hip, leaf_f64.hip); everybody else sees launch_leaf_family through dispatch_fast.h.
#pragma once

#include <algorithm>

#include "cardinality_pack.h"
#include "fast_state.h"
#include "kernels_leaf.h"
#include "kernels_hard.h"
#include "kernels_leaf_loop.h"
#include "launch.h"

namespace bn {

template <typename T>
using LeafFamilyFn = void (*)(T*, const T*, LeafFamilyArgs<T>);

// The instantiations launch_leaf_family chooses from, per parent cardinality (a cardinality pack
// holds the same ones): MODE = mode + 3 * state of leaf_family_p1; (OUT, FROM) of leaf_loop_p1;
// MODE of the reduce launches with four observations per lane
#define BN_FAMILY_MODES(X) X(1) X(2) X(4) X(5) X(7) X(8) X(10) X(11)
#define BN_LOOP_COMBOS(X)                                                                        \
  X(kLoopLight, kLoopChunk) X(kLoopLight, kLoopFresh) X(kLoopLight, kLoopFromArrays)            \
  X(kLoopLight, kLoopIngest) X(kLoopLight, kLoopIngestStates) X(kLoopLight, kLoopIngestKeep)    \
  X(kLoopLight, kLoopChunkCaller)                                                                \
  X(kLoopLast, kLoopChunk) X(kLoopLast, kLoopFresh) X(kLoopLast, kLoopFromArrays)
#define BN_PAIRS4_MODES(X)                                                                   \
  X(kLeafReduce + 3 * kLeafSteady) X(kLeafReduce + 3 * kLeafFresh) X(kLeafReduce + 3 * kLeafFromArrays) \
  X(kLeafReduce + 3 * kLeafRecompute) X(kLeafReduce + 3 * kLeafRecompute + kLeafCaller)

template <typename T>
constexpr int32_t pack_dtype() {
  return std::is_same<T, float>::value ? 0 : 1;
}

template <typename T, int D0, int MODE>
inline LeafFamilyFn<T> leaf_family_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return leaf_family_p1<T, D0, SV, kLeafVec, MODE>;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

// loop launches (kernels_leaf_loop.h): the kernel and the dynamic LDS its evidence rings need
template <typename T>
struct LeafLoopKernel {
  LeafFamilyFn<T> fn = nullptr;
  size_t lds_bytes = 0;
};

template <typename T, int D0, int OUT, int STATE>
inline LeafLoopKernel<T> leaf_loop_for_s(int32_t s, bool split) {
  constexpr bool kFromCaller = loop_reads_caller(STATE);
  constexpr bool kLight = loop_is_light(STATE);
  // blocks that share a chunk: instantiated for two-state leaves
  if (split)
    return s == 2 ? LeafLoopKernel<T>{leaf_loop_p1<T, D0, 2, OUT, STATE, true>,
                                      (size_t)kWavesPerBlock * LeafRing<T, 2, kFromCaller, kLight>::kWaveBytes}
                  : LeafLoopKernel<T>{};
  switch (s) {
#define X(SV)                                                                                       \
  case SV:                                                                                          \
    if constexpr (SV == 2 || (!loop_reads_caller(STATE) && STATE != kLoopIngestStates))             \
      return {leaf_loop_p1<T, D0, SV, OUT, STATE>,                                                  \
              (size_t)kWavesPerBlock * LeafRing<T, SV, kFromCaller, kLight>::kWaveBytes};           \
    break;
    BN_SMALL_DIMS(X)
#undef X
  }
  return {};
}

template <typename T, int OUT, int STATE>
inline LeafLoopKernel<T> leaf_loop_lookup(int32_t d0, int32_t s, bool split) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return leaf_loop_for_s<T, DV, OUT, STATE>(s, split);
    BN_PARENT_DIMS(X)
#undef X
  }
  if (const CardinalityPack* pack = cardinality_pack(d0)) {
    const LeafLoopRef r = pack->leaf_loop(pack_dtype<T>(), OUT, STATE, s, split ? 1 : 0);
    return {reinterpret_cast<LeafFamilyFn<T>>(const_cast<void*>(r.fn)), r.lds_bytes};
  }
  return {};
}

// reduce-mode launches of two-state leaves with four observations per lane
template <typename T, int MODE>
inline LeafFamilyFn<T> leaf_family_lookup_pairs4(int32_t d0) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return leaf_family_p1<T, DV, 2, 4, MODE>;
    BN_PARENT_DIMS(X)
#undef X
  }
  if (const CardinalityPack* pack = cardinality_pack(d0))
    return reinterpret_cast<LeafFamilyFn<T>>(const_cast<void*>(pack->leaf_family_pairs4(MODE)));
  return nullptr;
}

template <typename T, int MODE>
inline LeafFamilyFn<T> leaf_family_lookup(int32_t d0, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return leaf_family_for_s<T, DV, MODE>(s);
    BN_PARENT_DIMS(X)
#undef X
  }
  if (const CardinalityPack* pack = cardinality_pack(d0))
    return reinterpret_cast<LeafFamilyFn<T>>(const_cast<void*>(pack->leaf_family(pack_dtype<T>(), MODE, s)));
  return nullptr;
}

// the `abi` of a pack built from these headers (cardinality_pack.h)
constexpr int32_t pack_abi_of_build() {
  return kPackAbi + 1000 * (int32_t)(sizeof(LeafFamilyArgs<float>) + sizeof(LeafFamilyArgs<double>) + sizeof(WideArgs));
}

// two-parent families: the kernel of a (D0, D1, S) shape, nullptr for a shape without one
template <typename T, int MODE>
inline LeafFamilyFn<T> leaf_family_p2_lookup(int32_t d0, int32_t d1, int32_t s) {
#define X(A, B, C) \
  if (d0 == A && d1 == B && s == C) return leaf_family_p2<T, A, B, C, kLeafVec, MODE>;
  BN_TWO_PARENT_SHAPES(X)
#undef X
  if (const ShapePack* pack = shape_pack(d0, d1, s))
    return reinterpret_cast<LeafFamilyFn<T>>(const_cast<void*>(pack->leaf_family_p2(pack_dtype<T>(), MODE)));
  return nullptr;
}

// The launches of a step that reads compact hard evidence where the caller holds it (kernels_hard.h).
// Same contract as launch_leaf_family, for the three launches such a step consists of; -1 for any other.
using HardFn = void (*)(const float*, LeafFamilyArgs<float>);
using HardTablesFn = void (*)(const float*, LeafFamilyArgs<float>, float*);
struct HardKernels {
  HardFn first = nullptr, light = nullptr, reduce = nullptr;
  HardTablesFn tables = nullptr;
  size_t (*first_lds)(int32_t) = nullptr;  // LDS of the first launch: the staged rows of a chunk of that many families
};
template <int D0>
inline HardKernels hard_kernels_for() {
  if constexpr (D0 <= 16)
    return {hard_loop_first<D0>, hard_loop_light<D0>, hard_reduce_mfma<D0>, hard_tables_build<D0>,
            HardStage<D0, true>::bytes};
  return {};
}
inline HardKernels hard_kernels(int32_t d0) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return hard_kernels_for<DV>();
    BN_PARENT_DIMS(X)
#undef X
  }
  if (const CardinalityPack* pack = cardinality_pack(d0)) {
    const HardKernelRefs r = pack->hard();
    return {reinterpret_cast<HardFn>(const_cast<void*>(r.first)), reinterpret_cast<HardFn>(const_cast<void*>(r.light)),
            reinterpret_cast<HardFn>(const_cast<void*>(r.reduce)), reinterpret_cast<HardTablesFn>(const_cast<void*>(r.tables)),
            r.first_lds};
  }
  return {};
}

constexpr size_t kHardLdsMax = 64 * 1024;  // per block, so that at least two blocks share a CU

// the reduce launch's own cut of a group's K families: chunk length (whole groups of sixteen, as many as a wave
// keeps sums for)
inline int32_t hard_reduce_chunk_len(int32_t K) {
  const int32_t most = kHardSumGroups * kHardGroup;
  const int32_t chunks = (K + most - 1) / most;
  const int32_t len = (K + chunks - 1) / chunks;
  return (len + kHardGroup - 1) / kHardGroup * kHardGroup;
}

// Can the three launches serve this group (shape, room in LDS, room for the step's records)?
inline bool hard_family_fits(const Plan& p, const FactorGroup& g, const FusedFamily& fam) {
  const HardKernels kernels = hard_kernels(g.dims[0]);
  const int32_t K = (int32_t)g.nodes.size();
  if (!kernels.first || g.dims[1] != 2 || fam.src_stride == 0 || !fam.z_state) return false;
  if (hard_state_floats(K, g.dims[0]) > (size_t)K * (size_t)p.npad) return false;
  if (!hard_states_addressable(K, fam.src_stride, p.N)) return false;
  return kernels.first_lds(fam.chunk_len) <= kHardLdsMax;
}

inline int launch_hard_family(Plan& p, FactorGroup& g, FusedFamily& fam, WideNode& w, const float* cpts, int32_t fresh,
                              int32_t first_iteration, int32_t mode, hipStream_t s, bool with_counts, int32_t next_is_last) {
  const HardKernels kernels = hard_kernels(g.dims[0]);
  if (!hard_family_fits(p, g, fam)) return -1;
  // The step's records live where a step of this kind keeps nothing else: in the per-family loop-state
  // normalisers (its loop launches leave chunk products, its reduce launch forms the z again).
  LeafFamilyArgs<float> a{};
  a.hard_tab = static_cast<const float*>(fam.z_state);
  a.cpt_off = g.d_cpt;
  a.K = (int32_t)g.nodes.size();
  a.chunks = fam.chunks;
  a.chunk_len = fam.chunk_len;
  a.chunk_base = fam.chunk_base;
  a.part_mant = w.part_mant;
  a.part_exp = w.part_exp;
  a.zdiv_mant = w.zdiv_mant[w.zd_read];
  a.zdiv_exp = w.zdiv_exp[w.zd_read];
  a.zdiv_next_mant = w.zdiv_mant[w.zd_read ^ 1];
  a.zdiv_next_exp = w.zdiv_exp[w.zd_read ^ 1];
  a.scaled = w.scaled;
  a.scaled_prev = w.scaled_prev;
  a.npad = p.npad;
  a.ntiles = p.ntiles;
  a.N = p.N;
  a.em_partial = with_counts ? static_cast<float*>(fam.em_partial) : nullptr;
  a.slots = fam.slots;
  a.ll_part = fam.ll_part;
  a.src_base = fam.src_base;
  a.src_stride = fam.src_stride;
  std::copy(fast_state(p)->states_rows, fast_state(p)->states_rows + 6, a.hard_rows.e);
  HardFn fn = nullptr;
  size_t lds_bytes = 0;
  if (mode == kLeafLoop) {
    if (fresh) {
      if (next_is_last) return -1;  // (a reduce launch needs the normalisers of a light launch)
      fn = kernels.first;
      lds_bytes = kernels.first_lds(a.chunk_len);
    } else {
      // (the launch before the reduce launch is left out when a light launch has run: leaf_launch_folds)
      if (first_iteration || fam.z_form != kZPerChunk) return -1;
      fn = kernels.light;
      lds_bytes = kHardLogBytes;
    }
  } else if (mode == kLeafReduce) {
    if (fresh || first_iteration || !fam.hard_z_mant || fam.chunks > fam.ll_rows_cap) return -1;
    fn = kernels.reduce;
    a.zdiv_next_mant = fam.hard_z_mant;
    a.zdiv_next_exp = fam.hard_z_exp;
    a.z_rows = fam.chunks;
    a.chunk_len = hard_reduce_chunk_len(a.K);
    a.chunks = (a.K + a.chunk_len - 1) / a.chunk_len;
    a.slots = hard_sum_slots(p.ntiles);
  } else {
    return -1;
  }
  if (lds_bytes > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
    return -1;
  if (fn == kernels.first) {
    // the parameters are the step's: one set of records serves all its launches
    launch(kernels.tables, dim3((uint32_t)(((size_t)a.K * g.dims[0] + 255) / 256)), dim3(256), 0, s, cpts, a, static_cast<float*>(fam.z_state));
    if (!launch_ok()) return -1;
  }
  launch(fn, dim3(mode == kLeafReduce ? hard_sum_blocks(a.chunks, p.ntiles) : hard_blocks(a.chunks, p.ntiles)), dim3(256), lds_bytes, s, cpts, a);
  if (!launch_ok()) return -1;
  if (mode == kLeafLoop) {
    fam.z_form = kZPerChunk;
    fam.pending = false;
    fam.em_ready = false;
    w.partials_valid = true;
    if (fresh) w.products_cached = false;  // the first launch rewrote the chunks' partial products
    fam.hard_wants_light = fresh != 0;
    fam.hard_z_mant = fresh ? nullptr : a.zdiv_next_mant;
    fam.hard_z_exp = fresh ? nullptr : a.zdiv_next_exp;
  } else {
    fam.ll_rows = a.z_rows;
    fam.slots_used = a.slots;
    fam.em_by_form = true;
    fam.pending = true;
    fam.pending_fresh = fresh;
    fam.pending_first = first_iteration;
    fam.em_ready = with_counts;
  }
  return 1;
}

// w2: the second wide parent of a two-parent family group, nullptr for one-parent groups
template <typename T>
int launch_leaf_family(Plan& p, FactorGroup& g, FusedFamily& fam, WideNode& w, WideNode* w2, T* arena, const T* cpts,
                       int32_t fresh, int32_t first_iteration, int32_t mode, hipStream_t s, bool with_counts,
                       int32_t next_is_last) {
  // One-parent groups: the loop launch of the last but one iteration is left out when a reduce launch
  // follows and this iteration continues from loop state: all it would leave is one z per family,
  // and the reduce launch forms those again from the parent's two scaled products (kLeafRecompute).
  if (leaf_launch_folds(fam, mode == kLeafLoop, next_is_last, fresh, first_iteration)) {
    fam.z_form = kZRecompute;
    fam.pending = false;
    fam.em_ready = false;
    return 1;
  }
  if constexpr (std::is_same<T, float>::value) {
    if (fast_state(p)->states_in_caller && !w2 && g.dims[1] == 2)
      return launch_hard_family(p, g, fam, w, cpts, fresh, first_iteration, mode, s, with_counts, next_is_last);
  }
  fam.hard_wants_light = false;  // (launches of the other kind leave nothing for a reduce launch on state bytes)
  fam.hard_z_mant = nullptr;
  fam.hard_z_exp = nullptr;
  const int32_t state = fresh ? kLeafFresh
                              : (first_iteration ? kLeafFromArrays : (fam.z_form == kZRecompute ? kLeafRecompute : kLeafSteady));
  const bool ingest_wanted = fresh && !w2 && g.dims[1] == 2 && fast_state(p)->ingest_next;
  LeafFamilyFn<T> fn = nullptr;
  size_t lds_bytes = 0;
  const bool loop_p1 = mode == kLeafLoop && !w2;
  int32_t loop_out = kLoopLight;
  bool split = false;  // loop_p1 only: the waves of a block share one (chunk, tile, part) work item
  bool light = false;  // ... and the launch carries no per-state product
  if (loop_p1) {
    // one-parent loop launch: the launch before a full / reduce launch leaves one z per family,
    // any earlier one the chunk products of the z
    loop_out = next_is_last ? kLoopLast : kLoopLight;
    if (state == kLeafRecompute) return -1;  // no loop launch continues from there
    int32_t from = fresh ? kLoopFresh : (first_iteration ? kLoopFromArrays : kLoopChunk);
    const bool in_caller = fast_state(p)->evidence_in_caller;
    if (ingest_wanted) {
      if (loop_out != kLoopLight) return -1;  // callers ask for ingestion only with a loop launch to spare
      from = fast_state(p)->ingest_states ? kLoopIngestStates : (in_caller ? kLoopIngestKeep : kLoopIngest);
    } else if (in_caller) {
      if (from != kLoopChunk || loop_out != kLoopLight) return -1;  // (the launch before the reduce launch is left out)
      from = kLoopChunkCaller;
    }
    if (loop_is_light(from) && fam.z_form != kZPerChunk) return -1;
    // A launch that carries no per-state product needs half the registers: the waves of a block then
    // share a chunk at every batch size (four times the waves, each a quarter as long, six to
    // seven of them resident per SIMD instead of four).
    light = loop_is_light(from);
    split = fam.split > 1 || (light && g.dims[1] == 2 && !env_no_light_split());
    LeafLoopKernel<T> k;
    switch (loop_out * 8 + from) {
#define X(OUT, FROM)                                                      \
  case OUT * 8 + FROM:                                                    \
    k = leaf_loop_lookup<T, OUT, FROM>(g.dims[0], g.dims[1], split); \
    break;
      BN_LOOP_COMBOS(X)
#undef X
    }
    fn = k.fn;
    lds_bytes = k.lds_bytes;
    if (fn && lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return -1;
  } else if (w2) {
    if (ingest_wanted) return -1;
    switch (mode + 3 * state) {
#define X(M) \
  case M:    \
    fn = leaf_family_p2_lookup<T, M>(g.dims[0], g.dims[1], g.dims[2]); \
    break;
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
    }
  } else {
    if (ingest_wanted) return -1;  // the evidence would never reach the arena
    if (state == kLeafSteady && fam.z_form != kZPerFamily) return -1;
    switch (mode + 3 * state) {
#define X(M) \
  case M:    \
    fn = leaf_family_lookup<T, M>(g.dims[0], g.dims[1]); \
    break;
      BN_FAMILY_MODES(X)
#undef X
    }
  }
  // Reduce mode writes no partial products, so it is free to cut the families differently: with
  // four observations per lane (one wave per tile and chunk) the count sums cross the lanes half
  // as often per observation.
  bool wide4 = false;
  int32_t chunks_r = 0;
  if constexpr (std::is_same<T, float>::value) {
    if (mode == kLeafReduce && !w2 && g.dims[1] == 2 && p.tile == 256 && !env_reduce_v2()) {
      LeafFamilyFn<T> fn4 = nullptr;
      switch (state) {
        case kLeafSteady: fn4 = leaf_family_lookup_pairs4<T, kLeafReduce + 3 * kLeafSteady>(g.dims[0]); break;
        case kLeafFresh: fn4 = leaf_family_lookup_pairs4<T, kLeafReduce + 3 * kLeafFresh>(g.dims[0]); break;
        case kLeafFromArrays: fn4 = leaf_family_lookup_pairs4<T, kLeafReduce + 3 * kLeafFromArrays>(g.dims[0]); break;
        case kLeafRecompute:
          fn4 = fast_state(p)->evidence_in_caller
                    ? leaf_family_lookup_pairs4<T, kLeafReduce + 3 * kLeafRecompute + kLeafCaller>(g.dims[0])
                    : leaf_family_lookup_pairs4<T, kLeafReduce + 3 * kLeafRecompute>(g.dims[0]);
          break;
      }
      if (fast_state(p)->evidence_in_caller && state != kLeafRecompute) return -1;
      if (fn4) {
        // 140 VGPRs, three waves per SIMD: one round of resident waves (3 x 1024) per launch
        int64_t want = std::max<int64_t>(1, 3072 / p.ntiles);
        if (env_reduce_chunks() > 0) want = env_reduce_chunks();
        chunks_r = (int32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, fam.ll_rows_cap), (int64_t)g.nodes.size()));
        fn = fn4;
        wide4 = true;
      }
    }
  }
  if (!fn) return -1;
  if (w2 && (w2->chunks != fam.chunks || w.chunks != fam.chunks || w2->chunk_len != fam.chunk_len)) return -1;
  LeafFamilyArgs<T> a;
  a.f_in = g.d_in;
  a.f_out = g.d_out;
  a.cpt_off = g.d_cpt;
  a.obs_off = fam.d_obs_off;
  a.c_off = fam.d_c_off;
  a.K = (int32_t)g.nodes.size();
  a.chunks = fam.chunks;
  a.chunk_len = fam.chunk_len;
  a.chunk_base = fam.chunk_base;
  a.part_mant = w.part_mant;
  a.part_exp = w.part_exp;
  a.zdiv_mant = w.zdiv_mant[w.zd_read];
  a.zdiv_exp = w.zdiv_exp[w.zd_read];
  a.zdiv_next_mant = w.zdiv_mant[w.zd_read ^ 1];
  a.zdiv_next_exp = w.zdiv_exp[w.zd_read ^ 1];
  a.scaled = w.scaled;
  a.scaled_prev = w.scaled_prev;
  a.npad = p.npad;
  a.ntiles = p.ntiles;
  a.N = p.N;
  a.z_state = static_cast<T*>(fam.z_state);
  a.em_partial = with_counts ? static_cast<T*>(fam.em_partial) : nullptr;
  a.slots = fam.slots;
  a.ll_part = fam.ll_part;
  a.part_mant2 = w2 ? w2->part_mant : nullptr;
  a.part_exp2 = w2 ? w2->part_exp : nullptr;
  a.scaled2 = w2 ? w2->scaled : nullptr;
  a.m_state = static_cast<T*>(fam.m_state);
  a.src = fam.d_src;
  a.src_base = fam.src_base;
  a.src_stride = fam.src_stride;
  a.on_value = (T)fast_state(p)->ingest_on;
  a.off_value = (T)fast_state(p)->ingest_off;
  a.split = split ? kWavesPerBlock : 1;
  int64_t waves = (int64_t)fam.chunks * p.ntiles * leaf_parts(p);
  if (a.split > 1) {
    // one block per work item, and room for the quarters' products to meet (after the rings; a
    // launch without the per-state product reuses the rings)
    waves *= kWavesPerBlock;
    if (!light) lds_bytes += (size_t)(kWavesPerBlock - 1) * (g.dims[0] * 2 + 4) * 64 * (sizeof(double) + sizeof(int32_t));
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return -1;
  }
  if (!loop_p1 && !w2 && state == kLeafRecompute) {
    // the previous iteration's scaled product of the wave's observations, per lane
    lds_bytes = (size_t)kWavesPerBlock * g.dims[0] * 64 * (wide4 ? 4 : kLeafVec) * sizeof(T);
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return -1;
  }
  if (wide4) {
    const int32_t K = (int32_t)g.nodes.size();
    a.chunk_len = (K + chunks_r - 1) / chunks_r;
    a.chunks = (K + a.chunk_len - 1) / a.chunk_len;
    a.slots = (int32_t)p.ntiles;
    waves = (int64_t)a.chunks * p.ntiles;
  }
  launch(fn, dim3(wave_blocks(waves)), dim3(256), lds_bytes, s, arena, cpts, a);
  if (!launch_ok()) return -1;
  if (loop_p1) fam.z_form = loop_out == kLoopLast ? kZPerFamily : kZPerChunk;
  if (mode == kLeafReduce) {
    fam.ll_rows = a.chunks;
    fam.slots_used = a.slots;
    fam.em_by_form = false;
    fam.pending = true;
    fam.pending_fresh = fresh;
    fam.pending_first = first_iteration;
    fam.em_ready = with_counts;
  } else {
    fam.pending = false;
    fam.em_ready = false;
    w.partials_valid = true;
    if (w2) w2->partials_valid = true;
    // every launch but a loop launch without the per-state product rewrote its chunks' partial products
    if (!light) {
      w.products_cached = false;
      if (w2) w2->products_cached = false;
    }
  }
  return 1;
}

}  // namespace bn
