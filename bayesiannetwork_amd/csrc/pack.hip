// A cardinality pack: the fused wide-parent / leaf-family kernels instantiated for one more parent
// cardinality, BN_PACK_D0, as a shared object of its own (cardinality_pack.h).  It hands out kernel
// addresses and LDS sizes and nothing else: all launching stays with the library.
//   hipcc ... -DBN_PACK_D0=11 -shared -o libbn_hip_d11.so pack.hip        (make pack D0=11)
#ifndef BN_PACK_D0
#error "compile with -DBN_PACK_D0=<states of the wide parent>"
#endif

#include "dispatch_fast.h"
#include "dispatch_leaf.h"

namespace bn {
namespace {

constexpr int D0 = BN_PACK_D0;

template <typename T>
const void* family_of(int32_t mode, int32_t s) {
  switch (mode) {
#define X(M) \
  case M:    \
    return reinterpret_cast<const void*>(leaf_family_for_s<T, D0, M>(s));
    BN_FAMILY_MODES(X)
#undef X
  }
  return nullptr;
}

const void* pack_family(int32_t dtype, int32_t mode, int32_t s) {
  return dtype == 0 ? family_of<float>(mode, s) : family_of<double>(mode, s);
}

const void* pack_family_pairs4(int32_t mode) {
  switch (mode) {
#define X(M) \
  case M:    \
    return reinterpret_cast<const void*>(&leaf_family_p1<float, D0, 2, 4, M>);
    BN_PAIRS4_MODES(X)
#undef X
  }
  return nullptr;
}

template <typename T>
LeafLoopRef loop_of(int32_t out, int32_t from, int32_t s, bool split) {
  LeafLoopKernel<T> k;
  switch (out * 8 + from) {
#define X(OUT, FROM)                                  \
  case OUT * 8 + FROM:                                \
    k = leaf_loop_for_s<T, D0, OUT, FROM>(s, split);  \
    break;
    BN_LOOP_COMBOS(X)
#undef X
  }
  return {reinterpret_cast<const void*>(k.fn), k.lds_bytes};
}

LeafLoopRef pack_loop(int32_t dtype, int32_t out, int32_t from, int32_t s, int32_t split) {
  return dtype == 0 ? loop_of<float>(out, from, s, split != 0) : loop_of<double>(out, from, s, split != 0);
}

HardKernelRefs pack_hard() {
  const HardKernels k = hard_kernels_for<D0>();
  return {reinterpret_cast<const void*>(k.first), reinterpret_cast<const void*>(k.light),
          reinterpret_cast<const void*>(k.reduce), reinterpret_cast<const void*>(k.tables), k.first_lds};
}

WideKernelRefs pack_wide(int32_t dtype) { return dtype == 0 ? wide_kernels_for<float, D0>() : wide_kernels_for<double, D0>(); }

}  // namespace
}  // namespace bn

extern "C" const bn::CardinalityPack* bn_cardinality_pack() {
  static const bn::CardinalityPack pack = {bn::pack_abi_of_build(),        BN_PACK_D0,    bn::pack_family, bn::pack_family_pairs4,
                                           bn::pack_loop,       bn::pack_hard, bn::pack_wide};
  return &pack;
}
