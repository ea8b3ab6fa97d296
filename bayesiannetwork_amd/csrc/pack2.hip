// A shape pack: the fused kernel of leaf families with two wide parents (leaf_family_p2) instantiated
// for one more CPT shape [BN_PACK2_D0][BN_PACK2_D1][BN_PACK2_S], as a shared object of its own
// (cardinality_pack.h).  It hands out kernel addresses and nothing else.
//   hipcc ... -DBN_PACK2_D0=3 -DBN_PACK2_D1=3 -DBN_PACK2_S=4 -shared -o libbn_hip_p3x3x4.so pack2.hip   (make pack2 SHAPE=3x3x4)
#if !defined(BN_PACK2_D0) || !defined(BN_PACK2_D1) || !defined(BN_PACK2_S)
#error "compile with -DBN_PACK2_D0= -DBN_PACK2_D1= -DBN_PACK2_S= (states of the two parents and of the leaf)"
#endif

#include "dispatch_fast.h"
#include "dispatch_leaf.h"

namespace bn {
namespace {

template <typename T>
const void* p2_of(int32_t mode) {
  switch (mode) {
#define X(M) \
  case M:    \
    return reinterpret_cast<const void*>(&leaf_family_p2<T, BN_PACK2_D0, BN_PACK2_D1, BN_PACK2_S, kLeafVec, M>);
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
  }
  return nullptr;
}

const void* pack_p2(int32_t dtype, int32_t mode) { return dtype == 0 ? p2_of<float>(mode) : p2_of<double>(mode); }

}  // namespace
}  // namespace bn

extern "C" const bn::ShapePack* bn_shape_pack() {
  static const bn::ShapePack pack = {bn::pack_abi_of_build(), BN_PACK2_D0, BN_PACK2_D1, BN_PACK2_S, bn::pack_p2};
  return &pack;
}
