// Kernels for one more cardinality of a wide parent, in a shared object of their own.
//
// The fused wide-parent / leaf-family kernels keep a parent's states in registers, so the number of
// states D0 is a template argument, and the library proper carries the instantiations of a fixed
// list (BN_PARENT_DIMS).  Any other D0 is served by a "cardinality pack": pack.hip compiled with
// -DBN_PACK_D0=<D0> into libbn_hip_d<D0>.so next to the library (make pack D0=<D0>), which holds the
// same kernels instantiated for that D0 and hands out their addresses.  The library loads a pack
// the first time a plan meets its cardinality (dlopen; a HIP kernel of a loaded shared object is
// launched through its address like any other) and keeps all host-side logic itself: a pack
// contains no state and calls nothing outside itself.  No pack for a cardinality: the plan runs
// that node on the unfused kernels, as it always could.
#pragma once

#include <cstddef>
#include <cstdint>

namespace bn {

constexpr int32_t kPackAbi = 3;  // bumped whenever a kernel's arguments or this table change
// What a pack reports as its `abi`, and what the library expects of one: the number above together with the
// sizes of the structures its kernels take as arguments (a pack left over from a build with other structures
// would read its arguments wrongly: it is refused and the plan goes without it).  Defined where those structures
// are known (dispatch_leaf.h: pack_abi_of_build); the library's value reaches the loader through this function.
int32_t expected_pack_abi();

struct LeafLoopRef {
  const void* fn;
  size_t lds_bytes;
};
// the wide node's own kernels (kernels_wide.h), for dispatch_fast.h: launch_wide_node
struct WideKernelRefs {
  const void* product;         // wide_product_fast<S> (float) / wide_product_any<T, S>
  const void* combine_states;  // nullptr beyond 16 states
  const void* combine_fast;
  const void* emit;            // float only
  size_t (*combine_states_lds)(int32_t n_uncovered);
  size_t (*combine_lds)(int32_t waves);
};
// the launches over compact hard evidence (kernels_hard.h); all nullptr beyond 16 states
struct HardKernelRefs {
  const void *first, *light, *reduce, *tables;
  size_t (*first_lds)(int32_t chunk_len);
};

// dtype: 0 float, 1 double (bn_dtype).  mode / out / from: the kernels' template arguments
// (kernels_leaf.h: mode + 3 * state [+ kLeafCaller]; kernels_leaf_loop.h: kLoop...).
struct CardinalityPack {
  int32_t abi, d0;
  const void* (*leaf_family)(int32_t dtype, int32_t mode, int32_t s);
  const void* (*leaf_family_pairs4)(int32_t mode);  // float, two-state leaves, four observations per lane
  LeafLoopRef (*leaf_loop)(int32_t dtype, int32_t out, int32_t from, int32_t s, int32_t split);
  HardKernelRefs (*hard)();
  WideKernelRefs (*wide)(int32_t dtype);
};

// The pack for a parent of d0 states: loaded on first use, nullptr when there is none (or
// BN_NO_PACKS is set).  Thread-safe; a pack stays loaded for the life of the process.
const CardinalityPack* cardinality_pack(int32_t d0);

// The same for the fused kernel of leaf families with two wide parents (leaf_family_p2, CPT shape
// [D0][D1][S]): a fixed list of shapes is part of the library (BN_TWO_PARENT_SHAPES), any other is
// pack2.hip compiled with -DBN_PACK2_D0= -DBN_PACK2_D1= -DBN_PACK2_S= into libbn_hip_p<D0>x<D1>x<S>.so
// (make pack2 SHAPE=3x3x4).  mode: the kernel's MODE template argument, 0..8.
struct ShapePack {
  int32_t abi, d0, d1, s;
  const void* (*leaf_family_p2)(int32_t dtype, int32_t mode);
};
const ShapePack* shape_pack(int32_t d0, int32_t d1, int32_t s);

// where the library looks for packs (the directory it was loaded from); for messages and tools
const char* cardinality_pack_directory();

}  // namespace bn
