// The leaf-family kernels of float plans, in a translation unit of their own (they are most of the
// library's device code; the two element types compile side by side).
#include "dispatch_leaf.h"

namespace bn {
template int launch_leaf_family<float>(Plan&, FactorGroup&, FusedFamily&, WideNode&, WideNode*, float*, const float*, int32_t,
                                     int32_t, int32_t, hipStream_t, bool, int32_t);

bool hard_family_possible(const Plan& p, const FactorGroup& g, const FusedFamily& fam) { return hard_family_fits(p, g, fam); }

int32_t expected_pack_abi() { return pack_abi_of_build(); }
}  // namespace bn
