// Shape-specialised kernels for the shapes that carry the throughput.  Each launch_*
// helper returns 1 if it handled the group, 0 if the caller must use the generic kernel,
// and a negative value on a launch failure.
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace bn {

inline int64_t fast_scratch_elems(const Plan&) { return 0; }
inline int prepare_fast(Plan&) { return 0; }
inline void release_fast(Plan&) {}

template <typename T>
int launch_variable_fast(Plan&, VariableGroup&, T*, int32_t /*fresh*/, hipStream_t) {
  return 0;
}

template <typename T>
int launch_factor_fast(Plan&, FactorGroup&, T*, const T*, hipStream_t) {
  return 0;
}

template <typename T>
int launch_em_fast(Plan&, FactorGroup&, const T*, int32_t, double*, hipStream_t) {
  return 0;
}

}  // namespace bn
