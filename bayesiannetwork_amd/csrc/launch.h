// One way to launch every kernel of the library.  While a timing region is open (benchmark
// profiling), each kernel is dispatched with a (start, stop) event pair attached to the dispatch
// itself (hipExtLaunchKernel): the timestamps are the dispatch packet's own, so no marker packet
// sits between kernels and the timed stream runs as it does untimed.
#pragma once

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstddef>

namespace bn {

// Hands out the event pair for the next launch of the open region (and takes note of the kernel that
// is launched); false = launch untimed.
using LaunchEventSource = bool (*)(const void* kernel, hipEvent_t* start, hipEvent_t* stop);

inline LaunchEventSource& launch_event_source() {
  static LaunchEventSource source = nullptr;
  return source;
}

// While set (per thread), launch() does nothing: the host-side bookkeeping of a sequence of
// launches is replayed without dispatching them (the kernels themselves then run from a captured
// graph of the same sequence).
inline bool& dry_launches() {
  static thread_local bool dry = false;
  return dry;
}

template <typename T>
struct launch_param {
  using type = T;
};

// launch(kernel, grid, block, dynamic LDS bytes, stream, kernel arguments...): arguments convert
// to the kernel's parameter types as in a direct call.
template <typename... KArgs>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                   typename launch_param<KArgs>::type... args) {
  if (dry_launches()) return;
  void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {const_cast<void*>(static_cast<const void*>(&args))...};
  hipEvent_t start = nullptr, stop = nullptr;
  LaunchEventSource source = launch_event_source();
  if (source && source(reinterpret_cast<const void*>(kernel), &start, &stop)) {
    (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream, start, stop, 0);
  } else {
    (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream);
  }
}

}  // namespace bn
