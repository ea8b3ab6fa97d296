// C ABI of libbn_hip.so (declared in include/bn_hip.h): argument checking, device
// memory ownership and kernel launches.  No torch types, no exceptions across the
// boundary.
#include "../../include/bn_hip.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include <hip/hip_runtime.h>

#include "dispatch_fast.h"
#include "kernels_star.h"
#include "kernels_finish.h"
#include "error.h"
#include "launch.h"
#include "kernels_generic.h"
#include "plan.h"

// BN_STEP_GRAPH=1 in the environment: new plans start with captured launch sequences switched on.
// They are off unless asked for: on this runtime a replayed graph of the 784-leaf model's 16-kernel
// step measured 3 to 6 % slower than the same kernels launched one by one (133 vs 136 us per step at
// 100 observations), the step being bound by its chain of dependent dispatches either way.
inline bool step_graphs_default() {
  static const bool on = [] {
    const char* v = getenv("BN_STEP_GRAPH");
    return v && v[0] == '1';
  }();
  return on;
}

// A captured EM-statistics launch sequence (see bn_em_statistics).
struct StepGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int32_t seen = 0;       // calls that arrived with this signature
  int32_t ll_blocks = 0;  // partial sums the sequence leaves for the final log-likelihood sum
};

// What the one-launch EM step of a star network needs (kernels_star.h); set up on first use.
struct StarState {
  bool checked = false, ok = false;
  bool stream = false;  // star_em_stream (evidence streamed per iteration) instead of star_em_step (kept in registers)
  int32_t fgroup = -1, root = -1, d0 = 0, chunks = 0, chunk_len = 0, nslices = 0;
  double* part_mant = nullptr;
  int32_t* part_exp = nullptr;
  double* count_part = nullptr;
  double* root_part = nullptr;
  double* ll_part = nullptr;
  uint32_t* barrier = nullptr;  // arrival counters (kernels_star.h)
  int32_t* error = nullptr;     // pinned host memory the kernel can write: a barrier wait ran out
  uint32_t epoch = 0, slice_epoch = 0;
};

struct bn_plan {
  bn::Plan p;
  StarState star;
  bool star_wanted = true;  // bn_plan_enable_star_step
  std::map<std::string, StepGraph> graphs;  // by signature of the plan's state and the call's arguments
  hipStream_t capture_stream = nullptr;     // only ever captured on, never run
  double* accum_slot = nullptr;             // where a captured sequence leaves the count sums
  bool graphs_off = false;                  // set for the duration of a call that must not allocate
  bool graphs_wanted = step_graphs_default();  // bn_plan_enable_step_graphs
  int64_t graph_launches = 0;
  int64_t star_launches = 0;  // EM steps that ran as one persistent launch
  int64_t star_failures = 0;  // ... of which so many were found void (their blocks could not synchronise)
  uint32_t* finish_counter = nullptr;  // arrival counter of em_finish (kernels_finish.h), zero between launches
  int32_t finish_shape = 0;            // 0 not examined yet, 1 the plan's steps can end in em_finish, -1 they cannot
  // Float evidence of a whole-step call that turns out to be hard evidence in disguise - every row one of
  // (a, b) and (b, a) - is turned into state bytes first (compact_evidence): the step then reads 47 MB
  // per launch instead of 376.  bn_plan_enable_evidence_compaction
  bool compact_wanted = true;
  uint8_t* compact_states = nullptr;  // [observed][N] state bytes, then the kernel's verdict (CompactVerdict)
  void* compact_verdict_host = nullptr;  // pinned: where the verdict is copied for the host to read
  int32_t compact_skip = 0;           // calls to go before evidence that was not of that kind is looked at again
  float compact_rows[6] = {};         // what the state bytes of the last compaction stand for (HardEvidenceRows)
  // The batch after one of that kind is taken to be of that kind, with the same two rows: its step is enqueued
  // on that guess while the pass over its evidence runs, and the verdict is read when everything is enqueued
  // (no wait with an idle GPU).  A wrong guess - the parameters put back, the step run again - costs a step.
  bool compact_have_rows = false;     // compact_rows are the last looked-at batch's
  bool compact_guessing = false;      // the call at work runs on a guess
  int32_t compact_blocks = 0;         // verdicts of the pass at work
  hipEvent_t compact_event = nullptr; // the verdicts of the pass at work have reached the host
  void* compact_cpts_backup = nullptr;  // the parameters before a step that runs on a guess
  int64_t compactions = 0;            // whole-step calls that ran on state bytes made this way
};

namespace {
thread_local std::string g_error = "";
}

namespace bn {
int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}
}  // namespace bn

namespace {

using bn::fail;

#define BN_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t err__ = (expr);                                                               \
    if (err__ != hipSuccess)                                                                 \
      return fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));         \
  } while (0)

#define BN_LAUNCH_CHECK(what)                                                                \
  do {                                                                                       \
    hipError_t err__ = hipGetLastError();                                                    \
    if (err__ != hipSuccess)                                                                 \
      return fail(BN_ERR_HIP, std::string("launch of ") + what + ": " + hipGetErrorString(err__)); \
  } while (0)

template <typename U>
int upload(const std::vector<U>& host, U** dev, int64_t& bytes) {
  *dev = nullptr;
  if (host.empty()) return BN_OK;
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(U)));
  bytes += host.size() * sizeof(U);
  BN_HIP_CHECK(hipMemcpy(*dev, host.data(), host.size() * sizeof(U), hipMemcpyHostToDevice));
  return BN_OK;
}

bn::FactorDims make_dims(const std::vector<int32_t>& dims) {
  bn::FactorDims fd{};
  fd.arity = (int32_t)dims.size();
  fd.total = 0;
  fd.elems = 1;
  for (int32_t j = 0; j < bn::kMaxArity; ++j) {
    fd.d[j] = j < fd.arity ? dims[j] : 1;
    fd.base[j] = fd.total;
    if (j < fd.arity) {
      fd.total += dims[j];
      fd.elems *= dims[j];
    }
  }
  return fd;
}

constexpr int64_t kLdsLimit = 158 * 1024;

// Timing regions: one record per prof_begin/prof_end bracket, holding the event pair of every
// kernel launched inside it (see launch.h).
struct ProfRecord {
  std::string label;
  double alg_bytes;
  double alg_flops;
  std::vector<hipEvent_t> events;      // start, stop, start, stop, ...
  std::vector<const void*> kernels;    // the kernel of each pair
};
std::mutex g_prof_mutex;
std::vector<ProfRecord> g_prof;
std::atomic<bool> g_profiling{false};
bool g_region_open = false;
std::string g_prof_prefix;  // only labels starting with this are timed
int32_t g_prof_stride = 1;  // of the regions that qualify, every g_prof_stride-th is timed
int64_t g_prof_seen = 0;

std::vector<hipEvent_t> g_event_pool;  // events are recycled: creating one costs far more than using it

bool take_event(hipEvent_t* ev) {
  if (!g_event_pool.empty()) {
    *ev = g_event_pool.back();
    g_event_pool.pop_back();
    return true;
  }
  return hipEventCreate(ev) == hipSuccess;
}

bool next_launch_events(const void* kernel, hipEvent_t* start, hipEvent_t* stop) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (!g_region_open || g_prof.empty()) return false;
  if (!take_event(start)) return false;
  if (!take_event(stop)) {
    g_event_pool.push_back(*start);
    return false;
  }
  g_prof.back().events.push_back(*start);
  g_prof.back().events.push_back(*stop);
  g_prof.back().kernels.push_back(kernel);
  return true;
}

// alg_flops: the additions, multiplications and divisions of the launch's message updates as the
// formulas state them (0 = not counted for this kind of launch).
void prof_begin(const std::string& label, double alg_bytes, hipStream_t, double alg_flops = 0.0) {
  if (!g_profiling.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (!g_prof_prefix.empty() && label.compare(0, g_prof_prefix.size(), g_prof_prefix) != 0) return;
  if (g_prof_seen++ % g_prof_stride != 0) return;
  ProfRecord r;
  r.label = label;
  r.alg_bytes = alg_bytes;
  r.alg_flops = alg_flops;
  g_prof.push_back(std::move(r));
  g_region_open = true;
}

void prof_end(hipStream_t) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_region_open = false;
}

// Compulsory bytes of one launch (see bn_profile_report in bn_hip.h).
double variable_bytes(const bn::Plan& p, const bn::VariableGroup& g) {
  double per_obs = 0;
  for (size_t k = 0; k < g.nodes.size(); ++k) {
    int32_t live_out = 0;
    for (int32_t i = 0; i < g.num_inputs; ++i) live_out += g.out_off[k * g.num_inputs + i] != bn::kNoArray;
    per_obs += (double)g.num_states * g.num_inputs + (double)g.num_states * live_out + 1;
  }
  return per_obs * (double)p.N * (double)p.elem_size;
}

double factor_bytes(const bn::Plan& p, const bn::FactorGroup& g, bool read_only) {
  double total = 0;
  for (int32_t d : g.dims) total += d;
  double per_node = g.dims.size() == 1 ? (read_only ? 0 : total) : (read_only ? total : 2 * total);
  return per_node * (double)g.nodes.size() * (double)p.N * (double)p.elem_size;
}

std::string vlabel(const bn::VariableGroup& g) {
  return "variable_pass[I=" + std::to_string(g.num_inputs) + ",S=" + std::to_string(g.num_states) +
         ",K=" + std::to_string(g.nodes.size()) + "]";
}

std::string flabel(const char* what, const bn::FactorGroup& g) {
  std::string l = std::string(what) + "[";
  for (size_t i = 0; i < g.dims.size(); ++i) l += (i ? "x" : "") + std::to_string(g.dims[i]);
  return l + ",K=" + std::to_string(g.nodes.size()) + "]";
}
constexpr int32_t kLlBlocks = 1024;

// Reduction scratch of bn_em_statistics_static: partial sums of the log-likelihood and of the EM
// counts live here instead of in memory the plan allocated.  One copy per GPU, held until the
// process ends (8 MiB).  In host code the array's name is only a handle for
// hipGetSymbolAddress; kernels and memsets get the per-device address from static_scratch().
constexpr int64_t kStaticScratchElems = int64_t(1) << 20;
static __device__ __align__(256) double scratch[kStaticScratchElems];

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
double* g_static_scratch[kMaxDevices] = {};

// Address of `scratch` on the current device (looked up once per device).
int static_scratch(double** out) {
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return fail(BN_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(err));
  if (device < 0 || device >= kMaxDevices) return fail(BN_ERR_INVALID, "device index beyond the static scratch table");
  std::lock_guard<std::mutex> lock(g_static_mutex);
  if (!g_static_scratch[device]) {
    double* ptr = nullptr;
    err = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (err != hipSuccess || !ptr)
      return fail(BN_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(err));
    g_static_scratch[device] = ptr;
  }
  *out = g_static_scratch[device];
  return BN_OK;
}

// The plan's pointers are only valid on the device it was created on.
int on_plan_device(const bn::Plan& p) {
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess || device != p.device)
    return fail(BN_ERR_STATE, "the plan belongs to HIP device " + std::to_string(p.device) +
                                  " but device " + std::to_string(device) + " is current");
  return BN_OK;
}

void release(bn::Plan& p) {
  for (auto& g : p.vgroups) {
    if (g.d_in) (void)hipFree(g.d_in);
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.d_c) (void)hipFree(g.d_c);
  }
  for (auto& g : p.fgroups) {
    if (g.d_in) (void)hipFree(g.d_in);
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.d_cpt) (void)hipFree(g.d_cpt);
  }
  if (p.arena) (void)hipFree(p.arena);
  if (p.lazy_base) {
    if (p.lazy_reserved) {
      if (p.lazy_mapped) {
        (void)hipMemUnmap(p.lazy_base, p.lazy_bytes);
        (void)hipMemRelease(p.lazy_handle);
      }
      (void)hipMemAddressFree(p.lazy_base, p.lazy_bytes);
    } else {
      (void)hipFree(p.lazy_base);
    }
    p.lazy_base = nullptr;
  }
  if (p.cpts) (void)hipFree(p.cpts);
  if (p.scratch) (void)hipFree(p.scratch);
  if (p.d_evidence_ptrs) (void)hipFree(p.d_evidence_ptrs);
  if (p.d_obs_off) (void)hipFree(p.d_obs_off);
  if (p.d_obs_states) (void)hipFree(p.d_obs_states);
  if (p.d_fill_off) (void)hipFree(p.d_fill_off);
  if (p.d_fill_states) (void)hipFree(p.d_fill_states);
  if (p.c_before) (void)hipFree(p.c_before);
  if (p.d_row_off) (void)hipFree(p.d_row_off);
  if (p.d_row_len) (void)hipFree(p.d_row_len);
}

hipMemAllocationProp lazy_memory_prop(int device) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  return prop;
}

// Address range for the message arrays of the leaf-family candidates (Plan::lazy_begin): reserved
// without memory behind it where the runtime offers that, otherwise allocated straight away.
// Afterwards every offset that pointed into the lazy part of the layout is relative to `arena`
// again, so kernels keep addressing everything as arena + offset.
int place_lazy_region(bn::Plan& p) {
  if (p.lazy_elems <= 0) return BN_OK;
  constexpr size_t kGranule = size_t(2) << 20;
  p.lazy_bytes = (((size_t)p.lazy_elems * p.elem_size + kGranule - 1) / kGranule) * kGranule;
  void* base = nullptr;
  if (hipMemAddressReserve(&base, p.lazy_bytes, kGranule, nullptr, 0) == hipSuccess && base) {
    p.lazy_reserved = true;
    p.lazy_mapped = false;
  } else {
    (void)hipGetLastError();
    BN_HIP_CHECK(hipMalloc(&base, p.lazy_bytes));
    p.lazy_reserved = false;
    p.lazy_mapped = true;
    p.device_bytes += (int64_t)p.lazy_bytes;
  }
  p.lazy_base = base;
  const int64_t byte_gap = static_cast<char*>(base) - static_cast<char*>(p.arena);
  if (byte_gap % (int64_t)p.elem_size != 0) return fail(BN_ERR_HIP, "lazy region is not aligned to the element size");
  const int64_t delta = byte_gap / (int64_t)p.elem_size - p.lazy_begin;
  auto rebase = [&](int64_t& off) {
    if (off >= p.lazy_begin) off += delta;
  };
  for (auto& e : p.edges) {
    rebase(e.v2f);
    rebase(e.f2v);
  }
  for (auto& g : p.vgroups) {
    for (auto& off : g.in_off) rebase(off);
    for (auto& off : g.out_off) rebase(off);
  }
  for (auto& g : p.fgroups) {
    for (auto& off : g.in_off) rebase(off);
    for (auto& off : g.out_off) rebase(off);
  }
  p.lazy_begin = -1;  // offsets are final
  p.lazy_lo = byte_gap / (int64_t)p.elem_size;
  p.lazy_hi = p.lazy_lo + p.lazy_elems;
  return BN_OK;
}

// Set for the duration of a call that must not allocate (bn_em_statistics_static): backing the lazy
// region is then an error instead of an allocation.
thread_local bool t_no_mapping = false;
const char* const kNoMappingMessage =
    "the plan's message arrays have no memory yet and this call would have to allocate it: call "
    "bn_plan_map_message_arrays (or bn_iterate / any query) once beforehand, or bn_reset_messages "
    "if the plan continues from an earlier statistics call";

// Backs the lazy region with memory (first call that reads or writes message arrays).
int ensure_message_arrays(bn::Plan& p) {
  if (p.lazy_elems <= 0 || p.lazy_mapped) return BN_OK;
  if (t_no_mapping) return fail(BN_ERR_STATE, kNoMappingMessage);
  hipMemAllocationProp prop = lazy_memory_prop(p.device);
  BN_HIP_CHECK(hipMemCreate(&p.lazy_handle, p.lazy_bytes, &prop, 0));
  hipError_t err = hipMemMap(p.lazy_base, p.lazy_bytes, 0, p.lazy_handle, 0);
  if (err == hipSuccess) {
    hipMemAccessDesc access = {};
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    err = hipMemSetAccess(p.lazy_base, p.lazy_bytes, &access, 1);
    if (err != hipSuccess) (void)hipMemUnmap(p.lazy_base, p.lazy_bytes);
  }
  if (err != hipSuccess) {
    (void)hipMemRelease(p.lazy_handle);
    return fail(BN_ERR_HIP, std::string("mapping the message arrays: ") + hipGetErrorString(err));
  }
  p.lazy_mapped = true;
  p.device_bytes += (int64_t)p.lazy_bytes;
  return BN_OK;
}

int32_t em_chunks(const bn::Plan& p, int64_t active_nodes = -1) {
  if (active_nodes >= 0) {
    // only `active_nodes` of the nodes go through the partial sums (the rest is served by the
    // fused kernels' own sums): more chunks each, within the scratch that was allocated
    const int64_t slices = p.ntiles * (p.tile / 64);
    const int64_t room = p.cpt_total > 0 ? (p.scratch_elems - kLlBlocks) / p.cpt_total : 1;
    int64_t want = (4096 + active_nodes - 1) / std::max<int64_t>(active_nodes, 1);
    want = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 256), room));
    return (int32_t)std::min<int64_t>(want, slices);
  }
  // enough (node, chunk) work items to fill the chip, never more chunks than 64-sample slices
  const int64_t slices = p.ntiles * (p.tile / 64);
  int64_t want = (4096 + p.n_nodes - 1) / p.n_nodes;
  want = std::max<int64_t>(1, std::min<int64_t>(want, 256));
  return (int32_t)std::min<int64_t>(want, slices);
}

template <typename T>
int fill_arrays(bn::Plan& p, int64_t first_array, int64_t count, hipStream_t s) {
  T* arena = static_cast<T*>(p.arena);
  if (count <= 0) return BN_OK;
  const int64_t max_elems = (int64_t)p.fill_max_states * p.npad;
  const int32_t gx = (int32_t)std::min<int64_t>((max_elems + 255) / 256, 4096);
  for (int64_t first = 0; first < count; first += 65535) {
    const int32_t cnt = (int32_t)std::min<int64_t>(count - first, 65535);
    bn::launch(bn::fill_uniform<T>, dim3(gx, cnt), dim3(256), 0, s, arena,
                       p.d_fill_off + first_array + first, p.d_fill_states + first_array + first, p.npad);
    BN_LAUNCH_CHECK("fill_uniform");
  }
  return BN_OK;
}

// Back to the state right after construction.  Observation slots and normalisers are
// written now; the message arrays only get marked (Plan::fresh) and are filled on demand.
template <typename T>
int reset_messages_t(bn::Plan& p, hipStream_t s) {
  T* arena = static_cast<T*>(p.arena);
  p.obs_pending = p.fill_count > p.fill_messages;
  (void)arena;
  (void)s;
  p.c_stale = true;  // zeroed on demand: every iteration overwrites all of them anyway
  p.c_before_valid = false;
  p.fresh = true;
  bn::invalidate_fast(p);
  return BN_OK;
}

// A leaf-family iteration that was only reduced (bn_em_statistics) is completed before anything
// reads or changes messages, evidence or CPTs.
int complete_pending(bn::Plan& p, hipStream_t s) {
  if (bn::any_pending_fast(p))
    if (int rc = ensure_message_arrays(p)) return rc;  // the re-run writes the message arrays
  if (bn::complete_pending_fast(p, s) < 0) return fail(BN_ERR_HIP, "launch of the leaf-family kernel failed");
  return BN_OK;
}

// Normalisers read as zero between a reset and the first iteration.
int settle_normalisers(bn::Plan& p, hipStream_t s) {
  if (!p.c_stale) return BN_OK;
  BN_HIP_CHECK(hipMemsetAsync(static_cast<char*>(p.arena) + (size_t)p.c_base * p.elem_size, 0,
                              (size_t)p.n_nodes * p.npad * p.elem_size, s));
  p.c_stale = false;
  return BN_OK;
}

// Observation slots that were reset but never received evidence get their uniform value now.
int settle_observations(bn::Plan& p, hipStream_t s) {
  if (!p.obs_pending) return BN_OK;
  int rc = p.dtype == BN_F32 ? fill_arrays<float>(p, p.fill_messages, p.fill_count - p.fill_messages, s)
                             : fill_arrays<double>(p, p.fill_messages, p.fill_count - p.fill_messages, s);
  if (rc == BN_OK) p.obs_pending = false;
  return rc;
}

// Anything that reads message arrays directly needs them to hold real values.
int materialize(bn::Plan& p, hipStream_t s) {
  if (int rc = ensure_message_arrays(p)) return rc;
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = settle_observations(p, s)) return rc;
  if (!p.fresh) return BN_OK;
  int rc = p.dtype == BN_F32 ? fill_arrays<float>(p, 0, p.fill_messages, s)
                             : fill_arrays<double>(p, 0, p.fill_messages, s);
  if (rc == BN_OK) p.fresh = false;
  return rc;
}

// Set while bn_em_step runs its statistics: the plan's messages are reset right after the M-step, so
// the factor -> variable messages of the last iteration (which neither the log-likelihood nor the
// count sums read) are not computed.  Fused leaf families are exempt: their last launch IS the
// statistics.
thread_local bool t_discard_last_messages = false;

template <typename T>
int iterate_t(bn::Plan& p, int32_t iterations, int32_t reduce_last, hipStream_t s) {
  T* arena = static_cast<T*>(p.arena);
  const T* cpts = static_cast<const T*>(p.cpts);
  const int32_t subs = p.tile / 64;
  // message arrays are read when the iterations continue from stored messages, and written when the
  // last one runs in full mode; a statistics call on fresh messages touches neither
  if (iterations > 0 && (!p.fresh || reduce_last == 0))
    if (int rc = ensure_message_arrays(p)) return rc;
  // A tree of dense one-parent families: the first and the last iteration(s) run as separate variable
  // and factor passes (the first consumes the fresh / evidence state, the last writes the variable ->
  // factor messages and the normalisers that queries read); an even number of iterations in between
  // runs as one launch per factor group, each reading one generation of factor -> variable messages
  // and writing the other, so that the newest generation ends up in the usual arrays.
  // When the messages are still fresh (uniform), the first iteration is such a launch too: its only
  // inputs are the observation arrays.
  const int32_t tree_count = bn::tree_iterations_possible(p) && iterations >= 4 ? ((iterations - 2) & ~1) : 0;
  for (int32_t it = 0; it < iterations; ++it) {
    if (it == 0 && tree_count > 0 && p.fresh) {
      for (size_t fi = 0; fi < p.fgroups.size(); ++fi) {
        bn::FactorGroup& g = p.fgroups[fi];
        if (g.dims.size() == 1) {  // the prior message of a parentless node
          prof_begin(flabel("factor_pass", g), factor_bytes(p, g, false), s);
          if (bn::launch_factor_fast<T>(p, g, arena, cpts, 1, 1, 0, reduce_last, s, 0) <= 0)
            return fail(BN_ERR_HIP, "launch of factor_root_fast failed");
          prof_end(s);
          continue;
        }
        prof_begin(flabel("tree_iteration_fresh", g),
                   bn::fast_state(p)->tree.at((int32_t)fi).fresh_bytes_per_observation * (double)p.npad, s);
        if (!bn::launch_tree_iteration(p, fi, bn::kTreeFresh, false, s))
          return fail(BN_ERR_HIP, "launch of factor_tree_mfma failed");
        prof_end(s);
      }
      p.fresh = false;  // every factor -> variable message has been written (the others by the last iteration)
      bn::set_ingest_next(p, false);
      continue;
    }
    if (it >= 1 && it <= tree_count) {
      for (size_t fi = 0; fi < p.fgroups.size(); ++fi) {
        const bn::FactorGroup& g = p.fgroups[fi];
        if (g.dims.size() != 2) continue;
        const bn::TreeGroupTables& tables = bn::fast_state(p)->tree.at((int32_t)fi);
        const bool exact = it == tree_count;
        prof_begin(flabel(exact ? "tree_iteration_exact" : "tree_iteration", g),
                   (tables.bytes_per_observation + (exact ? tables.exact_bytes_per_observation : 0.0)) * (double)p.npad, s,
                   4.0 * g.dims[0] * g.dims[1] * (double)g.nodes.size() * (double)p.N);
        if (!bn::launch_tree_iteration(p, fi, (it - 1) & 1, exact, s))
          return fail(BN_ERR_HIP, "launch of factor_tree_mfma failed");
        prof_end(s);
      }
      continue;
    }
    const int32_t fresh = p.fresh ? 1 : 0;
    for (auto& g : p.vgroups) {
      const int32_t K = (int32_t)g.nodes.size();
      if (g_profiling.load(std::memory_order_relaxed)) {  // (labels and byte counts cost host time: only when asked for)
        std::string label = vlabel(g);
        double bytes = variable_bytes(p, g);
        bn::variable_profile(p, (size_t)(&g - p.vgroups.data()), label, bytes);
        prof_begin(label, bytes, s);
      }
      if (p.lazy_unmapped() && bn::variable_needs_arrays(p, (size_t)(&g - p.vgroups.data()), fresh))
        if (int rc = ensure_message_arrays(p)) return rc;
      int rc = bn::launch_variable_fast<T>(p, g, arena, fresh, s, it == 0 ? 1 : 0, reduce_last != 0 && it == iterations - 2);
      if (rc < 0) return fail(BN_ERR_HIP, "launch of a specialised variable kernel failed");
      if (rc == 0) {
        const int64_t blocks = (int64_t)K * p.ntiles * subs;
        const size_t lds = (size_t)g.num_states * 64 * 24;
        if (lds > 48 * 1024)
          BN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn::variable_pass_generic<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        bn::launch(bn::variable_pass_generic<T>, dim3((uint32_t)blocks), dim3(64), lds, s, arena, g.d_in,
                           g.d_out, g.d_c, K, g.num_inputs, g.num_states, p.tile, p.ntiles, fresh);
        BN_LAUNCH_CHECK("variable_pass_generic");
      }
      prof_end(s);
    }
    p.fresh = false;  // every variable -> factor message has been written; the factor pass writes the rest
    p.c_stale = false;  // and every normaliser
    for (auto& g : p.fgroups) {
      const int32_t K = (int32_t)g.nodes.size();
      if (t_discard_last_messages && reduce_last && it == iterations - 1 &&
          !bn::is_fused_family_group(p, (size_t)(&g - p.fgroups.data())))
        continue;
      if (g_profiling.load(std::memory_order_relaxed)) {
        std::string label = flabel("factor_pass", g);
        double bytes = factor_bytes(p, g, false);
        double flops = 0;
        bn::factor_profile(p, (size_t)(&g - p.fgroups.data()), label, bytes, &flops,
                           reduce_last == 1 && it == iterations - 1);
        prof_begin(label, bytes, s, flops);
      }
      if (p.lazy_unmapped() && bn::factor_needs_arrays(p, (size_t)(&g - p.fgroups.data()), fresh, it == 0 ? 1 : 0,
                                                       it == iterations - 1 ? 1 : 0, reduce_last))
        if (int rc = ensure_message_arrays(p)) return rc;
      int rc = bn::launch_factor_fast<T>(p, g, arena, cpts, fresh, it == 0 ? 1 : 0, it == iterations - 1 ? 1 : 0, reduce_last, s,
                                         it == iterations - 2 ? 1 : 0);
      if (rc < 0) return fail(BN_ERR_HIP, "launch of a specialised factor kernel failed");
      if (rc == 0) {
        bn::FactorDims fd = make_dims(g.dims);
        const int64_t blocks = (int64_t)K * p.ntiles * subs;
        const size_t lds = (size_t)fd.total * 64 * (8 + sizeof(T));
        if (lds > 48 * 1024)
          BN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn::factor_pass_generic<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        bn::launch(bn::factor_pass_generic<T>, dim3((uint32_t)blocks), dim3(64), lds, s, arena, cpts,
                           g.d_in, g.d_out, g.d_cpt, K, fd, p.tile, p.ntiles);
        BN_LAUNCH_CHECK("factor_pass_generic");
      }
      prof_end(s);
    }
    bn::set_ingest_next(p, false);  // the observation arrays now hold the evidence
  }
  return BN_OK;
}

template <typename T>
int em_accumulate_t(bn::Plan& p, double* accum, bool use_reduced, hipStream_t s) {
  std::vector<const bn::FactorGroup*> reduced_groups;
  const T* arena = static_cast<const T*>(p.arena);
  const T* cpts = static_cast<const T*>(p.cpts);
  int64_t active_nodes = 0;
  for (auto& g : p.fgroups)
    if (!(use_reduced && bn::em_is_reduced(p, (size_t)(&g - p.fgroups.data())))) active_nodes += (int64_t)g.nodes.size();
  int32_t nchunks = active_nodes < p.n_nodes ? em_chunks(p, active_nodes) : em_chunks(p);
  // never more chunks than the scratch in use has room for (the plan's own is sized for them)
  if (p.cpt_total > 0)
    nchunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(nchunks, (p.scratch_elems - kLlBlocks) / p.cpt_total));
  double* partial = p.scratch + kLlBlocks;
  for (auto& g : p.fgroups) {
    const int32_t K = (int32_t)g.nodes.size();
    if (use_reduced && bn::em_is_reduced(p, (size_t)(&g - p.fgroups.data()))) {
      reduced_groups.push_back(&g);  // finalised below, after the generic finalize has filled the rest
      continue;
    }
    if (p.lazy_unmapped() && bn::factor_needs_arrays(p, (size_t)(&g - p.fgroups.data()), 0, 1, 1, 0))
      if (int rc = ensure_message_arrays(p)) return rc;
    prof_begin(flabel("em_counts", g), factor_bytes(p, g, true), s);
    int rc = bn::launch_em_fast<T>(p, g, arena, nchunks, partial, s);
    if (rc < 0) return fail(BN_ERR_HIP, "launch of a specialised EM kernel failed");
    if (rc > 0) {
      prof_end(s);
      continue;
    }
    bn::FactorDims fd = make_dims(g.dims);
    const size_t lds = (size_t)fd.total * 65 * sizeof(T);
    if (lds > 48 * 1024)
      BN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn::em_partial_generic<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t gx = (uint32_t)((fd.elems + 255) / 256);
    for (int32_t first = 0; first < K; first += 65535) {
      const int32_t cnt = std::min(K - first, 65535);
      bn::launch(bn::em_partial_generic<T>, dim3(gx, nchunks, cnt), dim3(256), lds, s, arena,
                         g.d_in + (int64_t)first * fd.arity, fd, p.N, p.tile, p.ntiles, nchunks, partial,
                         g.d_cpt + first);
      BN_LAUNCH_CHECK("em_partial_generic");
    }
    prof_end(s);
  }
  prof_begin("em_finalize", 0.0, s);
  bn::FastState* st = bn::fast_state(p);
  bn::ListedEntries listed{nullptr, nullptr, 0, 0};
  if (st && !st->fused.empty() && reduced_groups.size() == st->fused.size()) {
    // every fused group brings its own counts: only the other nodes' entries went through `partial`;
    // they are finalised by the first group's launch below
    if (st->n_unfused_entries > 0) listed = bn::ListedEntries{partial, st->d_unfused_entries, st->n_unfused_entries, nchunks};
  } else {
    bn::launch(bn::em_finalize<T>, dim3((uint32_t)((p.cpt_total + 255) / 256)), dim3(256), 0, s, partial,
                       cpts, p.cpt_total, nchunks, accum);
    BN_LAUNCH_CHECK("em_finalize");
  }
  prof_end(s);
  for (const bn::FactorGroup* g : reduced_groups) {
    prof_begin(flabel("em_counts(from leaf_family sums)", *g), 0.0, s);
    if (bn::finalize_em_fast(p, (size_t)(g - p.fgroups.data()), accum, s, listed) <= 0)
      return fail(BN_ERR_HIP, "launch of em_finalize_slots failed");
    listed.count = 0;
    prof_end(s);
  }
  return BN_OK;
}

// The table of evidence source addresses, on the device (kernel arguments carry it over, a batch
// per launch); skipped when the sources are the same as last time.
int upload_evidence_table(bn::Plan& p, const void* const* evidence_dev, hipStream_t s) {
  const size_t count = p.observed.size();
  const bool same_table = p.evidence_on_device.size() == count &&
                          std::memcmp(p.evidence_on_device.data(), evidence_dev, count * sizeof(void*)) == 0;
  if (same_table) return BN_OK;
  for (size_t first = 0; first < count; first += bn::kAddressBatch) {
    const int32_t cnt = (int32_t)std::min<size_t>(count - first, bn::kAddressBatch);
    bn::AddressBatch batch;
    std::memcpy(batch.p, evidence_dev + first, (size_t)cnt * sizeof(void*));
    bn::launch(bn::store_addresses, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, batch, cnt,
               const_cast<const void**>(p.d_evidence_ptrs) + first);
    BN_LAUNCH_CHECK("store_addresses");
  }
  p.evidence_on_device.assign(evidence_dev, evidence_dev + count);
  return BN_OK;
}

// For an ingesting first iteration: per fused family group, the evidence source of every leaf in
// the group's own order (the leaf's all-ones stand-in array when it is not observed).
// strided_ok: the kernels that will read the sources take a base address and a distance instead of
// the table when the sources lie at equal distances (the leaf-family launches of a statistics call
// that is not captured into a graph: a graph would keep the base address of the call it was made from)
int upload_family_sources(bn::Plan& p, const void* const* evidence_dev, hipStream_t s,
                          const uint8_t* states_dev = nullptr, bool strided_ok = false) {
  bn::FastState* st = bn::fast_state(p);
  for (auto& kv : st->fused) {
    bn::FusedFamily& fam = kv.second;
    const bn::FactorGroup& g = p.fgroups[kv.first];
    if (fam.wide_index2 >= 0 || g.dims[1] != 2) continue;
    const char* ones = static_cast<const char*>(p.arena) + (size_t)p.ones_off * p.elem_size;
    std::vector<const void*> src(g.nodes.size());
    for (size_t k = 0; k < src.size(); ++k) {
      if (states_dev)  // row of the leaf in the [num_observed][N] state matrix (all leaves observed)
        src[k] = states_dev + (int64_t)fam.ev_index[k] * p.N;
      else
        src[k] = fam.ev_index[k] >= 0 ? evidence_dev[fam.ev_index[k]] : ones;
    }
    fam.src_base = nullptr;
    fam.src_stride = 0;
    if (strided_ok && src.size() >= 2) {
      const int64_t stride = static_cast<const char*>(src[1]) - static_cast<const char*>(src[0]);
      bool even = stride != 0;
      for (size_t k = 2; k < src.size() && even; ++k)
        even = static_cast<const char*>(src[k]) - static_cast<const char*>(src[k - 1]) == stride;
      if (even) {
        fam.src_base = static_cast<const unsigned char*>(src[0]);
        fam.src_stride = stride;
        continue;  // no table needed
      }
    }
    if (src == fam.src_on_device) continue;
    for (size_t first = 0; first < src.size(); first += bn::kAddressBatch) {
      const int32_t cnt = (int32_t)std::min<size_t>(src.size() - first, bn::kAddressBatch);
      bn::AddressBatch batch;
      std::memcpy(batch.p, src.data() + first, (size_t)cnt * sizeof(void*));
      bn::launch(bn::store_addresses, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, s, batch, cnt,
                 fam.d_src + first);
      BN_LAUNCH_CHECK("store_addresses");
    }
    fam.src_on_device.swap(src);
  }
  return BN_OK;
}

// Partial sums of the log-likelihood over all normaliser arrays into p.scratch; returns their number.
int ll_partial_launch(bn::Plan& p, hipStream_t s, int32_t* blocks_out) {
  const int64_t total = (int64_t)p.n_nodes * p.npad;
  const int32_t blocks = (int32_t)std::min<int64_t>(kLlBlocks, (total + 255) / 256);
  if (p.dtype == BN_F32) {
    bn::launch(bn::ll_partial<float>, dim3(blocks), dim3(256), 0, s,
                       static_cast<const float*>(p.arena) + p.c_base, (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  } else {
    bn::launch(bn::ll_partial<double>, dim3(blocks), dim3(256), 0, s,
                       static_cast<const double*>(p.arena) + p.c_base, (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  }
  BN_LAUNCH_CHECK("ll_partial");
  *blocks_out = blocks;
  return BN_OK;
}

// The launches of bn_em_statistics up to, not including, the final sum of the log-likelihood
// partials (which writes to the caller's address): iterations with the last one in reduce mode,
// log-likelihood partial sums into p.scratch (*ll_blocks of them), count sums into accum (if wanted).
int statistics_sums(bn::Plan& p, double* accum, hipStream_t s, int32_t* ll_blocks);

int statistics_launches(bn::Plan& p, int32_t num_iterations, int32_t reduce_last, double* accum, hipStream_t s,
                        int32_t* ll_blocks) {
  int rc = p.dtype == BN_F32 ? iterate_t<float>(p, num_iterations, reduce_last, s)
                             : iterate_t<double>(p, num_iterations, reduce_last, s);
  if (rc != BN_OK) return rc;
  return statistics_sums(p, accum, s, ll_blocks);
}

// The second half: log-likelihood partial sums into p.scratch, count sums into accum (if wanted).
int statistics_sums(bn::Plan& p, double* accum, hipStream_t s, int32_t* ll_blocks) {
  // the fused families left their normalisers as per-observation partial sums
  const int used = bn::ll_partials_reduced(p, p.scratch, kLlBlocks, s);
  if (used < 0) return fail(BN_ERR_HIP, "launch of ll_partial_listed failed");
  if (used > 0) {
    *ll_blocks = used;
  } else {
    if (int rc2 = complete_pending(p, s)) return rc2;
    if (int rc2 = settle_normalisers(p, s)) return rc2;
    if (int rc2 = ll_partial_launch(p, s, ll_blocks)) return rc2;
  }
  if (!accum) return BN_OK;
  return p.dtype == BN_F32 ? em_accumulate_t<float>(p, accum, true, s) : em_accumulate_t<double>(p, accum, true, s);
}

// BN_STEP_GRAPH=0 in the environment: never run launch sequences as captured graphs
bool step_graphs_enabled() {
  static const bool on = [] {
    const char* v = getenv("BN_STEP_GRAPH");
    return !(v && v[0] == '0');
  }();
  return on;
}
// batches up to this many tiles (of 256 float / 128 double observations) are launch-bound
constexpr int64_t kStepGraphMaxTiles = 64;

// BN_STAR_STEP=0 in the environment: never run an EM step as the one-launch star kernel
bool star_steps_enabled() {
  static const bool on = [] {
    const char* v = getenv("BN_STAR_STEP");
    return !(v && v[0] == '0');
  }();
  return on;
}

void release_star(bn_plan* plan) {
  StarState& st = plan->star;
  for (void* buffer : {(void*)st.part_mant, (void*)st.part_exp, (void*)st.count_part, (void*)st.root_part,
                       (void*)st.ll_part, (void*)st.barrier})
    if (buffer) (void)hipFree(buffer);
  if (st.error) (void)hipHostFree(st.error);
  st = StarState();
}

// The streaming one-launch kernel for a root of d0 states (nullptr: none), and the dynamic LDS a block
// of it needs with chunk_len families per block (*cpt_in_lds: their CPT rows included)
const void* stream_kernel(int dtype, int32_t d0) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return dtype == BN_F32 ? reinterpret_cast<const void*>(&bn::star_em_stream<float, DV>) \
                           : reinterpret_cast<const void*>(&bn::star_em_stream<double, DV>);
    BN_STAR_DIMS(X)
#undef X
  }
  return nullptr;
}
size_t stream_lds_bytes(int32_t d0, int32_t chunk_len, size_t elem_size, size_t budget, int32_t* cpt_in_lds) {
  constexpr size_t V = 2, NW = bn::kStarWaves;
  const size_t base = (size_t)2 * d0 * V * 64 * 12 + NW * V * 64 * 12 + NW * 64 * sizeof(double) + (size_t)d0 * V * 64 * 12;
  const size_t cpt_rows = (size_t)chunk_len * d0 * 2 * elem_size;
  *cpt_in_lds = base + cpt_rows <= budget ? 1 : 0;
  return base + (*cpt_in_lds ? cpt_rows : 0);
}

// Is the plan a star (kernels_star.h) small enough for the one-launch step?  Allocates its buffers.
int prepare_star(bn_plan* plan) {
  StarState& st = plan->star;
  if (st.checked) return BN_OK;
  st.checked = true;
  bn::Plan& p = plan->p;
  bn::FastState* fs = bn::fast_state(p);
  if (!fs || fs->fused.size() != 1 || fs->wide.size() != 1) return BN_OK;
  const auto& entry = *fs->fused.begin();
  const bn::FusedFamily& fam = entry.second;
  const bn::FactorGroup& g = p.fgroups[entry.first];
  const bn::WideNode& w = fs->wide[fam.wide_index];
  const int32_t K = (int32_t)g.nodes.size();
  if (fam.wide_index2 >= 0 || g.dims.size() != 2 || g.dims[1] != 2 || K != p.n_nodes - 1) return BN_OK;
  if (!p.parents[w.node].empty() || p.observed_slot_of[w.node] >= 0) return BN_OK;
  if (w.uncovered.size() != 1 || w.uncovered[0] != w.I - 1) return BN_OK;
  bool small = false;
  switch (g.dims[0]) {
#define X(DV) case DV: small = true; break;
    BN_STAR_DIMS(X)
#undef X
  }
  if (!small) return BN_OK;
  const int64_t nslices = p.npad / bn::kStarSlice;
  if (p.npad % bn::kStarSlice != 0 || nslices < 1) return BN_OK;
  // every block has to be resident at once: at most half of the device's CUs (a block fills one)
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device) != hipSuccess) {
    (void)hipGetLastError();
    return BN_OK;
  }
  int32_t chunks = 0, chunk_len = 0;
  {
    const int64_t max_blocks = std::min<int64_t>(bn::kStarMaxBlocks, cus / 2);
    if (nslices <= bn::kStarMaxSlices && max_blocks >= nslices) {
      chunks = (int32_t)std::min<int64_t>(bn::kStarChunks, max_blocks / nslices);
      chunks = std::max(1, std::min(chunks, K));
      chunk_len = (K + chunks - 1) / chunks;
      chunks = (K + chunk_len - 1) / chunk_len;
      if ((chunk_len + bn::kStarWaves - 1) / bn::kStarWaves > bn::kStarFamilies) chunks = 0;  // too many families for the registers
    }
  }
  if (chunks == 0) {
    // Larger batches: the streaming kernel, two of whose blocks fit on a CU (every block has to be
    // resident: twice as many blocks as the chip has CUs at most), a power of two of chunks per slice.
    // (measured on the 784-leaf model: ahead of one launch per pass up to about 3 000 observations -
    // 0.094 against 0.114 ms per step at 1 536, 0.099 against 0.118 at 2 048 -, level at 3 072 and
    // behind it from there on, where a wave has a dozen and more families to walk per iteration;
    // BN_STREAM_MAX_SLICES moves the limit, in slices of 128 observations)
    static const int64_t max_slices = [] {
      const char* v = getenv("BN_STREAM_MAX_SLICES");
      return v ? (int64_t)atoi(v) : (int64_t)24;
    }();
    if (nslices > std::min<int64_t>(max_slices, bn::kStreamMaxSlices)) return BN_OK;
    const void* kernel = stream_kernel(p.dtype, g.dims[0]);
    if (!kernel) return BN_OK;
    // the most chunks per slice whose blocks the runtime says are all resident at once
    chunks = 0;
    for (int32_t want = bn::kStarChunks; want >= 1 && chunks == 0; --want) {
      if (want > K) continue;
      const int32_t len = (K + want - 1) / want;
      int32_t in_lds = 0, per_cu = 0;
      // (two blocks per CU where the CPT rows leave room: 80 KiB each; otherwise one with up to 150 KiB)
      size_t lds = stream_lds_bytes(g.dims[0], len, p.elem_size, 80 * 1024, &in_lds);
      if (!in_lds) lds = stream_lds_bytes(g.dims[0], len, p.elem_size, 150 * 1024, &in_lds);
      if (lds > 48 * 1024 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * bn::kStarWaves, lds) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      if ((int64_t)want * nslices <= (int64_t)per_cu * cus) {
        chunk_len = len;
        chunks = (K + len - 1) / len;
      }
    }
    if (chunks == 0) return BN_OK;
    st.stream = true;
  }
  st.fgroup = entry.first;
  st.root = w.node;
  st.d0 = g.dims[0];
  st.chunks = chunks;
  st.chunk_len = chunk_len;
  st.nslices = (int32_t)nslices;
  const size_t part = (size_t)2 * chunks * st.d0 * p.npad;
  const size_t counts = (size_t)K * st.d0 * 2 * nslices;
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.part_mant), part * sizeof(double)));
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.part_exp), part * sizeof(int32_t)));
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.count_part), counts * sizeof(double)));
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.root_part), (size_t)nslices * st.d0 * sizeof(double)));
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.ll_part), (size_t)(chunks * nslices + nslices) * sizeof(double)));
  const size_t counters = (size_t)(2 + std::max<int64_t>(bn::kStarMaxSlices, nslices)) * bn::kStarCounterStride * sizeof(uint32_t);
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&st.barrier), counters));
  BN_HIP_CHECK(hipMemset(st.barrier, 0, counters));
  BN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&st.error), sizeof(int32_t), hipHostMallocMapped));
  *st.error = 0;
  p.device_bytes += (int64_t)(part * 12 + counts * 8);
  st.ok = true;
  return BN_OK;
}

template <typename T, int D0>
int launch_star(bn_plan* plan, int32_t num_iterations, int32_t average, double learning_rate, double regularization,
                double* ll_out_dev, void* new_cpts_dev, hipStream_t s, bool from_caller) {
  bn::Plan& p = plan->p;
  StarState& st = plan->star;
  bn::FastState* fs = bn::fast_state(p);
  const bn::FactorGroup& g = p.fgroups[st.fgroup];
  bn::StarArgs<T> a;
  a.arena = static_cast<T*>(p.arena);
  a.cpts = static_cast<T*>(p.cpts);
  a.new_cpts = static_cast<T*>(new_cpts_dev);
  a.obs_off = fs->fused.begin()->second.d_obs_off;
  a.src = from_caller ? fs->fused.begin()->second.d_src : nullptr;
  a.z_scratch = static_cast<T*>(fs->fused.begin()->second.z_state);
  a.src_base = from_caller && st.stream ? fs->fused.begin()->second.src_base : nullptr;
  a.src_stride = from_caller && st.stream ? fs->fused.begin()->second.src_stride : 0;
  a.cpt_off = g.d_cpt;
  a.root_cpt_off = p.cpt_off[st.root];
  a.K = (int32_t)g.nodes.size();
  a.C = st.chunks;
  a.chunk_len = st.chunk_len;
  a.nslices = st.nslices;
  a.iterations = num_iterations;
  a.N = p.N;
  a.npad = p.npad;
  a.part_mant = st.part_mant;
  a.part_exp = st.part_exp;
  a.count_part = st.count_part;
  a.root_part = st.root_part;
  a.ll_part = st.ll_part;
  a.barrier = st.barrier;
  a.epoch0 = st.epoch;
  a.slice_epoch0 = st.slice_epoch;
  a.error = st.error;
  a.learning_rate = learning_rate;
  a.regularization = regularization;
  a.ll_scale = average ? 1.0 / (double)p.N : 1.0;
  a.ll_out = ll_out_dev;
  constexpr int32_t V = 2, NW = bn::kStarWaves;
  a.cpt_in_lds = 0;
  size_t stream_lds = 0;
  if (st.stream) {  // as prepare_star sized it
    stream_lds = stream_lds_bytes(D0, st.chunk_len, sizeof(T), 80 * 1024, &a.cpt_in_lds);
    if (!a.cpt_in_lds) stream_lds = stream_lds_bytes(D0, st.chunk_len, sizeof(T), 150 * 1024, &a.cpt_in_lds);
  }
  const size_t lds = st.stream ? stream_lds
                               : (size_t)(NW / 2) * D0 * V * 64 * 12 + (size_t)NW * V * 64 * 12 + (size_t)D0 * V * 64 * 12 +
                                     (size_t)NW * 64 * sizeof(double) + (size_t)D0 * V * 64 * 12 +
                                     (size_t)st.chunk_len * D0 * 2 * sizeof(T);
  const void* kernel = st.stream ? reinterpret_cast<const void*>(&bn::star_em_stream<T, D0>)
                                 : reinterpret_cast<const void*>(&bn::star_em_step<T, D0>);
  static size_t lds_allowed[2][kMaxDevices] = {};  // per instantiation and device: the attribute is the device's
  size_t& allowed = lds_allowed[st.stream ? 1 : 0][p.device >= 0 && p.device < kMaxDevices ? p.device : 0];
  if (lds > 48 * 1024 && (p.device < 0 || p.device >= kMaxDevices || allowed < lds)) {
    BN_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (p.device >= 0 && p.device < kMaxDevices) allowed = lds;
  }
  void* args[] = {&a};
  const uint32_t blocks = (uint32_t)(st.chunks * st.nslices);
  static const bool cooperative = getenv("BN_STAR_COOPERATIVE") != nullptr;
  const hipError_t err = cooperative ? hipLaunchCooperativeKernel(kernel, dim3(blocks), dim3(64 * NW), args,
                                                                  (unsigned int)lds, s)
                                     : hipLaunchKernel(kernel, dim3(blocks), dim3(64 * NW), args, lds, s);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return 1;  // not launched: the caller takes the launch-per-pass route
  }
  st.epoch += blocks;                                                    // one grid-wide barrier per launch
  st.slice_epoch += (uint32_t)(num_iterations - 1) * (uint32_t)st.chunks;  // one slice barrier per loop iteration
  return BN_OK;
}

template <typename T>
int launch_star_for_d0(bn_plan* plan, int32_t num_iterations, int32_t average, double learning_rate,
                       double regularization, double* ll_out_dev, void* new_cpts_dev, hipStream_t s, bool from_caller) {
  switch (plan->star.d0) {
#define X(DV) \
  case DV:    \
    return launch_star<T, DV>(plan, num_iterations, average, learning_rate, regularization, ll_out_dev, new_cpts_dev, s, \
                              from_caller);
    BN_STAR_DIMS(X)
#undef X
  }
  return 1;
}

// Can the plan's whole-step calls end in em_finish (kernels_finish.h)?  A star: one parentless wide
// node, every other node a leaf of the one fused one-parent group.
bool finish_shape_ok(bn_plan* plan) {
  if (plan->finish_shape != 0) return plan->finish_shape > 0;
  plan->finish_shape = -1;
  static const bool enabled = [] {
    const char* v = getenv("BN_STEP_FINISH");
    return !(v && v[0] == '0');
  }();
  bn::Plan& p = plan->p;
  bn::FastState* fs = bn::fast_state(p);
  if (!enabled || !fs || fs->fused.size() != 1 || fs->wide.size() != 1) return false;
  const auto& entry = *fs->fused.begin();
  const bn::FactorGroup& g = p.fgroups[entry.first];
  const bn::WideNode& w = fs->wide[entry.second.wide_index];
  if (entry.second.wide_index2 >= 0 || g.dims.size() != 2 || (int32_t)g.nodes.size() != p.n_nodes - 1) return false;
  if (g.dims[0] > bn::kFinishMaxStates || g.dims[1] > bn::kFinishMaxStates) return false;
  if (!p.parents[w.node].empty()) return false;
  const bn::FactorGroup& rg = p.fgroups[p.fgroup_of[w.node]];
  if (rg.dims.size() != 1 || rg.nodes.size() != 1 || fs->n_unfused_entries != rg.dims[0]) return false;
  // (the root's sums are taken in the order of the count kernels for parentless nodes, em_p1_fast /
  // em_root_any, which serve any root of that many states)
  if (hipMalloc(reinterpret_cast<void**>(&plan->finish_counter), sizeof(uint32_t)) != hipSuccess ||
      hipMemset(plan->finish_counter, 0, sizeof(uint32_t)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  plan->finish_shape = 1;
  return true;
}

// The tail of a whole step as one launch: what statistics_sums + m_step_rows do.  Returns 1 when the
// plan's state does not allow it (the caller then takes those), BN_OK when launched.
// accum_dev: where the count sums go (the plan's own buffer for a whole step); statistics_only: no M-step
template <typename T>
int launch_finish(bn_plan* plan, double learning_rate, double regularization, double ll_scale, double* ll_out_dev,
                  void* new_cpts_dev, hipStream_t s, int32_t* ll_blocks, double* accum_dev = nullptr,
                  bool statistics_only = false) {
  bn::Plan& p = plan->p;
  bn::FastState* fs = bn::fast_state(p);
  const auto& entry = *fs->fused.begin();
  const bn::FusedFamily& fam = entry.second;
  if (!fam.pending || !fam.em_ready) return 1;  // the last iteration did not run in reduce mode with counts
  const bn::FactorGroup& g = p.fgroups[entry.first];
  const bn::WideNode& w = fs->wide[fam.wide_index];
  const bn::FactorGroup& rg = p.fgroups[p.fgroup_of[w.node]];
  bn::FinishArgs<T> a;
  a.arena = static_cast<const T*>(p.arena);
  a.c_offs = fs->d_ll_offs;
  a.n_listed = fs->n_ll_offs;
  a.ll_rows = fam.ll_part;
  a.ll_row_count = fam.ll_rows;
  a.N = p.N;
  a.npad = p.npad;
  // as ll_partials_reduced cuts the sum for a single fused group
  a.ll_blocks = (int32_t)std::min<int64_t>(kLlBlocks, (a.n_listed + a.ll_row_count) * ((p.npad + 255) / 256));
  if (a.ll_blocks <= 0) return 1;
  a.ll_partial = p.scratch;
  a.root_in_off = rg.in_off[0];
  a.root_cpt_off = rg.cpt_off[0];
  a.root_states = rg.dims[0];
  a.ntiles = p.ntiles;
  // as em_accumulate_t cuts the root's sum when the root is the only node that goes through the partial sums
  int32_t nchunks = em_chunks(p, 1);
  if (p.cpt_total > 0)
    nchunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(nchunks, (p.scratch_elems - kLlBlocks) / p.cpt_total));
  a.root_chunks = nchunks;
  a.root_partial = p.scratch + kLlBlocks + a.root_cpt_off * nchunks;
  a.em_partial = static_cast<const T*>(fam.em_partial);
  a.cpt_off = g.d_cpt;
  a.K = (int32_t)g.nodes.size();
  a.D0 = g.dims[0];
  a.S = g.dims[1];
  a.slots = fam.slots_used;
  a.forms = bn::form_sums_of(p, fam);
  a.cpts = static_cast<T*>(p.cpts);
  a.new_cpts = static_cast<T*>(new_cpts_dev);
  a.accum = accum_dev ? accum_dev : plan->accum_slot;
  a.statistics_only = statistics_only ? 1 : 0;
  a.learning_rate = learning_rate;
  a.regularization = regularization;
  a.ll_scale = ll_scale;
  a.ll_out = ll_out_dev;
  a.arrived = plan->finish_counter;
  const int64_t rows = (int64_t)a.K * a.D0;
  const uint32_t blocks = (uint32_t)((a.ll_blocks + bn::kFinishLlPerBlock - 1) / bn::kFinishLlPerBlock +
                                     (a.root_chunks + bn::kWavesPerBlock - 1) / bn::kWavesPerBlock +
                                     (rows + bn::kWavesPerBlock - 1) / bn::kWavesPerBlock);
  prof_begin("em_finish", 0.0, s);
  bn::launch(bn::em_finish<T>, dim3(blocks), dim3(256), 0, s, a);
  BN_LAUNCH_CHECK("em_finish");
  prof_end(s);
  *ll_blocks = a.ll_blocks;
  return BN_OK;
}

void drop_step_graphs(bn_plan* plan) {
  for (auto& kv : plan->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  plan->graphs.clear();
}

// Small batches are bound by the number of launches, not by what the kernels do.  A launch sequence
// (`body`) that depends only on the plan's state and on `key` runs as one captured graph from the
// third call on that finds the plan in the same state with the same key (captured on the second
// such call; the first may still allocate or map memory).  The host-side bookkeeping of the
// launches is then replayed without dispatching anything (bn::dry_launches).
//   body(stream, in_graph, &ll_blocks): in_graph = its kernels run from a graph, so every address
//   they are given must be the same in all later calls (plan-owned buffers only).
template <typename Body>
int run_sequence(bn_plan* plan, std::string key, int32_t num_iterations, hipStream_t s, bool* from_graph,
                 int32_t* ll_blocks, Body body) {
  bn::Plan& p = plan->p;
  *from_graph = false;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (s) (void)hipStreamIsCapturing(s, &capturing);
  (void)hipGetLastError();
  const bool graphable = step_graphs_enabled() && plan->graphs_wanted && !plan->graphs_off &&
                         !g_profiling.load(std::memory_order_relaxed) && p.ntiles <= kStepGraphMaxTiles &&
                         capturing == hipStreamCaptureStatusNone;
  if (!graphable) return body(s, false, ll_blocks);
  {
    auto put = [&](const auto& v) { key.append(reinterpret_cast<const char*>(&v), sizeof v); };
    put(num_iterations);
    put(p.fresh);
    put(p.c_stale);
    put(p.obs_pending);
    put(p.lazy_mapped);
    put(p.scratch);
    put(p.scratch_elems);
    bn::append_state_signature(p, key);
  }
  if (plan->graphs.size() > 16 && !plan->graphs.count(key)) drop_step_graphs(plan);
  StepGraph& g = plan->graphs[key];
  if (!g.exec && g.seen == 0) {
    g.seen = 1;
    return body(s, false, ll_blocks);
  }
  if (!plan->accum_slot) {
    BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&plan->accum_slot), (size_t)std::max<int64_t>(p.cpt_total, 1) * sizeof(double)));
    p.device_bytes += p.cpt_total * 8;
  }
  if (!g.exec) {
    if (!plan->capture_stream) BN_HIP_CHECK(hipStreamCreateWithFlags(&plan->capture_stream, hipStreamNonBlocking));
    BN_HIP_CHECK(hipStreamBeginCapture(plan->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = body(plan->capture_stream, true, ll_blocks);
    hipGraph_t graph = nullptr;
    const hipError_t ended = hipStreamEndCapture(plan->capture_stream, &graph);
    if (rc != BN_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      plan->graphs.erase(key);
      return rc;
    }
    if (ended != hipSuccess || !graph) {
      plan->graphs.erase(key);
      return fail(BN_ERR_HIP, std::string("capturing the launch sequence: ") + hipGetErrorString(ended));
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t made = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (made != hipSuccess) {
      (void)hipGraphDestroy(graph);
      plan->graphs.erase(key);
      return fail(BN_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(made));
    }
    g.graph = graph;
    g.exec = exec;
    g.ll_blocks = *ll_blocks;
  } else {
    bn::dry_launches() = true;
    const int rc = body(s, true, ll_blocks);
    bn::dry_launches() = false;
    if (rc != BN_OK) return rc;
    if (*ll_blocks != g.ll_blocks)
      return fail(BN_ERR_STATE, "a replayed launch sequence took another course than the captured one");
  }
  BN_HIP_CHECK(hipGraphLaunch(g.exec, s));
  ++plan->graph_launches;
  *from_graph = true;
  return BN_OK;
}

}  // namespace

extern "C" {

int bn_abi_version(void) { return 1; }

const char* bn_last_error(void) { return g_error.c_str(); }

int bn_plan_create(int dtype, int64_t num_observations, int32_t num_nodes, const int32_t* num_states,
                   const int32_t* parent_offsets, const int32_t* parent_indices, const int32_t* child_offsets,
                   const int32_t* child_indices, const int32_t* observed_nodes, int32_t num_observed,
                   bn_plan** out_plan) {
  if (!out_plan) return fail(BN_ERR_INVALID, "out_plan is null");
  *out_plan = nullptr;
  bn_plan* h = new (std::nothrow) bn_plan();
  if (!h) return fail(BN_ERR_INVALID, "out of host memory");
  bn::Plan& p = h->p;
  std::string err = bn::compile_plan(p, dtype, num_observations, num_nodes, num_states, parent_offsets,
                                     parent_indices, child_offsets, child_indices, observed_nodes, num_observed);
  if (!err.empty()) {
    delete h;
    return fail(BN_ERR_INVALID, err);
  }
  // shapes the generic kernels can hold in LDS
  for (auto& g : p.vgroups) {
    if ((int64_t)g.num_states * 64 * 24 > kLdsLimit) {
      delete h;
      return fail(BN_ERR_INVALID, "a node has more states than the variable kernel supports");
    }
  }
  for (auto& g : p.fgroups) {
    int64_t total = 0;
    for (int32_t d : g.dims) total += d;
    if (total * 64 * (8 + (int64_t)p.elem_size) > kLdsLimit || total * 65 * (int64_t)p.elem_size > kLdsLimit) {
      delete h;
      return fail(BN_ERR_INVALID, "a family's cardinalities sum to more than the factor kernel supports");
    }
  }

  auto bail = [&](int rc) {
    bn::release_fast(p);
    release(p);
    delete h;
    return rc;
  };
#define BN_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != BN_OK) return bail(rc__); \
  } while (0)
#define BN_TRY_HIP(expr)                                                                  \
  do {                                                                                    \
    hipError_t err__ = (expr);                                                            \
    if (err__ != hipSuccess)                                                              \
      return bail(fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__))); \
  } while (0)

  BN_TRY_HIP(hipGetDevice(&p.device));
  {
    double* unused = nullptr;
    BN_TRY(static_scratch(&unused));  // set-up for bn_em_statistics_static, once per device
  }
  p.device_bytes = 0;
  BN_TRY_HIP(hipMalloc(&p.arena, (size_t)p.arena_elems * p.elem_size));
  p.device_bytes += p.arena_elems * (int64_t)p.elem_size;
  BN_TRY(place_lazy_region(p));
  BN_TRY_HIP(hipMalloc(&p.cpts, (size_t)p.cpt_total * p.elem_size));
  p.device_bytes += p.cpt_total * (int64_t)p.elem_size;
  {
    // room for more chunks per entry when only a few nodes need partial sums (at most 32 MiB extra)
    const int64_t generous = std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)(4 << 20) / std::max<int64_t>(p.cpt_total, 1)));
    p.scratch_elems = kLlBlocks + p.cpt_total * std::max<int64_t>(em_chunks(p), generous);
  }
  BN_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&p.scratch), (size_t)p.scratch_elems * sizeof(double)));
  p.device_bytes += p.scratch_elems * 8;

  for (auto& g : p.vgroups) {
    BN_TRY(upload(g.in_off, &g.d_in, p.device_bytes));
    BN_TRY(upload(g.out_off, &g.d_out, p.device_bytes));
    BN_TRY(upload(g.c_off, &g.d_c, p.device_bytes));
  }
  for (auto& g : p.fgroups) {
    BN_TRY(upload(g.in_off, &g.d_in, p.device_bytes));
    BN_TRY(upload(g.out_off, &g.d_out, p.device_bytes));
    BN_TRY(upload(g.cpt_off, &g.d_cpt, p.device_bytes));
  }
  if (!p.observed.empty()) {
    std::vector<int32_t> obs_states;
    for (int32_t o : p.observed) obs_states.push_back(p.states[o]);
    BN_TRY(upload(p.obs_off, &p.d_obs_off, p.device_bytes));
    BN_TRY(upload(obs_states, &p.d_obs_states, p.device_bytes));
    BN_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&p.d_evidence_ptrs), p.observed.size() * sizeof(void*)));
    p.obs_all_pairs = p.tile % 2 == 0;
    for (int32_t st : obs_states) p.obs_all_pairs = p.obs_all_pairs && st == 2;
  }
  {
    // every message array and observation array starts uniform
    std::vector<int64_t> offs;
    std::vector<int32_t> sts;
    int32_t max_states = 1;
    for (const auto& e : p.edges) {
      offs.push_back(e.v2f);
      sts.push_back(e.states);
      offs.push_back(e.f2v);
      sts.push_back(e.states);
      max_states = std::max(max_states, e.states);
    }
    for (size_t i = 0; i < p.observed.size(); ++i) {
      offs.push_back(p.obs_off[i]);
      sts.push_back(p.states[p.observed[i]]);
    }
    p.fill_count = (int64_t)offs.size();
    p.fill_messages = 2 * (int64_t)p.edges.size();
    for (size_t i = 0; i < p.observed.size(); ++i) max_states = std::max(max_states, p.states[p.observed[i]]);
    p.fill_max_states = max_states;
    BN_TRY(upload(offs, &p.d_fill_off, p.device_bytes));
    BN_TRY(upload(sts, &p.d_fill_states, p.device_bytes));
  }
  {
    std::vector<int64_t> row_off;
    std::vector<int32_t> row_len;
    for (int32_t j = 0; j < p.n_nodes; ++j) {
      const int32_t S = p.states[j];
      int64_t elems = 1;
      for (int32_t d : p.cpt_dims[j]) elems *= d;
      for (int64_t r = 0; r < elems / S; ++r) {
        row_off.push_back(p.cpt_off[j] + r * S);
        row_len.push_back(S);
      }
    }
    p.n_rows = (int64_t)row_off.size();
    BN_TRY(upload(row_off, &p.d_row_off, p.device_bytes));
    BN_TRY(upload(row_len, &p.d_row_len, p.device_bytes));
  }
  if (p.ones_off >= 0) {
    const int64_t count = (int64_t)p.ones_states * p.npad;
    const dim3 grid((uint32_t)((count + 255) / 256));
    if (p.dtype == BN_F32) {
      bn::launch(bn::fill_region<float>, grid, dim3(256), 0, nullptr, static_cast<float*>(p.arena) + p.ones_off, count, 1.0f);
    } else {
      bn::launch(bn::fill_region<double>, grid, dim3(256), 0, nullptr, static_cast<double*>(p.arena) + p.ones_off, count, 1.0);
    }
    BN_TRY_HIP(hipGetLastError());
  }
  if (bn::prepare_fast(p) != 0) return bail(fail(BN_ERR_HIP, "device allocation for the specialised kernels failed"));
  // a candidate that no fused kernel serves has its messages on the ordinary kernels' path: its
  // arrays are needed from the first iteration on
  if (!bn::candidates_all_fused(p)) BN_TRY(ensure_message_arrays(p));
  BN_TRY(p.dtype == BN_F32 ? reset_messages_t<float>(p, nullptr) : reset_messages_t<double>(p, nullptr));
  BN_TRY_HIP(hipStreamSynchronize(nullptr));
#undef BN_TRY
#undef BN_TRY_HIP
  *out_plan = h;
  return BN_OK;
}

int bn_plan_destroy(bn_plan* plan) {
  if (!plan) return BN_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(plan->p.device);
  drop_step_graphs(plan);
  release_star(plan);
  if (plan->capture_stream) (void)hipStreamDestroy(plan->capture_stream);
  if (plan->accum_slot) (void)hipFree(plan->accum_slot);
  if (plan->finish_counter) (void)hipFree(plan->finish_counter);
  if (plan->compact_states) (void)hipFree(plan->compact_states);
  if (plan->compact_verdict_host) (void)hipHostFree(plan->compact_verdict_host);
  if (plan->compact_event) (void)hipEventDestroy(plan->compact_event);
  if (plan->compact_cpts_backup) (void)hipFree(plan->compact_cpts_backup);
  bn::release_fast(plan->p);
  release(plan->p);
  (void)hipSetDevice(prev);
  delete plan;
  return BN_OK;
}

int bn_plan_topology(const bn_plan* plan, int32_t* is_tree, int32_t* factor_graph_diameter) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  if (is_tree) *is_tree = plan->p.is_tree ? 1 : 0;
  if (factor_graph_diameter) *factor_graph_diameter = plan->p.diameter;
  return BN_OK;
}

int64_t bn_plan_cpt_elements(const bn_plan* plan) { return plan ? plan->p.cpt_total : -1; }

int64_t bn_plan_device_bytes(const bn_plan* plan) { return plan ? plan->p.device_bytes : -1; }

int64_t bn_plan_unmapped_bytes(const bn_plan* plan) {
  if (!plan) return -1;
  return plan->p.lazy_elems > 0 && !plan->p.lazy_mapped ? (int64_t)plan->p.lazy_bytes : 0;
}

int bn_plan_map_message_arrays(bn_plan* plan) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  if (int rc = on_plan_device(plan->p)) return rc;
  return ensure_message_arrays(plan->p);
}

int bn_plan_enable_step_graphs(bn_plan* plan, int32_t enabled) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  plan->graphs_wanted = enabled != 0;
  return BN_OK;
}

int64_t bn_plan_step_graph_launches(const bn_plan* plan) { return plan ? plan->graph_launches : -1; }

int64_t bn_plan_star_step_launches(const bn_plan* plan) { return plan ? plan->star_launches : -1; }

int64_t bn_plan_star_step_failures(const bn_plan* plan) {
  if (!plan) return -1;
  return plan->star_failures + (plan->star.error && *plan->star.error != 0 ? 1 : 0);
}

int32_t bn_star_barrier_verdict(uint32_t seen, uint32_t target) { return bn::star_verdict(seen, target); }

int32_t bn_cardinality_served(int32_t states) {
  if (bn::builtin_parent_dim(states)) return 1;
  return bn::cardinality_pack(states) ? 2 : 0;
}

int32_t bn_two_parent_shape_served(int32_t d0, int32_t d1, int32_t s) {
  if (bn::two_parent_shape(d0, d1, s)) return 1;
  return bn::shape_pack(d0, d1, s) ? 2 : 0;
}

const char* bn_cardinality_pack_directory(void) { return bn::cardinality_pack_directory(); }

int bn_plan_enable_evidence_compaction(bn_plan* plan, int32_t enabled) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  plan->compact_wanted = enabled != 0;
  return BN_OK;
}

int64_t bn_plan_evidence_compactions(const bn_plan* plan) { return plan ? plan->compactions : 0; }

int bn_plan_enable_star_step(bn_plan* plan, int32_t enabled) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  plan->star_wanted = enabled != 0;
  return BN_OK;
}

int bn_set_cpts(bn_plan* plan, const void* cpts_dev, void* stream) {
  if (!plan || !cpts_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (int rc = complete_pending(p, static_cast<hipStream_t>(stream))) return rc;
  BN_HIP_CHECK(hipMemcpyAsync(p.cpts, cpts_dev, (size_t)p.cpt_total * p.elem_size, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
  p.cpts_set = true;
  bn::cpts_changed_fast(p);
  return BN_OK;
}

int bn_reset_messages(bn_plan* plan, void* stream) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p.dtype == BN_F32 ? reset_messages_t<float>(p, s) : reset_messages_t<double>(p, s);
}

int bn_enter_evidence(bn_plan* plan, const void* const* evidence_dev, void* stream) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (p.observed.empty()) return BN_OK;
  if (!evidence_dev) return fail(BN_ERR_INVALID, "evidence_dev is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  const size_t count = p.observed.size();
  int32_t max_states = 1;
  for (size_t i = 0; i < count; ++i) {
    if (!evidence_dev[i]) return fail(BN_ERR_INVALID, "an evidence pointer is null");
    max_states = std::max(max_states, p.states[p.observed[i]]);
  }
  if (int rc = upload_evidence_table(p, evidence_dev, s)) return rc;
  const uint32_t gx = (uint32_t)(((int64_t)max_states * p.npad + 255) / 256);
  {
    double bytes = 0;
    for (int32_t o : p.observed) bytes += (double)p.states[o] * (double)p.N * (double)p.elem_size;
    prof_begin("evidence_scatter", bytes, s);
  }
  for (size_t first = 0; first < count; first += 65535) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(count - first, 65535);
    if (p.obs_all_pairs && p.dtype == BN_F32) {
      bn::launch(bn::evidence_scatter_pairs<float>, dim3((uint32_t)((p.npad + 511) / 512), cnt), dim3(256), 0, s,
                 static_cast<float*>(p.arena), (const void* const*)(p.d_evidence_ptrs + first), p.d_obs_off + first,
                 p.N, p.tile, p.npad);
    } else if (p.obs_all_pairs) {
      bn::launch(bn::evidence_scatter_pairs<double>, dim3((uint32_t)((p.npad + 511) / 512), cnt), dim3(256), 0, s,
                 static_cast<double*>(p.arena), (const void* const*)(p.d_evidence_ptrs + first), p.d_obs_off + first,
                 p.N, p.tile, p.npad);
    } else if ((size_t)64 * (max_states + 1) * p.elem_size <= 64 * 1024) {
      // both sides coalesced through an LDS transpose
      const size_t lds = (size_t)64 * (max_states + 1) * p.elem_size;
      const dim3 grid((uint32_t)(p.npad / 64), cnt);
      if (p.dtype == BN_F32) {
        if (lds > 48 * 1024)
          BN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn::evidence_scatter_lds<float>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        bn::launch(bn::evidence_scatter_lds<float>, grid, dim3(256), lds, s, static_cast<float*>(p.arena),
                   (const void* const*)(p.d_evidence_ptrs + first), p.d_obs_off + first, p.d_obs_states + first, p.N,
                   p.tile);
      } else {
        if (lds > 48 * 1024)
          BN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bn::evidence_scatter_lds<double>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        bn::launch(bn::evidence_scatter_lds<double>, grid, dim3(256), lds, s, static_cast<double*>(p.arena),
                   (const void* const*)(p.d_evidence_ptrs + first), p.d_obs_off + first, p.d_obs_states + first, p.N,
                   p.tile);
      }
    } else if (p.dtype == BN_F32) {
      bn::launch(bn::evidence_scatter<float>, dim3(gx, cnt), dim3(256), 0, s, static_cast<float*>(p.arena),
                         (const void* const*)(p.d_evidence_ptrs + first), p.d_obs_off + first,
                         p.d_obs_states + first, p.N, p.tile, p.ntiles);
    } else {
      bn::launch(bn::evidence_scatter<double>, dim3(gx, cnt), dim3(256), 0, s,
                         static_cast<double*>(p.arena), (const void* const*)(p.d_evidence_ptrs + first),
                         p.d_obs_off + first, p.d_obs_states + first, p.N, p.tile, p.ntiles);
    }
    BN_LAUNCH_CHECK("evidence_scatter");
  }
  p.obs_pending = false;
  prof_end(s);
  return BN_OK;
}

int bn_enter_evidence_states(bn_plan* plan, const uint8_t* states_dev, double on_value, double off_value,
                             void* stream) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (p.observed.empty()) return BN_OK;
  if (!states_dev) return fail(BN_ERR_INVALID, "states_dev is null");
  if (p.tile % 4 != 0) return fail(BN_ERR_STATE, "observation tile is not a multiple of four");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  const size_t count = p.observed.size();
  {
    double bytes = 0;
    for (int32_t o : p.observed) bytes += (double)p.states[o] * (double)p.N * (double)p.elem_size;
    prof_begin("evidence_from_states", bytes, s);
  }
  const uint32_t gx = (uint32_t)((p.npad + 1023) / 1024);
  for (size_t first = 0; first < count; first += 65535) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(count - first, 65535);
    if (p.dtype == BN_F32) {
      bn::launch(bn::evidence_from_states<float>, dim3(gx, cnt), dim3(256), 0, s, static_cast<float*>(p.arena),
                 states_dev + (int64_t)first * p.N, p.d_obs_off + first, p.d_obs_states + first, p.N, p.tile, p.npad,
                 (float)on_value, (float)off_value);
    } else {
      bn::launch(bn::evidence_from_states<double>, dim3(gx, cnt), dim3(256), 0, s, static_cast<double*>(p.arena),
                 states_dev + (int64_t)first * p.N, p.d_obs_off + first, p.d_obs_states + first, p.N, p.tile, p.npad,
                 on_value, off_value);
    }
    BN_LAUNCH_CHECK("evidence_from_states");
  }
  p.obs_pending = false;
  prof_end(s);
  return BN_OK;
}

int bn_iterate(bn_plan* plan, int32_t num_iterations, void* stream) {
  if (!plan) return fail(BN_ERR_INVALID, "plan is null");
  if (num_iterations < 0) return fail(BN_ERR_INVALID, "num_iterations must be >= 0");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = settle_observations(p, s)) return rc;
  return p.dtype == BN_F32 ? iterate_t<float>(p, num_iterations, 0, s) : iterate_t<double>(p, num_iterations, 0, s);
}

int bn_log_likelihood(bn_plan* plan, int32_t average, double* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = settle_normalisers(p, s)) return rc;
  const int64_t total = (int64_t)p.n_nodes * p.npad;
  const int32_t blocks = (int32_t)std::min<int64_t>(kLlBlocks, (total + 255) / 256);
  prof_begin("log_likelihood", (double)p.n_nodes * (double)p.N * (double)p.elem_size, s);
  if (p.dtype == BN_F32) {
    bn::launch(bn::ll_partial<float>, dim3(blocks), dim3(256), 0, s,
                       static_cast<const float*>(p.arena) + p.c_base, (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  } else {
    bn::launch(bn::ll_partial<double>, dim3(blocks), dim3(256), 0, s,
                       static_cast<const double*>(p.arena) + p.c_base, (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  }
  BN_LAUNCH_CHECK("ll_partial");
  const double scale = average ? 1.0 / (double)p.N : 1.0;
  bn::launch(bn::sum_partials, dim3(1), dim3(256), 0, s, p.scratch, blocks, scale, out_dev);
  BN_LAUNCH_CHECK("sum_partials");
  prof_end(s);
  return BN_OK;
}

int bn_normaliser_change(bn_plan* plan, double* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = settle_normalisers(p, s)) return rc;
  const int64_t total = (int64_t)p.n_nodes * p.npad;
  if (!p.c_before) {
    BN_HIP_CHECK(hipMalloc(&p.c_before, (size_t)total * p.elem_size));
    p.device_bytes += total * (int64_t)p.elem_size;
    p.c_before_valid = false;
  }
  const int32_t blocks = (int32_t)std::min<int64_t>(kLlBlocks, (total + 255) / 256);
  const int32_t before_is_zero = p.c_before_valid ? 0 : 1;
  if (p.dtype == BN_F32) {
    bn::launch(bn::normaliser_change_partial<float>, dim3(blocks), dim3(256), 0, s,
               static_cast<const float*>(p.arena) + p.c_base, static_cast<float*>(p.c_before), before_is_zero,
               (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  } else {
    bn::launch(bn::normaliser_change_partial<double>, dim3(blocks), dim3(256), 0, s,
               static_cast<const double*>(p.arena) + p.c_base, static_cast<double*>(p.c_before), before_is_zero,
               (int64_t)p.n_nodes, p.N, p.npad, p.scratch);
  }
  BN_LAUNCH_CHECK("normaliser_change_partial");
  bn::launch(bn::max_partials, dim3(1), dim3(256), 0, s, p.scratch, blocks, out_dev);
  BN_LAUNCH_CHECK("max_partials");
  p.c_before_valid = true;
  return BN_OK;
}

static int gather(bn::Plan& p, int64_t a_off, int64_t b_off, int32_t states, void* out_dev, hipStream_t s) {
  const int64_t total = p.N * states;
  const uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (p.dtype == BN_F32) {
    bn::launch(bn::gather_rows<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(p.arena),
                       a_off, b_off, static_cast<float*>(out_dev), p.N, states, p.tile);
  } else {
    bn::launch(bn::gather_rows<double>, dim3(blocks), dim3(256), 0, s, static_cast<const double*>(p.arena),
                       a_off, b_off, static_cast<double*>(out_dev), p.N, states, p.tile);
  }
  BN_LAUNCH_CHECK("gather_rows");
  return BN_OK;
}

int bn_local_log_likelihoods(bn_plan* plan, int32_t node, void* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (node < 0 || node >= p.n_nodes) return fail(BN_ERR_INVALID, "node out of range");
  if (int rc = complete_pending(p, static_cast<hipStream_t>(stream))) return rc;
  if (int rc = settle_normalisers(p, static_cast<hipStream_t>(stream))) return rc;
  return gather(p, p.c_off[node], bn::kNoArray, 1, out_dev, static_cast<hipStream_t>(stream));
}

int bn_marginal_single(bn_plan* plan, int32_t node, void* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (node < 0 || node >= p.n_nodes) return fail(BN_ERR_INVALID, "node out of range");
  const bn::EdgeLayout& e = p.edges[node];
  if (int rc = materialize(p, static_cast<hipStream_t>(stream))) return rc;
  return gather(p, e.f2v, e.v2f, e.states, out_dev, static_cast<hipStream_t>(stream));
}

int bn_get_message(bn_plan* plan, int32_t owner, int32_t slot, int32_t direction, void* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (owner < 0 || owner >= p.n_nodes) return fail(BN_ERR_INVALID, "owner out of range");
  if (slot == -2) {
    if (p.observed_slot_of[owner] < 0) return fail(BN_ERR_INVALID, "node is not observed");
    if (int rc = settle_observations(p, static_cast<hipStream_t>(stream))) return rc;
    return gather(p, p.obs_off[p.observed_slot_of[owner]], bn::kNoArray, p.states[owner], out_dev,
                  static_cast<hipStream_t>(stream));
  }
  if (slot < -1 || slot >= (int32_t)p.parents[owner].size()) return fail(BN_ERR_INVALID, "slot out of range");
  if (direction != 0 && direction != 1) return fail(BN_ERR_INVALID, "direction must be 0 or 1");
  const bn::EdgeLayout& e = slot < 0 ? p.edges[owner] : p.edges[p.parent_edge_base[owner] + slot];
  if (int rc = materialize(p, static_cast<hipStream_t>(stream))) return rc;
  return gather(p, direction == 0 ? e.v2f : e.f2v, bn::kNoArray, e.states, out_dev, static_cast<hipStream_t>(stream));
}

int bn_marginal_family(bn_plan* plan, int32_t node, void* out_dev, void* stream) {
  if (!plan || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (node < 0 || node >= p.n_nodes) return fail(BN_ERR_INVALID, "node out of range");
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = materialize(p, s)) return rc;
  bn::FamilyArgs fa{};
  fa.fd = make_dims(p.cpt_dims[node]);
  for (size_t q = 0; q < p.parents[node].size(); ++q) fa.in_off[q] = p.edges[p.parent_edge_base[node] + q].v2f;
  fa.in_off[p.parents[node].size()] = p.edges[node].v2f;
  fa.cpt_off = p.cpt_off[node];
  const int64_t total = p.N * fa.fd.elems;
  const uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (p.dtype == BN_F32) {
    bn::launch(bn::family_marginal<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(p.arena),
                       static_cast<const float*>(p.cpts), fa, static_cast<float*>(out_dev), p.N, p.tile);
  } else {
    bn::launch(bn::family_marginal<double>, dim3(blocks), dim3(256), 0, s,
                       static_cast<const double*>(p.arena), static_cast<const double*>(p.cpts), fa,
                       static_cast<double*>(out_dev), p.N, p.tile);
  }
  BN_LAUNCH_CHECK("family_marginal");
  return BN_OK;
}

int bn_em_accumulate(bn_plan* plan, double* accum_dev, void* stream) {
  if (!plan || !accum_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = materialize(p, s)) return rc;
  return p.dtype == BN_F32 ? em_accumulate_t<float>(p, accum_dev, false, s)
                           : em_accumulate_t<double>(p, accum_dev, false, s);
}

int bn_em_statistics(bn_plan* plan, int32_t num_iterations, int32_t average, double* ll_out_dev, double* accum_dev,
                     void* stream) {
  if (!plan || !ll_out_dev) return fail(BN_ERR_INVALID, "null argument");
  if (num_iterations < 1) return fail(BN_ERR_INVALID, "num_iterations must be >= 1");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = settle_observations(p, s)) return rc;
  const int32_t reduce_last = accum_dev ? 1 : 2;
  const double scale = average ? 1.0 / (double)p.N : 1.0;
  auto finish = [&](int32_t ll_blocks) -> int {
    prof_begin("log_likelihood", (double)p.N * 8.0, s);
    bn::launch(bn::sum_partials, dim3(1), dim3(256), 0, s, p.scratch, ll_blocks, scale, ll_out_dev);
    BN_LAUNCH_CHECK("sum_partials");
    prof_end(s);
    return BN_OK;
  };
  std::string key = "statistics";
  key.append(reinterpret_cast<const char*>(&reduce_last), sizeof reduce_last);
  bool from_graph = false;
  int32_t ll_blocks = 0;
  const int rc = run_sequence(plan, key, num_iterations, s, &from_graph, &ll_blocks,
                              [&](hipStream_t on, bool in_graph, int32_t* blocks) {
                                double* target = accum_dev && in_graph ? plan->accum_slot : accum_dev;
                                return statistics_launches(p, num_iterations, reduce_last, target, on, blocks);
                              });
  if (rc != BN_OK) return rc;
  if (from_graph && accum_dev)
    BN_HIP_CHECK(hipMemcpyAsync(accum_dev, plan->accum_slot, (size_t)p.cpt_total * sizeof(double), hipMemcpyDeviceToDevice, s));
  return finish(ll_blocks);
}

int bn_em_statistics_static(bn_plan* plan, int32_t num_iterations, int32_t average, double* ll_out_dev,
                            double* accum_dev, void* stream) {
  if (!plan || !ll_out_dev) return fail(BN_ERR_INVALID, "null argument");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  const int64_t needed = kLlBlocks + (accum_dev ? p.cpt_total : 0);
  if (needed > kStaticScratchElems)
    return fail(BN_ERR_INVALID, "bn_em_statistics_static: the network's CPTs hold " + std::to_string(p.cpt_total) +
                                    " entries; the static scratch has room for " +
                                    std::to_string(kStaticScratchElems - kLlBlocks) + " (use bn_em_statistics)");
  double* base = nullptr;
  if (int rc = static_scratch(&base)) return rc;
  // Never back the lazy message arrays from here.  Whether the call would need them is known before
  // anything is launched: it continues from stored messages or has a reduced iteration to complete,
  // or one of its launches touches an array of the lazy region (a group that no fused kernel serves).
  if (p.lazy_unmapped()) {
    bool needed = bn::any_pending_fast(p) || !p.fresh;
    const int32_t reduce_last = accum_dev ? 1 : 2;
    for (size_t gi = 0; gi < p.vgroups.size() && !needed; ++gi)
      needed = bn::variable_needs_arrays(p, gi, 1) || (num_iterations > 1 && bn::variable_needs_arrays(p, gi, 0, true));
    for (size_t fi = 0; fi < p.fgroups.size() && !needed; ++fi) {
      needed = bn::factor_needs_arrays(p, fi, 1, 1, num_iterations == 1 ? 1 : 0, reduce_last) ||
               (num_iterations > 1 && bn::factor_needs_arrays(p, fi, 0, 0, 1, reduce_last)) ||
               (num_iterations > 2 && bn::factor_needs_arrays(p, fi, 0, 0, 0, reduce_last)) ||
               (accum_dev && !bn::is_fused_family_group(p, fi) && bn::factor_needs_arrays(p, fi, 0, 1, 1, 0));
    }
    if (needed) return fail(BN_ERR_STATE, std::string("bn_em_statistics_static: ") + kNoMappingMessage);
  }
  struct NoMapping {
    bool before;
    ~NoMapping() { t_no_mapping = before; }
  } no_mapping{t_no_mapping};
  t_no_mapping = true;  // (a backstop: the check above has covered every launch of the call)
  // the same kernels as bn_em_statistics, with their partial sums in the static array
  struct Swap {
    bn::Plan& p;
    double* own;
    int64_t own_elems;
    ~Swap() {
      p.scratch = own;
      p.scratch_elems = own_elems;
    }
  } swap{p, p.scratch, p.scratch_elems};
  p.scratch = base;
  p.scratch_elems = kStaticScratchElems;
  struct NoGraphs {
    bn_plan* plan;
    bool before;
    ~NoGraphs() { plan->graphs_off = before; }
  } no_graphs{plan, plan->graphs_off};
  plan->graphs_off = true;  // capturing and instantiating a graph allocates
  return bn_em_statistics(plan, num_iterations, average, ll_out_dev, accum_dev, stream);
}

int bn_em_statistics_from(bn_plan* plan, const void* const* evidence_dev, int32_t num_iterations, int32_t average,
                          double* ll_out_dev, double* accum_dev, void* stream) {
  if (!plan || !ll_out_dev) return fail(BN_ERR_INVALID, "null argument");
  if (num_iterations < 1) return fail(BN_ERR_INVALID, "num_iterations must be >= 1");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // The first iteration can take the evidence straight from the caller's arrays when nothing has
  // been sent yet, it is followed by at least one more loop iteration (the last one may have to be
  // re-run later, after the caller's arrays are gone, and the one before it is not instantiated
  // for ingestion), and the arrays allow whole-lane loads (an even number of observations, so that
  // a lane's pair never straddles the end).
  bool ingest = !p.observed.empty() && evidence_dev && p.fresh && num_iterations >= 3 && p.N >= 2 && p.N % 2 == 0 &&
                bn::ingest_possible(p);
  if (ingest)
    for (size_t i = 0; i < p.observed.size() && ingest; ++i)
      ingest = evidence_dev[i] && reinterpret_cast<uintptr_t>(evidence_dev[i]) % 32 == 0;
  if (!ingest) {
    if (int rc = bn_enter_evidence(plan, evidence_dev, stream)) return rc;
    return bn_em_statistics(plan, num_iterations, average, ll_out_dev, accum_dev, stream);
  }
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = upload_family_sources(p, evidence_dev, s)) return rc;
  p.obs_pending = false;  // the first factor pass writes every observation array
  bn::set_ingest_next(p, true);
  const int rc = bn_em_statistics(plan, num_iterations, average, ll_out_dev, accum_dev, stream);
  bn::set_ingest_next(p, false);
  return rc;
}

int bn_em_statistics_from_states(bn_plan* plan, const uint8_t* states_dev, double on_value, double off_value,
                                 int32_t num_iterations, int32_t average, double* ll_out_dev, double* accum_dev,
                                 void* stream) {
  if (!plan || !ll_out_dev) return fail(BN_ERR_INVALID, "null argument");
  if (num_iterations < 1) return fail(BN_ERR_INVALID, "num_iterations must be >= 1");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // as bn_em_statistics_from; a lane reads its two observations as one 16-bit load, so N must be
  // even and the matrix 2-byte aligned, and every leaf of the ingesting groups must be observed
  const bool ingest = !p.observed.empty() && states_dev && p.fresh && num_iterations >= 3 && p.N >= 2 &&
                      p.N % 2 == 0 && reinterpret_cast<uintptr_t>(states_dev) % 2 == 0 &&
                      bn::ingest_states_possible(p);
  if (!ingest) {
    if (int rc = bn_enter_evidence_states(plan, states_dev, on_value, off_value, stream)) return rc;
    return bn_em_statistics(plan, num_iterations, average, ll_out_dev, accum_dev, stream);
  }
  if (int rc = complete_pending(p, s)) return rc;
  if (int rc = upload_family_sources(p, nullptr, s, states_dev)) return rc;
  p.obs_pending = false;  // the first factor pass writes every observation array
  bn::set_ingest_next(p, true, true, on_value, off_value);
  const int rc = bn_em_statistics(plan, num_iterations, average, ll_out_dev, accum_dev, stream);
  bn::set_ingest_next(p, false);
  return rc;
}

// Float evidence of a whole-step call that is hard evidence in disguise: the rows of all leaves and observations
// are two rows only, one per state (what expanding states with an on and an off value gives, or clamping
// one-hot rows).
// Then the step runs on state bytes (kernels_hard.h), which this turns the arrays into first: one pass
// over the evidence (evidence_two_rows) and a look at its verdict, which costs less than the two
// passes over the float arrays it saves from batches of kCompactMinN observations on.  *states: the
// state bytes (plan->compact_rows: what they stand for), or null when the evidence is not of that kind
// (or the step would not run on them).
constexpr int64_t kCompactMinN = 28672;
// What the blocks of the pass over the evidence have seen, once their verdicts are on the host: are the rows two
// rows only, and which (rows: HardEvidenceRows)?
bool compact_verdict_rows(const bn_plan* plan, int32_t blocks, float* rows) {
  const bn::CompactVerdict* host = static_cast<const bn::CompactVerdict*>(plan->compact_verdict_host);
  bn::CompactVerdict seen{};
  for (int32_t i = 0; i < 4; ++i) seen.lo[i] = 0x7f800000u;
  for (int32_t b = 0; b < blocks; ++b) {
    seen.other |= host[b].other;
    for (int32_t i = 0; i < 4; ++i) {
      seen.lo[i] = std::min(seen.lo[i], host[b].lo[i]);
      seen.hi[i] = std::max(seen.hi[i], host[b].hi[i]);
    }
  }
  bool two_rows = seen.other == 0;
  for (int32_t i = 0; i < 4; ++i) two_rows = two_rows && (seen.lo[i] == seen.hi[i] || seen.lo[i] > seen.hi[i]);
  const bool has0 = seen.lo[0] <= seen.hi[0], has1 = seen.lo[2] <= seen.hi[2];
  if (!two_rows || (!has0 && !has1)) return false;
  // the rows of the two states (a state no observation has takes the other's: nothing reads it), and for a
  // state byte beyond them - there is none - the smaller entries
  auto value = [](uint32_t bits) {
    float x;
    std::memcpy(&x, &bits, sizeof x);
    return x;
  };
  for (int32_t st = 0; st < 2; ++st) {
    const int32_t from = (st == 0 ? has0 : has1) ? st : 1 - st;
    rows[2 * st] = value(seen.lo[2 * from]);
    rows[2 * st + 1] = value(seen.lo[2 * from + 1]);
  }
  rows[4] = rows[5] = std::min(std::min(rows[0], rows[1]), std::min(rows[2], rows[3]));
  return true;
}

int compact_evidence(bn_plan* plan, const void* const* evidence_dev, int32_t num_iterations, hipStream_t s,
                     const uint8_t** states, bool may_guess) {
  *states = nullptr;
  plan->compact_guessing = false;
  bn::Plan& p = plan->p;
  static const bool off = getenv("BN_NO_EVIDENCE_COMPACTION") != nullptr;
  static const int64_t min_n = getenv("BN_COMPACT_MIN_N") ? atoll(getenv("BN_COMPACT_MIN_N")) : kCompactMinN;
  if (off || !plan->compact_wanted || p.dtype != BN_F32 || p.observed.empty() || !evidence_dev || num_iterations < 3 ||
      p.N < min_n || p.N % 2 != 0 || !bn::ingest_states_possible(p))
    return BN_OK;
  if (step_graphs_enabled() && plan->graphs_wanted && !plan->graphs_off) return BN_OK;  // (captured steps keep their route)
  if (plan->compact_skip > 0) {
    --plan->compact_skip;
    return BN_OK;
  }
  // the arrays at equal distances, in the order of the observed nodes
  const size_t K = p.observed.size();
  const char* base = static_cast<const char*>(evidence_dev[0]);
  const int64_t stride = K > 1 ? static_cast<const char*>(evidence_dev[1]) - base : (int64_t)p.N * 8;
  for (size_t i = 0; i < K; ++i)
    if (!evidence_dev[i] || static_cast<const char*>(evidence_dev[i]) != base + (int64_t)i * stride) return BN_OK;
  if (reinterpret_cast<uintptr_t>(base) % 16 != 0 || stride % 16 != 0) return BN_OK;  // (the kernel reads 16 bytes at a time)
  constexpr int32_t kBlocks = 1024;  // of the pass over the evidence: four per compute unit
  if (!plan->compact_states) {
    const size_t bytes = K * (size_t)p.N + 64 + kBlocks * sizeof(bn::CompactVerdict);
    BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&plan->compact_states), bytes));
    // pinned: where the blocks' verdicts are copied for the host to read
    BN_HIP_CHECK(hipHostMalloc(&plan->compact_verdict_host, kBlocks * sizeof(bn::CompactVerdict), hipHostMallocDefault));
    p.device_bytes += (int64_t)bytes;
  }
  // would the step run on them?  (the families' rows as the state bytes would give them)
  if (int rc = upload_family_sources(p, nullptr, s, plan->compact_states, true)) return rc;
  if (!bn::caller_states_possible(p, num_iterations)) {
    plan->compact_wanted = false;  // a property of the plan: not asked again
    return BN_OK;
  }
  bn::CompactVerdict* host = static_cast<bn::CompactVerdict*>(plan->compact_verdict_host);
  bn::CompactVerdict* verdict = reinterpret_cast<bn::CompactVerdict*>(plan->compact_states + (K * (size_t)p.N + 63) / 64 * 64);
  prof_begin("evidence_compaction", (double)K * (double)p.N * 9.0, s);
  const int64_t quads = (int64_t)K * ((p.N + 3) / 4);
  const int32_t blocks = (int32_t)std::min<int64_t>(kBlocks, (quads + 255) / 256);
  bn::launch(bn::evidence_two_rows, dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<const float*>(base), stride / 4,
             p.N, (int64_t)K, plan->compact_states, verdict);
  BN_LAUNCH_CHECK("evidence_two_rows");
  prof_end(s);
  BN_HIP_CHECK(hipMemcpyAsync(host, verdict, (size_t)blocks * sizeof(bn::CompactVerdict), hipMemcpyDeviceToHost, s));
  if (may_guess && plan->compact_have_rows) {
    // of the same kind as the last batch, with the same rows - to be confirmed by settle_guess
    if (!plan->compact_event) BN_HIP_CHECK(hipEventCreateWithFlags(&plan->compact_event, hipEventDisableTiming));
    BN_HIP_CHECK(hipEventRecord(plan->compact_event, s));
    plan->compact_guessing = true;
    plan->compact_blocks = blocks;
    *states = plan->compact_states;
    ++plan->compactions;
    return BN_OK;
  }
  BN_HIP_CHECK(hipStreamSynchronize(s));
  plan->compact_have_rows = compact_verdict_rows(plan, blocks, plan->compact_rows);
  if (!plan->compact_have_rows) {
    plan->compact_skip = 63;  // evidence of another kind: the next batches are taken as they come
    return BN_OK;
  }
  *states = plan->compact_states;
  ++plan->compactions;
  return BN_OK;
}

// The end of a call that ran on a guess (compact_evidence): did the guess hold?
int settle_guess(bn_plan* plan, bool* held) {
  *held = true;
  if (!plan->compact_guessing) return BN_OK;
  plan->compact_guessing = false;
  BN_HIP_CHECK(hipEventSynchronize(plan->compact_event));
  float rows[6];
  const bool two_rows = compact_verdict_rows(plan, plan->compact_blocks, rows);
  if (two_rows && std::memcmp(rows, plan->compact_rows, sizeof rows) == 0) return BN_OK;
  *held = false;
  plan->compact_have_rows = false;
  --plan->compactions;
  if (!two_rows) plan->compact_skip = 63;
  return BN_OK;
}

// Evidence of a whole-step call (bn_em_step, bn_em_step_statistics): read by the first iteration itself
// where bn_em_statistics_from[_states] would do so, and - the plan being reset when the call ends - by
// every launch from the caller's arrays where the kernels for that exist (no copy at all).
static int step_evidence(bn_plan* plan, const void* const* evidence_dev, const uint8_t* states_dev, double on_value,
                  double off_value, int32_t num_iterations, void* stream, const float* rows = nullptr) {
  bn::Plan& p = plan->p;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bool ingest = !p.observed.empty() && num_iterations >= 3 && p.N >= 2 && p.N % 2 == 0;
  if (ingest && states_dev) {
    ingest = reinterpret_cast<uintptr_t>(states_dev) % 2 == 0 && bn::ingest_states_possible(p);
  } else if (ingest) {
    ingest = bn::ingest_possible(p);
    for (size_t i = 0; i < p.observed.size() && ingest; ++i)
      ingest = evidence_dev[i] && reinterpret_cast<uintptr_t>(evidence_dev[i]) % 32 == 0;
  }
  if (!ingest) {
    const int rc = states_dev ? bn_enter_evidence_states(plan, states_dev, on_value, off_value, stream)
                              : bn_enter_evidence(plan, evidence_dev, stream);
    if (rc != BN_OK) return rc;
  } else {
    const bool graphs_possible = step_graphs_enabled() && plan->graphs_wanted && !plan->graphs_off;
    if (int rc = upload_family_sources(p, states_dev ? nullptr : evidence_dev, s, states_dev, !graphs_possible)) return rc;
    p.obs_pending = false;  // the first factor pass writes every observation array
    if (states_dev) {
      bn::set_ingest_next(p, true, true, on_value, off_value);
      // ... and neither would anything read a copy of the state bytes
      if (rows)  // (state bytes made from float evidence: compact_evidence has made sure the step runs on them)
        bn::set_states_in_caller(p, bn::caller_states_possible(p, num_iterations), rows);
      else
        bn::set_states_in_caller(p, bn::caller_states_possible(p, num_iterations), on_value, off_value);
    } else {
      bn::set_ingest_next(p, true);
      // the plan is reset when the step ends: nothing would ever read its copy of the evidence
      bn::set_evidence_in_caller(p, bn::caller_evidence_possible(p, num_iterations));
    }
  }
  return settle_observations(p, s);
}

int bn_em_step(bn_plan* plan, const void* const* evidence_dev, const uint8_t* states_dev, double on_value,
               double off_value, int32_t num_iterations, int32_t average, double learning_rate, double regularization,
               double* ll_out_dev, void* new_cpts_dev, void* stream) {
  if (!plan || !ll_out_dev || !new_cpts_dev) return fail(BN_ERR_INVALID, "null argument");
  if (num_iterations < 1) return fail(BN_ERR_INVALID, "num_iterations must be >= 1");
  if (regularization < 0.0 || regularization > 1.0) return fail(BN_ERR_INVALID, "regularization must be in [0, 1]");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  if (!p.observed.empty() && !evidence_dev && !states_dev) return fail(BN_ERR_INVALID, "no evidence given");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // messages as a machine built for this batch would have them
  if (int rc = p.dtype == BN_F32 ? reset_messages_t<float>(p, s) : reset_messages_t<double>(p, s)) return rc;
  // Small batch of a star network: the whole step as one cooperative launch (kernels_star.h)
  if (star_steps_enabled() && plan->star_wanted && num_iterations >= 2 && !g_profiling.load(std::memory_order_relaxed) && !plan->graphs_off) {
    if (int rc = prepare_star(plan)) return rc;
    if (plan->star.ok && *plan->star.error != 0) {
      // An earlier one-launch step could not synchronise its blocks (the GPU was shared with other
      // work).  That step changed nothing (kernels_star.h: parameters as before, NaN log-likelihood);
      // it is counted, and the plan launches per pass from now on.
      plan->star.ok = false;
      ++plan->star_failures;
      *plan->star.error = 0;
    }
    if (plan->star.ok) {
      // float evidence in 16-byte aligned arrays is read where it lies (the plan's copy would be
      // void after the step anyway); anything else goes into the observation arrays first
      bool from_caller = !p.observed.empty() && evidence_dev && !states_dev;
      for (size_t i = 0; i < p.observed.size() && from_caller; ++i)
        from_caller = evidence_dev[i] && reinterpret_cast<uintptr_t>(evidence_dev[i]) % 32 == 0;
      if (from_caller) {
        // (the streaming kernel takes a base address and a distance where the arrays lie evenly: no table)
        if (int rc = upload_family_sources(p, evidence_dev, s, nullptr, plan->star.stream)) return rc;
      } else {
        if (!p.observed.empty()) {
          const int rc = states_dev ? bn_enter_evidence_states(plan, states_dev, on_value, off_value, stream)
                                    : bn_enter_evidence(plan, evidence_dev, stream);
          if (rc != BN_OK) return rc;
        }
        if (int rc = settle_observations(p, s)) return rc;
      }
      const int launched = p.dtype == BN_F32
                               ? launch_star_for_d0<float>(plan, num_iterations, average, learning_rate, regularization,
                                                           ll_out_dev, new_cpts_dev, s, from_caller)
                               : launch_star_for_d0<double>(plan, num_iterations, average, learning_rate, regularization,
                                                            ll_out_dev, new_cpts_dev, s, from_caller);
      if (launched < 0) return launched;
      if (launched == BN_OK) {
        ++plan->star_launches;
        // new parameters, every stored message stale: as after bn_set_cpts + bn_reset_messages
        bn::cpts_changed_fast(p);
        return p.dtype == BN_F32 ? reset_messages_t<float>(p, s) : reset_messages_t<double>(p, s);
      }
      plan->star.ok = false;  // the runtime refused the cooperative launch: launch per pass from now on
    }
  }
  const uint8_t* compacted = nullptr;  // state bytes made from float evidence (they stand for plan->compact_rows)
  if (!states_dev) {
    if (int rc = compact_evidence(plan, evidence_dev, num_iterations, s, &compacted, true)) return rc;
    states_dev = compacted;
  }
  if (plan->compact_guessing) {
    // the parameters as they are, should the guess turn out wrong
    const size_t bytes = (size_t)p.cpt_total * p.elem_size;
    if (!plan->compact_cpts_backup) BN_HIP_CHECK(hipMalloc(&plan->compact_cpts_backup, std::max<size_t>(bytes, 1)));
    BN_HIP_CHECK(hipMemcpyAsync(plan->compact_cpts_backup, p.cpts, bytes, hipMemcpyDeviceToDevice, s));
  }
  if (int rc = step_evidence(plan, evidence_dev, states_dev, on_value, off_value, num_iterations, stream,
                             compacted ? plan->compact_rows : nullptr))
    return rc;
  if (!plan->accum_slot) {
    BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&plan->accum_slot), (size_t)std::max<int64_t>(p.cpt_total, 1) * sizeof(double)));
    p.device_bytes += p.cpt_total * 8;
  }
  std::string key = "step";
  key.append(reinterpret_cast<const char*>(&learning_rate), sizeof learning_rate);
  key.append(reinterpret_cast<const char*>(&regularization), sizeof regularization);
  bool from_graph = false;
  int32_t ll_blocks = 0;
  // the M-step launch also sums the log-likelihood partials into the caller's slot (not when it runs
  // from a graph: the caller's address may differ from call to call)
  bool ll_written = false;
  const double ll_scale = average ? 1.0 / (double)p.N : 1.0;
  const int rc = run_sequence(
      plan, key, num_iterations, s, &from_graph, &ll_blocks, [&](hipStream_t on, bool in_graph, int32_t* blocks) {
        t_discard_last_messages = true;
        int rc2 = p.dtype == BN_F32 ? iterate_t<float>(p, num_iterations, 1, on) : iterate_t<double>(p, num_iterations, 1, on);
        t_discard_last_messages = false;
        if (rc2) return rc2;
        if (finish_shape_ok(plan)) {
          // sums and M-step in one launch (the caller's addresses stay out of a captured graph)
          rc2 = p.dtype == BN_F32
                    ? launch_finish<float>(plan, learning_rate, regularization, ll_scale, in_graph ? nullptr : ll_out_dev,
                                           in_graph ? nullptr : new_cpts_dev, on, blocks)
                    : launch_finish<double>(plan, learning_rate, regularization, ll_scale, in_graph ? nullptr : ll_out_dev,
                                            in_graph ? nullptr : new_cpts_dev, on, blocks);
          if (rc2 < 0) return rc2;
          if (rc2 == BN_OK) {
            ll_written = !in_graph;
            bn::cpts_changed_fast(p);
            return p.dtype == BN_F32 ? reset_messages_t<float>(p, on) : reset_messages_t<double>(p, on);
          }
        }
        rc2 = statistics_sums(p, plan->accum_slot, on, blocks);
        if (rc2) return rc2;
        // M-step on the plan's own snapshot, in place; the caller's copy is written by the same kernel
        // unless the kernel runs from a graph (the caller's address may differ from call to call)
        const dim3 grid((uint32_t)((p.n_rows + 255) / 256));
        prof_begin("m_step", 0.0, on);
        if (p.dtype == BN_F32) {
          bn::launch(bn::m_step_rows<float>, grid, dim3(256), 0, on, (const double*)plan->accum_slot,
                     static_cast<const float*>(p.cpts), p.d_row_off, p.d_row_len, p.n_rows, learning_rate, regularization,
                     static_cast<float*>(p.cpts), in_graph ? nullptr : static_cast<float*>(new_cpts_dev),
                     (const double*)p.scratch, *blocks, ll_scale, in_graph ? nullptr : ll_out_dev);
        } else {
          bn::launch(bn::m_step_rows<double>, grid, dim3(256), 0, on, (const double*)plan->accum_slot,
                     static_cast<const double*>(p.cpts), p.d_row_off, p.d_row_len, p.n_rows, learning_rate,
                     regularization, static_cast<double*>(p.cpts), in_graph ? nullptr : static_cast<double*>(new_cpts_dev),
                     (const double*)p.scratch, *blocks, ll_scale, in_graph ? nullptr : ll_out_dev);
        }
        ll_written = !in_graph;
        BN_LAUNCH_CHECK("m_step_rows");
        prof_end(on);
        // with new parameters every stored message is stale: the plan is as after bn_set_cpts + bn_reset_messages
        bn::cpts_changed_fast(p);
        return p.dtype == BN_F32 ? reset_messages_t<float>(p, on) : reset_messages_t<double>(p, on);
      });
  bn::set_ingest_next(p, false);
  bn::set_evidence_in_caller(p, false);
  bn::set_states_in_caller(p, false);
  if (rc != BN_OK) return rc;
  if (from_graph)
    BN_HIP_CHECK(hipMemcpyAsync(new_cpts_dev, p.cpts, (size_t)p.cpt_total * p.elem_size, hipMemcpyDeviceToDevice, s));
  if (!ll_written) {
    prof_begin("log_likelihood", (double)p.N * 8.0, s);
    bn::launch(bn::sum_partials, dim3(1), dim3(256), 0, s, p.scratch, ll_blocks, ll_scale, ll_out_dev);
    BN_LAUNCH_CHECK("sum_partials");
    prof_end(s);
  }
  bool held = true;
  if (int rc2 = settle_guess(plan, &held)) return rc2;
  if (!held) {
    // the batch was not what it was taken for: the parameters as before, and the step again as the batch is
    BN_HIP_CHECK(hipMemcpyAsync(p.cpts, plan->compact_cpts_backup, (size_t)p.cpt_total * p.elem_size, hipMemcpyDeviceToDevice, s));
    bn::cpts_changed_fast(p);
    return bn_em_step(plan, evidence_dev, nullptr, on_value, off_value, num_iterations, average, learning_rate, regularization,
                      ll_out_dev, new_cpts_dev, stream);
  }
  return BN_OK;
}

int bn_em_step_statistics(bn_plan* plan, const void* const* evidence_dev, const uint8_t* states_dev, double on_value,
                          double off_value, int32_t num_iterations, int32_t average, double* ll_out_dev,
                          double* accum_dev, void* stream) {
  if (!plan || !ll_out_dev || !accum_dev) return fail(BN_ERR_INVALID, "null argument");
  if (num_iterations < 1) return fail(BN_ERR_INVALID, "num_iterations must be >= 1");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  if (!p.cpts_set) return fail(BN_ERR_STATE, "bn_set_cpts has not been called");
  if (!p.observed.empty() && !evidence_dev && !states_dev) return fail(BN_ERR_INVALID, "no evidence given");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = p.dtype == BN_F32 ? reset_messages_t<float>(p, s) : reset_messages_t<double>(p, s)) return rc;
  struct NoGraphs {  // the launches below are not captured (a graph would keep this call's addresses)
    bn_plan* plan;
    bool before;
    ~NoGraphs() { plan->graphs_off = before; }
  } no_graphs{plan, plan->graphs_off};
  plan->graphs_off = true;
  const uint8_t* compacted = nullptr;  // state bytes made from float evidence (they stand for plan->compact_rows)
  if (!states_dev) {
    if (int rc = compact_evidence(plan, evidence_dev, num_iterations, s, &compacted, true)) return rc;
    states_dev = compacted;
  }
  if (int rc = step_evidence(plan, evidence_dev, states_dev, on_value, off_value, num_iterations, stream,
                             compacted ? plan->compact_rows : nullptr))
    return rc;
  int32_t ll_blocks = 0;
  const double ll_scale = average ? 1.0 / (double)p.N : 1.0;
  t_discard_last_messages = true;
  int rc = BN_OK;
  bool finished = false;
  if (finish_shape_ok(plan)) {
    // star plans: all the sums in one launch, as in bn_em_step (there with the M-step behind them)
    rc = p.dtype == BN_F32 ? iterate_t<float>(p, num_iterations, 1, s) : iterate_t<double>(p, num_iterations, 1, s);
    if (rc == BN_OK) {
      const int done = p.dtype == BN_F32
                           ? launch_finish<float>(plan, -1.0, 0.0, ll_scale, ll_out_dev, nullptr, s, &ll_blocks, accum_dev, true)
                           : launch_finish<double>(plan, -1.0, 0.0, ll_scale, ll_out_dev, nullptr, s, &ll_blocks, accum_dev, true);
      if (done < 0) rc = done;
      finished = done == BN_OK;
      if (rc == BN_OK && !finished) rc = statistics_sums(p, accum_dev, s, &ll_blocks);
    }
  } else {
    rc = statistics_launches(p, num_iterations, 1, accum_dev, s, &ll_blocks);
  }
  t_discard_last_messages = false;
  bn::set_ingest_next(p, false);
  bn::set_evidence_in_caller(p, false);
  bn::set_states_in_caller(p, false);
  if (rc != BN_OK) return rc;
  if (!finished) {
    prof_begin("log_likelihood", (double)p.N * 8.0, s);
    bn::launch(bn::sum_partials, dim3(1), dim3(256), 0, s, p.scratch, ll_blocks, ll_scale, ll_out_dev);
    BN_LAUNCH_CHECK("sum_partials");
    prof_end(s);
  }
  // nobody will query this batch's messages: the plan is as after bn_reset_messages
  if (int rc2 = p.dtype == BN_F32 ? reset_messages_t<float>(p, s) : reset_messages_t<double>(p, s)) return rc2;
  bool held = true;
  if (int rc2 = settle_guess(plan, &held)) return rc2;
  if (!held)  // the batch was not what it was taken for: the sums again as the batch is
    return bn_em_step_statistics(plan, evidence_dev, nullptr, on_value, off_value, num_iterations, average, ll_out_dev, accum_dev,
                                 stream);
  return BN_OK;
}

int bn_m_step(bn_plan* plan, const double* count_sums_dev, const void* old_cpts_dev, double learning_rate,
              double regularization, void* new_cpts_dev, void* stream) {
  if (!plan || !count_sums_dev || !new_cpts_dev) return fail(BN_ERR_INVALID, "null argument");
  if (learning_rate >= 0.0 && !old_cpts_dev) return fail(BN_ERR_INVALID, "old_cpts_dev is null but a learning rate is given");
  if (regularization < 0.0 || regularization > 1.0) return fail(BN_ERR_INVALID, "regularization must be in [0, 1]");
  bn::Plan& p = plan->p;
  if (int rc = on_plan_device(p)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((uint32_t)((p.n_rows + 255) / 256));
  prof_begin("m_step", 0.0, s);
  if (p.dtype == BN_F32) {
    bn::launch(bn::m_step_rows<float>, grid, dim3(256), 0, s, count_sums_dev, static_cast<const float*>(old_cpts_dev),
               p.d_row_off, p.d_row_len, p.n_rows, learning_rate, regularization, static_cast<float*>(new_cpts_dev),
               static_cast<float*>(nullptr), static_cast<const double*>(nullptr), 0, 1.0, static_cast<double*>(nullptr));
  } else {
    bn::launch(bn::m_step_rows<double>, grid, dim3(256), 0, s, count_sums_dev, static_cast<const double*>(old_cpts_dev),
               p.d_row_off, p.d_row_len, p.n_rows, learning_rate, regularization, static_cast<double*>(new_cpts_dev),
               static_cast<double*>(nullptr), static_cast<const double*>(nullptr), 0, 1.0, static_cast<double*>(nullptr));
  }
  BN_LAUNCH_CHECK("m_step_rows");
  prof_end(s);
  return BN_OK;
}

int bn_profile_enable(int32_t enabled) {
  g_profiling.store(enabled != 0);
  bn::launch_event_source() = enabled ? &next_launch_events : nullptr;
  return BN_OK;
}

int bn_profile_only(const char* label_prefix) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof_prefix = label_prefix ? label_prefix : "";
  return BN_OK;
}

int bn_profile_stride(int32_t every) {
  if (every < 1) return fail(BN_ERR_INVALID, "stride must be at least 1");
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  g_prof_stride = every;
  g_prof_seen = 0;
  return BN_OK;
}

int bn_profile_report(char* buffer, int64_t capacity) {
  if (!buffer || capacity < 3) return fail(BN_ERR_INVALID, "bad argument");
  std::vector<ProfRecord> records;
  {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    records.swap(g_prof);
  }
  std::vector<std::string> labels;
  std::vector<double> total_ms, alg_bytes, alg_flops;
  std::vector<int64_t> launches;
  std::vector<std::vector<const void*>> label_kernels;
  for (auto& r : records) {
    double region_ms = 0.0;
    bool ok = !r.events.empty();
    for (size_t e = 0; e + 1 < r.events.size(); e += 2) {
      float ms = 0.f;
      if (hipEventSynchronize(r.events[e + 1]) == hipSuccess &&
          hipEventElapsedTime(&ms, r.events[e], r.events[e + 1]) == hipSuccess) {
        region_ms += ms;
      } else {
        ok = false;
      }
    }
    if (ok) {
      size_t i = std::find(labels.begin(), labels.end(), r.label) - labels.begin();
      if (i == labels.size()) {
        labels.push_back(r.label);
        total_ms.push_back(0.0);
        launches.push_back(0);
        alg_bytes.push_back(r.alg_bytes);
        alg_flops.push_back(r.alg_flops);
        label_kernels.emplace_back();
      }
      total_ms[i] += region_ms;
      launches[i] += 1;
      for (const void* k : r.kernels)
        if (std::find(label_kernels[i].begin(), label_kernels[i].end(), k) == label_kernels[i].end())
          label_kernels[i].push_back(k);
    }
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (hipEvent_t e : r.events) g_event_pool.push_back(e);
  }
  std::string json = "[";
  for (size_t i = 0; i < labels.size(); ++i) {
    // what the compiler made of the label's kernels: registers, static LDS and spill space of each
    // distinct instantiation, sorted - it changes whenever the generated code does
    std::vector<std::string> sigs;
    for (const void* k : label_kernels[i]) {
      hipFuncAttributes attr;
      if (hipFuncGetAttributes(&attr, k) != hipSuccess) {
        (void)hipGetLastError();
        continue;
      }
      char one[96];
      std::snprintf(one, sizeof(one), "v%d.l%zu.s%zu", attr.numRegs, (size_t)attr.sharedSizeBytes, (size_t)attr.localSizeBytes);
      sigs.push_back(one);
    }
    std::sort(sigs.begin(), sigs.end());
    std::string signature;
    for (const auto& one : sigs) signature += (signature.empty() ? "" : "|") + one;
    char num[200];
    std::snprintf(num, sizeof(num), "\", \"launches\": %lld, \"total_ms\": %.6f, \"alg_bytes\": %.1f, \"alg_flops\": %.1f, ",
                  (long long)launches[i], total_ms[i], alg_bytes[i], alg_flops[i]);
    json += std::string(i ? ", " : "") + "{\"kernel\": \"" + labels[i] + num + "\"signature\": \"" + signature + "\"}";
  }
  json += "]";
  if ((int64_t)json.size() + 1 > capacity) json = "[]";
  std::memcpy(buffer, json.c_str(), json.size() + 1);
  return BN_OK;
}

}  // extern "C"
