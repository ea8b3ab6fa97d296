// bn_exact_* entry points (declared in include/bn_hip.h): exact inference by enumeration.
#include "../../include/bn_hip.h"

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "error.h"
#include "exact.h"
#include "launch.h"

struct bn_exact {
  int dtype = BN_F32;
  size_t elem = 4;
  int device = 0;
  bn::ExactNet net{};
  std::vector<int32_t> observed;
  int32_t ev_rows = 0;
  int64_t cpt_total = 0;
  void* cpts = nullptr;            // snapshot of the flat CPT buffer
  bool have_cpts = false;
  int64_t n_obs = 0;               // 0 = no evidence entered yet
  int64_t ld = 0;                  // allocated observations per evidence row
  void* ev = nullptr;              // [ev_rows][ld]
  double* evidence_prob = nullptr; // c[n]
  double* table = nullptr;
  int64_t table_cap = 0;
};

namespace {

using bn::fail;

#define EX_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t err__ = (expr);                                                              \
    if (err__ != hipSuccess)                                                                \
      return fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));        \
  } while (0)

#define EX_LAUNCHED(what)                                                                   \
  do {                                                                                      \
    hipError_t err__ = hipGetLastError();                                                   \
    if (err__ != hipSuccess)                                                                \
      return fail(BN_ERR_HIP, std::string("launch of ") + what + ": " + hipGetErrorString(err__)); \
  } while (0)

int on_device(const bn_exact* x) {
  int current = -1;
  EX_HIP(hipGetDevice(&current));
  if (current != x->device) return fail(BN_ERR_STATE, "exact machine used on a different device than it was created on");
  return BN_OK;
}

// Enumeration order for a query: its nodes, then the other observed nodes, then the rest.
int make_query(const bn_exact* x, const int32_t* nodes, int32_t count, bn::ExactQuery& q) {
  const bn::ExactNet& net = x->net;
  std::vector<char> taken(net.n_nodes, 0);
  q = bn::ExactQuery{};
  q.bins = q.r_seen = q.r_hidden = 1;
  int32_t k = 0;
  for (int32_t i = 0; i < count; ++i) {
    const int32_t node = nodes[i];
    if (node < 0 || node >= net.n_nodes) return fail(BN_ERR_INVALID, "query node id out of range");
    if (taken[node]) return fail(BN_ERR_INVALID, "a node appears twice in the query");
    taken[node] = 1;
    q.order[k++] = node;
    q.bins *= net.states[node];
  }
  q.n_query = count;
  for (int32_t j = 0; j < net.n_nodes; ++j)
    if (!taken[j] && net.ev_row[j] >= 0) {
      q.order[k++] = j;
      q.r_seen *= net.states[j];
      ++q.n_seen;
    }
  for (int32_t j = 0; j < net.n_nodes; ++j)
    if (!taken[j] && net.ev_row[j] < 0) {
      q.order[k++] = j;
      q.r_hidden *= net.states[j];
      ++q.n_hidden;
    }
  if (q.bins * q.r_seen > bn::kExactMaxTable)
    return fail(BN_ERR_INVALID, "query too large for enumeration (query and observed nodes span more than 2^24 joint states)");
  return BN_OK;
}

int ensure_table(bn_exact* x, int64_t entries) {
  if (entries <= x->table_cap) return BN_OK;
  if (x->table) EX_HIP(hipFree(x->table));
  x->table = nullptr;
  x->table_cap = 0;
  EX_HIP(hipMalloc(reinterpret_cast<void**>(&x->table), (size_t)entries * sizeof(double)));
  x->table_cap = entries;
  return BN_OK;
}

template <typename T, typename OUT>
int run_query(bn_exact* x, const bn::ExactQuery& q, const double* evidence_prob, OUT* out, hipStream_t stream) {
  const int64_t entries = q.bins * q.r_seen;
  int rc = ensure_table(x, entries);
  if (rc != BN_OK) return rc;
  bn::launch(bn::exact_table<T>, dim3((unsigned)entries), dim3(256), 0, stream, x->net, q,
             static_cast<const T*>(x->cpts), x->table);
  EX_LAUNCHED("exact_table");
  const unsigned blocks_x = (unsigned)((x->n_obs + 255) / 256);
  const unsigned blocks_y = (unsigned)std::min<int64_t>(q.bins, 4096);
  bn::launch(bn::exact_posterior<T, OUT>, dim3(blocks_x, blocks_y), dim3(256), 0, stream, x->net, q,
             static_cast<const double*>(x->table), static_cast<const T*>(x->ev), x->ld, x->n_obs, evidence_prob, out);
  EX_LAUNCHED("exact_posterior");
  return BN_OK;
}

template <typename T>
int enter_evidence_t(bn_exact* x, int64_t n_obs, const void* const* evidence_dev, hipStream_t stream) {
  if (n_obs > x->ld || !x->evidence_prob) {
    if (x->ev) EX_HIP(hipFree(x->ev));
    if (x->evidence_prob) EX_HIP(hipFree(x->evidence_prob));
    x->ev = nullptr;
    x->evidence_prob = nullptr;
    x->ld = 0;
    const int64_t ld = (n_obs + 63) / 64 * 64;
    EX_HIP(hipMalloc(&x->ev, (size_t)std::max<int64_t>(x->ev_rows, 1) * ld * sizeof(T)));
    EX_HIP(hipMalloc(reinterpret_cast<void**>(&x->evidence_prob), (size_t)ld * sizeof(double)));
    x->ld = ld;
  }
  x->n_obs = n_obs;
  const unsigned blocks = (unsigned)((n_obs + 255) / 256);
  for (size_t i = 0; i < x->observed.size(); ++i) {
    const int32_t node = x->observed[i];
    if (evidence_dev) {
      bn::launch(bn::exact_evidence_rows<T>, dim3(blocks), dim3(256), 0, stream, static_cast<const T*>(evidence_dev[i]),
                 x->net.states[node], x->net.ev_row[node], x->ld, n_obs, static_cast<T*>(x->ev));
    } else {
      // no evidence: all-ones rows, the reference's state before enter_evidence (naive_inference_machine.py:31)
      T* row = static_cast<T*>(x->ev) + (int64_t)x->net.ev_row[node] * x->ld;
      const int64_t count = (int64_t)x->net.states[node] * x->ld;
      bn::launch(bn::exact_fill<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, row, count, T(1));
    }
    EX_LAUNCHED("exact evidence rows");
  }
  bn::ExactQuery q;
  int rc = make_query(x, nullptr, 0, q);
  if (rc != BN_OK) return rc;
  return run_query<T, double>(x, q, nullptr, x->evidence_prob, stream);
}

}  // namespace

extern "C" {

int bn_exact_create(int dtype, int32_t num_nodes, const int32_t* num_states, const int32_t* parent_offsets,
                    const int32_t* parent_indices, const int32_t* observed_nodes, int32_t num_observed,
                    bn_exact** out) {
  if (!out) return fail(BN_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (dtype != BN_F32 && dtype != BN_F64) return fail(BN_ERR_INVALID, "dtype must be BN_F32 or BN_F64");
  if (num_nodes < 1 || num_nodes > bn::kExactMaxNodes)
    return fail(BN_ERR_INVALID, "exact enumeration supports 1.." + std::to_string(bn::kExactMaxNodes) + " nodes");
  if (!num_states || !parent_offsets || (num_observed > 0 && !observed_nodes) || num_observed < 0)
    return fail(BN_ERR_INVALID, "NULL topology array");
  if (parent_offsets[0] != 0 || parent_offsets[num_nodes] > bn::kExactMaxParentRefs)
    return fail(BN_ERR_INVALID, "parent offsets must start at 0 and hold at most " +
                                    std::to_string(bn::kExactMaxParentRefs) + " entries");
  for (int32_t j = 0; j < num_nodes; ++j)
    if (parent_offsets[j + 1] < parent_offsets[j]) return fail(BN_ERR_INVALID, "parent offsets must not decrease");
  if (parent_offsets[num_nodes] > 0 && !parent_indices)
    return fail(BN_ERR_INVALID, "parent_indices is NULL although some node has parents");
  bn_exact* x = new (std::nothrow) bn_exact();
  if (!x) return fail(BN_ERR_INVALID, "out of host memory");
  x->dtype = dtype;
  x->elem = dtype == BN_F32 ? 4 : 8;
  bn::ExactNet& net = x->net;
  net.n_nodes = num_nodes;
  int64_t joint = 1, cpt_total = 0;
  for (int32_t j = 0; j < num_nodes; ++j) {
    if (num_states[j] < 1) {
      delete x;
      return fail(BN_ERR_INVALID, "a node needs at least one state");
    }
    net.states[j] = num_states[j];
    net.ev_row[j] = -1;
    joint *= num_states[j];
    if (joint > bn::kExactMaxJoint) {
      delete x;
      return fail(BN_ERR_INVALID, "network too large for enumeration (more than 2^30 joint states)");
    }
  }
  for (int32_t j = 0; j <= num_nodes; ++j) net.par_begin[j] = parent_offsets[j];
  for (int32_t j = 0; j < num_nodes; ++j) {
    if (parent_offsets[j + 1] < parent_offsets[j]) {
      delete x;
      return fail(BN_ERR_INVALID, "parent offsets must not decrease");
    }
    int64_t elems = num_states[j];
    for (int32_t a = parent_offsets[j]; a < parent_offsets[j + 1]; ++a) {
      const int32_t parent = parent_indices[a];
      if (parent < 0 || parent >= num_nodes || parent == j) {
        delete x;
        return fail(BN_ERR_INVALID, "parent id out of range");
      }
      net.par[a] = parent;
      elems *= num_states[parent];
    }
    net.cpt_off[j] = (int32_t)cpt_total;
    cpt_total += elems;
    if (cpt_total > (int64_t(1) << 30)) {
      delete x;
      return fail(BN_ERR_INVALID, "CPTs too large");
    }
  }
  x->cpt_total = cpt_total;
  int32_t rows = 0;
  for (int32_t i = 0; i < num_observed; ++i) {
    const int32_t node = observed_nodes[i];
    if (node < 0 || node >= num_nodes || net.ev_row[node] >= 0) {
      delete x;
      return fail(BN_ERR_INVALID, "observed node id out of range or listed twice");
    }
    net.ev_row[node] = rows;
    rows += num_states[node];
    x->observed.push_back(node);
  }
  x->ev_rows = rows;
  hipError_t err = hipGetDevice(&x->device);
  if (err == hipSuccess) err = hipMalloc(&x->cpts, (size_t)cpt_total * x->elem);
  if (err != hipSuccess) {
    delete x;
    return fail(BN_ERR_HIP, std::string("allocating the exact machine: ") + hipGetErrorString(err));
  }
  *out = x;
  return BN_OK;
}

int bn_exact_destroy(bn_exact* x) {
  if (!x) return BN_OK;
  (void)hipFree(x->cpts);
  (void)hipFree(x->ev);
  (void)hipFree(x->evidence_prob);
  (void)hipFree(x->table);
  delete x;
  return BN_OK;
}

int bn_exact_set_cpts(bn_exact* x, const void* cpts_dev, void* stream) {
  if (!x || !cpts_dev) return fail(BN_ERR_INVALID, "NULL argument");
  int rc = on_device(x);
  if (rc != BN_OK) return rc;
  EX_HIP(hipMemcpyAsync(x->cpts, cpts_dev, (size_t)x->cpt_total * x->elem, hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  x->have_cpts = true;
  return BN_OK;
}

int bn_exact_enter_evidence(bn_exact* x, int64_t num_observations, const void* const* evidence_dev, void* stream) {
  if (!x) return fail(BN_ERR_INVALID, "NULL argument");
  if (num_observations < 1) return fail(BN_ERR_INVALID, "num_observations must be at least 1");
  if (!x->have_cpts) return fail(BN_ERR_STATE, "bn_exact_set_cpts has not been called");
  if (evidence_dev)
    for (size_t i = 0; i < x->observed.size(); ++i)
      if (!evidence_dev[i]) return fail(BN_ERR_INVALID, "NULL evidence array");
  int rc = on_device(x);
  if (rc != BN_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return x->dtype == BN_F32 ? enter_evidence_t<float>(x, num_observations, evidence_dev, s)
                            : enter_evidence_t<double>(x, num_observations, evidence_dev, s);
}

int bn_exact_log_likelihood(bn_exact* x, int32_t average, double* total_out_dev, void* each_out_dev, void* stream) {
  if (!x) return fail(BN_ERR_INVALID, "NULL argument");
  if (x->n_obs == 0) return fail(BN_ERR_STATE, "no evidence entered");
  if (!total_out_dev && !each_out_dev) return fail(BN_ERR_INVALID, "both outputs are NULL");
  int rc = on_device(x);
  if (rc != BN_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double scale = average ? 1.0 / (double)x->n_obs : 1.0;
  if (x->dtype == BN_F32)
    bn::launch(bn::exact_log_evidence<float>, dim3(1), dim3(1024), 0, s, static_cast<const double*>(x->evidence_prob),
               x->n_obs, scale, static_cast<float*>(each_out_dev), total_out_dev);
  else
    bn::launch(bn::exact_log_evidence<double>, dim3(1), dim3(1024), 0, s, static_cast<const double*>(x->evidence_prob),
               x->n_obs, scale, static_cast<double*>(each_out_dev), total_out_dev);
  EX_LAUNCHED("exact_log_evidence");
  return BN_OK;
}

int bn_exact_marginal(bn_exact* x, const int32_t* nodes, int32_t num_query_nodes, void* out_dev, void* stream) {
  if (!x || !out_dev || (num_query_nodes > 0 && !nodes) || num_query_nodes < 0)
    return fail(BN_ERR_INVALID, "NULL argument");
  if (x->n_obs == 0) return fail(BN_ERR_STATE, "no evidence entered");
  int rc = on_device(x);
  if (rc != BN_OK) return rc;
  bn::ExactQuery q;
  rc = make_query(x, nodes, num_query_nodes, q);
  if (rc != BN_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return x->dtype == BN_F32 ? run_query<float, float>(x, q, x->evidence_prob, static_cast<float*>(out_dev), s)
                            : run_query<double, double>(x, q, x->evidence_prob, static_cast<double*>(out_dev), s);
}

}  // extern "C"
