// Exact inference by enumeration of the joint distribution: the on-device counterpart of the
// reference's brute-force machine (inference_machines/naive/naive_inference_machine.py:37-122),
// meant for checking sum-product results on small networks (trees, polytrees, loopy networks)
// where no stored answer exists.
//
// The reference materialises p(x_1..x_n) * prod_i e_i[n, x_i] as an [N, d_1, .., d_n] tensor and
// sums axes away.  Here a query over a node set Q is evaluated as
//
//     out[n, q] = ( sum_{o} table[q, o] * prod_{i observed, not in Q} e_i[n, o_i] )
//                 * prod_{i observed, in Q} e_i[n, q_i]  /  c[n]
//
//     table[q, o] = sum_{h} prod_nodes cpt_node[...]          (h = unobserved nodes not in Q)
//
// so everything that does not depend on the observation (the sum over nuisance nodes) is done
// once per query by `exact_table`, and the per-observation kernel only walks the joint states
// of the observed nodes.  c[n] is the same expression with Q empty (the probability of the
// evidence; its log is the reference's per-observation log-likelihood, :82-88).
//
// Evidence is kept transposed, one row per (observed node, state): ev[row][n], so a wave reads
// consecutive observations of one row; the enumeration index is uniform over the wave.
// All sums are carried in double, whatever the element type.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace bn {

constexpr int kExactMaxNodes = 32;
constexpr int kExactMaxParentRefs = 128;
constexpr int64_t kExactMaxJoint = int64_t(1) << 30;   // joint states enumerated per query
constexpr int64_t kExactMaxTable = int64_t(1) << 24;   // table entries per query

struct ExactNet {
  int32_t n_nodes;
  int32_t states[kExactMaxNodes];
  int32_t cpt_off[kExactMaxNodes];          // element offset of the node's CPT in the flat buffer
  int32_t par_begin[kExactMaxNodes + 1];    // parents of node j: par[par_begin[j] .. par_begin[j+1])
  int32_t par[kExactMaxParentRefs];         // in CPT axis order
  int32_t ev_row[kExactMaxNodes];           // first evidence row of the node, -1 = unobserved
};

// Enumeration order of one query: query nodes (output axes, first = slowest), then observed
// nodes outside the query, then the remaining nodes.
struct ExactQuery {
  int32_t order[kExactMaxNodes];
  int32_t n_query, n_seen, n_hidden;
  int64_t bins, r_seen, r_hidden;
};

// table[(q * r_seen + o)] = sum over hidden joint states of the product of all CPT entries.
// One block per table entry; its threads stride over the hidden joint states.
template <typename T>
__global__ __launch_bounds__(256) void exact_table(ExactNet net, ExactQuery qy, const T* __restrict__ cpts,
                                                   double* __restrict__ table) {
  __shared__ double partial[256];
  const int64_t entry = blockIdx.x;
  int32_t digit[kExactMaxNodes];
  {
    int64_t rem = entry;
    for (int32_t k = qy.n_query + qy.n_seen - 1; k >= 0; --k) {
      const int32_t node = qy.order[k];
      digit[node] = (int32_t)(rem % net.states[node]);
      rem /= net.states[node];
    }
  }
  double acc = 0.0;
  const int32_t first_hidden = qy.n_query + qy.n_seen;
  for (int64_t h = threadIdx.x; h < qy.r_hidden; h += blockDim.x) {
    int64_t rem = h;
    for (int32_t k = first_hidden + qy.n_hidden - 1; k >= first_hidden; --k) {
      const int32_t node = qy.order[k];
      digit[node] = (int32_t)(rem % net.states[node]);
      rem /= net.states[node];
    }
    double prob = 1.0;
    for (int32_t j = 0; j < net.n_nodes; ++j) {
      int32_t idx = 0;
      for (int32_t a = net.par_begin[j]; a < net.par_begin[j + 1]; ++a) idx = idx * net.states[net.par[a]] + digit[net.par[a]];
      idx = idx * net.states[j] + digit[j];
      prob *= (double)cpts[net.cpt_off[j] + idx];
    }
    acc += prob;
  }
  partial[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) partial[threadIdx.x] += partial[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) table[entry] = partial[0];
}

// One thread per observation; blockIdx.y strides over the output bins.
// OUT = double with evidence_prob == nullptr computes c[n] (bins = 1); otherwise the normalised
// posterior out[n * bins + q] in the element type.
template <typename T, typename OUT>
__global__ __launch_bounds__(256) void exact_posterior(ExactNet net, ExactQuery qy, const double* __restrict__ table,
                                                       const T* __restrict__ ev, int64_t ld, int64_t n_obs,
                                                       const double* __restrict__ evidence_prob,
                                                       OUT* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_obs) return;
  const double inv_c = evidence_prob ? 1.0 / evidence_prob[n] : 1.0;
  for (int64_t q = blockIdx.y; q < qy.bins; q += gridDim.y) {
    double acc = 0.0;
    for (int64_t o = 0; o < qy.r_seen; ++o) {
      double w = table[q * qy.r_seen + o];
      int64_t rem = o;
      for (int32_t k = qy.n_query + qy.n_seen - 1; k >= qy.n_query; --k) {
        const int32_t node = qy.order[k];
        const int32_t d = (int32_t)(rem % net.states[node]);
        rem /= net.states[node];
        w *= (double)ev[(int64_t)(net.ev_row[node] + d) * ld + n];
      }
      acc += w;
    }
    int64_t rem = q;
    for (int32_t k = qy.n_query - 1; k >= 0; --k) {
      const int32_t node = qy.order[k];
      const int32_t d = (int32_t)(rem % net.states[node]);
      rem /= net.states[node];
      if (net.ev_row[node] >= 0) acc *= (double)ev[(int64_t)(net.ev_row[node] + d) * ld + n];
    }
    out[n * qy.bins + q] = (OUT)(acc * inv_c);
  }
}

// ev[(row0 + s) * ld + n] = src[n * states + s]
template <typename T>
__global__ __launch_bounds__(256) void exact_evidence_rows(const T* __restrict__ src, int32_t states, int32_t row0,
                                                           int64_t ld, int64_t n_obs, T* __restrict__ ev) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_obs) return;
  for (int32_t s = 0; s < states; ++s) ev[(int64_t)(row0 + s) * ld + n] = src[n * states + s];
}

template <typename T>
__global__ __launch_bounds__(256) void exact_fill(T* __restrict__ dst, int64_t count, T value) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = value;
}

// out_each[n] = log c[n] (element type, may be null); out_total = sum_n log c[n] * scale (may be null).
template <typename T>
__global__ __launch_bounds__(1024) void exact_log_evidence(const double* __restrict__ evidence_prob, int64_t n_obs,
                                                           double scale, T* __restrict__ out_each,
                                                           double* __restrict__ out_total) {
  __shared__ double partial[1024];
  double acc = 0.0;
  for (int64_t n = threadIdx.x; n < n_obs; n += blockDim.x) {
    const double l = log(evidence_prob[n]);
    if (out_each) out_each[n] = (T)l;
    acc += l;
  }
  partial[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 512; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) partial[threadIdx.x] += partial[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0 && out_total) *out_total = partial[0] * scale;
}

}  // namespace bn
