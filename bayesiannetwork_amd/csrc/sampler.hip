// Ancestral sampling on the device: bn_sampler_* of include/bn_hip.h.
//
// One thread draws one joint sample: it walks the needed nodes in an order that puts parents before
// children, picks the CPT row its already-drawn parents select, and inverts that row's cumulative
// distribution at a uniform number from a counter-based generator (Philox-4x32-10 keyed by the
// seed, counter = (sample, position of the node in the walk)), so a draw depends on (seed, sample,
// node) only - not on the launch geometry or on which other nodes were asked for alongside it.
// Drawn states are kept [node][sample] (a wave reads and writes consecutive samples); a second
// kernel lays the requested nodes out as the caller's row-major [sample][node] int32 matrix.
#include "../../include/bn_hip.h"

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "error.h"

struct bn_sampler {
  int dtype = 0;
  int device = 0;
  int32_t n_nodes = 0;
  std::vector<int32_t> states;
  std::vector<std::vector<int32_t>> parents;
  std::vector<int64_t> cpt_off;
  int64_t cpt_total = 0;
  // device tables of the last request (rebuilt when the requested node list changes)
  std::vector<int32_t> request;
  int32_t n_walk = 0;
  int32_t* d_walk_states = nullptr;    // [n_walk] states of the node at each walk position
  int64_t* d_walk_cpt = nullptr;       // [n_walk] offset of its CPT in the flat buffer
  int32_t* d_walk_poff = nullptr;      // [n_walk + 1] CSR into d_walk_parents
  int32_t* d_walk_parents = nullptr;   // walk positions of the parents, CPT axis order
  int32_t* d_out_pos = nullptr;        // [n_request] walk position of each requested node
};

namespace {

using bn::fail;

#define BN_HIP_CHECK(expr)                                                            \
  do {                                                                                \
    hipError_t err__ = (expr);                                                        \
    if (err__ != hipSuccess)                                                          \
      return fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));  \
  } while (0)

struct Philox {
  uint32_t c[4];
  uint32_t k[2];
};

__device__ __forceinline__ void philox_round(Philox& s) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * s.c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c[1] ^ s.k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ s.c[3] ^ s.k[1];
  const uint32_t n3 = (uint32_t)p0;
  s.c[0] = n0;
  s.c[1] = n1;
  s.c[2] = n2;
  s.c[3] = n3;
  s.k[0] += 0x9E3779B9u;
  s.k[1] += 0xBB67AE85u;
}

// uniform in (0, 1) with 53 random bits, a function of (seed, sample, position) only
__device__ __forceinline__ double uniform53(uint64_t seed, uint64_t sample, uint32_t position) {
  Philox s;
  s.c[0] = (uint32_t)sample;
  s.c[1] = (uint32_t)(sample >> 32);
  s.c[2] = position;
  s.c[3] = 0x5a17u;
  s.k[0] = (uint32_t)seed;
  s.k[1] = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(s);
  const uint64_t bits = (((uint64_t)s.c[0] << 32) | s.c[1]) >> 11;  // 53 bits
  return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

// grid.x = ceil(num_samples / 256), block = 256
template <typename T>
__global__ void __launch_bounds__(256) sample_walk(const T* __restrict__ cpts, const int32_t* __restrict__ walk_states,
                                                  const int64_t* __restrict__ walk_cpt,
                                                  const int32_t* __restrict__ walk_poff,
                                                  const int32_t* __restrict__ walk_parents, int32_t n_walk,
                                                  int64_t num_samples, uint64_t seed, int32_t* __restrict__ drawn) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= num_samples) return;
  for (int32_t w = 0; w < n_walk; ++w) {
    const int32_t S = walk_states[w];
    // row of the CPT [parent_0, .., parent_{P-1}, state] selected by the drawn parents
    int64_t row = 0;
    for (int32_t q = walk_poff[w]; q < walk_poff[w + 1]; ++q) {
      const int32_t pw = walk_parents[q];
      row = row * walk_states[pw] + drawn[(int64_t)pw * num_samples + n];
    }
    const T* p = cpts + walk_cpt[w] + row * S;
    double total = 0.0;
    for (int32_t s = 0; s < S; ++s) total += (double)p[s];
    const double target = uniform53(seed, (uint64_t)n, (uint32_t)w) * total;
    double cum = 0.0;
    int32_t pick = S - 1;
    for (int32_t s = 0; s < S; ++s) {
      cum += (double)p[s];
      if (target < cum) {
        pick = s;
        break;
      }
    }
    // a state of probability zero is never returned (rounding could only put target at the very end)
    while (pick > 0 && (double)p[pick] <= 0.0) --pick;
    drawn[(int64_t)w * num_samples + n] = pick;
  }
}

// out[n][j] = drawn[out_pos[j]][n]: a 64 x 64 tile through LDS so that both sides are coalesced
__global__ void __launch_bounds__(256) sample_layout(const int32_t* __restrict__ drawn, const int32_t* __restrict__ out_pos,
                                                    int32_t n_out, int64_t num_samples, int32_t* __restrict__ out) {
  __shared__ int32_t tile[64][65];
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int32_t j0 = blockIdx.y * 64;
  const int32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int32_t r = ty; r < 64; r += 4) {  // r: requested node within the tile, tx: sample
    const int32_t j = j0 + r;
    const int64_t n = n0 + tx;
    tile[r][tx] = (j < n_out && n < num_samples) ? drawn[(int64_t)out_pos[j] * num_samples + n] : 0;
  }
  __syncthreads();
  for (int32_t r = ty; r < 64; r += 4) {  // r: sample within the tile, tx: requested node
    const int64_t n = n0 + r;
    const int32_t j = j0 + tx;
    if (j < n_out && n < num_samples) out[n * n_out + j] = tile[tx][r];
  }
}

void free_tables(bn_sampler* s) {
  for (void* p : {(void*)s->d_walk_states, (void*)s->d_walk_cpt, (void*)s->d_walk_poff, (void*)s->d_walk_parents,
                  (void*)s->d_out_pos})
    if (p) (void)hipFree(p);
  s->d_walk_states = nullptr;
  s->d_walk_cpt = nullptr;
  s->d_walk_poff = nullptr;
  s->d_walk_parents = nullptr;
  s->d_out_pos = nullptr;
  s->request.clear();
  s->n_walk = 0;
}

template <typename U>
int upload(const std::vector<U>& host, U** dev) {
  *dev = nullptr;
  const size_t count = std::max<size_t>(host.size(), 1);
  BN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dev), count * sizeof(U)));
  if (!host.empty()) BN_HIP_CHECK(hipMemcpy(*dev, host.data(), host.size() * sizeof(U), hipMemcpyHostToDevice));
  return BN_OK;
}

// Walk order for a request: every requested node and its ancestors, parents first (depth-first,
// iterative), in the order the reference's recursion visits them (samplers/torch_sampler.py:43-52).
// Host only.  position[j] = place of node j in `walk`, -1 for nodes the request does not need.
int walk_order(const std::vector<std::vector<int32_t>>& parents, const std::vector<int32_t>& request,
               std::vector<int32_t>& position, std::vector<int32_t>& walk) {
  const size_t n_nodes = parents.size();
  position.assign(n_nodes, -1);
  walk.clear();
  // A node is unvisited, open (its parents have been pushed and it waits below them) or positioned.
  // The open nodes of the stack are a chain in which each is a parent of the one below it, so
  // meeting an open node among the parents of the node on top is exactly a cycle; a dense graph
  // may push an unvisited node many times, which is not one.
  std::vector<char> open(n_nodes, 0);
  for (int32_t j : request) {
    std::vector<int32_t> stack{j};
    while (!stack.empty()) {
      const int32_t cur = stack.back();
      if (position[cur] >= 0) {
        stack.pop_back();
        continue;
      }
      bool ready = true;
      for (int32_t q : parents[cur]) {
        if (position[q] >= 0) continue;
        if (open[q] || open[cur]) return fail(BN_ERR_INVALID, "the parent relation has a cycle");
        stack.push_back(q);
        ready = false;
      }
      if (!ready) {
        open[cur] = 1;
        continue;
      }
      stack.pop_back();
      open[cur] = 0;
      position[cur] = (int32_t)walk.size();
      walk.push_back(cur);
    }
  }
  return BN_OK;
}

int build_tables(bn_sampler* s, const int32_t* nodes, int32_t count) {
  std::vector<int32_t> request(nodes, nodes + count);
  if (s->d_out_pos && request == s->request) return BN_OK;
  free_tables(s);
  std::vector<int32_t> position, walk;
  if (int rc = walk_order(s->parents, request, position, walk)) return rc;
  std::vector<int32_t> w_states, w_poff{0}, w_parents, out_pos;
  std::vector<int64_t> w_cpt;
  for (int32_t j : walk) {
    w_states.push_back(s->states[j]);
    w_cpt.push_back(s->cpt_off[j]);
    for (int32_t q : s->parents[j]) w_parents.push_back(position[q]);
    w_poff.push_back((int32_t)w_parents.size());
  }
  for (int32_t j : request) out_pos.push_back(position[j]);
  if (int rc = upload(w_states, &s->d_walk_states)) return rc;
  if (int rc = upload(w_cpt, &s->d_walk_cpt)) return rc;
  if (int rc = upload(w_poff, &s->d_walk_poff)) return rc;
  if (int rc = upload(w_parents, &s->d_walk_parents)) return rc;
  if (int rc = upload(out_pos, &s->d_out_pos)) return rc;
  s->request.swap(request);
  s->n_walk = (int32_t)walk.size();
  return BN_OK;
}

}  // namespace

extern "C" {

int bn_sampler_create(int dtype, int32_t num_nodes, const int32_t* num_states, const int32_t* parent_offsets,
                      const int32_t* parent_indices, bn_sampler** out_sampler) {
  if (!out_sampler) return fail(BN_ERR_INVALID, "out_sampler is null");
  *out_sampler = nullptr;
  if (dtype != BN_F32 && dtype != BN_F64) return fail(BN_ERR_INVALID, "dtype must be BN_F32 or BN_F64");
  if (num_nodes < 1 || !num_states || !parent_offsets) return fail(BN_ERR_INVALID, "bad topology arguments");
  if (parent_offsets[0] != 0) return fail(BN_ERR_INVALID, "parent_offsets must start at 0");
  if (parent_offsets[num_nodes] > 0 && !parent_indices) return fail(BN_ERR_INVALID, "parent_indices is null");
  bn_sampler* s = new (std::nothrow) bn_sampler();
  if (!s) return fail(BN_ERR_INVALID, "out of host memory");
  s->dtype = dtype;
  s->n_nodes = num_nodes;
  s->states.assign(num_states, num_states + num_nodes);
  s->parents.resize(num_nodes);
  for (int32_t j = 0; j < num_nodes; ++j) {
    if (num_states[j] < 1 || parent_offsets[j + 1] < parent_offsets[j]) {
      delete s;
      return fail(BN_ERR_INVALID, "bad state count or parent offsets");
    }
    for (int32_t q = parent_offsets[j]; q < parent_offsets[j + 1]; ++q) {
      if (parent_indices[q] < 0 || parent_indices[q] >= num_nodes || parent_indices[q] == j) {
        delete s;
        return fail(BN_ERR_INVALID, "parent index out of range");
      }
      s->parents[j].push_back(parent_indices[q]);
    }
  }
  for (int32_t j = 0; j < num_nodes; ++j) {
    int64_t elems = s->states[j];
    for (int32_t q : s->parents[j]) elems *= s->states[q];
    s->cpt_off.push_back(s->cpt_total);
    s->cpt_total += elems;
  }
  if (hipGetDevice(&s->device) != hipSuccess) {
    delete s;
    return fail(BN_ERR_HIP, "hipGetDevice failed");
  }
  *out_sampler = s;
  return BN_OK;
}

int bn_sampler_destroy(bn_sampler* sampler) {
  if (!sampler) return BN_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(sampler->device);
  free_tables(sampler);
  (void)hipSetDevice(prev);
  delete sampler;
  return BN_OK;
}

int64_t bn_sampler_cpt_elements(const bn_sampler* sampler) { return sampler ? sampler->cpt_total : -1; }

int32_t bn_sampler_walk_order(int32_t num_nodes, const int32_t* parent_offsets, const int32_t* parent_indices,
                              const int32_t* nodes, int32_t num_requested, int32_t* order_out) {
  if (num_nodes < 1 || !parent_offsets || num_requested < 0 || (num_requested > 0 && !nodes) || !order_out)
    return fail(BN_ERR_INVALID, "bad argument");
  if (parent_offsets[0] != 0 || (parent_offsets[num_nodes] > 0 && !parent_indices))
    return fail(BN_ERR_INVALID, "bad parent offsets");
  std::vector<std::vector<int32_t>> parents(num_nodes);
  for (int32_t j = 0; j < num_nodes; ++j) {
    if (parent_offsets[j + 1] < parent_offsets[j]) return fail(BN_ERR_INVALID, "bad parent offsets");
    for (int32_t q = parent_offsets[j]; q < parent_offsets[j + 1]; ++q) {
      if (parent_indices[q] < 0 || parent_indices[q] >= num_nodes || parent_indices[q] == j)
        return fail(BN_ERR_INVALID, "parent index out of range");
      parents[j].push_back(parent_indices[q]);
    }
  }
  std::vector<int32_t> request(nodes, nodes + num_requested), position, walk;
  for (int32_t j : request)
    if (j < 0 || j >= num_nodes) return fail(BN_ERR_INVALID, "requested node out of range");
  if (int rc = walk_order(parents, request, position, walk)) return rc;
  std::copy(walk.begin(), walk.end(), order_out);
  return (int32_t)walk.size();
}

int64_t bn_sampler_workspace_bytes(bn_sampler* sampler, const int32_t* nodes, int32_t num_requested, int64_t num_samples) {
  if (!sampler || num_requested < 0 || num_samples < 0 || (num_requested > 0 && !nodes)) return -1;
  for (int32_t i = 0; i < num_requested; ++i)
    if (nodes[i] < 0 || nodes[i] >= sampler->n_nodes) return -1;
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess || device != sampler->device) return -1;
  if (build_tables(sampler, nodes, num_requested) != BN_OK) return -1;
  return std::max<int64_t>((int64_t)sampler->n_walk * num_samples * (int64_t)sizeof(int32_t), 4);
}

int bn_sampler_sample(bn_sampler* sampler, const void* cpts_dev, const int32_t* nodes, int32_t num_requested,
                      int64_t num_samples, uint64_t seed, void* workspace_dev, int32_t* out_dev, void* stream) {
  if (!sampler) return fail(BN_ERR_INVALID, "sampler is null");
  if (num_requested < 0 || num_samples < 0) return fail(BN_ERR_INVALID, "negative count");
  if (num_requested == 0 || num_samples == 0) return BN_OK;
  if (!cpts_dev || !nodes || !workspace_dev || !out_dev) return fail(BN_ERR_INVALID, "null argument");
  for (int32_t i = 0; i < num_requested; ++i)
    if (nodes[i] < 0 || nodes[i] >= sampler->n_nodes) return fail(BN_ERR_INVALID, "requested node out of range");
  int device = -1;
  if (hipGetDevice(&device) != hipSuccess || device != sampler->device)
    return fail(BN_ERR_STATE, "the sampler belongs to another HIP device than the current one");
  if (int rc = build_tables(sampler, nodes, num_requested)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int32_t* drawn = static_cast<int32_t*>(workspace_dev);
  const dim3 grid((uint32_t)((num_samples + 255) / 256));
  if (sampler->dtype == BN_F32) {
    hipLaunchKernelGGL(sample_walk<float>, grid, dim3(256), 0, s, static_cast<const float*>(cpts_dev), sampler->d_walk_states,
                       sampler->d_walk_cpt, sampler->d_walk_poff, sampler->d_walk_parents, sampler->n_walk, num_samples, seed,
                       drawn);
  } else {
    hipLaunchKernelGGL(sample_walk<double>, grid, dim3(256), 0, s, static_cast<const double*>(cpts_dev),
                       sampler->d_walk_states, sampler->d_walk_cpt, sampler->d_walk_poff, sampler->d_walk_parents,
                       sampler->n_walk, num_samples, seed, drawn);
  }
  BN_HIP_CHECK(hipGetLastError());
  const dim3 lgrid((uint32_t)((num_samples + 63) / 64), (uint32_t)((num_requested + 63) / 64));
  hipLaunchKernelGGL(sample_layout, lgrid, dim3(256), 0, s, drawn, sampler->d_out_pos, num_requested, num_samples, out_dev);
  BN_HIP_CHECK(hipGetLastError());
  return BN_OK;
}

}  // extern "C"
