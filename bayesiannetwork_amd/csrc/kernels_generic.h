// Shape-agnostic kernels: any fan-in, any cardinality up to kMaxStatesGeneric, any arity
// up to kMaxArity.  One lane = one observation; a 64-lane workgroup handles one node and
// one 64-observation slice of a tile, and keeps its per-lane vectors in LDS as
// [index][lane] so that runtime-indexed accesses never leave the chip and never
// bank-conflict (lane is the fastest axis).  The specialised kernels in kernels_fast.h
// replace these for the shapes that matter for throughput; results are interchangeable.
#pragma once

#include <hip/hip_runtime.h>

#include "plan.h"

namespace bn {

template <typename T>
__device__ __forceinline__ T dev_log(T x);
template <>
__device__ __forceinline__ float dev_log<float>(float x) {
  return logf(x);
}
template <>
__device__ __forceinline__ double dev_log<double>(double x) {
  return log(x);
}
template <typename T>
__device__ __forceinline__ T dev_exp(T x);
template <>
__device__ __forceinline__ float dev_exp<float>(float x) {
  return expf(x);
}
template <>
__device__ __forceinline__ double dev_exp<double>(double x) {
  return exp(x);
}

// element address inside a message array
__device__ __forceinline__ int64_t msg_index(int64_t base, int64_t tile_id, int32_t states, int32_t s,
                                             int32_t tile, int32_t lane_in_tile) {
  return base + (tile_id * states + s) * (int64_t)tile + lane_in_tile;
}

// ---------------------------------------------------------------------------------
// Variable pass.  For one node and 64 observations, with in_i[s] the messages waiting at
// the variable (VariableNodeGroup.calculate_outputs, spa_v3/factor_graph.py:112-144):
//
//   P[s]      = prod_i in_i[s]                                   (exp of the reference's x2)
//   c         = log sum_s P[s]                                   (local log-likelihood, :133-144)
//   to child factor i :  out_i[s] = (P[s]/in_i[s]) / sum_s' (P[s']/in_i[s'])     (:120,:127)
//   to local factor   :  out[s]   = (P[s]/in_local[s]) / sum_s' P[s']            (:138)
//
// The reference evaluates these through log / sum / softmax / exp to survive underflow.
// Here the product is carried as a double mantissa plus an integer exponent per state
// (renormalised every four float factors, every double factor), which is the same quantity without any
// transcendental and without the ~1e-7*|log| rounding that a float32 sum of several
// hundred logs picks up.  An exactly-zero input gives P = 0 and 0/0 = NaN on its own
// outgoing message and 0 on the others, as -inf - -inf does in the reference.
//
// grid.x = K * ntiles * (tile / 64), block = 64, dynamic LDS = S*64*24 bytes.
constexpr int32_t kDeadExponent = -(1 << 28);

// A running product kept as a double is brought back to [0.5, 1) before it can leave the double
// range: four float factors always fit (each >= 2^-149), a single double factor may already be
// close to the limit.
template <typename T>
__device__ __forceinline__ bool renorm_due(int32_t factors_done) {
  return sizeof(T) == 8 ? true : (factors_done & 3) == 0;
}

__device__ __forceinline__ int32_t clamp_shift(int32_t e) { return e < -4000 ? -4000 : (e > 4000 ? 4000 : e); }

template <typename T>
__global__ void __launch_bounds__(64) variable_pass_generic(T* __restrict__ arena,
                                                           const int64_t* __restrict__ in_off,
                                                           const int64_t* __restrict__ out_off,
                                                           const int64_t* __restrict__ c_off, int32_t K,
                                                           int32_t I, int32_t S, int32_t tile,
                                                           int64_t ntiles, int32_t fresh) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double* mant = reinterpret_cast<double*>(lds_raw);                       // [S][64]
  double* quo = mant + (size_t)S * 64;                                     // [S][64]
  int32_t* expo = reinterpret_cast<int32_t*>(quo + (size_t)S * 64);        // [S][64]
  int32_t* qexp = expo + (size_t)S * 64;                                   // [S][64]

  const int32_t lane = threadIdx.x;
  const int32_t subs = tile / 64;
  int64_t w = blockIdx.x;
  const int32_t sub = (int32_t)(w % subs);
  w /= subs;
  const int64_t t = w % ntiles;
  const int32_t k = (int32_t)(w / ntiles);
  if (k >= K) return;
  const int32_t lt = sub * 64 + lane;

  const int64_t* in_k = in_off + (int64_t)k * I;
  const int64_t* out_k = out_off + (int64_t)k * I;
  // while the plan is fresh, messages from factors still have their initial value 1/S and
  // are not read (the observation slot, whose output offset is -1, always is)
  const T uniform = (T)1 / (T)S;
  auto input = [&](int32_t i, int32_t s) -> T {
    if (fresh && out_k[i] != kNoArray) return uniform;
    return arena[msg_index(in_k[i], t, S, s, tile, lt)];
  };

  // P[s] as mantissa * 2^exponent
  int32_t emax = kDeadExponent;
  for (int32_t s = 0; s < S; ++s) {
    double m = 1.0;
    int32_t e = 0;
    for (int32_t i = 0; i < I; ++i) {
      m *= (double)input(i, s);
      if (renorm_due<T>(i + 1) || i == I - 1) {
        int de = 0;
        m = frexp(m, &de);
        e += de;
      }
    }
    if (m == 0.0) e = kDeadExponent;
    mant[s * 64 + lane] = m;
    expo[s * 64 + lane] = e;
    emax = e > emax ? e : emax;
  }
  double z = 0.0;
  for (int32_t s = 0; s < S; ++s) z += ldexp(mant[s * 64 + lane], clamp_shift(expo[s * 64 + lane] - emax));
  arena[msg_index(c_off[k], t, 1, 0, tile, lt)] = (T)((double)emax * 0.6931471805599453 + log(z));

  for (int32_t i = 0; i < I; ++i) {
    const int64_t dst = out_k[i];
    if (dst == kNoArray) continue;
    // leave-one-out quotient, again as mantissa * 2^exponent
    int32_t xmax = kDeadExponent;
    for (int32_t s = 0; s < S; ++s) {
      int de = 0;
      const double mm = frexp((double)input(i, s), &de);
      const double q = mant[s * 64 + lane] / mm;
      int32_t x = expo[s * 64 + lane] - de;
      if (q == 0.0) x = kDeadExponent;
      quo[s * 64 + lane] = q;
      qexp[s * 64 + lane] = x;
      xmax = x > xmax ? x : xmax;
    }
    if (i == I - 1) {
      for (int32_t s = 0; s < S; ++s)
        arena[msg_index(dst, t, S, s, tile, lt)] =
            (T)(ldexp(quo[s * 64 + lane], clamp_shift(qexp[s * 64 + lane] - emax)) / z);
    } else {
      double sum = 0.0;
      for (int32_t s = 0; s < S; ++s) {
        const double v = ldexp(quo[s * 64 + lane], clamp_shift(qexp[s * 64 + lane] - xmax));
        quo[s * 64 + lane] = v;
        sum += v;
      }
      for (int32_t s = 0; s < S; ++s) arena[msg_index(dst, t, S, s, tile, lt)] = (T)(quo[s * 64 + lane] / sum);
    }
  }
}

// ---------------------------------------------------------------------------------
// Factor pass: out_i[x_i] = sum over the other axes of cpt[x] * prod_{j != i} in_j[x_j]
// for every slot i at once (FactorNodeGroup.calculate_outputs,
// spa_v3/factor_graph.py:251-298, which issues one einsum per slot).  The CPT is walked
// once in row-major order with an odometer; each entry is a wave-uniform scalar.
//
// grid.x = K * ntiles * (tile/64), block = 64, dynamic LDS = sum(dims)*64*(sizeof(T)+8).
struct FactorDims {
  int32_t arity;
  int32_t d[kMaxArity];
  int32_t base[kMaxArity];  // prefix sums of d
  int32_t total;            // sum of d
  int64_t elems;            // product of d
};

template <typename T>
__global__ void __launch_bounds__(64) factor_pass_generic(T* __restrict__ arena, const T* __restrict__ cpts,
                                                         const int64_t* __restrict__ in_off,
                                                         const int64_t* __restrict__ out_off,
                                                         const int64_t* __restrict__ cpt_off, int32_t K,
                                                         FactorDims fd, int32_t tile, int64_t ntiles) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double* acc = reinterpret_cast<double*>(lds_raw);                       // [total][64]
  T* in = reinterpret_cast<T*>(lds_raw + (size_t)fd.total * 64 * 8);      // [total][64]

  const int32_t lane = threadIdx.x;
  const int32_t subs = tile / 64;
  int64_t w = blockIdx.x;
  const int32_t sub = (int32_t)(w % subs);
  w /= subs;
  const int64_t t = w % ntiles;
  const int32_t k = (int32_t)(w / ntiles);
  if (k >= K) return;
  const int32_t lt = sub * 64 + lane;
  const int32_t A = fd.arity;

#pragma unroll
  for (int32_t j = 0; j < kMaxArity; ++j) {
    if (j < A) {
      const int64_t src = in_off[(int64_t)k * A + j];
      for (int32_t x = 0; x < fd.d[j]; ++x) {
        in[(fd.base[j] + x) * 64 + lane] = arena[msg_index(src, t, fd.d[j], x, tile, lt)];
        acc[(fd.base[j] + x) * 64 + lane] = 0.0;
      }
    }
  }

  const T* cpt = cpts + cpt_off[k];
  int32_t idx[kMaxArity];
#pragma unroll
  for (int32_t j = 0; j < kMaxArity; ++j) idx[j] = 0;

  for (int64_t e = 0; e < fd.elems; ++e) {
    const double c = (double)cpt[e];
    double v[kMaxArity], pre[kMaxArity + 1], suf[kMaxArity + 1];
    pre[0] = 1.0;
#pragma unroll
    for (int32_t j = 0; j < kMaxArity; ++j) {
      v[j] = (j < A) ? (double)in[(fd.base[j] + idx[j]) * 64 + lane] : 1.0;
      pre[j + 1] = pre[j] * v[j];
    }
    suf[kMaxArity] = 1.0;
#pragma unroll
    for (int32_t j = kMaxArity - 1; j >= 0; --j) suf[j] = suf[j + 1] * v[j];
#pragma unroll
    for (int32_t j = 0; j < kMaxArity; ++j) {
      if (j < A) acc[(fd.base[j] + idx[j]) * 64 + lane] += c * pre[j] * suf[j + 1];
    }
    // advance the odometer, last axis fastest
    bool carry = true;
#pragma unroll
    for (int32_t j = kMaxArity - 1; j >= 0; --j) {
      if (j < A && carry) {
        if (++idx[j] == fd.d[j]) {
          idx[j] = 0;
        } else {
          carry = false;
        }
      }
    }
  }

#pragma unroll
  for (int32_t j = 0; j < kMaxArity; ++j) {
    if (j < A) {
      const int64_t dst = out_off[(int64_t)k * A + j];
      for (int32_t x = 0; x < fd.d[j]; ++x)
        arena[msg_index(dst, t, fd.d[j], x, tile, lt)] = (T)acc[(fd.base[j] + x) * 64 + lane];
    }
  }
}

// Is float evidence of two-state leaves hard evidence in disguise - does every leaf's every observation
// have one of two rows?  A row counts for the state whose entry is the larger one (state 1 when they are
// equal) and its state is written to states[leaf][n]; the rows of a state are all the same exactly when
// the smallest and the largest value seen of either entry coincide, which is what the verdicts hold, one
// per block of the launch (as bit patterns: the entries must not be negative, or `other` is set).  ev: the first leaf's array
// ([N][2], 16-byte aligned), the others leaf_stride floats apart (a multiple of four); N even.  A thread takes four
// observations at a time, a grid of any size the lot.
struct CompactVerdict {
  int32_t other;           // some entry is negative or not a number
  uint32_t lo[4], hi[4];   // of (state 0: entry 0, entry 1; state 1: entry 0, entry 1); lo > hi: no such row
  int32_t pad[7];
};
static __global__ void __launch_bounds__(256) evidence_two_rows(const float* __restrict__ ev, int64_t leaf_stride, int64_t N,
                                                        int64_t leaves, uint8_t* __restrict__ states, CompactVerdict* verdict) {
  const int64_t quads_per_leaf = (N + 3) / 4;
  const int64_t total = leaves * quads_per_leaf;
  uint32_t lo[4] = {0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u}, hi[4] = {0u, 0u, 0u, 0u};
  bool other = false;
  // (leaf, four observations) of the thread's items without a division per item: one here, additions from there
  const int64_t step = (int64_t)gridDim.x * 256;
  const int64_t step_leaves = step / quads_per_leaf, step_quads = step - step_leaves * quads_per_leaf;
  int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t leaf = item / quads_per_leaf, quad = item - leaf * quads_per_leaf;
  for (; item < total; item += step) {
    const int64_t n0 = quad * 4;
    const float* rows = ev + leaf * leaf_stride + n0 * 2;
    const bool whole = n0 + 4 <= N;  // (else two observations are left: N is even)
    const float4 first = *reinterpret_cast<const float4*>(rows);
    const float4 second = whole ? *reinterpret_cast<const float4*>(rows + 4) : first;
    const float x[4] = {first.x, first.z, second.x, second.z}, y[4] = {first.y, first.w, second.y, second.w};
    uint32_t packed = 0;
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      other |= !(x[v] >= 0.f && y[v] >= 0.f);
      const bool zero = x[v] > y[v];
      packed |= (zero ? 0u : 1u) << (8 * v);
      const uint32_t bx = __builtin_bit_cast(uint32_t, x[v]), by = __builtin_bit_cast(uint32_t, y[v]);
      // (a row of the other state leaves the extremes alone: it enters as the value they start from)
      lo[0] = min(lo[0], zero ? bx : 0x7f800000u);
      hi[0] = max(hi[0], zero ? bx : 0u);
      lo[1] = min(lo[1], zero ? by : 0x7f800000u);
      hi[1] = max(hi[1], zero ? by : 0u);
      lo[2] = min(lo[2], zero ? 0x7f800000u : bx);
      hi[2] = max(hi[2], zero ? 0u : bx);
      lo[3] = min(lo[3], zero ? 0x7f800000u : by);
      hi[3] = max(hi[3], zero ? 0u : by);
    }
    if (whole)
      *reinterpret_cast<uint32_t*>(states + leaf * N + n0) = packed;
    else
      *reinterpret_cast<uint16_t*>(states + leaf * N + n0) = (uint16_t)packed;
    quad += step_quads;
    leaf += step_leaves;
    if (quad >= quads_per_leaf) {
      quad -= quads_per_leaf;
      ++leaf;
    }
  }
  // the wave's extremes, then the block's: every block leaves a verdict of its own (atomics of a few thousand
  // waves on the same eight words took 1.5 ms)
  __shared__ uint32_t part[4][9];
#pragma unroll
  for (int32_t i = 0; i < 4; ++i) {
#pragma unroll
    for (int32_t o = 32; o > 0; o >>= 1) {
      const uint32_t l = __shfl_xor(lo[i], o, 64), h = __shfl_xor(hi[i], o, 64);
      lo[i] = l < lo[i] ? l : lo[i];
      hi[i] = h > hi[i] ? h : hi[i];
    }
  }
  other = __any((int)other) != 0;
  const int32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int32_t i = 0; i < 4; ++i) {
      part[wave][i] = lo[i];
      part[wave][4 + i] = hi[i];
    }
    part[wave][8] = other ? 1u : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    CompactVerdict mine{};
#pragma unroll
    for (int32_t i = 0; i < 4; ++i) {
      mine.lo[i] = 0x7f800000u;
      for (int32_t wv = 0; wv < 4; ++wv) {
        mine.lo[i] = part[wv][i] < mine.lo[i] ? part[wv][i] : mine.lo[i];
        mine.hi[i] = part[wv][4 + i] > mine.hi[i] ? part[wv][4 + i] : mine.hi[i];
      }
    }
    for (int32_t wv = 0; wv < 4; ++wv) mine.other |= (int32_t)part[wv][8];
    verdict[blockIdx.x] = mine;
  }
}


// ---------------------------------------------------------------------------------
// Evidence: ev[n, s] (row-major, caller's buffer) -> observation array in tile layout.
// Pure copy, bit-exact (VariableNodeGroup.set_observations, spa_v3/factor_graph.py:166-170).
// Padding observations get 1.0.   grid = (ceil(S*npad/256), num_observed), block = 256.
template <typename T>
__global__ void __launch_bounds__(256) evidence_scatter(T* __restrict__ arena,
                                                       const void* const* __restrict__ sources,
                                                       const int64_t* __restrict__ obs_off,
                                                       const int32_t* __restrict__ obs_states, int64_t N,
                                                       int32_t tile, int64_t ntiles) {
  const int32_t o = blockIdx.y;
  const int32_t S = obs_states[o];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // position inside the array
  if (idx >= (int64_t)S * ntiles * tile) return;
  const int32_t lt = (int32_t)(idx % tile);
  const int64_t rest = idx / tile;
  const int32_t s = (int32_t)(rest % S);
  const int64_t t = rest / S;
  const int64_t n = t * tile + lt;
  const T* src = static_cast<const T*>(sources[o]);
  arena[obs_off[o] + idx] = n < N ? src[n * S + s] : (T)1;
}

// The same copy with both sides coalesced: a block takes 64 observations of one observed node,
// reads their S * 64 consecutive source elements into LDS and writes them out row by row
// (the element-per-thread version above reads with a stride of S elements).
//   grid = (npad / 64, num_observed), block = 256, dynamic LDS = 64 * (max S + 1) * sizeof(T)
template <typename T>
__global__ void __launch_bounds__(256) evidence_scatter_lds(T* __restrict__ arena,
                                                           const void* const* __restrict__ sources,
                                                           const int64_t* __restrict__ obs_off,
                                                           const int32_t* __restrict__ obs_states, int64_t N,
                                                           int32_t tile) {
  extern __shared__ __align__(16) unsigned char scatter_lds[];
  T* stage = reinterpret_cast<T*>(scatter_lds);  // [64][S + 1]
  const int32_t o = blockIdx.y;
  const int32_t S = obs_states[o];
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const T* src = static_cast<const T*>(sources[o]);
  const int32_t count = 64 * S;
  for (int32_t idx = threadIdx.x; idx < count; idx += 256) {
    const int32_t l = idx / S, st = idx - l * S;
    stage[l * (S + 1) + st] = n0 + l < N ? src[n0 * S + idx] : (T)1;
  }
  __syncthreads();
  const int64_t t = n0 / tile;
  const int32_t lt0 = (int32_t)(n0 % tile);
  T* dst = arena + obs_off[o] + t * S * (int64_t)tile + lt0;
  for (int32_t idx = threadIdx.x; idx < count; idx += 256) {
    const int32_t st = idx >> 6, l = idx & 63;
    dst[(int64_t)st * tile + l] = stage[l * (S + 1) + st];
  }
}

// Writes a batch of addresses, handed over as kernel arguments, into a device table.  Kernel
// arguments are captured at launch, so the caller's table needs no staging buffer and the upload is
// ordered with the stream without any host-side wait.
constexpr int32_t kAddressBatch = 448;  // 3.5 KiB of the 4 KiB kernel-argument space
struct AddressBatch {
  const void* p[kAddressBatch];
};
static __global__ void __launch_bounds__(256) store_addresses(AddressBatch batch, int32_t count, const void** __restrict__ dst) {
  const int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (i < count) dst[i] = batch.p[i];
}

// Two-state evidence, two observations per lane: one [e0 e1 | e0 e1] group in (16 bytes of float,
// 32 of double), one pair out to each of the two state rows.  Sources that are not 16-byte aligned
// are read element-wise.   grid = (ceil(npad / 512), num_observed); tile must be even
template <typename T>
__global__ void __launch_bounds__(256) evidence_scatter_pairs(T* __restrict__ arena,
                                                             const void* const* __restrict__ sources,
                                                             const int64_t* __restrict__ obs_off, int64_t N,
                                                             int32_t tile, int64_t npad) {
  typedef T pair_t __attribute__((ext_vector_type(2)));
  const int32_t o = blockIdx.y;
  const int64_t n0 = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  if (n0 >= npad) return;
  const T* src = static_cast<const T*>(sources[o]);
  T x[4] = {(T)1, (T)1, (T)1, (T)1};
  if (n0 + 1 < N) {
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const pair_t a = *reinterpret_cast<const pair_t*>(src + 2 * n0);
      const pair_t b = *reinterpret_cast<const pair_t*>(src + 2 * n0 + 2);
      x[0] = a[0];
      x[1] = a[1];
      x[2] = b[0];
      x[3] = b[1];
    } else {
#pragma unroll
      for (int32_t i = 0; i < 4; ++i) x[i] = src[2 * n0 + i];
    }
  } else if (n0 < N) {
    x[0] = src[2 * n0];
    x[1] = src[2 * n0 + 1];
  }
  const int64_t t = n0 / tile;
  const int32_t lt = (int32_t)(n0 % tile);
  T* dst = arena + obs_off[o] + t * 2 * tile + lt;
  pair_t row0, row1;
  row0[0] = x[0];
  row0[1] = x[2];
  row1[0] = x[1];
  row1[1] = x[3];
  *reinterpret_cast<pair_t*>(dst) = row0;
  *reinterpret_cast<pair_t*>(dst + tile) = row1;
}

// Hard evidence from state indices: row s of the observation slot gets `on` where the observed
// state is s and `off` elsewhere; four observations per thread (one 4-byte read of the indices,
// 16-byte stores for float).  grid = (ceil(npad / 1024), num_observed); tile % 4 == 0
template <typename T>
__global__ void __launch_bounds__(256) evidence_from_states(T* __restrict__ arena,
                                                           const uint8_t* __restrict__ states_in,
                                                           const int64_t* __restrict__ obs_off,
                                                           const int32_t* __restrict__ obs_states, int64_t N,
                                                           int32_t tile, int64_t npad, T on, T off) {
  const int32_t o = blockIdx.y;
  const int64_t n0 = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
  if (n0 >= npad) return;
  const int32_t S = obs_states[o];
  const uint8_t* src = states_in + (int64_t)o * N;
  int32_t st[4];
  if (n0 + 3 < N && ((reinterpret_cast<uintptr_t>(src + n0) & 3) == 0)) {
    const uint32_t packed = *reinterpret_cast<const uint32_t*>(src + n0);
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) st[v] = (int32_t)((packed >> (8 * v)) & 0xff);
  } else {
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) st[v] = n0 + v < N ? (int32_t)src[n0 + v] : -1;  // padding: handled below
  }
  const int64_t t = n0 / tile;
  const int32_t lt = (int32_t)(n0 % tile);
  T* dst = arena + obs_off[o] + t * S * tile + lt;
  for (int32_t s = 0; s < S; ++s) {
    T row[4];
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) row[v] = n0 + v < N ? (st[v] == s ? on : off) : (T)1;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(dst + (int64_t)s * tile) = make_float4(row[0], row[1], row[2], row[3]);
    } else {
      *reinterpret_cast<double2*>(dst + (int64_t)s * tile) = make_double2(row[0], row[1]);
      *reinterpret_cast<double2*>(dst + (int64_t)s * tile + 2) = make_double2(row[2], row[3]);
    }
  }
}

// fill one region with a constant.  grid.x = ceil(count/256)
template <typename T>
__global__ void __launch_bounds__(256) fill_region(T* __restrict__ dst, int64_t count, T value) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < count) dst[idx] = value;
}

// fill many message arrays with 1/states.  grid = (ceil(max_elems/256), num_arrays)
template <typename T>
__global__ void __launch_bounds__(256) fill_uniform(T* __restrict__ arena, const int64_t* __restrict__ offs,
                                                   const int32_t* __restrict__ states, int64_t npad) {
  const int32_t a = blockIdx.y;
  const int64_t count = (int64_t)states[a] * npad;
  const T value = (T)1 / (T)states[a];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (int64_t)gridDim.x * 256)
    arena[offs[a] + idx] = value;
}

// ---------------------------------------------------------------------------------
// Tile layout -> row-major [N, S] product of up to two messages (single-node marginal,
// SpaInferenceMachine._infer_single_node, spa_v3/spa_inference_machine.py:72-81; with
// b_off == kNoArray it is a plain message read-out).   grid.x = ceil(N*S/256)
template <typename T>
__global__ void __launch_bounds__(256) gather_rows(const T* __restrict__ arena, int64_t a_off, int64_t b_off,
                                                  T* __restrict__ out, int64_t N, int32_t S, int32_t tile) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * S) return;
  const int64_t n = idx / S;
  const int32_t s = (int32_t)(idx % S);
  const int64_t t = n / tile;
  const int32_t lt = (int32_t)(n % tile);
  T v = arena[msg_index(a_off, t, S, s, tile, lt)];
  if (b_off != kNoArray) v *= arena[msg_index(b_off, t, S, s, tile, lt)];
  out[idx] = v;
}

// ---------------------------------------------------------------------------------
// Family marginal, materialised: out[n, x] = cpt[x] * prod_j in_j[n, x_j]
// (SpaInferenceMachine._infer_node_with_parents, spa_v3/spa_inference_machine.py:89-104).
// grid.x = ceil(N*elems/256)
struct FamilyArgs {
  FactorDims fd;
  int64_t in_off[kMaxArity];
  int64_t cpt_off;
};

template <typename T>
__global__ void __launch_bounds__(256) family_marginal(const T* __restrict__ arena, const T* __restrict__ cpts,
                                                      FamilyArgs fa, T* __restrict__ out, int64_t N,
                                                      int32_t tile) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * fa.fd.elems) return;
  const int64_t n = idx / fa.fd.elems;
  int64_t e = idx % fa.fd.elems;
  const int64_t t = n / tile;
  const int32_t lt = (int32_t)(n % tile);
  double prod = (double)cpts[fa.cpt_off + e];
#pragma unroll
  for (int32_t j = kMaxArity - 1; j >= 0; --j) {
    if (j < fa.fd.arity) {
      const int32_t x = (int32_t)(e % fa.fd.d[j]);
      e /= fa.fd.d[j];
      prod *= (double)arena[msg_index(fa.in_off[j], t, fa.fd.d[j], x, tile, lt)];
    }
  }
  out[idx] = (T)prod;
}

// ---------------------------------------------------------------------------------
// Sum of the normalisers c[node, n] over all nodes and the real observations.
// Stage 1: grid.x blocks of 256 write one double each; stage 2 (one block) adds them in
// index order, so the result does not depend on scheduling.
template <typename T>
__global__ void __launch_bounds__(256) ll_partial(const T* __restrict__ c, int64_t n_nodes, int64_t N,
                                                 int64_t npad, double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  const int64_t total = n_nodes * npad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t n = idx % npad;
    if (n < N) acc += (double)c[idx];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// M-step, one thread per CPT row.  Products and sums are kept separate (no fused multiply-add),
// so the result is the one the same formulas give when evaluated operation by operation.
// ll_out != nullptr: block 0 also adds up `ll_count` partial sums of the log-likelihood exactly as
// sum_partials does (same association) and writes ll_scale times the total there - the last launch
// of an EM step then delivers both results.
template <typename T>
__global__ void __launch_bounds__(256) m_step_rows(const double* __restrict__ counts, const T* old_cpts,
                                                  const int64_t* __restrict__ row_off,
                                                  const int32_t* __restrict__ row_len, int64_t rows,
                                                  double learning_rate, double regularization, T* out, T* out2,
                                                  const double* __restrict__ ll_partial, int32_t ll_count,
                                                  double ll_scale, double* __restrict__ ll_out) {
#pragma clang fp contract(off)
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < rows) {
    const int64_t off = row_off[r];
    const int32_t S = row_len[r];
    double sum = 0.0;
    for (int32_t j = 0; j < S; ++j) sum += counts[off + j];
    const double uniform = 1.0 / (double)S;
    for (int32_t j = 0; j < S; ++j) {
      double x = counts[off + j] / sum;
      if (regularization != 0.0) {
        const double a = (1.0 - regularization) * x;
        const double b = regularization * uniform;
        x = a + b;
      }
      if (learning_rate >= 0.0) {
        const double a = (1.0 - learning_rate) * (double)old_cpts[off + j];
        const double b = learning_rate * x;
        x = a + b;
      }
      out[off + j] = (T)x;  // may be old_cpts itself: each entry is read before it is written, by the same thread
      if (out2) out2[off + j] = (T)x;
    }
  }
  if (ll_out == nullptr || blockIdx.x != 0) return;
  __shared__ double red[256];
  double acc = 0.0;
  for (int32_t i = threadIdx.x; i < ll_count; i += 256) acc += ll_partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) ll_out[0] = red[0] * ll_scale;
}

// Same sum when part of it already exists as per-observation partial sums: adds the normalisers
// of the listed arrays (offsets into the arena, npad elements each) and `extra_rows` rows of
// npad doubles.  grid.x = number of partials written
template <typename T>
__global__ void __launch_bounds__(256) ll_partial_listed(const T* __restrict__ arena,
                                                        const int64_t* __restrict__ c_offs, int64_t n_listed,
                                                        const double* __restrict__ extra, int64_t extra_rows,
                                                        int64_t N, int64_t npad, double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  const int64_t per_row = ((npad + 255) / 256) * 256;  // a row never shares a thread stride with the next
  const int64_t total = (n_listed + extra_rows) * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / per_row;
    const int64_t n = idx - row * per_row;
    if (n >= N) continue;
    acc += row < n_listed ? (double)arena[c_offs[row] + n] : extra[(row - n_listed) * npad + n];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// max |c - before| over the real observations of all nodes, and before <- c.  One partial per block.
template <typename T>
__global__ void __launch_bounds__(256) normaliser_change_partial(const T* __restrict__ c, T* __restrict__ before,
                                                                int32_t before_is_zero, int64_t n_nodes, int64_t N,
                                                                int64_t npad, double* __restrict__ partial) {
  __shared__ double red[256];
  double worst = 0.0;
  const int64_t total = n_nodes * npad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const T now = c[idx];
    if (idx % npad < N) {
      const double d = fabs((double)now - (before_is_zero ? 0.0 : (double)before[idx]));
      worst = d > worst || d != d ? d : worst;  // a NaN is reported, not hidden
    }
    before[idx] = now;
  }
  red[threadIdx.x] = worst;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) {
      const double o = red[threadIdx.x + step];
      if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

static __global__ void __launch_bounds__(256) max_partials(const double* __restrict__ partial, int32_t count,
                                                   double* __restrict__ out) {
  __shared__ double red[256];
  double worst = 0.0;
  for (int32_t i = threadIdx.x; i < count; i += 256) {
    const double o = partial[i];
    if (o > worst || o != o) worst = o;
  }
  red[threadIdx.x] = worst;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) {
      const double o = red[threadIdx.x + step];
      if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

static __global__ void __launch_bounds__(256) sum_partials(const double* __restrict__ partial, int32_t count,
                                                   double scale, double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int32_t i = threadIdx.x; i < count; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ---------------------------------------------------------------------------------
// EM expected counts, shape-agnostic: a thread owns one CPT entry x and sums
// prod_j in_j[n, x_j] over the observations of the block's tile range; inputs for 64
// observations at a time are staged in LDS as [row][65].  Output: one double per
// (entry, tile chunk); em_finalize adds the chunks in order and applies the CPT factor.
//
// grid = (ceil(elems/256), nchunks, K), block = 256, dynamic LDS = total*65*sizeof(T).
template <typename T>
__global__ void __launch_bounds__(256) em_partial_generic(const T* __restrict__ arena,
                                                         const int64_t* __restrict__ in_off, FactorDims fd,
                                                         int64_t N, int32_t tile, int64_t ntiles,
                                                         int32_t nchunks, double* __restrict__ partial,
                                                         const int64_t* __restrict__ cpt_off) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* in = reinterpret_cast<T*>(lds_raw);  // [total][65]
  const int32_t k = blockIdx.z;
  const int32_t chunk = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t A = fd.arity;

  int32_t row[kMaxArity];
  {
    int64_t r = e < fd.elems ? e : 0;
#pragma unroll
    for (int32_t j = kMaxArity - 1; j >= 0; --j) {
      if (j < A) {
        row[j] = fd.base[j] + (int32_t)(r % fd.d[j]);
        r /= fd.d[j];
      } else {
        row[j] = 0;
      }
    }
  }

  const int32_t subs = tile / 64;
  const int64_t slices = ntiles * subs;
  const int64_t per = (slices + nchunks - 1) / nchunks;
  const int64_t first = chunk * per;
  const int64_t last = first + per < slices ? first + per : slices;

  double acc = 0.0;
  for (int64_t sl = first; sl < last; ++sl) {
    const int64_t t = sl / subs;
    const int32_t sub = (int32_t)(sl % subs);
    const int64_t n0 = t * tile + sub * 64;
    int32_t cnt = (int32_t)(N - n0 < 64 ? N - n0 : 64);
    if (cnt <= 0) break;
    __syncthreads();
    for (int32_t r = threadIdx.x; r < fd.total * 64; r += 256) {
      const int32_t rr = r / 64, ll = r % 64;
      // which slot does row rr belong to
      int32_t j = 0;
#pragma unroll
      for (int32_t q = 1; q < kMaxArity; ++q)
        if (q < A && rr >= fd.base[q]) j = q;
      in[rr * 65 + ll] =
          arena[msg_index(in_off[(int64_t)k * A + j], t, fd.d[j], rr - fd.base[j], tile, sub * 64 + ll)];
    }
    __syncthreads();
    if (e < fd.elems) {
      for (int32_t n = 0; n < cnt; ++n) {
        double prod = 1.0;
#pragma unroll
        for (int32_t j = 0; j < kMaxArity; ++j)
          if (j < A) prod *= (double)in[row[j] * 65 + n];
        acc += prod;
      }
    }
  }
  if (e < fd.elems) partial[(cpt_off[k] + e) * nchunks + chunk] = acc;
}

// accum[x] = cpt[x] * sum over chunks.   grid.x = ceil(total_entries/256)
template <typename T>
__global__ void __launch_bounds__(256) em_finalize(const double* __restrict__ partial, const T* __restrict__ cpts,
                                                  int64_t entries, int32_t nchunks, double* __restrict__ accum) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= entries) return;
  double acc = 0.0;
  for (int32_t c = 0; c < nchunks; ++c) acc += partial[e * nchunks + c];
  accum[e] = acc * (double)cpts[e];
}

// The same for a listed subset of the entries (those no fused kernel produces itself), one wave
// per entry: the chunk partials of an entry are contiguous, and there are few entries but many
// chunks when only a handful of nodes go through the partial sums.
//   grid.x = ceil(count / 4), block = 256
template <typename T>
__global__ void __launch_bounds__(256) em_finalize_listed(const double* __restrict__ partial, const T* __restrict__ cpts,
                                                         const int32_t* __restrict__ entries, int32_t count,
                                                         int32_t nchunks, double* __restrict__ accum) {
  const int32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= count) return;
  const int64_t e = entries[i];
  // fixed association: lane l sums chunks l, l + 64, ..; then the butterfly
  double acc = 0.0;
  for (int32_t c = lane; c < nchunks; c += 64) acc += partial[e * nchunks + c];
#pragma unroll
  for (int32_t off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) accum[e] = acc * (double)cpts[e];
}

}  // namespace bn
