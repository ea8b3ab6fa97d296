// Shape-specialised kernels for the shapes that carry the throughput.
//
// Common structure: one wavefront owns one (node, tile) work item and moves every message row
// with one 16-byte load/store per lane (1 KiB per wave instruction, fully coalesced): a tile is 256
// float32 observations, four per lane, or 128 float64 observations, two per lane.  Cardinalities
// and fan-in are template parameters, so all per-lane vectors live in registers and every loop
// unrolls; CPT entries are wave-uniform and arrive through scalar loads.  All of these kernels
// stream each input once and each output once: they are HBM-bandwidth bound by construction.
// The register-resident variable pass, the small factor kernels and their EM count kernels are
// templates over the element type; the many-state kernels (kernels_dense.h) and the standalone
// wide-node kernels (kernels_wide.h) exist for float32 only.
//
// Each launch_* helper returns 1 if it handled the group, 0 if the caller must use the
// generic kernel, and a negative value on a launch failure.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_generic.h"
#include "plan.h"

namespace bn {

constexpr int32_t kTile = 256;       // float32 tile: 64 lanes x 4 observations
// observations per tile for a plan of element type T (a message row is 1 KiB either way)
template <typename T>
struct tile_of {
  static constexpr int32_t value = sizeof(T) == 4 ? 256 : 128;
};
constexpr int32_t kWavesPerBlock = 4;
constexpr int32_t kWideMaxStates = 16;

struct f4 {
  float v[4];
};

__device__ __forceinline__ f4 load4(const float* __restrict__ base, int64_t row, int32_t lane) {
  const float4 x = *reinterpret_cast<const float4*>(base + row * kTile + lane * 4);
  f4 r;
  r.v[0] = x.x;
  r.v[1] = x.y;
  r.v[2] = x.z;
  r.v[3] = x.w;
  return r;
}

__device__ __forceinline__ void store4(float* __restrict__ base, int64_t row, int32_t lane, const f4& r) {
  *reinterpret_cast<float4*>(base + row * kTile + lane * 4) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
}

// One message row of a tile as seen by a lane, for either element type: W = 4 floats or 2
// doubles, 16 bytes per lane either way.
template <typename T>
struct rowv {
  static constexpr int32_t W = tile_of<T>::value / 64;
  T v[W];
};

template <typename T>
__device__ __forceinline__ rowv<T> loadw(const T* __restrict__ base, int64_t row, int32_t lane) {
  typedef T vec_t __attribute__((ext_vector_type(rowv<T>::W)));
  const vec_t x = *reinterpret_cast<const vec_t*>(base + row * tile_of<T>::value + lane * rowv<T>::W);
  rowv<T> r;
#pragma unroll
  for (int32_t v = 0; v < rowv<T>::W; ++v) r.v[v] = x[v];
  return r;
}

template <typename T>
__device__ __forceinline__ void storew(T* __restrict__ base, int64_t row, int32_t lane, const rowv<T>& r) {
  typedef T vec_t __attribute__((ext_vector_type(rowv<T>::W)));
  vec_t x;
#pragma unroll
  for (int32_t v = 0; v < rowv<T>::W; ++v) x[v] = r.v[v];
  *reinterpret_cast<vec_t*>(base + row * tile_of<T>::value + lane * rowv<T>::W) = x;
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }

// work item of this wave, or false if past the end.  items = K * ntiles, node-major.
__device__ __forceinline__ bool wave_item(int64_t items, int64_t ntiles, int32_t& k, int64_t& t, int32_t& lane) {
  // readfirstlane tells the compiler the wave index is uniform: everything derived from it (node,
  // tile, table offsets, CPT entries) then lives in scalar registers and scalar loads
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= items) return false;
  k = (int32_t)(w / ntiles);
  t = w % ntiles;
  return true;
}

// ---------------------------------------------------------------------------------
// Factor pass, no parent: the message to the variable is the prior itself
// (FactorNodeGroup._construct_einsum_equation_for_output with no inputs,
// spa_v3/factor_graph.py:276-284).
template <typename T>
__global__ void __launch_bounds__(256) factor_root_fast(T* __restrict__ arena, const T* __restrict__ cpts,
                                                       const int64_t* __restrict__ out_off,
                                                       const int64_t* __restrict__ cpt_off, int32_t K, int32_t S,
                                                       int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  T* dst = arena + out_off[k];
  const T* cpt = cpts + cpt_off[k];
  for (int32_t s = 0; s < S; ++s) {
    const T c = cpt[s];
    rowv<T> r;
#pragma unroll
    for (int32_t v = 0; v < rowv<T>::W; ++v) r.v[v] = c;
    storew<T>(dst, t * S + s, lane, r);
  }
}

// ---------------------------------------------------------------------------------
// Factor pass, one parent with D0 states, own variable with S states:
//   to parent   out[a] = sum_b cpt[a][b] * in_own[b]
//   to own var  out[b] = sum_a cpt[a][b] * in_parent[a]
// (the two einsums of FactorNodeGroup.calculate_outputs, spa_v3/factor_graph.py:251-258).
template <typename T, int D0, int S>
__global__ void __launch_bounds__(256) factor_p1_fast(T* __restrict__ arena, const T* __restrict__ cpts,
                                                     const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ out_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K,
                                                     int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const T* in_par = arena + in_off[2 * k];
  const T* in_own = arena + in_off[2 * k + 1];
  T* out_par = arena + out_off[2 * k];
  T* out_own = arena + out_off[2 * k + 1];
  const T* cpt = cpts + cpt_off[k];

  rowv<T> a[D0], b[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) a[i] = loadw<T>(in_par, t * D0 + i, lane);
#pragma unroll
  for (int32_t j = 0; j < S; ++j) b[j] = loadw<T>(in_own, t * S + j, lane);

  rowv<T> oa[D0], ob[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t v = 0; v < rowv<T>::W; ++v) oa[i].v[v] = T(0);
#pragma unroll
  for (int32_t j = 0; j < S; ++j)
#pragma unroll
    for (int32_t v = 0; v < rowv<T>::W; ++v) ob[j].v[v] = T(0);

#pragma unroll
  for (int32_t i = 0; i < D0; ++i) {
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      const T c = cpt[i * S + j];
#pragma unroll
      for (int32_t v = 0; v < rowv<T>::W; ++v) {
        oa[i].v[v] = fma_t(c, b[j].v[v], oa[i].v[v]);
        ob[j].v[v] = fma_t(c, a[i].v[v], ob[j].v[v]);
      }
    }
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) storew<T>(out_par, t * D0 + i, lane, oa[i]);
#pragma unroll
  for (int32_t j = 0; j < S; ++j) storew<T>(out_own, t * S + j, lane, ob[j]);
}

// ---------------------------------------------------------------------------------
// Factor pass, two parents (D0, D1) and own variable (S): three contractions of one
// [D0, D1, S] CPT, each leaving one axis free.
template <typename T, int D0, int D1, int S>
__global__ void __launch_bounds__(256) factor_p2_fast(T* __restrict__ arena, const T* __restrict__ cpts,
                                                     const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ out_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K,
                                                     int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const T* in0 = arena + in_off[3 * k];
  const T* in1 = arena + in_off[3 * k + 1];
  const T* in2 = arena + in_off[3 * k + 2];
  const T* cpt = cpts + cpt_off[k];

  rowv<T> a[D0], b[D1], c[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) a[i] = loadw<T>(in0, t * D0 + i, lane);
#pragma unroll
  for (int32_t i = 0; i < D1; ++i) b[i] = loadw<T>(in1, t * D1 + i, lane);
#pragma unroll
  for (int32_t i = 0; i < S; ++i) c[i] = loadw<T>(in2, t * S + i, lane);

  rowv<T> oa[D0], ob[D1], oc[S];
#pragma unroll
  for (int32_t v = 0; v < rowv<T>::W; ++v) {
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) oa[i].v[v] = T(0);
#pragma unroll
    for (int32_t i = 0; i < D1; ++i) ob[i].v[v] = T(0);
#pragma unroll
    for (int32_t i = 0; i < S; ++i) oc[i].v[v] = T(0);
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < D1; ++j)
#pragma unroll
      for (int32_t l = 0; l < S; ++l) {
        const T w = cpt[(i * D1 + j) * S + l];
#pragma unroll
        for (int32_t v = 0; v < rowv<T>::W; ++v) {
          oa[i].v[v] = fma_t(w, b[j].v[v] * c[l].v[v], oa[i].v[v]);
          ob[j].v[v] = fma_t(w, a[i].v[v] * c[l].v[v], ob[j].v[v]);
          oc[l].v[v] = fma_t(w, a[i].v[v] * b[j].v[v], oc[l].v[v]);
        }
      }
  T* o0 = arena + out_off[3 * k];
  T* o1 = arena + out_off[3 * k + 1];
  T* o2 = arena + out_off[3 * k + 2];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) storew<T>(o0, t * D0 + i, lane, oa[i]);
#pragma unroll
  for (int32_t i = 0; i < D1; ++i) storew<T>(o1, t * D1 + i, lane, ob[i]);
#pragma unroll
  for (int32_t i = 0; i < S; ++i) storew<T>(o2, t * S + i, lane, oc[i]);
}

// ---------------------------------------------------------------------------------
// Variable pass for a small fan-in I (<= 4) and S states, all in registers.
// P[s] = prod_i in_i[s] fits a double without rescaling (at most four float factors),
// leave-one-out products are formed by multiplication, not division.
// Same quantities as variable_pass_generic.
template <typename T, int I, int S>
__global__ void __launch_bounds__(256) variable_small_fast(T* __restrict__ arena,
                                                          const int64_t* __restrict__ in_off,
                                                          const int64_t* __restrict__ out_off,
                                                          const int64_t* __restrict__ c_off, int32_t K,
                                                          int64_t ntiles, int32_t fresh) {
  constexpr int32_t W = rowv<T>::W;
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const T uniform = T(1) / (T)S;

  rowv<T> in[I][S];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const bool skip = fresh && out_off[(int64_t)k * I + i] != kNoArray;
    const T* src = arena + in_off[(int64_t)k * I + i];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      if (skip) {
#pragma unroll
        for (int32_t v = 0; v < W; ++v) in[i][s].v[v] = uniform;
      } else {
        in[i][s] = loadw<T>(src, t * S + s, lane);
      }
    }
  }

  // leave-one-out products loo[i][s] = prod_{j != i} in_j[s], and the full product
  double loo[I][S][W];
  double z[W];
#pragma unroll
  for (int32_t v = 0; v < W; ++v) z[v] = 0.0;

  if constexpr (std::is_same<T, float>::value) {
    // at most four float factors: the product fits a double as it is
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
#pragma unroll
      for (int32_t v = 0; v < W; ++v) {
        double pre = 1.0;
#pragma unroll
        for (int32_t i = 0; i < I; ++i) {
          loo[i][s][v] = pre;
          pre *= (double)in[i][s].v[v];
        }
        z[v] += pre;
        double suf = 1.0;
#pragma unroll
        for (int32_t i = I - 1; i >= 0; --i) {
          loo[i][s][v] *= suf;
          suf *= (double)in[i][s].v[v];
        }
      }
    }
    // normaliser c = log sum_s P[s]
    {
      rowv<T> c;
#pragma unroll
      for (int32_t v = 0; v < W; ++v) {
        int e = 0;
        const double m = frexp(z[v], &e);
        c.v[v] = (T)((double)e * 0.6931471805599453 + (double)logf((float)m));
      }
      storew<T>(arena + c_off[k], t, lane, c);
    }
  } else {
    // double factors can take a plain product out of range (the reference works on logarithms
    // here, factor_graph.py:114-138): mantissas and exponents are multiplied apart, and every
    // sum over the states is taken relative to its largest exponent
    int32_t ex[I][S][W];  // exponent that goes with loo[i][s]
    int32_t emax[W];
    double full[S][W];
    int32_t fex[S][W];
#pragma unroll
    for (int32_t v = 0; v < W; ++v) emax[v] = kDeadExponent;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
#pragma unroll
      for (int32_t v = 0; v < W; ++v) {
        double m[I];
        int32_t e[I];
#pragma unroll
        for (int32_t i = 0; i < I; ++i) {
          int de = 0;
          m[i] = frexp((double)in[i][s].v[v], &de);
          e[i] = m[i] == 0.0 ? kDeadExponent : de;
        }
        double pre = 1.0;
        int32_t pre_e = 0;
#pragma unroll
        for (int32_t i = 0; i < I; ++i) {
          loo[i][s][v] = pre;
          ex[i][s][v] = pre_e;
          pre *= m[i];
          pre_e += e[i];
        }
        full[s][v] = pre;
        fex[s][v] = pre_e;
        emax[v] = pre_e > emax[v] ? pre_e : emax[v];
        double suf = 1.0;
        int32_t suf_e = 0;
#pragma unroll
        for (int32_t i = I - 1; i >= 0; --i) {
          loo[i][s][v] *= suf;
          ex[i][s][v] += suf_e;
          suf *= m[i];
          suf_e += e[i];
        }
      }
    }
    rowv<T> c;
#pragma unroll
    for (int32_t v = 0; v < W; ++v) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s) z[v] += ldexp(full[s][v], clamp_shift(fex[s][v] - emax[v]));
      c.v[v] = (T)((double)emax[v] * 0.6931471805599453 + log(z[v]));
    }
    storew<T>(arena + c_off[k], t, lane, c);
    // bring every leave-one-out product to a common scale: the local slot shares the full
    // product's (its message is loo / sum P), the others their own largest exponent
#pragma unroll
    for (int32_t i = 0; i < I; ++i) {
#pragma unroll
      for (int32_t v = 0; v < W; ++v) {
        int32_t top = emax[v];
        if (i != I - 1) {
          top = kDeadExponent;
#pragma unroll
          for (int32_t s = 0; s < S; ++s) top = ex[i][s][v] > top ? ex[i][s][v] : top;
        }
#pragma unroll
        for (int32_t s = 0; s < S; ++s) loo[i][s][v] = ldexp(loo[i][s][v], clamp_shift(ex[i][s][v] - top));
      }
    }
  }

#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const int64_t dst = out_off[(int64_t)k * I + i];
    if (dst == kNoArray) continue;
    double inv[W];
    if (i == I - 1) {
#pragma unroll
      for (int32_t v = 0; v < W; ++v) inv[v] = 1.0 / z[v];
    } else {
#pragma unroll
      for (int32_t v = 0; v < W; ++v) {
        double sum = 0.0;
#pragma unroll
        for (int32_t s = 0; s < S; ++s) sum += loo[i][s][v];
        inv[v] = 1.0 / sum;
      }
    }
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      rowv<T> o;
#pragma unroll
      for (int32_t v = 0; v < W; ++v) o.v[v] = (T)(loo[i][s][v] * inv[v]);
      storew<T>(arena + dst, t * S + s, lane, o);
    }
  }
}

// ---------------------------------------------------------------------------------
// EM expected counts, one parent (D0) + own variable (S), or no parent (D0 = 1, the
// parent message is the constant 1): per lane running sums of in_parent[a] * in_own[b]
// over the observations of the wave's tile range (double running sums), then a wave reduction.  The CPT factor
// and the sum over chunks are applied by em_finalize.
template <typename T, int D0, int S, bool HAS_PARENT>
__global__ void __launch_bounds__(256) em_p1_fast(const T* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                 const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                 int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)K * nchunks) return;
  const int32_t k = (int32_t)(w / nchunks);
  const int32_t chunk = (int32_t)(w % nchunks);
  constexpr int32_t A = HAS_PARENT ? 2 : 1;
  const T* in_par = HAS_PARENT ? arena + in_off[(int64_t)k * A] : nullptr;
  const T* in_own = arena + in_off[(int64_t)k * A + (A - 1)];

  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  double acc[D0][S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) acc[i][j] = 0.0;

  for (int64_t t = t0; t < t1; ++t) {
    rowv<T> a[D0], b[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      if (HAS_PARENT) {
        a[i] = loadw<T>(in_par, t * D0 + i, lane);
      } else {
        {
#pragma unroll
          for (int32_t v = 0; v < rowv<T>::W; ++v) a[i].v[v] = T(1);
        }
      }
    }
#pragma unroll
    for (int32_t j = 0; j < S; ++j) b[j] = loadw<T>(in_own, t * S + j, lane);
    const int64_t n0 = t * tile_of<T>::value + lane * rowv<T>::W;
#pragma unroll
    for (int32_t v = 0; v < rowv<T>::W; ++v) {
      const double live = (n0 + v < N) ? 1.0 : 0.0;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        const double av = (double)a[i].v[v] * live;
#pragma unroll
        for (int32_t j = 0; j < S; ++j) acc[i][j] = fma(av, (double)b[j].v[v], acc[i][j]);
      }
    }
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      double x = acc[i][j];
#pragma unroll
      for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0) partial[(cpt_off[k] + i * S + j) * nchunks + chunk] = x;
    }
}

// The count sums of parentless nodes with any number of states: em_p1_fast<T, 1, S, false> with S at
// run time.  The same per-lane running sums in the same order (so the two agree bit for bit), four
// states at a time (their loads are in flight together; the node's rows are read once in all).
template <typename T>
__global__ void __launch_bounds__(256) em_root_any(const T* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                  const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                  int64_t ntiles, int32_t nchunks, double* __restrict__ partial, int32_t S) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)K * nchunks) return;
  const int32_t k = (int32_t)(w / nchunks);
  const int32_t chunk = (int32_t)(w % nchunks);
  const T* in_own = arena + in_off[k];
  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
  for (int32_t j0 = 0; j0 < S; j0 += 4) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t t = t0; t < t1; ++t) {
      rowv<T> row[4];
#pragma unroll
      for (int32_t u = 0; u < 4; ++u) row[u] = loadw<T>(in_own, t * S + (j0 + u < S ? j0 + u : j0), lane);
      const int64_t n0 = t * tile_of<T>::value + lane * rowv<T>::W;
#pragma unroll
      for (int32_t u = 0; u < 4; ++u)
#pragma unroll
        for (int32_t v = 0; v < rowv<T>::W; ++v) {
          const double live = (n0 + v < N) ? 1.0 : 0.0;
          acc[u] = fma(1.0 * live, (double)row[u].v[v], acc[u]);
        }
    }
#pragma unroll
    for (int32_t u = 0; u < 4; ++u) {
      double x = acc[u];
#pragma unroll
      for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0 && j0 + u < S) partial[(cpt_off[k] + j0 + u) * nchunks + chunk] = x;
    }
  }
}

template <typename T, int D0, int D1, int S>
__global__ void __launch_bounds__(256) em_p2_fast(const T* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                 const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                 int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)K * nchunks) return;
  const int32_t k = (int32_t)(w / nchunks);
  const int32_t chunk = (int32_t)(w % nchunks);
  const T* in0 = arena + in_off[(int64_t)k * 3];
  const T* in1 = arena + in_off[(int64_t)k * 3 + 1];
  const T* in2 = arena + in_off[(int64_t)k * 3 + 2];

  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  double acc[D0][D1][S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < D1; ++j)
#pragma unroll
      for (int32_t l = 0; l < S; ++l) acc[i][j][l] = 0.0;

  for (int64_t t = t0; t < t1; ++t) {
    rowv<T> a[D0], b[D1], c[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) a[i] = loadw<T>(in0, t * D0 + i, lane);
#pragma unroll
    for (int32_t i = 0; i < D1; ++i) b[i] = loadw<T>(in1, t * D1 + i, lane);
#pragma unroll
    for (int32_t i = 0; i < S; ++i) c[i] = loadw<T>(in2, t * S + i, lane);
    const int64_t n0 = t * tile_of<T>::value + lane * rowv<T>::W;
#pragma unroll
    for (int32_t v = 0; v < rowv<T>::W; ++v) {
      const double live = (n0 + v < N) ? 1.0 : 0.0;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        const double av = (double)a[i].v[v] * live;
#pragma unroll
        for (int32_t j = 0; j < D1; ++j) {
          const double ab = av * (double)b[j].v[v];
#pragma unroll
          for (int32_t l = 0; l < S; ++l) acc[i][j][l] = fma(ab, (double)c[l].v[v], acc[i][j][l]);
        }
      }
    }
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < D1; ++j)
#pragma unroll
      for (int32_t l = 0; l < S; ++l) {
        double x = acc[i][j][l];
#pragma unroll
        for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) partial[(cpt_off[k] + (i * D1 + j) * S + l) * nchunks + chunk] = x;
      }
}

// =================================================================================
// Host side: dispatch tables
// =================================================================================

#define BN_SMALL_DIMS(X) X(2) X(3) X(4) X(5) X(10)
// cardinalities of a wide parent (the D0 of the leaf-family kernels, the S of the wide-node kernels) whose
// instantiations are part of the library; any other is served by a cardinality pack (cardinality_pack.h)
#define BN_PARENT_DIMS(X) X(2) X(3) X(4) X(5) X(10)
// root cardinalities of the one-launch EM step of small star networks (kernels_star.h)
#define BN_STAR_DIMS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
// (D0, D1, S) of the leaf families with two wide parents that have a fused kernel (CPT [D0][D1][S])
#define BN_TWO_PARENT_SHAPES(X) \
  X(2, 2, 2) X(2, 2, 3) X(2, 2, 4) X(2, 3, 2) X(3, 2, 2) X(2, 3, 3) X(3, 2, 3) X(3, 3, 2) X(3, 3, 3) X(4, 4, 2)
inline bool two_parent_shape(int32_t d0, int32_t d1, int32_t s) {
#define X(A, B, C) \
  if (d0 == A && d1 == B && s == C) return true;
  BN_TWO_PARENT_SHAPES(X)
#undef X
  return false;
}
// cardinalities of the one-parent factor and EM count kernels (cheap to instantiate densely)
#define BN_P1_DIMS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)

template <typename T>
using FactorFn = void (*)(T*, const T*, const int64_t*, const int64_t*, const int64_t*, int32_t, int64_t);
using FactorP1Fn = FactorFn<float>;
template <typename T>
using EmFn = void (*)(const T*, const int64_t*, const int64_t*, int32_t, int64_t, int64_t, int32_t, double*);
using EmP1Fn = EmFn<float>;
template <typename T>
using VarFn = void (*)(T*, const int64_t*, const int64_t*, const int64_t*, int32_t, int64_t, int32_t);
using VarSmallFn = VarFn<float>;

template <typename T, int D0>
inline FactorFn<T> factor_p1_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return factor_p1_fast<T, D0, SV>;
    BN_P1_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T>
inline FactorFn<T> factor_p1_lookup(int32_t d0, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return factor_p1_for_s<T, DV>(s);
    BN_P1_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T, int D0>
inline EmFn<T> em_p1_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return em_p1_fast<T, D0, SV, true>;
    BN_P1_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T>
inline EmFn<T> em_p1_lookup(int32_t d0, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return em_p1_for_s<T, DV>(s);
    BN_P1_DIMS(X)
#undef X
  }
  return nullptr;
}

// Two-parent families: every shape with cardinalities 2..5 whose CPT has at most 40 entries (the
// EM kernel keeps one double accumulator per entry in registers).
#define BN_P2_DIMS(X) X(2) X(3) X(4) X(5)
constexpr int32_t kP2MaxEntries = 40;

template <typename T, int D0, int D1>
inline FactorFn<T> factor_p2_for_s(int32_t s) {
  switch (s) {
#define X(SV)                                                          \
  case SV:                                                             \
    if constexpr (D0 * D1 * SV <= kP2MaxEntries) return factor_p2_fast<T, D0, D1, SV>; \
    break;
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T, int D0>
inline FactorFn<T> factor_p2_for_d1(int32_t d1, int32_t s) {
  switch (d1) {
#define X(DV) \
  case DV:    \
    return factor_p2_for_s<T, D0, DV>(s);
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T>
inline FactorFn<T> factor_p2_lookup(int32_t d0, int32_t d1, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return factor_p2_for_d1<T, DV>(d1, s);
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T, int D0, int D1>
inline EmFn<T> em_p2_for_s(int32_t s) {
  switch (s) {
#define X(SV)                                                      \
  case SV:                                                         \
    if constexpr (D0 * D1 * SV <= kP2MaxEntries) return em_p2_fast<T, D0, D1, SV>; \
    break;
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T, int D0>
inline EmFn<T> em_p2_for_d1(int32_t d1, int32_t s) {
  switch (d1) {
#define X(DV) \
  case DV:    \
    return em_p2_for_s<T, D0, DV>(s);
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T>
inline EmFn<T> em_p2_lookup(int32_t d0, int32_t d1, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return em_p2_for_d1<T, DV>(d1, s);
    BN_P2_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T>
inline EmFn<T> em_root_lookup(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return em_p1_fast<T, 1, SV, false>;
    BN_P1_DIMS(X)
#undef X
  }
  return nullptr;
}

template <typename T, int I>
inline VarFn<T> var_small_for_s(int32_t s) {
  switch (s) {
#define X(SV)                                                        \
  case SV:                                                           \
    if constexpr (I * SV <= 12) return variable_small_fast<T, I, SV>; \
    break;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

// whole node in registers: fan-in up to 4 and at most 12 message entries per observation
template <typename T>
inline VarFn<T> var_small_lookup(int32_t fan_in, int32_t s) {
  switch (fan_in) {
    case 1:
      return var_small_for_s<T, 1>(s);
    case 2:
      return var_small_for_s<T, 2>(s);
    case 3:
      return var_small_for_s<T, 3>(s);
    case 4:
      return var_small_for_s<T, 4>(s);
  }
  return nullptr;
}

inline uint32_t wave_blocks(int64_t items) { return (uint32_t)((items + kWavesPerBlock - 1) / kWavesPerBlock); }

inline bool launch_ok() { return hipGetLastError() == hipSuccess; }

}  // namespace bn
