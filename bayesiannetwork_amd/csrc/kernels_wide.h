// Wide variable nodes (fan-in in the hundreds) and the leaf-family kernels fused around them
// (the leaf-family kernels themselves are in kernels_leaf.h).
//
// A wide node's product P[s] = prod_i in_i[s] cannot be formed by one wave without reading
// megabytes per tile, so it is built from PARTIAL products over chunks of inputs, each a
// double mantissa + int exponent per (state, observation):
//
//   wide_product   one wave per (chunk, tile) multiplies the chunk's inputs         (standalone)
//   wide_combine   one wave per tile multiplies the partials and the few inputs that no
//                  partial covers, writes c = log sum P, the scaled product
//                  P'[s] = P[s] / 2^emax (doubles), and the outgoing messages of the
//                  uncovered inputs (always including the message to the local factor)
//   wide_emit      one wave per (chunk, tile): out_i[s] = (P'[s]/in_i[s]) / sum_s' (...)  (standalone)
//
// When the wide node's children are LEAF FAMILIES — factor nodes with that single parent
// whose own variable has no children — one kernel does the whole iteration for them:
//
//   leaf_family_p1 one wave per (chunk of families, tile); per family k it
//                    reads  the own variable's observation ev and the leaf's normaliser
//                           z = sum_s ev[s] * (factor's previous message to the leaf)[s],
//                    forms  the parent's message to the factor from P' (the wide emission; the
//                           factor's previous message to the parent is cpt . ev / z_prev and the
//                           emission is normalised, so q = cpt . ev is all it needs) and the own
//                           variable's message ev / z with c = log z (its whole variable pass),
//                    runs   both contractions with the CPT,
//                    writes the leaf's next z, and
//                    folds  the new message to the parent into the chunk's partial product for
//                           the next iteration's wide_combine.
//   No message touches HBM inside the loop; all four are stored only in the last iteration of
//   a bn_iterate call, for marginal / EM queries, and read back by the first iteration of the
//   next call.  Per observation and iteration this moves S+1 floats in and 1 out per family (4
//   floats for the 784-leaf model's (10,2) families) instead of the 61 of the unfused passes.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_fast.h"

namespace bn {

struct WideArgs {
  const int64_t* in_off;   // [I]  this node's input offsets (f2v arrays / observation)
  const int64_t* out_off;  // [I]  matching output offsets (v2f arrays), kNoArray = none
  int64_t c_off;
  int32_t I;
  // partial products
  int32_t chunks, chunk_len, n_covered;
  const int32_t* covered;    // [n_covered] input slots represented in the partials, chunk order
  const int32_t* chunk_first;  // [chunks + 1] chunk c holds covered[chunk_first[c] .. chunk_first[c + 1])
  const int32_t* uncovered;  // [n_uncovered] input slots folded by wide_combine; last = local slot
  int32_t n_uncovered;
  double* part_mant;  // [chunks][S][npad]
  int32_t* part_exp;  // [chunks][S][npad]
  // per-observation divisor of each chunk's partial product (mantissa in [0.5, 1), exponent), kept
  // by the leaf-family kernels of a fused node; nullptr = the partial products are complete
  const double* zdiv_mant;  // [chunks][npad]
  const int32_t* zdiv_exp;  // [chunks][npad]
  // the product of all chunk partials per state: prod_mode 1 = store it after forming it,
  // 2 = read it instead of the partials (they have not changed since it was stored), 0 = neither
  double* prod_mant;  // [S][npad]
  int32_t* prod_exp;  // [S][npad]
  int32_t prod_mode;
  // Two iterations' combines in one launch (prod_mode 2, wide_combine_fast): when no kernel runs
  // between them (the loop launch in between is folded into the reduce launch that follows) and the
  // uncovered inputs do not change from one to the next, the launch first forms the earlier
  // iteration's P' from the stored product and the earlier divisors and writes it to early_scaled
  // (nothing else of that iteration is ever read), then does the later iteration's whole combine.
  const double* early_zdiv_mant;  // nullptr = a single combine
  const int32_t* early_zdiv_exp;
  double* early_scaled;  // [S][npad]
  // A combine on fresh messages (wide_combine_fast) of a parentless node can also write the message of
  // the node's own factor - the prior, the same for every observation - which nothing reads before
  // the next combine: the factor pass then has no launch to make for that factor.
  const void* root_cpt;  // [S] of the plan's element type, nullptr = not wanted
  int64_t root_out;      // the message array
  double* scaled;     // [S][npad]   P'[s]
  int64_t npad;
  int64_t ntiles;
  int32_t fresh;
  // while fresh, the covered inputs are all 1/S: their product (1/S)^n_covered is passed in
  // as mantissa * 2^exponent and no partial is read
  double fresh_mant;
  int32_t fresh_exp;
};

// 1 / prod_chunks zdiv for observation n, as a mantissa (0.5, 1] and an exponent
__device__ __forceinline__ void divisor_product(const WideArgs& a, const double* __restrict__ zdiv_mant,
                                                const int32_t* __restrict__ zdiv_exp, int64_t n, double& inv_m,
                                                int32_t& inv_e) {
  double m = 1.0;
  int32_t e = 0;
  // mantissas lie in [0.5, 1): sixteen of them multiply without leaving double range
  for (int32_t ch = 0; ch < a.chunks; ch += 16) {
    double zm[16];
    int32_t ze[16];
#pragma unroll
    for (int32_t u = 0; u < 16; ++u) {
      const bool live = ch + u < a.chunks;
      const int64_t at = (int64_t)(live ? ch + u : ch) * a.npad + n;
      zm[u] = live ? zdiv_mant[at] : 1.0;
      ze[u] = live ? zdiv_exp[at] : 0;
    }
#pragma unroll
    for (int32_t u = 0; u < 16; ++u) {
      m *= zm[u];
      e += ze[u];
    }
    int de = 0;
    m = frexp(m, &de);
    e += de;
  }
  inv_m = 1.0 / m;
  inv_e = -e;
}

template <int S>
__device__ __forceinline__ void renorm(double (&m)[S][4], int32_t (&e)[S][4]) {
#pragma unroll
  for (int32_t s = 0; s < S; ++s)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      int de = 0;
      m[s][v] = frexp(m[s][v], &de);
      e[s][v] += de;
    }
}

template <int S>
__device__ __forceinline__ void store_partial(const WideArgs& a, int32_t chunk, int64_t t, int32_t lane,
                                              const double (&m)[S][4], const int32_t (&e)[S][4]) {
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    const int64_t row = ((int64_t)chunk * S + s) * a.npad + t * kTile + lane * 4;
    *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(m[s][0], m[s][1]);
    *reinterpret_cast<double2*>(a.part_mant + row + 2) = make_double2(m[s][2], m[s][3]);
    *reinterpret_cast<int4*>(a.part_exp + row) = make_int4(e[s][0], e[s][1], e[s][2], e[s][3]);
  }
}

template <int S>
__global__ void __launch_bounds__(256) wide_product_fast(const float* __restrict__ arena, WideArgs a) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles) return;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;

  double m[S][4];
  int32_t e[S][4];
#pragma unroll
  for (int32_t s = 0; s < S; ++s)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      m[s][v] = 1.0;
      e[s][v] = 0;
    }
  const int32_t first = a.chunk_first[chunk], last = a.chunk_first[chunk + 1];
  const double u = 1.0 / (double)S;
  for (int32_t j = first; j < last; ++j) {
    if (a.fresh) {
      // covered inputs are factor messages, all still 1/S
#pragma unroll
      for (int32_t s = 0; s < S; ++s)
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) m[s][v] *= u;
    } else {
      const float* src = arena + a.in_off[a.covered[j]];
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        const f4 x = load4(src, t * S + s, lane);
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) m[s][v] *= (double)x.v[v];
      }
    }
    if (((j - first) & 3) == 3) renorm<S>(m, e);
  }
  renorm<S>(m, e);
  store_partial<S>(a, chunk, t, lane, m, e);
}

// One observation per lane; a block owns 64 observations and its waves share the chunks: wave w
// multiplies the partials of chunks 4w..4w+3, 4(w+nw).., requested four chunks at a time, hands its
// product to wave 0 through LDS, and wave 0 finishes.  The launch is short and latency-bound, so
// it wants the chunk walk as parallel as it can get.
//   grid.x = ntiles * 4, block = 64 * nw (nw <= kCombineMaxWaves), dynamic LDS = combine_lds_bytes<S>(nw)
constexpr int32_t kCombineMaxWaves = 8;
template <int S>
constexpr size_t combine_lds_bytes(int32_t nw) {
  return nw > 1 ? (size_t)nw * S * 64 * (sizeof(double) + sizeof(int32_t)) : 0;
}

template <typename T, int S>
__global__ void __launch_bounds__(64 * kCombineMaxWaves) wide_combine_fast(T* __restrict__ arena, WideArgs a) {
  extern __shared__ __align__(16) unsigned char combine_lds[];
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t nw = (int32_t)(blockDim.x >> 6);
  const int64_t slice = blockIdx.x;
  const int64_t t = slice / (tile_of<T>::value / 64);
  const int32_t col = (int32_t)(slice % (tile_of<T>::value / 64)) * 64 + lane;
  const int64_t n = t * tile_of<T>::value + col;

  double m[S];
  int32_t e[S];
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    m[s] = (a.fresh && wave == 0) ? a.fresh_mant : 1.0;
    e[s] = (a.fresh && wave == 0) ? a.fresh_exp : 0;
  }
  // partial mantissas lie in [0.5, 1) (or are 0): a few dozen of them multiply without leaving
  // double range, so one renormalisation per group of four loads is plenty
  const bool stored = !a.fresh && a.prod_mode == 2;
  if (stored && wave == 0) {
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      m[s] = a.prod_mant[(int64_t)s * a.npad + n];
      e[s] = a.prod_exp[(int64_t)s * a.npad + n];
    }
  }
  if (a.fresh && a.root_cpt && wave == 0) {
    const T* prior = static_cast<const T*>(a.root_cpt);
#pragma unroll
    for (int32_t s = 0; s < S; ++s) arena[a.root_out + (t * S + s) * tile_of<T>::value + col] = prior[s];
  }
  if (stored && wave == 0 && a.early_zdiv_mant) {
    double m1[S];
    int32_t e1[S];
    double inv_m;
    int32_t inv_e;
    divisor_product(a, a.early_zdiv_mant, a.early_zdiv_exp, n, inv_m, inv_e);
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      m1[s] = m[s] * inv_m;
      e1[s] = e[s] + inv_e;
    }
    for (int32_t u = 0; u < a.n_uncovered; ++u) {
      const T* src = arena + a.in_off[a.uncovered[u]] + t * S * tile_of<T>::value + col;
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        int de = 0;
        m1[s] = frexp(m1[s] * (double)src[s * tile_of<T>::value], &de);
        e1[s] += de;
      }
    }
    int32_t emax1 = kDeadExponent;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      int de = 0;
      m1[s] = frexp(m1[s], &de);
      e1[s] = m1[s] == 0.0 ? kDeadExponent : e1[s] + de;
      emax1 = e1[s] > emax1 ? e1[s] : emax1;
    }
#pragma unroll
    for (int32_t s = 0; s < S; ++s) a.early_scaled[(int64_t)s * a.npad + n] = ldexp(m1[s], clamp_shift(e1[s] - emax1));
  }
  for (int32_t ch = wave * 4; ch < (a.fresh || stored ? 0 : a.chunks); ch += nw * 4) {
    double pm[4][S];
    int32_t pe[4][S];
#pragma unroll
    for (int32_t u = 0; u < 4; ++u) {
      const bool live = ch + u < a.chunks;
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        const int64_t row = ((int64_t)(live ? ch + u : ch) * S + s) * a.npad + n;
        pm[u][s] = a.part_mant[row];
        pe[u][s] = a.part_exp[row];
        if (!live) {
          pm[u][s] = 1.0;
          pe[u][s] = 0;
        }
      }
    }
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
#pragma unroll
      for (int32_t u = 0; u < 4; ++u) {
        m[s] *= pm[u][s];
        e[s] += pe[u][s];
      }
      int de = 0;
      m[s] = frexp(m[s], &de);
      e[s] += de;
    }
  }
  if (nw > 1) {
    double* lds_m = reinterpret_cast<double*>(combine_lds);                      // [nw][S][64]
    int32_t* lds_e = reinterpret_cast<int32_t*>(combine_lds + (size_t)nw * S * 64 * sizeof(double));
    if (wave > 0) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        lds_m[(wave * S + s) * 64 + lane] = m[s];
        lds_e[(wave * S + s) * 64 + lane] = e[s];
      }
    }
    __syncthreads();
    if (wave > 0) return;
    for (int32_t w = 1; w < nw; ++w) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        int de = 0;
        m[s] = frexp(m[s] * lds_m[(w * S + s) * 64 + lane], &de);
        e[s] += lds_e[(w * S + s) * 64 + lane] + de;
      }
    }
  }
  if (!a.fresh && a.prod_mode == 1) {
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      a.prod_mant[(int64_t)s * a.npad + n] = m[s];
      a.prod_exp[(int64_t)s * a.npad + n] = e[s];
    }
  }
  if (!a.fresh && a.zdiv_mant) {
    double inv_m;
    int32_t inv_e;
    divisor_product(a, a.zdiv_mant, a.zdiv_exp, n, inv_m, inv_e);
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      m[s] *= inv_m;
      e[s] += inv_e;
    }
  }
  // inputs no partial covers
  const T uniform = 1.0f / (T)S;
  for (int32_t u = 0; u < a.n_uncovered; ++u) {
    const int32_t i = a.uncovered[u];
    const bool skip = a.fresh && a.out_off[i] != kNoArray;
    const T* src = arena + a.in_off[i] + t * S * tile_of<T>::value + col;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const T x = skip ? uniform : src[s * tile_of<T>::value];
      int de = 0;
      m[s] = frexp(m[s] * (double)x, &de);
      e[s] += de;
    }
  }

  int32_t emax = kDeadExponent;
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    int de = 0;
    m[s] = frexp(m[s], &de);
    e[s] = m[s] == 0.0 ? kDeadExponent : e[s] + de;
    emax = e[s] > emax ? e[s] : emax;
  }
  double z = 0.0;
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    m[s] = ldexp(m[s], clamp_shift(e[s] - emax));  // P'[s]
    z += m[s];
  }
  arena[a.c_off + n] = (T)((double)emax * 0.6931471805599453 + log(z));
#pragma unroll
  for (int32_t s = 0; s < S; ++s) a.scaled[(int64_t)s * a.npad + n] = m[s];
  // outgoing messages of the uncovered inputs
  for (int32_t u = 0; u < a.n_uncovered; ++u) {
    const int32_t i = a.uncovered[u];
    const int64_t dst = a.out_off[i];
    if (dst == kNoArray) continue;
    const bool skip = a.fresh != 0;
    const T* src = arena + a.in_off[i] + t * S * tile_of<T>::value + col;
    double q[S];
    double sum = 0.0;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const T x = skip ? uniform : src[s * tile_of<T>::value];
      q[s] = m[s] / (double)x;
      sum += q[s];
    }
    const bool local = (i == a.I - 1);
    sum = 1.0 / (local ? z : sum);
    T* out = arena + dst + t * S * tile_of<T>::value + col;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) out[s * tile_of<T>::value] = (T)(q[s] * sum);
  }
}

// The same combine for small batches, where a launch is a handful of blocks and its duration is the
// length of its longest dependent chain: one wave per state, so that a wave's loads of all chunk
// partials are in flight together (one round trip instead of one per group of chunks) and no wave
// merges the others' products one after the other.
//   grid.x = ntiles * tile / 64, block = 64 * S, dynamic LDS = combine_states_lds_bytes<S>(n_uncovered)
template <int S>
constexpr size_t combine_states_lds_bytes(int32_t n_uncovered) {
  return (size_t)S * 64 * (sizeof(double) + sizeof(int32_t)) + (size_t)n_uncovered * S * 64 * sizeof(double);
}

template <typename T, int S>
__global__ void __launch_bounds__(64 * S) wide_combine_states(T* __restrict__ arena, WideArgs a) {
  extern __shared__ __align__(16) unsigned char combine_lds[];
  double* lds_m = reinterpret_cast<double*>(combine_lds);                                    // [S][64]
  int32_t* lds_e = reinterpret_cast<int32_t*>(combine_lds + (size_t)S * 64 * sizeof(double));  // [S][64]
  double* lds_q = reinterpret_cast<double*>(combine_lds + (size_t)S * 64 * (sizeof(double) + sizeof(int32_t)));  // [n_uncovered][S][64]
  const int32_t s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // this wave's state
  const int64_t slice = blockIdx.x;
  const int64_t t = slice / (tile_of<T>::value / 64);
  const int32_t col = (int32_t)(slice % (tile_of<T>::value / 64)) * 64 + lane;
  const int64_t n = t * tile_of<T>::value + col;

  double m = a.fresh ? a.fresh_mant : 1.0;
  int32_t e = a.fresh ? a.fresh_exp : 0;
  if (!a.fresh) {
    const bool stored = a.prod_mode == 2;
    if (stored) {
      m = a.prod_mant[(int64_t)s * a.npad + n];
      e = a.prod_exp[(int64_t)s * a.npad + n];
    }
    // mantissas lie in [0.5, 1) or are 0: sixteen of them multiply without leaving double range
    for (int32_t ch = 0; ch < (stored ? 0 : a.chunks); ch += 16) {
      double pm[16];
      int32_t pe[16];
#pragma unroll
      for (int32_t u = 0; u < 16; ++u) {
        const bool live = ch + u < a.chunks;
        const int64_t row = ((int64_t)(live ? ch + u : ch) * S + s) * a.npad + n;
        pm[u] = a.part_mant[row];
        pe[u] = a.part_exp[row];
        if (!live) {
          pm[u] = 1.0;
          pe[u] = 0;
        }
      }
#pragma unroll
      for (int32_t u = 0; u < 16; ++u) {
        m *= pm[u];
        e += pe[u];
      }
      int de = 0;
      m = frexp(m, &de);
      e += de;
    }
    if (a.prod_mode == 1) {
      a.prod_mant[(int64_t)s * a.npad + n] = m;
      a.prod_exp[(int64_t)s * a.npad + n] = e;
    }
    if (a.zdiv_mant) {
      double inv_m;
      int32_t inv_e;
      divisor_product(a, a.zdiv_mant, a.zdiv_exp, n, inv_m, inv_e);
      m *= inv_m;
      e += inv_e;
    }
  }
  // inputs no partial covers (their values are needed again for the outgoing messages)
  const T uniform = 1.0f / (T)S;
  T x[kMaxUncovered];
#pragma unroll
  for (int32_t u = 0; u < kMaxUncovered; ++u) {
    if (u < a.n_uncovered) {
      const int32_t i = a.uncovered[u];
      const bool skip = a.fresh && a.out_off[i] != kNoArray;
      x[u] = skip ? uniform : arena[a.in_off[i] + (t * S + s) * tile_of<T>::value + col];
      int de = 0;
      m = frexp(m * (double)x[u], &de);
      e += de;
    }
  }
  {
    int de = 0;
    m = frexp(m, &de);
    e = m == 0.0 ? kDeadExponent : e + de;
  }
  lds_m[s * 64 + lane] = m;
  lds_e[s * 64 + lane] = e;
  __syncthreads();
  int32_t emax = kDeadExponent;
#pragma unroll
  for (int32_t j = 0; j < S; ++j) {
    const int32_t ej = lds_e[j * 64 + lane];
    emax = ej > emax ? ej : emax;
  }
  double z = 0.0;
#pragma unroll
  for (int32_t j = 0; j < S; ++j) z += ldexp(lds_m[j * 64 + lane], clamp_shift(lds_e[j * 64 + lane] - emax));
  const double scaled = ldexp(m, clamp_shift(e - emax));  // P'[s]
  if (s == 0) arena[a.c_off + n] = (T)((double)emax * 0.6931471805599453 + log(z));
  a.scaled[(int64_t)s * a.npad + n] = scaled;
  // outgoing messages of the uncovered inputs: P' / input, normalised over the states (the message to
  // the local factor by z, as the reference's variable pass does)
#pragma unroll
  for (int32_t u = 0; u < kMaxUncovered; ++u) {
    if (u < a.n_uncovered) {
      const T xin = (a.fresh != 0) ? uniform : x[u];
      lds_q[(u * S + s) * 64 + lane] = scaled / (double)xin;
    }
  }
  __syncthreads();
#pragma unroll
  for (int32_t u = 0; u < kMaxUncovered; ++u) {
    if (u < a.n_uncovered) {
      const int32_t i = a.uncovered[u];
      const int64_t dst = a.out_off[i];
      if (dst != kNoArray) {
        double sum = 0.0;
#pragma unroll
        for (int32_t j = 0; j < S; ++j) sum += lds_q[(u * S + j) * 64 + lane];
        const bool local = (i == a.I - 1);
        sum = 1.0 / (local ? z : sum);
        arena[dst + (t * S + s) * tile_of<T>::value + col] = (T)(lds_q[(u * S + s) * 64 + lane] * sum);
      }
    }
  }
}

// divisors of the partial products := 1 (after the partial products were rebuilt from stored messages)
template <typename E>
__global__ void __launch_bounds__(256) wide_divisor_one(double* __restrict__ zdiv_mant, E* __restrict__ zdiv_exp,
                                                       int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    zdiv_mant[i] = 0.5;
    zdiv_exp[i] = 1;
  }
}

template <int S>
__global__ void __launch_bounds__(256) wide_emit_fast(float* __restrict__ arena, WideArgs a) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles) return;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;

  double p[S][4];
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    const int64_t row = (int64_t)s * a.npad + t * kTile + lane * 4;
    const double2 lo = *reinterpret_cast<const double2*>(a.scaled + row);
    const double2 hi = *reinterpret_cast<const double2*>(a.scaled + row + 2);
    p[s][0] = lo.x;
    p[s][1] = lo.y;
    p[s][2] = hi.x;
    p[s][3] = hi.y;
  }
  const int32_t first = a.chunk_first[chunk], last = a.chunk_first[chunk + 1];
  const float uniform = 1.0f / (float)S;
  for (int32_t j = first; j < last; ++j) {
    const int32_t i = a.covered[j];
    const int64_t dst = a.out_off[i];
    if (dst == kNoArray) continue;
    const float* src = arena + a.in_off[i];
    double q[S][4];
    double sum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      f4 x;
      if (a.fresh) {
        x = f4{{uniform, uniform, uniform, uniform}};
      } else {
        x = load4(src, t * S + s, lane);
      }
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) {
        q[s][v] = p[s][v] / (double)x.v[v];
        sum[v] += q[s][v];
      }
    }
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) sum[v] = 1.0 / sum[v];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      f4 o;
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) o.v[v] = (float)(q[s][v] * sum[v]);
      store4(arena + dst, t * S + s, lane, o);
    }
  }
}

}  // namespace bn
