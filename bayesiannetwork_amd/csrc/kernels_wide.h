// Wide variable nodes (fan-in in the hundreds) and the leaf-family kernel fused around them.
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
  const int64_t* out_off;  // [I]  matching output offsets (v2f arrays), -1 = dead
  int64_t c_off;
  int32_t I;
  // partial products
  int32_t chunks, chunk_len, n_covered;
  const int32_t* covered;    // [n_covered] input slots represented in the partials, chunk order
  const int32_t* uncovered;  // [n_uncovered] input slots folded by wide_combine; last = local slot
  int32_t n_uncovered;
  double* part_mant;  // [chunks][S][npad]
  int32_t* part_exp;  // [chunks][S][npad]
  double* scaled;     // [S][npad]   P'[s]
  int64_t npad;
  int64_t ntiles;
  int32_t fresh;
  // while fresh, the covered inputs are all 1/S: their product (1/S)^n_covered is passed in
  // as mantissa * 2^exponent and no partial is read
  double fresh_mant;
  int32_t fresh_exp;
};

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
  const int32_t first = chunk * a.chunk_len;
  const int32_t last = first + a.chunk_len < a.n_covered ? first + a.chunk_len : a.n_covered;
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

template <int S>
__global__ void __launch_bounds__(64 * kCombineMaxWaves) wide_combine_fast(float* __restrict__ arena, WideArgs a) {
  extern __shared__ __align__(16) unsigned char combine_lds[];
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t nw = (int32_t)(blockDim.x >> 6);
  const int64_t slice = blockIdx.x;
  const int64_t t = slice / (kTile / 64);
  const int32_t col = (int32_t)(slice % (kTile / 64)) * 64 + lane;
  const int64_t n = t * kTile + col;

  double m[S];
  int32_t e[S];
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    m[s] = (a.fresh && wave == 0) ? a.fresh_mant : 1.0;
    e[s] = (a.fresh && wave == 0) ? a.fresh_exp : 0;
  }
  // partial mantissas lie in [0.5, 1) (or are 0): a few dozen of them multiply without leaving
  // double range, so one renormalisation per group of four loads is plenty
  for (int32_t ch = wave * 4; ch < (a.fresh ? 0 : a.chunks); ch += nw * 4) {
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
  // inputs no partial covers
  const float uniform = 1.0f / (float)S;
  for (int32_t u = 0; u < a.n_uncovered; ++u) {
    const int32_t i = a.uncovered[u];
    const bool skip = a.fresh && a.out_off[i] >= 0;
    const float* src = arena + a.in_off[i] + t * S * kTile + col;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const float x = skip ? uniform : src[s * kTile];
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
  arena[a.c_off + n] = (float)((double)emax * 0.6931471805599453 + log(z));
#pragma unroll
  for (int32_t s = 0; s < S; ++s) a.scaled[(int64_t)s * a.npad + n] = m[s];
  // outgoing messages of the uncovered inputs
  for (int32_t u = 0; u < a.n_uncovered; ++u) {
    const int32_t i = a.uncovered[u];
    const int64_t dst = a.out_off[i];
    if (dst < 0) continue;
    const bool skip = a.fresh != 0;
    const float* src = arena + a.in_off[i] + t * S * kTile + col;
    double q[S];
    double sum = 0.0;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const float x = skip ? uniform : src[s * kTile];
      q[s] = m[s] / (double)x;
      sum += q[s];
    }
    const bool local = (i == a.I - 1);
    sum = 1.0 / (local ? z : sum);
    float* out = arena + dst + t * S * kTile + col;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) out[s * kTile] = (float)(q[s] * sum);
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
  const int32_t first = chunk * a.chunk_len;
  const int32_t last = first + a.chunk_len < a.n_covered ? first + a.chunk_len : a.n_covered;
  const float uniform = 1.0f / (float)S;
  for (int32_t j = first; j < last; ++j) {
    const int32_t i = a.covered[j];
    const int64_t dst = a.out_off[i];
    if (dst < 0) continue;
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

// ---------------------------------------------------------------------------------
// Leaf families with one (wide) parent: see the file header.
struct LeafFamilyArgs {
  const int64_t* f_in;     // [K*arity]  v2f offsets of the factor group (parent slots, own slot)
  const int64_t* f_out;    // [K*arity]  f2v offsets
  const int64_t* cpt_off;  // [K]
  const int64_t* obs_off;  // [K]    observation array of the own variable, -1 = not observed
  const int64_t* c_off;    // [K]    normaliser of the own variable
  int32_t K;
  int32_t chunks, chunk_len;
  double* part_mant;
  int32_t* part_exp;
  const double* scaled;  // parent's P'
  int64_t npad;
  int64_t ntiles;
  float* z_state;  // [K][npad] loop state: the leaf's normaliser z for the coming iteration
  // kLeafReduce only
  int64_t N;
  float* em_partial;  // [K][D0*S][slots] per-wave sums of (message from parent) x (message from own variable); nullptr = not wanted
  int32_t slots;      // ntiles * waves per tile
  double* ll_part;    // [chunks][npad] sum over the chunk's families of c = log z, per observation
  // two-parent families (leaf_family_p2) only: the second wide parent's partial products and P',
  // and the loop state that holds the factor's previous messages to both parents
  double* part_mant2;
  int32_t* part_exp2;
  const double* scaled2;
  float* m_state;  // [K][D0 + D1][npad]
};
// Template parameter MODE
//   kLeafLoop    inside a bn_iterate call: only the loop state (the next z) is written;
//   kLeafFull    last iteration of a bn_iterate call: everything queries may read is written
//                (both variable -> factor messages, the message to the parent, c = log z);
//   kLeafReduce  last iteration of an EM step (bn_em_statistics): nothing of the message state is
//                written; the iteration's values are used on the fly for the per-observation sums
//                of c = log z over each chunk's families and for the per-wave EM count sums.  The stored state stays one iteration behind, and the
//                host re-runs this iteration in kLeafFull mode if anything asks for messages.
constexpr int32_t kLeafLoop = 0, kLeafFull = 1, kLeafReduce = 2;
// Where the iteration's input state comes from, also a compile-time choice (folded into the
// kernel's MODE template argument as MODE = mode + 3 * state):
//   kLeafSteady      the loop state a kLeafLoop launch left (z_state);
//   kLeafFresh       nothing has been sent yet: every message still has its initial uniform value;
//   kLeafFromArrays  the message arrays a kLeafFull launch wrote.
constexpr int32_t kLeafSteady = 0, kLeafFresh = 1, kLeafFromArrays = 2;

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sums E per-lane values over the 64 lanes of the wave with fewer operations than E full
// reductions: the first two steps are a reduce-scatter inside each quad (after them a lane only
// carries a quarter of the entries: those of its position in the quad), the rest adds the 16
// quads (row rotations, then the two cross-row exchanges).  Afterwards lane q < 4 holds in
// out[f] the total of entry index wave_sum_entry<E>(q, f) (or nothing if that is >= E).
template <int E>
struct WaveSums {
  static constexpr int H1 = (E + 1) / 2, H2 = (H1 + 1) / 2;
  float out[H2];
  __device__ __forceinline__ static int32_t entry(int32_t q, int32_t f) {
    const int32_t e = f + ((q >> 1) & 1) * H2;  // index after the first step
    return e < H1 ? e + (q & 1) * H1 : E;
  }
  __device__ __forceinline__ void reduce(const float (&p)[E], int32_t lane) {
    const bool odd = lane & 1, upper = lane & 2;
    float y[H1];
#pragma unroll
    for (int32_t e = 0; e < H1; ++e) {
      const float a = p[e], b = e + H1 < E ? p[e + H1] : 0.f;
      y[e] = (odd ? b : a) + dpp_move<0xB1>(odd ? a : b);  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int32_t f = 0; f < H2; ++f) {
      const float a = y[f], b = f + H2 < H1 ? y[f + H2] : 0.f;
      float w = (upper ? b : a) + dpp_move<0x4E>(upper ? a : b);  // quad_perm [2,3,0,1]
      w += dpp_move<0x124>(w);                                     // row_ror:4
      w += dpp_move<0x128>(w);                                     // row_ror:8
      w += __shfl_xor(w, 16, 64);
      w += __shfl_xor(w, 32, 64);
      out[f] = w;
    }
  }
};

// V floats of one lane as a hardware vector: arithmetic on it compiles to the packed
// two-float instructions (v_pk_mul/add/fma_f32) when V is even.
template <int V>
using fv = float __attribute__((ext_vector_type(V)));

template <int V>
__device__ __forceinline__ fv<V> loadv(const float* __restrict__ base, int64_t row, int32_t col) {
  return *reinterpret_cast<const fv<V>*>(base + row * kTile + col);
}

template <int V>
__device__ __forceinline__ void storev(float* __restrict__ base, int64_t row, int32_t col, const fv<V>& r) {
  *reinterpret_cast<fv<V>*>(base + row * kTile + col) = r;
}

template <int V>
__device__ __forceinline__ fv<V> splat(float x) {
  fv<V> r;
#pragma unroll
  for (int32_t v = 0; v < V; ++v) r[v] = x;
  return r;
}

template <int V>
__device__ __forceinline__ fv<V> rcpv(const fv<V>& x) {
  fv<V> r;
#pragma unroll
  for (int32_t v = 0; v < V; ++v) r[v] = __builtin_amdgcn_rcpf(x[v]);
  return r;
}

// V observations per lane: a wave covers 64*V observations, i.e. 4/V work items per tile.
//
// The loop over the chunk's families is software-pipelined by hand: the small loop state of
// family k+1 (its pointers, observation, normaliser) is requested before the arithmetic of
// family k, so that with a few waves per SIMD the memory latency of one family hides behind the
// arithmetic of the previous one.
struct LeafPtrs {  // wave-uniform
  float* f2v_par;
  float* f2v_own;
  float* v2f_par;
  float* v2f_own;
  float* c;
  float* z;          // loop-state normaliser of this family
  const float* obs;  // the observation, or an all-ones array when the leaf is not observed
  const float* cpt;
};

template <int S, int V>
struct LeafState {
  fv<V> ev[S];
  fv<V> z;  // the leaf's normaliser for this iteration: sum_s ev[s] * (previous message from the factor)[s]
};

__device__ __forceinline__ LeafPtrs leaf_ptrs(float* __restrict__ arena, const float* __restrict__ cpts,
                                              const LeafFamilyArgs& a, int32_t k) {
  LeafPtrs p;
  p.f2v_par = arena + a.f_out[2 * k];
  p.f2v_own = arena + a.f_out[2 * k + 1];
  p.v2f_par = arena + a.f_in[2 * k];
  p.v2f_own = arena + a.f_in[2 * k + 1];
  p.c = arena + a.c_off[k];
  p.z = a.z_state + (int64_t)k * a.npad;
  p.obs = arena + a.obs_off[k];
  p.cpt = cpts + a.cpt_off[k];
  return p;
}

// z of one leaf from its evidence and the factor's previous message
template <int S, int V>
__device__ __forceinline__ fv<V> leaf_normaliser(const fv<V>* ev, const fv<V>* from_factor) {
  fv<V> z = ev[0] * from_factor[0];
#pragma unroll
  for (int32_t j = 1; j < S; ++j) z = __builtin_elementwise_fma(ev[j], from_factor[j], z);
  return z;
}

// 1/x to within an ulp: hardware seed plus one Newton step
template <int V>
__device__ __forceinline__ fv<V> recipv(const fv<V>& x) {
  const fv<V> r = rcpv<V>(x);
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-x, r, splat<V>(1.f)), r, r);
}

template <int S, int V, int STATE>
__device__ __forceinline__ LeafState<S, V> leaf_state(const LeafPtrs& p, const LeafFamilyArgs& a, int64_t t,
                                                      int32_t col) {
  LeafState<S, V> st;
#pragma unroll
  for (int32_t j = 0; j < S; ++j) st.ev[j] = loadv<V>(p.obs, t * S + j, col);
  if constexpr (STATE != kLeafSteady) {
    fv<V> own[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j)
      own[j] = STATE == kLeafFresh ? splat<V>(1.0f / (float)S) : loadv<V>(p.f2v_own, t * S + j, col);
    st.z = leaf_normaliser<S, V>(st.ev, own);
  } else {
    st.z = loadv<V>(p.z, t, col);
  }
  return st;
}

// The partial product's exponents, one per (state, observation): int16 pairs when V is 2 (packed
// adds, half the registers; chunks are capped at kLeafMaxChunkLen families so that even messages
// at the bottom of the float range cannot overflow 16 bits), plain int32 otherwise.
constexpr int32_t kLeafMaxChunkLen = 192;
template <int V>
using ev_t = typename std::conditional<V == 2, short, int32_t>::type __attribute__((ext_vector_type(V)));

template <int D0, int S, int V, int MODE>
__global__ void __launch_bounds__(256) leaf_family_p1(float* __restrict__ arena, const float* __restrict__ cpts,
                                                     LeafFamilyArgs a) {
  constexpr bool FULL = MODE % 3 == kLeafFull;
  constexpr bool REDUCE = MODE % 3 == kLeafReduce;
  constexpr int32_t STATE = MODE / 3;
  constexpr int32_t kParts = 4 / V;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles * kParts) return;
  const int32_t part = (int32_t)(w % kParts);
  w /= kParts;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;
  const int32_t col = part * 64 * V + lane * V;  // first observation of this lane inside the tile

  double pm[D0][V];
  ev_t<V> pe[D0];
#pragma unroll
  for (int32_t s = 0; s < D0; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      pm[s][v] = 1.0;
      pe[s][v] = 0;
    }

  const int32_t first = chunk * a.chunk_len;
  const int32_t last = first + a.chunk_len < a.K ? first + a.chunk_len : a.K;
  if (first >= last) return;
  // The parent's scaled product P' (largest state = 1) for this lane's observations, in float.
  // A state whose share is below float range reads as zero: its message to a leaf factor would
  // be below 2^-100 of the leading state's and cannot change any sum it enters.
  fv<V> pf[D0];
  {
    const double* scaled = a.scaled + t * kTile + col;
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      double pv[V];
      if constexpr (V == 4) {
        const double2 lo = *reinterpret_cast<const double2*>(scaled + (int64_t)i * a.npad);
        const double2 hi = *reinterpret_cast<const double2*>(scaled + (int64_t)i * a.npad + 2);
        pv[0] = lo.x;
        pv[1] = lo.y;
        pv[2] = hi.x;
        pv[3] = hi.y;
      } else if constexpr (V == 2) {
        const double2 lo = *reinterpret_cast<const double2*>(scaled + (int64_t)i * a.npad);
        pv[0] = lo.x;
        pv[1] = lo.y;
      } else {
        pv[0] = scaled[(int64_t)i * a.npad];
      }
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pf[i][v] = (float)pv[v];
    }
  }

  // loop mode keeps the leaves' normalisers out of the per-state products: prod_k (q_k / z_k) is
  // carried as prod_k q_k per state and prod_k z_k per observation, divided once at the end
  fv<V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  double pz[V];
  int32_t pze[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
    pz[v] = 1.0;
    pze[v] = 0;
  }

  LeafPtrs ptr = leaf_ptrs(arena, cpts, a, first);
  LeafState<S, V> cur = leaf_state<S, V, STATE>(ptr, a, t, col);
  for (int32_t k = first; k < last; ++k) {
    const int32_t kn = k + 1 < last ? k + 1 : k;
    const LeafPtrs ptr_next = leaf_ptrs(arena, cpts, a, kn);
    const LeafState<S, V> nxt = leaf_state<S, V, STATE>(ptr_next, a, t, col);
    const float* cpt = ptr.cpt;

    // q = cpt . ev.  The leaf's message to the factor is ev / z, so the factor's message to the
    // parent is q / z; and because the parent's message back is normalised over its states, q
    // also stands in for the factor's previous message to the parent (q / z_prev) in the quotient.
    fv<V> q[D0];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      q[i] = cpt[i * S] * cur.ev[0];
#pragma unroll
      for (int32_t j = 1; j < S; ++j) q[i] = __builtin_elementwise_fma(splat<V>(cpt[i * S + j]), cur.ev[j], q[i]);
    }
    // parent's message to this factor, before normalisation: r = P' / (previous message to the parent)
    fv<V> r[D0];
    fv<V> inv_sum = splat<V>(0.f);
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      fv<V> prev;
      if constexpr (STATE == kLeafFresh) {
        prev = splat<V>(1.0f / (float)D0);
      } else if constexpr (STATE == kLeafFromArrays) {
        prev = loadv<V>(ptr.f2v_par, t * D0 + i, col);
      } else {
        prev = q[i];
      }
      r[i] = pf[i] * rcpv<V>(prev);
      inv_sum += r[i];
    }
    inv_sum = rcpv<V>(inv_sum);
    if constexpr (!FULL && !REDUCE) {
      // Loop mode needs neither message explicitly.  The leaf's next normaliser is
      //   z' = sum_s ev[s] * (cpt^T t)[s] = sum_i t[i] * q[i],  t = r / sum r,
      // which in the steady state (r = P'/q) is sum(P') / sum(r).
      fv<V> zn;
      if constexpr (STATE == kLeafSteady) {
        zn = pf_sum * inv_sum;
      } else {
        zn = r[0] * q[0];
#pragma unroll
        for (int32_t i = 1; i < D0; ++i) zn = __builtin_elementwise_fma(r[i], q[i], zn);
        zn *= inv_sum;
      }
      storev<V>(ptr.z, t, col, zn);
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
#pragma unroll
        for (int32_t i = 0; i < D0; ++i) pm[i][v] *= (double)q[i][v];
        pz[v] *= (double)cur.z[v];
      }
      if (((k - first) & 3) == 3) {
#pragma unroll
        for (int32_t i = 0; i < D0; ++i) {
          ev_t<V> de_v;
#pragma unroll
          for (int32_t v = 0; v < V; ++v) {
            int de = 0;
            pm[i][v] = frexp(pm[i][v], &de);
            de_v[v] = de;
          }
          pe[i] += de_v;
        }
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          pz[v] = frexp(pz[v], &de);
          pze[v] += de;
        }
      }
      ptr = ptr_next;
      cur = nxt;
      continue;
    }
    if constexpr (REDUCE) {
      // sum_k log z_k = log prod_k z_k: the product is carried like in loop mode, one log at the end
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pz[v] *= (double)cur.z[v];
      if (((k - first) & 3) == 3) {
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          pz[v] = frexp(pz[v], &de);
          pze[v] += de;
        }
      }
      if (a.em_partial) {
        // EM counts of this family over the wave's observations (padding masked), wave-reduced:
        // sum_n t[i] * o[j] with t = r / sum r and o = ev / z; both scalings go onto the S-vector
        const fv<V> scale = recipv<V>(cur.z) * inv_sum;
        fv<V> o[S];
#pragma unroll
        for (int32_t j = 0; j < S; ++j) {
          o[j] = cur.ev[j] * scale;
#pragma unroll
          for (int32_t v = 0; v < V; ++v)
            if (t * kTile + col + v >= a.N) o[j][v] = 0.f;
        }
        float* dst = a.em_partial + ((int64_t)k * (D0 * S)) * a.slots + (t * kParts + part);
        float per_lane[D0 * S];
#pragma unroll
        for (int32_t i = 0; i < D0; ++i)
#pragma unroll
          for (int32_t j = 0; j < S; ++j) {
            const fv<V> prod = r[i] * o[j];
            float x = prod[0];
#pragma unroll
            for (int32_t v = 1; v < V; ++v) x += prod[v];
            per_lane[i * S + j] = x;
          }
        WaveSums<D0 * S> sums;
        sums.reduce(per_lane, lane);
        if (lane < 4) {
#pragma unroll
          for (int32_t f = 0; f < WaveSums<D0 * S>::H2; ++f) {
            const int32_t e = WaveSums<D0 * S>::entry(lane, f);
            if (e < D0 * S) dst[(int64_t)e * a.slots] = sums.out[f];
          }
        }
      }
      ptr = ptr_next;
      cur = nxt;
      continue;
    }
    // own (leaf) variable: c = log z, message to the factor = ev / z
    const fv<V> inv_z = recipv<V>(cur.z);
    {
      fv<V> c;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        const float mz = frexpf(cur.z[v], &de);
        c[v] = (float)((double)de * 0.6931471805599453 + (double)logf(mz));
      }
      storev<V>(ptr.c, t, col, c);
    }
    fv<V> new_par[D0], new_own[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) new_par[i] = q[i] * inv_z;
    if constexpr (FULL || REDUCE) {
      fv<V> to_f_own[S];
#pragma unroll
      for (int32_t j = 0; j < S; ++j) to_f_own[j] = cur.ev[j] * inv_z;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) r[i] *= inv_sum;  // now the normalised message
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) storev<V>(ptr.v2f_par, t * D0 + i, col, r[i]);
#pragma unroll
      for (int32_t j = 0; j < S; ++j) storev<V>(ptr.v2f_own, t * S + j, col, to_f_own[j]);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) storev<V>(ptr.f2v_par, t * D0 + i, col, new_par[i]);
    }
    // factor's message to the leaf: cpt^T . (normalised message from the parent)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) new_own[j] = cpt[j] * r[0];
#pragma unroll
    for (int32_t i = 1; i < D0; ++i)
#pragma unroll
      for (int32_t j = 0; j < S; ++j) new_own[j] = __builtin_elementwise_fma(splat<V>(cpt[i * S + j]), r[i], new_own[j]);
#pragma unroll
    for (int32_t j = 0; j < S; ++j) storev<V>(ptr.f2v_own, t * S + j, col, new_own[j]);
    // fold into the partial product for the parent's next combine
#pragma unroll
    for (int32_t i = 0; i < D0; ++i)
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pm[i][v] *= (double)new_par[i][v];
    if (((k - first) & 3) == 3) {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        ev_t<V> de_v;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          pm[i][v] = frexp(pm[i][v], &de);
          de_v[v] = de;
        }
        pe[i] += de_v;
      }
    }
    ptr = ptr_next;
    cur = nxt;
  }
  if constexpr (REDUCE) {
    double* dst = a.ll_part + (int64_t)chunk * a.npad + t * kTile + col;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      const double m = frexp(pz[v], &de);
      dst[v] = (double)(pze[v] + de) * 0.6931471805599453 + log(m);
    }
    return;
  }
  double inv_pz[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
    int de = 0;
    inv_pz[v] = 1.0 / frexp(pz[v], &de);  // pz stayed 1 in full mode
    pze[v] += de;
  }
#pragma unroll
  for (int32_t s = 0; s < D0; ++s) {
    int32_t ex[V];
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      pm[s][v] = frexp(pm[s][v] * inv_pz[v], &de);
      ex[v] = (int32_t)pe[s][v] + de - pze[v];
    }
    const int64_t row = ((int64_t)chunk * D0 + s) * a.npad + t * kTile + col;
    if constexpr (V == 4) {
      *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(pm[s][0], pm[s][1]);
      *reinterpret_cast<double2*>(a.part_mant + row + 2) = make_double2(pm[s][2], pm[s][3]);
      *reinterpret_cast<int4*>(a.part_exp + row) = make_int4(ex[0], ex[1], ex[2], ex[3]);
    } else if constexpr (V == 2) {
      *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(pm[s][0], pm[s][1]);
      *reinterpret_cast<int2*>(a.part_exp + row) = make_int2(ex[0], ex[1]);
    } else {
      a.part_mant[row] = pm[s][0];
      a.part_exp[row] = ex[0];
    }
  }
}

// ---------------------------------------------------------------------------------
// Leaf families with TWO wide parents (CPT [D0][D1][S], e.g. the loopy 784-leaf model's
// Y | F1, F2).  Same structure as leaf_family_p1; differences:
//   * the factor's previous messages to the parents depend on the other parent's message, so they
//     cannot be rebuilt from the evidence: the loop state is m1[D0], m2[D1] and the leaf's next z;
//   * the kernel maintains the chunk partial products of both parents.
// Per family and observation the loop moves D0+D1+1+S floats in and D0+D1+1 out (7 + 5 for the
// (2,2,2) families) instead of the 36 of the unfused passes.
template <int N, int V>
__device__ __forceinline__ void load_scaled(const double* __restrict__ scaled, int64_t npad, fv<V>* out) {
#pragma unroll
  for (int32_t i = 0; i < N; ++i)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) out[i][v] = (float)scaled[(int64_t)i * npad + v];
}

template <int N, int V>
__device__ __forceinline__ void store_partials(double* __restrict__ mant, int32_t* __restrict__ exps, int64_t base,
                                               int64_t npad, double (&pm)[N][V], const ev_t<V>* pe) {
#pragma unroll
  for (int32_t s = 0; s < N; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      mant[base + (int64_t)s * npad + v] = frexp(pm[s][v], &de);
      exps[base + (int64_t)s * npad + v] = (int32_t)pe[s][v] + de;
    }
}

template <int D0, int D1, int S, int V, int MODE>
__global__ void __launch_bounds__(256) leaf_family_p2(float* __restrict__ arena, const float* __restrict__ cpts,
                                                     LeafFamilyArgs a) {
  constexpr bool FULL = MODE % 3 == kLeafFull;
  constexpr bool REDUCE = MODE % 3 == kLeafReduce;
  constexpr int32_t STATE = MODE / 3;
  constexpr int32_t kParts = 4 / V;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles * kParts) return;
  const int32_t part = (int32_t)(w % kParts);
  w /= kParts;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;
  const int32_t col = part * 64 * V + lane * V;
  const int32_t first = chunk * a.chunk_len;
  const int32_t last = first + a.chunk_len < a.K ? first + a.chunk_len : a.K;
  if (first >= last) return;

  double pm1[D0][V], pm2[D1][V];
  ev_t<V> pe1[D0], pe2[D1];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      pm1[i][v] = 1.0;
      pe1[i][v] = 0;
    }
#pragma unroll
    for (int32_t i = 0; i < D1; ++i) {
      pm2[i][v] = 1.0;
      pe2[i][v] = 0;
    }
  }
  fv<V> pf1[D0], pf2[D1];
  load_scaled<D0, V>(a.scaled + t * kTile + col, a.npad, pf1);
  load_scaled<D1, V>(a.scaled2 + t * kTile + col, a.npad, pf2);

  double ll_acc[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) ll_acc[v] = 0.0;

  for (int32_t k = first; k < last; ++k) {
    const float* cpt = cpts + a.cpt_off[k];
    float* f2v_p1 = arena + a.f_out[3 * k];
    float* f2v_p2 = arena + a.f_out[3 * k + 1];
    float* f2v_own = arena + a.f_out[3 * k + 2];
    float* m_state = a.m_state + (int64_t)k * (D0 + D1) * a.npad;
    float* z_state = a.z_state + (int64_t)k * a.npad;
    const float* obs = arena + a.obs_off[k];

    fv<V> ev[S], prev1[D0], prev2[D1], z;
#pragma unroll
    for (int32_t j = 0; j < S; ++j) ev[j] = loadv<V>(obs, t * S + j, col);
    if constexpr (STATE == kLeafSteady) {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) prev1[i] = loadv<V>(m_state, t * (D0 + D1) + i, col);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) prev2[i] = loadv<V>(m_state, t * (D0 + D1) + D0 + i, col);
      z = loadv<V>(z_state, t, col);
    } else {
      fv<V> own[S];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i)
        prev1[i] = STATE == kLeafFresh ? splat<V>(1.0f / (float)D0) : loadv<V>(f2v_p1, t * D0 + i, col);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i)
        prev2[i] = STATE == kLeafFresh ? splat<V>(1.0f / (float)D1) : loadv<V>(f2v_p2, t * D1 + i, col);
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
        own[j] = STATE == kLeafFresh ? splat<V>(1.0f / (float)S) : loadv<V>(f2v_own, t * S + j, col);
      z = leaf_normaliser<S, V>(ev, own);
    }

    // both parents' messages to this factor: (P'/previous message) normalised over the states
    fv<V> t1[D0], t2[D1];
    {
      fv<V> sum = splat<V>(0.f);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        t1[i] = pf1[i] * rcpv<V>(prev1[i]);
        sum += t1[i];
      }
      sum = rcpv<V>(sum);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) t1[i] *= sum;
      sum = splat<V>(0.f);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) {
        t2[i] = pf2[i] * rcpv<V>(prev2[i]);
        sum += t2[i];
      }
      sum = rcpv<V>(sum);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) t2[i] *= sum;
    }
    // the leaf's variable pass: c = log z, message to the factor = ev / z
    const fv<V> inv_z = recipv<V>(z);
    fv<V> o[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j) o[j] = ev[j] * inv_z;
    if (FULL || REDUCE) {
      fv<V> c;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        const float mz = frexpf(z[v], &de);
        c[v] = (float)((double)de * 0.6931471805599453 + (double)logf(mz));
        if (REDUCE) ll_acc[v] += (double)c[v];
      }
      if (FULL) storev<V>(arena + a.c_off[k], t, col, c);
    }
    if constexpr (REDUCE) {
      if (!a.em_partial) continue;  // only the log-likelihood sums are wanted
      // EM counts over the wave's observations (padding masked), wave-reduced
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
#pragma unroll
        for (int32_t v = 0; v < V; ++v)
          if (t * kTile + col + v >= a.N) o[j][v] = 0.f;
      float* dst = a.em_partial + ((int64_t)k * (D0 * D1 * S)) * a.slots + (t * kParts + part);
      float per_lane[D0 * D1 * S];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i)
#pragma unroll
        for (int32_t l = 0; l < D1; ++l) {
          const fv<V> tt = t1[i] * t2[l];
#pragma unroll
          for (int32_t j = 0; j < S; ++j) {
            const fv<V> prod = tt * o[j];
            float x = prod[0];
#pragma unroll
            for (int32_t v = 1; v < V; ++v) x += prod[v];
            per_lane[(i * D1 + l) * S + j] = x;
          }
        }
      WaveSums<D0 * D1 * S> sums;
      sums.reduce(per_lane, lane);
      if (lane < 4) {
#pragma unroll
        for (int32_t f = 0; f < WaveSums<D0 * D1 * S>::H2; ++f) {
          const int32_t e = WaveSums<D0 * D1 * S>::entry(lane, f);
          if (e < D0 * D1 * S) dst[(int64_t)e * a.slots] = sums.out[f];
        }
      }
      continue;
    }
    // contractions: A[i][l] = sum_j cpt[i][l][j] o[j];  m1 = A t2;  m2 = A^T t1;  own = cpt^T (t1 x t2)
    fv<V> m1[D0], m2[D1], own[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) m1[i] = splat<V>(0.f);
#pragma unroll
    for (int32_t l = 0; l < D1; ++l) m2[l] = splat<V>(0.f);
#pragma unroll
    for (int32_t j = 0; j < S; ++j) own[j] = splat<V>(0.f);
#pragma unroll
    for (int32_t i = 0; i < D0; ++i)
#pragma unroll
      for (int32_t l = 0; l < D1; ++l) {
        const float* row = cpt + (i * D1 + l) * S;
        fv<V> A = row[0] * o[0];
#pragma unroll
        for (int32_t j = 1; j < S; ++j) A = __builtin_elementwise_fma(splat<V>(row[j]), o[j], A);
        m1[i] = __builtin_elementwise_fma(A, t2[l], m1[i]);
        m2[l] = __builtin_elementwise_fma(A, t1[i], m2[l]);
        const fv<V> tt = t1[i] * t2[l];
#pragma unroll
        for (int32_t j = 0; j < S; ++j) own[j] = __builtin_elementwise_fma(splat<V>(row[j]), tt, own[j]);
      }
    if (FULL) {
      float* v2f_p1 = arena + a.f_in[3 * k];
      float* v2f_p2 = arena + a.f_in[3 * k + 1];
      float* v2f_own = arena + a.f_in[3 * k + 2];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        storev<V>(v2f_p1, t * D0 + i, col, t1[i]);
        storev<V>(f2v_p1, t * D0 + i, col, m1[i]);
      }
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) {
        storev<V>(v2f_p2, t * D1 + i, col, t2[i]);
        storev<V>(f2v_p2, t * D1 + i, col, m2[i]);
      }
#pragma unroll
      for (int32_t j = 0; j < S; ++j) {
        storev<V>(v2f_own, t * S + j, col, o[j]);
        storev<V>(f2v_own, t * S + j, col, own[j]);
      }
    } else {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) storev<V>(m_state, t * (D0 + D1) + i, col, m1[i]);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) storev<V>(m_state, t * (D0 + D1) + D0 + i, col, m2[i]);
      storev<V>(z_state, t, col, leaf_normaliser<S, V>(ev, own));
    }
    // fold into both parents' partial products
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) pm1[i][v] *= (double)m1[i][v];
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) pm2[i][v] *= (double)m2[i][v];
    }
    if (((k - first) & 3) == 3) {
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
#pragma unroll
        for (int32_t i = 0; i < D0; ++i) {
          int de = 0;
          pm1[i][v] = frexp(pm1[i][v], &de);
          pe1[i][v] += de;
        }
#pragma unroll
        for (int32_t i = 0; i < D1; ++i) {
          int de = 0;
          pm2[i][v] = frexp(pm2[i][v], &de);
          pe2[i][v] += de;
        }
      }
    }
  }
  if constexpr (REDUCE) {
    double* dst = a.ll_part + (int64_t)chunk * a.npad + t * kTile + col;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) dst[v] = ll_acc[v];
    return;
  }
  store_partials<D0, V>(a.part_mant, a.part_exp, (int64_t)chunk * D0 * a.npad + t * kTile + col, a.npad, pm1, pe1);
  store_partials<D1, V>(a.part_mant2, a.part_exp2, (int64_t)chunk * D1 * a.npad + t * kTile + col, a.npad, pm2, pe2);
}

// accum[entry] = cpt[entry] * sum over the per-wave sums written by the kLeafReduce kernel.
// One wave per CPT entry.   grid.x = ceil(entries / 4), block = 256
__global__ void __launch_bounds__(256) em_finalize_slots(const float* __restrict__ em_partial,
                                                        const float* __restrict__ cpts,
                                                        const int64_t* __restrict__ cpt_off, int32_t K,
                                                        int32_t entries_per_node, int32_t slots,
                                                        double* __restrict__ accum) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (item >= (int64_t)K * entries_per_node) return;
  const int32_t k = (int32_t)(item / entries_per_node);
  const int32_t e = (int32_t)(item % entries_per_node);
  const float* src = em_partial + item * slots;
  double acc = 0.0;
  for (int32_t sl = lane; sl < slots; sl += 64) acc += (double)src[sl];
#pragma unroll
  for (int32_t off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) accum[cpt_off[k] + e] = acc * (double)cpts[cpt_off[k] + e];
}

}  // namespace bn
