// Shape-specialised float32 kernels for the shapes that carry the throughput.
//
// Common structure: one wavefront owns one (node, tile) work item — 256 observations,
// four per lane — and moves every message row with one 16-byte load/store per lane
// (1 KiB per wave instruction, fully coalesced).  Cardinalities and fan-in are template
// parameters, so all per-lane vectors live in registers and every loop unrolls; CPT
// entries are wave-uniform and arrive through scalar loads.  All of these kernels stream
// each input once and each output once: they are HBM-bandwidth bound by construction.
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
constexpr int32_t kWavesPerBlock = 4;
constexpr int32_t kWideFanIn = 24;   // variable nodes with more inputs take the chunked path
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

// work item of this wave, or false if past the end.  items = K * ntiles, node-major.
__device__ __forceinline__ bool wave_item(int64_t items, int64_t ntiles, int32_t& k, int64_t& t, int32_t& lane) {
  const int32_t wave = threadIdx.x >> 6;
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
__global__ void __launch_bounds__(256) factor_root_fast(float* __restrict__ arena, const float* __restrict__ cpts,
                                                       const int64_t* __restrict__ out_off,
                                                       const int64_t* __restrict__ cpt_off, int32_t K, int32_t S,
                                                       int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  float* dst = arena + out_off[k];
  const float* cpt = cpts + cpt_off[k];
  for (int32_t s = 0; s < S; ++s) {
    const float c = cpt[s];
    f4 r{{c, c, c, c}};
    store4(dst, t * S + s, lane, r);
  }
}

// ---------------------------------------------------------------------------------
// Factor pass, one parent with D0 states, own variable with S states:
//   to parent   out[a] = sum_b cpt[a][b] * in_own[b]
//   to own var  out[b] = sum_a cpt[a][b] * in_parent[a]
// (the two einsums of FactorNodeGroup.calculate_outputs, spa_v3/factor_graph.py:251-258).
template <int D0, int S>
__global__ void __launch_bounds__(256) factor_p1_fast(float* __restrict__ arena, const float* __restrict__ cpts,
                                                     const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ out_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K,
                                                     int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const float* in_par = arena + in_off[2 * k];
  const float* in_own = arena + in_off[2 * k + 1];
  float* out_par = arena + out_off[2 * k];
  float* out_own = arena + out_off[2 * k + 1];
  const float* cpt = cpts + cpt_off[k];

  f4 a[D0], b[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) a[i] = load4(in_par, t * D0 + i, lane);
#pragma unroll
  for (int32_t j = 0; j < S; ++j) b[j] = load4(in_own, t * S + j, lane);

  f4 oa[D0], ob[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) oa[i].v[v] = 0.f;
#pragma unroll
  for (int32_t j = 0; j < S; ++j)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) ob[j].v[v] = 0.f;

#pragma unroll
  for (int32_t i = 0; i < D0; ++i) {
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      const float c = cpt[i * S + j];
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) {
        oa[i].v[v] = fmaf(c, b[j].v[v], oa[i].v[v]);
        ob[j].v[v] = fmaf(c, a[i].v[v], ob[j].v[v]);
      }
    }
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) store4(out_par, t * D0 + i, lane, oa[i]);
#pragma unroll
  for (int32_t j = 0; j < S; ++j) store4(out_own, t * S + j, lane, ob[j]);
}

// ---------------------------------------------------------------------------------
// Factor pass, two parents (D0, D1) and own variable (S): three contractions of one
// [D0, D1, S] CPT, each leaving one axis free.
template <int D0, int D1, int S>
__global__ void __launch_bounds__(256) factor_p2_fast(float* __restrict__ arena, const float* __restrict__ cpts,
                                                     const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ out_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K,
                                                     int64_t ntiles) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const float* in0 = arena + in_off[3 * k];
  const float* in1 = arena + in_off[3 * k + 1];
  const float* in2 = arena + in_off[3 * k + 2];
  const float* cpt = cpts + cpt_off[k];

  f4 a[D0], b[D1], c[S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) a[i] = load4(in0, t * D0 + i, lane);
#pragma unroll
  for (int32_t i = 0; i < D1; ++i) b[i] = load4(in1, t * D1 + i, lane);
#pragma unroll
  for (int32_t i = 0; i < S; ++i) c[i] = load4(in2, t * S + i, lane);

  f4 oa[D0], ob[D1], oc[S];
#pragma unroll
  for (int32_t v = 0; v < 4; ++v) {
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) oa[i].v[v] = 0.f;
#pragma unroll
    for (int32_t i = 0; i < D1; ++i) ob[i].v[v] = 0.f;
#pragma unroll
    for (int32_t i = 0; i < S; ++i) oc[i].v[v] = 0.f;
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < D1; ++j)
#pragma unroll
      for (int32_t l = 0; l < S; ++l) {
        const float w = cpt[(i * D1 + j) * S + l];
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) {
          oa[i].v[v] = fmaf(w, b[j].v[v] * c[l].v[v], oa[i].v[v]);
          ob[j].v[v] = fmaf(w, a[i].v[v] * c[l].v[v], ob[j].v[v]);
          oc[l].v[v] = fmaf(w, a[i].v[v] * b[j].v[v], oc[l].v[v]);
        }
      }
  float* o0 = arena + out_off[3 * k];
  float* o1 = arena + out_off[3 * k + 1];
  float* o2 = arena + out_off[3 * k + 2];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) store4(o0, t * D0 + i, lane, oa[i]);
#pragma unroll
  for (int32_t i = 0; i < D1; ++i) store4(o1, t * D1 + i, lane, ob[i]);
#pragma unroll
  for (int32_t i = 0; i < S; ++i) store4(o2, t * S + i, lane, oc[i]);
}

// ---------------------------------------------------------------------------------
// Variable pass for a small fan-in I (<= 4) and S states, all in registers.
// P[s] = prod_i in_i[s] fits a double without rescaling (at most four float factors),
// leave-one-out products are formed by multiplication, not division.
// Same quantities as variable_pass_generic.
template <int I, int S>
__global__ void __launch_bounds__(256) variable_small_fast(float* __restrict__ arena,
                                                          const int64_t* __restrict__ in_off,
                                                          const int64_t* __restrict__ out_off,
                                                          const int64_t* __restrict__ c_off, int32_t K,
                                                          int64_t ntiles, int32_t fresh) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const float uniform = 1.0f / (float)S;

  f4 in[I][S];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const bool skip = fresh && out_off[(int64_t)k * I + i] >= 0;
    const float* src = arena + in_off[(int64_t)k * I + i];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      if (skip) {
        in[i][s] = f4{{uniform, uniform, uniform, uniform}};
      } else {
        in[i][s] = load4(src, t * S + s, lane);
      }
    }
  }

  // leave-one-out products loo[i][s] = prod_{j != i} in_j[s], and the full product
  double loo[I][S][4];
  double z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
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
    f4 c;
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      int e = 0;
      const double m = frexp(z[v], &e);
      c.v[v] = (float)((double)e * 0.6931471805599453 + (double)logf((float)m));
    }
    store4(arena + c_off[k], t, lane, c);
  }

#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const int64_t dst = out_off[(int64_t)k * I + i];
    if (dst < 0) continue;
    double inv[4];
    if (i == I - 1) {
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) inv[v] = 1.0 / z[v];
    } else {
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) {
        double sum = 0.0;
#pragma unroll
        for (int32_t s = 0; s < S; ++s) sum += loo[i][s][v];
        inv[v] = 1.0 / sum;
      }
    }
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      f4 o;
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) o.v[v] = (float)(loo[i][s][v] * inv[v]);
      store4(arena + dst, t * S + s, lane, o);
    }
  }
}

// ---------------------------------------------------------------------------------
// Wide variable node (fan-in in the hundreds): three kernels.
//   wide_product : each wave multiplies a chunk of the inputs -> partial product per state,
//                  as double mantissa + int exponent            grid (chunks x tiles)
//   wide_combine : multiplies the partials and the remaining inputs, writes c, the message
//                  to the local factor, and P'[s] = P[s] / 2^emax as doubles  grid (tiles)
//   wide_emit    : out_i[s] = (P'[s]/in_i[s]) / sum_s'(P'[s']/in_i[s'])  grid (chunks x tiles)
// Chunked inputs are the first n_chunked = chunks * chunk_len input slots of the node (its
// children); the rest (at most a few: tail children, observation, local factor) are folded in
// by wide_combine.
struct WideArgs {
  const int64_t* in_off;   // [I] for this node
  const int64_t* out_off;  // [I]
  int64_t c_off;
  int32_t I;
  int32_t chunks, chunk_len;
  double* part_mant;  // [chunks][S][npad]
  int32_t* part_exp;  // [chunks][S][npad]
  double* scaled;     // [S][npad]   P'[s]
  int64_t npad;
  int64_t ntiles;
  int32_t fresh;
};

template <int S>
__global__ void __launch_bounds__(256) wide_product_fast(const float* __restrict__ arena, WideArgs a) {
  const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
  if (!a.fresh) {
    for (int32_t j = 0; j < a.chunk_len; ++j) {
      const float* src = arena + a.in_off[first + j];
#pragma unroll
      for (int32_t s = 0; s < S; ++s) {
        const f4 x = load4(src, t * S + s, lane);
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) m[s][v] *= (double)x.v[v];
      }
      if ((j & 3) == 3 || j == a.chunk_len - 1) {
#pragma unroll
        for (int32_t s = 0; s < S; ++s)
#pragma unroll
          for (int32_t v = 0; v < 4; ++v) {
            int de = 0;
            m[s][v] = frexp(m[s][v], &de);
            e[s][v] += de;
          }
      }
    }
  } else {
    // all chunked inputs still hold 1/S
    const double u = 1.0 / (double)S;
    for (int32_t j = 0; j < a.chunk_len; ++j) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s)
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) m[s][v] *= u;
      if ((j & 3) == 3 || j == a.chunk_len - 1) {
#pragma unroll
        for (int32_t s = 0; s < S; ++s)
#pragma unroll
          for (int32_t v = 0; v < 4; ++v) {
            int de = 0;
            m[s][v] = frexp(m[s][v], &de);
            e[s][v] += de;
          }
      }
    }
  }
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    const int64_t row = ((int64_t)chunk * S + s) * a.npad + t * kTile + lane * 4;
    *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(m[s][0], m[s][1]);
    *reinterpret_cast<double2*>(a.part_mant + row + 2) = make_double2(m[s][2], m[s][3]);
    *reinterpret_cast<int4*>(a.part_exp + row) = make_int4(e[s][0], e[s][1], e[s][2], e[s][3]);
  }
}

template <int S>
__global__ void __launch_bounds__(256) wide_combine_fast(float* __restrict__ arena, WideArgs a) {
  const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (t >= a.ntiles) return;

  double m[S][4];
  int32_t e[S][4];
#pragma unroll
  for (int32_t s = 0; s < S; ++s)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      m[s][v] = 1.0;
      e[s][v] = 0;
    }
  for (int32_t ch = 0; ch < a.chunks; ++ch) {
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const int64_t row = ((int64_t)ch * S + s) * a.npad + t * kTile + lane * 4;
      const double2 lo = *reinterpret_cast<const double2*>(a.part_mant + row);
      const double2 hi = *reinterpret_cast<const double2*>(a.part_mant + row + 2);
      const int4 ex = *reinterpret_cast<const int4*>(a.part_exp + row);
      m[s][0] *= lo.x;
      m[s][1] *= lo.y;
      m[s][2] *= hi.x;
      m[s][3] *= hi.y;
      e[s][0] += ex.x;
      e[s][1] += ex.y;
      e[s][2] += ex.z;
      e[s][3] += ex.w;
    }
    if ((ch & 3) == 3) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s)
#pragma unroll
        for (int32_t v = 0; v < 4; ++v) {
          int de = 0;
          m[s][v] = frexp(m[s][v], &de);
          e[s][v] += de;
        }
    }
  }
  // the inputs that were not chunked
  const int32_t n_chunked = a.chunks * a.chunk_len;
  f4 local[S];
  const float uniform = 1.0f / (float)S;
  for (int32_t i = n_chunked; i < a.I; ++i) {
    const bool skip = a.fresh && a.out_off[i] >= 0;
    const float* src = arena + a.in_off[i];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      f4 x;
      if (skip) {
        x = f4{{uniform, uniform, uniform, uniform}};
      } else {
        x = load4(src, t * S + s, lane);
      }
      if (i == a.I - 1) local[s] = x;
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) {
        int de = 0;
        m[s][v] = frexp(m[s][v] * (double)x.v[v], &de);
        e[s][v] += de;
      }
    }
  }

  int32_t emax[4];
  double z[4];
#pragma unroll
  for (int32_t v = 0; v < 4; ++v) {
    emax[v] = kDeadExponent;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      int de = 0;
      m[s][v] = frexp(m[s][v], &de);
      e[s][v] = m[s][v] == 0.0 ? kDeadExponent : e[s][v] + de;
      emax[v] = e[s][v] > emax[v] ? e[s][v] : emax[v];
    }
    z[v] = 0.0;
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      m[s][v] = ldexp(m[s][v], clamp_shift(e[s][v] - emax[v]));  // P'[s]
      z[v] += m[s][v];
    }
  }
  {
    f4 c;
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) c.v[v] = (float)((double)emax[v] * 0.6931471805599453 + log(z[v]));
    store4(arena + a.c_off, t, lane, c);
  }
  // message to the local factor: (P'/in_local) / z
  {
    float* dst = arena + a.out_off[a.I - 1];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      f4 o;
#pragma unroll
      for (int32_t v = 0; v < 4; ++v) o.v[v] = (float)(m[s][v] / (double)local[s].v[v] / z[v]);
      store4(dst, t * S + s, lane, o);
    }
  }
#pragma unroll
  for (int32_t s = 0; s < S; ++s) {
    const int64_t row = (int64_t)s * a.npad + t * kTile + lane * 4;
    *reinterpret_cast<double2*>(a.scaled + row) = make_double2(m[s][0], m[s][1]);
    *reinterpret_cast<double2*>(a.scaled + row + 2) = make_double2(m[s][2], m[s][3]);
  }
}

template <int S>
__global__ void __launch_bounds__(256) wide_emit_fast(float* __restrict__ arena, WideArgs a, int32_t emit_chunks,
                                                     int32_t emit_len) {
  const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)emit_chunks * a.ntiles) return;
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
  const int32_t first = chunk * emit_len;
  const int32_t last = first + emit_len < a.I - 1 ? first + emit_len : a.I - 1;  // local slot excluded
  const float uniform = 1.0f / (float)S;
  for (int32_t i = first; i < last; ++i) {
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
// EM expected counts, one parent (D0) + own variable (S), or no parent (D0 = 1, the
// parent message is the constant 1): per lane running sums of in_parent[a] * in_own[b]
// over the observations of the wave's tile range, then a wave reduction.  The CPT factor
// and the sum over chunks are applied by em_finalize.
template <int D0, int S, bool HAS_PARENT>
__global__ void __launch_bounds__(256) em_p1_fast(const float* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                 const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                 int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)K * nchunks) return;
  const int32_t k = (int32_t)(w / nchunks);
  const int32_t chunk = (int32_t)(w % nchunks);
  constexpr int32_t A = HAS_PARENT ? 2 : 1;
  const float* in_par = HAS_PARENT ? arena + in_off[(int64_t)k * A] : nullptr;
  const float* in_own = arena + in_off[(int64_t)k * A + (A - 1)];

  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  float acc[D0][S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) acc[i][j] = 0.f;

  for (int64_t t = t0; t < t1; ++t) {
    f4 a[D0], b[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      if (HAS_PARENT) {
        a[i] = load4(in_par, t * D0 + i, lane);
      } else {
        a[i] = f4{{1.f, 1.f, 1.f, 1.f}};
      }
    }
#pragma unroll
    for (int32_t j = 0; j < S; ++j) b[j] = load4(in_own, t * S + j, lane);
    const int64_t n0 = t * kTile + lane * 4;
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      const float live = (n0 + v < N) ? 1.f : 0.f;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        const float av = a[i].v[v] * live;
#pragma unroll
        for (int32_t j = 0; j < S; ++j) acc[i][j] = fmaf(av, b[j].v[v], acc[i][j]);
      }
    }
  }
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      double x = (double)acc[i][j];
#pragma unroll
      for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0) partial[(cpt_off[k] + i * S + j) * nchunks + chunk] = x;
    }
}

template <int D0, int D1, int S>
__global__ void __launch_bounds__(256) em_p2_fast(const float* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                 const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                 int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)K * nchunks) return;
  const int32_t k = (int32_t)(w / nchunks);
  const int32_t chunk = (int32_t)(w % nchunks);
  const float* in0 = arena + in_off[(int64_t)k * 3];
  const float* in1 = arena + in_off[(int64_t)k * 3 + 1];
  const float* in2 = arena + in_off[(int64_t)k * 3 + 2];

  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  float acc[D0][D1][S];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i)
#pragma unroll
    for (int32_t j = 0; j < D1; ++j)
#pragma unroll
      for (int32_t l = 0; l < S; ++l) acc[i][j][l] = 0.f;

  for (int64_t t = t0; t < t1; ++t) {
    f4 a[D0], b[D1], c[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) a[i] = load4(in0, t * D0 + i, lane);
#pragma unroll
    for (int32_t i = 0; i < D1; ++i) b[i] = load4(in1, t * D1 + i, lane);
#pragma unroll
    for (int32_t i = 0; i < S; ++i) c[i] = load4(in2, t * S + i, lane);
    const int64_t n0 = t * kTile + lane * 4;
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      const float live = (n0 + v < N) ? 1.f : 0.f;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        const float av = a[i].v[v] * live;
#pragma unroll
        for (int32_t j = 0; j < D1; ++j) {
          const float ab = av * b[j].v[v];
#pragma unroll
          for (int32_t l = 0; l < S; ++l) acc[i][j][l] = fmaf(ab, c[l].v[v], acc[i][j][l]);
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
        double x = (double)acc[i][j][l];
#pragma unroll
        for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) partial[(cpt_off[k] + (i * D1 + j) * S + l) * nchunks + chunk] = x;
      }
}

// =================================================================================
// Host side: dispatch tables
// =================================================================================

#define BN_SMALL_DIMS(X) X(2) X(3) X(4) X(5) X(10)

using FactorP1Fn = void (*)(float*, const float*, const int64_t*, const int64_t*, const int64_t*, int32_t, int64_t);
using EmP1Fn = void (*)(const float*, const int64_t*, const int64_t*, int32_t, int64_t, int64_t, int32_t, double*);
using VarSmallFn = void (*)(float*, const int64_t*, const int64_t*, const int64_t*, int32_t, int64_t, int32_t);

template <int D0>
inline FactorP1Fn factor_p1_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return factor_p1_fast<D0, SV>;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

inline FactorP1Fn factor_p1_lookup(int32_t d0, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return factor_p1_for_s<DV>(s);
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

template <int D0>
inline EmP1Fn em_p1_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return em_p1_fast<D0, SV, true>;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

inline EmP1Fn em_p1_lookup(int32_t d0, int32_t s) {
  switch (d0) {
#define X(DV) \
  case DV:    \
    return em_p1_for_s<DV>(s);
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

inline EmP1Fn em_root_lookup(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return em_p1_fast<1, SV, false>;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

template <int I>
inline VarSmallFn var_small_for_s(int32_t s) {
  switch (s) {
#define X(SV) \
  case SV:    \
    return variable_small_fast<I, SV>;
    BN_SMALL_DIMS(X)
#undef X
  }
  return nullptr;
}

inline VarSmallFn var_small_lookup(int32_t fan_in, int32_t s) {
  switch (fan_in) {
    case 1:
      return var_small_for_s<1>(s);
    case 2:
      return var_small_for_s<2>(s);
    case 3:
      return var_small_for_s<3>(s);
    case 4:
      return var_small_for_s<4>(s);
  }
  return nullptr;
}

inline uint32_t wave_blocks(int64_t items) { return (uint32_t)((items + kWavesPerBlock - 1) / kWavesPerBlock); }

inline bool launch_ok() { return hipGetLastError() == hipSuccess; }

// ---- wide variable nodes: scratch layout --------------------------------------------
struct WidePlan {
  int32_t chunks = 0, chunk_len = 0;      // product chunks over the first chunks*chunk_len inputs
  int32_t emit_chunks = 0, emit_len = 0;  // emission chunks over inputs [0, I-1)
  double* part_mant = nullptr;
  int32_t* part_exp = nullptr;
  double* scaled = nullptr;
};

inline bool is_wide(const Plan& p, const VariableGroup& g) {
  return p.dtype == 0 && g.num_inputs > kWideFanIn && g.num_states <= kWideMaxStates;
}

inline void wide_geometry(const Plan& p, const VariableGroup& g, WidePlan& w) {
  // aim for ~4096 waves per launch, chunks of at least 8 inputs
  const int32_t I = g.num_inputs;
  int64_t want = (4096 + p.ntiles - 1) / p.ntiles;
  int32_t chunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(want, I / 8));
  w.chunk_len = (I - 2) / chunks;  // the last two slots (observation or last child, local factor) are never chunked
  w.chunks = chunks;
  w.emit_len = w.chunk_len;
  w.emit_chunks = ((I - 1) + w.emit_len - 1) / w.emit_len;
}

}  // namespace bn

// The per-plan wide scratch lives in a side table keyed by (plan, group index) so that plan.h
// stays free of kernel-specific state.
#include <map>

namespace bn {

struct FastState {
  std::map<int32_t, WidePlan> wide;  // variable group index -> scratch
};

inline std::map<const Plan*, FastState>& fast_states() {
  static std::map<const Plan*, FastState> states;
  return states;
}

inline int64_t fast_scratch_elems(const Plan&) { return 0; }

inline int prepare_fast(Plan& p) {
  FastState st;
  for (size_t gi = 0; gi < p.vgroups.size(); ++gi) {
    const VariableGroup& g = p.vgroups[gi];
    if (!is_wide(p, g)) continue;
    WidePlan w;
    wide_geometry(p, g, w);
    const size_t S = g.num_states;
    const size_t nodes = g.nodes.size();
    // scratch is reused node after node within the group (launches are stream-ordered)
    (void)nodes;
    if (hipMalloc(reinterpret_cast<void**>(&w.part_mant), (size_t)w.chunks * S * p.npad * sizeof(double)) != hipSuccess)
      return -2;
    if (hipMalloc(reinterpret_cast<void**>(&w.part_exp), (size_t)w.chunks * S * p.npad * sizeof(int32_t)) != hipSuccess)
      return -2;
    if (hipMalloc(reinterpret_cast<void**>(&w.scaled), S * p.npad * sizeof(double)) != hipSuccess) return -2;
    p.device_bytes += (int64_t)((size_t)w.chunks * S * p.npad * 12 + S * p.npad * 8);
    st.wide[(int32_t)gi] = w;
  }
  fast_states()[&p] = st;
  return 0;
}

inline void release_fast(Plan& p) {
  auto it = fast_states().find(&p);
  if (it == fast_states().end()) return;
  for (auto& kv : it->second.wide) {
    if (kv.second.part_mant) (void)hipFree(kv.second.part_mant);
    if (kv.second.part_exp) (void)hipFree(kv.second.part_exp);
    if (kv.second.scaled) (void)hipFree(kv.second.scaled);
  }
  fast_states().erase(it);
}

template <int S>
inline bool launch_wide_node(const Plan& p, const VariableGroup& g, const WidePlan& w, size_t k, float* arena,
                             int32_t fresh, hipStream_t s) {
  WideArgs a;
  a.in_off = g.d_in + k * g.num_inputs;
  a.out_off = g.d_out + k * g.num_inputs;
  a.c_off = g.c_off[k];
  a.I = g.num_inputs;
  a.chunks = w.chunks;
  a.chunk_len = w.chunk_len;
  a.part_mant = w.part_mant;
  a.part_exp = w.part_exp;
  a.scaled = w.scaled;
  a.npad = p.npad;
  a.ntiles = p.ntiles;
  a.fresh = fresh;
  hipLaunchKernelGGL(wide_product_fast<S>, dim3(wave_blocks((int64_t)w.chunks * p.ntiles)), dim3(256), 0, s, arena, a);
  if (!launch_ok()) return false;
  hipLaunchKernelGGL(wide_combine_fast<S>, dim3(wave_blocks(p.ntiles)), dim3(256), 0, s, arena, a);
  if (!launch_ok()) return false;
  hipLaunchKernelGGL(wide_emit_fast<S>, dim3(wave_blocks((int64_t)w.emit_chunks * p.ntiles)), dim3(256), 0, s, arena, a,
                     w.emit_chunks, w.emit_len);
  return launch_ok();
}

template <typename T>
int launch_variable_fast(Plan& p, VariableGroup& g, T* arena_t, int32_t fresh, hipStream_t s) {
  if constexpr (!std::is_same<T, float>::value) {
    return 0;
  } else {
    float* arena = arena_t;
    const int32_t K = (int32_t)g.nodes.size();
    if (is_wide(p, g)) {
      auto it = fast_states().find(&p);
      if (it == fast_states().end()) return 0;
      auto wit = it->second.wide.find((int32_t)(&g - p.vgroups.data()));
      if (wit == it->second.wide.end()) return 0;
      for (size_t k = 0; k < g.nodes.size(); ++k) {
        bool ok = false;
        switch (g.num_states) {
#define X(SV)                                                    \
  case SV:                                                       \
    ok = launch_wide_node<SV>(p, g, wit->second, k, arena, fresh, s); \
    break;
          BN_SMALL_DIMS(X)
#undef X
          default:
            return k == 0 ? 0 : -1;
        }
        if (!ok) return -1;
      }
      return 1;
    }
    if (g.num_inputs * g.num_states > 12) return 0;  // per-lane state would not fit the register budget
    VarSmallFn fn = var_small_lookup(g.num_inputs, g.num_states);
    if (!fn) return 0;
    hipLaunchKernelGGL(fn, dim3(wave_blocks((int64_t)K * p.ntiles)), dim3(256), 0, s, arena, g.d_in, g.d_out, g.d_c, K,
                       p.ntiles, fresh);
    return launch_ok() ? 1 : -1;
  }
}

template <typename T>
int launch_factor_fast(Plan& p, FactorGroup& g, T* arena_t, const T* cpts_t, hipStream_t s) {
  if constexpr (!std::is_same<T, float>::value) {
    return 0;
  } else {
    float* arena = arena_t;
    const float* cpts = cpts_t;
    const int32_t K = (int32_t)g.nodes.size();
    const uint32_t blocks = wave_blocks((int64_t)K * p.ntiles);
    if (g.dims.size() == 1) {
      hipLaunchKernelGGL(factor_root_fast, dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_out, g.d_cpt, K, g.dims[0],
                         p.ntiles);
      return launch_ok() ? 1 : -1;
    }
    if (g.dims.size() == 2) {
      FactorP1Fn fn = factor_p1_lookup(g.dims[0], g.dims[1]);
      if (!fn) return 0;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt, K, p.ntiles);
      return launch_ok() ? 1 : -1;
    }
    if (g.dims.size() == 3 && g.dims[0] == 2 && g.dims[1] == 2 && g.dims[2] == 2) {
      hipLaunchKernelGGL((factor_p2_fast<2, 2, 2>), dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt,
                         K, p.ntiles);
      return launch_ok() ? 1 : -1;
    }
    return 0;
  }
}

template <typename T>
int launch_em_fast(Plan& p, FactorGroup& g, const T* arena_t, int32_t nchunks, double* partial, hipStream_t s) {
  if constexpr (!std::is_same<T, float>::value) {
    return 0;
  } else {
    const float* arena = arena_t;
    const int32_t K = (int32_t)g.nodes.size();
    const uint32_t blocks = wave_blocks((int64_t)K * nchunks);
    if (g.dims.size() == 1) {
      EmP1Fn fn = em_root_lookup(g.dims[0]);
      if (!fn) return 0;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N, p.ntiles, nchunks, partial);
      return launch_ok() ? 1 : -1;
    }
    if (g.dims.size() == 2) {
      EmP1Fn fn = em_p1_lookup(g.dims[0], g.dims[1]);
      if (!fn) return 0;
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N, p.ntiles, nchunks, partial);
      return launch_ok() ? 1 : -1;
    }
    if (g.dims.size() == 3 && g.dims[0] == 2 && g.dims[1] == 2 && g.dims[2] == 2) {
      hipLaunchKernelGGL((em_p2_fast<2, 2, 2>), dim3(blocks), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N, p.ntiles,
                         nchunks, partial);
      return launch_ok() ? 1 : -1;
    }
    return 0;
  }
}

}  // namespace bn
