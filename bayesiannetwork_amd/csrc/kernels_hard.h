// The leaf-family launches of a whole EM step (bn_em_step, bn_em_step_statistics) whose evidence is
// compact hard evidence - one uint8 state per observed two-state leaf and observation - read where
// the caller holds it, in every launch (nothing is copied: the plan is reset when the step ends).
//
// The evidence of a leaf can take three forms only (state 0, state 1, any other state: both rows at
// the off value), so everything that depends on it and on the family's CPT alone exists in three
// versions per family: q[y][d] = cpt[d] . ev(y), 1 / q[y][d] and log2 q[y][d].  hard_tables_build
// writes these once per step (the parameters do not change inside one), q and 1 / q with the very
// operations the float kernels apply to the expanded evidence.  No product with the evidence and
// no reciprocal per state is left in the family loops, and a family's evidence is 2 bytes per lane
// instead of 16:
//   * a dot product with a family's row is computed for both common forms at once (the rows are the
//     same for all lanes: two fused multiply-adds per state serve a lane's two observations) and
//     every observation keeps the result of its own form;
//   * the per-state product of the q over a chunk's families is carried as a sum of logarithms in
//     double, one fused multiply-add per family and state (hard_loop_first);
//   * the count sums factor into sums by evidence form, which are matrix products over the
//     observations and run on the matrix cores (hard_reduce_mfma).
// A block's four waves work on the same chunk of families (each on its own 128 observations) and
// stage the chunk's rows in LDS once; every wave then reads a family's rows from there as it goes.
//
//   hard_loop_first   the call's first iteration on fresh messages: what leaf_loop_p1 does in its
//                     fresh / ingesting state (prod q per state, prod z of iterations 1 and 2)
//   hard_loop_light   a later loop iteration: prod z of the next iteration from the parent's P'
//   hard_reduce_mfma  the last iteration of a statistics call, with the normalisers of the iteration
//                     before recomputed (leaf_family_p1 in reduce mode, kLeafRecompute)
// Float plans, two-state leaves, parents of at most 16 states, two observations per lane, the
// leaves' state rows at equal distances in memory (LeafFamilyArgs::src_base / src_stride).
#pragma once

#include "kernels_leaf.h"
#include "kernels_leaf_loop.h"

namespace bn {

// floats of a group's records (kHardRows rows of D0 per family: kernels_leaf.h)
constexpr size_t hard_table_floats(int32_t K, int32_t d0) { return (size_t)K * kHardRows * d0; }
// ... and of everything a step keeps per group: the records, then the third-form sums [K][2][D0]
constexpr size_t hard_state_floats(int32_t K, int32_t d0) { return hard_table_floats(K, d0) + (size_t)K * 2 * d0; }

typedef const __attribute__((address_space(4))) float* const_float_ptr;  // read through the scalar cache
typedef const __attribute__((address_space(4))) double* const_double_ptr;

// blocks of a launch: every chunk takes ceil(waves per chunk / 4) of them
inline uint32_t hard_blocks(int32_t chunks, int64_t ntiles) {
  return (uint32_t)(chunks * ((ntiles * 2 + kWavesPerBlock - 1) / kWavesPerBlock));
}

// Where a wave works: the block's chunk of families, the wave's 128 observations
template <int D0>
struct HardWave {
  int32_t lane, wave, chunk, part, first, last, col;
  int64_t t, obs0;
  bool active;            // false: the block's chunk has fewer waves than the block (the wave only helps staging)
  const uint8_t* states;  // + k * stride: the row of leaf k; this lane's pair sits at `at`
  int64_t stride, at;

  __device__ __forceinline__ void place(const LeafFamilyArgs<float>& a, int32_t chunk_len) {
    constexpr int32_t V = 2;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lane = threadIdx.x & 63;
    const int32_t per_chunk = (int32_t)(a.ntiles * 2);
    const int32_t blocks_per_chunk = (per_chunk + kWavesPerBlock - 1) / kWavesPerBlock;
    chunk = (int32_t)blockIdx.x / blocks_per_chunk;
    int32_t item = ((int32_t)blockIdx.x % blocks_per_chunk) * kWavesPerBlock + wave;
    active = item < per_chunk;
    item = active ? item : per_chunk - 1;
    t = item >> 1;
    part = item & 1;
    col = part * 64 * V + lane * V;
    obs0 = t * 256 + col;
    first = chunk * chunk_len;
    last = first + chunk_len < a.K ? first + chunk_len : a.K;
    states = a.src_base;
    stride = a.src_stride;
    // a lane's two observations are two consecutive bytes (N is even); lanes past the end read the
    // last real pair (their results are finite and masked out of every sum)
    at = obs0 + 1 >= a.N ? a.N - 2 : obs0;
  }
  // the two state bytes of this lane's observations of leaf k
  __device__ __forceinline__ uint32_t pair(int32_t k) const {
    return *reinterpret_cast<const uint16_t*>(states + (int64_t)k * stride + at);
  }
};

// What the count sums of a family are made of when its evidence is hard: with
//   G[y][d] = sum over the observations n of form y of  P'[d][n] * scale[n]
// (hard_reduce_mfma delivers G[0] and G[1] where the float kernels deliver one sum per CPT entry)
//   counts[d][j] = cpt[d][j] * (ev(0)[j] * G[0][d] / q[0][d] + ev(1)[j] * G[1][d] / q[1][d])
// (hard_count).  An observation of the third form (both rows off) would need a G[2] of its own;
// instead it is added to G[0] and G[1] with the two weights per (family, d) that make the formula
// come out at its contribution off * G[2][d] / q[2][d] to both j (rows 6 and 7 of the record).

// the records of all families of the group, the operations of the float kernels: q = cpt[d][0] * ev0,
// then fma(cpt[d][1], ev1, q); 1 / q by the hardware reciprocal
template <int D0>
__global__ void __launch_bounds__(256) hard_tables_build(const float* __restrict__ cpts, LeafFamilyArgs<float> a, float* tab) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.K * D0) return;
  const int32_t k = (int32_t)(e / D0), d = (int32_t)(e % D0);
  const float* cpt = cpts + a.cpt_off[k];
  float* rec = tab + (size_t)k * (kHardRows * D0) + d;
  float rq[kHardForms], qs[kHardForms];
#pragma unroll
  for (int32_t y = 0; y < kHardForms; ++y) {
    float q = cpt[d * 2] * hard_ev(y, 0, a.on_value, a.off_value);
    q = fmaf(cpt[d * 2 + 1], hard_ev(y, 1, a.on_value, a.off_value), q);
    qs[y] = q;
    rq[y] = fast_rcp(q);
    rec[(kHardQ + y) * D0] = q;
    rec[(kHardRq + y) * D0] = rq[y];
  }
  // x0, x1 with  rq0 * on * x0 + rq1 * off * x1 = rq0 * off * x0 + rq1 * on * x1 = rq2 * off
  const double on = a.on_value, off = a.off_value;
  const double target = (double)rq[2] * off;
  const bool same = on == off;
  rec[(kHardThird + 0) * D0] = (float)(same ? 1.0 : target / ((double)rq[0] * (on + off)));
  rec[(kHardThird + 1) * D0] = (float)(same ? 0.0 : target / ((double)rq[1] * (on + off)));
  double* logs = reinterpret_cast<double*>(tab + (size_t)k * (kHardRows * D0) + (size_t)kHardLog * D0) + d;
  double lg[kHardForms];
#pragma unroll
  for (int32_t y = 0; y < kHardForms; ++y) lg[y] = qs[y] > 0.f ? log2((double)qs[y]) : kHardLogOfZero;
  logs[0] = lg[0];
  logs[D0] = lg[1] - lg[0];
  logs[2 * D0] = lg[2] - lg[0];
  // sums of the third-form observations ([K][2][D0], behind the records)
  float* third = tab + hard_table_floats(a.K, D0) + (size_t)k * (2 * D0) + d;
  third[0] = 0.f;
  third[D0] = 0.f;
}

// The rows a launch reads per family, staged in LDS by the whole block: per family `kStride` floats,
// the record's floats [from, from + count) first and - hard_loop_first - the doubles' second row after
template <int D0, bool WITH_LOG>
struct HardStage {
  static constexpr int32_t kFloats = 2 * D0;                              // two rows of one kind (forms 0 and 1)
  static constexpr int32_t kStride = ((WITH_LOG ? 4 : 2) * D0 + 3) / 4 * 4;  // 16-byte steps
  static constexpr size_t bytes(int32_t chunk_len) { return (size_t)chunk_len * kStride * sizeof(float); }

  // call with all threads of the block; from_row: kHardQ or kHardRq
  __device__ __forceinline__ static void fill(const float* tab, int32_t first, int32_t last, int32_t from_row, float* lds) {
    constexpr int32_t kPer = (WITH_LOG ? 4 : 2) * D0;
    const int32_t total = (last - first) * kPer;
    for (int32_t e = (int32_t)threadIdx.x; e < total; e += 256) {
      const int32_t kl = e / kPer, x = e - kl * kPer;
      const float* rec = tab + (size_t)(first + kl) * (kHardRows * D0);
      // the second row of doubles (log2 q[1] - log2 q[0]) lies 2 * D0 floats behind the first
      lds[kl * kStride + x] = x < kFloats ? rec[from_row * D0 + x] : rec[(kHardLog + 2) * D0 + (x - kFloats)];
    }
    __syncthreads();
  }
};

// one row of a record, the same for all lanes
template <int D0>
struct HardRow {
  float x[D0];
  // family k's row straight from the record, through the scalar cache (the rare forms)
  __device__ __forceinline__ void load(const float* tab, int32_t k, int32_t row) {
    const_float_ptr src = (const_float_ptr)(tab + (size_t)k * (kHardRows * D0) + (size_t)row * D0);
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) x[d] = src[d];
  }
};

// the two staged rows of a family (forms 0 and 1)
template <int D0>
struct HardRows {
  HardRow<D0> r0, r1;
  __device__ __forceinline__ void load(const float* staged) {
    const float* p = static_cast<const float*>(__builtin_assume_aligned(staged, 16));
    float both[2 * D0];
#pragma unroll
    for (int32_t i = 0; i < 2 * D0; ++i) both[i] = p[i];
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      r0.x[d] = both[d];
      r1.x[d] = both[D0 + d];
    }
  }
};

// one of the record's rows of doubles (0: log2 q[0], 1, 2: the other forms' difference to it)
template <int D0>
struct HardLogRow {
  double x[D0];
  __device__ __forceinline__ void load(const float* tab, int32_t k, int32_t row) {
    const_double_ptr src = (const_double_ptr)(tab + (size_t)k * (kHardRows * D0) + (size_t)kHardLog * D0) + (size_t)row * D0;
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) x[d] = src[d];
  }
  __device__ __forceinline__ void load_staged(const float* staged) {
    const double* p = static_cast<const double*>(__builtin_assume_aligned(staged, 8));
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) x[d] = p[d];
  }
};

// 2^total as a mantissa in [0.5, 1) and an exponent
__device__ __forceinline__ double hard_exp2_split(double total, int32_t* exponent) {
  const double whole = floor(total) + 1.0;
  *exponent = (int32_t)whole;
  return exp2(total - whole);
}

// sum_d x[d] * row[d], the chain of fused multiply-adds the float kernels run per observation
template <int D0>
__device__ __forceinline__ fv<float, 2> hard_dot(const fv<float, 2>* x, const HardRow<D0>& row) {
  fv<float, 2> acc = splat<float, 2>(0.f);
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) acc = __builtin_elementwise_fma(x[d], splat<float, 2>(row.x[d]), acc);
  return acc;
}

// Which form do this lane's two observations have?  one: state 1; third: any state beyond
struct HardForms {
  bool one[2], third[2];
  __device__ __forceinline__ explicit HardForms(uint32_t pair) {
    one[0] = (pair & 0xffu) == 1u;
    one[1] = (pair >> 8) == 1u;
    third[0] = (pair & 0xfeu) != 0;
    third[1] = (pair & 0xfe00u) != 0;
  }
  __device__ __forceinline__ bool any_third() const { return __any((int)(third[0] || third[1])) != 0; }
};

// the dot product with the row of each observation's own form: both common forms are computed and
// every observation keeps its own; third_row: the record's row of the third form
template <int D0>
__device__ __forceinline__ fv<float, 2> hard_dot_own(const fv<float, 2>* x, const float* tab, int32_t k, int32_t third_row,
                                                     const HardRows<D0>& rows, const HardForms& f, bool any_third) {
  const fv<float, 2> s0 = hard_dot<D0>(x, rows.r0), s1 = hard_dot<D0>(x, rows.r1);
  fv<float, 2> out;
  out[0] = f.one[0] ? s1[0] : s0[0];
  out[1] = f.one[1] ? s1[1] : s0[1];
  if (any_third) {
    HardRow<D0> r2;
    r2.load(tab, k, third_row);
    const fv<float, 2> s2 = hard_dot<D0>(x, r2);
    out[0] = f.third[0] ? s2[0] : out[0];
    out[1] = f.third[1] ? s2[1] : out[1];
  }
  return out;
}

// hard_dot_own for two families at once: four independent chains of fused multiply-adds, so that no
// instruction has to wait for the one before it (a packed multiply-add that depends on its
// predecessor costs a wait state)
template <int D0>
__device__ __forceinline__ void hard_dot_own2(const fv<float, 2>* x, const float* tab, int32_t k_a, int32_t k_b, int32_t third_row,
                                              const HardRows<D0>& ra, const HardRows<D0>& rb, const HardForms& fa,
                                              const HardForms& fb, bool any_third, fv<float, 2>* out_a, fv<float, 2>* out_b) {
  fv<float, 2> a0 = splat<float, 2>(0.f), a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) {
    a0 = __builtin_elementwise_fma(x[d], splat<float, 2>(ra.r0.x[d]), a0);
    b0 = __builtin_elementwise_fma(x[d], splat<float, 2>(rb.r0.x[d]), b0);
    a1 = __builtin_elementwise_fma(x[d], splat<float, 2>(ra.r1.x[d]), a1);
    b1 = __builtin_elementwise_fma(x[d], splat<float, 2>(rb.r1.x[d]), b1);
  }
#pragma unroll
  for (int32_t v = 0; v < 2; ++v) {
    (*out_a)[v] = fa.one[v] ? a1[v] : a0[v];
    (*out_b)[v] = fb.one[v] ? b1[v] : b0[v];
  }
  if (any_third) {
    HardRow<D0> r2a, r2b;
    r2a.load(tab, k_a, third_row);
    r2b.load(tab, k_b, third_row);
    const fv<float, 2> sa = hard_dot<D0>(x, r2a), sb = hard_dot<D0>(x, r2b);
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      (*out_a)[v] = fa.third[v] ? sa[v] : (*out_a)[v];
      (*out_b)[v] = fb.third[v] ? sb[v] : (*out_b)[v];
    }
  }
}

template <int D0>
__device__ __forceinline__ void hard_load_scaled(const double* scaled, int64_t npad, int64_t obs0, fv<float, 2>* pf) {
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) {
    const double2 lo = *reinterpret_cast<const double2*>(scaled + (int64_t)i * npad + obs0);
    pf[i][0] = (float)lo.x;
    pf[i][1] = (float)lo.y;
    asm volatile("" : "+v"(pf[i]));  // converted here (or the doubles stay live and every use converts again)
  }
}

__device__ __forceinline__ void hard_store_divisor(double* mant, int32_t* exp, int64_t at, const double* m_in, const int32_t* e_in) {
  double2 m;
  int2 e;
  int de = 0;
  m.x = frexp(m_in[0], &de);
  e.x = e_in[0] + de;
  m.y = frexp(m_in[1], &de);
  e.y = e_in[1] + de;
  *reinterpret_cast<double2*>(mant + at) = m;
  *reinterpret_cast<int2*>(exp + at) = e;
}

// families whose state bytes are in flight per wave: a request is 128 bytes only, so it takes many of
// them to cover the memory latency
// (the loops are unrolled over one ring; a second ring behind it doubles the distance)
constexpr int32_t kHardRingFirst = 4, kHardRingLight = 8, kHardRing = 8;

// ---------------------------------------------------------------------------------
// The per-state product of the q over the chunk's families is carried as a sum of logarithms in
// double: log2 q[0][d] summed over the chunk is the same for every observation (the wave adds it
// up once, at the end), and an observation adds the difference of its own form, one fused
// multiply-add per family and state.  (The float kernels multiply the q up in double; both are the
// product of the same float factors to some 1e-13.)
template <int D0>
__global__ void __launch_bounds__(256) hard_loop_first(const float* __restrict__ cpts, LeafFamilyArgs<float> a) {
  constexpr int32_t V = 2, S = 2;
  using Stage = HardStage<D0, true>;
  extern __shared__ __align__(16) float hard_staged[];
  HardWave<D0> w;
  w.place(a, a.chunk_len);
  const float* tab = a.hard_tab;
  Stage::fill(tab, w.first, w.last, kHardQ, hard_staged);
  if (!w.active) return;

  fv<float, V> pf[D0];
  hard_load_scaled<D0>(a.scaled, a.npad, w.obs0, pf);
  // r = P' / (uniform previous message), the same for every family (leaf_loop_p1, fresh state)
  fv<float, V> r[D0], sum_r = splat<float, V>(0.f);
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) {
    r[d] = pf[d] * rcpv<float, V>(splat<float, V>(1.0f / (float)D0));
    sum_r += r[d];
  }
  const fv<float, V> inv_sum_r = rcpv<float, V>(sum_r);
  double lacc[D0][V];
#pragma unroll
  for (int32_t d = 0; d < D0; ++d)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) lacc[d][v] = 0.0;
  int32_t n_one[V] = {0, 0}, n_third[V] = {0, 0};
  double pz[V] = {1.0, 1.0};
  int32_t pze[V] = {0, 0};
  int32_t until_renorm = 6;  // six float factors cannot leave the double range

  uint32_t ring[kHardRingFirst], behind[kHardRingFirst];
#pragma unroll
  for (int32_t u = 0; u < kHardRingFirst; ++u) {
    ring[u] = w.pair(w.first + u < w.last ? w.first + u : w.last - 1);
    behind[u] = w.pair(w.first + kHardRingFirst + u < w.last ? w.first + kHardRingFirst + u : w.last - 1);
  }
  for (int32_t k0 = w.first; k0 < w.last; k0 += kHardRingFirst) {
#pragma unroll
    for (int32_t u = 0; u < kHardRingFirst; ++u) {
      const int32_t k = k0 + u;
      if (k >= w.last) break;
      const uint32_t pair = ring[u];
      ring[u] = behind[u];
      behind[u] = w.pair(k + 2 * kHardRingFirst < w.last ? k + 2 * kHardRingFirst : w.last - 1);
      const float* staged = hard_staged + (size_t)(k - w.first) * Stage::kStride;
      HardRows<D0> q;
      q.load(staged);
      const HardForms f(pair);
      const bool any_third = f.any_third();
      // the leaf's next normaliser from uniform previous messages: sum(r q) / sum(r)
      const fv<float, V> acc = hard_dot_own<D0>(r, tab, k, kHardQ + 2, q, f, any_third);
      const fv<float, V> zn = acc * inv_sum_r;
      __builtin_amdgcn_sched_barrier(0);  // the float rows are done with before the row of doubles is fetched
      const double* step = static_cast<const double*>(__builtin_assume_aligned(staged + Stage::kFloats, 8));
      double is_one[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        pz[v] *= (double)zn[v];
        is_one[v] = f.one[v] ? 1.0 : 0.0;
        n_one[v] += f.one[v] ? 1 : 0;
      }
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const double x = step[d];
#pragma unroll
        for (int32_t v = 0; v < V; ++v) lacc[d][v] = fma(is_one[v], x, lacc[d][v]);
        if (d == D0 / 2) __builtin_amdgcn_sched_barrier(0);
      }
      if (any_third) {
        HardLogRow<D0> third;
        third.load(tab, k, 2);
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          const double is_third = f.third[v] ? 1.0 : 0.0;
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) lacc[d][v] = fma(is_third, third.x[d], lacc[d][v]);
          n_third[v] += f.third[v] ? 1 : 0;
        }
      }
      if (--until_renorm == 0) {
        until_renorm = 6;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) pze[v] += frexp_in_place(pz[v]);
      }
    }
  }
  const int64_t zat = (int64_t)(a.chunk_base + w.chunk) * a.npad + w.obs0;
  hard_store_divisor(a.zdiv_next_mant, a.zdiv_next_exp, zat, pz, pze);  // prod z of the next iteration
  // prod z of this iteration: the leaf's first normaliser depends on the evidence form only
  {
    double lz[kHardForms];
#pragma unroll
    for (int32_t y = 0; y < kHardForms; ++y) {
      float z = hard_ev(y, 0, a.on_value, a.off_value) * (1.0f / (float)S);
      z = fmaf(hard_ev(y, 1, a.on_value, a.off_value), 1.0f / (float)S, z);
      lz[y] = z > 0.f ? log2((double)z) : kHardLogOfZero;
    }
    double2 m;
    int2 e;
    const int32_t n = w.last - w.first;
    m.x = hard_exp2_split((double)(n - n_one[0] - n_third[0]) * lz[0] + ((double)n_one[0] * lz[1] + (double)n_third[0] * lz[2]), &e.x);
    m.y = hard_exp2_split((double)(n - n_one[1] - n_third[1]) * lz[0] + ((double)n_one[1] * lz[1] + (double)n_third[1] * lz[2]), &e.y);
    *reinterpret_cast<double2*>(a.zdiv_mant + zat) = m;
    *reinterpret_cast<int2*>(a.zdiv_exp + zat) = e;
  }
  // sum over the chunk of log2 q[0][d]: lanes take families, then the butterfly
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) {
    double base = 0.0;
    for (int32_t k = w.first + w.lane; k < w.last; k += 64)
      base += reinterpret_cast<const double*>(tab + (size_t)k * (kHardRows * D0) + (size_t)kHardLog * D0)[d];
#pragma unroll
    for (int32_t o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o, 64);
    double2 m;
    int2 e;
    m.x = hard_exp2_split(base + lacc[d][0], &e.x);
    m.y = hard_exp2_split(base + lacc[d][1], &e.y);
    const int64_t row = ((int64_t)(a.chunk_base + w.chunk) * D0 + d) * a.npad + w.obs0;
    *reinterpret_cast<double2*>(a.part_mant + row) = m;
    *reinterpret_cast<int2*>(a.part_exp + row) = e;
  }
}

// ---------------------------------------------------------------------------------
template <int D0>
__global__ void __launch_bounds__(256) hard_loop_light(const float* __restrict__ cpts, LeafFamilyArgs<float> a) {
  constexpr int32_t V = 2;
  using Stage = HardStage<D0, false>;
  extern __shared__ __align__(16) float hard_staged[];
  HardWave<D0> w;
  w.place(a, a.chunk_len);
  const float* tab = a.hard_tab;
  Stage::fill(tab, w.first, w.last, kHardRq, hard_staged);
  if (!w.active) return;

  fv<float, V> pf[D0];
  hard_load_scaled<D0>(a.scaled, a.npad, w.obs0, pf);
  fv<float, V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  double pz[V] = {1.0, 1.0};
  int32_t pze[V] = {0, 0};
  int32_t until_renorm = 6;

  uint32_t ring[kHardRingLight], behind[kHardRingLight];
#pragma unroll
  for (int32_t u = 0; u < kHardRingLight; ++u) {
    ring[u] = w.pair(w.first + u < w.last ? w.first + u : w.last - 1);
    behind[u] = w.pair(w.first + kHardRingLight + u < w.last ? w.first + kHardRingLight + u : w.last - 1);
  }
  // two families per turn (the second one of an odd chunk's last turn is the first one again and discarded)
  for (int32_t k0 = w.first; k0 < w.last; k0 += kHardRingLight) {
#pragma unroll
    for (int32_t u = 0; u < kHardRingLight; u += 2) {
      const int32_t k = k0 + u;
      if (k >= w.last) break;
      const bool both = k + 1 < w.last;
      const uint32_t pair_a = ring[u], pair_b = ring[u + 1];
      ring[u] = behind[u];
      ring[u + 1] = behind[u + 1];
      behind[u] = w.pair(k + 2 * kHardRingLight < w.last ? k + 2 * kHardRingLight : w.last - 1);
      behind[u + 1] = w.pair(k + 1 + 2 * kHardRingLight < w.last ? k + 1 + 2 * kHardRingLight : w.last - 1);
      HardRows<D0> rq_a, rq_b;
      rq_a.load(hard_staged + (size_t)(k - w.first) * Stage::kStride);
      rq_b.load(hard_staged + (size_t)(both ? k + 1 - w.first : k - w.first) * Stage::kStride);
      const HardForms fa(pair_a), fb(pair_b);
      // z' = sum(P') / sum(P' / q)
      fv<float, V> sum_a, sum_b;
      hard_dot_own2<D0>(pf, tab, k, both ? k + 1 : k, kHardRq + 2, rq_a, rq_b, fa, fb, fa.any_third() || fb.any_third(), &sum_a, &sum_b);
      const fv<float, V> zn_a = pf_sum * rcpv<float, V>(sum_a);
      fv<float, V> zn_b = pf_sum * rcpv<float, V>(sum_b);
      if (!both) zn_b = splat<float, V>(1.0f);
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pz[v] *= (double)zn_a[v] * (double)zn_b[v];
      until_renorm -= 2;
      if (until_renorm <= 0) {
        until_renorm = 6;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) pze[v] += frexp_in_place(pz[v]);
      }
    }
  }
  hard_store_divisor(a.zdiv_next_mant, a.zdiv_next_exp, (int64_t)(a.chunk_base + w.chunk) * a.npad + w.obs0, pz, pze);
}

// ---------------------------------------------------------------------------------
// The count sums on the matrix cores.  The leaves' messages to the parent are the q of the step
// from the second iteration on, so the parent's scaled products of two iterations differ by a
// factor per observation only, and a family's joint posterior of an observation n of form y is
//   cpt[d][j] * ev(y)[j] / q[y][d] * b[d][n],   b[.][n] = P'[.][n] / sum_d P'[d][n]
// with the parent's posterior b the same for all families.  The sums by form are therefore
//   G[d][(family, form)] = sum_n b[d][n] * S[(family, form)][n],   S = 1 where n has that form, else 0,
// a product of a 16 x 128 and a 128 x 16 matrix per eight families and wave: b sits in registers in
// the operand layout for the whole chunk, and S comes straight from the state bytes (a lane reads
// the 32 bytes of its column and quarter of the wave's observations).  Observations of the third
// form (rare) are added with atomics to sums of their own (hard_third_sums), already weighted for
// forms 0 and 1.  What is left per family and observation is the leaf's normaliser of this
// iteration for the log-likelihood, formed again from the previous P' as hard_loop_light forms it.
//
// a.chunks / a.chunk_len: the launch's own cut of the families (it writes no partial products, so it
// need not follow the parent's chunks); a.slots: per-wave sum slots per entry (ntiles * 2).
// em_partial[k][d * 2 + y][slot]: the wave's share of G[y][d] (see hard_count).
constexpr int32_t kHardGroup = 8;  // families per matrix product

// [K][2][D0] sums of the third-form observations, behind the records
template <int D0>
__device__ __forceinline__ float* hard_third_sums(const float* tab, int32_t K) {
  return const_cast<float*>(tab) + hard_table_floats(K, D0);
}

// one observation of the third form: b[d], weighted for forms 0 and 1, onto the family's sums
template <int D0>
__device__ __forceinline__ void hard_third_add(const float* tab, const double* scaled, int64_t npad, int32_t K, int32_t k,
                                               int64_t obs) {
  const float* rec = tab + (size_t)k * (kHardRows * D0);
  float* third = hard_third_sums<D0>(tab, K) + (size_t)k * (2 * D0);
  // (rare: compact loops, so that the callers stay small and hold nothing for this in registers)
  float sum = 0.f;
#pragma unroll 1
  for (int32_t d = 0; d < D0; ++d) sum += (float)scaled[(int64_t)d * npad + obs];
  const float scale = fast_rcp(sum);
#pragma unroll 1
  for (int32_t d = 0; d < D0; ++d) {
    const float x = (float)scaled[(int64_t)d * npad + obs] * scale;
    atomicAdd(third + d, x * rec[kHardThird * D0 + d]);
    atomicAdd(third + D0 + d, x * rec[(kHardThird + 1) * D0 + d]);
  }
}

// the state bytes [n0, n0 + 32) of one leaf's row (any alignment; n0 and N are even).  WHOLE: they all lie
// before the row's end; otherwise bytes past the end read as state 0 (pairs of bytes from addresses
// that always exist, all in flight together: a wave of the last tile must not take longer than the others)
template <bool WHOLE>
__device__ __forceinline__ void hard_states32(const uint8_t* row, int64_t n0, int64_t N, uint32_t* out) {
  if constexpr (WHOLE) {
    uint4 lo, hi;
    __builtin_memcpy(&lo, row + n0, 16);
    __builtin_memcpy(&hi, row + n0 + 16, 16);
    out[0] = lo.x, out[1] = lo.y, out[2] = lo.z, out[3] = lo.w;
    out[4] = hi.x, out[5] = hi.y, out[6] = hi.z, out[7] = hi.w;
  } else {
    uint16_t two[16];
#pragma unroll
    for (int32_t x = 0; x < 16; ++x) {
      const int64_t n = n0 + 2 * x;
      __builtin_memcpy(&two[x], row + (n + 2 <= N ? n : 0), 2);
    }
#pragma unroll
    for (int32_t x = 0; x < 8; ++x) {
      const int64_t n = n0 + 4 * x;
      const uint32_t lo = n + 2 <= N ? two[2 * x] : 0u, hi = n + 4 <= N ? two[2 * x + 1] : 0u;
      out[x] = lo | (hi << 16);
    }
  }
}

// x + the same value of the lanes 8, 4, 2 and 1 further on in the row of 16: every lane ends with the row's sum
__device__ __forceinline__ float hard_row_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int32_t, x), 0x128, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int32_t, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int32_t, x), 0x122, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int32_t, x), 0x121, 0xf, 0xf, false));
  return x;
}

template <int D0>
__global__ void __launch_bounds__(256) hard_reduce_mfma(const float* __restrict__ cpts, LeafFamilyArgs<float> a) {
  static_assert(D0 <= 16, "one 16-row operand holds the parent's states");
  static_assert(kHardRing == kHardGroup, "one ring of state bytes per matrix product");
  constexpr int32_t V = 2, E = D0 * 2;
  using Stage = HardStage<D0, false>;
  typedef float float4v __attribute__((ext_vector_type(4)));
  extern __shared__ __align__(16) float hard_staged[];  // the chunk's rows
  HardWave<D0> w;
  w.place(a, a.chunk_len);
  const float* tab = a.hard_tab;
  Stage::fill(tab, w.first, w.last, kHardRq, hard_staged);
  if (!w.active) return;

  // ---- the leaves' normalisers of this iteration, two families per turn: z = sum(P') / sum(P' / q) of
  // the previous P' (the operations of the loop launch that was left out)
  bool saw_third = false;  // any observation of the third form among the wave's (the same in all lanes)
  {
    fv<float, V> prev[D0];
    hard_load_scaled<D0>(a.scaled_prev, a.npad, w.obs0, prev);
    fv<float, V> prev_sum = prev[0];
#pragma unroll
    for (int32_t i = 1; i < D0; ++i) prev_sum += prev[i];
    double pz[V] = {1.0, 1.0};
    int32_t pze[V] = {0, 0};
    int32_t until_renorm = 6;  // six float factors cannot leave the double range
    uint32_t ring[kHardRing], behind[kHardRing];
#pragma unroll
    for (int32_t u = 0; u < kHardRing; ++u) {
      ring[u] = w.pair(w.first + u < w.last ? w.first + u : w.last - 1);
      behind[u] = w.pair(w.first + kHardRing + u < w.last ? w.first + kHardRing + u : w.last - 1);
    }
    for (int32_t k0 = w.first; k0 < w.last; k0 += kHardRing) {
#pragma unroll
      for (int32_t u = 0; u < kHardRing; u += 2) {
        const int32_t k = k0 + u;
        if (k >= w.last) break;
        const bool both = k + 1 < w.last;
        const uint32_t pair_a = ring[u], pair_b = ring[u + 1];
        ring[u] = behind[u];
        ring[u + 1] = behind[u + 1];
        behind[u] = w.pair(k + 2 * kHardRing < w.last ? k + 2 * kHardRing : w.last - 1);
        behind[u + 1] = w.pair(k + 1 + 2 * kHardRing < w.last ? k + 1 + 2 * kHardRing : w.last - 1);
        HardRows<D0> rq_a, rq_b;
        rq_a.load(hard_staged + (size_t)(k - w.first) * Stage::kStride);
        rq_b.load(hard_staged + (size_t)(both ? k + 1 - w.first : k - w.first) * Stage::kStride);
        const HardForms fa(pair_a), fb(pair_b);
        fv<float, V> sum_a, sum_b;
        const bool any_third = fa.any_third() || fb.any_third();
        saw_third |= any_third;
        hard_dot_own2<D0>(prev, tab, k, both ? k + 1 : k, kHardRq + 2, rq_a, rq_b, fa, fb, any_third, &sum_a, &sum_b);
        const fv<float, V> z_a = prev_sum * rcpv<float, V>(sum_a);
        fv<float, V> z_b = prev_sum * rcpv<float, V>(sum_b);
        if (!both) z_b = splat<float, V>(1.0f);
#pragma unroll
        for (int32_t v = 0; v < V; ++v) pz[v] *= (double)z_a[v] * (double)z_b[v];
        until_renorm -= 2;
        if (until_renorm <= 0) {
          until_renorm = 6;
#pragma unroll
          for (int32_t v = 0; v < V; ++v) pze[v] += frexp_in_place(pz[v]);
        }
      }
    }
    double* dst = a.ll_part + (int64_t)w.chunk * a.npad + w.obs0;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      const double m = frexp(pz[v], &de);
      dst[v] = (double)(pze[v] + de) * 0.6931471805599453 + log(m);
    }
  }
  if (!a.em_partial) return;
  if (saw_third) {
    // (rare) every lane looks at its own two observations of every family once more
#pragma unroll 1
    for (int32_t k = w.first; k < w.last; ++k) {
      const HardForms f(w.pair(k));
#pragma unroll
      for (int32_t v = 0; v < V; ++v)
        if (f.third[v] && w.obs0 + v < a.N) hard_third_add<D0>(tab, a.scaled, a.npad, a.K, k, w.obs0 + v);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // nothing of the second pass is started (and held in registers) during the first

  // ---- the sums by form.  b as the 16 x 4 operand of each of the 32 steps: lane (i = lane % 16, kk = lane / 16)
  // holds state i of observations 32 kk .. 32 kk + 31 of the wave's 128 (zero for the states and
  // observations there are not)
  const int32_t op_i = w.lane & 15, op_kk = w.lane >> 4;
  const int64_t op_n0 = w.t * 256 + w.part * 128 + op_kk * 32;
  float p_op[32];
  {
    const double* src = a.scaled + (int64_t)(op_i < D0 ? op_i : 0) * a.npad + op_n0;
    double2 raw[16];  // all in flight together
#pragma unroll
    for (int32_t t = 0; t < 16; ++t) raw[t] = *reinterpret_cast<const double2*>(src + 2 * t);
#pragma unroll
    for (int32_t t = 0; t < 32; t += 2) {
      p_op[t] = (op_i < D0 && op_n0 + t < a.N) ? (float)raw[t >> 1].x : 0.f;
      p_op[t + 1] = (op_i < D0 && op_n0 + t + 1 < a.N) ? (float)raw[t >> 1].y : 0.f;
    }
    // converted here and now (left alone the conversions sink into the loop, and the doubles stay live)
#pragma unroll
    for (int32_t t = 0; t < 32; ++t) asm volatile("" : "+v"(p_op[t]));
#pragma unroll
    for (int32_t t = 0; t < 32; ++t) {
      const float sum = hard_row_sum(p_op[t]);
      p_op[t] *= op_n0 + t < a.N ? fast_rcp(sum) : 0.f;
    }
  }
  const int32_t slot = (int32_t)(w.t * 2 + w.part);
  // the state bytes of column j = (family k0 + j / 2, form j % 2) in the operand layout, one group ahead
  const int32_t op_k = op_i >> 1;
  const uint32_t op_y = (uint32_t)(op_i & 1) * 0x01010101u;
  auto sums = [&](auto whole) {
  constexpr bool kWhole = decltype(whole)::value;
  uint32_t sb[8];
  hard_states32<kWhole>(w.states + (int64_t)(w.first + op_k < w.last ? w.first + op_k : w.last - 1) * w.stride, op_n0, a.N, sb);
  for (int32_t k0 = w.first; k0 < w.last; k0 += kHardGroup) {
    uint32_t sb_next[8];
    {
      const int32_t kn = k0 + kHardGroup + op_k;
      hard_states32<kWhole>(w.states + (int64_t)(kn < w.last ? kn : w.last - 1) * w.stride, op_n0, a.N, sb_next);
    }
    const int32_t k = k0 + op_k;
    // one byte per observation: 1 where it has this column's form
    uint32_t is[8], beyond = 0;
#pragma unroll
    for (int32_t x = 0; x < 8; ++x) {
      beyond |= sb[x] & 0xfefefefeu;
      is[x] = (sb[x] ^ op_y ^ 0x01010101u) & 0x01010101u;
    }
    if (beyond) {
      // third-form observations belong to neither column (their share goes to sums of their own, below)
#pragma unroll
      for (int32_t x = 0; x < 8; ++x) {
        const uint32_t t = sb[x] & 0xfefefefeu;
        is[x] &= ~(((((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u) >> 7);
      }
    }
    float4v sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int32_t t = 0; t < 32; ++t) {
      const float s = (float)((is[t >> 2] >> (8 * (t & 3))) & 0xffu);
      sum = __builtin_amdgcn_mfma_f32_16x16x4f32(p_op[t], s, sum, 0, 0, 0);
    }
    // lane (j, kk) holds states 4 kk .. 4 kk + 3 of column j
    if (k < w.last) {
      const int32_t y = op_i & 1;
#pragma unroll
      for (int32_t x = 0; x < 4; ++x) {
        const int32_t d = op_kk * 4 + x;
        if (d < D0) a.em_partial[((int64_t)k * E + d * 2 + y) * a.slots + slot] = sum[x];
      }
    }
#pragma unroll
    for (int32_t x = 0; x < 8; ++x) sb[x] = sb_next[x];
  }
  };
  // (a wave whose observations all exist - all but the last tile's - reads its state bytes without looking)
  if (w.t * 256 + w.part * 128 + 128 <= a.N)
    sums(std::true_type{});
  else
    sums(std::false_type{});
}

}  // namespace bn
