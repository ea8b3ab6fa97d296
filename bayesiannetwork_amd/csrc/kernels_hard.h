// The leaf-family launches of a whole EM step (bn_em_step, bn_em_step_statistics) whose evidence is
// compact hard evidence - one uint8 state per observed two-state leaf and observation - read where
// the caller holds it, in every launch (nothing is copied: the plan is reset when the step ends).
//
// The evidence of a leaf can take three forms only (state 0, state 1, any other state: both rows at
// the off value), so everything that depends on it and on the family's CPT alone exists in three
// versions per family: q[y][d] = cpt[d] . ev(y), 1 / q[y][d] and log2 q[y][d].  hard_tables_build
// writes these once per step (the parameters do not change inside one), q and 1 / q with the very
// operations the float kernels apply to the expanded evidence.  No product with the evidence and
// no reciprocal per state is left in the family loops:
//   * the per-state product of the q over a chunk's families is carried as a sum of logarithms in
//     double, one fused multiply-add per family and state (hard_loop_first, which stages its
//     chunk's rows in LDS once per block and computes a dot product with a family's row for both
//     common forms at once - the rows are the same for all lanes - every observation keeping the
//     result of its own form);
//   * the dot products of the later launches' normalisers are matrix products of the parent's P'
//     (128 observations x states) with the families' rows of 1 / q (hard_loop_light);
//   * the count sums factor into sums by evidence form, which are matrix products of the parent's
//     posterior with the 0 / 1 pattern of the state bytes (hard_reduce_mfma).
//
//   hard_loop_first   the call's first iteration on fresh messages: what leaf_loop_p1 does in its
//                     fresh / ingesting state (prod q per state, prod z of iterations 1 and 2)
//   hard_loop_light   a later loop iteration: prod z of the next iteration from the parent's P'
//   hard_reduce_mfma  the last iteration of a statistics call: the count sums (the leaves'
//                     normalisers are those of the last hard_loop_light launch)
// Float plans, two-state leaves, parents of at most 16 states, the leaves' state rows at equal
// distances in memory (LeafFamilyArgs::src_base / src_stride), an even number of observations.
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
    float q = cpt[d * 2] * hard_ev(a.hard_rows, y, 0);
    q = fmaf(cpt[d * 2 + 1], hard_ev(a.hard_rows, y, 1), q);
    qs[y] = q;
    rq[y] = fast_rcp(q);
    rec[(kHardQ + y) * D0] = q;
    rec[(kHardRq + y) * D0] = rq[y];
  }
  // x0, x1 with  rq0 * ev(0)[j] * x0 + rq1 * ev(1)[j] * x1 = rq2 * ev(2)[j]  for both j
  {
    const double e00 = hard_ev(a.hard_rows, 0, 0), e01 = hard_ev(a.hard_rows, 0, 1), e10 = hard_ev(a.hard_rows, 1, 0),
                 e11 = hard_ev(a.hard_rows, 1, 1);
    const double t0 = (double)rq[2] * hard_ev(a.hard_rows, 2, 0), t1 = (double)rq[2] * hard_ev(a.hard_rows, 2, 1);
    const double det = e00 * e11 - e10 * e01;
    const bool none = det == 0.0;  // (the two forms' rows in line: an observation of the third form counts for form 0)
    rec[(kHardThird + 0) * D0] = (float)(none ? 1.0 : (t0 * e11 - t1 * e10) / (det * (double)rq[0]));
    rec[(kHardThird + 1) * D0] = (float)(none ? 0.0 : (t1 * e00 - t0 * e01) / (det * (double)rq[1]));
  }
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
constexpr int32_t kHardRingFirst = 4;

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
      float z = hard_ev(a.hard_rows, y, 0) * (1.0f / (float)S);
      z = fmaf(hard_ev(a.hard_rows, y, 1), 1.0f / (float)S, z);
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
// hard_loop_light and hard_reduce_mfma: everything on the matrix cores.
//
// The leaves' normalisers.  z[n][k] = sum(P'[.][n]) / sum_d P'[d][n] / q[k][y(n, k)][d] needs, per family
// and observation, the dot product of the parent's P' with the row 1 / q of the observation's form.
// For 16 families that is, per common form, a product of a 128 x D0 matrix (the wave's observations)
// and a D0 x 16 matrix (column j = family k0 + j): three v_mfma_f32_16x16x4_f32 per 16 observations
// and form for up to 12 states.  Every lane ends up with both forms' dot products of ONE family for
// 32 observations, keeps the one each observation's state byte names and adds up the logarithms:
// prod_k z = sum(P')^K / prod_k (dot product of the own form).  The sums over the families are taken
// once, at the end of the chunk.
//
// The count sums (reduce launch).  The leaves' messages to the parent are the q of the step from the
// second iteration on, so the parent's scaled products of two iterations differ by a factor per
// observation only, and a family's joint posterior of an observation n of form y is
//   cpt[d][j] * ev(y)[j] / q[y][d] * b[d][n],   b[.][n] = P'[.][n] / sum_d P'[d][n]
// with the parent's posterior b the same for all families.  The sums by form are therefore
//   G[y][d][family] = sum_n b[d][n] * S[y][family][n],   S[y] = 1 where n has form y, else 0,
// per 16 families and wave a product of a 16 x 128 and a 128 x 16 matrix for form 1, whose S is the
// state byte itself; form 0 gets what is left of sum_n b[d][n], which is the same for all families
// (a group of families with an observation of the third form has its form 0 multiplied out as well).
//
// A lane (j = lane % 16, kk = lane / 16) reads the 32 state bytes of family k0 + j for the wave's
// observations 32 kk .. 32 kk + 31 per 16 families; nothing goes through LDS and no sum crosses the
// lanes before the end.  Observations of the third form (rare) fall out of both forms; a wave that
// meets one goes over its families again at the end, lane by lane.
constexpr int32_t kHardGroup = 16;  // families per matrix product
constexpr int32_t kHardAhead = 2;   // groups in flight per wave besides the one at work
constexpr int32_t kHardLogFloats = 32 * 64;  // a wave's sums of logarithms in LDS
constexpr size_t kHardLogBytes = (size_t)kWavesPerBlock * kHardLogFloats * sizeof(float);  // dynamic LDS of a block

// [K][2][D0] sums of the third-form observations, behind the records
template <int D0>
__device__ __forceinline__ float* hard_third_sums(const float* tab, int32_t K) {
  return const_cast<float*>(tab) + hard_table_floats(K, D0);
}

// one observation of the third form: b[d], weighted for forms 0 and 1, onto the family's sums.
// form0: the wave's sum slot of the family's entry (d = 0, form 0), entries 2 * slots floats apart per state: the
// observation is taken out of form 0's sums again
template <int D0>
__device__ __forceinline__ void hard_third_add(const float* tab, const double* scaled, int64_t npad, int32_t K, int32_t k,
                                               int64_t obs, float* form0, int32_t slots) {
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
    atomicAdd(form0 + (int64_t)d * 2 * slots, -x);
  }
}

// The state bytes through a buffer descriptor over the whole array of rows: a read past its end returns
// zeros instead of faulting, so a lane may always read the 32 bytes of its observations, whether
// they all exist or not (past a row's end it reads the next row's first bytes).
struct HardStates {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t stride;
  __device__ __forceinline__ HardStates(const LeafFamilyArgs<float>& a)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src_base), 0,
                                               (int32_t)((int64_t)(a.K - 1) * a.src_stride + a.N), 0x00020000)),
        stride((uint32_t)a.src_stride) {}
  // the bytes [n0, n0 + 32) of row k (any alignment)
  __device__ __forceinline__ void load32(int32_t k, uint32_t n0, uint32_t* out) const {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t at = (uint32_t)k * stride + n0;
    const u4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, at, 0, 0);
    const u4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, at + 16, 0, 0);
    out[0] = lo.x, out[1] = lo.y, out[2] = lo.z, out[3] = lo.w;
    out[4] = hi.x, out[5] = hi.y, out[6] = hi.z, out[7] = hi.w;
  }
  // the two bytes at n0 of row k
  __device__ __forceinline__ uint32_t load2(int32_t k, uint32_t n0) const {
    return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (uint32_t)k * stride + n0, 0, 0);
  }
};
// can the launches address the array that way (32-bit offsets)?
inline bool hard_states_addressable(int32_t K, int64_t stride, int64_t N) {
  return stride > 0 && (int64_t)(K - 1) * stride + N + 64 < ((int64_t)1 << 31);
}

// x + the same value of the lanes 8, 4, 2 and 1 further on in the row of 16: every lane ends with the row's sum
__device__ __forceinline__ float hard_row_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int32_t, x), __builtin_bit_cast(int32_t, x), 0x128, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int32_t, x), __builtin_bit_cast(int32_t, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int32_t, x), __builtin_bit_cast(int32_t, x), 0x122, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int32_t, x), __builtin_bit_cast(int32_t, x), 0x121, 0xf, 0xf, false));
  return x;
}

typedef float hard_float4 __attribute__((ext_vector_type(4)));

// What a wave holds of one group of 16 families: lane (j, kk) has the 32 state bytes of family k0 + j for the
// observations 32 kk .. and - NS > 0, for the normalisers - as the right operand of their products 1 / q
// of that family per form for the states 4 s + kk.
template <int NS>
struct HardGroupData {
  uint32_t bytes[8];
  uint32_t cut;  // the two bytes of the dword the row's end cuts (waves of the last tile only)
  float rq0[NS > 0 ? NS : 1], rq1[NS > 0 ? NS : 1];
};

// Where a lane's state bytes are, and how to get a group's
template <int D0, int NS>
struct HardGroupLoader {
  HardStates states;
  const float* tab;
  int32_t op_i, op_kk, last, cut;
  int64_t op_n0, N;
  bool whole;  // do the wave's observations all exist?
  __device__ __forceinline__ HardGroupLoader(const LeafFamilyArgs<float>& a, const HardWave<D0>& w) : states(a) {
    tab = a.hard_tab;
    op_i = w.lane & 15;
    op_kk = w.lane >> 4;
    last = w.last;
    N = a.N;
    const int64_t wave_n0 = w.t * 256 + w.part * 128;
    whole = wave_n0 + 128 <= a.N;
    op_n0 = wave_n0 + op_kk * 32;
    // the dword of a lane's 32 bytes that the end of the row cuts in two (N is even): its first two bytes are
    // read on their own, since a dword that reaches past the end of the whole array reads as zero
    const int64_t left0 = a.N - op_n0;
    cut = (!whole && left0 > 0 && left0 < 32 && (left0 & 3) == 2) ? (int32_t)(left0 >> 2) : -1;
  }
  __device__ __forceinline__ void load(int32_t k0, HardGroupData<NS>& g) const {
    const int32_t k = k0 + op_i < last ? k0 + op_i : last - 1;
    states.load32(k, (uint32_t)op_n0, g.bytes);
    if (!whole) g.cut = states.load2(k, (uint32_t)op_n0 + (cut >= 0 ? 4 * cut : 0));
    if constexpr (NS > 0) {
      const float* rec = tab + (size_t)k * (kHardRows * D0) + (size_t)kHardRq * D0;
#pragma unroll
      for (int32_t s = 0; s < NS; ++s) {
        const int32_t d = 4 * s + op_kk;
        const float x0 = rec[d < D0 ? d : 0], x1 = rec[D0 + (d < D0 ? d : 0)];
        g.rq0[s] = d < D0 ? x0 : 0.f;
        g.rq1[s] = d < D0 ? x1 : 0.f;
      }
    }
  }
  // a group's bytes as they came in -> as they count: observations that do not exist read as state 0 (what
  // lies behind the row's end is not theirs).  Returns whether a state beyond 1 is among them.
  __device__ __forceinline__ bool settle(int32_t k0, HardGroupData<NS>& g) const {
    if (!whole) {
#pragma unroll
      for (int32_t x = 0; x < 8; ++x) {
        const int64_t left = N - (op_n0 + 4 * x);
        g.bytes[x] = left >= 4 ? g.bytes[x] : (x == cut ? g.cut : 0u);
      }
    }
    uint32_t beyond = 0;
#pragma unroll
    for (int32_t x = 0; x < 8; ++x) beyond |= g.bytes[x] & 0xfefefefeu;
    return __any((int)(beyond != 0 && k0 + op_i < last)) != 0;
  }
};

// The normalisers of a wave's chunk from the P' in z_src: log2 of prod_k z for this lane's two
// observations (w.obs0, + 1)
template <int D0>
__device__ __forceinline__ void hard_chunk_normalisers(const LeafFamilyArgs<float>& a, const HardWave<D0>& w, const double* z_src,
                                                       double* log2_z) {
  constexpr int32_t V = 2, NS = (D0 + 3) / 4;
  const float* tab = a.hard_tab;
  const HardGroupLoader<D0, NS> loader(a, w);
  const int32_t op_i = loader.op_i, op_kk = loader.op_kk;
  const int64_t wave_n0 = w.t * 256 + w.part * 128;

  // P' as the left operand of the products: step (o, s) takes rows m = observation 32 (m / 4) + 4 o + m % 4
  // and columns d = 4 s .. 4 s + 3; lane (m = lane % 16, lane / 16) holds entry (m, 4 s + lane / 16).
  // Observations that do not exist stand in with (1, 0, 0, ..): any positive sum.
  float z_op[8][NS];
  {
    const int64_t n_m = wave_n0 + 32 * (op_i >> 2) + (op_i & 3);
    double raw[8][NS];
#pragma unroll
    for (int32_t o = 0; o < 8; ++o)
#pragma unroll
      for (int32_t s = 0; s < NS; ++s) {
        const int32_t d = 4 * s + op_kk;
        raw[o][s] = z_src[(int64_t)(d < D0 ? d : 0) * a.npad + n_m + 4 * o];
      }
#pragma unroll
    for (int32_t o = 0; o < 8; ++o)
#pragma unroll
      for (int32_t s = 0; s < NS; ++s) {
        const int32_t d = 4 * s + op_kk;
        z_op[o][s] = n_m + 4 * o < a.N ? (d < D0 ? (float)raw[o][s] : 0.f) : (d == 0 ? 1.f : 0.f);
        asm volatile("" : "+v"(z_op[o][s]));  // converted here (or the doubles stay live)
      }
  }
  // sum over the families of log2(dot product of the own form), for the observations 32 kk + 4 o + r: 32 sums
  // per lane, kept in LDS (the four of step o of lane L at [o * 64 + L] of the wave's 512 vectors; only this
  // lane touches them)
  extern __shared__ __align__(16) float hard_logs[];
  hard_float4* lacc = reinterpret_cast<hard_float4*>(hard_logs) + (size_t)w.wave * (kHardLogFloats / 4) + w.lane;
#pragma unroll
  for (int32_t o = 0; o < 8; ++o) lacc[o * 64] = hard_float4{0.f, 0.f, 0.f, 0.f};
  bool saw_third = false;

  // one group: THIRD = some observation of the wave has a state beyond 1 for one of its families
  auto work = [&](auto third_tag, int32_t k0, HardGroupData<NS>& g) {
    constexpr bool THIRD = decltype(third_tag)::value;
    const bool valid = k0 + op_i < w.last;
    const hard_float4 zero = {0.f, 0.f, 0.f, 0.f};
    // the dot products of step o + 1 are on their way while those of step o are looked at
    auto dots = [&](int32_t o, hard_float4* dot0, hard_float4* dot1) {
      *dot0 = zero;
      *dot1 = zero;
#pragma unroll
      for (int32_t s = 0; s < NS; ++s) {
        *dot0 = __builtin_amdgcn_mfma_f32_16x16x4f32(z_op[o][s], g.rq0[s], *dot0, 0, 0, 0);
        *dot1 = __builtin_amdgcn_mfma_f32_16x16x4f32(z_op[o][s], g.rq1[s], *dot1, 0, 0, 0);
      }
    };
    hard_float4 next0, next1;
    dots(0, &next0, &next1);
#pragma unroll
    for (int32_t o = 0; o < 8; ++o) {
      const hard_float4 dot0 = next0, dot1 = next1;
      if (o + 1 < 8) dots(o + 1, &next0, &next1);
      const uint32_t four = g.bytes[o];
      hard_float4 logs4 = lacc[o * 64];
#pragma unroll
      for (int32_t r = 0; r < 4; ++r) {
        const uint32_t y = (four >> (8 * r)) & 0xffu;
        float x = y == 0u ? dot0[r] : dot1[r];
        if constexpr (THIRD) x = y > 1u ? 1.0f : x;  // (seen to at the end)
        x = valid ? x : 1.0f;
        logs4[r] += __builtin_amdgcn_logf(x);
      }
      lacc[o * 64] = logs4;
      __builtin_amdgcn_sched_barrier(0);  // (or the dot products of all eight steps are formed first and held)
    }
  };

  HardGroupData<NS> ring[kHardAhead + 1];
#pragma unroll
  for (int32_t g = 0; g <= kHardAhead; ++g) loader.load(w.first + g * kHardGroup, ring[g]);
  for (int32_t k0 = w.first; k0 < w.last; k0 += (kHardAhead + 1) * kHardGroup) {
#pragma unroll
    for (int32_t g = 0; g <= kHardAhead; ++g) {
      const int32_t kg = k0 + g * kHardGroup;
      if (kg >= w.last) break;
      if (loader.settle(kg, ring[g])) {
        saw_third = true;
        work(std::true_type{}, kg, ring[g]);
      } else {
        work(std::false_type{}, kg, ring[g]);
      }
      loader.load(kg + (kHardAhead + 1) * kHardGroup, ring[g]);
    }
  }

  // ---- the end of the chunk: the sums over the families; lane L keeps those of the observations 2 L and 2 L + 1
  // of the wave's 128 (which sit at o = (L % 16) / 2, r = 2 (L % 2) and + 1 in its row of 16 lanes)
  float logs[V] = {0.f, 0.f};
#pragma unroll
  for (int32_t o = 0; o < 8; ++o) {
    const hard_float4 mine = lacc[o * 64];
#pragma unroll
    for (int32_t r = 0; r < 4; ++r) {
      const float total = hard_row_sum(mine[r]);
      if (op_i == 2 * o + (r >> 1)) logs[r & 1] = total;
    }
  }
  // sum(P') per observation, and whatever the products left out
  fv<float, V> pf[D0];
  hard_load_scaled<D0>(z_src, a.npad, w.obs0, pf);
  fv<float, V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  if (saw_third) {
    // (rare) every lane looks at its own two observations of every family once more
#pragma unroll 1
    for (int32_t k = w.first; k < w.last; ++k) {
      const HardForms f(w.pair(k));
      if (!f.third[0] && !f.third[1]) continue;
      HardRow<D0> r2;
      r2.load(tab, k, kHardRq + 2);
      const fv<float, V> dot = hard_dot<D0>(pf, r2);
#pragma unroll
      for (int32_t v = 0; v < V; ++v)
        if (f.third[v]) logs[v] += __builtin_amdgcn_logf(dot[v]);
    }
  }
  const double families = (double)(w.last - w.first);
#pragma unroll
  for (int32_t v = 0; v < V; ++v) log2_z[v] = families * log2((double)pf_sum[v]) - (double)logs[v];
}

// a later loop iteration: prod z of the next iteration from the parent's P'
template <int D0>
__global__ void __launch_bounds__(256) hard_loop_light(const float* __restrict__ cpts, LeafFamilyArgs<float> a) {
  static_assert(D0 <= 16, "one 16-row operand holds the parent's states");
  HardWave<D0> w;
  w.place(a, a.chunk_len);
  if (!w.active) return;
  double log2_z[2];
  hard_chunk_normalisers<D0>(a, w, a.scaled, log2_z);
  double2 m;
  int2 e;
  m.x = hard_exp2_split(log2_z[0], &e.x);
  m.y = hard_exp2_split(log2_z[1], &e.y);
  const int64_t at = (int64_t)(a.chunk_base + w.chunk) * a.npad + w.obs0;
  *reinterpret_cast<double2*>(a.zdiv_next_mant + at) = m;
  *reinterpret_cast<int2*>(a.zdiv_next_exp + at) = e;
}

// The last iteration of a statistics call.  The leaves' normalisers of this iteration are those the last
// hard_loop_light launch formed: the parent's P' that launch had and the P' of the iteration before
// this one are both the product of the same messages (the q of the step), so they differ by a factor per
// observation, which the normalisers do not see.  a.zdiv_next_*: that launch's products, a.z_rows of
// them (the parent's chunks of this group) - they become the rows of a.ll_part.
//
// The count sums: a wave takes a chunk of at most kHardSumGroups * 16 families and kHardSumBlocks
// blocks of 128 observations, one after the other, and keeps the sums of all its families in
// registers until the end (half the per-wave sums to write, and to read back, that a wave per
// 128 observations would leave).  a.chunks / a.chunk_len: the launch's own cut of the families;
// a.slots: per-wave sum slots per entry; em_partial[k][d * 2 + y][slot]: the wave's share of G[y][d]
// (see hard_count), not formed when a.em_partial is null.
constexpr int32_t kHardSumGroups = 8;  // groups of 16 families per chunk, at most
constexpr int32_t kHardSumBlocks = 2;  // blocks of 128 observations per wave

// waves (and sum slots per entry) per chunk of families, blocks of the launch
inline int32_t hard_sum_slots(int64_t ntiles) { return (int32_t)((ntiles * 2 + kHardSumBlocks - 1) / kHardSumBlocks); }
inline uint32_t hard_sum_blocks(int32_t chunks, int64_t ntiles) {
  return (uint32_t)(chunks * ((hard_sum_slots(ntiles) + kWavesPerBlock - 1) / kWavesPerBlock));
}

template <int D0>
__global__ void __launch_bounds__(256) hard_reduce_mfma(const float* __restrict__ cpts, LeafFamilyArgs<float> a) {
  static_assert(D0 <= 16, "one 16-row operand holds the parent's states");
  constexpr int32_t V = 2, E = D0 * 2;
  const float* tab = a.hard_tab;
  // the wave's chunk of families and its blocks of observations
  const int32_t per_chunk = a.slots;
  const int32_t blocks_per_chunk = (per_chunk + kWavesPerBlock - 1) / kWavesPerBlock;
  const int32_t chunk = (int32_t)blockIdx.x / blocks_per_chunk;
  const int32_t slot = ((int32_t)blockIdx.x % blocks_per_chunk) * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (slot >= per_chunk) return;
  const int32_t parts = (int32_t)(a.ntiles * 2);  // blocks of 128 observations there are
  HardWave<D0> w;  // (placed on one block of observations after the other)
  auto place = [&](int32_t q) {
    const int32_t item = slot * kHardSumBlocks + q;
    w.lane = threadIdx.x & 63;
    w.chunk = chunk;
    w.t = item >> 1;
    w.part = item & 1;
    w.col = w.part * 128 + w.lane * V;
    w.obs0 = w.t * 256 + w.col;
    w.first = chunk * a.chunk_len;
    w.last = w.first + a.chunk_len < a.K ? w.first + a.chunk_len : a.K;
    w.states = a.src_base;
    w.stride = a.src_stride;
    w.at = w.obs0 + 1 >= a.N ? a.N - 2 : w.obs0;
    w.active = true;
    return item < parts;
  };

  // ---- the normalisers' rows: chunk c of the launch converts the parent's chunks c, c + a.chunks, ..
  // (all loads first: one memory latency for the lot)
  {
    constexpr int32_t kMost = 4;  // rows per chunk of the launch at a time
    for (int32_t row0 = chunk; row0 < a.z_rows; row0 += kMost * a.chunks) {
      double2 m[kHardSumBlocks][kMost];
      int2 e[kHardSumBlocks][kMost];
#pragma unroll
      for (int32_t q = 0; q < kHardSumBlocks; ++q) {
        const bool there = place(q);
#pragma unroll
        for (int32_t u = 0; u < kMost; ++u) {
          const int32_t row = row0 + u * a.chunks;
          const int64_t at = (int64_t)(a.chunk_base + (there && row < a.z_rows ? row : chunk)) * a.npad + (there ? w.obs0 : 0);
          m[q][u] = *reinterpret_cast<const double2*>(a.zdiv_next_mant + at);
          e[q][u] = *reinterpret_cast<const int2*>(a.zdiv_next_exp + at);
        }
      }
#pragma unroll
      for (int32_t q = 0; q < kHardSumBlocks; ++q) {
        if (!place(q)) break;
#pragma unroll
        for (int32_t u = 0; u < kMost; ++u) {
          const int32_t row = row0 + u * a.chunks;
          if (row >= a.z_rows) break;
          double2 ll;  // (the mantissa lies in [0.5, 1): its logarithm in float is good to 1e-7 absolute)
          ll.x = ((double)e[q][u].x + (double)__builtin_amdgcn_logf((float)m[q][u].x)) * 0.6931471805599453;
          ll.y = ((double)e[q][u].y + (double)__builtin_amdgcn_logf((float)m[q][u].y)) * 0.6931471805599453;
          *reinterpret_cast<double2*>(a.ll_part + (int64_t)row * a.npad + w.obs0) = ll;
        }
      }
    }
  }
  if (!a.em_partial) return;

  // ---- the count sums
  const int32_t op_i = (threadIdx.x & 63) & 15, op_kk = (threadIdx.x & 63) >> 4;
  hard_float4 ones[kHardSumGroups];  // sum_n b[4 kk + x][n] [n has form 1 for family 16 g + j], over all the wave's blocks
#pragma unroll
  for (int32_t g = 0; g < kHardSumGroups; ++g) ones[g] = hard_float4{0.f, 0.f, 0.f, 0.f};
  float rest[4] = {0.f, 0.f, 0.f, 0.f};  // sum_n b[4 kk + x][n]: what a family's form 0 gets when form 1 has had its share
  bool saw_third = false;
  // P' of a block of observations in the operand layout, as it is in memory (all requests in flight together)
  auto load_raw = [&](int32_t q, double2* raw) {
    place(q);
    const int64_t n0 = w.t * 256 + w.part * 128 + op_kk * 32;
    const double* src = a.scaled + (int64_t)(op_i < D0 ? op_i : 0) * a.npad + n0;
#pragma unroll
    for (int32_t t = 0; t < 16; ++t) raw[t] = *reinterpret_cast<const double2*>(src + 2 * t);
  };
  double2 raw[16];
  load_raw(0, raw);
#pragma unroll 1
  for (int32_t q = 0; q < kHardSumBlocks; ++q) {
    if (!place(q)) break;
    const HardGroupLoader<D0, 0> loader(a, w);
    const int64_t op_n0 = loader.op_n0;
    HardGroupData<0> ring[kHardAhead + 1];
#pragma unroll
    for (int32_t g = 0; g <= kHardAhead; ++g) loader.load(w.first + g * kHardGroup, ring[g]);
    // b as the left operand of the products: lane (i = lane % 16, kk) holds state i of observations
    // 32 kk .. 32 kk + 31 of the block (zero for the states and observations there are not)
    float p_op[32];
    {
#pragma unroll
      for (int32_t t = 0; t < 32; t += 2) {
        p_op[t] = (op_i < D0 && op_n0 + t < a.N) ? (float)raw[t >> 1].x : 0.f;
        p_op[t + 1] = (op_i < D0 && op_n0 + t + 1 < a.N) ? (float)raw[t >> 1].y : 0.f;
      }
      // converted here and now (left alone the conversions sink into the loop, and the doubles stay live)
#pragma unroll
      for (int32_t t = 0; t < 32; ++t) asm volatile("" : "+v"(p_op[t]));
      __builtin_amdgcn_sched_barrier(0);
      // the next block's P' is on its way while this one is worked on (npad covers whole tiles: the block
      // after the last one reads the last one's again)
      if (q + 1 < kHardSumBlocks) {
        load_raw(slot * kHardSumBlocks + q + 1 < parts ? q + 1 : q, raw);
        place(q);
      }
      float total = 0.f;
#pragma unroll
      for (int32_t t = 0; t < 32; ++t) {
        const float sum = hard_row_sum(p_op[t]);
        p_op[t] *= op_n0 + t < a.N ? fast_rcp(sum) : 0.f;
        total += p_op[t];
        if (t % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // (four sums at a time, not 32)
      }
      // over the four lanes of the state, then to the lanes that will hold the state's sums
      total += __shfl_xor(total, 16, 64);
      total += __shfl_xor(total, 32, 64);
#pragma unroll
      for (int32_t x = 0; x < 4; ++x) rest[x] += __shfl(total, op_kk * 4 + x, 64);
    }
#pragma unroll
    for (int32_t g = 0; g < kHardSumGroups; ++g) {
      const int32_t kg = w.first + g * kHardGroup;
      if (kg >= w.last) break;
      HardGroupData<0>& cur = ring[g % (kHardAhead + 1)];
      saw_third |= loader.settle(kg, cur);
      // two chains of products (a product that continues the one before it has to wait for it)
      hard_float4 one_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int32_t o = 0; o < 8; ++o) {
        const uint32_t four = cur.bytes[o];
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) {
          const float is_one = ((four >> (8 * r)) & 0xffu) == 1u ? 1.f : 0.f;
          if (r & 1)
            one_b = __builtin_amdgcn_mfma_f32_16x16x4f32(p_op[4 * o + r], is_one, one_b, 0, 0, 0);
          else
            ones[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(p_op[4 * o + r], is_one, ones[g], 0, 0, 0);
        }
      }
      ones[g] += one_b;
      if (g + kHardAhead + 1 < kHardSumGroups) loader.load(kg + (kHardAhead + 1) * kHardGroup, cur);
    }
  }
  // lane (j, kk) holds states 4 kk .. 4 kk + 3 of family 16 g + j
#pragma unroll
  for (int32_t g = 0; g < kHardSumGroups; ++g) {
    const int32_t k = chunk * a.chunk_len + g * kHardGroup + op_i;
    if (g * kHardGroup + op_i < a.chunk_len && k < a.K) {
#pragma unroll
      for (int32_t x = 0; x < 4; ++x) {
        const int32_t d = op_kk * 4 + x;
        if (d < D0) {
          float* dst = a.em_partial + ((int64_t)k * E + d * 2) * a.slots + slot;
          dst[0] = rest[x] - ones[g][x];
          dst[a.slots] = ones[g][x];
        }
      }
    }
  }
  if (saw_third) {
    // (rare) every lane looks at its own two observations of every family once more: an observation of the
    // third form goes to the sums of its own and out of form 0's, where `rest` has put it
    __threadfence();
#pragma unroll 1
    for (int32_t q = 0; q < kHardSumBlocks; ++q) {
      if (!place(q)) break;
#pragma unroll 1
      for (int32_t k = w.first; k < w.last; ++k) {
        const HardForms f(w.pair(k));
#pragma unroll
        for (int32_t v = 0; v < V; ++v)
          if (f.third[v] && w.obs0 + v < a.N)
            hard_third_add<D0>(tab, a.scaled, a.npad, a.K, k, w.obs0 + v, a.em_partial + (int64_t)k * E * a.slots + slot, a.slots);
      }
    }
  }
}

}  // namespace bn
