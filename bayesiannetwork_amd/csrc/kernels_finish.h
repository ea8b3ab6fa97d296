// The tail of a whole EM step (bn_em_step) of a star-shaped plan - one parentless wide node whose
// other families all belong to one fused one-parent leaf group - as ONE launch instead of four:
// log-likelihood partial sums (ll_partial_listed), the root's count sums (em_p1_fast without a
// parent), the leaf families' count sums finalised from the reduce launch's per-wave sums
// (em_finalize_slots) and the M-step (m_step_rows), each with the association of the kernel it
// stands for, so the step's results are bit for bit those of the separate launches.
//
// Blocks take roles by index: [0, ll_blocks) sum normalisers, [ll_blocks, ll_blocks + root_blocks)
// sum the root's posterior over their tile ranges, the rest finalise one CPT row per wave and apply
// the M-step to it.  What needs everybody's results - the final log-likelihood sum, the root's row -
// is done by the block that finishes last: every block counts itself in when its results are in
// memory, and the one that sees the count complete carries on (results travel through relaxed
// agent-scope atomics, as between the blocks of star_em_step: no fence, no cache write-back).
#pragma once

#include "kernels_fast.h"
#include "kernels_star.h"

namespace bn {

template <typename T>
struct FinishArgs {
  const T* arena;
  // log-likelihood partial sums (the arguments of ll_partial_listed)
  const int64_t* c_offs;
  int64_t n_listed;
  const double* ll_rows;
  int64_t ll_row_count;
  int64_t N, npad;
  int32_t ll_blocks;
  double* ll_partial;  // [ll_blocks]
  // the root's count sums: its variable's message to its own factor, summed over the observations
  int64_t root_in_off;    // that message array
  int64_t root_cpt_off;   // the root's CPT in the flat buffer
  int32_t root_states;
  int64_t ntiles;
  int32_t root_chunks;
  double* root_partial;   // [root_states][root_chunks]
  // the leaf families' count sums
  const T* em_partial;     // [K][entries][slots]
  const int64_t* cpt_off;  // [K]
  int32_t K, D0, S, slots;
  // M-step
  T* cpts;       // the plan's snapshot: old parameters in, new ones out
  T* new_cpts;   // the caller's copy (may be null)
  double* accum; // count sums, flat CPT layout (may be null)
  double learning_rate, regularization;
  double ll_scale;
  double* ll_out;  // may be null: the caller sums ll_partial itself
  uint32_t* arrived;  // zero between launches
  FormSums forms;     // tab != nullptr: em_partial holds sums by evidence form (two-state leaves)
  // 1: the statistics only (a rank's share of a sharded step: the sums go to accum and ll_out, the
  // parameters are left alone)
  int32_t statistics_only;
};

// one CPT row: target = counts / sum, regularised and blended as in m_step_rows (no fused multiply-add)
template <typename T>
__device__ __forceinline__ void finish_row(const FinishArgs<T>& a, int64_t off, int32_t len, const double* counts) {
#pragma clang fp contract(off)
  double sum = 0.0;
  for (int32_t j = 0; j < len; ++j) sum += counts[j];
  const double uniform = 1.0 / (double)len;
  for (int32_t j = 0; j < len; ++j) {
    double x = counts[j] / sum;
    if (a.regularization != 0.0) {
      const double p = (1.0 - a.regularization) * x;
      const double u = a.regularization * uniform;
      x = p + u;
    }
    if (a.learning_rate >= 0.0) {
      const double old = (1.0 - a.learning_rate) * (double)a.cpts[off + j];
      const double fresh = a.learning_rate * x;
      x = old + fresh;
    }
    a.cpts[off + j] = (T)x;
    if (a.new_cpts) a.new_cpts[off + j] = (T)x;
  }
}

constexpr int32_t kFinishMaxStates = 64;  // of the root, and of a leaf

constexpr int32_t kFinishLlPerBlock = 8;  // log-likelihood partial sums one block computes, one after the other

template <typename T>
__global__ void __launch_bounds__(256) em_finish(FinishArgs<T> a) {
  // every product here is rounded on its own, as in the separate kernels, where it travels through
  // memory before it is added to anything
#pragma clang fp contract(off)
  __shared__ double red[kFinishLlPerBlock * 256];
  __shared__ double root_counts[kFinishMaxStates];
  __shared__ int32_t last_block;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t b = (int32_t)blockIdx.x;
  const int32_t ll_role = (a.ll_blocks + kFinishLlPerBlock - 1) / kFinishLlPerBlock;
  const int32_t root_role = (a.root_chunks + kWavesPerBlock - 1) / kWavesPerBlock;

  if (b >= ll_role + root_role) {
    // ---- em_finalize_slots + m_step_rows: one wave per CPT row of the leaf families; nobody waits for these
    const int64_t row = (int64_t)(b - ll_role - root_role) * kWavesPerBlock + wave;
    if (row >= (int64_t)a.K * a.D0) return;
    const int32_t k = (int32_t)(row / a.D0), d = (int32_t)(row % a.D0);
    const int64_t off = a.cpt_off[k] + (int64_t)d * a.S;
    auto entry_sum = [&](int32_t j) {
      const T* src = a.em_partial + ((int64_t)k * a.D0 * a.S + (int64_t)d * a.S + j) * a.slots;
      double acc = 0.0;
      for (int32_t sl = lane; sl < a.slots; sl += 64) acc += (double)src[sl];
#pragma unroll
      for (int32_t o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      return acc;
    };
    if (a.S <= 4) {  // the usual leaves: the row stays in registers
      double counts[4];
#pragma unroll
      for (int32_t j = 0; j < 4; ++j) counts[j] = j < a.S ? entry_sum(j) : 0.0;
      if (a.forms.tab) {
        const float* rec = a.forms.tab + (size_t)k * (kHardRows * a.D0);
        const double g0 = counts[0] + hard_third_sum(a.forms.tab, a.K, a.D0, k, 0, d);
        const double g1 = counts[1] + hard_third_sum(a.forms.tab, a.K, a.D0, k, 1, d);
#pragma unroll
        for (int32_t j = 0; j < 2; ++j)
          counts[j] = hard_count(g0, g1, rec[kHardRq * a.D0 + d], rec[(kHardRq + 1) * a.D0 + d], j, a.forms.rows,
                                 (double)a.cpts[off + j]);
      } else {
#pragma unroll
        for (int32_t j = 0; j < 4; ++j)
          if (j < a.S) counts[j] = counts[j] * (double)a.cpts[off + j];
      }
      if (lane == 0) {
        if (a.accum) {
#pragma unroll
          for (int32_t j = 0; j < 4; ++j)
            if (j < a.S) a.accum[off + j] = counts[j];
        }
        if (!a.statistics_only) finish_row(a, off, a.S, counts);
      }
    } else {
      double counts[kFinishMaxStates];
      for (int32_t j = 0; j < a.S; ++j) counts[j] = entry_sum(j) * (double)a.cpts[off + j];
      if (lane == 0) {
        if (a.accum)
          for (int32_t j = 0; j < a.S; ++j) a.accum[off + j] = counts[j];
        if (!a.statistics_only) finish_row(a, off, a.S, counts);
      }
    }
    return;
  }

  if (b < ll_role) {
    // ---- ll_partial_listed: the partial sums that blocks b * 8 .. b * 8 + 7 of ll_blocks would form, side
    // by side (their loads are in flight together), each with that kernel's association
    const int64_t per_row = ((a.npad + 255) / 256) * 256;
    const int64_t total = (a.n_listed + a.ll_row_count) * per_row;
    const int32_t part0 = b * kFinishLlPerBlock;
    double acc[kFinishLlPerBlock];
#pragma unroll
    for (int32_t u = 0; u < kFinishLlPerBlock; ++u) acc[u] = 0.0;
    for (int64_t base = (int64_t)threadIdx.x; base < total; base += (int64_t)a.ll_blocks * 256) {
#pragma unroll
      for (int32_t u = 0; u < kFinishLlPerBlock; ++u) {
        const int64_t idx = base + (int64_t)(part0 + u) * 256;
        if (part0 + u >= a.ll_blocks || idx >= total) continue;
        const int64_t row = idx / per_row;
        const int64_t n = idx - row * per_row;
        if (n >= a.N) continue;
        acc[u] += row < a.n_listed ? (double)a.arena[a.c_offs[row] + n] : a.ll_rows[(row - a.n_listed) * a.npad + n];
      }
    }
#pragma unroll
    for (int32_t u = 0; u < kFinishLlPerBlock; ++u) red[u * 256 + threadIdx.x] = acc[u];
    __syncthreads();
    for (int32_t step = 128; step > 0; step >>= 1) {
      if ((int32_t)threadIdx.x < step) {
#pragma unroll
        for (int32_t u = 0; u < kFinishLlPerBlock; ++u) red[u * 256 + threadIdx.x] += red[u * 256 + threadIdx.x + step];
      }
      __syncthreads();
    }
    if ((int32_t)threadIdx.x < kFinishLlPerBlock && part0 + (int32_t)threadIdx.x < a.ll_blocks)
      shared_store(a.ll_partial + part0 + threadIdx.x, red[threadIdx.x * 256]);
  } else {
    // ---- em_p1_fast without a parent: wave = chunk of tiles, per-lane running sums per state
    const int32_t chunk = (b - ll_role) * kWavesPerBlock + wave;
    if (chunk < a.root_chunks) {
      const T* in_own = a.arena + a.root_in_off;
      const int64_t per = (a.ntiles + a.root_chunks - 1) / a.root_chunks;
      const int64_t t0 = chunk * per;
      const int64_t t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
      for (int32_t j0 = 0; j0 < a.root_states; j0 += 4) {  // four states at a time: their loads are in flight together
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t t = t0; t < t1; ++t) {
          rowv<T> row[4];
#pragma unroll
          for (int32_t u = 0; u < 4; ++u)
            row[u] = loadw<T>(in_own, t * a.root_states + (j0 + u < a.root_states ? j0 + u : j0), lane);
          const int64_t n0 = t * tile_of<T>::value + lane * rowv<T>::W;
#pragma unroll
          for (int32_t u = 0; u < 4; ++u)
#pragma unroll
            for (int32_t v = 0; v < rowv<T>::W; ++v) {
              const double live = (n0 + v < a.N) ? 1.0 : 0.0;
              acc[u] = fma(1.0 * live, (double)row[u].v[v], acc[u]);
            }
        }
#pragma unroll
        for (int32_t u = 0; u < 4; ++u) {
          double x = acc[u];
#pragma unroll
          for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
          if (lane == 0 && j0 + u < a.root_states) shared_store(a.root_partial + (int64_t)(j0 + u) * a.root_chunks + chunk, x);
        }
      }
    }
  }

  // ---- who is last (of the blocks whose results the two sums below need)?
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t before = __hip_atomic_fetch_add(a.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_block = before + 1 == (uint32_t)(ll_role + root_role);
    if (last_block) __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_block) return;

  // ---- the root's row: its chunk sums as em_finalize_slots adds listed entries (one wave per entry)
  for (int32_t j = wave; j < a.root_states; j += kWavesPerBlock) {
    double acc = 0.0;
    for (int32_t c = lane; c < a.root_chunks; c += 64) acc += shared_load(a.root_partial + (int64_t)j * a.root_chunks + c);
#pragma unroll
    for (int32_t o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) root_counts[j] = acc * (double)a.cpts[a.root_cpt_off + j];
  }
  // ---- the log-likelihood: the partial sums in the association of sum_partials
  double acc = 0.0;
  for (int32_t i = threadIdx.x; i < a.ll_blocks; i += 256) acc += shared_load(a.ll_partial + i);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int32_t step = 128; step > 0; step >>= 1) {
    if ((int32_t)threadIdx.x < step) red[threadIdx.x] += red[threadIdx.x + step];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (a.ll_out) a.ll_out[0] = red[0] * a.ll_scale;
    if (a.accum)
      for (int32_t j = 0; j < a.root_states; ++j) a.accum[a.root_cpt_off + j] = root_counts[j];
    if (!a.statistics_only) finish_row(a, a.root_cpt_off, a.root_states, root_counts);
  }
}

}  // namespace bn
