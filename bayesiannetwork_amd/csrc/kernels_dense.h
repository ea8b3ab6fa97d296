// High-cardinality (32 / 64 state) single-parent families: the two factor contractions and
// the EM count matrix are real GEMMs and run on the matrix cores with the exact-float32
// instruction v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain in k order, so numerics match
// the VALU kernels); the variable pass for many-state nodes streams state by state.
//
// The tile layout [tile][state][256 observations] is already the operand shape MFMA wants:
// with OUTPUT^T[state_out][obs] = W[state_out][state_in] * MSG^T[state_in][obs]
//   B operand  lane l holds MSG^T[k = l>>5][j = l&31]  -> 32 consecutive observations of one
//              message row per half-wave: whole 128-byte lines straight from HBM,
//   A operand  lane l holds W[i = l&31][k = l>>5]      -> the CPT (or its transpose), staged
//              once per workgroup in LDS with a 65-float row pitch (conflict-free both ways),
//   D          column = observation, rows = output states -> each accumulator register is
//              stored as 128-byte runs of consecutive observations.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_fast.h"

namespace bn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int32_t kCptPitch = 65;  // LDS row pitch (floats) for a <= 64-column CPT

__device__ __forceinline__ int32_t mfma_row(int32_t reg, int32_t lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// ---------------------------------------------------------------------------------
// Factor pass, one parent (D0 states) and own variable (S states), D0, S in {32, 64}.
// One workgroup (4 waves) per (node, tile); wave w owns observations [64w, 64w+64).
//   to own variable  out_own^T[b][n] = sum_a cpt[a][b] * in_par^T[a][n]
//   to parent        out_par^T[a][n] = sum_b cpt[a][b] * in_own^T[b][n]
// At least three waves per SIMD (<= 168 VGPRs): with the k loops fully unrolled the compiler would
// take a few registers more and drop to two, which costs more than the tighter allocation does.
template <int D0, int S>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) factor_p1_mfma(float* __restrict__ arena, const float* __restrict__ cpts,
                                                     const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ out_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K,
                                                     int64_t ntiles) {
  __shared__ float w_lds[D0 * kCptPitch];
  const int64_t item = blockIdx.x;
  const int32_t k = (int32_t)(item / ntiles);
  const int64_t t = item % ntiles;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t half = lane >> 5, col = lane & 31;

  const float* cpt = cpts + cpt_off[k];
  for (int32_t e = threadIdx.x; e < D0 * S; e += 256) w_lds[(e / S) * kCptPitch + (e % S)] = cpt[e];
  __syncthreads();

  const float* in_par = arena + in_off[2 * k] + t * D0 * kTile + wave * 64;
  const float* in_own = arena + in_off[2 * k + 1] + t * S * kTile + wave * 64;
  float* out_par = arena + out_off[2 * k] + t * D0 * kTile + wave * 64;
  float* out_own = arena + out_off[2 * k + 1] + t * S * kTile + wave * 64;

  // ---- to own variable: i = b (S/32 tiles), k = a (D0/2 steps), j = observation (2 tiles)
  {
    f32x16 acc[S / 32][2];
#pragma unroll
    for (int32_t it = 0; it < S / 32; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
#pragma unroll
    for (int32_t ks = 0; ks < D0 / 2; ++ks) {
      const int32_t a = 2 * ks + half;
      float bfrag[2];
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt) bfrag[jt] = in_par[(int64_t)a * kTile + 32 * jt + col];
#pragma unroll
      for (int32_t it = 0; it < S / 32; ++it) {
        const float afrag = w_lds[a * kCptPitch + 32 * it + col];  // W[i=b][k=a] = cpt[a][b]
#pragma unroll
        for (int32_t jt = 0; jt < 2; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag[jt], acc[it][jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int32_t it = 0; it < S / 32; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r)
          out_own[(int64_t)(32 * it + mfma_row(r, lane)) * kTile + 32 * jt + col] = acc[it][jt][r];
  }
  // ---- to parent: i = a (D0/32 tiles), k = b (S/2 steps), j = observation
  {
    f32x16 acc[D0 / 32][2];
#pragma unroll
    for (int32_t it = 0; it < D0 / 32; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r) acc[it][jt][r] = 0.f;
#pragma unroll
    for (int32_t ks = 0; ks < S / 2; ++ks) {
      const int32_t b = 2 * ks + half;
      float bfrag[2];
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt) bfrag[jt] = in_own[(int64_t)b * kTile + 32 * jt + col];
#pragma unroll
      for (int32_t it = 0; it < D0 / 32; ++it) {
        const float afrag = w_lds[(32 * it + col) * kCptPitch + b];  // W[i=a][k=b] = cpt[a][b]
#pragma unroll
        for (int32_t jt = 0; jt < 2; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag[jt], acc[it][jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int32_t it = 0; it < D0 / 32; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r)
          out_par[(int64_t)(32 * it + mfma_row(r, lane)) * kTile + 32 * jt + col] = acc[it][jt][r];
  }
}

// ---------------------------------------------------------------------------------
// One whole flooding iteration for a dense family (parent P with D0 states -> X with S states) in
// one kernel: the variable passes of P and X, as far as this family's factor needs them, folded
// into the factor pass.  The factor's two inputs are not read as stored messages but formed from
// what P and X receive from elsewhere:
//   u_P[a] = prod of P's inputs other than this factor's message  (its own factor's message, the
//            other child's factor message, its observation): 1 or 2 arrays,
//   u_X[b] = prod of X's inputs other than this factor's message  (its children's factor messages
//            or its observation): 0, 1 or 2 arrays,
// normalised as the variable pass does (VariableNodeGroup.calculate_outputs, spa_v3/factor_graph.py:
// 112-144): P's message to a child's factor by its own sum over the states, X's message to its own
// factor by the sum of X's FULL product, i.e. sum_b u_X[b] * (this factor's previous message to
// X)[b] — that message is read for the normaliser only.  Then contracted with the CPT as in
// factor_p1_mfma (FactorNodeGroup.calculate_outputs, :251-298).  Normalisation commutes with the contraction, so the unnormalised products are the
// MFMA operands — they stay in registers, 64 per lane and side — and the accumulators are scaled
// by 1 / sum afterwards.  With two inputs each is first scaled by a power of two so that its
// largest entry per observation is in [0.5, 1): a product of two float messages that are tiny
// throughout does not underflow.
// Only factor -> variable messages exist between such iterations (a variable -> factor message
// costs a write and a read of 64 floats per observation and edge: 2.3 KB per family, observation and
// iteration in two passes, 1.3 KB here); they alternate between the two arrays of every edge, so
// no launch reads what it writes.   Same geometry as factor_p1_mfma.
struct TreeFamilyTables {
  const int64_t* in_par;  // [K][2] arrays whose product is u_P (kNoArray = none)
  const int64_t* in_own;  // [K][2] arrays whose product is u_X
  const int64_t* own_prev;  // [K] the factor's previous message to X (weights of X's normaliser)
  const int64_t* out;     // [K][2] where the messages to P and to X go
  const int64_t* cpt_off;
};

// u[ks] = product of the (one or two) arrays for state 2 * ks + half and observation column `col` of
// the 32-column block at `base` (already offset to tile, wave and column block); inv = 1 / sum
// over the states of u (times the weight array's entry, when there is one).
template <int KD>
__device__ __forceinline__ void tree_operand(const float* __restrict__ arena, int64_t first, int64_t second,
                                             int64_t weight, int64_t base, int32_t half, int32_t col,
                                             float (&u)[KD / 2], float& inv) {
  // 2^-exponent of the largest entry of the observation's column over all KD states (this lane holds
  // every other state, the lane 32 further on the rest)
  auto column_scale = [&](float mx) {
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    int e = 0;
    (void)frexpf(mx, &e);
    return ldexpf(1.0f, mx > 0.f ? -e : 0);
  };
  // the weights travel together with the inputs (one memory round trip for all three arrays)
  float w[KD / 2];
  if (weight != kNoArray) {
    const float* wgt = arena + weight + base + half * kTile + col;
#pragma unroll
    for (int32_t ks = 0; ks < KD / 2; ++ks) w[ks] = wgt[(int64_t)(2 * ks) * kTile];
  }
  if (first == kNoArray) {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 2; ++ks) u[ks] = 1.0f;
  } else {
    const float* a0 = arena + first + base + half * kTile + col;
#pragma unroll
    for (int32_t ks = 0; ks < KD / 2; ++ks) u[ks] = a0[(int64_t)(2 * ks) * kTile];
    if (second != kNoArray) {
      const float* a1 = arena + second + base + half * kTile + col;
      float v[KD / 2];
#pragma unroll
      for (int32_t ks = 0; ks < KD / 2; ++ks) v[ks] = a1[(int64_t)(2 * ks) * kTile];
      float mu = 0.f, mv = 0.f;
#pragma unroll
      for (int32_t ks = 0; ks < KD / 2; ++ks) {
        mu = fmaxf(mu, u[ks]);
        mv = fmaxf(mv, v[ks]);
      }
      // each input scaled so that its largest entry is in [0.5, 1) before the two are multiplied
      const float su = column_scale(mu), sv = column_scale(mv);
#pragma unroll
      for (int32_t ks = 0; ks < KD / 2; ++ks) u[ks] = (u[ks] * su) * (v[ks] * sv);
    }
  }
  double sum = 0.0;
  if (weight != kNoArray) {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 2; ++ks) sum += (double)(u[ks] * w[ks]);
  } else {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 2; ++ks) sum += (double)u[ks];
  }
  sum += __shfl_xor(sum, 32, 64);
  inv = (float)(1.0 / sum);
}

// EXACT: X's message to its own factor is normalised as the reference does (weights = the factor's
// previous message to X).  Without it that message is normalised by its own sum, which changes the
// scale of this launch's message to the parent and nothing else; the scale of a message does not
// survive the next normalisation, so only the last launch before a regular iteration (whose
// normalisers the log-likelihood reads) has to be EXACT, and the others save reading one array.
template <int D0, int S, bool EXACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EXACT && D0 == 64 && S == 64 ? 2 : 3))) factor_tree_mfma(float* __restrict__ arena, const float* __restrict__ cpts,
                                                       TreeFamilyTables tab, int32_t K, int64_t ntiles) {
  __shared__ float w_lds[D0 * kCptPitch];
  const int64_t item = blockIdx.x;
  const int32_t k = (int32_t)(item / ntiles);
  const int64_t t = item % ntiles;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t half = lane >> 5, col = lane & 31;

  const float* cpt = cpts + tab.cpt_off[k];
  for (int32_t e = threadIdx.x; e < D0 * S; e += 256) w_lds[(e / S) * kCptPitch + (e % S)] = cpt[e];
  __syncthreads();
  const int64_t in_par0 = tab.in_par[2 * k], in_par1 = tab.in_par[2 * k + 1];
  const int64_t in_own0 = tab.in_own[2 * k], in_own1 = tab.in_own[2 * k + 1];
  const int64_t own_prev = EXACT ? tab.own_prev[k] : kNoArray;
  const int64_t out_par_off = tab.out[2 * k], out_own_off = tab.out[2 * k + 1];

  // the wave's 64 observations as two blocks of 32 columns, one after the other (one block's
  // operands and accumulators in registers at a time)
#pragma unroll 1
  for (int32_t jt = 0; jt < 2; ++jt) {
    const int32_t column = wave * 64 + 32 * jt;
    // ---- to own variable: out_own^T[b][n] = sum_a cpt[a][b] * u_P[a][n] / sum_a u_P[a][n]
    {
      float u[D0 / 2], inv;
      tree_operand<D0>(arena, in_par0, in_par1, kNoArray, t * D0 * kTile + column, half, col, u, inv);
      __builtin_amdgcn_sched_barrier(0);  // keep the phases apart: the scheduler otherwise pulls the next
                                          // phase's loads up and runs out of registers
      f32x16 acc[S / 32];
#pragma unroll
      for (int32_t it = 0; it < S / 32; ++it)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r) acc[it][r] = 0.f;
#pragma unroll
      for (int32_t ks = 0; ks < D0 / 2; ++ks) {
        const int32_t a = 2 * ks + half;
#pragma unroll
        for (int32_t it = 0; it < S / 32; ++it)  // W[i=b][k=a] = cpt[a][b]
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_lds[a * kCptPitch + 32 * it + col], u[ks], acc[it], 0, 0, 0);
      }
      float* out_own = arena + out_own_off + t * S * kTile + column + col;
#pragma unroll
      for (int32_t it = 0; it < S / 32; ++it)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r)
          out_own[(int64_t)(32 * it + mfma_row(r, lane)) * kTile] = acc[it][r] * inv;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- to parent: out_par^T[a][n] = sum_b cpt[a][b] * u_X[b][n] / (normaliser of X's message)
    {
      float u[S / 2], inv;
      tree_operand<S>(arena, in_own0, in_own1, own_prev, t * S * kTile + column, half, col, u, inv);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[D0 / 32];
#pragma unroll
      for (int32_t it = 0; it < D0 / 32; ++it)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r) acc[it][r] = 0.f;
#pragma unroll
      for (int32_t ks = 0; ks < S / 2; ++ks) {
        const int32_t b = 2 * ks + half;
#pragma unroll
        for (int32_t it = 0; it < D0 / 32; ++it)  // W[i=a][k=b] = cpt[a][b]
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_lds[(32 * it + col) * kCptPitch + b], u[ks], acc[it], 0, 0, 0);
      }
      float* out_par = arena + out_par_off + t * D0 * kTile + column + col;
#pragma unroll
      for (int32_t it = 0; it < D0 / 32; ++it)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r)
          out_par[(int64_t)(32 * it + mfma_row(r, lane)) * kTile] = acc[it][r] * inv;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------
// EM count matrix for the same families: stat[a][b] = sum_n in_par[n][a] * in_own[n][b]
// (the CPT factor is applied by em_finalize) = a [D0 x N] x [N x S] GEMM with the
// observations as the k dimension.  One workgroup per (node, chunk of tiles); 64 observations
// of both messages are staged in LDS per step (coalesced 16-byte loads, 65-float pitch), the
// four waves own the four 32x32 quadrants of the count matrix (D0 = S = 64) or split it
// accordingly.  Padding observations are zeroed while staging.
template <int D0, int S>
__global__ void __launch_bounds__(256) em_p1_mfma(const float* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                 const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                 int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  __shared__ float par_lds[D0 * kCptPitch];
  __shared__ float own_lds[S * kCptPitch];
  const int32_t k = blockIdx.x / nchunks;
  const int32_t chunk = blockIdx.x % nchunks;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t half = lane >> 5, col = lane & 31;
  constexpr int32_t kQuads = (D0 / 32) * (S / 32);   // 32x32 output tiles
  constexpr int32_t kPerWave = (kQuads + 3) / 4;     // tiles per wave

  const float* in_par = arena + in_off[2 * k];
  const float* in_own = arena + in_off[2 * k + 1];
  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  // The matrix instruction accumulates in float: a tile's 256 observations at a time, then the
  // tile's sums go into double running sums (a float running sum over a chunk of thousands of
  // observations loses up to 4e-5 relative on the small entries of a 64 x 64 count matrix).
  f32x16 acc[kPerWave];
  double total[kPerWave][16];
#pragma unroll
  for (int32_t q = 0; q < kPerWave; ++q)
#pragma unroll
    for (int32_t r = 0; r < 16; ++r) {
      acc[q][r] = 0.f;
      total[q][r] = 0.0;
    }

  for (int64_t t = t0; t < t1; ++t) {
    if (t > t0) {
#pragma unroll
      for (int32_t q = 0; q < kPerWave; ++q)
#pragma unroll
        for (int32_t r = 0; r < 16; ++r) {
          total[q][r] += (double)acc[q][r];
          acc[q][r] = 0.f;
        }
    }
    for (int32_t quarter = 0; quarter < 4; ++quarter) {
      const int64_t n0 = t * kTile + quarter * 64;
      if (n0 >= N) break;
      __syncthreads();
      // stage [rows][64 observations] of both messages; thread -> (row, 4 observations)
      for (int32_t e = threadIdx.x; e < D0 * 16; e += 256) {
        const int32_t row = e >> 4, c4 = (e & 15) * 4;
        const float4 x = *reinterpret_cast<const float4*>(in_par + (t * D0 + row) * kTile + quarter * 64 + c4);
        float* dst = par_lds + row * kCptPitch + c4;
        dst[0] = n0 + c4 + 0 < N ? x.x : 0.f;
        dst[1] = n0 + c4 + 1 < N ? x.y : 0.f;
        dst[2] = n0 + c4 + 2 < N ? x.z : 0.f;
        dst[3] = n0 + c4 + 3 < N ? x.w : 0.f;
      }
      for (int32_t e = threadIdx.x; e < S * 16; e += 256) {
        const int32_t row = e >> 4, c4 = (e & 15) * 4;
        const float4 x = *reinterpret_cast<const float4*>(in_own + (t * S + row) * kTile + quarter * 64 + c4);
        float* dst = own_lds + row * kCptPitch + c4;
        dst[0] = x.x;
        dst[1] = x.y;
        dst[2] = x.z;
        dst[3] = x.w;
      }
      __syncthreads();
#pragma unroll
      for (int32_t q = 0; q < kPerWave; ++q) {
        const int32_t quad = wave * kPerWave + q;
        if (quad < kQuads) {
          const int32_t it = quad / (S / 32), jt = quad % (S / 32);
#pragma unroll 8
          for (int32_t ks = 0; ks < 32; ++ks) {
            const int32_t n = 2 * ks + half;
            const float afrag = par_lds[(32 * it + col) * kCptPitch + n];  // A[i=a][k=n]
            const float bfrag = own_lds[(32 * jt + col) * kCptPitch + n];  // B[k=n][j=b]
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag, bfrag, acc[q], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int32_t q = 0; q < kPerWave; ++q) {
    const int32_t quad = wave * kPerWave + q;
    if (quad < kQuads) {
      const int32_t it = quad / (S / 32), jt = quad % (S / 32);
#pragma unroll
      for (int32_t r = 0; r < 16; ++r) {
        const int32_t a = 32 * it + mfma_row(r, lane), b = 32 * jt + col;
        partial[(cpt_off[k] + (int64_t)a * S + b) * nchunks + chunk] = total[q][r] + (double)acc[q][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Variable pass for nodes with 8*SQ states and a small fan-in I (<= 4), every input read once:
// a block of eight waves owns (node, 128 observations); wave w keeps states [w*SQ, (w+1)*SQ) of
// all inputs in registers (two observations per lane, 8-byte accesses), the waves exchange their
// partial normalisers through LDS, and each writes the outgoing messages of its own states.
// Same quantities as variable_pass_generic.   grid.x = K * ntiles * 2, block = 512
template <int I, int SQ>
__global__ void __launch_bounds__(512) variable_split_fast(float* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                          const int64_t* __restrict__ out_off,
                                                          const int64_t* __restrict__ c_off, int32_t K,
                                                          int64_t ntiles, int32_t fresh) {
  constexpr int32_t NW = 8, S = NW * SQ;
  __shared__ double part[NW][I + 1][2][64];
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t item = blockIdx.x;  // (node, tile, half tile)
  const int32_t k = (int32_t)(item / (ntiles * 2));
  const int64_t rest = item % (ntiles * 2);
  const int64_t t = rest >> 1;
  const int32_t col = (int32_t)(rest & 1) * 128 + lane * 2;
  const float uniform = 1.0f / (float)S;

  float2 in[I][SQ];
  float* dst[I];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const int64_t o = out_off[(int64_t)k * I + i];
    dst[i] = o != kNoArray ? arena + o + (t * S + wave * SQ) * kTile + col : nullptr;
    const bool skip = fresh && o != kNoArray;  // a message slot that still holds its initial uniform value
    const float* src = arena + in_off[(int64_t)k * I + i] + (t * S + wave * SQ) * kTile + col;
#pragma unroll
    for (int32_t q = 0; q < SQ; ++q)
      in[i][q] = skip ? make_float2(uniform, uniform) : *reinterpret_cast<const float2*>(src + (int64_t)q * kTile);
  }
  // partial normalisers over this wave's states: z = sum_s prod_i in_i[s], sums_i = sum_s prod_{j != i} in_j[s]
  double z[2] = {0.0, 0.0}, sums[I][2];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) sums[i][0] = sums[i][1] = 0.0;
#pragma unroll
  for (int32_t q = 0; q < SQ; ++q)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double loo[I], pre = 1.0;
#pragma unroll
      for (int32_t i = 0; i < I; ++i) {
        loo[i] = pre;
        pre *= (double)(v ? in[i][q].y : in[i][q].x);
      }
      z[v] += pre;
      double suf = 1.0;
#pragma unroll
      for (int32_t i = I - 1; i >= 0; --i) {
        sums[i][v] += loo[i] * suf;
        suf *= (double)(v ? in[i][q].y : in[i][q].x);
      }
    }
#pragma unroll
  for (int32_t v = 0; v < 2; ++v) {
#pragma unroll
    for (int32_t i = 0; i < I; ++i) part[wave][i][v][lane] = sums[i][v];
    part[wave][I][v][lane] = z[v];
  }
  __syncthreads();
#pragma unroll
  for (int32_t v = 0; v < 2; ++v) {
    z[v] = 0.0;
#pragma unroll
    for (int32_t i = 0; i < I; ++i) sums[i][v] = 0.0;
#pragma unroll
    for (int32_t w = 0; w < NW; ++w) {
      z[v] += part[w][I][v][lane];
#pragma unroll
      for (int32_t i = 0; i < I; ++i) sums[i][v] += part[w][i][v][lane];
    }
  }
  if (wave == 0) {
    float2 c;
    int e = 0;
    double m = frexp(z[0], &e);
    c.x = (float)((double)e * 0.6931471805599453 + (double)logf((float)m));
    m = frexp(z[1], &e);
    c.y = (float)((double)e * 0.6931471805599453 + (double)logf((float)m));
    *reinterpret_cast<float2*>(arena + c_off[k] + t * kTile + col) = c;
  }
#pragma unroll
  for (int32_t v = 0; v < 2; ++v) {
#pragma unroll
    for (int32_t i = 0; i < I - 1; ++i) sums[i][v] = 1.0 / sums[i][v];
    sums[I - 1][v] = 1.0 / z[v];  // the message to the local factor is normalised by the full product
  }
#pragma unroll
  for (int32_t q = 0; q < SQ; ++q) {
    float2 out[I];
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double loo[I], pre = 1.0;
#pragma unroll
      for (int32_t i = 0; i < I; ++i) {
        loo[i] = pre;
        pre *= (double)(v ? in[i][q].y : in[i][q].x);
      }
      double suf = 1.0;
#pragma unroll
      for (int32_t i = I - 1; i >= 0; --i) {
        const float o = (float)(loo[i] * suf * sums[i][v]);
        if (v) out[i].y = o; else out[i].x = o;
        suf *= (double)(v ? in[i][q].y : in[i][q].x);
      }
    }
#pragma unroll
    for (int32_t i = 0; i < I; ++i)
      if (dst[i]) *reinterpret_cast<float2*>(dst[i] + (int64_t)q * kTile) = out[i];
  }
}

// =================================================================================
// float64 plans (the element type of the reference's own experiment scripts,
// dev_experiments/mnist/mnist.py:32): the same three kernels on v_mfma_f64_16x16x4_f64 and on
// mantissa / exponent products.  A tile holds 128 observations.
// =================================================================================
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int32_t kTile64 = 128;

// Factor pass, one parent, D0, S in {32, 64}, float64.  One workgroup (4 waves) per (node, tile);
// wave w owns observations [32 w, 32 w + 32) as two 16-wide column tiles.
//   A operand  lane l holds W[i = l & 15][k = l >> 4]   (CPT or its transpose, from LDS, pitch 65)
//   B operand  lane l holds MSG^T[k = l >> 4][j = l & 15]: 16 lanes read 128 consecutive bytes
//   D          register r of lane l is row 4 r + (l >> 4), column l & 15 (tools/ubench/mfma_f64_layout.hip)
template <int D0, int S>
__global__ void __launch_bounds__(256) factor_p1_mfma_f64(double* __restrict__ arena, const double* __restrict__ cpts,
                                                         const int64_t* __restrict__ in_off,
                                                         const int64_t* __restrict__ out_off,
                                                         const int64_t* __restrict__ cpt_off, int32_t K,
                                                         int64_t ntiles) {
  __shared__ double w_lds[D0 * kCptPitch];
  const int64_t item = blockIdx.x;
  const int32_t k = (int32_t)(item / ntiles);
  const int64_t t = item % ntiles;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t kq = lane >> 4, col = lane & 15;

  const double* cpt = cpts + cpt_off[k];
  for (int32_t e = threadIdx.x; e < D0 * S; e += 256) w_lds[(e / S) * kCptPitch + (e % S)] = cpt[e];
  __syncthreads();

  const double* in_par = arena + in_off[2 * k] + t * D0 * kTile64 + wave * 32;
  const double* in_own = arena + in_off[2 * k + 1] + t * S * kTile64 + wave * 32;
  double* out_par = arena + out_off[2 * k] + t * D0 * kTile64 + wave * 32;
  double* out_own = arena + out_off[2 * k + 1] + t * S * kTile64 + wave * 32;

  // ---- to own variable: out_own^T[b][n] = sum_a cpt[a][b] * in_par^T[a][n]
  {
    f64x4 acc[S / 16][2];
#pragma unroll
    for (int32_t it = 0; it < S / 16; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) acc[it][jt][r] = 0.0;
#pragma unroll
    for (int32_t ks = 0; ks < D0 / 4; ++ks) {
      const int32_t a = 4 * ks + kq;
      double bfrag[2];
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt) bfrag[jt] = in_par[(int64_t)a * kTile64 + 16 * jt + col];
#pragma unroll
      for (int32_t it = 0; it < S / 16; ++it) {
        const double afrag = w_lds[a * kCptPitch + 16 * it + col];  // W[i=b][k=a] = cpt[a][b]
#pragma unroll
        for (int32_t jt = 0; jt < 2; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag[jt], acc[it][jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int32_t it = 0; it < S / 16; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r)
          out_own[(int64_t)(16 * it + 4 * r + kq) * kTile64 + 16 * jt + col] = acc[it][jt][r];
  }
  // ---- to parent: out_par^T[a][n] = sum_b cpt[a][b] * in_own^T[b][n]
  {
    f64x4 acc[D0 / 16][2];
#pragma unroll
    for (int32_t it = 0; it < D0 / 16; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) acc[it][jt][r] = 0.0;
#pragma unroll
    for (int32_t ks = 0; ks < S / 4; ++ks) {
      const int32_t b = 4 * ks + kq;
      double bfrag[2];
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt) bfrag[jt] = in_own[(int64_t)b * kTile64 + 16 * jt + col];
#pragma unroll
      for (int32_t it = 0; it < D0 / 16; ++it) {
        const double afrag = w_lds[(16 * it + col) * kCptPitch + b];  // W[i=a][k=b] = cpt[a][b]
#pragma unroll
        for (int32_t jt = 0; jt < 2; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag[jt], acc[it][jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int32_t it = 0; it < D0 / 16; ++it)
#pragma unroll
      for (int32_t jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r)
          out_par[(int64_t)(16 * it + 4 * r + kq) * kTile64 + 16 * jt + col] = acc[it][jt][r];
  }
}

// u[ks] = product of the (one or two) arrays for state 4 * ks + kq and observation column `col` of
// the 16-column block at `base`; inv = 1 / sum over the states of u (times the weight array's
// entry, when there is one).  float64 twin of tree_operand.
template <int KD>
__device__ __forceinline__ void tree_operand_f64(const double* __restrict__ arena, int64_t first, int64_t second,
                                                 int64_t weight, int64_t base, int32_t kq, int32_t col,
                                                 double (&u)[KD / 4], double& inv) {
  // 2^-exponent of the column's largest entry over all KD states (a lane holds every fourth state)
  auto column_scale = [&](double mx) {
    mx = fmax(mx, __shfl_xor(mx, 16, 64));
    mx = fmax(mx, __shfl_xor(mx, 32, 64));
    int e = 0;
    (void)frexp(mx, &e);
    return ldexp(1.0, mx > 0.0 ? -e : 0);
  };
  double w[KD / 4];
  if (weight != kNoArray) {
    const double* wgt = arena + weight + base + kq * kTile64 + col;
#pragma unroll
    for (int32_t ks = 0; ks < KD / 4; ++ks) w[ks] = wgt[(int64_t)(4 * ks) * kTile64];
  }
  if (first == kNoArray) {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 4; ++ks) u[ks] = 1.0;
  } else {
    const double* a0 = arena + first + base + kq * kTile64 + col;
#pragma unroll
    for (int32_t ks = 0; ks < KD / 4; ++ks) u[ks] = a0[(int64_t)(4 * ks) * kTile64];
    if (second != kNoArray) {
      const double* a1 = arena + second + base + kq * kTile64 + col;
      double v[KD / 4];
#pragma unroll
      for (int32_t ks = 0; ks < KD / 4; ++ks) v[ks] = a1[(int64_t)(4 * ks) * kTile64];
      double mu = 0.0, mv = 0.0;
#pragma unroll
      for (int32_t ks = 0; ks < KD / 4; ++ks) {
        mu = fmax(mu, u[ks]);
        mv = fmax(mv, v[ks]);
      }
      const double su = column_scale(mu), sv = column_scale(mv);
#pragma unroll
      for (int32_t ks = 0; ks < KD / 4; ++ks) u[ks] = (u[ks] * su) * (v[ks] * sv);
    }
  }
  double sum = 0.0;
  if (weight != kNoArray) {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 4; ++ks) sum += u[ks] * w[ks];
  } else {
#pragma unroll
    for (int32_t ks = 0; ks < KD / 4; ++ks) sum += u[ks];
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  inv = 1.0 / sum;
}

// factor_tree_mfma for float64 plans: same tables, same two contractions, on the float64 matrix
// instruction with the operand layout of factor_p1_mfma_f64; a wave's 32 observations as two
// blocks of 16 columns, one after the other.
template <int D0, int S, bool EXACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
factor_tree_mfma_f64(double* __restrict__ arena, const double* __restrict__ cpts, TreeFamilyTables tab, int32_t K,
                     int64_t ntiles) {
  __shared__ double w_lds[D0 * kCptPitch];
  const int64_t item = blockIdx.x;
  const int32_t k = (int32_t)(item / ntiles);
  const int64_t t = item % ntiles;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t kq = lane >> 4, col = lane & 15;

  const double* cpt = cpts + tab.cpt_off[k];
  for (int32_t e = threadIdx.x; e < D0 * S; e += 256) w_lds[(e / S) * kCptPitch + (e % S)] = cpt[e];
  __syncthreads();
  const int64_t in_par0 = tab.in_par[2 * k], in_par1 = tab.in_par[2 * k + 1];
  const int64_t in_own0 = tab.in_own[2 * k], in_own1 = tab.in_own[2 * k + 1];
  const int64_t own_prev = EXACT ? tab.own_prev[k] : kNoArray;
  const int64_t out_par_off = tab.out[2 * k], out_own_off = tab.out[2 * k + 1];

#pragma unroll 1
  for (int32_t jt = 0; jt < 2; ++jt) {
    const int32_t column = wave * 32 + 16 * jt;
    // ---- to own variable
    {
      double u[D0 / 4], inv;
      tree_operand_f64<D0>(arena, in_par0, in_par1, kNoArray, t * D0 * kTile64 + column, kq, col, u, inv);
      __builtin_amdgcn_sched_barrier(0);
      f64x4 acc[S / 16];
#pragma unroll
      for (int32_t it = 0; it < S / 16; ++it)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) acc[it][r] = 0.0;
#pragma unroll
      for (int32_t ks = 0; ks < D0 / 4; ++ks) {
        const int32_t a = 4 * ks + kq;
#pragma unroll
        for (int32_t it = 0; it < S / 16; ++it)  // W[i=b][k=a] = cpt[a][b]
          acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(w_lds[a * kCptPitch + 16 * it + col], u[ks], acc[it], 0, 0, 0);
      }
      double* out_own = arena + out_own_off + t * S * kTile64 + column + col;
#pragma unroll
      for (int32_t it = 0; it < S / 16; ++it)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) out_own[(int64_t)(16 * it + 4 * r + kq) * kTile64] = acc[it][r] * inv;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- to parent
    {
      double u[S / 4], inv;
      tree_operand_f64<S>(arena, in_own0, in_own1, own_prev, t * S * kTile64 + column, kq, col, u, inv);
      __builtin_amdgcn_sched_barrier(0);
      f64x4 acc[D0 / 16];
#pragma unroll
      for (int32_t it = 0; it < D0 / 16; ++it)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) acc[it][r] = 0.0;
#pragma unroll
      for (int32_t ks = 0; ks < S / 4; ++ks) {
        const int32_t b = 4 * ks + kq;
#pragma unroll
        for (int32_t it = 0; it < D0 / 16; ++it)  // W[i=a][k=b] = cpt[a][b]
          acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(w_lds[(16 * it + col) * kCptPitch + b], u[ks], acc[it], 0, 0, 0);
      }
      double* out_par = arena + out_par_off + t * D0 * kTile64 + column + col;
#pragma unroll
      for (int32_t it = 0; it < D0 / 16; ++it)
#pragma unroll
        for (int32_t r = 0; r < 4; ++r) out_par[(int64_t)(16 * it + 4 * r + kq) * kTile64] = acc[it][r] * inv;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// EM count matrix of the same families in float64: stat[a][b] = sum_n in_par[n][a] * in_own[n][b].
// One workgroup per (node, chunk of tiles); 32 observations of both messages are staged in LDS
// per step (pitch 33), each wave owns a quarter of the 16x16 output tiles.
template <int D0, int S>
__global__ void __launch_bounds__(256) em_p1_mfma_f64(const double* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                     const int64_t* __restrict__ cpt_off, int32_t K, int64_t N,
                                                     int64_t ntiles, int32_t nchunks, double* __restrict__ partial) {
  constexpr int32_t kPitch = 33;
  __shared__ double par_lds[D0 * kPitch];
  __shared__ double own_lds[S * kPitch];
  const int32_t k = blockIdx.x / nchunks;
  const int32_t chunk = blockIdx.x % nchunks;
  if (k >= K) return;
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t kq = lane >> 4, col = lane & 15;
  constexpr int32_t kTilesOut = (D0 / 16) * (S / 16);  // 16x16 output tiles
  constexpr int32_t kPerWave = kTilesOut / 4;
  static_assert(kTilesOut % 4 == 0, "output tiles split evenly over four waves");

  const double* in_par = arena + in_off[2 * k];
  const double* in_own = arena + in_off[2 * k + 1];
  const int64_t per = (ntiles + nchunks - 1) / nchunks;
  const int64_t t0 = chunk * per;
  const int64_t t1 = t0 + per < ntiles ? t0 + per : ntiles;

  f64x4 acc[kPerWave];
#pragma unroll
  for (int32_t q = 0; q < kPerWave; ++q)
#pragma unroll
    for (int32_t r = 0; r < 4; ++r) acc[q][r] = 0.0;

  for (int64_t t = t0; t < t1; ++t) {
    for (int32_t part = 0; part < kTile64 / 32; ++part) {
      const int64_t n0 = t * kTile64 + part * 32;
      if (n0 >= N) break;
      __syncthreads();
      // stage [rows][32 observations]; thread -> (row, 2 observations); padding observations as zero
      for (int32_t e = threadIdx.x; e < D0 * 16; e += 256) {
        const int32_t row = e >> 4, c2 = (e & 15) * 2;
        const double2 x = *reinterpret_cast<const double2*>(in_par + (t * D0 + row) * kTile64 + part * 32 + c2);
        par_lds[row * kPitch + c2] = n0 + c2 < N ? x.x : 0.0;
        par_lds[row * kPitch + c2 + 1] = n0 + c2 + 1 < N ? x.y : 0.0;
      }
      for (int32_t e = threadIdx.x; e < S * 16; e += 256) {
        const int32_t row = e >> 4, c2 = (e & 15) * 2;
        const double2 x = *reinterpret_cast<const double2*>(in_own + (t * S + row) * kTile64 + part * 32 + c2);
        own_lds[row * kPitch + c2] = x.x;
        own_lds[row * kPitch + c2 + 1] = x.y;
      }
      __syncthreads();
#pragma unroll
      for (int32_t q = 0; q < kPerWave; ++q) {
        const int32_t tile = wave * kPerWave + q;
        const int32_t it = tile / (S / 16), jt = tile % (S / 16);
#pragma unroll
        for (int32_t ks = 0; ks < 8; ++ks) {
          const int32_t n = 4 * ks + kq;
          const double afrag = par_lds[(16 * it + col) * kPitch + n];  // A[i=a][k=n]
          const double bfrag = own_lds[(16 * jt + col) * kPitch + n];  // B[k=n][j=b]
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag, acc[q], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int32_t q = 0; q < kPerWave; ++q) {
    const int32_t tile = wave * kPerWave + q;
    const int32_t it = tile / (S / 16), jt = tile % (S / 16);
#pragma unroll
    for (int32_t r = 0; r < 4; ++r) {
      const int32_t a = 16 * it + 4 * r + kq, b = 16 * jt + col;
      partial[(cpt_off[k] + (int64_t)a * S + b) * nchunks + chunk] = acc[q][r];
    }
  }
}

// Variable pass for nodes with 8*SQ states and a small fan-in I (<= 4) in float64: as
// variable_split_fast, a block of eight waves owns (node, tile of 128 observations) and wave w keeps
// states [w*SQ, (w+1)*SQ) of all inputs in registers.  Products of doubles can leave the double
// range (the reference works on logarithms here, factor_graph.py:114-138), so every product is a
// mantissa with its exponent apart, and every sum over the states is taken relative to the
// largest exponent among its terms: first the waves agree on those largest exponents, then on the
// sums.   grid.x = K * ntiles, block = 512
template <int I, int SQ>
__global__ void __launch_bounds__(512) variable_split_f64(double* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                         const int64_t* __restrict__ out_off,
                                                         const int64_t* __restrict__ c_off, int32_t K,
                                                         int64_t ntiles, int32_t fresh) {
  constexpr int32_t NW = 8, S = NW * SQ;
  __shared__ int32_t top_lds[NW][I + 1][2][64];
  __shared__ double sum_lds[NW][I + 1][2][64];
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t item = blockIdx.x;  // (node, tile)
  const int32_t k = (int32_t)(item / ntiles);
  const int64_t t = item % ntiles;
  const int32_t col = lane * 2;
  const double uniform = 1.0 / (double)S;

  double2 in[I][SQ];
  double* dst[I];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const int64_t o = out_off[(int64_t)k * I + i];
    dst[i] = o != kNoArray ? arena + o + (t * S + wave * SQ) * kTile64 + col : nullptr;
    const bool skip = fresh && o != kNoArray;  // a message slot that still holds its initial uniform value
    const double* src = arena + in_off[(int64_t)k * I + i] + (t * S + wave * SQ) * kTile64 + col;
#pragma unroll
    for (int32_t q = 0; q < SQ; ++q)
      in[i][q] = skip ? make_double2(uniform, uniform) : *reinterpret_cast<const double2*>(src + (int64_t)q * kTile64);
  }
  // Term i < I of state q: the leave-one-out product prod_{j != i} in_j; term I: the full product.
  // Both as mantissa * 2^exponent (exponent kDeadExponent for an exact zero).
  auto terms = [&](int32_t q, int32_t v, double (&m)[I + 1], int32_t (&e)[I + 1]) {
    double fm[I];
    int32_t fe[I];
#pragma unroll
    for (int32_t i = 0; i < I; ++i) {
      int de = 0;
      fm[i] = frexp(v ? in[i][q].y : in[i][q].x, &de);
      fe[i] = de;
    }
    double pre = 1.0, suf = 1.0;
    int32_t pre_e = 0, suf_e = 0;
#pragma unroll
    for (int32_t i = 0; i < I; ++i) {
      m[i] = pre;
      e[i] = pre_e;
      pre *= fm[i];
      pre_e += fe[i];
    }
    m[I] = pre;
    e[I] = pre_e;
#pragma unroll
    for (int32_t i = I - 1; i >= 0; --i) {
      m[i] *= suf;
      e[i] += suf_e;
      suf *= fm[i];
      suf_e += fe[i];
    }
#pragma unroll
    for (int32_t i = 0; i <= I; ++i) {
      int de = 0;
      m[i] = frexp(m[i], &de);  // at most four mantissas multiplied: far from the range's end
      e[i] = m[i] == 0.0 ? kDeadExponent : e[i] + de;
    }
  };
  // 1. largest exponent of every term kind over all states
  int32_t top[I + 1][2];
#pragma unroll
  for (int32_t i = 0; i <= I; ++i) top[i][0] = top[i][1] = kDeadExponent;
#pragma unroll
  for (int32_t q = 0; q < SQ; ++q)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double m[I + 1];
      int32_t e[I + 1];
      terms(q, v, m, e);
#pragma unroll
      for (int32_t i = 0; i <= I; ++i) top[i][v] = e[i] > top[i][v] ? e[i] : top[i][v];
    }
#pragma unroll
  for (int32_t i = 0; i <= I; ++i)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) top_lds[wave][i][v][lane] = top[i][v];
  __syncthreads();
#pragma unroll
  for (int32_t i = 0; i <= I; ++i)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      int32_t x = kDeadExponent;
#pragma unroll
      for (int32_t w = 0; w < NW; ++w) x = top_lds[w][i][v][lane] > x ? top_lds[w][i][v][lane] : x;
      top[i][v] = x;
    }
  // 2. sums over the states relative to those exponents
  double sums[I + 1][2];
#pragma unroll
  for (int32_t i = 0; i <= I; ++i) sums[i][0] = sums[i][1] = 0.0;
#pragma unroll
  for (int32_t q = 0; q < SQ; ++q)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double m[I + 1];
      int32_t e[I + 1];
      terms(q, v, m, e);
#pragma unroll
      for (int32_t i = 0; i <= I; ++i) sums[i][v] += ldexp(m[i], clamp_shift(e[i] - top[i][v]));
    }
#pragma unroll
  for (int32_t i = 0; i <= I; ++i)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) sum_lds[wave][i][v][lane] = sums[i][v];
  __syncthreads();
#pragma unroll
  for (int32_t i = 0; i <= I; ++i)
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double x = 0.0;
#pragma unroll
      for (int32_t w = 0; w < NW; ++w) x += sum_lds[w][i][v][lane];
      sums[i][v] = x;
    }
  if (wave == 0) {
    double2 c;
    c.x = (double)top[I][0] * 0.6931471805599453 + log(sums[I][0]);
    c.y = (double)top[I][1] * 0.6931471805599453 + log(sums[I][1]);
    *reinterpret_cast<double2*>(arena + c_off[k] + t * kTile64 + col) = c;
  }
  // 3. outgoing messages: leave-one-out product over its sum; the message to the local factor
  //    (slot I - 1) is normalised by the full product's sum instead
#pragma unroll
  for (int32_t v = 0; v < 2; ++v) {
#pragma unroll
    for (int32_t i = 0; i < I - 1; ++i) sums[i][v] = 1.0 / sums[i][v];
    sums[I - 1][v] = 1.0 / sums[I][v];
    top[I - 1][v] = top[I][v];
  }
#pragma unroll
  for (int32_t q = 0; q < SQ; ++q) {
    double2 out[I];
#pragma unroll
    for (int32_t v = 0; v < 2; ++v) {
      double m[I + 1];
      int32_t e[I + 1];
      terms(q, v, m, e);
#pragma unroll
      for (int32_t i = 0; i < I; ++i) {
        const double o = ldexp(m[i], clamp_shift(e[i] - top[i][v])) * sums[i][v];
        if (v) out[i].y = o; else out[i].x = o;
      }
    }
#pragma unroll
    for (int32_t i = 0; i < I; ++i)
      if (dst[i]) *reinterpret_cast<double2*>(dst[i] + (int64_t)q * kTile64) = out[i];
  }
}

// ---------------------------------------------------------------------------------
// Variable pass for nodes with many states and a small fan-in I (<= 4): one wave per
// (node, tile), four observations per lane, states streamed one at a time (products are per
// state; only the normalisers couple states).  Pass 1 accumulates the normalisers, pass 2
// re-reads the inputs (mostly from L2 / Infinity Cache) and writes the normalised messages.
// Same quantities as variable_pass_generic.
template <int I>
__global__ void __launch_bounds__(256) variable_big_fast(float* __restrict__ arena, const int64_t* __restrict__ in_off,
                                                        const int64_t* __restrict__ out_off,
                                                        const int64_t* __restrict__ c_off, int32_t K, int32_t S,
                                                        int64_t ntiles, int32_t fresh) {
  int32_t k, lane;
  int64_t t;
  if (!wave_item((int64_t)K * ntiles, ntiles, k, t, lane)) return;
  const float uniform = 1.0f / (float)S;
  const float* src[I];
  float* dst[I];
  bool skip[I];
#pragma unroll
  for (int32_t i = 0; i < I; ++i) {
    const int64_t o = out_off[(int64_t)k * I + i];
    src[i] = arena + in_off[(int64_t)k * I + i];
    dst[i] = o != kNoArray ? arena + o : nullptr;
    skip[i] = fresh && o != kNoArray;
  }

  double sums[I][4];
  double z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int32_t i = 0; i < I; ++i)
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) sums[i][v] = 0.0;

  for (int32_t s = 0; s < S; ++s) {
    f4 in[I];
#pragma unroll
    for (int32_t i = 0; i < I; ++i) in[i] = skip[i] ? f4{{uniform, uniform, uniform, uniform}} : load4(src[i], t * S + s, lane);
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      double loo[I];
      double pre = 1.0;
#pragma unroll
      for (int32_t i = 0; i < I; ++i) {
        loo[i] = pre;
        pre *= (double)in[i].v[v];
      }
      z[v] += pre;
      double suf = 1.0;
#pragma unroll
      for (int32_t i = I - 1; i >= 0; --i) {
        sums[i][v] += loo[i] * suf;
        suf *= (double)in[i].v[v];
      }
    }
  }
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
  for (int32_t v = 0; v < 4; ++v) {
#pragma unroll
    for (int32_t i = 0; i < I - 1; ++i) sums[i][v] = 1.0 / sums[i][v];
    sums[I - 1][v] = 1.0 / z[v];  // the message to the local factor is normalised by the full product
  }
  for (int32_t s = 0; s < S; ++s) {
    f4 in[I];
#pragma unroll
    for (int32_t i = 0; i < I; ++i) in[i] = skip[i] ? f4{{uniform, uniform, uniform, uniform}} : load4(src[i], t * S + s, lane);
    f4 out[I];
#pragma unroll
    for (int32_t v = 0; v < 4; ++v) {
      double loo[I];
      double pre = 1.0;
#pragma unroll
      for (int32_t i = 0; i < I; ++i) {
        loo[i] = pre;
        pre *= (double)in[i].v[v];
      }
      double suf = 1.0;
#pragma unroll
      for (int32_t i = I - 1; i >= 0; --i) {
        out[i].v[v] = (float)(loo[i] * suf * sums[i][v]);
        suf *= (double)in[i].v[v];
      }
    }
#pragma unroll
    for (int32_t i = 0; i < I; ++i)
      if (dst[i]) store4(dst[i], t * S + s, lane, out[i]);
  }
}

}  // namespace bn
