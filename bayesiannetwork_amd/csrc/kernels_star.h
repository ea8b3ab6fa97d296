// A whole mini-batch EM step of a "star" network in one cooperative launch, for small batches.
//
// Star: one wide root Q (D0 states) whose children are all two-state leaf families of one shape,
// i.e. the latent-class model of the reference's MNIST experiment (dev_experiments/mnist/mnist.py).
// At the reference's own batch sizes (100, 1 000) a step made of one launch per pass is bound by
// its chain of sixteen dependent dispatches; here the chain is T grid-wide barriers inside one
// kernel.  The arithmetic is that of the per-pass kernels (wide_combine_states, leaf_loop_p1,
// leaf_family_p1 in reduce mode, em_finalize*, m_step_rows): every one of the T flooding iterations
// is run; what the persistent kernel saves is memory round trips, not passes:
//   * a wave keeps the evidence of its few families in registers for all T iterations;
//   * q = cpt . ev of a family is the same in every iteration, so its product over the block's
//     families is formed once and only the product of the leaf normalisers is renewed per iteration.
//
//   block (c, sl): chunk c of the families x slice sl of 128 observations, 8 waves that split the
//   chunk's families.  Per iteration: the block's partial product goes to global memory, a
//   grid-wide barrier, then every block multiplies the partial products of its slice (redundantly
//   per chunk) to obtain the root's scaled product P' for the next iteration.
//
// The grid must be co-resident: it is kept to half the chip's CUs (kStarMaxBlocks; launched through
// hipLaunchCooperativeKernel, which checks residency, if BN_STAR_COOPERATIVE is set: that launch
// measured 21 us slower per step).  The barrier's spin is bounded, so a launch that cannot make
// progress ends instead of hanging: all blocks then agree that the step is void (star_give_up), the
// parameters stay as they were, the log-likelihood is NaN and a host-visible flag is set; the next
// bn_em_step call on the plan counts the failure and goes back to one launch per pass.
#pragma once

#include "kernels_leaf.h"

namespace bn {

constexpr int32_t kStarWaves = 8;        // waves per block
constexpr int32_t kStarFamilies = 8;     // families a wave can keep in registers
// chunks of families (= blocks per slice): every block reads all of its slice's partial products
// in every iteration, so the traffic of an iteration grows with the square of this
constexpr int32_t kStarChunks = 16;
constexpr int32_t kStarSlice = 128;      // observations per block (two per lane)
// Every block must be resident for the barriers to work.  A block fills a CU (LDS), the chip has
// 256 of them and nothing else runs on the stream while the kernel does: half the chip at most.
constexpr int32_t kStarMaxBlocks = 128;
constexpr int32_t kStarCounterStride = 64;  // uint32 elements between two barrier counters (256 bytes)
constexpr int32_t kStarMaxSlices = 16;

template <typename T>
struct StarArgs {
  T* arena;
  T* cpts;       // the plan's CPT snapshot: read during the step, overwritten by the M-step
  T* new_cpts;   // the caller's copy of the new CPTs
  const int64_t* obs_off;  // [K] observation array (or the all-ones array) of each leaf
  // non-null: the evidence is read from the caller's [N][2] arrays instead (one pointer per leaf;
  // for a leaf that is not observed, its all-ones array in the arena, which has the same shape up
  // to N).  The plan's own copy is not needed: the step ends with the plan reset.
  const void* const* src;
  // star_em_stream only: the leaves' normalisers between two iterations ([K][npad]; every lane reads
  // back what it wrote itself)
  T* z_scratch;
  // ... and, when the caller's arrays lie at equal distances, leaf k's is src_base + k * src_stride
  // bytes and `src` is not read (src_stride = 0: `src` is)
  const unsigned char* src_base;
  int64_t src_stride;
  int32_t cpt_in_lds;  // star_em_stream: the launch has room for the block's CPT rows in LDS
  const int64_t* cpt_off;  // [K] element offset of each leaf's CPT
  int64_t root_cpt_off;
  int32_t K, C, chunk_len, nslices, iterations;
  int64_t N, npad;
  double* part_mant;   // [2][C][D0][npad] block partial products, alternating between iterations
  int32_t* part_exp;
  double* count_part;  // [K][D0 * 2][nslices] count sums per slice
  double* root_part;   // [nslices][D0] sums of the root's posterior
  double* ll_part;     // [C * nslices + nslices] log-likelihood sums per block (+ the root's normaliser per slice)
  // Arrival counters, never reset (epochs continue from launch to launch), 256 bytes apart:
  // [0] all blocks (the barrier before the M-step), [1 + sl] the blocks of slice sl (the
  // iterations only couple the blocks of one slice: fewer arrivals per counter, no shared line)
  uint32_t* barrier;
  uint32_t epoch0;        // of the grid-wide counter
  uint32_t slice_epoch0;  // of every slice counter
  int32_t* error;      // host-visible; set to 1 by a block whose barrier wait ran out
  double learning_rate, regularization, ll_scale;
  double* ll_out;
};

// What the blocks hand each other inside the launch goes through device-coherent accesses
// (relaxed atomics at agent scope: written through, never served from a stale cache), so that a
// barrier needs no release / acquire fence: on this GPU such a fence writes back and invalidates the
// whole L2 of the XCD, once per block and barrier.
template <typename U>
__device__ __forceinline__ void shared_store(U* p, U v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename U>
__device__ __forceinline__ U shared_load(const U* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A counter that has been given up on carries this on top of its arrivals.  Epochs advance by at most
// a few thousand per launch, so a distance of half of it from the target cannot be mistaken.
constexpr uint32_t kStarPoison = 0x40000000u;
constexpr int32_t kStarWait = 0, kStarPass = 1, kStarAbort = 2;

// What a block that waits for `target` arrivals makes of the counter's value.
__host__ __device__ constexpr int32_t star_verdict(uint32_t seen, uint32_t target) {
  const int32_t ahead = (int32_t)(seen - target);
  return ahead >= (int32_t)(kStarPoison / 2) ? kStarAbort : (ahead >= 0 ? kStarPass : kStarWait);
}

// Gives the counter up, unless all blocks have arrived meanwhile: a compare-and-swap, so that in the
// one order all blocks see either the target was reached first (everybody passes, this block too) or
// the poison came first (everybody who looks afterwards aborts, nobody has passed before).
__device__ __forceinline__ int32_t star_give_up(uint32_t* counter, uint32_t target) {
  uint32_t seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    const int32_t verdict = star_verdict(seen, target);
    if (verdict != kStarWait) return verdict;
    if (__hip_atomic_compare_exchange_strong(counter, &seen, seen + kStarPoison, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT))
      return kStarAbort;
  }
}

// All blocks of a counter arrive, then all leave - or all learn that the launch is given up.  Every
// wave first waits for its own coherent stores to have been performed; thread 0 then counts the
// block in and waits for the epoch's total.  The bounded wait makes every wave reach the end of the
// kernel whatever happens; the verdict is the same for every block of the counter.
__device__ __forceinline__ bool star_barrier(uint32_t* counter, uint32_t target, int32_t* error) {
  __shared__ int32_t ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int32_t verdict = kStarWait;
    for (int32_t spin = 0; spin < (1 << 22) && verdict == kStarWait; ++spin) {
      verdict = star_verdict(shared_load(counter), target);
      if (verdict == kStarWait) __builtin_amdgcn_s_sleep(2);
    }
    if (verdict == kStarWait) verdict = star_give_up(counter, target);
    if (verdict == kStarAbort) shared_store(error, 1);
    ok = verdict == kStarPass;
  }
  __syncthreads();
  return ok != 0;
}

template <typename T, int D0>
__global__ void __launch_bounds__(64 * kStarWaves) star_em_step(StarArgs<T> a) {
  constexpr int32_t S = 2, V = 2, E = D0 * S, NW = kStarWaves, F = kStarFamilies;
  constexpr int32_t kTileT = tile_of<T>::value;
  extern __shared__ __align__(16) unsigned char star_lds[];
  // LDS: [0, merge)   tree merge of the waves' prod q: up to NW/2 writers x (D0*V doubles + D0*V ints) x 64 lanes
  //      then         per-wave prod z (V doubles + V ints per lane), all NW waves
  //      then         the combine's per-state products [D0][V] doubles + ints per lane
  double* merge_m = reinterpret_cast<double*>(star_lds);
  int32_t* merge_e = reinterpret_cast<int32_t*>(merge_m + (size_t)(NW / 2) * D0 * V * 64);
  double* z_m = reinterpret_cast<double*>(merge_e + (size_t)(NW / 2) * D0 * V * 64);
  int32_t* z_e = reinterpret_cast<int32_t*>(z_m + (size_t)NW * V * 64);
  double* st_m = reinterpret_cast<double*>(z_e + (size_t)NW * V * 64);
  int32_t* st_e = reinterpret_cast<int32_t*>(st_m + (size_t)D0 * V * 64);
  double* red = reinterpret_cast<double*>(st_e + (size_t)D0 * V * 64);  // [NW * 64] block reductions

  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t c = (int32_t)blockIdx.x / a.nslices, sl = (int32_t)blockIdx.x % a.nslices;
  const int32_t nblocks = a.C * a.nslices;
  const int64_t obs0 = (int64_t)sl * kStarSlice + lane * V;  // first of this lane's two observations
  const int64_t t = obs0 / kTileT;
  const int32_t col = (int32_t)(obs0 % kTileT);
  uint32_t slice_epoch = a.slice_epoch0;
  uint32_t* slice_counter = a.barrier + (size_t)(1 + sl) * kStarCounterStride;

  // the block's families, split evenly over its waves
  const int32_t fb0 = c * a.chunk_len;
  const int32_t fb1 = fb0 + a.chunk_len < a.K ? fb0 + a.chunk_len : a.K;
  const int32_t per_wave = (fb1 - fb0 + NW - 1) / NW;
  const int32_t lo = fb0 + wave * per_wave < fb1 ? fb0 + wave * per_wave : fb1;
  const int32_t hi = lo + per_wave < fb1 ? lo + per_wave : fb1;
  const int32_t nf = hi - lo;  // <= F (checked by the host)

  fv<T, V> ev[F][S];
#pragma unroll
  for (int32_t f = 0; f < F; ++f)
#pragma unroll
    for (int32_t j = 0; j < S; ++j)
      ev[f][j] = f < nf && !a.src ? loadv<T, V>(a.arena + a.obs_off[lo + f], t * S + j, col) : splat<T, V>((T)1);
  if (a.src) {
    // [observation][state]: the lane's two observations x two states are 4 consecutive elements
    // (padding observations, and the second one of an odd batch's last lane, count as all ones)
#pragma unroll
    for (int32_t f = 0; f < F; ++f) {
      if (f < nf && obs0 < a.N) {
        const T* rows = static_cast<const T*>(a.src[lo + f]) + obs0 * S;
        if (obs0 + 1 < a.N) {
          const fv<T, 4> raw = *reinterpret_cast<const fv<T, 4>*>(rows);
          ev[f][0][0] = raw[0];
          ev[f][1][0] = raw[1];
          ev[f][0][1] = raw[2];
          ev[f][1][1] = raw[3];
        } else {
          ev[f][0][0] = rows[0];
          ev[f][1][0] = rows[1];
        }
      }
    }
  }
  const T* prior = a.cpts + a.root_cpt_off;
  // the CPT rows of the block's families, fetched by all threads at once (two memory round trips
  // for the lot instead of two per family) and read from LDS from then on
  T* cpt_lds = reinterpret_cast<T*>(red + NW * 64 + (size_t)D0 * V * 64 * 3 / 2);  // after the block's prod q
  for (int32_t i = threadIdx.x; i < (fb1 - fb0) * E; i += 64 * NW)
    cpt_lds[i] = a.cpts[a.cpt_off[fb0 + i / E] + i % E];
  __syncthreads();

  // q = cpt . ev of family f
  auto family_q = [&](int32_t f, fv<T, V>* q) {
    const T* cpt = cpt_lds + (size_t)(lo - fb0 + f) * E;
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      q[d] = cpt[d * S] * ev[f][0];
#pragma unroll
      for (int32_t j = 1; j < S; ++j) q[d] = __builtin_elementwise_fma(splat<T, V>(cpt[d * S + j]), ev[f][j], q[d]);
    }
  };

  // ---- prod_k q_k[d] over the wave's families, then over the block (kept by wave 0)
  double qm[D0][V];
  int32_t qe[D0][V];
#pragma unroll
  for (int32_t d = 0; d < D0; ++d)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      qm[d][v] = 1.0;
      qe[d][v] = 0;
    }
#pragma unroll
  for (int32_t f = 0; f < F; ++f) {
    if (f < nf) {
      fv<T, V> q[D0];
      family_q(f, q);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          qm[d][v] = frexp(qm[d][v] * (double)q[d][v], &de);
          qe[d][v] += de;
        }
    }
  }
  for (int32_t step = NW / 2; step >= 1; step >>= 1) {
    // waves [step, 2 step) hand their products to waves [0, step)
    __syncthreads();
    if (wave >= step && wave < 2 * step) {
      double* dm = merge_m + (size_t)(wave - step) * D0 * V * 64;
      int32_t* de = merge_e + (size_t)(wave - step) * D0 * V * 64;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          dm[(d * V + v) * 64 + lane] = qm[d][v];
          de[(d * V + v) * 64 + lane] = qe[d][v];
        }
    }
    __syncthreads();
    if (wave < step) {
      const double* dm = merge_m + (size_t)wave * D0 * V * 64;
      const int32_t* de = merge_e + (size_t)wave * D0 * V * 64;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int x = 0;
          qm[d][v] = frexp(qm[d][v] * dm[(d * V + v) * 64 + lane], &x);
          qe[d][v] += de[(d * V + v) * 64 + lane] + x;
        }
    }
  }

  // the block's prod q stays in LDS (wave 0 reads it once per iteration): no registers across the loop
  double* bq_m = red + NW * 64;                                    // [D0][V][64]
  int32_t* bq_e = reinterpret_cast<int32_t*>(bq_m + (size_t)D0 * V * 64);
  if (wave == 0) {
#pragma unroll
    for (int32_t d = 0; d < D0; ++d)
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        bq_m[(d * V + v) * 64 + lane] = qm[d][v];
        bq_e[(d * V + v) * 64 + lane] = qe[d][v];
      }
  }

  // The root's scaled product P' (largest state 1) and its normaliser c_Q for this lane's
  // observations, from the blocks' partial products of the iteration before.
  fv<T, V> pf[D0];
  fv<T, V> pf_sum;
  double cq[V] = {0.0, 0.0};
  auto combine = [&](int32_t buffer) {
    const double* pm = a.part_mant + (size_t)buffer * a.C * D0 * a.npad;
    const int32_t* pe = a.part_exp + (size_t)buffer * a.C * D0 * a.npad;
    for (int32_t d = wave; d < D0; d += NW) {
      double m[V] = {1.0, 1.0};
      int32_t e[V] = {0, 0};
      for (int32_t ch = 0; ch < a.C; ch += 16) {
        double2 xm[16];
        int2 xe[16];
#pragma unroll
        for (int32_t u = 0; u < 16; ++u) {
          const bool live = ch + u < a.C;
          const int64_t row = ((int64_t)(live ? ch + u : ch) * D0 + d) * a.npad + obs0;
          xm[u] = make_double2(shared_load(pm + row), shared_load(pm + row + 1));
          xe[u] = make_int2(shared_load(pe + row), shared_load(pe + row + 1));
          if (!live) {
            xm[u] = make_double2(1.0, 1.0);
            xe[u] = make_int2(0, 0);
          }
        }
#pragma unroll
        for (int32_t u = 0; u < 16; ++u) {
          m[0] *= xm[u].x;
          m[1] *= xm[u].y;
          e[0] += xe[u].x;
          e[1] += xe[u].y;
        }
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          m[v] = frexp(m[v], &de);
          e[v] += de;
        }
      }
      const double x = (double)prior[d];  // the local factor's message: the prior
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        m[v] = frexp(m[v] * x, &de);
        e[v] = m[v] == 0.0 ? kDeadExponent : e[v] + de;
        st_m[(d * V + v) * 64 + lane] = m[v];
        st_e[(d * V + v) * 64 + lane] = e[v];
      }
    }
    __syncthreads();
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int32_t emax = kDeadExponent;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const int32_t x = st_e[(d * V + v) * 64 + lane];
        emax = x > emax ? x : emax;
      }
      double z = 0.0;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const double scaled = ldexp(st_m[(d * V + v) * 64 + lane], clamp_shift(st_e[(d * V + v) * 64 + lane] - emax));
        z += scaled;
        pf[d][v] = (T)scaled;
      }
      cq[v] = (double)(T)((double)emax * 0.6931471805599453 + log(z));
    }
    pf_sum = pf[0];
#pragma unroll
    for (int32_t d = 1; d < D0; ++d) pf_sum += pf[d];
    __syncthreads();  // st_* may be rewritten by the next combine
  };

  // ---- iterations 1 .. T-1: new leaf normalisers, the block's partial product for the root
  fv<T, V> zk[F];  // the leaf normaliser the coming iteration uses
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) pf[d] = splat<T, V>((T)1);
  pf_sum = splat<T, V>((T)D0);
  bool alive = true;
  for (int32_t it = 1; it < a.iterations && alive; ++it) {
    if (it > 1) combine((it - 1) & 1);
    double zm[V] = {1.0, 1.0};  // prod of this iteration's leaf normalisers over the wave's families
    int32_t ze[V] = {0, 0};
#pragma unroll
    for (int32_t f = 0; f < F; ++f) {
      if (f < nf) {
        fv<T, V> q[D0];
        family_q(f, q);
        fv<T, V> zc, zn;
        if (it == 1) {
          // nothing has been sent yet: every message is uniform
          fv<T, V> own[S];
#pragma unroll
          for (int32_t j = 0; j < S; ++j) own[j] = splat<T, V>((T)1 / (T)S);
          zc = leaf_normaliser<T, S, V>(ev[f], own);
          fv<T, V> sum_r = splat<T, V>(0.f), acc = splat<T, V>(0.f);
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) {
            const fv<T, V> r = pf[d] * rcpv<T, V>(splat<T, V>((T)1 / (T)D0));
            sum_r += r;
            acc = __builtin_elementwise_fma(r, q[d], acc);
          }
          zn = acc * rcpv<T, V>(sum_r);
        } else {
          zc = zk[f];
          fv<T, V> sum_r = splat<T, V>(0.f);
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) sum_r = __builtin_elementwise_fma(pf[d], rcpv<T, V>(q[d]), sum_r);
          zn = pf_sum * rcpv<T, V>(sum_r);
        }
        zk[f] = zn;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          zm[v] = frexp(zm[v] * (double)zc[v], &de);
          ze[v] += de;
        }
      }
    }
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      z_m[(wave * V + v) * 64 + lane] = zm[v];
      z_e[(wave * V + v) * 64 + lane] = ze[v];
    }
    __syncthreads();
    if (wave == 0) {
      // the block's partial product of the messages to the root: prod q / prod z
      double inv_z[V];
      int32_t ez[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        double m = 1.0;
        int32_t e = 0;
#pragma unroll
        for (int32_t w = 0; w < NW; ++w) {
          int de = 0;
          m = frexp(m * z_m[(w * V + v) * 64 + lane], &de);
          e += z_e[(w * V + v) * 64 + lane] + de;
        }
        inv_z[v] = 1.0 / m;
        ez[v] = e;
      }
      double* pm = a.part_mant + (size_t)(it & 1) * a.C * D0 * a.npad;
      int32_t* pe = a.part_exp + (size_t)(it & 1) * a.C * D0 * a.npad;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        double m[V];
        int32_t e[V];
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          m[v] = frexp(bq_m[(d * V + v) * 64 + lane] * inv_z[v], &de);
          e[v] = bq_e[(d * V + v) * 64 + lane] + de - ez[v];
        }
        const int64_t row = ((int64_t)c * D0 + d) * a.npad + obs0;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          shared_store(pm + row + v, m[v]);
          shared_store(pe + row + v, e[v]);
        }
      }
    }
    slice_epoch += (uint32_t)a.C;
    alive = star_barrier(slice_counter, slice_epoch, a.error);
    // a block that leaves the iterations will never arrive at the grid-wide barrier: it gives that
    // one up as well, so that the other slices' blocks do not wait for it
    if (!alive && threadIdx.x == 0) (void)star_give_up(a.barrier, a.epoch0 + (uint32_t)nblocks);
  }

  // ---- iteration T on the fly: log-likelihood and count sums
  if (alive) {
    combine((a.iterations - 1) & 1);
    double zm[V] = {1.0, 1.0};
    int32_t ze[V] = {0, 0};
#pragma unroll
    for (int32_t f = 0; f < F; ++f) {
      if (f < nf) {
        const int32_t k = lo + f;
        fv<T, V> q[D0], r[D0];
        family_q(f, q);
        fv<T, V> inv_sum = splat<T, V>(0.f);
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) {
          r[d] = pf[d] * rcpv<T, V>(q[d]);
          inv_sum += r[d];
        }
        inv_sum = rcpv<T, V>(inv_sum);
        const fv<T, V> z = zk[f];
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          zm[v] = frexp(zm[v] * (double)z[v], &de);
          ze[v] += de;
        }
        // counts: sum_n t[d] * o[j], t = r / sum r, o = ev / z (padding observations masked)
        const fv<T, V> scale = recipv<T, V>(z) * inv_sum;
        fv<T, V> o[S];
#pragma unroll
        for (int32_t j = 0; j < S; ++j) {
          o[j] = ev[f][j] * scale;
#pragma unroll
          for (int32_t v = 0; v < V; ++v)
            if (obs0 + v >= a.N) o[j][v] = 0.f;
        }
        T per_lane[E];
#pragma unroll
        for (int32_t d = 0; d < D0; ++d)
#pragma unroll
          for (int32_t j = 0; j < S; ++j) {
            T x = r[d][0] * o[j][0];
#pragma unroll
            for (int32_t v = 1; v < V; ++v) x = fma_t(r[d][v], o[j][v], x);
            per_lane[d * S + j] = x;
          }
        WaveSums<T, E> sums;
        sums.reduce(per_lane, lane);
        if (lane < 4) {
#pragma unroll
          for (int32_t g = 0; g < WaveSums<T, E>::H2; ++g) {
            const int32_t e = WaveSums<T, E>::entry(lane, g);
            if (e < E) shared_store(a.count_part + ((int64_t)k * E + e) * a.nslices + sl, (double)sums.out[g]);
          }
        }
      }
    }
    // sum over the wave's families of log z, per observation, then over the block's observations
    double mine = 0.0;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      const double m = frexp(zm[v], &de);
      if (obs0 + v < a.N) mine += (double)(ze[v] + de) * 0.6931471805599453 + log(m);
    }
    // root: its normaliser and its posterior, by the blocks of chunk 0 (one per slice), wave 0
    double root_mine[D0];
    if (c == 0 && wave == 0) {
#pragma unroll
      for (int32_t v = 0; v < V; ++v)
        if (obs0 + v < a.N) mine += cq[v];
      double inv[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        double z = 0.0;
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) z += (double)pf[d][v];
        inv[v] = obs0 + v < a.N ? 1.0 / z : 0.0;
      }
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) root_mine[d] = (double)pf[d][0] * inv[0] + (double)pf[d][1] * inv[1];
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        double x = root_mine[d];
#pragma unroll
        for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) shared_store(a.root_part + (int64_t)sl * D0 + d, x);
      }
    }
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) red[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int32_t w = 0; w < NW; ++w) total += red[w];
      shared_store(a.ll_part + blockIdx.x, total);
    }
    alive = star_barrier(a.barrier, a.epoch0 + (uint32_t)nblocks, a.error);
  }
  if (!alive) {
    // A barrier was given up (the same verdict in every block): no result.  The step leaves the
    // parameters as they were - the plan's snapshot untouched, the caller's buffer filled with them -
    // and the caller sees a NaN log-likelihood and the error flag.
    if (sl != 0) return;
    if (a.new_cpts) {
      for (int32_t i = threadIdx.x; i < (fb1 - fb0) * E; i += 64 * NW) {
        const int64_t off = a.cpt_off[fb0 + i / E] + i % E;
        a.new_cpts[off] = a.cpts[off];
      }
      if (blockIdx.x == 0 && threadIdx.x < D0) a.new_cpts[a.root_cpt_off + threadIdx.x] = a.cpts[a.root_cpt_off + threadIdx.x];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.ll_out = __builtin_nan("");
    return;
  }

  // ---- M-step (EmBatchOptimizer._m_step, optimizers/em_batch_optimizer.py:70-84): the blocks of
  // slice 0 finalise their chunk's families, block 0 also the root and the log-likelihood
  if (sl != 0) return;
  {
#pragma clang fp contract(off)
  const double gamma = a.regularization, rate = a.learning_rate;
  auto new_row = [&](int64_t off, int32_t len, const double* counts) {
    double sum = 0.0;
    for (int32_t j = 0; j < len; ++j) sum += counts[j];
    const double uniform = 1.0 / (double)len;
    for (int32_t j = 0; j < len; ++j) {
      double x = counts[j] / sum;
      if (gamma != 0.0) {
        const double p = (1.0 - gamma) * x;
        const double u = gamma * uniform;
        x = p + u;
      }
      if (rate >= 0.0) {
        const double old = (1.0 - rate) * (double)a.cpts[off + j];
        const double fresh = rate * x;
        x = old + fresh;
      }
      a.cpts[off + j] = (T)x;
      if (a.new_cpts) a.new_cpts[off + j] = (T)x;
    }
  };
  for (int32_t row = threadIdx.x; row < (fb1 - fb0) * D0; row += 64 * NW) {
    const int32_t k = fb0 + row / D0, d = row % D0;
    const int64_t off = a.cpt_off[k] + (int64_t)d * S;
    double counts[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      double x = 0.0;
      const double* src = a.count_part + ((int64_t)k * E + d * S + j) * a.nslices;
      for (int32_t s2 = 0; s2 < a.nslices; ++s2) x += shared_load(src + s2);
      counts[j] = x * (double)a.cpts[off + j];
    }
    new_row(off, S, counts);
  }
  if (blockIdx.x == 0) {
    // every thread fetches its share of the blocks' sums at once (one memory round trip), then the
    // block adds them up in a fixed order
    double mine = 0.0;
    for (int32_t b = threadIdx.x; b < nblocks; b += 64 * NW) mine += shared_load(a.ll_part + b);
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    double root_count = 0.0;  // thread d < D0: the root's count of state d
    if (threadIdx.x < D0)
      for (int32_t s2 = 0; s2 < a.nslices; ++s2) root_count += shared_load(a.root_part + (int64_t)s2 * D0 + threadIdx.x);
    __syncthreads();  // red[] was read by thread 0 before the barrier
    if (lane == 0) red[wave] = mine;
    if (threadIdx.x < D0) red[NW + threadIdx.x] = root_count;
    __syncthreads();
    if (threadIdx.x == 0) {
      double counts[D0];
      for (int32_t d = 0; d < D0; ++d) counts[d] = red[NW + d];
      new_row(a.root_cpt_off, D0, counts);
      double ll = 0.0;
      for (int32_t w = 0; w < NW; ++w) ll += red[w];
      *a.ll_out = ll * a.ll_scale;
    }
  }
  }
}


// ---------------------------------------------------------------------------------
// The same one-launch EM step for batches whose evidence no longer fits the registers of the resident
// waves (about a thousand to a few tens of thousands of observations): star_em_stream.
//
//   block (c, sl) as in star_em_step, but a wave walks its families one after the other in every
//   iteration, reading their evidence again (the block reads nothing but its own chunk's evidence:
//   the re-reads come from the L2 of the XCD it sits on) and keeping a family's normaliser for the
//   next iteration in global scratch instead of a register.  The first pass forms prod q and the
//   first iteration's values together; the waves' products meet in LDS two at a time (30 KB instead
//   of a buffer per pair), so that two blocks fit on a CU: the launch may have twice as many blocks
//   as the chip has CUs, four waves per SIMD, each with half as many families to walk.  After the grid-wide barrier ALL blocks share the M-step (a wave
//   per CPT row, the lanes fetching the slices' count sums side by side).
constexpr int32_t kStreamMaxSlices = 256;

template <typename T, int D0>
__global__ void __launch_bounds__(64 * kStarWaves) star_em_stream(StarArgs<T> a) {
  constexpr int32_t S = 2, V = 2, E = D0 * S, NW = kStarWaves;
  constexpr int32_t kTileT = tile_of<T>::value;
  extern __shared__ __align__(16) unsigned char star_lds[];
  // LDS: two merge buffers (D0*V doubles + D0*V ints per lane each; the combine's per-state products
  // live in the first one, which is free once the first pass has ended), the waves' prod z, block
  // reductions, the block's prod q, the block's CPT rows
  double* merge_m = reinterpret_cast<double*>(star_lds);
  int32_t* merge_e = reinterpret_cast<int32_t*>(merge_m + (size_t)2 * D0 * V * 64);
  double* z_m = reinterpret_cast<double*>(merge_e + (size_t)2 * D0 * V * 64);
  int32_t* z_e = reinterpret_cast<int32_t*>(z_m + (size_t)NW * V * 64);
  double* st_m = merge_m;
  int32_t* st_e = merge_e;
  double* red = reinterpret_cast<double*>(z_e + (size_t)NW * V * 64);  // [NW * 64]
  double* bq_m = red + NW * 64;                                          // [D0][V][64]
  int32_t* bq_e = reinterpret_cast<int32_t*>(bq_m + (size_t)D0 * V * 64);
  // the CPT rows of the block's families, when the host found room for them (a.cpt_in_lds): fetched by
  // all threads at once instead of two scalar round trips per family and iteration
  T* cpt_lds = reinterpret_cast<T*>(bq_e + (size_t)D0 * V * 64);

  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int32_t c = (int32_t)blockIdx.x / a.nslices, sl = (int32_t)blockIdx.x % a.nslices;
  const int32_t nblocks = a.C * a.nslices;
  const int64_t obs0 = (int64_t)sl * kStarSlice + lane * V;
  const int64_t t = obs0 / kTileT;
  const int32_t col = (int32_t)(obs0 % kTileT);
  uint32_t slice_epoch = a.slice_epoch0;
  uint32_t* slice_counter = a.barrier + (size_t)(1 + sl) * kStarCounterStride;

  const int32_t fb0 = c * a.chunk_len;
  const int32_t fb1 = fb0 + a.chunk_len < a.K ? fb0 + a.chunk_len : a.K;
  const int32_t per_wave = (fb1 - fb0 + NW - 1) / NW;
  const int32_t lo = fb0 + wave * per_wave < fb1 ? fb0 + wave * per_wave : fb1;
  const int32_t hi = lo + per_wave < fb1 ? lo + per_wave : fb1;

  // this lane's evidence of family k (padding observations, and the second one of an odd batch's
  // last lane, count as all ones)
  auto load_ev = [&](int32_t k, fv<T, V>* ev) {
    if (a.src || a.src_stride) {
      ev[0] = splat<T, V>((T)1);
      ev[1] = splat<T, V>((T)1);
      if (obs0 < a.N) {
        const T* rows = (a.src_stride ? reinterpret_cast<const T*>(a.src_base + (int64_t)k * a.src_stride)
                                      : static_cast<const T*>(a.src[k])) + obs0 * S;
        if (obs0 + 1 < a.N) {
          const fv<T, 4> raw = *reinterpret_cast<const fv<T, 4>*>(rows);
          ev[0][0] = raw[0];
          ev[1][0] = raw[1];
          ev[0][1] = raw[2];
          ev[1][1] = raw[3];
        } else {
          ev[0][0] = rows[0];
          ev[1][0] = rows[1];
        }
      }
    } else {
#pragma unroll
      for (int32_t j = 0; j < S; ++j) ev[j] = loadv<T, V>(a.arena + a.obs_off[k], t * S + j, col);
    }
  };
  if (a.cpt_in_lds) {
    for (int32_t i = threadIdx.x; i < (fb1 - fb0) * E; i += 64 * NW) cpt_lds[i] = a.cpts[a.cpt_off[fb0 + i / E] + i % E];
    __syncthreads();
  }
  auto family_q = [&](int32_t k, const fv<T, V>* ev, fv<T, V>* q) {
    const T* cpt = a.cpt_in_lds ? cpt_lds + (size_t)(k - fb0) * E : a.cpts + a.cpt_off[k];  // (wave-uniform)
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      q[d] = cpt[d * S] * ev[0];
#pragma unroll
      for (int32_t j = 1; j < S; ++j) q[d] = __builtin_elementwise_fma(splat<T, V>(cpt[d * S + j]), ev[j], q[d]);
    }
  };
  T* zs = a.z_scratch + t * kTileT + col;  // + k * npad: this lane's two normalisers of family k
  // The wave's families one after the other, the evidence (and, with_z, the stored normaliser) of
  // four of them in flight: body(k, ev, z).
  constexpr int32_t kRing = 4;
  auto walk = [&](bool with_z, auto&& body) {
    fv<T, V> ring_ev[kRing][S];
    fv<T, V> ring_z[kRing];
#pragma unroll
    for (int32_t u = 0; u < kRing; ++u) {
      ring_z[u] = splat<T, V>((T)1);
      if (lo + u < hi) {
        load_ev(lo + u, ring_ev[u]);
        if (with_z) ring_z[u] = *reinterpret_cast<const fv<T, V>*>(zs + (int64_t)(lo + u) * a.npad);
      }
    }
    for (int32_t k0 = lo; k0 < hi; k0 += kRing) {
#pragma unroll
      for (int32_t u = 0; u < kRing; ++u) {
        const int32_t k = k0 + u;
        if (k < hi) {
          body(k, ring_ev[u], ring_z[u]);
          if (k + kRing < hi) {
            load_ev(k + kRing, ring_ev[u]);
            if (with_z) ring_z[u] = *reinterpret_cast<const fv<T, V>*>(zs + (int64_t)(k + kRing) * a.npad);
          }
        }
      }
    }
  };
  const T* prior = a.cpts + a.root_cpt_off;

  // the block's partial product prod q / prod z of this iteration's messages to the root, into the
  // partial-product buffer of iteration `it`
  auto publish = [&](int32_t it, const double* zm, const int32_t* ze) {
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      z_m[(wave * V + v) * 64 + lane] = zm[v];
      z_e[(wave * V + v) * 64 + lane] = ze[v];
    }
    __syncthreads();
    if (wave == 0) {
      double inv_z[V];
      int32_t ez[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        double m = 1.0;
        int32_t e = 0;
#pragma unroll
        for (int32_t w = 0; w < NW; ++w) {
          int de = 0;
          m = frexp(m * z_m[(w * V + v) * 64 + lane], &de);
          e += z_e[(w * V + v) * 64 + lane] + de;
        }
        inv_z[v] = 1.0 / m;
        ez[v] = e;
      }
      double* pm = a.part_mant + (size_t)(it & 1) * a.C * D0 * a.npad;
      int32_t* pe = a.part_exp + (size_t)(it & 1) * a.C * D0 * a.npad;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const int64_t row = ((int64_t)c * D0 + d) * a.npad + obs0;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          const double m = frexp(bq_m[(d * V + v) * 64 + lane] * inv_z[v], &de);
          shared_store(pm + row + v, m);
          shared_store(pe + row + v, bq_e[(d * V + v) * 64 + lane] + de - ez[v]);
        }
      }
    }
  };

  // ---- first pass: prod_k q_k[d] over the wave's families, and iteration 1 (every message uniform)
  double qm[D0][V];
  int32_t qe[D0][V];
#pragma unroll
  for (int32_t d = 0; d < D0; ++d)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      qm[d][v] = 1.0;
      qe[d][v] = 0;
    }
  {
    double zm[V] = {1.0, 1.0};
    int32_t ze[V] = {0, 0};
    int32_t until = 4;  // four float factors cannot leave the double range
    walk(false, [&](int32_t k, const fv<T, V>* ev, const fv<T, V>&) {
      fv<T, V> q[D0];
      family_q(k, ev, q);
      fv<T, V> own[S];
#pragma unroll
      for (int32_t j = 0; j < S; ++j) own[j] = splat<T, V>((T)1 / (T)S);
      const fv<T, V> zc = leaf_normaliser<T, S, V>(ev, own);
      fv<T, V> sum_r = splat<T, V>(0.f), acc = splat<T, V>(0.f);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const fv<T, V> r = splat<T, V>((T)1) * rcpv<T, V>(splat<T, V>((T)1 / (T)D0));
        sum_r += r;
        acc = __builtin_elementwise_fma(r, q[d], acc);
      }
      const fv<T, V> zn = acc * rcpv<T, V>(sum_r);
      *reinterpret_cast<fv<T, V>*>(zs + (int64_t)k * a.npad) = zn;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        zm[v] *= (double)zc[v];
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) qm[d][v] *= (double)q[d][v];
      }
      if (--until == 0 || sizeof(T) == 8) {
        until = 4;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          zm[v] = frexp(zm[v], &de);
          ze[v] += de;
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) {
            qm[d][v] = frexp(qm[d][v], &de);
            qe[d][v] += de;
          }
        }
      }
    });
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      zm[v] = frexp(zm[v], &de);
      ze[v] += de;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        qm[d][v] = frexp(qm[d][v], &de);
        qe[d][v] += de;
      }
    }
    // the waves' prod q meet in wave 0, two writers at a time: (4,5)->(0,1), (6,7)->(2,3), (2,3)->(0,1), 1->0
    auto hand_over = [&](int32_t from0, int32_t to0, int32_t count) {
      __syncthreads();
      if (wave >= from0 && wave < from0 + count) {
        double* dm = merge_m + (size_t)(wave - from0) * D0 * V * 64;
        int32_t* de = merge_e + (size_t)(wave - from0) * D0 * V * 64;
#pragma unroll
        for (int32_t d = 0; d < D0; ++d)
#pragma unroll
          for (int32_t v = 0; v < V; ++v) {
            dm[(d * V + v) * 64 + lane] = qm[d][v];
            de[(d * V + v) * 64 + lane] = qe[d][v];
          }
      }
      __syncthreads();
      if (wave >= to0 && wave < to0 + count) {
        const double* dm = merge_m + (size_t)(wave - to0) * D0 * V * 64;
        const int32_t* de = merge_e + (size_t)(wave - to0) * D0 * V * 64;
#pragma unroll
        for (int32_t d = 0; d < D0; ++d)
#pragma unroll
          for (int32_t v = 0; v < V; ++v) {
            int x = 0;
            qm[d][v] = frexp(qm[d][v] * dm[(d * V + v) * 64 + lane], &x);
            qe[d][v] += de[(d * V + v) * 64 + lane] + x;
          }
      }
    };
    static_assert(NW == 8, "the hand-over schedule is written for eight waves");
    hand_over(4, 0, 2);
    hand_over(6, 2, 2);
    hand_over(2, 0, 2);
    hand_over(1, 0, 1);
    if (wave == 0) {
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          bq_m[(d * V + v) * 64 + lane] = qm[d][v];
          bq_e[(d * V + v) * 64 + lane] = qe[d][v];
        }
    }
    publish(1, zm, ze);
  }
  slice_epoch += (uint32_t)a.C;
  bool alive = star_barrier(slice_counter, slice_epoch, a.error);
  if (!alive && threadIdx.x == 0) (void)star_give_up(a.barrier, a.epoch0 + (uint32_t)nblocks);

  // The root's scaled product P' (largest state 1) and its normaliser c_Q for this lane's
  // observations, from the blocks' partial products of the iteration before.
  fv<T, V> pf[D0];
  fv<T, V> pf_sum;
  double cq[V] = {0.0, 0.0};
  auto combine = [&](int32_t buffer) {
    const double* pm = a.part_mant + (size_t)buffer * a.C * D0 * a.npad;
    const int32_t* pe = a.part_exp + (size_t)buffer * a.C * D0 * a.npad;
    for (int32_t d = wave; d < D0; d += NW) {
      double m[V] = {1.0, 1.0};
      int32_t e[V] = {0, 0};
      for (int32_t ch = 0; ch < a.C; ch += 16) {
        double2 xm[16];
        int2 xe[16];
#pragma unroll
        for (int32_t u = 0; u < 16; ++u) {
          const bool live = ch + u < a.C;
          const int64_t row = ((int64_t)(live ? ch + u : ch) * D0 + d) * a.npad + obs0;
          xm[u] = make_double2(shared_load(pm + row), shared_load(pm + row + 1));
          xe[u] = make_int2(shared_load(pe + row), shared_load(pe + row + 1));
          if (!live) {
            xm[u] = make_double2(1.0, 1.0);
            xe[u] = make_int2(0, 0);
          }
        }
#pragma unroll
        for (int32_t u = 0; u < 16; ++u) {
          m[0] *= xm[u].x;
          m[1] *= xm[u].y;
          e[0] += xe[u].x;
          e[1] += xe[u].y;
        }
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          m[v] = frexp(m[v], &de);
          e[v] += de;
        }
      }
      const double x = (double)prior[d];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        m[v] = frexp(m[v] * x, &de);
        e[v] = m[v] == 0.0 ? kDeadExponent : e[v] + de;
        st_m[(d * V + v) * 64 + lane] = m[v];
        st_e[(d * V + v) * 64 + lane] = e[v];
      }
    }
    __syncthreads();
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int32_t emax = kDeadExponent;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const int32_t x = st_e[(d * V + v) * 64 + lane];
        emax = x > emax ? x : emax;
      }
      double z = 0.0;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const double scaled = ldexp(st_m[(d * V + v) * 64 + lane], clamp_shift(st_e[(d * V + v) * 64 + lane] - emax));
        z += scaled;
        pf[d][v] = (T)scaled;
      }
      cq[v] = (double)(T)((double)emax * 0.6931471805599453 + log(z));
    }
    pf_sum = pf[0];
#pragma unroll
    for (int32_t d = 1; d < D0; ++d) pf_sum += pf[d];
    __syncthreads();
  };

  // ---- iterations 2 .. T-1: the leaves' next normalisers from P', the block's partial product renewed
  for (int32_t it = 2; it < a.iterations && alive; ++it) {
    combine((it - 1) & 1);
    double zm[V] = {1.0, 1.0};
    int32_t ze[V] = {0, 0};
    int32_t until = 4;
    walk(true, [&](int32_t k, const fv<T, V>* ev, const fv<T, V>& zc) {
      fv<T, V> q[D0];
      family_q(k, ev, q);
      fv<T, V> sum_r = splat<T, V>(0.f);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) sum_r = __builtin_elementwise_fma(pf[d], rcpv<T, V>(q[d]), sum_r);
      const fv<T, V> zn = pf_sum * rcpv<T, V>(sum_r);
      *reinterpret_cast<fv<T, V>*>(zs + (int64_t)k * a.npad) = zn;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) zm[v] *= (double)zc[v];
      if (--until == 0 || sizeof(T) == 8) {
        until = 4;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          zm[v] = frexp(zm[v], &de);
          ze[v] += de;
        }
      }
    });
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      zm[v] = frexp(zm[v], &de);
      ze[v] += de;
    }
    publish(it, zm, ze);
    slice_epoch += (uint32_t)a.C;
    alive = star_barrier(slice_counter, slice_epoch, a.error);
    if (!alive && threadIdx.x == 0) (void)star_give_up(a.barrier, a.epoch0 + (uint32_t)nblocks);
  }

  // ---- iteration T on the fly: log-likelihood and count sums
  if (alive) {
    combine((a.iterations - 1) & 1);
    double zm[V] = {1.0, 1.0};
    int32_t ze[V] = {0, 0};
    int32_t until = 4;
    walk(true, [&](int32_t k, const fv<T, V>* ev, const fv<T, V>& z) {
      fv<T, V> q[D0], r[D0];
      family_q(k, ev, q);
      fv<T, V> inv_sum = splat<T, V>(0.f);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        r[d] = pf[d] * rcpv<T, V>(q[d]);
        inv_sum += r[d];
      }
      inv_sum = rcpv<T, V>(inv_sum);
#pragma unroll
      for (int32_t v = 0; v < V; ++v) zm[v] *= (double)z[v];
      if (--until == 0 || sizeof(T) == 8) {
        until = 4;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          zm[v] = frexp(zm[v], &de);
          ze[v] += de;
        }
      }
      // counts: sum_n t[d] * o[j], t = r / sum r, o = ev / z (padding observations masked)
      const fv<T, V> scale = recipv<T, V>(z) * inv_sum;
      fv<T, V> o[S];
#pragma unroll
      for (int32_t j = 0; j < S; ++j) {
        o[j] = ev[j] * scale;
#pragma unroll
        for (int32_t v = 0; v < V; ++v)
          if (obs0 + v >= a.N) o[j][v] = 0.f;
      }
      T per_lane[E];
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t j = 0; j < S; ++j) {
          T x = r[d][0] * o[j][0];
#pragma unroll
          for (int32_t v = 1; v < V; ++v) x = fma_t(r[d][v], o[j][v], x);
          per_lane[d * S + j] = x;
        }
      WaveSums<T, E> sums;
      sums.reduce(per_lane, lane);
      if (lane < 4) {
#pragma unroll
        for (int32_t g = 0; g < WaveSums<T, E>::H2; ++g) {
          const int32_t e = WaveSums<T, E>::entry(lane, g);
          if (e < E) shared_store(a.count_part + ((int64_t)k * E + e) * a.nslices + sl, (double)sums.out[g]);
        }
      }
    });
    double mine = 0.0;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      const double m = frexp(zm[v], &de);
      if (obs0 + v < a.N) mine += (double)(ze[v] + de) * 0.6931471805599453 + log(m);
    }
    // root: its normaliser and its posterior, by the blocks of chunk 0 (one per slice), wave 0
    if (c == 0 && wave == 0) {
#pragma unroll
      for (int32_t v = 0; v < V; ++v)
        if (obs0 + v < a.N) mine += cq[v];
      double inv[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        double zsum = 0.0;
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) zsum += (double)pf[d][v];
        inv[v] = obs0 + v < a.N ? 1.0 / zsum : 0.0;
      }
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        double x = (double)pf[d][0] * inv[0] + (double)pf[d][1] * inv[1];
#pragma unroll
        for (int32_t off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) shared_store(a.root_part + (int64_t)sl * D0 + d, x);
      }
    }
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) red[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int32_t w = 0; w < NW; ++w) total += red[w];
      shared_store(a.ll_part + blockIdx.x, total);
    }
    alive = star_barrier(a.barrier, a.epoch0 + (uint32_t)nblocks, a.error);
  }

  // ---- M-step (EmBatchOptimizer._m_step, optimizers/em_batch_optimizer.py:70-84), or - the launch
  // given up - the old parameters into the caller's buffer: every block takes a share of the rows
  const int64_t rows = (int64_t)a.K * D0;
  const int64_t per_block = (rows + nblocks - 1) / nblocks;
  const int64_t row0 = (int64_t)blockIdx.x * per_block;
  const int64_t row1 = row0 + per_block < rows ? row0 + per_block : rows;
  if (!alive) {
    if (a.new_cpts) {
      for (int64_t i = row0 * S + threadIdx.x; i < row1 * S; i += 64 * NW) {
        const int64_t off = a.cpt_off[i / E] + i % E;
        a.new_cpts[off] = a.cpts[off];
      }
      if (blockIdx.x == 0 && threadIdx.x < D0) a.new_cpts[a.root_cpt_off + threadIdx.x] = a.cpts[a.root_cpt_off + threadIdx.x];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.ll_out = __builtin_nan("");
    return;
  }
  {
#pragma clang fp contract(off)
  const double gamma = a.regularization, rate = a.learning_rate;
  auto new_row = [&](int64_t off, int32_t len, const double* counts) {
    double sum = 0.0;
    for (int32_t j = 0; j < len; ++j) sum += counts[j];
    const double uniform = 1.0 / (double)len;
    for (int32_t j = 0; j < len; ++j) {
      double x = counts[j] / sum;
      if (gamma != 0.0) {
        const double p = (1.0 - gamma) * x;
        const double u = gamma * uniform;
        x = p + u;
      }
      if (rate >= 0.0) {
        const double old = (1.0 - rate) * (double)a.cpts[off + j];
        const double fresh = rate * x;
        x = old + fresh;
      }
      a.cpts[off + j] = (T)x;
      if (a.new_cpts) a.new_cpts[off + j] = (T)x;
    }
  };
  // a wave per row: the lanes fetch the slices' sums side by side, then add them up
  for (int64_t row = row0 + wave; row < row1; row += NW) {
    const int32_t k = (int32_t)(row / D0), d = (int32_t)(row % D0);
    const int64_t off = a.cpt_off[k] + (int64_t)d * S;
    double counts[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j) {
      const double* src = a.count_part + ((int64_t)k * E + d * S + j) * a.nslices;
      double x = 0.0;
      for (int32_t s2 = lane; s2 < a.nslices; s2 += 64) x += shared_load(src + s2);
#pragma unroll
      for (int32_t o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
      counts[j] = x * (double)a.cpts[off + j];
    }
    if (lane == 0) new_row(off, S, counts);
  }
  if (blockIdx.x == 0) {
    double mine = 0.0;
    for (int32_t b = threadIdx.x; b < nblocks; b += 64 * NW) mine += shared_load(a.ll_part + b);
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    __syncthreads();  // red[] was read by thread 0 before the barrier
    if (lane == 0) red[wave] = mine;
    // the root's counts: wave w sums states w, w + 8, ..
    for (int32_t d = wave; d < D0; d += NW) {
      double x = 0.0;
      for (int32_t s2 = lane; s2 < a.nslices; s2 += 64) x += shared_load(a.root_part + (int64_t)s2 * D0 + d);
#pragma unroll
      for (int32_t o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
      if (lane == 0) red[NW + d] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double counts[D0];
      for (int32_t d = 0; d < D0; ++d) counts[d] = red[NW + d];
      new_row(a.root_cpt_off, D0, counts);
      double ll = 0.0;
      for (int32_t w = 0; w < NW; ++w) ll += red[w];
      *a.ll_out = ll * a.ll_scale;
    }
  }
  }
}

}  // namespace bn
