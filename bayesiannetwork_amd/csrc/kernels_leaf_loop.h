// Loop launches of the one-parent leaf-family scheme (see kernels_wide.h / kernels_leaf.h): the
// iterations of a bn_iterate / bn_em_statistics call that are not the last one.  Such a launch
// writes nothing a query could read, only the state the next iteration starts from.
//
// What differs from leaf_family_p1:
//   * The evidence of the chunk's families reaches the wave through a private ring of LDS slots
//     filled by LDS-DMA (global_load_lds_dwordx4): several families are in flight per wave without
//     holding registers, so HBM latency hides behind the arithmetic at four waves per SIMD.
//   * Per-family table entries and CPT rows are read with scalar loads one family ahead.
//   * The leaves' normalisers z_k are only needed as their product over the chunk (it scales the
//     parent's partial product).  The chunk's partial product of the messages to the parent is
//     prod_k q_k[s] / prod_k z_k with q_k = cpt_k . ev_k.  The numerator does not change from one
//     iteration of a call to the next (same CPTs, same evidence), so only the call's first loop
//     launch forms it (per state, double mantissa and exponent) and writes it to the partial-product
//     array; the denominators go to the parent's two divisor arrays (LeafFamilyArgs::zdiv_*), which
//     alternate: the launch of iteration t leaves prod z of iteration t + 1 in the array the combine
//     of iteration t has just read (the first launch also writes that of its own iteration into the
//     other one), and the combine divides.  Later launches carry no per-state product at all and
//     read nothing but the evidence and the parent's scaled product.
//   * Only a launch that a full launch follows (kLoopLast) writes one z per family and observation,
//     which that kernel reads; a reduce launch recomputes them (kLeafRecompute).
#pragma once

#include "kernels_leaf.h"

namespace bn {

constexpr int32_t kLoopLight = 0, kLoopLast = 1;
// where the iteration's input state comes from
//   kLoopChunk         an earlier loop launch of the same call (nothing is read: the state is in the
//                      parent's divisor arrays)
//   kLoopFresh         nothing has been sent yet (uniform messages)
//   kLoopFromArrays    the message arrays a full launch wrote
//   kLoopIngest        as kLoopFresh, evidence read from the caller's [N][2] arrays and copied into
//                      the arena on the way (two-state leaves)
//   kLoopIngestStates  as kLoopIngest, from one uint8 state per observation
//   kLoopIngestKeep    as kLoopIngest without the copy: every launch of the call reads the caller's
//                      arrays (an EM step, after which the plan's copy would be void anyway)
//   kLoopChunkCaller   as kLoopChunk, evidence read from the caller's arrays
constexpr int32_t kLoopChunk = 0, kLoopFresh = 1, kLoopFromArrays = 2, kLoopIngest = 3, kLoopIngestStates = 4,
                  kLoopIngestKeep = 5, kLoopChunkCaller = 6;
constexpr bool loop_reads_caller(int32_t state) {
  return state == kLoopIngest || state == kLoopIngestKeep || state == kLoopChunkCaller;
}
constexpr bool loop_is_light(int32_t state) { return state == kLoopChunk || state == kLoopChunkCaller; }

typedef const __attribute__((address_space(4))) int64_t* const_table_ptr;  // read through the scalar cache

// One LDS-DMA instruction: lane l moves 16 bytes from base + voff(l) to LDS byte address lds_dst + 16 l.
// The compiler does not count it: its completion is waited for with wait_vmcnt below.
__device__ __forceinline__ void lds_dma16(const void* base, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_dst), "s"(base)
      : "memory");
}

// Waits until at most N of the wave's vector-memory operations are outstanding (they complete in
// issue order, so everything but the N youngest has arrived).
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// frexp of a non-negative double in place; returns the exponent (0 for 0)
__device__ __forceinline__ int32_t frexp_in_place(double& x) {
  int32_t e;
  asm("v_frexp_exp_i32_f64_e32 %0, %1" : "=v"(e) : "v"(x));
  asm("v_frexp_mant_f64_e32 %0, %0" : "+v"(x));
  return e;
}

// Geometry of the evidence ring for element type T and S states (two observations per lane).
// LIGHT: the launch carries no per-state product (STATE == kLoopChunk): it runs with half the
// registers, so twice the waves are resident and each needs half as many families in flight.
template <typename T, int S, bool FROM_CALLER, bool LIGHT = false>
struct LeafRing {
  static constexpr int32_t kRowBytes = 64 * 2 * (int32_t)sizeof(T);        // one state row of the wave's observations
  static constexpr int32_t kFamilyBytes = S * kRowBytes;                    // the same bytes in the caller's [N][S] layout
  static constexpr int32_t kDmas = (kFamilyBytes + 1023) / 1024;            // DMA instructions per family
  static constexpr int32_t kSlotBytes = kDmas * 1024;
  static constexpr int32_t kBudget = LIGHT ? 4096 : 8192;
  static constexpr int32_t kMaxDepth = LIGHT ? 4 : 8;
  static constexpr int32_t kDepth =
      kBudget / kSlotBytes >= kMaxDepth ? kMaxDepth : (kBudget / kSlotBytes >= 2 ? kBudget / kSlotBytes : 2);
  static constexpr int32_t kWaveBytes = kDepth * kSlotBytes;
};

// SPLIT: the waves of a block share one (chunk, tile, part) work item, each taking a quarter of the
// chunk's families (LeafFamilyArgs::split; an instantiation of its own so that the merge of the
// quarters costs the ordinary launches no registers).
template <typename T, int D0, int S, int OUT, int STATE, bool SPLIT = false>
__global__ void __launch_bounds__(256) leaf_loop_p1(T* __restrict__ arena, const T* __restrict__ cpts,
                                                   LeafFamilyArgs<T> a) {
  constexpr int32_t V = 2;
  constexpr bool LAST = OUT == kLoopLast;
  // INGEST: the evidence comes from the caller's [N][S] arrays; COPY: ... and is written to the arena
  constexpr bool INGEST = loop_reads_caller(STATE), INGEST_STATES = STATE == kLoopIngestStates;
  constexpr bool COPY = STATE == kLoopIngest || INGEST_STATES;
  constexpr bool FRESH = STATE == kLoopFresh || STATE == kLoopIngest || STATE == kLoopIngestKeep || INGEST_STATES;
  // the first loop launch of a call forms prod q per state; the others reuse what it wrote
  constexpr bool PROD = !loop_is_light(STATE);
  static_assert(!(INGEST || INGEST_STATES) || S == 2, "evidence is taken from the caller's arrays for two-state leaves only");
  using Ring = LeafRing<T, S, INGEST, !PROD>;
  constexpr int32_t R = Ring::kDepth, ND = Ring::kDmas;
  // vector-memory operations issued per family besides the DMAs (they sit between the DMAs in the
  // queue, so the counted waits allow for them): the z store of kLoopLast, the arena copy of the
  // evidence when ingesting
  constexpr int32_t kOther = (LAST ? 1 : 0) + (COPY ? S : 0);
  constexpr int32_t kAhead = (R - 1) * (ND + kOther);
  constexpr int32_t kParts = tile_of<T>::value / (64 * V);
  constexpr int32_t E = D0 * S;
  extern __shared__ __align__(16) unsigned char ring_raw[];

  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr bool split = SPLIT;  // the block's waves share one work item
  int64_t w = split ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles * kParts) return;
  const int32_t part = (int32_t)(w % kParts);
  w /= kParts;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;
  const int32_t col = part * 64 * V + lane * V;  // first observation of this lane inside the tile
  const int64_t obs0 = t * tile_of<T>::value + col;
  int32_t first = chunk * a.chunk_len;
  int32_t last = first + a.chunk_len < a.K ? first + a.chunk_len : a.K;
  if (first >= last) return;
  // A wave of a sharing block whose quarter is empty walks the chunk's last family instead and
  // contributes nothing (it has to reach the block's barrier anyway).
  bool spare = false;
  if (split) {
    const int32_t quarter = (last - first + kWavesPerBlock - 1) / kWavesPerBlock;
    const int32_t lo = first + wave * quarter;
    const int32_t hi = lo + quarter < last ? lo + quarter : last;
    spare = lo >= hi;
    first = spare ? last - 1 : lo;
    last = spare ? last : hi;
  }
  const int32_t n = last - first;

  const_table_ptr obs_tab = (const_table_ptr)a.obs_off;
  const_table_ptr cpt_tab = (const_table_ptr)a.cpt_off;
  const_table_ptr src_tab = (const_table_ptr)a.src;
  // the caller's array of leaf k, as an address held in 64 bits
  auto caller_source = [&](int32_t k) -> int64_t {
    return a.src_stride ? (int64_t)(uintptr_t)a.src_base + (int64_t)k * a.src_stride : src_tab[k];
  };

  // Byte offset, from the start of a family's evidence, of the 16 bytes this lane fetches in each
  // of the family's DMA instructions.  Arena layout: [tile][state][observation]; the wave's piece of
  // state row r is kRowBytes long.  Caller layout: [observation][state], the wave's observations
  // are one contiguous run (clamped at the end of the array: padding observations read real ones).
  uint32_t voff[ND];
#pragma unroll
  for (int32_t j = 0; j < ND; ++j) {
    const int32_t at = j * 1024 + lane * 16;  // byte within the slot
    if constexpr (INGEST) {
      int64_t byte = (t * tile_of<T>::value + part * 64 * V) * (int64_t)(S * sizeof(T)) + at;
      const int64_t end = a.N * (int64_t)(S * sizeof(T)) - 16;
      byte = byte < end ? byte : end;
      voff[j] = (uint32_t)byte;
    } else {
      int32_t row = at / Ring::kRowBytes;
      row = row < S ? row : S - 1;
      voff[j] = (uint32_t)(((t * S + row) * tile_of<T>::value + part * 64 * V) * (int64_t)sizeof(T) + at % Ring::kRowBytes);
    }
  }
  const uint32_t ring = (uint32_t)(uintptr_t)ring_raw + (uint32_t)wave * Ring::kWaveBytes;
  const unsigned char* my_ring = ring_raw + wave * Ring::kWaveBytes;
  auto family_source = [&](int32_t k) -> const void* {
    if constexpr (INGEST) return reinterpret_cast<const void*>(caller_source(k));
    return arena + obs_tab[k];
  };
  auto request = [&](const void* source, int32_t slot) {
#pragma unroll
    for (int32_t j = 0; j < ND; ++j) lds_dma16(source, voff[j], ring + (uint32_t)(slot * Ring::kSlotBytes + j * 1024));
  };
  // this lane's evidence of the family in `slot`
  auto read_slot = [&](int32_t slot, fv<T, V>* ev) {
    const unsigned char* p = my_ring + slot * Ring::kSlotBytes;
    if constexpr (INGEST) {
      // [observation][state]: two observations x two states are consecutive
      const fv<T, 4> raw = *reinterpret_cast<const fv<T, 4>*>(p + lane * (4 * sizeof(T)));
      ev[0][0] = raw[0];
      ev[1][0] = raw[1];
      ev[0][1] = raw[2];
      ev[1][1] = raw[3];
    } else {
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
        ev[j] = *reinterpret_cast<const fv<T, V>*>(p + j * Ring::kRowBytes + lane * (V * sizeof(T)));
    }
  };
  // compact hard evidence: this lane's two observations are two consecutive bytes (N is even)
  auto states_pair = [&](int32_t k) -> uint32_t {
    const bool tail = obs0 + 1 >= a.N;
    const int64_t at = tail ? a.N - 2 : obs0;
    return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(caller_source(k)) + at);
  };
  auto expand_states = [&](uint32_t pair, fv<T, V>* ev) {
    const uint32_t s0 = pair & 0xffu, s1 = pair >> 8;
    ev[0][0] = s0 == 0 ? a.on_value : a.off_value;
    ev[1][0] = s0 == 1 ? a.on_value : a.off_value;
    ev[0][1] = s1 == 0 ? a.on_value : a.off_value;
    ev[1][1] = s1 == 1 ? a.on_value : a.off_value;
  };

  double pm[D0][V];
  ev_t<V> pe[D0];
#pragma unroll
  for (int32_t s = 0; s < D0; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      pm[s][v] = 1.0;
      pe[s][v] = 0;
    }
  // the parent's scaled product P' (largest state = 1) for this lane's observations, in T
  fv<T, V> pf[D0];
#pragma unroll
  for (int32_t i = 0; i < D0; ++i) {
    const double2 lo = *reinterpret_cast<const double2*>(a.scaled + (int64_t)i * a.npad + obs0);
    pf[i][0] = (T)lo.x;
    pf[i][1] = (T)lo.y;
  }
  fv<T, V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  // prod_k z_k of this iteration (read, or multiplied up below) and of the next one
  double zin_m[V] = {1.0, 1.0};
  int32_t zin_e[V] = {0, 0};
  double pz[V] = {1.0, 1.0};
  int32_t pze[V] = {0, 0};

  // the loads above have arrived before the first DMA is counted
  wait_vmcnt<0>();
  fv<T, V> cur_ev[S], nxt_ev[S];
  uint32_t nxt_pair = 0;
  if constexpr (INGEST_STATES) {
    expand_states(states_pair(first), cur_ev);
  } else {
    for (int32_t i = 0; i < R && i < n; ++i) request(family_source(first + i), i);
    if (n >= R) {
      wait_vmcnt<(R - 1) * ND>();
    } else {
      wait_vmcnt<0>();
    }
    read_slot(0, cur_ev);
  }
  // table entries and CPT rows travel one family ahead through the scalar cache
  int64_t ahead_off = 0;
  if constexpr (!INGEST_STATES) {
    const int32_t ka = first + R < last ? first + R : last - 1;
    ahead_off = INGEST ? caller_source(ka) : obs_tab[ka];
  }
  int64_t cpt_next_off = cpt_tab[first + 1 < last ? first + 1 : first];
  T cpt_cur[E];
  {
    const T* c = cpts + cpt_tab[first];
#pragma unroll
    for (int32_t j = 0; j < E; ++j) cpt_cur[j] = c[j];
  }
  int32_t until_renorm = sizeof(T) == 8 ? 1 : 6;  // six float factors cannot leave the double range

  for (int32_t i = 0; i < n; ++i) {
    const int32_t k = first + i;
    T cpt_nxt[E];
    {
      const T* c = cpts + cpt_next_off;
#pragma unroll
      for (int32_t j = 0; j < E; ++j) cpt_nxt[j] = c[j];
    }
    cpt_next_off = cpt_tab[k + 2 < last ? k + 2 : last - 1];
    const int32_t in = i + 1 < n ? i + 1 : i;
    if constexpr (INGEST_STATES) {
      nxt_pair = states_pair(first + in);
    } else {
      if (i + R < n) {
        // the slot of family i is free once its evidence is in registers
#pragma unroll
        for (int32_t j = 0; j < S; ++j) asm volatile("" ::"v"(cur_ev[j]));
        const void* source = INGEST ? reinterpret_cast<const void*>(ahead_off) : static_cast<const void*>(arena + ahead_off);
        request(source, i % R);
        const int32_t ka = k + R + 1 < last ? k + R + 1 : last - 1;
        ahead_off = INGEST ? caller_source(ka) : obs_tab[ka];
        // family i + 1 must have arrived.  Younger than its DMAs are, in the steady state, R - 1
        // families' worth of operations; during the first R - 1 iterations fewer of the other
        // operations have been issued, and only the DMAs can be relied on to be there
        if (kOther == 0 || i >= R - 1) {
          wait_vmcnt<kAhead>();
        } else {
          wait_vmcnt<(R - 1) * ND>();
        }
      } else {
        wait_vmcnt<0>();
      }
      read_slot(in % R, nxt_ev);
    }
    if constexpr (COPY) {
      // the arena copy of the evidence (an unobserved leaf rewrites the shared all-ones array)
      T* dst = arena + obs_tab[k];
#pragma unroll
      for (int32_t j = 0; j < S; ++j) storev<T, V>(dst, t * S + j, col, cur_ev[j]);
    }

    // q = cpt . ev: the factor's message to the parent is q / z, and q also stands in for the
    // previous such message in the parent's emission (which is normalised over the states)
    fv<T, V> q[D0];
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      q[d] = cpt_cur[d * S] * cur_ev[0];
#pragma unroll
      for (int32_t j = 1; j < S; ++j) q[d] = __builtin_elementwise_fma(splat<T, V>(cpt_cur[d * S + j]), cur_ev[j], q[d]);
    }
    // this iteration's z of the leaf where it is not part of the state that was read
    if constexpr (PROD) {
      fv<T, V> own[S];
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
        own[j] = FRESH ? splat<T, V>((T)1 / (T)S) : loadv<T, V>(arena + a.f_out[2 * k + 1], t * S + j, col);
      const fv<T, V> z = leaf_normaliser<T, S, V>(cur_ev, own);
#pragma unroll
      for (int32_t v = 0; v < V; ++v) zin_m[v] *= (double)z[v];
    }
    // the leaf's next normaliser z' = sum_i t_i q_i with t = r / sum r, r = P' / (previous message
    // to the parent); in the steady state r = P' / q and z' = sum(P') / sum(r)
    fv<T, V> zn;
    if constexpr (!PROD) {
      fv<T, V> sum_r = splat<T, V>(0.f);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) sum_r = __builtin_elementwise_fma(pf[d], rcpv<T, V>(q[d]), sum_r);
      zn = pf_sum * rcpv<T, V>(sum_r);
    } else {
      fv<T, V> sum_r = splat<T, V>(0.f), acc = splat<T, V>(0.f);
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        const fv<T, V> prev = FRESH ? splat<T, V>((T)1 / (T)D0) : loadv<T, V>(arena + a.f_out[2 * k], t * D0 + d, col);
        const fv<T, V> r = pf[d] * rcpv<T, V>(prev);
        sum_r += r;
        acc = __builtin_elementwise_fma(r, q[d], acc);
      }
      zn = acc * rcpv<T, V>(sum_r);
    }
    if constexpr (LAST) storev<T, V>(a.z_state + (int64_t)k * a.npad, t, col, zn);
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      if constexpr (PROD) {
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) pm[d][v] *= (double)q[d][v];
      }
      if constexpr (!LAST) pz[v] *= (double)zn[v];
    }
    if (--until_renorm == 0) {
      until_renorm = sizeof(T) == 8 ? 1 : 6;
      if constexpr (PROD) {
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) {
          ev_t<V> de_v;
#pragma unroll
          for (int32_t v = 0; v < V; ++v) de_v[v] = frexp_in_place(pm[d][v]);
          pe[d] += de_v;
        }
      }
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        if constexpr (!LAST) pze[v] += frexp_in_place(pz[v]);
        if constexpr (PROD) zin_e[v] += frexp_in_place(zin_m[v]);
      }
    }
    if constexpr (INGEST_STATES) {
      expand_states(nxt_pair, cur_ev);
    } else {
#pragma unroll
      for (int32_t j = 0; j < S; ++j) cur_ev[j] = nxt_ev[j];
    }
#pragma unroll
    for (int32_t j = 0; j < E; ++j) cpt_cur[j] = cpt_nxt[j];
  }
  if constexpr (SPLIT) {
    // the four quarters' products meet in wave 0 (every factor brought to [0.5, 1) first).  With the
    // per-state product they need room of their own after the rings; without it they fit into the
    // ring of the wave that writes them, which it has finished reading by then.
    constexpr int32_t kDoubles = PROD ? D0 * V + 2 * V : V, kInts = kDoubles;
    static_assert(PROD || kDoubles * 64 * 12 <= Ring::kWaveBytes, "a wave's record does not fit into its ring");
    auto record_m = [&](int32_t wv) -> double* {  // wave wv's mantissas (wv = 1..)
      if constexpr (PROD)
        return reinterpret_cast<double*>(ring_raw + kWavesPerBlock * Ring::kWaveBytes) + (size_t)(wv - 1) * kDoubles * 64;
      return reinterpret_cast<double*>(ring_raw + wv * Ring::kWaveBytes);
    };
    auto record_e = [&](int32_t wv) -> int32_t* {
      if constexpr (PROD)
        return reinterpret_cast<int32_t*>(record_m(kWavesPerBlock)) + (size_t)(wv - 1) * kInts * 64;
      return reinterpret_cast<int32_t*>(record_m(wv) + kDoubles * 64);
    };
    constexpr int32_t kZ = PROD ? D0 * V : 0;  // where the z products start in a wave's record
    if constexpr (PROD) {
#pragma unroll
      for (int32_t d = 0; d < D0; ++d)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          pe[d][v] += (short)frexp_in_place(pm[d][v]);
          if (spare) {
            pm[d][v] = 0.5;
            pe[d][v] = 1;
          }
        }
    }
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      pze[v] += frexp_in_place(pz[v]);
      if constexpr (PROD) zin_e[v] += frexp_in_place(zin_m[v]);
      if (spare) {
        pz[v] = 0.5;
        pze[v] = 1;
        zin_m[v] = 0.5;
        zin_e[v] = 1;
      }
    }
    if (wave > 0) {
      double* dm = record_m(wave);
      int32_t* de = record_e(wave);
      if constexpr (PROD) {
#pragma unroll
        for (int32_t d = 0; d < D0; ++d)
#pragma unroll
          for (int32_t v = 0; v < V; ++v) {
            dm[(d * V + v) * 64 + lane] = pm[d][v];
            de[(d * V + v) * 64 + lane] = pe[d][v];
          }
      }
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        dm[(kZ + v) * 64 + lane] = pz[v];
        de[(kZ + v) * 64 + lane] = pze[v];
        if constexpr (PROD) {
          dm[(kZ + V + v) * 64 + lane] = zin_m[v];
          de[(kZ + V + v) * 64 + lane] = zin_e[v];
        }
      }
    }
    __syncthreads();
    if (wave > 0) return;
    for (int32_t j = 0; j < kWavesPerBlock - 1; ++j) {
      const double* dm = record_m(j + 1);
      const int32_t* de = record_e(j + 1);
      if constexpr (PROD) {
#pragma unroll
        for (int32_t d = 0; d < D0; ++d)
#pragma unroll
          for (int32_t v = 0; v < V; ++v) {
            pm[d][v] *= dm[(d * V + v) * 64 + lane];
            pe[d][v] += (short)de[(d * V + v) * 64 + lane];
          }
      }
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        pz[v] *= dm[(kZ + v) * 64 + lane];
        pze[v] += de[(kZ + v) * 64 + lane];
        if constexpr (PROD) {
          zin_m[v] *= dm[(kZ + V + v) * 64 + lane];
          zin_e[v] += de[(kZ + V + v) * 64 + lane];
        }
      }
    }
  }
  const int64_t zat = (int64_t)(a.chunk_base + chunk) * a.npad + obs0;
  if constexpr (!LAST) {
    // prod z of the next iteration: what the combine after the next one divides by
    double2 m;
    int2 e;
    int de = 0;
    m.x = frexp(pz[0], &de);
    e.x = pze[0] + de;
    m.y = frexp(pz[1], &de);
    e.y = pze[1] + de;
    *reinterpret_cast<double2*>(a.zdiv_next_mant + zat) = m;
    *reinterpret_cast<int2*>(a.zdiv_next_exp + zat) = e;
  }
  if constexpr (PROD) {
    // prod z of this iteration: what the next combine divides by
    double2 m;
    int2 e;
    int de = 0;
    m.x = frexp(zin_m[0], &de);
    e.x = zin_e[0] + de;
    m.y = frexp(zin_m[1], &de);
    e.y = zin_e[1] + de;
    *reinterpret_cast<double2*>(a.zdiv_mant + zat) = m;
    *reinterpret_cast<int2*>(a.zdiv_exp + zat) = e;
  }
  // ... and its numerator prod q, once per call
  if constexpr (PROD) {
#pragma unroll
    for (int32_t s = 0; s < D0; ++s) {
      int32_t ex[V];
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        pm[s][v] = frexp(pm[s][v], &de);
        ex[v] = (int32_t)pe[s][v] + de;
      }
      const int64_t row = ((int64_t)(a.chunk_base + chunk) * D0 + s) * a.npad + obs0;
      *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(pm[s][0], pm[s][1]);
      *reinterpret_cast<int2*>(a.part_exp + row) = make_int2(ex[0], ex[1]);
    }
  }
}

}  // namespace bn
