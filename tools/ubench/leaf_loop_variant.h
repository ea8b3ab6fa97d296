// Experimental steady-state loop launch: evidence through a per-wave LDS ring filled by LDS-DMA.
#pragma once
#include "kernels_leaf.h"

namespace bnx {
using namespace bn;

constexpr int32_t kRing = 8;           // families in flight per wave
constexpr int32_t kSlotBytes = 1024;   // S(2) rows x 128 observations x 4 bytes

typedef const __attribute__((address_space(4))) int64_t* const_i64_ptr;

// one 1 KiB LDS-DMA: lane l moves 16 bytes from base + voff to lds_dst + 16 * l
__device__ __forceinline__ void dma16(const void* base, uint32_t voff, uint32_t lds_dst) {
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

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int D0, int S, int V>
__global__ void __launch_bounds__(256) leaf_loop_v2(float* __restrict__ arena, const float* __restrict__ cpts,
                                                    LeafFamilyArgs<float> a) {
  using T = float;
  static_assert(S == 2 && V == 2, "prototype");
  extern __shared__ __align__(16) unsigned char ring_raw[];
  constexpr int32_t kParts = tile_of<T>::value / (64 * V);
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
  const int32_t n = last - first;

  const uint32_t ring = (uint32_t)(uintptr_t)ring_raw + (uint32_t)wave * (kRing * kSlotBytes);
  // lane l < 32 fetches 16 bytes of state row 0, lane l >= 32 of state row 1
  const uint32_t voff = (uint32_t)((((t * S + (lane >> 5)) * tile_of<T>::value) + part * 64 * V + (lane & 31) * 4) * 4);
  const_i64_ptr obs_tab = (const_i64_ptr)a.obs_off;
  const_i64_ptr cpt_tab = (const_i64_ptr)a.cpt_off;

  double pm[D0][V];
  ev_t<V> pe[D0];
#pragma unroll
  for (int32_t s = 0; s < D0; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      pm[s][v] = 1.0;
      pe[s][v] = 0;
    }
  fv<T, V> pf[D0];
  {
    const double* scaled = a.scaled + t * tile_of<T>::value + col;
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      const double2 lo = *reinterpret_cast<const double2*>(scaled + (int64_t)i * a.npad);
      pf[i][0] = (T)lo.x;
      pf[i][1] = (T)lo.y;
    }
  }
  fv<T, V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  double pz[V] = {1.0, 1.0};
  int32_t pze[V] = {0, 0};

  // everything the compiler's own loads needed has arrived before the first DMA is counted
  wait_vm<0>();
  for (int32_t i = 0; i < kRing && i < n; ++i)
    dma16(arena + obs_tab[first + i], voff, ring + (uint32_t)i * kSlotBytes);

  const unsigned char* my_ring = ring_raw + wave * (kRing * kSlotBytes);
  auto read_ev = [&](int32_t slot, fv<T, V>* ev) {
    ev[0] = *reinterpret_cast<const fv<T, V>*>(my_ring + slot * kSlotBytes + lane * 8);
    ev[1] = *reinterpret_cast<const fv<T, V>*>(my_ring + slot * kSlotBytes + 512 + lane * 8);
  };

  // family `first` is the oldest of the min(kRing, n) requests in flight
  if (n >= kRing) wait_vm<kRing - 1>(); else wait_vm<0>();
  fv<T, V> cur_ev[S], nxt_ev[S];
  read_ev(0, cur_ev);
  const T* z_base = a.z_state + t * tile_of<T>::value + col;
  fv<T, V> cur_z = *reinterpret_cast<const fv<T, V>*>(z_base + (int64_t)first * a.npad);

  for (int32_t i = 0; i < n; ++i) {
    const int32_t k = first + i;
    const T* cpt = cpts + cpt_tab[k];
    // refill the slot family i came from, then make sure family i + 1 has landed
    const int32_t in = i + 1 < n ? i + 1 : i;
    if (i + kRing < n) {
      // the slot's old contents must be in registers before it is overwritten
      asm volatile("" ::"v"(cur_ev[0]), "v"(cur_ev[1]));
      dma16(arena + obs_tab[k + kRing], voff, ring + (uint32_t)(i % kRing) * kSlotBytes);
      wait_vm<kRing - 1>();
    } else {
      wait_vm<0>();
    }
    read_ev(in % kRing, nxt_ev);
    const fv<T, V> nxt_z = *reinterpret_cast<const fv<T, V>*>(z_base + (int64_t)(first + in) * a.npad);

    fv<T, V> q[D0];
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      q[d] = cpt[d * S] * cur_ev[0];
      q[d] = __builtin_elementwise_fma(splat<T, V>(cpt[d * S + 1]), cur_ev[1], q[d]);
    }
    fv<T, V> inv_sum = splat<T, V>(0.f);
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) inv_sum = __builtin_elementwise_fma(pf[d], rcpv<T, V>(q[d]), inv_sum);
    inv_sum = rcpv<T, V>(inv_sum);
    const fv<T, V> zn = pf_sum * inv_sum;
    *reinterpret_cast<fv<T, V>*>(a.z_state + (int64_t)k * a.npad + t * tile_of<T>::value + col) = zn;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) pm[d][v] *= (double)q[d][v];
      pz[v] *= (double)cur_z[v];
    }
    if (((i + 1) & 3) == 0) {
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        ev_t<V> de_v;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          pm[d][v] = frexp(pm[d][v], &de);
          de_v[v] = de;
        }
        pe[d] += de_v;
      }
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        pz[v] = frexp(pz[v], &de);
        pze[v] += de;
      }
    }
    cur_ev[0] = nxt_ev[0];
    cur_ev[1] = nxt_ev[1];
    cur_z = nxt_z;
  }
  double inv_pz[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
    int de = 0;
    inv_pz[v] = 1.0 / frexp(pz[v], &de);
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
    const int64_t row = ((int64_t)(a.chunk_base + chunk) * D0 + s) * a.npad + t * tile_of<T>::value + col;
    *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(pm[s][0], pm[s][1]);
    *reinterpret_cast<int2*>(a.part_exp + row) = make_int2(ex[0], ex[1]);
  }
}


// in-place frexp of a positive double: mantissa back into x, exponent returned
__device__ __forceinline__ int32_t frexp_inplace(double& x) {
  int32_t e;
  asm("v_frexp_exp_i32_f64_e32 %0, %1" : "=v"(e) : "v"(x));
  asm("v_frexp_mant_f64_e32 %0, %0" : "+v"(x));
  return e;
}

// Loop launch with the leaves' normalisers carried per chunk: reads prod_k z_k of its chunk (per
// observation) as mantissa / exponent, writes either the next iteration's chunk product (WRITE_Z =
// false) or the next z of every family (WRITE_Z = true, the iteration before a full / reduce launch).
template <int D0, int S, int V, bool WRITE_Z, bool PAIR = false>
__global__ void __launch_bounds__(256) leaf_loop_v3(float* __restrict__ arena, const float* __restrict__ cpts,
                                                    LeafFamilyArgs<float> a, double* __restrict__ zc_mant,
                                                    int32_t* __restrict__ zc_exp) {
  using T = float;
  static_assert(S == 2 && V == 2, "prototype");
  extern __shared__ __align__(16) unsigned char ring_raw[];
  constexpr int32_t kParts = tile_of<T>::value / (64 * V);
  constexpr int32_t E = D0 * S;
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
  const int32_t n = last - first;

  const uint32_t ring = (uint32_t)(uintptr_t)ring_raw + (uint32_t)wave * (kRing * kSlotBytes);
  const uint32_t voff = (uint32_t)((((t * S + (lane >> 5)) * tile_of<T>::value) + part * 64 * V + (lane & 31) * 4) * 4);
  const_i64_ptr obs_tab = (const_i64_ptr)a.obs_off;
  const_i64_ptr cpt_tab = (const_i64_ptr)a.cpt_off;

  double pm[D0][V];
  ev_t<V> pe[D0];
#pragma unroll
  for (int32_t s = 0; s < D0; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      pm[s][v] = 1.0;
      pe[s][v] = 0;
    }
  fv<T, V> pf[D0];
  const int64_t obs0 = t * tile_of<T>::value + col;
  {
    const double* scaled = a.scaled + obs0;
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      const double2 lo = *reinterpret_cast<const double2*>(scaled + (int64_t)i * a.npad);
      pf[i][0] = (T)lo.x;
      pf[i][1] = (T)lo.y;
    }
  }
  fv<T, V> pf_sum = pf[0];
#pragma unroll
  for (int32_t i = 1; i < D0; ++i) pf_sum += pf[i];
  // this chunk's product of the leaves' current normalisers
  const double2 zin_m = *reinterpret_cast<const double2*>(zc_mant + (int64_t)chunk * a.npad + obs0);
  const int2 zin_e = *reinterpret_cast<const int2*>(zc_exp + (int64_t)chunk * a.npad + obs0);
  double pz[V] = {1.0, 1.0};  // product of the NEXT normalisers
  int32_t pze[V] = {0, 0};

  wait_vm<0>();
  for (int32_t i = 0; i < kRing && i < n; ++i)
    dma16(arena + obs_tab[first + i], voff, ring + (uint32_t)i * kSlotBytes);
  int64_t obs_ahead = obs_tab[first + kRing < last ? first + kRing : last - 1];
  int64_t cpt_next_off = cpt_tab[first + 1 < last ? first + 1 : first];
  float cpt_cur[E];
  {
    const T* c = cpts + cpt_tab[first];
#pragma unroll
    for (int32_t j = 0; j < E; ++j) cpt_cur[j] = c[j];
  }

  const unsigned char* my_ring = ring_raw + wave * (kRing * kSlotBytes);
  auto read_ev = [&](int32_t slot, fv<T, V>* ev) {
    ev[0] = *reinterpret_cast<const fv<T, V>*>(my_ring + slot * kSlotBytes + lane * 8);
    ev[1] = *reinterpret_cast<const fv<T, V>*>(my_ring + slot * kSlotBytes + 512 + lane * 8);
  };
  if (n >= kRing) wait_vm<kRing - 1>(); else wait_vm<0>();
  fv<T, V> cur_ev[S], nxt_ev[S];
  read_ev(0, cur_ev);
  int32_t until_renorm = 6;
  fv<T, V> held[D0], held_z = splat<T, V>(1.f);
#pragma unroll
  for (int32_t d = 0; d < D0; ++d) held[d] = splat<T, V>(1.f);

  for (int32_t i = 0; i < n; ++i) {
    const int32_t k = first + i;
    // next family's CPT and the table entries after it, one iteration ahead
    float cpt_nxt[E];
    {
      const T* c = cpts + cpt_next_off;
#pragma unroll
      for (int32_t j = 0; j < E; ++j) cpt_nxt[j] = c[j];
    }
    cpt_next_off = cpt_tab[k + 2 < last ? k + 2 : last - 1];
    if (i + kRing < n) {
      asm volatile("" ::"v"(cur_ev[0]), "v"(cur_ev[1]));
      dma16(arena + obs_ahead, voff, ring + (uint32_t)(i % kRing) * kSlotBytes);
      obs_ahead = obs_tab[k + kRing + 1 < last ? k + kRing + 1 : last - 1];
      wait_vm<kRing - 1>();
    } else {
      wait_vm<0>();
    }
    const int32_t in = i + 1 < n ? i + 1 : i;
    read_ev(in % kRing, nxt_ev);

    fv<T, V> q[D0];
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) {
      q[d] = cpt_cur[d * S] * cur_ev[0];
      q[d] = __builtin_elementwise_fma(splat<T, V>(cpt_cur[d * S + 1]), cur_ev[1], q[d]);
    }
    fv<T, V> inv_sum = splat<T, V>(0.f);
#pragma unroll
    for (int32_t d = 0; d < D0; ++d) inv_sum = __builtin_elementwise_fma(pf[d], rcpv<T, V>(q[d]), inv_sum);
    inv_sum = rcpv<T, V>(inv_sum);
    const fv<T, V> zn = pf_sum * inv_sum;
    if constexpr (WRITE_Z) *reinterpret_cast<fv<T, V>*>(a.z_state + (int64_t)k * a.npad + obs0) = zn;
    if constexpr (PAIR) {
      // families are folded in pairs: the first of a pair only parks its q
      if ((i & 1) == 0 && i + 1 < n) {
#pragma unroll
        for (int32_t d = 0; d < D0; ++d) held[d] = q[d];
        held_z = zn;
      } else {
        if ((i & 1) == 1) {
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) q[d] *= held[d];
        }
        const fv<T, V> zz = (i & 1) == 1 ? zn * held_z : zn;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
#pragma unroll
          for (int32_t d = 0; d < D0; ++d) pm[d][v] *= (double)q[d][v];
          if constexpr (!WRITE_Z) pz[v] *= (double)zz[v];
        }
      }
    } else {
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) pm[d][v] *= (double)q[d][v];
      if constexpr (!WRITE_Z) pz[v] *= (double)zn[v];
    }
    }
    if (--until_renorm == 0) {
      until_renorm = 6;
#pragma unroll
      for (int32_t d = 0; d < D0; ++d) {
        ev_t<V> de_v;
#pragma unroll
        for (int32_t v = 0; v < V; ++v) de_v[v] = (short)frexp_inplace(pm[d][v]);
        pe[d] += de_v;
      }
      if constexpr (!WRITE_Z) {
#pragma unroll
        for (int32_t v = 0; v < V; ++v) pze[v] += frexp_inplace(pz[v]);
      }
    }
    cur_ev[0] = nxt_ev[0];
    cur_ev[1] = nxt_ev[1];
#pragma unroll
    for (int32_t j = 0; j < E; ++j) cpt_cur[j] = cpt_nxt[j];
  }
  if constexpr (!WRITE_Z) {
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      pz[v] = frexp(pz[v], &de);
      pze[v] += de;
    }
    *reinterpret_cast<double2*>(zc_mant + (int64_t)chunk * a.npad + obs0) = make_double2(pz[0], pz[1]);
    *reinterpret_cast<int2*>(zc_exp + (int64_t)chunk * a.npad + obs0) = make_int2(pze[0], pze[1]);
  }
  const double inv_z[V] = {1.0 / zin_m.x, 1.0 / zin_m.y};
  const int32_t ze[V] = {zin_e.x, zin_e.y};
#pragma unroll
  for (int32_t s = 0; s < D0; ++s) {
    int32_t ex[V];
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      pm[s][v] = frexp(pm[s][v] * inv_z[v], &de);
      ex[v] = (int32_t)pe[s][v] + de - ze[v];
    }
    const int64_t row = ((int64_t)(a.chunk_base + chunk) * D0 + s) * a.npad + obs0;
    *reinterpret_cast<double2*>(a.part_mant + row) = make_double2(pm[s][0], pm[s][1]);
    *reinterpret_cast<int2*>(a.part_exp + row) = make_int2(ex[0], ex[1]);
  }
}

inline void launch_variant(dim3 grid, dim3 block, float* arena, const float* cpts, const LeafFamilyArgs<float>& a) {
  leaf_loop_v2<10, 2, 2><<<grid, block, 4 * kRing * kSlotBytes>>>(arena, cpts, a);
}
inline void launch_v3_pair(dim3 grid, dim3 block, float* arena, const float* cpts, const LeafFamilyArgs<float>& a,
                      double* zc_mant, int32_t* zc_exp) {
  leaf_loop_v3<10, 2, 2, false, true><<<grid, block, 4 * kRing * kSlotBytes>>>(arena, cpts, a, zc_mant, zc_exp);
}
template <bool WRITE_Z>
inline void launch_v3(dim3 grid, dim3 block, float* arena, const float* cpts, const LeafFamilyArgs<float>& a,
                      double* zc_mant, int32_t* zc_exp) {
  leaf_loop_v3<10, 2, 2, WRITE_Z><<<grid, block, 4 * kRing * kSlotBytes>>>(arena, cpts, a, zc_mant, zc_exp);
}
}  // namespace bnx
