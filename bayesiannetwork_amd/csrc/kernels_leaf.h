// Leaf-family kernels (see kernels_wide.h for the scheme they belong to), for float and double
// plans alike: T is the plan's element type, a tile holds tile_of<T>::value observations.
#pragma once

#include "kernels_wide.h"

namespace bn {

// ---------------------------------------------------------------------------------
// Leaf families with one (wide) parent: see the file header.
// The evidence rows the three forms of compact hard evidence stand for: e[2 y + j] = row entry j of an
// observation of form y (0, 1: the leaf's states; 2: neither).  State bytes given with an on and an off
// value: (on, off), (off, on), (off, off).
struct HardEvidenceRows {
  float e[6];
};

template <typename T>
struct LeafFamilyArgs {
  const int64_t* f_in;     // [K*arity]  v2f offsets of the factor group (parent slots, own slot)
  const int64_t* f_out;    // [K*arity]  f2v offsets
  const int64_t* cpt_off;  // [K]
  const int64_t* obs_off;  // [K]    observation array of the own variable, -1 = not observed
  const int64_t* c_off;    // [K]    normaliser of the own variable
  int32_t K;
  int32_t chunks, chunk_len;  // this group's chunks of families
  int32_t chunk_base;         // where they sit among the parent's chunk partials
  double* part_mant;
  int32_t* part_exp;
  // what the parent's combine divides each chunk's partial product by, per observation (mantissa in
  // [0.5, 1) and exponent): loop launches keep prod_k z_k here and leave it out of the partial
  // product, a full launch writes 1
  double* zdiv_mant;   // [parent's chunks][npad]  the array the parent's next combine reads
  int32_t* zdiv_exp;   // [parent's chunks][npad]
  double* zdiv_next_mant;  // the array the combine after that reads
  int32_t* zdiv_next_exp;
  const double* scaled_prev;  // parent's P' of the previous iteration (kLeafRecompute)
  const double* scaled;  // parent's P'
  int64_t npad;
  int64_t ntiles;
  T* z_state;  // [K][npad] loop state: the leaf's normaliser z for the coming iteration
  // kLeafReduce only
  int64_t N;
  T* em_partial;  // [K][D0*S][slots] per-wave sums of (message from parent) x (message from own variable); nullptr = not wanted
  int32_t slots;      // ntiles * waves per tile
  double* ll_part;    // [chunks][npad] sum over the chunk's families of c = log z, per observation
  // two-parent families (leaf_family_p2) only: the second wide parent's partial products and P',
  // and the loop state that holds the factor's previous messages to both parents
  double* part_mant2;
  int32_t* part_exp2;
  const double* scaled2;
  T* m_state;  // [K][D0 + D1][npad]  (ratios to each parent's last state: rows D0 - 1 and D0 + D1 - 1 unused)
  // ingesting loop launches only: per family, the caller's evidence array ([N][2], row-major) of its
  // leaf, or the leaf's stand-in observation array in the arena (all ones) when it is not observed
  const void* const* src;
  // ... or, when those arrays lie at equal distances (slices of one buffer, what stacking the
  // evidence and unbinding it gives), leaf k's array is src_base + k * src_stride bytes and no table
  // is read (src_stride = 0: use the table)
  const unsigned char* src_base;
  int64_t src_stride;
  // compact hard evidence read in place (kernels_hard.h): the step's tables of q and 1 / q per evidence form
  const float* hard_tab;
  T on_value, off_value;  // evidence given as states: what an observed / any other state's row receives
  HardEvidenceRows hard_rows;  // ... and the rows of the forms of compact hard evidence read in place (kernels_hard.h)
  // leaf_loop_p1 only: 1, or kWavesPerBlock when the waves of a block share one (chunk, tile, part)
  // work item, each taking a quarter of the chunk's families (small batches: as many waves as
  // before, a quarter of the chunk partials for the parent's combine to read)
  int32_t split;
  // hard_reduce_mfma only: the rows of normalisers' products the last hard_loop_light launch left (in zdiv_next_*)
  int32_t z_rows;
};
// Template parameter MODE
//   kLeafLoop    inside a bn_iterate call: only the loop state is written (one-parent families run
//                these launches on leaf_loop_p1, kernels_leaf_loop.h; leaf_family_p2 has the mode);
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
//   kLeafFromArrays  the message arrays a kLeafFull launch wrote;
//   kLeafRecompute   like kLeafSteady, but the leaves' normalisers are not read: z = sum(P'prev) /
//                    sum(P'prev / q) is formed again from the parent's scaled product of the previous
//                    iteration (exactly what the loop launch of that iteration would have stored),
//                    so that launch does not have to run at all.  P'prev is kept in LDS, per lane.
constexpr int32_t kLeafSteady = 0, kLeafFresh = 1, kLeafFromArrays = 2, kLeafRecompute = 3;
constexpr int32_t kLeafCaller = 12;  // added to MODE: evidence from the caller's arrays (leaf_state)
// dynamic LDS of a kLeafRecompute launch
template <typename T, int D0, int V>
constexpr size_t leaf_recompute_lds_bytes() {
  return (size_t)kWavesPerBlock * D0 * 64 * V * sizeof(T);
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int64_t bits = __builtin_bit_cast(int64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bits, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bits >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, (int64_t)(((uint64_t)hi << 32) | lo));
}

// hardware reciprocal estimate for float, exact division for double
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }

// Sums E per-lane values over the 64 lanes of the wave with fewer operations than E full
// reductions: the first two steps are a reduce-scatter inside each quad (after them a lane only
// carries a quarter of the entries: those of its position in the quad), the rest adds the 16
// quads (row rotations, then the two cross-row exchanges).  Afterwards lane q < 4 holds in
// out[f] the total of entry index wave_sum_entry<E>(q, f) (or nothing if that is >= E).
template <typename T, int E>
struct WaveSums {
  static constexpr int H1 = (E + 1) / 2, H2 = (H1 + 1) / 2;
  T out[H2];
  __device__ __forceinline__ static int32_t entry(int32_t q, int32_t f) {
    const int32_t e = f + ((q >> 1) & 1) * H2;  // index after the first step
    return e < H1 ? e + (q & 1) * H1 : E;
  }
  // the two steps inside the quads: afterwards out[f] is the quad's sum of this lane's entries
  __device__ __forceinline__ void scatter(const T (&p)[E], int32_t lane) {
    const bool odd = lane & 1, upper = lane & 2;
    T y[H1];
#pragma unroll
    for (int32_t e = 0; e < H1; ++e) {
      const T a = p[e], b = e + H1 < E ? p[e + H1] : 0.f;
      y[e] = (odd ? b : a) + dpp_move<0xB1>(odd ? a : b);  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int32_t f = 0; f < H2; ++f) {
      const T a = y[f], b = f + H2 < H1 ? y[f + H2] : 0.f;
      out[f] = (upper ? b : a) + dpp_move<0x4E>(upper ? a : b);  // quad_perm [2,3,0,1]
    }
  }
  // the sum over the sixteen quads: a chain of dependent cross-lane steps with little to overlap
  // among themselves, so callers run it next to independent arithmetic (the next family's)
  __device__ __forceinline__ void finish() {
#pragma unroll
    for (int32_t f = 0; f < H2; ++f) {
      T w = out[f];
      w += dpp_move<0x124>(w);  // row_ror:4
      w += dpp_move<0x128>(w);  // row_ror:8
      w += __shfl_xor(w, 16, 64);
      w += __shfl_xor(w, 32, 64);
      out[f] = w;
    }
  }
  __device__ __forceinline__ void reduce(const T (&p)[E], int32_t lane) {
    scatter(p, lane);
    finish();
  }
};

// V floats of one lane as a hardware vector: arithmetic on it compiles to the packed
// two-T instructions (v_pk_mul/add/fma_f32) when V is even.
template <typename T, int V>
using fv = T __attribute__((ext_vector_type(V)));

template <typename T, int V>
__device__ __forceinline__ fv<T, V> loadv(const T* __restrict__ base, int64_t row, int32_t col) {
  return *reinterpret_cast<const fv<T, V>*>(base + row * tile_of<T>::value + col);
}

template <typename T, int V>
__device__ __forceinline__ void storev(T* __restrict__ base, int64_t row, int32_t col, const fv<T, V>& r) {
  *reinterpret_cast<fv<T, V>*>(base + row * tile_of<T>::value + col) = r;
}

template <typename T, int V>
__device__ __forceinline__ fv<T, V> splat(T x) {
  fv<T, V> r;
#pragma unroll
  for (int32_t v = 0; v < V; ++v) r[v] = x;
  return r;
}

template <typename T, int V>
__device__ __forceinline__ fv<T, V> rcpv(const fv<T, V>& x) {
  fv<T, V> r;
#pragma unroll
  for (int32_t v = 0; v < V; ++v) r[v] = fast_rcp(x[v]);
  return r;
}

// V observations per lane: a wave covers 64*V observations, i.e. 4/V work items per tile.
//
// The loop over the chunk's families is software-pipelined by hand: the small loop state of
// family k+1 (its pointers, observation, normaliser) is requested before the arithmetic of
// family k, so that with a few waves per SIMD the memory latency of one family hides behind the
// arithmetic of the previous one.
template <typename T>
struct LeafPtrs {  // wave-uniform
  T* f2v_par;
  T* f2v_own;
  T* v2f_par;
  T* v2f_own;
  T* c;
  T* z;          // loop-state normaliser of this family
  const T* obs;  // the observation, or an all-ones array when the leaf is not observed
  const T* cpt;
};

template <typename T, int S, int V>
struct LeafState {
  fv<T, V> ev[S];
  fv<T, V> z;  // the leaf's normaliser for this iteration: sum_s ev[s] * (previous message from the factor)[s]
};

template <typename T, bool CALLER = false>
__device__ __forceinline__ LeafPtrs<T> leaf_ptrs(T* __restrict__ arena, const T* __restrict__ cpts,
                                                 const LeafFamilyArgs<T>& a, int32_t k) {
  LeafPtrs<T> p;
  p.f2v_par = arena + a.f_out[2 * k];
  p.f2v_own = arena + a.f_out[2 * k + 1];
  p.v2f_par = arena + a.f_in[2 * k];
  p.v2f_own = arena + a.f_in[2 * k + 1];
  p.c = arena + a.c_off[k];
  p.z = a.z_state + (int64_t)k * a.npad;
  if constexpr (CALLER) {
    // (a launch that reads the caller's arrays here requires them at equal distances: no table)
    p.obs = reinterpret_cast<const T*>(a.src_base + (int64_t)k * a.src_stride);
  } else {
    p.obs = arena + a.obs_off[k];
  }
  p.cpt = cpts + a.cpt_off[k];
  return p;
}

// z of one leaf from its evidence and the factor's previous message
template <typename T, int S, int V>
__device__ __forceinline__ fv<T, V> leaf_normaliser(const fv<T, V>* ev, const fv<T, V>* from_factor) {
  fv<T, V> z = ev[0] * from_factor[0];
#pragma unroll
  for (int32_t j = 1; j < S; ++j) z = __builtin_elementwise_fma(ev[j], from_factor[j], z);
  return z;
}

// 1/x to within an ulp: hardware seed plus one Newton step
template <typename T, int V>
__device__ __forceinline__ fv<T, V> recipv(const fv<T, V>& x) {
  const fv<T, V> r = rcpv<T, V>(x);
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-x, r, splat<T, V>(1.f)), r, r);
}

// CALLER: the evidence is read from the caller's [N][S] array of the leaf (p.obs points there): a
// lane's observations are one contiguous run of V * S values, fetched in 16-byte pieces that are
// clamped to the end of the array (padding observations then read real ones; they are masked out of
// every sum).  Two-state leaves of float plans with four observations per lane.
template <typename T, int S, int V, int STATE, bool CALLER = false>
__device__ __forceinline__ LeafState<T, S, V> leaf_state(const LeafPtrs<T>& p, const LeafFamilyArgs<T>& a, int64_t t,
                                                      int32_t col) {
  LeafState<T, S, V> st;
  if constexpr (CALLER) {
    static_assert(S == 2 && V == 4 && sizeof(T) == 4, "caller-side evidence: two-state float leaves, four observations per lane");
    // (32-bit byte offsets: an evidence array of two-state float rows stays below 4 GB up to 2^29 observations)
    const uint32_t last_piece = (uint32_t)(a.N * (int64_t)(S * sizeof(T)) - 16);
    uint32_t at0 = (uint32_t)((t * tile_of<T>::value + col) * (int64_t)(S * sizeof(T))), at1 = at0 + 16;
    at0 = at0 < last_piece ? at0 : last_piece;
    at1 = at1 < last_piece ? at1 : last_piece;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.obs);
    const fv<T, 4> x = *reinterpret_cast<const fv<T, 4>*>(base + at0);
    const fv<T, 4> y = *reinterpret_cast<const fv<T, 4>*>(base + at1);
    st.ev[0][0] = x[0];
    st.ev[1][0] = x[1];
    st.ev[0][1] = x[2];
    st.ev[1][1] = x[3];
    st.ev[0][2] = y[0];
    st.ev[1][2] = y[1];
    st.ev[0][3] = y[2];
    st.ev[1][3] = y[3];
  } else {
#pragma unroll
    for (int32_t j = 0; j < S; ++j) st.ev[j] = loadv<T, V>(p.obs, t * S + j, col);
  }
  if constexpr (STATE == kLeafRecompute) {
    st.z = splat<T, V>((T)1);  // formed in the family loop
  } else if constexpr (STATE != kLeafSteady) {
    fv<T, V> own[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j)
      own[j] = STATE == kLeafFresh ? splat<T, V>(1.0f / (T)S) : loadv<T, V>(p.f2v_own, t * S + j, col);
    st.z = leaf_normaliser<T, S, V>(st.ev, own);
  } else {
    st.z = loadv<T, V>(p.z, t, col);
  }
  return st;
}

// The partial product's exponents, one per (state, observation): int16 pairs when V is 2 (packed
// adds, half the registers; chunks are capped at kLeafMaxChunkLen families so that even messages
// at the bottom of the T range cannot overflow 16 bits), plain int32 otherwise.
constexpr int32_t kLeafMaxChunkLen = 192;
template <int V>
using ev_t = typename std::conditional<V == 2, short, int32_t>::type __attribute__((ext_vector_type(V)));

template <typename T, int D0, int S, int V, int MODE>
__global__ void __launch_bounds__(256) leaf_family_p1(T* __restrict__ arena, const T* __restrict__ cpts,
                                                     LeafFamilyArgs<T> a) {
  constexpr bool FULL = MODE % 3 == kLeafFull;
  constexpr bool REDUCE = MODE % 3 == kLeafReduce;
  static_assert(FULL || REDUCE, "loop launches of one-parent families run on leaf_loop_p1");
  // MODE = mode + 3 * state (+ kLeafCaller when the evidence is read from the caller's arrays)
  constexpr bool CALLER = MODE >= kLeafCaller;
  constexpr int32_t STATE = (MODE % kLeafCaller) / 3;
  static_assert(!CALLER || REDUCE, "only a reduce launch reads the caller's arrays: it stores nothing that would outlive them");
  constexpr int32_t kParts = tile_of<T>::value / (64 * V);
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
  // The parent's scaled product P' (largest state = 1) for this lane's observations, in T.
  // A state whose share is below T range reads as zero: its message to a leaf factor would
  // be below 2^-100 of the leading state's and cannot change any sum it enters.
  fv<T, V> pf[D0];
  {
    const double* scaled = a.scaled + t * tile_of<T>::value + col;
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
      for (int32_t v = 0; v < V; ++v) pf[i][v] = (T)pv[v];
    }
  }

  // kLeafRecompute: the previous iteration's P' for this lane's observations, in LDS
  extern __shared__ __align__(16) unsigned char recompute_raw[];
  fv<T, V>* prev_pf = reinterpret_cast<fv<T, V>*>(recompute_raw) + (size_t)wave * D0 * 64 + lane;  // [D0] stride 64
  fv<T, V> prev_sum = splat<T, V>((T)0);
  if constexpr (STATE == kLeafRecompute) {
    const double* scaled = a.scaled_prev + t * tile_of<T>::value + col;
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      fv<T, V> x;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) x[v] = (T)scaled[(int64_t)i * a.npad + v];
      prev_pf[i * 64] = x;
      prev_sum = i == 0 ? x : prev_sum + x;
    }
  }

  // reduce mode: prod_k z_k per observation (one log at the end gives sum_k log z_k)
  double pz[V];
  int32_t pze[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
    pz[v] = 1.0;
    pze[v] = 0;
  }

  // reduce mode: the count sums of the previous family after their steps inside the quads
  constexpr bool kHoldSums = true;
  WaveSums<T, D0 * S> held;
  T* held_dst = nullptr;

  constexpr bool kCptAhead = REDUCE && D0 * S <= 24;
  T cpt_held[kCptAhead ? D0 * S : 1];
  if constexpr (kCptAhead) {
    const T* c = cpts + a.cpt_off[first];
#pragma unroll
    for (int32_t j = 0; j < D0 * S; ++j) cpt_held[j] = c[j];
  }
  // kDeep: the loop state is requested two families ahead instead of one (where the registers allow:
  // the launch that reads the caller's arrays, which keeps no addresses per family)
  constexpr bool kDeep = CALLER;
  LeafPtrs<T> ptr = leaf_ptrs<T, CALLER>(arena, cpts, a, first);
  LeafState<T, S, V> cur = leaf_state<T, S, V, STATE, CALLER>(ptr, a, t, col);
  LeafPtrs<T> ptr_mid = ptr;
  LeafState<T, S, V> mid = cur;
  if constexpr (kDeep) {
    ptr_mid = leaf_ptrs<T, CALLER>(arena, cpts, a, first + 1 < last ? first + 1 : first);
    mid = leaf_state<T, S, V, STATE, CALLER>(ptr_mid, a, t, col);
  }
  for (int32_t k = first; k < last; ++k) {
    const int32_t ahead = kDeep ? 2 : 1;
    const int32_t kn = k + ahead < last ? k + ahead : last - 1;
    const LeafPtrs<T> ptr_far = leaf_ptrs<T, CALLER>(arena, cpts, a, kn);
    const LeafState<T, S, V> far = leaf_state<T, S, V, STATE, CALLER>(ptr_far, a, t, col);
    const LeafPtrs<T> ptr_next = kDeep ? ptr_mid : ptr_far;
    const LeafState<T, S, V> nxt = kDeep ? mid : far;
    if constexpr (kDeep) {
      ptr_mid = ptr_far;
      mid = far;
    }
    // CPT rows: small ones travel one family ahead through the scalar cache (requested here for the
    // next family, the current family's were requested a turn ago), so that no family starts by
    // waiting for its scalar loads
    T cpt_next[kCptAhead ? D0 * S : 1];
    if constexpr (kCptAhead) {
#pragma unroll
      for (int32_t j = 0; j < D0 * S; ++j) cpt_next[j] = ptr_next.cpt[j];
    }
    const T* cpt = kCptAhead ? cpt_held : ptr.cpt;

    // q = cpt . ev.  The leaf's message to the factor is ev / z, so the factor's message to the
    // parent is q / z; and because the parent's message back is normalised over its states, q
    // also stands in for the factor's previous message to the parent (q / z_prev) in the quotient.
    fv<T, V> q[D0];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) {
      q[i] = cpt[i * S] * cur.ev[0];
#pragma unroll
      for (int32_t j = 1; j < S; ++j) q[i] = __builtin_elementwise_fma(splat<T, V>(cpt[i * S + j]), cur.ev[j], q[i]);
    }
    // parent's message to this factor, before normalisation: r = P' / (previous message to the parent)
    fv<T, V> r[D0];
    fv<T, V> inv_sum = splat<T, V>(0.f);
    if constexpr (STATE == kLeafRecompute) {
      // the same operations, in the same order, as the loop launch's z' = sum(P') / sum(P' / q)
      fv<T, V> sum_r = splat<T, V>(0.f);
      // read again for every family (held in registers over the loop, P'prev would cost a wave per SIMD)
      uint32_t prev_at = (uint32_t)(((size_t)wave * D0 * 64 + lane) * sizeof(fv<T, V>));
      asm volatile("" : "+v"(prev_at));
      const fv<T, V>* prev_here = reinterpret_cast<const fv<T, V>*>(recompute_raw + prev_at);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        const fv<T, V> rq = rcpv<T, V>(q[i]);
        sum_r = __builtin_elementwise_fma(prev_here[i * 64], rq, sum_r);
        r[i] = pf[i] * rq;
        inv_sum += r[i];
      }
      cur.z = prev_sum * rcpv<T, V>(sum_r);
    } else {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        fv<T, V> prev;
        if constexpr (STATE == kLeafFresh) {
          prev = splat<T, V>(1.0f / (T)D0);
        } else if constexpr (STATE == kLeafFromArrays) {
          prev = loadv<T, V>(ptr.f2v_par, t * D0 + i, col);
        } else {
          prev = q[i];
        }
        r[i] = pf[i] * rcpv<T, V>(prev);
        inv_sum += r[i];
      }
    }
    inv_sum = rcpv<T, V>(inv_sum);
    if constexpr (REDUCE) {
      // sum_k log z_k = log prod_k z_k: the product is carried like in loop mode, one log at the end
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pz[v] *= (double)cur.z[v];
      if (renorm_due<T>(k - first + 1)) {
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
        const fv<T, V> scale = recipv<T, V>(cur.z) * inv_sum;
        fv<T, V> o[S];
#pragma unroll
        for (int32_t j = 0; j < S; ++j) {
          o[j] = cur.ev[j] * scale;
#pragma unroll
          for (int32_t v = 0; v < V; ++v)
            if (t * tile_of<T>::value + col + v >= a.N) o[j][v] = 0.f;
        }
        T* dst = a.em_partial + ((int64_t)k * (D0 * S)) * a.slots + (t * kParts + part);
        T per_lane[D0 * S];
#pragma unroll
        for (int32_t i = 0; i < D0; ++i)
#pragma unroll
          for (int32_t j = 0; j < S; ++j) {
            // a chain of fused multiply-adds: half the issue slots of packed products plus adds
            T x = r[i][0] * o[j][0];
#pragma unroll
            for (int32_t v = 1; v < V; ++v) x = fma_t(r[i][v], o[j][v], x);
            per_lane[i * S + j] = x;
          }
        // the sums over the wave of the family before this one are finished here, next to this
        // family's arithmetic, and this family's wait for the next turn
        WaveSums<T, D0 * S> sums;
        sums.scatter(per_lane, lane);
        if constexpr (!kHoldSums) {
          sums.finish();
          if (lane < 4) {
#pragma unroll
            for (int32_t f = 0; f < WaveSums<T, D0 * S>::H2; ++f) {
              const int32_t e = WaveSums<T, D0 * S>::entry(lane, f);
              if (e < D0 * S) dst[(int64_t)e * a.slots] = sums.out[f];
            }
          }
        } else if (held_dst) {
          held.finish();
          if (lane < 4) {
#pragma unroll
            for (int32_t f = 0; f < WaveSums<T, D0 * S>::H2; ++f) {
              const int32_t e = WaveSums<T, D0 * S>::entry(lane, f);
              if (e < D0 * S) held_dst[(int64_t)e * a.slots] = held.out[f];
            }
          }
        }
        if constexpr (kHoldSums) {
          held = sums;
          held_dst = dst;
        }
      }
      if constexpr (kCptAhead) {
#pragma unroll
        for (int32_t j = 0; j < D0 * S; ++j) cpt_held[j] = cpt_next[j];
      }
      ptr = ptr_next;
      cur = nxt;
      continue;
    }
    // own (leaf) variable: c = log z, message to the factor = ev / z
    const fv<T, V> inv_z = recipv<T, V>(cur.z);
    {
      fv<T, V> c;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        const T mz = frexp(cur.z[v], &de);
        c[v] = (T)((double)de * 0.6931471805599453 + (double)log(mz));
      }
      storev<T, V>(ptr.c, t, col, c);
    }
    fv<T, V> new_par[D0], new_own[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) new_par[i] = q[i] * inv_z;
    if constexpr (FULL || REDUCE) {
      fv<T, V> to_f_own[S];
#pragma unroll
      for (int32_t j = 0; j < S; ++j) to_f_own[j] = cur.ev[j] * inv_z;
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) r[i] *= inv_sum;  // now the normalised message
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) storev<T, V>(ptr.v2f_par, t * D0 + i, col, r[i]);
#pragma unroll
      for (int32_t j = 0; j < S; ++j) storev<T, V>(ptr.v2f_own, t * S + j, col, to_f_own[j]);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) storev<T, V>(ptr.f2v_par, t * D0 + i, col, new_par[i]);
    }
    // factor's message to the leaf: cpt^T . (normalised message from the parent)
#pragma unroll
    for (int32_t j = 0; j < S; ++j) new_own[j] = cpt[j] * r[0];
#pragma unroll
    for (int32_t i = 1; i < D0; ++i)
#pragma unroll
      for (int32_t j = 0; j < S; ++j) new_own[j] = __builtin_elementwise_fma(splat<T, V>(cpt[i * S + j]), r[i], new_own[j]);
#pragma unroll
    for (int32_t j = 0; j < S; ++j) storev<T, V>(ptr.f2v_own, t * S + j, col, new_own[j]);
    // fold into the partial product for the parent's next combine
#pragma unroll
    for (int32_t i = 0; i < D0; ++i)
#pragma unroll
      for (int32_t v = 0; v < V; ++v) pm[i][v] *= (double)new_par[i][v];
    if (renorm_due<T>(k - first + 1)) {
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
    if (held_dst) {
      held.finish();
      if (lane < 4) {
#pragma unroll
        for (int32_t f = 0; f < WaveSums<T, D0 * S>::H2; ++f) {
          const int32_t e = WaveSums<T, D0 * S>::entry(lane, f);
          if (e < D0 * S) held_dst[(int64_t)e * a.slots] = held.out[f];
        }
      }
    }
    double* dst = a.ll_part + (int64_t)chunk * a.npad + t * tile_of<T>::value + col;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      const double m = frexp(pz[v], &de);
      dst[v] = (double)(pze[v] + de) * 0.6931471805599453 + log(m);
    }
    return;
  }
  // the partial product holds the whole messages (q / z): nothing is left for the combine to divide by
#pragma unroll
  for (int32_t v = 0; v < V; ++v) {
    const int64_t at = (int64_t)(a.chunk_base + chunk) * a.npad + t * tile_of<T>::value + col + v;
    a.zdiv_mant[at] = 0.5;
    a.zdiv_exp[at] = 1;
    a.zdiv_next_mant[at] = 0.5;
    a.zdiv_next_exp[at] = 1;
  }
#pragma unroll
  for (int32_t s = 0; s < D0; ++s) {
    int32_t ex[V];
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      int de = 0;
      pm[s][v] = frexp(pm[s][v], &de);
      ex[v] = (int32_t)pe[s][v] + de;
    }
    const int64_t row = ((int64_t)(a.chunk_base + chunk) * D0 + s) * a.npad + t * tile_of<T>::value + col;
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
// Chunk partial products of a wide node from the stored messages, for either element type (the
// float32-only wide_product_fast moves 16 bytes per lane; this one serves float64 plans, where the
// partials are otherwise always maintained by the leaf-family kernels and only have to be rebuilt
// when an iteration starts from message arrays that no such kernel wrote).
//   grid.x = ceil(chunks * ntiles * parts / 4), block = 256, two observations per lane
template <typename T, int S>
__global__ void __launch_bounds__(256) wide_product_any(const T* __restrict__ arena, WideArgs a) {
  constexpr int32_t V = 2, kParts = tile_of<T>::value / (64 * V);
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (w >= (int64_t)a.chunks * a.ntiles * kParts) return;
  const int32_t part = (int32_t)(w % kParts);
  w /= kParts;
  const int32_t chunk = (int32_t)(w / a.ntiles);
  const int64_t t = w % a.ntiles;
  const int32_t col = part * 64 * V + lane * V;
  double m[S][V];
  int32_t e[S][V];
#pragma unroll
  for (int32_t s = 0; s < S; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      m[s][v] = 1.0;
      e[s][v] = 0;
    }
  const int32_t first = a.chunk_first[chunk], last = a.chunk_first[chunk + 1];
  for (int32_t j = first; j < last; ++j) {
    const T* src = arena + a.in_off[a.covered[j]];
#pragma unroll
    for (int32_t s = 0; s < S; ++s) {
      const fv<T, V> x = a.fresh ? splat<T, V>((T)1 / (T)S) : loadv<T, V>(src, t * S + s, col);
#pragma unroll
      for (int32_t v = 0; v < V; ++v) m[s][v] *= (double)x[v];
    }
    if (renorm_due<T>(j - first + 1) || j + 1 == last) {
#pragma unroll
      for (int32_t s = 0; s < S; ++s)
#pragma unroll
        for (int32_t v = 0; v < V; ++v) {
          int de = 0;
          m[s][v] = frexp(m[s][v], &de);
          e[s][v] += de;
        }
    }
  }
#pragma unroll
  for (int32_t s = 0; s < S; ++s)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
      const int64_t row = ((int64_t)chunk * S + s) * a.npad + t * tile_of<T>::value + col + v;
      a.part_mant[row] = m[s][v];
      a.part_exp[row] = e[s][v];
    }
}

// ---------------------------------------------------------------------------------
// Leaf families with TWO wide parents (CPT [D0][D1][S], e.g. the loopy 784-leaf model's
// Y | F1, F2).  Same structure as leaf_family_p1; differences:
//   * the factor's previous messages to the parents depend on the other parent's message, so they
//     cannot be rebuilt from the evidence: the loop state is m1[D0], m2[D1] and the leaf's next z;
//   * the kernel maintains the chunk partial products of both parents.
// Per family and observation the loop moves D0+D1-1+S floats in and D0+D1-1 out (5 + 3 for the
// (2,2,2) families: the loop state holds ratios of the previous messages) instead of the 36 of the
// unfused passes.
template <typename T, int N, int V>
__device__ __forceinline__ void load_scaled(const double* __restrict__ scaled, int64_t npad, fv<T, V>* out) {
#pragma unroll
  for (int32_t i = 0; i < N; ++i)
#pragma unroll
    for (int32_t v = 0; v < V; ++v) out[i][v] = (T)scaled[(int64_t)i * npad + v];
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

template <typename T, int D0, int D1, int S, int V, int MODE>
__global__ void __launch_bounds__(256) leaf_family_p2(T* __restrict__ arena, const T* __restrict__ cpts,
                                                     LeafFamilyArgs<T> a) {
  constexpr bool FULL = MODE % 3 == kLeafFull;
  constexpr bool REDUCE = MODE % 3 == kLeafReduce;
  constexpr int32_t STATE = MODE / 3;
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
  fv<T, V> pf1[D0], pf2[D1];
  load_scaled<T, D0, V>(a.scaled + t * tile_of<T>::value + col, a.npad, pf1);
  load_scaled<T, D1, V>(a.scaled2 + t * tile_of<T>::value + col, a.npad, pf2);

  double ll_acc[V];
#pragma unroll
  for (int32_t v = 0; v < V; ++v) ll_acc[v] = 0.0;

  for (int32_t k = first; k < last; ++k) {
    const T* cpt = cpts + a.cpt_off[k];
    T* f2v_p1 = arena + a.f_out[3 * k];
    T* f2v_p2 = arena + a.f_out[3 * k + 1];
    T* f2v_own = arena + a.f_out[3 * k + 2];
    T* m_state = a.m_state + (int64_t)k * (D0 + D1) * a.npad;
    T* z_state = a.z_state + (int64_t)k * a.npad;
    const T* obs = arena + a.obs_off[k];

    fv<T, V> ev[S], prev1[D0], prev2[D1], z;
#pragma unroll
    for (int32_t j = 0; j < S; ++j) ev[j] = loadv<T, V>(obs, t * S + j, col);
    if constexpr (STATE == kLeafSteady) {
      // the parents' messages back are normalised over the states, so of the factor's previous
      // messages only ratios are kept: (last state's value) / (state i's value), which the quotient
      // P' / previous message needs as a factor (rows i and D0 + i of the state array; the rows of
      // the two last states are not used)
#pragma unroll
      for (int32_t i = 0; i < D0 - 1; ++i) prev1[i] = loadv<T, V>(m_state, t * (D0 + D1) + i, col);
      prev1[D0 - 1] = splat<T, V>((T)1);
#pragma unroll
      for (int32_t i = 0; i < D1 - 1; ++i) prev2[i] = loadv<T, V>(m_state, t * (D0 + D1) + D0 + i, col);
      prev2[D1 - 1] = splat<T, V>((T)1);
      z = loadv<T, V>(z_state, t, col);
    } else {
      fv<T, V> own[S];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i)
        prev1[i] = STATE == kLeafFresh ? splat<T, V>(1.0f / (T)D0) : loadv<T, V>(f2v_p1, t * D0 + i, col);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i)
        prev2[i] = STATE == kLeafFresh ? splat<T, V>(1.0f / (T)D1) : loadv<T, V>(f2v_p2, t * D1 + i, col);
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
        own[j] = STATE == kLeafFresh ? splat<T, V>(1.0f / (T)S) : loadv<T, V>(f2v_own, t * S + j, col);
      z = leaf_normaliser<T, S, V>(ev, own);
    }

    // both parents' messages to this factor: (P'/previous message) normalised over the states
    fv<T, V> t1[D0], t2[D1];
    {
      fv<T, V> sum = splat<T, V>(0.f);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        t1[i] = STATE == kLeafSteady ? pf1[i] * prev1[i] : pf1[i] * rcpv<T, V>(prev1[i]);
        sum += t1[i];
      }
      sum = rcpv<T, V>(sum);
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) t1[i] *= sum;
      sum = splat<T, V>(0.f);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) {
        t2[i] = STATE == kLeafSteady ? pf2[i] * prev2[i] : pf2[i] * rcpv<T, V>(prev2[i]);
        sum += t2[i];
      }
      sum = rcpv<T, V>(sum);
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) t2[i] *= sum;
    }
    // the leaf's variable pass: c = log z, message to the factor = ev / z
    const fv<T, V> inv_z = recipv<T, V>(z);
    fv<T, V> o[S];
#pragma unroll
    for (int32_t j = 0; j < S; ++j) o[j] = ev[j] * inv_z;
    if (FULL || REDUCE) {
      fv<T, V> c;
#pragma unroll
      for (int32_t v = 0; v < V; ++v) {
        int de = 0;
        const T mz = frexp(z[v], &de);
        c[v] = (T)((double)de * 0.6931471805599453 + (double)log(mz));
        if (REDUCE) ll_acc[v] += (double)c[v];
      }
      if (FULL) storev<T, V>(arena + a.c_off[k], t, col, c);
    }
    if constexpr (REDUCE) {
      if (!a.em_partial) continue;  // only the log-likelihood sums are wanted
      // EM counts over the wave's observations (padding masked), wave-reduced
#pragma unroll
      for (int32_t j = 0; j < S; ++j)
#pragma unroll
        for (int32_t v = 0; v < V; ++v)
          if (t * tile_of<T>::value + col + v >= a.N) o[j][v] = 0.f;
      T* dst = a.em_partial + ((int64_t)k * (D0 * D1 * S)) * a.slots + (t * kParts + part);
      T per_lane[D0 * D1 * S];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i)
#pragma unroll
        for (int32_t l = 0; l < D1; ++l) {
          const fv<T, V> tt = t1[i] * t2[l];
#pragma unroll
          for (int32_t j = 0; j < S; ++j) {
            const fv<T, V> prod = tt * o[j];
            T x = prod[0];
#pragma unroll
            for (int32_t v = 1; v < V; ++v) x += prod[v];
            per_lane[(i * D1 + l) * S + j] = x;
          }
        }
      WaveSums<T, D0 * D1 * S> sums;
      sums.reduce(per_lane, lane);
      if (lane < 4) {
#pragma unroll
        for (int32_t f = 0; f < WaveSums<T, D0 * D1 * S>::H2; ++f) {
          const int32_t e = WaveSums<T, D0 * D1 * S>::entry(lane, f);
          if (e < D0 * D1 * S) dst[(int64_t)e * a.slots] = sums.out[f];
        }
      }
      continue;
    }
    // contractions: A[i][l] = sum_j cpt[i][l][j] o[j];  m1 = A t2;  m2 = A^T t1;  own = cpt^T (t1 x t2)
    fv<T, V> m1[D0], m2[D1], own[S];
#pragma unroll
    for (int32_t i = 0; i < D0; ++i) m1[i] = splat<T, V>(0.f);
#pragma unroll
    for (int32_t l = 0; l < D1; ++l) m2[l] = splat<T, V>(0.f);
#pragma unroll
    for (int32_t j = 0; j < S; ++j) own[j] = splat<T, V>(0.f);
#pragma unroll
    for (int32_t i = 0; i < D0; ++i)
#pragma unroll
      for (int32_t l = 0; l < D1; ++l) {
        const T* row = cpt + (i * D1 + l) * S;
        fv<T, V> A = row[0] * o[0];
#pragma unroll
        for (int32_t j = 1; j < S; ++j) A = __builtin_elementwise_fma(splat<T, V>(row[j]), o[j], A);
        m1[i] = __builtin_elementwise_fma(A, t2[l], m1[i]);
        m2[l] = __builtin_elementwise_fma(A, t1[i], m2[l]);
        const fv<T, V> tt = t1[i] * t2[l];
#pragma unroll
        for (int32_t j = 0; j < S; ++j) own[j] = __builtin_elementwise_fma(splat<T, V>(row[j]), tt, own[j]);
      }
    if (FULL) {
      T* v2f_p1 = arena + a.f_in[3 * k];
      T* v2f_p2 = arena + a.f_in[3 * k + 1];
      T* v2f_own = arena + a.f_in[3 * k + 2];
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) {
        storev<T, V>(v2f_p1, t * D0 + i, col, t1[i]);
        storev<T, V>(f2v_p1, t * D0 + i, col, m1[i]);
      }
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) {
        storev<T, V>(v2f_p2, t * D1 + i, col, t2[i]);
        storev<T, V>(f2v_p2, t * D1 + i, col, m2[i]);
      }
#pragma unroll
      for (int32_t j = 0; j < S; ++j) {
        storev<T, V>(v2f_own, t * S + j, col, o[j]);
        storev<T, V>(f2v_own, t * S + j, col, own[j]);
      }
    } else {
      // (reciprocals to within an ulp: the next iteration multiplies by the ratio, nothing else is rounded)
#pragma unroll
      for (int32_t i = 0; i < D0 - 1; ++i)
        storev<T, V>(m_state, t * (D0 + D1) + i, col, m1[D0 - 1] * recipv<T, V>(m1[i]));
#pragma unroll
      for (int32_t i = 0; i < D1 - 1; ++i)
        storev<T, V>(m_state, t * (D0 + D1) + D0 + i, col, m2[D1 - 1] * recipv<T, V>(m2[i]));
      storev<T, V>(z_state, t, col, leaf_normaliser<T, S, V>(ev, own));
    }
    // fold into both parents' partial products
#pragma unroll
    for (int32_t v = 0; v < V; ++v) {
#pragma unroll
      for (int32_t i = 0; i < D0; ++i) pm1[i][v] *= (double)m1[i][v];
#pragma unroll
      for (int32_t i = 0; i < D1; ++i) pm2[i][v] *= (double)m2[i][v];
    }
    if (renorm_due<T>(k - first + 1)) {
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
    double* dst = a.ll_part + (int64_t)chunk * a.npad + t * tile_of<T>::value + col;
#pragma unroll
    for (int32_t v = 0; v < V; ++v) dst[v] = ll_acc[v];
    return;
  }
  store_partials<D0, V>(a.part_mant, a.part_exp, (int64_t)(a.chunk_base + chunk) * D0 * a.npad + t * tile_of<T>::value + col, a.npad, pm1, pe1);
  store_partials<D1, V>(a.part_mant2, a.part_exp2, (int64_t)(a.chunk_base + chunk) * D1 * a.npad + t * tile_of<T>::value + col, a.npad, pm2, pe2);
}

// accum[entry] = cpt[entry] * sum over the per-wave sums written by the kLeafReduce kernel.
// One wave per CPT entry.   grid.x = ceil(entries / 4) (+ ceil(listed.count / 4)), block = 256
// The blocks past the group's own also finalise a list of other entries from their chunk partial
// sums (em_finalize_listed's work: the entries no fused group delivers), so that an EM step of a
// network with a few unfused nodes needs one finalising launch, not two.
struct ListedEntries {
  const double* partial;   // [entry][nchunks]
  const int32_t* entries;  // [count] flat CPT positions
  int32_t count;           // 0 = none
  int32_t nchunks;
};

// Count sums a reduce launch over compact hard evidence left by evidence form instead of by CPT entry
// (kernels_hard.h: hard_reduce, hard_count): the group's records and what the state bytes stand for.
// tab == nullptr: sums by entry.
struct FormSums {
  const float* tab;
  HardEvidenceRows rows;  // what the evidence forms stand for
};

constexpr int32_t kHardForms = 3;  // evidence forms of a two-state leaf: state 0, state 1, neither
// One record per family, the same for all lanes and read through the scalar cache: rows of D0 floats
//   0..2  q[y][d] = cpt[d] . ev(y) for the three evidence forms
//   3..5  1 / q[y][d]
//   6..7  how an observation of the third form enters the sums of the first two (hard_reduce)
//   8..13 three rows of doubles: log2 q[0][d], log2 q[1][d] - log2 q[0][d], log2 q[2][d] - log2 q[0][d]
//         (a q of zero counts as 2^kHardLogOfZero)
constexpr int32_t kHardRows = 14, kHardQ = 0, kHardRq = 3, kHardThird = 6, kHardLog = 8;
constexpr double kHardLogOfZero = -1.0e6;

// evidence row j of a two-state leaf whose evidence has form y (0, 1; 2: neither state)
__device__ __forceinline__ float hard_ev(const HardEvidenceRows& rows, int32_t y, int32_t j) { return rows.e[2 * y + j]; }

// counts[d][j] of a family from its sums by form, G[y][d] = sum over the observations of form y of
// P'[d] * scale:  cpt[d][j] * (ev(0)[j] * G[0][d] / q[0][d] + ev(1)[j] * G[1][d] / q[1][d])
// the sums of family k's third-form observations for form y, state d (behind the K records)
__device__ __forceinline__ double hard_third_sum(const float* tab, int32_t K, int32_t d0, int32_t k, int32_t y, int32_t d) {
  return (double)tab[(size_t)K * (kHardRows * d0) + (size_t)k * (2 * d0) + (size_t)y * d0 + d];
}

__device__ __forceinline__ double hard_count(double g0, double g1, float rq0, float rq1, int32_t j, const HardEvidenceRows& rows,
                                             double cpt) {
#pragma clang fp contract(off)
  const double x0 = ((double)rq0 * (double)hard_ev(rows, 0, j)) * g0;
  const double x1 = ((double)rq1 * (double)hard_ev(rows, 1, j)) * g1;
  return (x0 + x1) * cpt;
}



template <typename T>
__global__ void __launch_bounds__(256) em_finalize_slots(const T* __restrict__ em_partial,
                                                        const T* __restrict__ cpts,
                                                        const int64_t* __restrict__ cpt_off, int32_t K,
                                                        int32_t entries_per_node, int32_t slots,
                                                        double* __restrict__ accum, ListedEntries listed, FormSums forms) {
  const int32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t own = (((int64_t)K * entries_per_node + kWavesPerBlock - 1) / kWavesPerBlock) * kWavesPerBlock;
  if (item >= own) {
    const int64_t i = item - own;
    if (i >= listed.count) return;
    const int64_t e = listed.entries[i];
    // fixed association: lane l sums chunks l, l + 64, ..; then the butterfly
    double acc = 0.0;
    for (int32_t c = lane; c < listed.nchunks; c += 64) acc += listed.partial[e * listed.nchunks + c];
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) accum[e] = acc * (double)cpts[e];
    return;
  }
  if (item >= (int64_t)K * entries_per_node) return;
  const int32_t k = (int32_t)(item / entries_per_node);
  const int32_t e = (int32_t)(item % entries_per_node);
  auto slot_sum = [&](int64_t entry) {
    const T* src = em_partial + entry * slots;
    double acc = 0.0;
    for (int32_t sl = lane; sl < slots; sl += 64) acc += (double)src[sl];
#pragma unroll
    for (int32_t off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
  };
  if (forms.tab) {
    // two-state leaves: entry e = (d, j); the sums of forms 0 and 1 lie where entries (d, 0), (d, 1) would
    const int32_t d = e >> 1, j = e & 1, d0 = entries_per_node >> 1;
    const double g0 = slot_sum(item - j) + hard_third_sum(forms.tab, K, d0, k, 0, d);
    const double g1 = slot_sum(item - j + 1) + hard_third_sum(forms.tab, K, d0, k, 1, d);
    const float* rec = forms.tab + (size_t)k * (kHardRows * d0);
    if (lane == 0)
      accum[cpt_off[k] + e] = hard_count(g0, g1, rec[kHardRq * d0 + d], rec[(kHardRq + 1) * d0 + d], j, forms.rows,
                                         (double)cpts[cpt_off[k] + e]);
    return;
  }
  const double acc = slot_sum(item);
  if (lane == 0) accum[cpt_off[k] + e] = acc * (double)cpts[cpt_off[k] + e];
}


}  // namespace bn
