// State the specialised path keeps per plan (which groups run on the wide-node / leaf-family
// kernels, their scratch and loop state); shared by the translation units that launch them.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>

#include <hip/hip_runtime.h>

#include "plan.h"

namespace bn {

// observations per lane in the leaf-family kernel: 4 leaves one wave per SIMD, 1 doubles the
// per-family instruction overhead (both measured slower on the 784-leaf model)
constexpr int32_t kLeafVec = 2;
// waves a leaf-family launch needs per tile (a wave covers 64 * kLeafVec observations)
inline int32_t leaf_parts(const Plan& p) { return p.tile / (64 * kLeafVec); }

struct WideNode {
  int32_t node = -1, vgroup = -1, vpos = -1;
  int32_t S = 0, I = 0;
  int32_t fused_fgroup = -1;  // first factor group whose leaf-family kernel maintains partials of this node, or -1
  std::vector<int32_t> fused_fgroups;  // all of them, in the order their families appear in `covered`
  int32_t chunks = 0, chunk_len = 0;
  std::vector<int32_t> covered, uncovered;
  int32_t* d_covered = nullptr;
  int32_t* d_uncovered = nullptr;
  int32_t* d_chunk_first = nullptr;  // [chunks + 1] chunk boundaries in `covered`
  double* part_mant = nullptr;
  int32_t* part_exp = nullptr;
  // P' of the last combine and of the one before (the combine writes over the older one and the two
  // swap; a reduce launch that recomputes the leaves' normalisers reads both)
  double* scaled = nullptr;
  double* scaled_prev = nullptr;
  // fused nodes: per chunk and observation, what the combine divides the chunk's partial product by.
  // Two arrays that alternate: the next combine reads zdiv_*[zd_read], the one after it the other
  // (see kernels_leaf_loop.h); every combine on stored state flips zd_read.
  double* zdiv_mant[2] = {nullptr, nullptr};  // [chunks][npad] each, one allocation
  int32_t* zdiv_exp[2] = {nullptr, nullptr};
  int32_t zd_read = 0;
  // the product over all chunk partials, per state (mantissa, exponent), kept from the first combine
  // after they were written for the combines that follow while they stay the same (loop launches
  // that only move the divisors); only for nodes whose fused groups are all one-parent groups
  double* prod_mant = nullptr;  // [S][npad]
  int32_t* prod_exp = nullptr;  // [S][npad]
  bool products_cached = false;
  // the last combine also did the next iteration's (WideArgs::early_*): that iteration's variable pass
  // has nothing left to do for this node
  bool combined_ahead = false;
  bool partials_valid = false;
  // some covered / uncovered slot's message arrays lie in the plan's lazy region
  bool covered_lazy = false, uncovered_lazy = false;
};

struct FusedFamily {
  int32_t wide_index = -1;
  int32_t wide_index2 = -1;     // second wide parent of two-parent families, else -1
  int32_t chunk_base = 0, chunks = 0, chunk_len = 0;  // this group's range of the parent's chunk partials
  // loop launches of a one-parent group: waves that share a chunk (1, or kWavesPerBlock for small batches)
  int32_t split = 1;
  void* m_state = nullptr;      // [K][D0 + D1][npad] of the plan's element type: two-parent loop state (previous messages to the parents)
  int64_t* d_obs_off = nullptr;
  int64_t* d_c_off = nullptr;
  std::vector<int32_t> ev_index;   // [K] position of each leaf in the evidence order, -1 = not observed
  const void** d_src = nullptr;    // [K] evidence source of each leaf for an ingesting launch
  std::vector<const void*> src_on_device;  // what d_src holds
  // the sources lie at equal distances: leaf k's is src_base + k * src_stride (0 = d_src is used)
  const unsigned char* src_base = nullptr;
  int64_t src_stride = 0;
  void* z_state = nullptr;      // [K][npad] loop-state normalisers (plan's element type)
  // where the leaves' normalisers of the coming iteration are: kZPerFamily (z_state, written by a
  // loop launch that a full launch follows), kZPerChunk (only their chunk products, in the parent's
  // divisor arrays: what loop launches need), kZRecompute (as kZPerChunk, and the loop launch of the
  // last but one iteration was left out: the reduce launch forms them again from the parent's two
  // scaled products), kZNone after a reset
  int32_t z_form = 0;
  void* em_partial = nullptr;   // [K][entries][slots] (plan's element type), written by the reduce-mode kernel
  int32_t slots = 0;
  double* ll_part = nullptr;    // [chunks][npad], written by the reduce-mode kernel
  int32_t ll_rows = 0;          // rows the last reduce-mode launch wrote
  int32_t ll_rows_cap = 0;      // rows allocated
  int32_t slots_used = 0;       // per-wave sum slots the last reduce-mode launch wrote
  bool em_by_form = false;      // ... as sums by evidence form (kernels_hard.h), not by CPT entry
  // State bytes read in place (kernels_hard.h): the products of the leaves' normalisers the last
  // hard_loop_light launch left, per chunk of the parent - the reduce launch turns them into its rows of
  // normalisers (null: no such launch since the messages were fresh) - and whether the call's first launch
  // has run without one yet (then the loop launch before the reduce launch is not left out)
  double* hard_z_mant = nullptr;
  int32_t* hard_z_exp = nullptr;
  bool hard_wants_light = false;
  // the last iteration ran in reduce mode: stored messages are one iteration behind until
  // complete_pending_fast re-runs it in full mode with these flags
  bool pending = false;
  int32_t pending_fresh = 0, pending_first = 0;
  bool em_ready = false;  // em_partial holds the counts of the current state
};

constexpr int32_t kZNone = 0, kZPerFamily = 1, kZPerChunk = 2, kZRecompute = 3;

// Device tables of one dense factor group for the fused tree iteration (kernels_dense.h,
// factor_tree_mfma), one set per generation: generation g reads the factor -> variable messages
// from the f2v arrays (g = 0) or from the v2f arrays (g = 1) of the edges and writes the other;
// set 2 is the first iteration on fresh messages (every message still uniform: only the
// observation arrays are inputs) and writes the f2v arrays.
constexpr int32_t kTreeFresh = 2;
struct TreeGroupTables {
  int64_t* in_par[3] = {nullptr, nullptr, nullptr};
  int64_t* in_own[3] = {nullptr, nullptr, nullptr};
  int64_t* own_prev[3] = {nullptr, nullptr, nullptr};
  int64_t* out[3] = {nullptr, nullptr, nullptr};
  double bytes_per_observation = 0;        // what one launch moves per observation (profiling)
  double exact_bytes_per_observation = 0;  // more for the launch that reads the own_prev arrays
  double fresh_bytes_per_observation = 0;  // the first iteration on fresh messages
};

struct FastState {
  // fused tree iterations: every non-root family is a dense 32 / 64-state one-parent family whose
  // two variables each receive at most two other messages
  bool tree_ok = false;
  std::map<int32_t, TreeGroupTables> tree;  // by factor group
  // normaliser arrays of the nodes no fused family serves (the rest of a log-likelihood sum
  // comes from the families' ll_part rows when their last iteration ran in reduce mode)
  int64_t* d_ll_offs = nullptr;
  int64_t n_ll_offs = 0;
  // CPT entries of the nodes no fused family serves (the only ones the generic EM sums have to
  // finalise when every fused group delivers its own counts)
  int32_t* d_unfused_entries = nullptr;
  int32_t n_unfused_entries = 0;
  std::vector<WideNode> wide;
  std::map<int32_t, int32_t> wide_of_node;   // node id -> index in `wide`
  std::map<int32_t, FusedFamily> fused;       // factor group index -> fused family
  std::vector<char> vgroup_skip;              // variable groups whose every node a fused kernel serves
  std::vector<char> vgroup_wide;              // variable groups served by the wide-node path
  // groups with a message array in the plan's lazy region (Plan::in_lazy)
  std::vector<char> vgroup_lazy, fgroup_lazy;
  // parentless factor groups whose output (the prior, broadcast over observations) is already in
  // place: it only changes with the CPTs
  std::vector<char> root_written;
  // Every observed node is a two-state leaf of a one-parent fused family: the first iteration
  // after a reset can read the evidence from the caller's arrays and write the arena copy itself.
  bool ingest_ok = false;
  bool ingest_next = false;  // the coming first iteration does so (evidence pointers are on the device)
  bool ingest_states = false;  // ... from uint8 states (bn_enter_evidence_states' format) instead of float arrays
  bool ingest_all_observed = false;  // every leaf of every ingesting group is observed (the uint8 route needs it)
  double ingest_on = 1.0, ingest_off = 0.0;
  // An EM step (bn_em_step) on such a plan: no launch of the call copies the evidence, they all read
  // the caller's arrays (the plan is reset when the step ends, the copy would never be read again).
  bool evidence_in_caller = false;
  // ... and the same for compact hard evidence: every launch reads the caller's state bytes (kernels_hard.h)
  bool states_in_caller = false;
  // what the state bytes stand for, for the whole step: row entry j of form y at [2 y + j] (kernels_leaf.h, HardEvidenceRows)
  float states_rows[6] = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f};
};

// The plan owns its FastState through Plan::fast (opaque there, so that plan.h needs no kernel types).
inline FastState* fast_state(const Plan& p) { return static_cast<FastState*>(p.fast); }

// Tuning overrides read once per process (benchmark experiments only):
//   BN_WIDE_CHUNKS=n    chunks per wide node        BN_REDUCE_CHUNKS=n  chunks of a reduce-mode launch
//   BN_REDUCE_V2=1      reduce-mode launches with two observations per lane, like the other modes
inline int32_t env_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
inline int32_t env_wide_chunks() {
  static const int32_t v = env_int("BN_WIDE_CHUNKS");
  return v;
}
inline int32_t env_reduce_chunks() {
  static const int32_t v = env_int("BN_REDUCE_CHUNKS");
  return v;
}
// BN_SHARE_MAX=n  largest number of waves per chunk (tiles x parts) at which the waves of a block share a chunk
// BN_COMBINE_STATES_MAX=n  largest number of tiles at which the per-state combine kernel runs (default: always;
//                          measured faster at every batch size of the 784-leaf model)
inline int32_t env_share_max() {
  static const int32_t v = getenv("BN_SHARE_MAX") ? env_int("BN_SHARE_MAX") : 16;
  return v;
}
inline int32_t env_combine_states_max() {
  static const int32_t v = getenv("BN_COMBINE_STATES_MAX") ? env_int("BN_COMBINE_STATES_MAX") : (1 << 30);
  return v;
}
// BN_NO_LIGHT_SPLIT=1  loop launches without the per-state product run one wave per chunk like the others
inline bool env_no_light_split() {
  static const bool v = getenv("BN_NO_LIGHT_SPLIT") != nullptr;
  return v;
}
// BN_NO_RECOMPUTE=1  the loop launch before a reduce launch always runs (and stores the leaves' normalisers)
inline bool env_no_recompute() {
  static const bool v = getenv("BN_NO_RECOMPUTE") != nullptr;
  return v;
}
// BN_COPY_EVIDENCE=1  an EM step copies the evidence into the plan like bn_em_statistics_from does
inline bool env_copy_evidence() {
  static const bool v = getenv("BN_COPY_EVIDENCE") != nullptr;
  return v;
}
// BN_NO_DOUBLE_COMBINE=1  every iteration's combine is a launch of its own
inline bool env_no_double_combine() {
  static const bool v = getenv("BN_NO_DOUBLE_COMBINE") != nullptr;
  return v;
}
inline bool env_reduce_v2() {
  static const bool v = getenv("BN_REDUCE_V2") != nullptr;
  return v;
}

// Is the loop launch of a one-parent group left out in this iteration (the reduce launch that follows
// forms the leaves' normalisers again)?  loop_mode: the launch would be a loop launch; next_is_last:
// 0, or what the next iteration's launch will be (1 a full launch, 2 a reduce launch).
inline bool leaf_launch_folds(const FusedFamily& fam, bool loop_mode, int32_t next_is_last, int32_t fresh,
                              int32_t first_iteration) {
  return loop_mode && fam.wide_index2 < 0 && next_is_last == 2 && !fresh && !first_iteration &&
         fam.z_form == kZPerChunk && !fam.hard_wants_light && !env_no_recompute();
}

}  // namespace bn
