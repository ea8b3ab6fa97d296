// Host side of the specialised path (float32; the wide-parent / leaf-family part also float64): which groups get which kernel, the scratch
// the wide / fused kernels need, and the launches.  Each launch_* helper returns 1 if it
// handled the group, 0 if the caller must use the generic kernel, and a negative value on
// a launch failure.
#pragma once

#include "launch.h"
#include "cardinality_pack.h"
#include "fast_state.h"
#include "kernels_leaf.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "kernels_dense.h"
#include "kernels_fast.h"
#include "kernels_wide.h"
#include "plan.h"

namespace bn {

inline bool small_dim(int32_t d) {
  switch (d) {
#define X(V) \
  case V:    \
    return true;
    BN_SMALL_DIMS(X)
#undef X
  }
  return false;
}

// cardinalities of a wide parent whose kernels are part of the library proper ...
inline bool builtin_parent_dim(int32_t d) {
  switch (d) {
#define X(V) \
  case V:    \
    return true;
    BN_PARENT_DIMS(X)
#undef X
  }
  return false;
}
// ... and all that can be served: those, and any with a cardinality pack next to the library
inline bool parent_dim(int32_t d) { return builtin_parent_dim(d) || cardinality_pack(d) != nullptr; }

inline bool is_wide_group(const Plan& p, const VariableGroup& g) {
  return g.num_inputs > kWideFanIn && parent_dim(g.num_states);
}

template <typename U>
inline bool dev_upload(const std::vector<U>& host, U** dev, int64_t& bytes) {
  *dev = nullptr;
  if (host.empty()) return true;
  if (hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(U)) != hipSuccess) return false;
  bytes += (int64_t)(host.size() * sizeof(U));
  return hipMemcpy(*dev, host.data(), host.size() * sizeof(U), hipMemcpyHostToDevice) == hipSuccess;
}

inline void release_fast(Plan& p) {
  FastState* st = fast_state(p);
  if (!st) return;
  for (auto& w : st->wide) {
    if (w.d_covered) (void)hipFree(w.d_covered);
    if (w.d_uncovered) (void)hipFree(w.d_uncovered);
    if (w.d_chunk_first) (void)hipFree(w.d_chunk_first);
    if (w.part_mant) (void)hipFree(w.part_mant);
    if (w.part_exp) (void)hipFree(w.part_exp);
    if (w.zdiv_mant[0]) (void)hipFree(w.zdiv_mant[0]);
    if (w.zdiv_exp[0]) (void)hipFree(w.zdiv_exp[0]);
    if (w.prod_mant) (void)hipFree(w.prod_mant);
    if (w.prod_exp) (void)hipFree(w.prod_exp);
    if (w.scaled) (void)hipFree(w.scaled);
    if (w.scaled_prev) (void)hipFree(w.scaled_prev);
  }
  for (auto& kv : st->fused) {
    if (kv.second.d_obs_off) (void)hipFree(kv.second.d_obs_off);
    if (kv.second.d_c_off) (void)hipFree(kv.second.d_c_off);
    if (kv.second.d_src) (void)hipFree(kv.second.d_src);
    if (kv.second.em_partial) (void)hipFree(kv.second.em_partial);
    if (kv.second.z_state) (void)hipFree(kv.second.z_state);
    if (kv.second.ll_part) (void)hipFree(kv.second.ll_part);
    if (kv.second.m_state) (void)hipFree(kv.second.m_state);
  }
  for (auto& kv : st->tree)
    for (int32_t gen = 0; gen < 3; ++gen) {
      if (kv.second.in_par[gen]) (void)hipFree(kv.second.in_par[gen]);
      if (kv.second.in_own[gen]) (void)hipFree(kv.second.in_own[gen]);
      if (kv.second.own_prev[gen]) (void)hipFree(kv.second.own_prev[gen]);
      if (kv.second.out[gen]) (void)hipFree(kv.second.out[gen]);
    }
  if (st->d_ll_offs) (void)hipFree(st->d_ll_offs);
  if (st->d_unfused_entries) (void)hipFree(st->d_unfused_entries);
  delete st;
  p.fast = nullptr;
}

inline void invalidate_fast(Plan& p) {
  if (FastState* st = fast_state(p)) {
    for (auto& w : st->wide) {
      w.partials_valid = false;
      w.products_cached = false;
      w.combined_ahead = false;
      // nothing is carried over a reset: the alternating buffers start from the same side every time
      // (a captured launch sequence holds their addresses)
      w.zd_read = 0;
      if (w.scaled > w.scaled_prev) std::swap(w.scaled, w.scaled_prev);
    }
    for (auto& kv : st->fused) {
      kv.second.pending = false;
      kv.second.em_ready = false;
      kv.second.z_form = kZNone;
      kv.second.hard_z_mant = nullptr;
      kv.second.hard_z_exp = nullptr;
      kv.second.hard_wants_light = false;
    }
    std::fill(st->root_written.begin(), st->root_written.end(), 0);
  }
}

inline bool ingest_possible(const Plan& p) {
  FastState* st = fast_state(p);
  return st && st->ingest_ok;
}

// Can every launch of an EM step read float evidence from the caller's arrays?  The step must ingest
// (ingest_possible), leave out the loop launch before the reduce launch (three iterations at least),
// and every fused group must have the reduce kernel that reads those arrays (float plans, two-state
// leaves, four observations per lane; the arrays of a group's leaves at equal distances, as
// upload_family_sources found them).
inline bool caller_evidence_possible(const Plan& p, int32_t num_iterations) {
  FastState* st = fast_state(p);
  if (!st || !st->ingest_ok || p.dtype != 0 || p.tile != 256 || num_iterations < 3) return false;
  if (env_copy_evidence() || env_no_recompute() || env_reduce_v2()) return false;
  for (const auto& kv : st->fused)
    if (kv.second.wide_index2 >= 0 || p.fgroups[kv.first].dims[1] != 2 || kv.second.src_stride == 0) return false;
  return !st->fused.empty();
}

inline void set_evidence_in_caller(Plan& p, bool on) {
  if (FastState* st = fast_state(p)) st->evidence_in_caller = on;
}

// Can the launches that read compact hard evidence in place serve this group (its shape, room in LDS
// for a chunk's rows, room for the step's records)?  Defined next to those kernels (leaf_f32.hip).
bool hard_family_possible(const Plan& p, const FactorGroup& g, const FusedFamily& fam);

// The same question for compact hard evidence (kernels_hard.h): float plans, every leaf of every
// fused group an observed two-state leaf under one parent of at most 16 states, their state rows at
// equal distances.
inline bool caller_states_possible(const Plan& p, int32_t num_iterations) {
  FastState* st = fast_state(p);
  if (!st || !st->ingest_ok || !st->ingest_all_observed || p.dtype != 0 || p.tile != 256 || num_iterations < 3) return false;
  if (env_copy_evidence() || env_no_recompute()) return false;
  for (const auto& kv : st->fused)
    if (kv.second.wide_index2 >= 0 || !hard_family_possible(p, p.fgroups[kv.first], kv.second)) return false;
  return !st->fused.empty();
}

// rows: what the three forms of the state bytes stand for (6 floats: kernels_leaf.h, HardEvidenceRows)
inline void set_states_in_caller(Plan& p, bool on, const float* rows = nullptr) {
  if (FastState* st = fast_state(p)) {
    st->states_in_caller = on;
    if (rows) std::copy(rows, rows + 6, st->states_rows);
  }
}
inline void set_states_in_caller(Plan& p, bool on, double on_value, double off_value) {
  const float rows[6] = {(float)on_value, (float)off_value, (float)off_value, (float)on_value, (float)off_value, (float)off_value};
  set_states_in_caller(p, on, rows);
}

inline void set_ingest_next(Plan& p, bool on, bool from_states = false, double on_value = 1.0, double off_value = 0.0) {
  if (FastState* st = fast_state(p)) {
    st->ingest_next = on;
    st->ingest_states = on && from_states;
    st->ingest_on = on_value;
    st->ingest_off = off_value;
  }
}

inline bool ingest_states_possible(const Plan& p) {
  FastState* st = fast_state(p);
  return st && st->ingest_ok && st->ingest_all_observed;
}

inline void cpts_changed_fast(Plan& p) {
  if (FastState* st = fast_state(p)) std::fill(st->root_written.begin(), st->root_written.end(), 0);
}

// Tables for the fused tree iteration, if the plan qualifies (FastState::tree_ok).
inline int prepare_tree(Plan& p, FastState& st) {
  st.tree_ok = false;
  if (!st.wide.empty() || !st.fused.empty()) return 0;
  auto dense = [](int32_t d) { return d == 32 || d == 64; };
  bool any = false;
  for (const FactorGroup& g : p.fgroups) {
    if (g.dims.size() == 1) continue;
    if (g.dims.size() != 2 || !dense(g.dims[0]) || !dense(g.dims[1])) return 0;
    any = true;
  }
  if (!any) return 0;
  auto generation = [&](const EdgeLayout& e, int32_t gen) {
    // the message of a parentless factor (the prior) never changes: one array for both generations
    if (e.owner == e.var && p.parents[e.owner].empty()) return e.f2v;
    return gen == 0 ? e.f2v : e.v2f;
  };
  // messages a variable receives, except the one from factor `except_owner`
  auto others = [&](int32_t var, int32_t except_owner, int32_t gen, std::vector<int64_t>& out) {
    if (except_owner != var && gen != kTreeFresh) out.push_back(generation(p.edges[var], gen));
    for (int32_t child : p.children[var]) {
      if (child == except_owner || gen == kTreeFresh) continue;
      const auto& cp = p.parents[child];
      const int32_t pos = (int32_t)(std::find(cp.begin(), cp.end(), var) - cp.begin());
      out.push_back(generation(p.edges[p.parent_edge_base[child] + pos], gen));
    }
    if (p.observed_slot_of[var] >= 0) out.push_back(p.obs_off[p.observed_slot_of[var]]);
  };
  std::map<int32_t, std::vector<int64_t>> host[3][4];  // [generation][in_par, in_own, own_prev, out]
  for (size_t fi = 0; fi < p.fgroups.size(); ++fi) {
    const FactorGroup& g = p.fgroups[fi];
    if (g.dims.size() != 2) continue;
    double arrays = 0, fresh_arrays = 0;
    for (int32_t gen = 0; gen < 3; ++gen) {
      const int32_t to = gen == 0 ? 1 : 0;  // the generation this one writes
      for (int32_t j : g.nodes) {
        const int32_t parent = p.parents[j][0];
        std::vector<int64_t> par, own;
        others(parent, j, gen, par);
        others(j, j, gen, own);
        if ((par.empty() && gen != kTreeFresh) || par.size() > 2 || own.size() > 2) return 0;
        const double moved = (double)par.size() * g.dims[0] + (double)own.size() * g.dims[1] + g.dims[0] + g.dims[1];
        if (gen == 0) arrays += moved;
        if (gen == kTreeFresh) fresh_arrays += moved;
        par.resize(2, kNoArray);
        own.resize(2, kNoArray);
        const EdgeLayout& e_own = p.edges[j];
        const EdgeLayout& e_par = p.edges[p.parent_edge_base[j]];
        auto& h = host[gen];
        h[0][(int32_t)fi].insert(h[0][(int32_t)fi].end(), par.begin(), par.end());
        h[1][(int32_t)fi].insert(h[1][(int32_t)fi].end(), own.begin(), own.end());
        h[2][(int32_t)fi].push_back(generation(e_own, gen == kTreeFresh ? 0 : gen));  // not read when fresh
        h[3][(int32_t)fi].push_back(generation(e_par, to));
        h[3][(int32_t)fi].push_back(generation(e_own, to));
      }
    }
    st.tree[(int32_t)fi].bytes_per_observation = arrays * (double)p.elem_size;
    st.tree[(int32_t)fi].fresh_bytes_per_observation = fresh_arrays * (double)p.elem_size;
    st.tree[(int32_t)fi].exact_bytes_per_observation = (double)g.nodes.size() * g.dims[1] * (double)p.elem_size;
  }
  for (auto& kv : st.tree) {
    for (int32_t gen = 0; gen < 3; ++gen) {
      if (!dev_upload(host[gen][0][kv.first], &kv.second.in_par[gen], p.device_bytes) ||
          !dev_upload(host[gen][1][kv.first], &kv.second.in_own[gen], p.device_bytes) ||
          !dev_upload(host[gen][2][kv.first], &kv.second.own_prev[gen], p.device_bytes) ||
          !dev_upload(host[gen][3][kv.first], &kv.second.out[gen], p.device_bytes))
        return -2;
    }
  }
  st.tree_ok = true;
  return 0;
}

// One fused flooding iteration of generation `gen` (0: f2v arrays -> v2f arrays, 1: back,
// kTreeFresh: uniform messages -> f2v arrays) for the dense factor group fi.  Returns false on a launch failure.
inline bool launch_tree_iteration(Plan& p, size_t fi, int32_t gen, bool exact, hipStream_t s) {
  FastState* st = fast_state(p);
  const FactorGroup& g = p.fgroups[fi];
  const TreeGroupTables& t = st->tree.at((int32_t)fi);
  TreeFamilyTables tab{t.in_par[gen], t.in_own[gen], t.own_prev[gen], t.out[gen], g.d_cpt};
  const int32_t K = (int32_t)g.nodes.size();
  const dim3 grid((uint32_t)((int64_t)K * p.ntiles));
  const int32_t d0 = g.dims[0], sv = g.dims[1];
  if (p.dtype == 0) {
    using Fn = void (*)(float*, const float*, TreeFamilyTables, int32_t, int64_t);
    Fn fn = d0 == 64 ? (sv == 64 ? (exact ? factor_tree_mfma<64, 64, true> : factor_tree_mfma<64, 64, false>)
                                 : (exact ? factor_tree_mfma<64, 32, true> : factor_tree_mfma<64, 32, false>))
                     : (sv == 64 ? (exact ? factor_tree_mfma<32, 64, true> : factor_tree_mfma<32, 64, false>)
                                 : (exact ? factor_tree_mfma<32, 32, true> : factor_tree_mfma<32, 32, false>));
    launch(fn, grid, dim3(256), 0, s, static_cast<float*>(p.arena), static_cast<const float*>(p.cpts), tab, K, p.ntiles);
  } else {
    using Fn = void (*)(double*, const double*, TreeFamilyTables, int32_t, int64_t);
    Fn fn = d0 == 64 ? (sv == 64 ? (exact ? factor_tree_mfma_f64<64, 64, true> : factor_tree_mfma_f64<64, 64, false>)
                                 : (exact ? factor_tree_mfma_f64<64, 32, true> : factor_tree_mfma_f64<64, 32, false>))
                     : (sv == 64 ? (exact ? factor_tree_mfma_f64<32, 64, true> : factor_tree_mfma_f64<32, 64, false>)
                                 : (exact ? factor_tree_mfma_f64<32, 32, true> : factor_tree_mfma_f64<32, 32, false>));
    launch(fn, grid, dim3(256), 0, s, static_cast<double*>(p.arena), static_cast<const double*>(p.cpts), tab, K, p.ntiles);
  }
  return launch_ok();
}

inline bool is_fused_family_group(const Plan& p, size_t fi) {
  FastState* st = fast_state(p);
  return st && st->fused.count((int32_t)fi) != 0;
}

inline bool tree_iterations_possible(const Plan& p) {
  static const bool enabled = [] {
    const char* e = std::getenv("BN_TREE_FUSED");
    return !(e && e[0] == '0');
  }();
  FastState* st = fast_state(p);
  return enabled && st && st->tree_ok;
}

inline int prepare_fast(Plan& p) {
  // built in place, so that release_fast frees whatever a failed preparation had allocated
  p.fast = new FastState;
  FastState& st = *fast_state(p);
  st.vgroup_skip.assign(p.vgroups.size(), 0);
  st.root_written.assign(p.fgroups.size(), 0);
  // wide variable nodes
  for (size_t gi = 0; gi < p.vgroups.size(); ++gi) {
    const VariableGroup& g = p.vgroups[gi];
    if (!is_wide_group(p, g)) continue;
    for (size_t k = 0; k < g.nodes.size(); ++k) {
      WideNode w;
      w.node = g.nodes[k];
      w.vgroup = (int32_t)gi;
      w.vpos = (int32_t)k;
      w.S = g.num_states;
      w.I = g.num_inputs;
      st.wide_of_node[w.node] = (int32_t)st.wide.size();
      st.wide.push_back(std::move(w));
    }
  }
  // Leaf families around one or two wide parents.  A wide node can carry several one-parent family
  // groups (children of different cardinalities), each with its own range of chunks; a two-parent
  // group has both its parents to itself.
  std::vector<char> handled(p.n_nodes, 0);
  struct Candidate {
    int32_t fi, num_parents, widx[2];
  };
  std::vector<Candidate> candidates;
  for (size_t fi = 0; fi < p.fgroups.size(); ++fi) {
    const FactorGroup& g = p.fgroups[fi];
    const int32_t num_parents = (int32_t)g.dims.size() - 1;
    if (num_parents == 2) {
      // the two-parent shapes with a fused kernel: in the library, or as a shape pack next to it
      if (!two_parent_shape(g.dims[0], g.dims[1], g.dims[2]) && !shape_pack(g.dims[0], g.dims[1], g.dims[2])) continue;
    } else if (num_parents != 1 || !parent_dim(g.dims[0]) || !small_dim(g.dims[1])) {
      continue;
    }
    const std::vector<int32_t>& family_parents = p.parents[g.nodes[0]];
    Candidate c{(int32_t)fi, num_parents, {-1, -1}};
    bool ok = g.nodes.size() >= 16;
    for (int32_t q = 0; q < num_parents && ok; ++q) {
      auto wit = st.wide_of_node.find(family_parents[q]);
      ok = wit != st.wide_of_node.end();
      if (ok) c.widx[q] = wit->second;
    }
    ok = ok && (num_parents == 1 || c.widx[0] != c.widx[1]);
    for (int32_t j : g.nodes) ok = ok && p.parents[j] == family_parents && p.children[j].empty();
    if (ok) candidates.push_back(c);
  }
  std::vector<char> accepted(p.fgroups.size(), 0), exclusive(st.wide.size(), 0);
  for (const Candidate& c : candidates) {
    if (c.num_parents != 2) continue;
    const int32_t K = (int32_t)p.fgroups[c.fi].nodes.size();
    bool ok = true;
    for (int32_t q = 0; q < 2; ++q) ok = ok && !exclusive[c.widx[q]] && st.wide[c.widx[q]].I - K <= kMaxUncovered;
    if (!ok) continue;
    accepted[c.fi] = 1;
    exclusive[c.widx[0]] = exclusive[c.widx[1]] = 1;
  }
  for (size_t wi = 0; wi < st.wide.size(); ++wi) {
    if (exclusive[wi]) continue;
    int32_t total = 0, largest = -1;
    for (const Candidate& c : candidates) {
      if (c.num_parents != 1 || c.widx[0] != (int32_t)wi) continue;
      const int32_t K = (int32_t)p.fgroups[c.fi].nodes.size();
      total += K;
      if (largest < 0 || K > (int32_t)p.fgroups[largest].nodes.size()) largest = c.fi;
    }
    if (largest < 0) continue;
    if (st.wide[wi].I - total <= kMaxUncovered) {
      for (const Candidate& c : candidates)
        if (c.num_parents == 1 && c.widx[0] == (int32_t)wi) accepted[c.fi] = 1;
    } else if (st.wide[wi].I - (int32_t)p.fgroups[largest].nodes.size() <= kMaxUncovered) {
      accepted[largest] = 1;
    }
  }
  for (const Candidate& c : candidates) {
    if (!accepted[c.fi]) continue;
    const FactorGroup& g = p.fgroups[c.fi];
    const std::vector<int32_t>& family_parents = p.parents[g.nodes[0]];
    // covered slots of each parent, in the factor group's node order
    for (int32_t q = 0; q < c.num_parents; ++q) {
      WideNode& w = st.wide[c.widx[q]];
      const auto& ch = p.children[family_parents[q]];
      for (int32_t j : g.nodes) w.covered.push_back((int32_t)(std::find(ch.begin(), ch.end(), j) - ch.begin()));
      w.fused_fgroups.push_back(c.fi);
      if (w.fused_fgroup < 0) w.fused_fgroup = c.fi;
    }
    FusedFamily& fam = st.fused[c.fi];  // in place: release_fast frees it whatever happens below
    fam.wide_index = c.widx[0];
    fam.wide_index2 = c.widx[1];
    std::vector<int64_t> obs, cs;
    for (int32_t j : g.nodes) {
      obs.push_back(p.observed_slot_of[j] >= 0 ? p.obs_off[p.observed_slot_of[j]] : p.ones_off);
      fam.ev_index.push_back(p.observed_slot_of[j]);
      cs.push_back(p.c_off[j]);
      handled[j] = 1;
    }
    if (!dev_upload(obs, &fam.d_obs_off, p.device_bytes) || !dev_upload(cs, &fam.d_c_off, p.device_bytes)) return -2;
    if (hipMalloc(reinterpret_cast<void**>(&fam.d_src), g.nodes.size() * sizeof(void*)) != hipSuccess) return -2;
    p.device_bytes += (int64_t)(g.nodes.size() * sizeof(void*));
    fam.slots = (int32_t)(p.ntiles * leaf_parts(p));
    const size_t em_elems = g.nodes.size() * (size_t)g.cpt_elems * fam.slots;
    if (hipMalloc(reinterpret_cast<void**>(&fam.em_partial), em_elems * (size_t)p.elem_size) != hipSuccess) return -2;
    p.device_bytes += (int64_t)(em_elems * (size_t)p.elem_size);
    if (hipMalloc(reinterpret_cast<void**>(&fam.z_state), g.nodes.size() * (size_t)p.npad * (size_t)p.elem_size) != hipSuccess)
      return -2;
    p.device_bytes += (int64_t)(g.nodes.size() * (size_t)p.npad * (size_t)p.elem_size);
    if (c.num_parents == 2) {
      const size_t m_elems = g.nodes.size() * (size_t)(g.dims[0] + g.dims[1]) * p.npad;
      if (hipMalloc(reinterpret_cast<void**>(&fam.m_state), m_elems * (size_t)p.elem_size) != hipSuccess) return -2;
      p.device_bytes += (int64_t)(m_elems * (size_t)p.elem_size);
    }
  }
  // The standalone wide kernels (product / emission) exist for float32 only: a float64 plan takes
  // the wide path for a variable group only if a leaf-family kernel serves every node of it, and
  // gives up the fusion of the others (their groups then run on the generic kernels).
  st.vgroup_wide.assign(p.vgroups.size(), 0);
  auto unfuse = [&](int32_t fi) {
    auto it = st.fused.find(fi);
    if (it == st.fused.end()) return;
    FusedFamily& fam = it->second;
    for (void* buffer : {(void*)fam.d_obs_off, (void*)fam.d_c_off, (void*)fam.d_src, fam.em_partial, fam.z_state, fam.m_state})
      if (buffer) (void)hipFree(buffer);
    for (int32_t j : p.fgroups[fi].nodes) handled[j] = 0;
    for (int32_t widx : {fam.wide_index, fam.wide_index2}) {
      if (widx < 0) continue;
      st.wide[widx].fused_fgroup = -1;
      st.wide[widx].fused_fgroups.clear();
      st.wide[widx].covered.clear();
    }
    st.fused.erase(it);
  };
  for (bool changed = true; changed;) {
    changed = false;
    for (size_t gi = 0; gi < p.vgroups.size(); ++gi) {
      const VariableGroup& g = p.vgroups[gi];
      if (!is_wide_group(p, g)) continue;
      size_t fused_nodes = 0;
      for (int32_t j : g.nodes) fused_nodes += st.wide[st.wide_of_node[j]].fused_fgroup >= 0 ? 1 : 0;
      st.vgroup_wide[gi] = (p.dtype == 0 || fused_nodes == g.nodes.size()) ? 1 : 0;
      if (st.vgroup_wide[gi] || fused_nodes == 0) continue;
      for (int32_t j : g.nodes) {
        const std::vector<int32_t> groups = st.wide[st.wide_of_node[j]].fused_fgroups;  // copy: unfuse edits it
        for (int32_t fi : groups) {
          unfuse(fi);
          changed = true;
        }
      }
    }
  }
  for (size_t gi = 0; gi < p.vgroups.size(); ++gi) {
    bool all = true;
    for (int32_t j : p.vgroups[gi].nodes) all = all && handled[j];
    st.vgroup_skip[gi] = all ? 1 : 0;
  }
  {
    std::vector<char> ingestible(p.n_nodes, 0);
    for (const auto& kv : st.fused)
      if (kv.second.wide_index2 < 0 && p.fgroups[kv.first].dims[1] == 2)
        for (int32_t j : p.fgroups[kv.first].nodes) ingestible[j] = 1;
    st.ingest_ok = kLeafVec == 2 && !p.observed.empty();
    for (int32_t o : p.observed) st.ingest_ok = st.ingest_ok && ingestible[o];
    st.ingest_all_observed = st.ingest_ok;
    for (const auto& kv : st.fused)
      if (kv.second.wide_index2 < 0 && p.fgroups[kv.first].dims[1] == 2)
        for (int32_t e : kv.second.ev_index) st.ingest_all_observed = st.ingest_all_observed && e >= 0;
  }
  {
    std::vector<int64_t> offs;
    for (int32_t j = 0; j < p.n_nodes; ++j)
      if (!handled[j]) offs.push_back(p.c_off[j]);
    st.n_ll_offs = (int64_t)offs.size();
    if (!dev_upload(offs, &st.d_ll_offs, p.device_bytes)) return -2;
    std::vector<int32_t> entries;
    for (size_t fi = 0; fi < p.fgroups.size(); ++fi) {
      if (st.fused.count((int32_t)fi)) continue;
      const FactorGroup& g = p.fgroups[fi];
      for (size_t k = 0; k < g.nodes.size(); ++k)
        for (int64_t x = 0; x < g.cpt_elems; ++x) entries.push_back((int32_t)(g.cpt_off[k] + x));
    }
    st.n_unfused_entries = (int32_t)entries.size();
    if (!dev_upload(entries, &st.d_unfused_entries, p.device_bytes)) return -2;
  }
  // chunk geometry + scratch
  for (auto& w : st.wide) {
    if (p.dtype != 0 && w.fused_fgroup < 0) continue;  // not on the wide path in a float64 plan
    if (w.fused_fgroup < 0) {
      for (int32_t i = 0; i < w.I - 2; ++i) w.covered.push_back(i);
      w.uncovered = {w.I - 2, w.I - 1};
    } else {
      std::vector<char> is_covered(w.I, 0);
      for (int32_t slot : w.covered) is_covered[slot] = 1;
      w.uncovered.clear();
      for (int32_t i = 0; i < w.I; ++i)
        if (!is_covered[i]) w.uncovered.push_back(i);
    }
    const int32_t n = (int32_t)w.covered.size();
    // one round of resident waves (4 per SIMD x 1024 SIMDs) per launch, never a partly filled
    // second one: measured best on the 784-leaf model (8 chunks at 60 000 observations; 7 and 9
    // are both 5 % slower, 13 and more pay in the combine)
    int64_t want = std::max<int64_t>(1, 4096 / (p.ntiles * leaf_parts(p)));
    // small batches cannot fill the chip anyway; there the combine sets the limit: its waves
    // each walk a share of the chunks, three groups of four at most
    want = std::min<int64_t>(want, 4 * kCombineMaxWaves * 3);
    // in between (a few thousand observations) the combine, which reads every chunk's partial
    // product, costs as much as the leaf-family launch itself: 32 chunks measured best at 7 500
    if (p.ntiles >= 16) want = std::min<int64_t>(want, 32);
    if (env_wide_chunks() > 0) want = env_wide_chunks();
    int32_t chunks = (int32_t)std::max<int64_t>(1, std::min<int64_t>(want, n / 8));
    // Small batches: the waves of a block share a chunk (FusedFamily::split), so the same number of
    // waves leaves a quarter of the partials for the combine to read.  One-parent groups only (the
    // loop launches of two-parent groups have no such mode).
    bool share = w.fused_fgroup >= 0 && p.ntiles * leaf_parts(p) <= env_share_max() && chunks >= 2 * kWavesPerBlock;
    for (int32_t fi : w.fused_fgroups) share = share && st.fused[fi].wide_index2 < 0 && p.fgroups[fi].dims[1] == 2;
    if (share) chunks /= kWavesPerBlock;
    // the leaf-family kernel keeps a chunk's exponents in 16 bits
    if (w.fused_fgroup >= 0) chunks = std::max(chunks, (n + kLeafMaxChunkLen - 1) / kLeafMaxChunkLen);
    w.chunk_len = (n + chunks - 1) / chunks;
    w.chunks = (n + w.chunk_len - 1) / w.chunk_len;
    std::vector<int32_t> chunk_first;
    if (w.fused_fgroup >= 0) {
      // every fused group gets whole chunks of its own, one group after the other
      w.chunks = 0;
      int32_t group_first = 0;
      for (int32_t fi : w.fused_fgroups) {
        FusedFamily& fam = st.fused[fi];
        const int32_t members = (int32_t)p.fgroups[fi].nodes.size();
        fam.chunk_len = w.chunk_len;
        fam.split = share ? kWavesPerBlock : 1;
        fam.chunks = (members + w.chunk_len - 1) / w.chunk_len;
        fam.chunk_base = w.chunks;
        for (int32_t c = 0; c < fam.chunks; ++c) chunk_first.push_back(group_first + c * w.chunk_len);
        w.chunks += fam.chunks;
        group_first += members;
      }
    } else {
      for (int32_t c = 0; c < w.chunks; ++c) chunk_first.push_back(c * w.chunk_len);
    }
    chunk_first.push_back(n);
    if (!dev_upload(chunk_first, &w.d_chunk_first, p.device_bytes)) return -2;
    if (!dev_upload(w.covered, &w.d_covered, p.device_bytes) || !dev_upload(w.uncovered, &w.d_uncovered, p.device_bytes))
      return -2;
    const size_t rows = (size_t)w.chunks * w.S * p.npad;
    if (hipMalloc(reinterpret_cast<void**>(&w.part_mant), rows * sizeof(double)) != hipSuccess) return -2;
    if (hipMalloc(reinterpret_cast<void**>(&w.part_exp), rows * sizeof(int32_t)) != hipSuccess) return -2;
    if (hipMalloc(reinterpret_cast<void**>(&w.scaled), (size_t)w.S * p.npad * sizeof(double)) != hipSuccess) return -2;
    if (hipMalloc(reinterpret_cast<void**>(&w.scaled_prev), (size_t)w.S * p.npad * sizeof(double)) != hipSuccess) return -2;
    p.device_bytes += (int64_t)(rows * 12 + (size_t)w.S * p.npad * 16);
    if (w.fused_fgroup >= 0) {
      // divisors of the chunk partial products: 1 until a loop launch of a one-parent group says otherwise
      const size_t zd = (size_t)w.chunks * p.npad;
      if (hipMalloc(reinterpret_cast<void**>(&w.zdiv_mant[0]), 2 * zd * sizeof(double)) != hipSuccess) return -2;
      if (hipMalloc(reinterpret_cast<void**>(&w.zdiv_exp[0]), 2 * zd * sizeof(int32_t)) != hipSuccess) return -2;
      w.zdiv_mant[1] = w.zdiv_mant[0] + zd;
      w.zdiv_exp[1] = w.zdiv_exp[0] + zd;
      p.device_bytes += (int64_t)(zd * 24);
      wide_divisor_one<int32_t><<<dim3((uint32_t)std::min<size_t>(1024, (2 * zd + 255) / 256)), dim3(256), 0, 0>>>(
          w.zdiv_mant[0], w.zdiv_exp[0], (int64_t)(2 * zd));
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(0) != hipSuccess) return -2;
      bool one_parent_only = true;
      for (int32_t fi : w.fused_fgroups) one_parent_only = one_parent_only && st.fused[fi].wide_index2 < 0;
      if (one_parent_only) {
        if (hipMalloc(reinterpret_cast<void**>(&w.prod_mant), (size_t)w.S * p.npad * sizeof(double)) != hipSuccess) return -2;
        if (hipMalloc(reinterpret_cast<void**>(&w.prod_exp), (size_t)w.S * p.npad * sizeof(int32_t)) != hipSuccess) return -2;
        p.device_bytes += (int64_t)((size_t)w.S * p.npad * 12);
      }
    }
    for (int32_t fi : w.fused_fgroups) {
      FusedFamily& fam = st.fused[fi];
      if (fam.ll_part) continue;  // a two-parent group is visited from both its parents
      fam.ll_rows = fam.chunks;
      // rows for the reduce launch's own cut of the families: at small batches it wants many short chunks
      // (about three resident waves per SIMD: 3 072 / tiles), each of which leaves one row
      fam.ll_rows_cap = std::max(fam.chunks, (int32_t)std::min<int64_t>(128, std::max<int64_t>(32, 3072 / std::max<int64_t>(1, p.ntiles) + 1)));
      if (hipMalloc(reinterpret_cast<void**>(&fam.ll_part), (size_t)fam.ll_rows_cap * p.npad * sizeof(double)) != hipSuccess)
        return -2;
      p.device_bytes += (int64_t)((size_t)fam.ll_rows_cap * p.npad * sizeof(double));
    }
  }
  if (int rc = prepare_tree(p, st)) return rc;
  // which launches touch the lazy region (see variable_needs_arrays / factor_needs_arrays)
  st.vgroup_lazy.assign(p.vgroups.size(), 0);
  st.fgroup_lazy.assign(p.fgroups.size(), 0);
  auto any_lazy = [&](const std::vector<int64_t>& offs) {
    for (int64_t off : offs)
      if (off != kNoArray && p.in_lazy(off)) return true;
    return false;
  };
  for (size_t gi = 0; gi < p.vgroups.size(); ++gi)
    st.vgroup_lazy[gi] = any_lazy(p.vgroups[gi].in_off) || any_lazy(p.vgroups[gi].out_off) ? 1 : 0;
  for (size_t fi = 0; fi < p.fgroups.size(); ++fi)
    st.fgroup_lazy[fi] = any_lazy(p.fgroups[fi].in_off) || any_lazy(p.fgroups[fi].out_off) ? 1 : 0;
  for (auto& w : st.wide) {
    const VariableGroup& g = p.vgroups[w.vgroup];
    auto slot_lazy = [&](int32_t slot) {
      const int64_t in = g.in_off[(size_t)w.vpos * g.num_inputs + slot], out = g.out_off[(size_t)w.vpos * g.num_inputs + slot];
      return p.in_lazy(in) || (out != kNoArray && p.in_lazy(out));
    };
    for (int32_t slot : w.covered) w.covered_lazy = w.covered_lazy || slot_lazy(slot);
    for (int32_t slot : w.uncovered) w.uncovered_lazy = w.uncovered_lazy || slot_lazy(slot);
  }
  return 0;
}

// Does the variable-pass launch of group gi read or write a message array of the lazy region?
// (Asked while that region has no memory yet: the caller then backs it before the launch.)
// after_first: the question is about a later iteration of a call whose first one runs on fresh
// messages (asked before the call starts: the fused kernels will have written the partial products by then)
inline bool variable_needs_arrays(const Plan& p, size_t gi, int32_t fresh, bool after_first = false) {
  FastState* st = fast_state(p);
  if (!st) return true;
  if (st->vgroup_skip[gi]) return false;
  if (!st->vgroup_wide[gi]) return st->vgroup_lazy[gi] != 0;
  for (int32_t j : p.vgroups[gi].nodes) {
    const WideNode& w = st->wide[st->wide_of_node[j]];
    const bool fused = w.fused_fgroup >= 0;
    if (w.uncovered_lazy) return true;
    // covered slots: read by the standalone product, read and written by the standalone emission
    if (w.covered_lazy && (!fused || (!fresh && !w.partials_valid && !after_first))) return true;
  }
  return false;
}

// The same for the factor-pass launch of group fi.  A fused group touches its arrays when it starts
// from stored messages or runs in full mode.
inline bool factor_needs_arrays(const Plan& p, size_t fi, int32_t fresh, int32_t first_iteration, int32_t last_iteration,
                                int32_t reduce_last) {
  FastState* st = fast_state(p);
  if (!st) return true;
  if (!st->fgroup_lazy[fi]) return false;
  if (!st->fused.count((int32_t)fi)) return true;
  const bool from_arrays = !fresh && first_iteration;
  const bool full = last_iteration && !reduce_last;
  return from_arrays || full;
}

inline WideArgs wide_args(const Plan& p, const VariableGroup& g, const WideNode& w, int32_t fresh) {
  WideArgs a;
  a.in_off = g.d_in + (size_t)w.vpos * g.num_inputs;
  a.out_off = g.d_out + (size_t)w.vpos * g.num_inputs;
  a.c_off = g.c_off[w.vpos];
  a.I = w.I;
  a.chunks = w.chunks;
  a.chunk_len = w.chunk_len;
  a.n_covered = (int32_t)w.covered.size();
  a.covered = w.d_covered;
  a.chunk_first = w.d_chunk_first;
  a.uncovered = w.d_uncovered;
  a.n_uncovered = (int32_t)w.uncovered.size();
  a.part_mant = w.part_mant;
  a.part_exp = w.part_exp;
  a.zdiv_mant = w.zdiv_mant[w.zd_read];
  a.zdiv_exp = w.zdiv_exp[w.zd_read];
  a.prod_mant = w.prod_mant;
  a.prod_exp = w.prod_exp;
  a.prod_mode = (fresh || !w.prod_mant) ? 0 : (w.products_cached ? 2 : 1);
  a.early_zdiv_mant = nullptr;
  a.early_zdiv_exp = nullptr;
  a.early_scaled = nullptr;
  a.root_cpt = nullptr;
  a.root_out = 0;
  a.scaled = w.scaled_prev;  // the combine writes over the older of the two, then they swap
  a.npad = p.npad;
  a.ntiles = p.ntiles;
  a.fresh = fresh;
  {
    // (1/S)^n as mantissa * 2^exponent
    const double log2p = -(double)w.covered.size() * std::log2((double)w.S);
    const double fl = std::floor(log2p);
    a.fresh_mant = std::exp2(log2p - fl);
    a.fresh_exp = (int32_t)fl;
  }
  return a;
}

// The wide node's kernels for S states as addresses (all of them take (arena, WideArgs))
template <typename T, int S>
inline WideKernelRefs wide_kernels_for() {
  WideKernelRefs k{};
  if constexpr (std::is_same<T, float>::value) {
    k.product = reinterpret_cast<const void*>(&wide_product_fast<S>);
    k.emit = reinterpret_cast<const void*>(&wide_emit_fast<S>);
  } else {
    k.product = reinterpret_cast<const void*>(&wide_product_any<T, S>);
  }
  if constexpr (S <= 16) {  // (one wave per state: a block of 64 * S threads)
    k.combine_states = reinterpret_cast<const void*>(&wide_combine_states<T, S>);
    k.combine_states_lds = [](int32_t n) -> size_t { return combine_states_lds_bytes<S>(n); };
  }
  k.combine_fast = reinterpret_cast<const void*>(&wide_combine_fast<T, S>);
  k.combine_lds = [](int32_t nw) -> size_t { return combine_lds_bytes<S>(nw); };
  return k;
}

template <typename T>
inline WideKernelRefs wide_kernels(int32_t states) {
  switch (states) {
#define X(SV) \
  case SV:    \
    return wide_kernels_for<T, SV>();
    BN_PARENT_DIMS(X)
#undef X
  }
  if (const CardinalityPack* pack = cardinality_pack(states)) return pack->wide(std::is_same<T, float>::value ? 0 : 1);
  return WideKernelRefs{};
}

template <typename T>
inline bool launch_wide_node(const Plan& p, const VariableGroup& g, WideNode& w, T* arena, int32_t fresh,
                             hipStream_t s, bool next_folds = false) {
  using WideFn = void (*)(T*, WideArgs);
  using WideReadFn = void (*)(const T*, WideArgs);
  const int32_t S = g.num_states;
  const WideKernelRefs k = wide_kernels<T>(S);
  if (!k.combine_fast) return false;
  if (w.combined_ahead) {  // the previous iteration's launch did this combine too
    w.combined_ahead = false;
    return true;
  }
  const bool fused = w.fused_fgroup >= 0;
  if (!fresh && (!fused || !w.partials_valid)) w.products_cached = false;  // the partial products are about to be rebuilt
  WideArgs a = wide_args(p, g, w, fresh);
  // This iteration's loop launches are all left out (next_folds: the caller checked every fused group
  // of the node) and nothing else feeds the node but its own parentless factor, whose message does
  // not change: the next iteration's combine has all its inputs now and runs in this launch.
  const bool two = next_folds && fused && a.prod_mode == 2 && w.zdiv_mant[0] && p.parents[w.node].empty() &&
                   w.uncovered.size() == 1 && w.uncovered[0] == w.I - 1 && !env_no_double_combine();
  if (two) {
    a.early_zdiv_mant = w.zdiv_mant[w.zd_read];
    a.early_zdiv_exp = w.zdiv_exp[w.zd_read];
    a.early_scaled = w.scaled_prev;
    a.zdiv_mant = w.zdiv_mant[w.zd_read ^ 1];
    a.zdiv_exp = w.zdiv_exp[w.zd_read ^ 1];
    a.scaled = w.scaled;  // the older product is not needed any more: no launch reads it
  }
  // fresh messages, a parentless node whose factor is alone in its group: its message rides along
  int32_t root_group = -1;
  if (fresh && fused && p.parents[w.node].empty()) {
    FastState* st = fast_state(p);
    const int32_t fi = p.fgroup_of[w.node];
    const FactorGroup& rg = p.fgroups[fi];
    if (rg.dims.size() == 1 && rg.nodes.size() == 1 && !st->root_written[fi] && !p.in_lazy(rg.out_off[0])) {
      a.root_cpt = static_cast<const char*>(p.cpts) + (size_t)rg.cpt_off[0] * p.elem_size;
      a.root_out = rg.out_off[0];
      root_group = fi;
    }
  }
  if (!fresh && (!fused || !w.partials_valid)) {
    if (w.zdiv_mant[0]) {
      // partial products rebuilt from the stored messages are complete
      const int64_t zd = 2 * (int64_t)w.chunks * p.npad;
      launch(wide_divisor_one<int32_t>, dim3((uint32_t)std::min<int64_t>(1024, (zd + 255) / 256)), dim3(256), 0, s,
             w.zdiv_mant[0], w.zdiv_exp[0], zd);
      if (!launch_ok()) return false;
    }
    // (the float kernel moves 16 bytes per lane: one wave per chunk and tile)
    const int64_t waves = (int64_t)w.chunks * p.ntiles * (std::is_same<T, float>::value ? 1 : leaf_parts(p));
    launch(reinterpret_cast<WideReadFn>(const_cast<void*>(k.product)), dim3(wave_blocks(waves)), dim3(256), 0, s, arena, a);
    if (!launch_ok()) return false;
  }
  // (large batches that also store the product over the chunks: measured 22 against 27 us at 60 000
  // observations on the kernel with one observation per lane and a wave per four chunks)
  const bool states_kernel = a.prod_mode == 1 ? p.ntiles < 64 : a.prod_mode != 2;
  // (one wave per state: a block of 64 * S threads, so up to 16 states)
  if (k.combine_states && !fresh && states_kernel && p.ntiles <= env_combine_states_max() &&
      (int32_t)w.uncovered.size() <= kMaxUncovered) {
    // a handful of blocks: the launch lasts as long as its longest chain, so one wave per state
    const size_t lds = k.combine_states_lds(a.n_uncovered);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(k.combine_states, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return false;
    launch(reinterpret_cast<WideFn>(const_cast<void*>(k.combine_states)), dim3((uint32_t)(p.ntiles * (p.tile / 64))), dim3(64 * S), lds, s, arena, a);
  } else {
    // waves per block: one per group of four chunks, none to spare while the messages are fresh
    int32_t nw = (fresh || a.prod_mode == 2) ? 1 : std::max(1, std::min(kCombineMaxWaves, (w.chunks + 3) / 4));
    while (nw > 1 && k.combine_lds(nw) > 128 * 1024) --nw;  // (a wave then takes more of the chunks)
    const size_t lds = k.combine_lds(nw);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(k.combine_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return false;
    launch(reinterpret_cast<WideFn>(const_cast<void*>(k.combine_fast)), dim3((uint32_t)(p.ntiles * (p.tile / 64))), dim3(64 * nw), lds, s, arena, a);
  }
  if (!launch_ok()) return false;
  if (a.prod_mode == 1) w.products_cached = true;
  if (root_group >= 0) fast_state(p)->root_written[root_group] = 1;
  if (two) {
    w.combined_ahead = true;  // (two swaps of the scaled products, two flips of the divisor arrays)
  } else {
    std::swap(w.scaled, w.scaled_prev);
    if (!fresh && w.zdiv_mant[0]) w.zd_read ^= 1;
  }
  if (!fused) {
    if (!k.emit) return false;
    launch(reinterpret_cast<WideFn>(const_cast<void*>(k.emit)), dim3(wave_blocks((int64_t)w.chunks * p.ntiles)), dim3(256), 0, s, arena, a);
    if (!launch_ok()) return false;
  }
  return true;
}

// Label and compulsory bytes of the variable-pass launch for group gi when a special path
// serves it (profiling only).  Returns false when the defaults apply.
inline bool variable_profile(const Plan& p, size_t gi, std::string& label, double& bytes) {
  FastState* st = fast_state(p);
  if (!st) return false;
  const VariableGroup& g = p.vgroups[gi];
  if (st->vgroup_skip[gi]) {
    label = "variable_pass(fused into leaf_family)[I=" + std::to_string(g.num_inputs) + ",S=" +
            std::to_string(g.num_states) + "]";
    bytes = 0;
    return true;
  }
  if (!st->vgroup_wide[gi]) return false;
  bool any_fused = false;
  for (int32_t j : g.nodes) any_fused = any_fused || st->wide[st->wide_of_node[j]].fused_fgroup >= 0;
  if (!any_fused) return false;
  label = "wide_combine[I=" + std::to_string(g.num_inputs) + ",S=" + std::to_string(g.num_states) + "]";
  // what is left of the variable pass once the covered slots are the leaf-family kernel's: the other
  // slots' messages in and out, and the normaliser
  bytes = 0;
  for (int32_t j : g.nodes) {
    const WideNode& w = st->wide[st->wide_of_node[j]];
    const bool fused = w.fused_fgroup >= 0;
    double live_out = 0;
    std::vector<char> is_covered(w.I, 0);
    if (fused)
      for (int32_t slot : w.covered) is_covered[slot] = 1;
    for (int32_t slot = 0; slot < w.I; ++slot)
      if (!is_covered[slot]) live_out += g.out_off[(size_t)w.vpos * g.num_inputs + slot] != kNoArray;
    const double slots_in = fused ? (double)w.uncovered.size() : (double)w.I;
    bytes += ((double)w.S * (slots_in + live_out) + 1.0) * (double)p.N * (double)p.elem_size;
  }
  return true;
}

// flops: the additions, multiplications and divisions of one flooding iteration of the group's
// families as the message formulas state them (with_counts: plus the expected-count products of
// the last iteration of a statistics call); written only for fused one-parent groups
inline bool factor_profile(const Plan& p, size_t fi, std::string& label, double& bytes, double* flops = nullptr,
                           bool with_counts = false) {
  FastState* st = fast_state(p);
  if (!st) return false;
  auto it = st->fused.find((int32_t)fi);
  if (it == st->fused.end()) return false;
  const FactorGroup& g = p.fgroups[fi];
  if (flops && it->second.wide_index2 < 0) {
    // per family and observation: message to the parent D0 (2S - 1); the parent's leave-one-out
    // message, normalised, D0 divisions + D0 additions + D0 multiplications; message to the leaf
    // S (2 D0 - 1); the leaf's normaliser and its message back 3 S; counts 3 D0 S
    const double d0 = g.dims[0], sv = g.dims[1];
    const double per = d0 * (2 * sv - 1) + 3 * d0 + sv * (2 * d0 - 1) + 3 * sv + (with_counts ? 3 * d0 * sv : 0.0);
    *flops = per * (double)g.nodes.size() * (double)p.N;
  }
  const WideNode& w = st->wide[it->second.wide_index];
  // the factor pass itself + the covered part of the wide parent's variable pass + the leaf
  // variables' whole variable pass (inputs observation + factor message, outputs message + c)
  double leaf = 0;
  for (int32_t j : g.nodes) {
    const int32_t fan_in = 1 + (p.observed_slot_of[j] >= 0 ? 1 : 0);
    leaf += (double)p.states[j] * fan_in + (double)p.states[j] + 1;
  }
  bytes += 2.0 * w.S * (double)g.nodes.size() * (double)p.N * (double)p.elem_size +
           leaf * (double)p.N * (double)p.elem_size;
  if (it->second.wide_index2 >= 0) {
    const WideNode& w2 = st->wide[it->second.wide_index2];
    bytes += 2.0 * w2.S * (double)g.nodes.size() * (double)p.N * (double)p.elem_size;
  }
  label = "leaf_family[";
  for (size_t d = 0; d < g.dims.size(); ++d) label += (d ? "x" : "") + std::to_string(g.dims[d]);
  label += ",K=" + std::to_string(g.nodes.size()) + "]";
  return true;
}

template <typename T>
// last_but_one_before_reduce: this is the iteration before the last one and the last one ends in reduce mode
int launch_variable_fast(Plan& p, VariableGroup& g, T* arena_t, int32_t fresh, hipStream_t s,
                         int32_t first_iteration = 0, bool last_but_one_before_reduce = false) {
  FastState* st = fast_state(p);
  const size_t gi = (size_t)(&g - p.vgroups.data());
  if (st && st->vgroup_skip[gi]) return 1;  // done inside the leaf-family kernel
  if (st && st->vgroup_wide[gi]) {
    for (int32_t j : g.nodes) {
      WideNode& w = st->wide[st->wide_of_node[j]];
      // will the factor pass of this iteration leave out every loop launch that feeds this node?
      bool next_folds = last_but_one_before_reduce && !w.fused_fgroups.empty();
      for (int32_t fi : w.fused_fgroups)
        next_folds = next_folds && leaf_launch_folds(st->fused[fi], true, 2, fresh, first_iteration);
      if (!launch_wide_node<T>(p, g, w, arena_t, fresh, s, next_folds)) return -1;
    }
    return 1;
  }
  if (g.num_inputs > 4) return 0;
  if (VarFn<T> fn = var_small_lookup<T>(g.num_inputs, g.num_states)) {
    // whole node in registers
    launch(fn, dim3(wave_blocks((int64_t)g.nodes.size() * p.ntiles)), dim3(256), 0, s, arena_t, g.d_in, g.d_out, g.d_c,
           (int32_t)g.nodes.size(), p.ntiles, fresh);
    return launch_ok() ? 1 : -1;
  }
  if constexpr (!std::is_same<T, float>::value) {
    // 32 or 64 states in float64: states split over the eight waves of a block (one tile of 128 observations)
    if ((g.num_states == 32 || g.num_states == 64) && g.num_inputs >= 1 && g.num_inputs <= 4) {
      double* arena = arena_t;
      const int32_t K = (int32_t)g.nodes.size();
      const dim3 blocks((uint32_t)((int64_t)K * p.ntiles));
      const bool wide64 = g.num_states == 64;
      switch (g.num_inputs) {
#define X(IV)                                                                                                  \
  case IV:                                                                                                     \
    if (wide64)                                                                                                \
      launch(variable_split_f64<IV, 8>, blocks, dim3(512), 0, s, arena, g.d_in, g.d_out, g.d_c, K, p.ntiles, fresh); \
    else                                                                                                       \
      launch(variable_split_f64<IV, 4>, blocks, dim3(512), 0, s, arena, g.d_in, g.d_out, g.d_c, K, p.ntiles, fresh);  \
    break;
        X(1) X(2) X(3) X(4)
#undef X
      }
      return launch_ok() ? 1 : -1;
    }
    return 0;
  } else {
    float* arena = arena_t;
    const int32_t K = (int32_t)g.nodes.size();
    // 32 or 64 states: the states split over the eight waves of a block, inputs read once
    if ((g.num_states == 32 || g.num_states == 64) && g.num_inputs >= 1 && g.num_inputs <= 4) {
      const dim3 blocks((uint32_t)((int64_t)K * p.ntiles * 2));
      const bool wide64 = g.num_states == 64;
      switch (g.num_inputs) {
#define X(IV)                                                                                                   \
  case IV:                                                                                                      \
    if (wide64)                                                                                                 \
      launch(variable_split_fast<IV, 8>, blocks, dim3(512), 0, s, arena, g.d_in, g.d_out, g.d_c, K, p.ntiles, fresh); \
    else                                                                                                        \
      launch(variable_split_fast<IV, 4>, blocks, dim3(512), 0, s, arena, g.d_in, g.d_out, g.d_c, K, p.ntiles, fresh);  \
    break;
        X(1) X(2) X(3) X(4)
#undef X
      }
      return launch_ok() ? 1 : -1;
    }
    // many states (or a shape without a register-resident instance): stream state by state
    const dim3 grid(wave_blocks((int64_t)K * p.ntiles));
    switch (g.num_inputs) {
      case 1:
        launch(variable_big_fast<1>, grid, dim3(256), 0, s, arena, g.d_in, g.d_out, g.d_c, K, g.num_states,
                           p.ntiles, fresh);
        break;
      case 2:
        launch(variable_big_fast<2>, grid, dim3(256), 0, s, arena, g.d_in, g.d_out, g.d_c, K, g.num_states,
                           p.ntiles, fresh);
        break;
      case 3:
        launch(variable_big_fast<3>, grid, dim3(256), 0, s, arena, g.d_in, g.d_out, g.d_c, K, g.num_states,
                           p.ntiles, fresh);
        break;
      default:
        launch(variable_big_fast<4>, grid, dim3(256), 0, s, arena, g.d_in, g.d_out, g.d_c, K, g.num_states,
                           p.ntiles, fresh);
        break;
    }
    return launch_ok() ? 1 : -1;
  }
}

// w2: the second wide parent of a two-parent family group, nullptr for one-parent groups.
// next_is_last: 0, or what the next iteration's launch will be: 1 a full launch, 2 a reduce launch.
// Defined in dispatch_leaf.h and instantiated for float / double in translation units of their own.
template <typename T>
int launch_leaf_family(Plan& p, FactorGroup& g, FusedFamily& fam, WideNode& w, WideNode* w2, T* arena, const T* cpts,
                       int32_t fresh, int32_t first_iteration, int32_t mode, hipStream_t s, bool with_counts = true,
                       int32_t next_is_last = 0);
extern template int launch_leaf_family<float>(Plan&, FactorGroup&, FusedFamily&, WideNode&, WideNode*, float*, const float*,
                                              int32_t, int32_t, int32_t, hipStream_t, bool, int32_t);
extern template int launch_leaf_family<double>(Plan&, FactorGroup&, FusedFamily&, WideNode&, WideNode*, double*,
                                               const double*, int32_t, int32_t, int32_t, hipStream_t, bool, int32_t);

inline bool any_pending_fast(const Plan& p) {
  FastState* st = fast_state(p);
  if (!st) return false;
  for (auto& kv : st->fused)
    if (kv.second.pending) return true;
  return false;
}

// No kernel but the fused leaf-family ones touches the message arrays of the layout's leaf-family
// candidates (Plan::lazy_candidate): each candidate is in a fused group, and the variable group it
// belongs to is skipped as a whole (a group that also holds other nodes runs the ordinary variable
// kernel over all its members, the fused leaves included).
inline bool candidates_all_fused(const Plan& p) {
  FastState* st = fast_state(p);
  std::vector<char> fused(p.n_nodes, 0);
  if (st)
    for (auto& kv : st->fused)
      for (int32_t j : p.fgroups[kv.first].nodes) fused[j] = 1;
  for (int32_t j = 0; j < p.n_nodes; ++j) {
    if (!p.lazy_candidate[j]) continue;
    if (!fused[j] || !st->vgroup_skip[p.vgroup_of[j]]) return false;
  }
  return true;
}

// Re-run, in full mode, every leaf-family iteration that was only reduced.  Must precede
// anything that reads or changes messages, evidence or CPTs.
inline int complete_pending_fast(Plan& p, hipStream_t s) {
  FastState* st = fast_state(p);
  if (!st) return 0;
  for (auto& kv : st->fused) {
    FusedFamily& fam = kv.second;
    if (!fam.pending) continue;
    FactorGroup& g = p.fgroups[kv.first];
    WideNode* w2 = fam.wide_index2 >= 0 ? &st->wide[fam.wide_index2] : nullptr;
    const int rc = p.dtype == 0
                       ? launch_leaf_family<float>(p, g, fam, st->wide[fam.wide_index], w2, static_cast<float*>(p.arena),
                                                   static_cast<const float*>(p.cpts), fam.pending_fresh,
                                                   fam.pending_first, kLeafFull, s)
                       : launch_leaf_family<double>(p, g, fam, st->wide[fam.wide_index], w2,
                                                    static_cast<double*>(p.arena), static_cast<const double*>(p.cpts),
                                                    fam.pending_fresh, fam.pending_first, kLeafFull, s);
    if (rc < 0) return rc;
  }
  return 0;
}

// Everything of the specialised path's host state that decides which kernels an iteration
// launches and with which arguments, as bytes appended to `key`.
inline void append_state_signature(const Plan& p, std::string& key) {
  FastState* st = fast_state(p);
  if (!st) return;
  auto put = [&](const auto& v) { key.append(reinterpret_cast<const char*>(&v), sizeof v); };
  put(st->ingest_next);
  put(st->evidence_in_caller);
  put(st->states_in_caller);
  put(st->states_rows);
  put(st->ingest_states);
  put(st->ingest_on);
  put(st->ingest_off);
  key.append(st->root_written.begin(), st->root_written.end());
  for (const auto& w : st->wide) {
    put(w.partials_valid);
    put(w.products_cached);
    put(w.zd_read);
    put(w.scaled);  // which of its two buffers is the current one
  }
  for (const auto& kv : st->fused) {
    const FusedFamily& fam = kv.second;
    put(fam.pending);
    put(fam.pending_fresh);
    put(fam.pending_first);
    put(fam.em_ready);
    put(fam.z_form);
    put(fam.src_base);
    put(fam.src_stride);
    put(fam.ll_rows);
    put(fam.slots_used);
    put(fam.em_by_form);
    put(fam.hard_z_mant);
    put(fam.hard_wants_light);
  }
}

inline bool em_is_reduced(const Plan& p, size_t fi) {
  FastState* st = fast_state(p);
  if (!st) return false;
  auto it = st->fused.find((int32_t)fi);
  return it != st->fused.end() && it->second.em_ready;
}

// Log-likelihood partial sums when every fused family's last iteration ran in reduce mode (their
// normalisers exist only as ll_part rows).  Returns the number of partials written, 0 if the
// plan is not in that state, -1 on a launch failure.
inline int ll_partials_reduced(Plan& p, double* partial, int32_t max_blocks, hipStream_t s) {
  FastState* st = fast_state(p);
  if (!st || st->fused.empty()) return 0;
  for (auto& kv : st->fused)
    if (!kv.second.pending) return 0;
  // every family gets an equal share of the partial-sum slots (the kernel is grid-stride, so a
  // launch with fewer blocks than rows is still complete)
  const int32_t share = max_blocks / (int32_t)st->fused.size();
  if (share < 1) return -1;
  int32_t used = 0;
  bool listed_done = false;
  for (auto& kv : st->fused) {
    const FusedFamily& fam = kv.second;
    const int64_t n_listed = listed_done ? 0 : st->n_ll_offs;
    const int64_t rows = n_listed + fam.ll_rows;
    const int32_t blocks = (int32_t)std::min<int64_t>(share, rows * ((p.npad + 255) / 256));
    if (blocks <= 0) return -1;
    if (p.dtype == 0) {
      launch(ll_partial_listed<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(p.arena),
             st->d_ll_offs, n_listed, fam.ll_part, (int64_t)fam.ll_rows, p.N, p.npad, partial + used);
    } else {
      launch(ll_partial_listed<double>, dim3(blocks), dim3(256), 0, s, static_cast<const double*>(p.arena),
             st->d_ll_offs, n_listed, fam.ll_part, (int64_t)fam.ll_rows, p.N, p.npad, partial + used);
    }
    if (!launch_ok()) return -1;
    used += blocks;
    listed_done = true;
  }
  return used;
}

// how the last reduce-mode launch of a fused group left its count sums
inline FormSums form_sums_of(const Plan& p, const FusedFamily& fam) {
  FastState* st = fast_state(p);
  FormSums forms{};
  if (!fam.em_by_form) return forms;
  forms.tab = static_cast<const float*>(fam.z_state);
  std::copy(st->states_rows, st->states_rows + 6, forms.rows.e);
  return forms;
}

// EM counts of a fused group straight from the reduce-mode sums, if they describe the current state.
// listed: other entries to finalise from their chunk partial sums in the same launch (count 0 = none).
inline int finalize_em_fast(Plan& p, size_t fi, double* accum, hipStream_t s, ListedEntries listed = ListedEntries{nullptr, nullptr, 0, 0}) {
  FastState* st = fast_state(p);
  if (!st) return 0;
  auto it = st->fused.find((int32_t)fi);
  if (it == st->fused.end() || !it->second.em_ready) return 0;
  FactorGroup& g = p.fgroups[fi];
  const int64_t items = (int64_t)g.nodes.size() * g.cpt_elems;
  const uint32_t blocks = wave_blocks(items) + (uint32_t)((listed.count + kWavesPerBlock - 1) / kWavesPerBlock);
  const FormSums forms = form_sums_of(p, it->second);
  if (p.dtype == 0) {
    launch(em_finalize_slots<float>, dim3(blocks), dim3(256), 0, s,
           static_cast<const float*>(it->second.em_partial), static_cast<const float*>(p.cpts), g.d_cpt,
           (int32_t)g.nodes.size(), (int32_t)g.cpt_elems, it->second.slots_used, accum, listed, forms);
  } else {
    launch(em_finalize_slots<double>, dim3(blocks), dim3(256), 0, s,
           static_cast<const double*>(it->second.em_partial), static_cast<const double*>(p.cpts), g.d_cpt,
           (int32_t)g.nodes.size(), (int32_t)g.cpt_elems, it->second.slots_used, accum, listed, forms);
  }
  return launch_ok() ? 1 : -1;
}


template <typename T>
int launch_factor_fast(Plan& p, FactorGroup& g, T* arena_t, const T* cpts_t, int32_t fresh, int32_t first_iteration,
                       int32_t last_iteration, int32_t reduce_last, hipStream_t s, int32_t next_is_last = 0) {
  // next_is_last: this is the last but one iteration (told apart further below: 1 = a full launch follows, 2 = a reduce launch)
  if (next_is_last) next_is_last = reduce_last ? 2 : 1;
  FastState* st = fast_state(p);
  const size_t fi = (size_t)(&g - p.fgroups.data());
  if (st) {
    auto it = st->fused.find((int32_t)fi);
    if (it != st->fused.end()) {
      WideNode& w = st->wide[it->second.wide_index];
      const int32_t mode = !last_iteration ? kLeafLoop : (reduce_last ? kLeafReduce : kLeafFull);
      WideNode* w2 = it->second.wide_index2 >= 0 ? &st->wide[it->second.wide_index2] : nullptr;
      // reduce_last == 2: only the log-likelihood sums are wanted from the reduce-mode launch
      return launch_leaf_family<T>(p, g, it->second, w, w2, arena_t, cpts_t, fresh, first_iteration, mode, s,
                                   reduce_last != 2, next_is_last);
    }
  }
  {
    T* arena = arena_t;
    const T* cpts = cpts_t;
    const int32_t K = (int32_t)g.nodes.size();
    const uint32_t blocks = wave_blocks((int64_t)K * p.ntiles);
    if (g.dims.size() == 1) {
      if (st && st->root_written[fi]) return 1;
      launch(factor_root_fast<T>, dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_out, g.d_cpt, K, g.dims[0],
                         p.ntiles);
      if (!launch_ok()) return -1;
      if (st) st->root_written[fi] = 1;
      return 1;
    }
    if (g.dims.size() == 2) {
      const int32_t d0 = g.dims[0], sv = g.dims[1];
      if constexpr (std::is_same<T, float>::value) {
        if ((d0 == 32 || d0 == 64) && (sv == 32 || sv == 64)) {
          // dense contraction on the matrix cores, one workgroup per (node, tile)
          const dim3 grid((uint32_t)((int64_t)K * p.ntiles));
          FactorP1Fn dense = d0 == 64 ? (sv == 64 ? factor_p1_mfma<64, 64> : factor_p1_mfma<64, 32>)
                                      : (sv == 64 ? factor_p1_mfma<32, 64> : factor_p1_mfma<32, 32>);
          launch(dense, grid, dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt, K, p.ntiles);
          return launch_ok() ? 1 : -1;
        }
      } else {
        if ((d0 == 32 || d0 == 64) && (sv == 32 || sv == 64)) {
          const dim3 grid((uint32_t)((int64_t)K * p.ntiles));
          FactorFn<double> dense = d0 == 64 ? (sv == 64 ? factor_p1_mfma_f64<64, 64> : factor_p1_mfma_f64<64, 32>)
                                            : (sv == 64 ? factor_p1_mfma_f64<32, 64> : factor_p1_mfma_f64<32, 32>);
          launch(dense, grid, dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt, K, p.ntiles);
          return launch_ok() ? 1 : -1;
        }
      }
      FactorFn<T> fn = factor_p1_lookup<T>(d0, sv);
      if (!fn) return 0;
      launch(fn, dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt, K, p.ntiles);
      return launch_ok() ? 1 : -1;
    }
    if (g.dims.size() == 3) {
      FactorFn<T> fn = factor_p2_lookup<T>(g.dims[0], g.dims[1], g.dims[2]);
      if (!fn) return 0;
      launch(fn, dim3(blocks), dim3(256), 0, s, arena, cpts, g.d_in, g.d_out, g.d_cpt, K, p.ntiles);
      return launch_ok() ? 1 : -1;
    }
    return 0;
  }
}

template <typename T>
int launch_em_fast(Plan& p, FactorGroup& g, const T* arena, int32_t nchunks, double* partial, hipStream_t s) {
  const int32_t K = (int32_t)g.nodes.size();
  const uint32_t blocks = wave_blocks((int64_t)K * nchunks);
  EmFn<T> fn = nullptr;
  if (g.dims.size() == 1) {
    fn = em_root_lookup<T>(g.dims[0]);
    if (!fn) {  // any other number of states
      launch(em_root_any<T>, dim3(blocks), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N, p.ntiles, nchunks, partial,
             g.dims[0]);
      return launch_ok() ? 1 : -1;
    }
  } else if (g.dims.size() == 2) {
    const int32_t d0 = g.dims[0], sv = g.dims[1];
    if constexpr (std::is_same<T, float>::value) {
      if ((d0 == 32 || d0 == 64) && (sv == 32 || sv == 64)) {
        EmP1Fn dense = d0 == 64 ? (sv == 64 ? em_p1_mfma<64, 64> : em_p1_mfma<64, 32>)
                                : (sv == 64 ? em_p1_mfma<32, 64> : em_p1_mfma<32, 32>);
        launch(dense, dim3((uint32_t)((int64_t)K * nchunks)), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N,
               p.ntiles, nchunks, partial);
        return launch_ok() ? 1 : -1;
      }
    } else {
      if ((d0 == 32 || d0 == 64) && (sv == 32 || sv == 64)) {
        EmFn<double> dense = d0 == 64 ? (sv == 64 ? em_p1_mfma_f64<64, 64> : em_p1_mfma_f64<64, 32>)
                                      : (sv == 64 ? em_p1_mfma_f64<32, 64> : em_p1_mfma_f64<32, 32>);
        launch(dense, dim3((uint32_t)((int64_t)K * nchunks)), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N,
               p.ntiles, nchunks, partial);
        return launch_ok() ? 1 : -1;
      }
    }
    fn = em_p1_lookup<T>(d0, sv);
  } else if (g.dims.size() == 3) {
    fn = em_p2_lookup<T>(g.dims[0], g.dims[1], g.dims[2]);
  }
  if (!fn) return 0;
  launch(fn, dim3(blocks), dim3(256), 0, s, arena, g.d_in, g.d_cpt, K, p.N, p.ntiles, nchunks, partial);
  return launch_ok() ? 1 : -1;
}

}  // namespace bn
