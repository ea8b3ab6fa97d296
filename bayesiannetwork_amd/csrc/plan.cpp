// Plan compiler (host only): validates the topology, lays out the message arena and
// builds the per-group offset tables the kernels index with.
//
// Grouping follows the reference's launch units — variable nodes keyed by
// (num_inputs, num_states), factor nodes keyed by CPT shape
// (spa_v3/factor_graph.py:339-395) — because those keys are exactly what a kernel
// specialisation depends on.  Slot order inside a node is the reference's:
// variable node [child_0 .. child_{C-1}, (observation), local factor]
// (factor_graph.py:75,146-170), factor node [parent_0 .. parent_{P-1}, own variable]
// (factor_graph.py:300-318).
#include "plan.h"

#include <algorithm>
#include <deque>
#include <cstdlib>
#include <map>
#include <sstream>

namespace bn {

namespace {

int32_t tree_diameter(const std::vector<std::vector<int32_t>>& adj) {
  // two breadth-first sweeps; exact on trees
  auto sweep = [&](int32_t start, std::vector<int32_t>& dist) {
    dist.assign(adj.size(), -1);
    std::deque<int32_t> queue{start};
    dist[start] = 0;
    int32_t last = start;
    while (!queue.empty()) {
      int32_t v = queue.front();
      queue.pop_front();
      last = v;
      for (int32_t w : adj[v])
        if (dist[w] < 0) {
          dist[w] = dist[v] + 1;
          queue.push_back(w);
        }
    }
    return last;
  };
  std::vector<int32_t> dist;
  int32_t far = sweep(0, dist);
  int32_t far2 = sweep(far, dist);
  return dist[far2];
}

}  // namespace

std::string compile_plan(Plan& p, int dtype, int64_t num_observations, int32_t num_nodes,
                         const int32_t* num_states, const int32_t* parent_offsets,
                         const int32_t* parent_indices, const int32_t* child_offsets,
                         const int32_t* child_indices, const int32_t* observed_nodes,
                         int32_t num_observed) {
  if (dtype != 0 && dtype != 1) return "dtype must be BN_F32 or BN_F64";
  if (num_observations < 1) return "num_observations must be >= 1";
  if (num_nodes < 1) return "num_nodes must be >= 1";
  if (!num_states || !parent_offsets || !child_offsets) return "null topology array";
  if (num_observed < 0 || (num_observed > 0 && !observed_nodes)) return "bad observed_nodes";
  if (parent_offsets[0] != 0 || child_offsets[0] != 0) return "CSR offsets must start at 0";
  for (int32_t j = 0; j < num_nodes; ++j)  // before anything is indexed with them
    if (parent_offsets[j + 1] < parent_offsets[j] || child_offsets[j + 1] < child_offsets[j])
      return "CSR offsets must be non-decreasing";
  if ((parent_offsets[num_nodes] > 0 && !parent_indices) || (child_offsets[num_nodes] > 0 && !child_indices))
    return "null index array with a non-empty CSR";

  p.dtype = dtype;
  p.elem_size = dtype == 0 ? 4 : 8;
  p.N = num_observations;
  p.tile = dtype == 0 ? 256 : 128;  // 64 lanes x 16 bytes
  p.ntiles = (p.N + p.tile - 1) / p.tile;
  p.npad = p.ntiles * p.tile;
  p.n_nodes = num_nodes;

  p.states.assign(num_states, num_states + num_nodes);
  p.parents.resize(num_nodes);
  p.children.resize(num_nodes);
  for (int32_t j = 0; j < num_nodes; ++j) {
    if (p.states[j] < 1) return "num_states must be >= 1";
    if (parent_offsets[j + 1] < parent_offsets[j] || child_offsets[j + 1] < child_offsets[j])
      return "CSR offsets must be non-decreasing";
    for (int32_t e = parent_offsets[j]; e < parent_offsets[j + 1]; ++e) {
      int32_t q = parent_indices[e];
      if (q < 0 || q >= num_nodes || q == j) return "parent index out of range";
      p.parents[j].push_back(q);
    }
    for (int32_t e = child_offsets[j]; e < child_offsets[j + 1]; ++e) {
      int32_t q = child_indices[e];
      if (q < 0 || q >= num_nodes || q == j) return "child index out of range";
      p.children[j].push_back(q);
    }
    if ((int)p.parents[j].size() + 1 > kMaxArity) {
      std::ostringstream os;
      os << "node " << j << " has " << p.parents[j].size() << " parents; at most " << kMaxArity - 1
         << " are supported";
      return os.str();
    }
  }
  // parents and children must describe the same edge set
  int64_t n_parent_edges = 0, n_child_edges = 0;
  for (int32_t j = 0; j < num_nodes; ++j) {
    n_parent_edges += p.parents[j].size();
    n_child_edges += p.children[j].size();
    for (int32_t q : p.parents[j])
      if (std::find(p.children[q].begin(), p.children[q].end(), j) == p.children[q].end())
        return "parent/child lists disagree";
  }
  if (n_parent_edges != n_child_edges) return "parent/child lists disagree";

  p.observed.assign(observed_nodes, observed_nodes + num_observed);
  p.observed_slot_of.assign(num_nodes, -1);
  for (int32_t i = 0; i < num_observed; ++i) {
    int32_t o = p.observed[i];
    if (o < 0 || o >= num_nodes) return "observed node out of range";
    if (p.observed_slot_of[o] >= 0) return "node observed twice";
    p.observed_slot_of[o] = i;
  }

  // ---- CPT shapes and flat offsets
  p.cpt_dims.resize(num_nodes);
  p.cpt_off.resize(num_nodes);
  p.cpt_total = 0;
  for (int32_t j = 0; j < num_nodes; ++j) {
    int64_t elems = 1;
    for (int32_t q : p.parents[j]) {
      p.cpt_dims[j].push_back(p.states[q]);
      elems *= p.states[q];
    }
    p.cpt_dims[j].push_back(p.states[j]);
    elems *= p.states[j];
    p.cpt_off[j] = p.cpt_total;
    p.cpt_total += elems;
  }

  // ---- arena layout: every array is a whole number of tiles, so offsets stay 1 KiB aligned
  int64_t cursor = 0;
  auto take = [&](int64_t states) {
    int64_t off = cursor;
    cursor += states * p.npad;
    return off;
  };
  p.edges.clear();
  p.parent_edge_base.resize(num_nodes);
  for (int32_t j = 0; j < num_nodes; ++j) p.edges.push_back({j, j, p.states[j], 0, 0});
  for (int32_t j = 0; j < num_nodes; ++j) {
    p.parent_edge_base[j] = (int64_t)p.edges.size();
    for (int32_t q : p.parents[j]) p.edges.push_back({j, q, p.states[q], 0, 0});
  }
  // Leaf-family candidates: childless nodes with one or two parents, all of them wide.  Their
  // message arrays go to the end of the layout (the lazy region, see plan.h).
  p.lazy_candidate.assign(num_nodes, 0);
  const char* lazy_knob = getenv("BN_LAZY_ARRAYS");  // "0": every array in the arena, allocated at creation
  const bool lazy_allowed = !(lazy_knob && lazy_knob[0] == '0');
  for (int32_t j = 0; j < num_nodes && lazy_allowed; ++j) {
    if (!p.children[j].empty() || p.parents[j].empty() || p.parents[j].size() > 2) continue;
    bool wide = true;
    for (int32_t q : p.parents[j]) {
      const int32_t fan_in = (int32_t)p.children[q].size() + 1 + (p.observed_slot_of[q] >= 0 ? 1 : 0);
      wide = wide && fan_in > kWideFanIn;
    }
    p.lazy_candidate[j] = wide ? 1 : 0;
  }
  // factor -> variable messages first, grouped by receiving variable (its fan-in is read
  // together by the variable pass), then variable -> factor grouped by receiving factor.
  std::vector<std::vector<int64_t>> by_var(num_nodes);
  for (int64_t e = 0; e < (int64_t)p.edges.size(); ++e) by_var[p.edges[e].var].push_back(e);
  auto lay_messages = [&](bool lazy) {
    for (int32_t v = 0; v < num_nodes; ++v)
      for (int64_t e : by_var[v])
        if ((p.lazy_candidate[p.edges[e].owner] != 0) == lazy) p.edges[e].f2v = take(p.edges[e].states);
    for (int32_t j = 0; j < num_nodes; ++j) {
      if ((p.lazy_candidate[j] != 0) != lazy) continue;
      for (size_t q = 0; q < p.parents[j].size(); ++q) {
        EdgeLayout& e = p.edges[p.parent_edge_base[j] + q];
        e.v2f = take(e.states);
      }
      p.edges[j].v2f = take(p.states[j]);
    }
  };
  lay_messages(false);
  p.obs_off.resize(num_observed);
  for (int32_t i = 0; i < num_observed; ++i) p.obs_off[i] = take(p.states[p.observed[i]]);
  p.c_base = cursor;
  p.c_off.resize(num_nodes);
  for (int32_t j = 0; j < num_nodes; ++j) p.c_off[j] = take(1);
  // an all-ones array that stands in for the observation of unobserved childless nodes, so that
  // kernels serving observed and unobserved leaves alike read both the same way
  {
    int32_t widest = 0;
    for (int32_t j = 0; j < num_nodes; ++j)
      if (p.children[j].empty() && p.observed_slot_of[j] < 0) widest = std::max(widest, p.states[j]);
    p.ones_off = widest > 0 ? take(widest) : -1;
    p.ones_states = widest;
  }
  p.arena_elems = cursor;
  p.lazy_begin = cursor;
  lay_messages(true);
  p.lazy_elems = cursor - p.lazy_begin;

  // ---- variable groups
  {
    std::map<std::pair<int32_t, int32_t>, std::vector<int32_t>> keyed;
    for (int32_t j = 0; j < num_nodes; ++j) {
      int32_t fan_in = (int32_t)p.children[j].size() + 1 + (p.observed_slot_of[j] >= 0 ? 1 : 0);
      keyed[{fan_in, p.states[j]}].push_back(j);
    }
    p.vgroups.clear();
    p.vgroup_of.assign(num_nodes, -1);
    for (auto& kv : keyed) {
      VariableGroup g;
      g.num_inputs = kv.first.first;
      g.num_states = kv.first.second;
      g.nodes = kv.second;
      for (int32_t j : g.nodes) {
        p.vgroup_of[j] = (int32_t)p.vgroups.size();
        // children slots: the edge (factor of child, variable j) is the child's parent edge
        for (int32_t child : p.children[j]) {
          const auto& cp = p.parents[child];
          int32_t pos = (int32_t)(std::find(cp.begin(), cp.end(), j) - cp.begin());
          const EdgeLayout& e = p.edges[p.parent_edge_base[child] + pos];
          g.in_off.push_back(e.f2v);
          g.out_off.push_back(e.v2f);
        }
        if (p.observed_slot_of[j] >= 0) {
          g.in_off.push_back(p.obs_off[p.observed_slot_of[j]]);
          g.out_off.push_back(kNoArray);
        }
        g.in_off.push_back(p.edges[j].f2v);
        g.out_off.push_back(p.edges[j].v2f);
        g.c_off.push_back(p.c_off[j]);
      }
      p.vgroups.push_back(std::move(g));
    }
  }

  // ---- factor groups
  {
    std::map<std::vector<int32_t>, std::vector<int32_t>> keyed;
    for (int32_t j = 0; j < num_nodes; ++j) keyed[p.cpt_dims[j]].push_back(j);
    p.fgroups.clear();
    p.fgroup_of.assign(num_nodes, -1);
    p.fgroup_pos.assign(num_nodes, -1);
    for (auto& kv : keyed) {
      FactorGroup g;
      g.dims = kv.first;
      g.cpt_elems = 1;
      for (int32_t d : g.dims) g.cpt_elems *= d;
      g.nodes = kv.second;
      for (int32_t j : g.nodes) {
        p.fgroup_of[j] = (int32_t)p.fgroups.size();
        p.fgroup_pos[j] = (int32_t)(g.cpt_off.size());
        for (size_t q = 0; q < p.parents[j].size(); ++q) {
          const EdgeLayout& e = p.edges[p.parent_edge_base[j] + q];
          g.in_off.push_back(e.v2f);
          g.out_off.push_back(e.f2v);
        }
        g.in_off.push_back(p.edges[j].v2f);
        g.out_off.push_back(p.edges[j].f2v);
        g.cpt_off.push_back(p.cpt_off[j]);
      }
      p.fgroups.push_back(std::move(g));
    }
  }

  // ---- topology facts: is the DAG's underlying graph a tree; factor-graph diameter
  {
    std::vector<std::vector<int32_t>> und(num_nodes);
    for (int32_t j = 0; j < num_nodes; ++j)
      for (int32_t q : p.parents[j]) {
        und[j].push_back(q);
        und[q].push_back(j);
      }
    std::vector<char> seen(num_nodes, 0);
    std::vector<int32_t> stack{0};
    seen[0] = 1;
    int32_t reached = 1;
    while (!stack.empty()) {
      int32_t v = stack.back();
      stack.pop_back();
      for (int32_t w : und[v])
        if (!seen[w]) {
          seen[w] = 1;
          ++reached;
          stack.push_back(w);
        }
    }
    p.is_tree = (reached == num_nodes) && (n_parent_edges == num_nodes - 1);
    p.diameter = -1;
    if (p.is_tree) {
      // factor graph: vertex j = variable j, num_nodes + j = factor j
      std::vector<std::vector<int32_t>> adj(2 * (size_t)num_nodes);
      for (const EdgeLayout& e : p.edges) {
        adj[e.var].push_back(num_nodes + e.owner);
        adj[num_nodes + e.owner].push_back(e.var);
      }
      p.diameter = tree_diameter(adj);
    }
  }
  return "";
}

}  // namespace bn
