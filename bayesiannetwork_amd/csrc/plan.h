// Host-side plan: topology -> groups, message arena layout, device index tables.
//
// Message layout ("tile-major, state-middle, sample-minor"): a message with S
// states over N observations is stored as [num_tiles][S][TILE] elements, TILE
// consecutive observations per tile.  One wavefront (64 lanes x VEC elements,
// VEC * sizeof(T) = 16 bytes) covers a whole tile row with one 1 KiB load, the
// softmax / leave-one-out arithmetic over S is private to a lane, and no
// kernel on the message path needs a cross-lane reduction.  Observations
// beyond N in the last tile are padding: they carry finite values and are
// masked out of every reduction over observations.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace bn {

constexpr int kMaxArity = 8;        // parents + 1
// Variable nodes with more inputs than this are "wide": their product is built from chunk partials,
// and childless families under them are candidates for the fused leaf-family kernels.
constexpr int32_t kWideFanIn = 24;
constexpr int kMaxStatesGeneric = 256;
// inputs of a wide node that its combine kernel folds in one by one (everything else reaches it as
// chunk partial products)
constexpr int32_t kMaxUncovered = 8;
// Stands for "no array" where an offset is expected.  Real offsets may be negative (an array can
// lie below the arena's base address), so the sign of an offset says nothing.
constexpr int64_t kNoArray = INT64_MIN;

struct EdgeLayout {
  int32_t owner;   // node whose factor is one end
  int32_t var;     // node whose variable is the other end
  int32_t states;  // states of `var`
  int64_t v2f;     // arena element offset of the variable -> factor message
  int64_t f2v;     // arena element offset of the factor -> variable message
};

// One launch unit of the variable pass: nodes with equal (num_inputs, num_states).
struct VariableGroup {
  int32_t num_inputs = 0;
  int32_t num_states = 0;
  std::vector<int32_t> nodes;
  // [K * I] arena offsets; in: message waiting at the variable (f2v of an edge, or the
  // observation array); out: where the matching outgoing message goes (v2f of the edge),
  // -1 for the observation slot whose output nobody reads.
  std::vector<int64_t> in_off, out_off;
  std::vector<int64_t> c_off;  // [K]
  // device copies
  int64_t* d_in = nullptr;
  int64_t* d_out = nullptr;
  int64_t* d_c = nullptr;
};

// One launch unit of the factor pass: nodes with equal CPT shape.
struct FactorGroup {
  std::vector<int32_t> dims;  // [parent dims..., own states]
  int64_t cpt_elems = 0;      // product of dims
  std::vector<int32_t> nodes;
  std::vector<int64_t> in_off, out_off;  // [K * A]: v2f / f2v offsets of each slot's edge
  std::vector<int64_t> cpt_off;          // [K] element offsets into the flat CPT buffer
  int64_t* d_in = nullptr;
  int64_t* d_out = nullptr;
  int64_t* d_cpt = nullptr;
};

struct Plan {
  int dtype = 0;
  int device = 0;
  int64_t N = 0;
  int32_t tile = 0;
  int64_t ntiles = 0;
  int64_t npad = 0;
  size_t elem_size = 4;

  int32_t n_nodes = 0;
  std::vector<int32_t> states;
  std::vector<std::vector<int32_t>> parents, children;
  std::vector<int32_t> observed;          // evidence order
  std::vector<int32_t> observed_slot_of;  // node -> index in `observed`, or -1

  std::vector<EdgeLayout> edges;          // [0, n_nodes): local edges; then parent edges in CSR order
  std::vector<int64_t> parent_edge_base;  // node -> first parent-edge id
  std::vector<int64_t> obs_off;           // per observed index: arena offset of the observation array
  std::vector<int64_t> c_off;             // per node
  std::vector<int64_t> cpt_off;           // per node, into flat CPT buffer
  std::vector<std::vector<int32_t>> cpt_dims;
  int64_t cpt_total = 0;
  int64_t arena_elems = 0;  // elements of the arena proper (allocated at creation)
  int64_t c_base = 0;  // c arrays are contiguous: [n_nodes][npad] starting here
  // Message arrays of leaf-family candidates (childless nodes with one or two wide parents) are laid
  // out after everything else, from lazy_begin on: the fused kernels keep those messages in
  // registers and only a query makes them write the arrays, so their memory is a separate region
  // that is backed on first use (an EM-only run never needs it).
  int64_t lazy_begin = 0, lazy_elems = 0;
  std::vector<char> lazy_candidate;  // per node
  // final offsets (relative to `arena`) of the lazy region: [lazy_lo, lazy_hi)
  int64_t lazy_lo = 0, lazy_hi = 0;
  bool in_lazy(int64_t off) const { return off >= lazy_lo && off < lazy_hi; }
  bool lazy_unmapped() const { return lazy_elems > 0 && !lazy_mapped; }

  std::vector<VariableGroup> vgroups;
  std::vector<FactorGroup> fgroups;
  std::vector<int32_t> vgroup_of, fgroup_of, fgroup_pos;  // node -> group, position in factor group

  bool is_tree = false;
  int32_t diameter = -1;

  // device memory
  void* arena = nullptr;
  // the lazy region: an address range reserved at creation (offsets of its arrays are relative to
  // `arena` like all others), mapped to memory by the first call that reads or writes messages
  void* lazy_base = nullptr;
  size_t lazy_bytes = 0;
  bool lazy_reserved = false;  // address range from hipMemAddressReserve (else an ordinary allocation)
  bool lazy_mapped = false;
  hipMemGenericAllocationHandle_t lazy_handle{};
  void* cpts = nullptr;        // flat CPT snapshot
  double* scratch = nullptr;   // reduction partials
  int64_t scratch_elems = 0;
  void* fast = nullptr;  // state of the specialised float32 kernels (dispatch_fast.h), owned by the plan
  void** d_evidence_ptrs = nullptr;  // device array of evidence source addresses
  std::vector<const void*> evidence_on_device;  // the table d_evidence_ptrs holds (empty = nothing uploaded yet)
  // CPT rows (last axis) of all nodes: start element in the flat CPT buffer and length
  int64_t* d_row_off = nullptr;
  int32_t* d_row_len = nullptr;
  int64_t n_rows = 0;
  int64_t ones_off = -1;  // arena offset of ones_states * npad elements equal to 1 (never written after creation), or -1
  int32_t ones_states = 0;
  void* c_before = nullptr;     // snapshot of the normalisers for bn_normaliser_change (allocated on first use)
  bool c_before_valid = false;  // false: the snapshot counts as all zero
  bool obs_all_pairs = false;  // every observed node has two states
  bool c_stale = false;        // normalisers are to read as zero (reset since the last iteration)
  int64_t* d_obs_off = nullptr;
  int32_t* d_obs_states = nullptr;
  int64_t* d_fill_off = nullptr;    // every message / observation array, for the uniform reset
  int32_t* d_fill_states = nullptr;
  int64_t fill_count = 0;
  int64_t fill_messages = 0;  // the first fill_messages arrays are messages, the rest observation slots
  int32_t fill_max_states = 1;
  int64_t device_bytes = 0;
  bool cpts_set = false;
  // true while every message still has its initial uniform value and the arrays have not
  // been written: the first variable pass then uses 1/S directly instead of reading them
  bool fresh = false;
  // observation slots still owe their uniform reset value (skipped when evidence arrives first)
  bool obs_pending = false;

};

// Builds everything host-side (no HIP calls).  Returns "" or an error message.
std::string compile_plan(Plan& p, int dtype, int64_t num_observations, int32_t num_nodes,
                         const int32_t* num_states, const int32_t* parent_offsets,
                         const int32_t* parent_indices, const int32_t* child_offsets,
                         const int32_t* child_indices, const int32_t* observed_nodes,
                         int32_t num_observed);

}  // namespace bn
