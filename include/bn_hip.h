/*
 * bn_hip.h — C ABI of the MI355X (gfx950) sum-product / EM engine.
 *
 * One shared library, libbn_hip.so, plain pointers and sizes only.  Every
 * function returns 0 on success and a negative code on failure;
 * bn_last_error() then holds a message for the calling thread.  Device
 * pointers are ordinary HIP device addresses (e.g. tensor.data_ptr()); the
 * stream argument is a hipStream_t passed as void* (NULL = default stream).
 * A plan is bound to the HIP device it was created on (calls made while another
 * device is current fail with BN_ERR_STATE), owns all message buffers, and is
 * not thread-safe; the caller keeps input/output buffers
 * alive until the stream has run the call.
 *
 * The reference package has no FFI layer: its seam is the Python ABC
 * IInferenceMachine (inference_machines/abstractions.py:13-33) implemented by
 * spa_v3/spa_inference_machine.py.  Each entry point below names the piece of
 * that implementation it replaces.
 *
 * Layout conventions
 *   - node ids index the network's node order (BayesianNetwork.nodes,
 *     bayesian_network.py:46-48);
 *   - a CPT is row-major [parent_0, ..., parent_{P-1}, states]
 *     (bayesian_network.py:17-23); "flat CPT buffer" = all CPTs concatenated in
 *     node order, in the plan's dtype;
 *   - evidence / marginals handed across this boundary are row-major
 *     [num_observations, ...] exactly as the reference's tensors.
 */
#ifndef BN_HIP_H
#define BN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bn_plan bn_plan;

enum { BN_F32 = 0, BN_F64 = 1 };

enum {
  BN_OK = 0,
  BN_ERR_INVALID = -1, /* bad argument / inconsistent topology */
  BN_ERR_HIP = -2,     /* a HIP runtime call failed */
  BN_ERR_STATE = -3    /* call not valid in the plan's current state */
};

/* ABI version of this header (bumped on any signature change). */
int bn_abi_version(void);

/* Message for the last failure on this thread (never NULL). */
const char* bn_last_error(void);

/*
 * Compile a plan: grouping of variable / factor nodes, message arena,
 * per-group index tables, uniform initial messages.
 * Replaces FactorGraph.__init__ + VariableNodeGroup/FactorNodeGroup.__init__
 * (spa_v3/factor_graph.py:327-424, :15-110, :174-249).
 *
 *   dtype              BN_F32 / BN_F64
 *   num_observations   N >= 1 (the reference maps 0 to 1, factor_graph.py:34-35)
 *   num_states[j]      states of node j
 *   parent_offsets     CSR offsets, num_nodes+1 entries
 *   parent_indices     parents of node j in CPT axis order
 *   child_offsets/child_indices
 *                      children of node j in variable-slot order
 *                      (BayesianNetwork.children_of order)
 *   observed_nodes     node ids in evidence order
 */
int bn_plan_create(int dtype, int64_t num_observations, int32_t num_nodes, const int32_t* num_states,
                   const int32_t* parent_offsets, const int32_t* parent_indices,
                   const int32_t* child_offsets, const int32_t* child_indices,
                   const int32_t* observed_nodes, int32_t num_observed, bn_plan** out_plan);

int bn_plan_destroy(bn_plan* plan);

/*
 * Topology facts computed by the plan compiler; replace nx.is_tree(network.G)
 * and nx.diameter(factor_graph.G) (spa_v3/spa_inference_machine.py:41-46).
 * diameter is -1 when the factor graph is not connected.
 */
int bn_plan_topology(const bn_plan* plan, int32_t* is_tree, int32_t* factor_graph_diameter);

/* Total element count of the flat CPT buffer / of the EM statistics buffer. */
int64_t bn_plan_cpt_elements(const bn_plan* plan);

/* Bytes of device memory the plan holds (message arena + tables + scratch). */
int64_t bn_plan_device_bytes(const bn_plan* plan);

/*
 * Bytes of message arrays that have an address but no memory yet.  Where a wide node's childless
 * families run on the fused kernels, their messages live in registers and the arrays a query
 * would read (FactorNodeGroup / VariableNodeGroup buffers of those families,
 * spa_v3/factor_graph.py:48-60, :197-213) are only backed by memory when the first call that reads
 * or writes messages arrives (bn_iterate, marginals, bn_get_message, bn_em_accumulate); a run of
 * bn_em_statistics* calls never needs them.  0 once they are backed (then counted in
 * bn_plan_device_bytes) or when the plan has no such arrays.
 */
int64_t bn_plan_unmapped_bytes(const bn_plan* plan);

/*
 * Back those arrays with memory now (what the first call that needs them would do by itself), for
 * callers that set a plan up where they may allocate and use it where they may not
 * (bn_em_statistics_static).  No effect when they are backed already.  No counterpart in the
 * reference, whose buffers all exist from construction (spa_v3/factor_graph.py:48-60, :197-213).
 */
int bn_plan_map_message_arrays(bn_plan* plan);

/*
 * Small batches are bound by the chain of dependent kernel launches of a call (the reference's own
 * batch size is 100, dev_experiments/mnist/mnist.py:38).  On request, bn_em_statistics* and
 * bn_em_step run their launch sequence as one captured graph from the third call on that finds the
 * plan in the same state with the same arguments (the second such call captures; batches of more
 * than 16384 float32 / 8192 float64 observations are always launched kernel by kernel).  Results are
 * bit-identical either way.  bn_plan_enable_step_graphs(plan, 1) switches it on for a plan, 0 off
 * (the default, unless BN_STEP_GRAPH=1 is in the environment when the plan is created: replaying
 * measured slightly slower than launching on the runtime this was built against; BN_STEP_GRAPH=0
 * forbids it for the process); bn_plan_step_graph_launches counts the graph launches a plan has made.
 */
int bn_plan_enable_step_graphs(bn_plan* plan, int32_t enabled);
int64_t bn_plan_step_graph_launches(const bn_plan* plan);

/*
 * bn_em_step on a small batch (at most 1024 observations) of a star network — one root whose
 * children are all two-state leaves of one CPT shape, the latent-class model of
 * dev_experiments/mnist/mnist.py — runs as ONE persistent launch of at most 128 blocks: the blocks
 * keep their families' evidence in registers for all iterations and meet at a barrier per iteration
 * instead of ending a kernel per pass (the step is otherwise bound by its chain of sixteen
 * dependent dispatches).  Same iterations, same formulas; sums are taken in another order, so
 * results agree with the launch-per-pass route to rounding (2e-6 relative in float32), not bit for
 * bit.  The barriers need all blocks on the GPU at once; their wait is bounded, and a step whose
 * blocks could not meet (a GPU shared with other work) is void as a whole: all blocks agree on it
 * before any parameter is touched, the plan's CPT snapshot stays as it was, new_cpts_dev receives
 * those unchanged parameters and *ll_out_dev a NaN.  The next bn_em_step on the plan notices, counts
 * the failure and launches per pass from then on.  bn_plan_star_step_failures returns that count
 * plus a failure not yet noticed; it reads a flag the kernel writes, so it is exact once the stream
 * has run the step (after a synchronisation), and a caller that must not lose a step checks it then.
 * On by default; bn_plan_enable_star_step(plan, 0) switches a plan back to one launch per pass
 * (BN_STAR_STEP=0 in the environment: the whole process); bn_plan_star_step_launches counts the
 * steps a plan has run this way.
 */
int bn_plan_enable_star_step(bn_plan* plan, int32_t enabled);
int64_t bn_plan_star_step_launches(const bn_plan* plan);
int64_t bn_plan_star_step_failures(const bn_plan* plan);
/*
 * Float evidence that is hard evidence in disguise.  bn_em_step / bn_em_step_statistics look at float
 * evidence of batches of 28 672 observations and more before they start (float32 plans whose observed
 * nodes are all two-state leaves, evidence arrays at equal distances): when the rows of all leaves and
 * observations are two rows only, one whose first entry is the larger and one whose second is - what
 * the reference's one-hot evidence and its clamped variants are (spa_v3/spa_inference_machine.py:115-117
 * enters whatever rows it is given) - the step runs on one state byte per leaf and observation, the
 * two states standing for those two rows: one pass over the evidence writes the bytes, the host reads
 * its verdict, and every launch then reads an eighth of the bytes it would have read.  The first such
 * batch of a plan costs a stream synchronisation inside the call; the batch after one of that kind is
 * taken to be of that kind too (same rows): its step is enqueued on that guess and the verdict is read
 * when everything is enqueued - if the guess was wrong the parameters are put back and the step runs
 * again, with the results of a call that never guessed.  Results agree with the float route to rounding (2e-6 relative), not bit for
 * bit.  Evidence of another kind is taken as before (and the next 63 calls do not look).  On by
 * default; bn_plan_enable_evidence_compaction(plan, 0) or BN_NO_EVIDENCE_COMPACTION=1 in the
 * environment switch it off; bn_plan_evidence_compactions counts the calls that ran this way.
 */
int bn_plan_enable_evidence_compaction(bn_plan* plan, int32_t enabled);
int64_t bn_plan_evidence_compactions(const bn_plan* plan);
/* The decision rule of that kernel's barrier, for tests: 0 = keep waiting, 1 = all blocks have
 * arrived, 2 = the launch was given up; `seen` is the arrival counter's value, `target` the value
 * that means "all arrived".  Host-only, needs no GPU. */
int32_t bn_star_barrier_verdict(uint32_t seen, uint32_t target);

/*
 * Which wide-parent cardinalities run on the fused kernels?  Those kernels keep a parent's states in
 * registers, so every number of states is an instantiation of its own: a fixed list is part of the
 * library, and any other is served by a "cardinality pack", libbn_hip_d<states>.so next to the
 * library (built by `make pack D0=<states>` from the same sources; the packs of the Makefile's PACKS
 * list are built with the library), loaded the first time a plan needs it.  Without either, a node
 * with many children and that many states runs on the unfused kernels (correct, and some hundred
 * times slower at 784 children).  Returns 1 = part of the library, 2 = a pack is present (this loads
 * it), 0 = neither.  Host-only, needs no GPU.  (Has no counterpart in the reference, which treats
 * every shape alike: spa_v3/factor_graph.py:339-395.)
 */
int32_t bn_cardinality_served(int32_t states);
/* The same question for the fused kernel of leaf families with two many-children parents (CPT shape
 * [d0][d1][s]): ten shapes are part of the library, any other is a "shape pack",
 * libbn_hip_p<d0>x<d1>x<s>.so (`make pack2 SHAPE=<d0>x<d1>x<s>`, ten seconds of hipcc; the Makefile's
 * PACKS2 are built with the library).  1 = in the library, 2 = pack present (loaded by this call), 0 = neither:
 * such families run on the unfused kernels. */
int32_t bn_two_parent_shape_served(int32_t d0, int32_t d1, int32_t s);
/* The directory the library looks for packs in (its own). */
const char* bn_cardinality_pack_directory(void);

/*
 * Snapshot the CPTs (device pointer to the flat CPT buffer, plan dtype).
 * Replaces torch.stack(node.cpt ...) in FactorNodeGroup.__init__
 * (spa_v3/factor_graph.py:192).
 */
int bn_set_cpts(bn_plan* plan, const void* cpts_dev, void* stream);

/* Put every message back to uniform 1/S, as after construction
 * (spa_v3/factor_graph.py:48-60, :197-213). */
int bn_reset_messages(bn_plan* plan, void* stream);

/*
 * Copy evidence into the observation slot of every observed node.
 * evidence_dev[i] is the device address of a contiguous
 * [num_observations, num_states(observed_nodes[i])] array in the plan dtype;
 * the array of addresses itself lives in host memory.  Values are copied
 * bit-for-bit.  Replaces FactorGraph.enter_evidence +
 * VariableNodeGroup.set_observations (spa_v3/factor_graph.py:453-473, :166-170).
 */
int bn_enter_evidence(bn_plan* plan, const void* const* evidence_dev, void* stream);

/*
 * Compact hard evidence: states_dev[o * num_observations + n] (uint8) is the observed state of
 * the o-th observed node in observation n.  Writes on_value into that state's row of the node's
 * observation slot and off_value into its other rows: exactly the tensor
 * one_hot(state) * (on_value - off_value) + off_value that Evidence would have carried
 * (inference_machines/evidence.py:8-45; the soft-clamped one-hot evidence of
 * dev_experiments), at one byte per observed node and observation instead of 4 * num_states.
 * A state >= the node's number of states leaves all rows at off_value.
 */
int bn_enter_evidence_states(bn_plan* plan, const uint8_t* states_dev, double on_value, double off_value,
                             void* stream);

/*
 * Run num_iterations flooding iterations: all variable-to-factor updates, then
 * all factor-to-variable updates.  Replaces FactorGraph.iterate,
 * VariableNodeGroup.calculate_outputs and FactorNodeGroup.calculate_outputs
 * (spa_v3/factor_graph.py:446-451, :112-144, :251-298).
 */
int bn_iterate(bn_plan* plan, int32_t num_iterations, void* stream);

/*
 * Sum over nodes and observations of the per-node normalisers c[node, n]
 * written by the last variable pass; divided by num_observations when
 * average != 0.  One double is written to out_dev.  Replaces
 * SpaInferenceMachine.log_likelihood (spa_v3/spa_inference_machine.py:128-137).
 */
int bn_log_likelihood(bn_plan* plan, int32_t average, double* out_dev, void* stream);

/* c[node, :] -> out_dev[num_observations] (plan dtype).  Replaces reading
 * VariableNodeGroup.local_log_likelihoods (spa_v3/factor_graph.py:144). */
int bn_local_log_likelihoods(bn_plan* plan, int32_t node, void* out_dev, void* stream);

/*
 * out_dev[N, S] = (local factor -> variable message) * (variable -> local
 * factor message).  Replaces SpaInferenceMachine._infer_single_node
 * (spa_v3/spa_inference_machine.py:72-81).
 */
int bn_marginal_single(bn_plan* plan, int32_t node, void* out_dev, void* stream);

/*
 * out_dev[N, d_0..d_{P-1}, S] = cpt * product of the factor's input messages
 * (not renormalised).  Replaces SpaInferenceMachine._infer_node_with_parents
 * (spa_v3/spa_inference_machine.py:89-104).
 */
int bn_marginal_family(bn_plan* plan, int32_t node, void* out_dev, void* stream);

/*
 * EM expected counts: for every node the SUM over observations of its family
 * marginal, written as doubles to accum_dev laid out like the flat CPT buffer.
 * Dividing by num_observations gives exactly
 * infer_nodes_with_parents(all nodes)[i].mean(dim=0)
 * (optimizers/em_batch_optimizer.py:61-68, optimizers/em_optimizer.py:48-55)
 * without materialising the [N, ...] tensors.
 */
int bn_em_accumulate(bn_plan* plan, double* accum_dev, void* stream);

/*
 * One EM step's statistics in one call: num_iterations flooding iterations, then the
 * log-likelihood (as bn_log_likelihood) and the expected-count sums (as bn_em_accumulate).
 * Equivalent to bn_iterate + bn_log_likelihood + bn_em_accumulate, but the last iteration may
 * be evaluated on the fly for the statistics without storing its messages; they are brought up
 * to date automatically by the next call that reads or changes messages, evidence or CPTs.
 * accum_dev may be NULL: then only the log-likelihood is produced (inference without an
 * E-step: SpaInferenceMachine.log_likelihood on a machine that has not iterated yet,
 * spa_v3/spa_inference_machine.py:121-137).
 * Replaces the body of EmBatchOptimizer.optimize's step up to the M-step
 * (optimizers/em_batch_optimizer.py:44-48: enter_evidence excluded).
 */
int bn_em_statistics(bn_plan* plan, int32_t num_iterations, int32_t average, double* ll_out_dev, double* accum_dev,
                     void* stream);

/*
 * bn_em_statistics for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): same arguments, same results, same state
 * afterwards, and the call itself allocates and frees nothing.  No counterpart in the reference.
 * The partial sums of its reductions (log-likelihood, expected counts) go to a static device array
 * of 2^20 doubles (8 MiB per GPU, held until the process ends) instead of memory the plan owns;
 * the message state still lives in the plan, which bn_plan_create allocated.  Fails with
 * BN_ERR_INVALID when 1024 + bn_plan_cpt_elements(plan) exceeds the array.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (different streams or threads) race.
 * A plan may still hold message arrays without memory (bn_plan_unmapped_bytes > 0).  A call on a
 * freshly reset plan whose observed families all run on the fused kernels never touches them; any
 * other call would have to back them, which this entry point refuses: it returns BN_ERR_STATE
 * before launching anything (a second call without bn_reset_messages in between, a plan with
 * groups on the unfused kernels).  bn_plan_map_message_arrays at set-up time removes the condition.
 */
int bn_em_statistics_static(bn_plan* plan, int32_t num_iterations, int32_t average, double* ll_out_dev,
                            double* accum_dev, void* stream);

/*
 * bn_enter_evidence followed by bn_em_statistics, as one call: same arguments, same results, same
 * state afterwards (the observation slots hold the evidence).  When nothing has been sent since
 * the last reset, at least three iterations are asked for, num_observations is even, every observed node is a two-state leaf
 * served by a fused leaf-family kernel and the evidence arrays are 32-byte aligned, the first
 * iteration reads the evidence from the caller's arrays and writes the plan's copy on the way,
 * which saves the separate copy pass; otherwise it is literally the two calls.  The evidence
 * arrays are not referenced after the call's work on the stream has finished.
 * Replaces enter_evidence + log_likelihood + _e_step of one optimizer step
 * (optimizers/em_batch_optimizer.py:44-48, optimizers/em_optimizer.py:34-41).
 */
int bn_em_statistics_from(bn_plan* plan, const void* const* evidence_dev, int32_t num_iterations, int32_t average,
                          double* ll_out_dev, double* accum_dev, void* stream);

/*
 * The same for compact hard evidence: bn_enter_evidence_states followed by bn_em_statistics, as
 * one call (arguments of both).  The first iteration reads the state bytes itself and writes the
 * observation slots on the way when, in addition to the conditions above, every leaf of the fused
 * groups is observed; otherwise it is literally the two calls.
 */
int bn_em_statistics_from_states(bn_plan* plan, const uint8_t* states_dev, double on_value, double off_value,
                                 int32_t num_iterations, int32_t average, double* ll_out_dev, double* accum_dev,
                                 void* stream);

/*
 * One whole mini-batch EM step in a single call (EmBatchOptimizer.optimize's loop body,
 * optimizers/em_batch_optimizer.py:37-59: a machine for the batch, enter_evidence, log_likelihood,
 * the E-step's family marginals, _m_step):
 *   bn_reset_messages; evidence as in bn_em_statistics_from (evidence_dev, states_dev == NULL) or
 *   bn_em_statistics_from_states (states_dev with on_value / off_value, evidence_dev == NULL);
 *   num_iterations iterations; *ll_out_dev = log-likelihood (mean if average); count sums into a
 *   buffer of the plan; bn_m_step from them with the plan's current CPT snapshot as the old CPTs
 *   (learning_rate < 0: no blending); the result becomes the plan's snapshot (as bn_set_cpts would
 *   make it) and is written to new_cpts_dev (flat, plan dtype).
 * Afterwards the plan is as after bn_set_cpts + bn_reset_messages: evidence has to be entered again
 * before anything is queried.  Same kernels and results as the separate calls; the point is the
 * number of host calls and launches per step, which is what bounds small batches (see
 * bn_plan_enable_step_graphs: this sequence, M-step included, runs as one captured graph).
 * Residency: the evidence arrays are read by kernels of this call on `stream` until its last launch
 * has run (the plan's own copy would be void after the step, so a step may not make one at all):
 * they must stay unchanged until then, as any stream-ordered input.
 */
int bn_em_step(bn_plan* plan, const void* const* evidence_dev, const uint8_t* states_dev, double on_value,
               double off_value, int32_t num_iterations, int32_t average, double learning_rate, double regularization,
               double* ll_out_dev, void* new_cpts_dev, void* stream);

/*
 * The statistics half of bn_em_step, for callers that add the sums up over several devices before
 * the M-step (the loop body of optimizers/em_batch_optimizer.py:37-59 on every rank up to the
 * E-step's sums, an all-reduce of [count sums, log-likelihood, N] in between, bn_m_step after it):
 *   bn_reset_messages; evidence as in bn_em_step; num_iterations iterations; *ll_out_dev =
 *   log-likelihood (mean if average); expected-count sums (flat, network order, float64) to accum_dev.
 * Afterwards the plan is as after bn_reset_messages (CPT snapshot unchanged): nobody queries the
 * messages of a batch that is only summed, so - as in bn_em_step - the plan keeps no copy of the
 * evidence where its kernels can read the caller's arrays directly.  Same residency rule as
 * bn_em_step; same results as bn_em_statistics_from[_states] on a reset plan.
 */
int bn_em_step_statistics(bn_plan* plan, const void* const* evidence_dev, const uint8_t* states_dev, double on_value,
                          double off_value, int32_t num_iterations, int32_t average, double* ll_out_dev,
                          double* accum_dev, void* stream);

/*
 * M-step for the whole network in one launch.  For every CPT row (last axis) of every node:
 *   target = count_sums_row / sum(count_sums_row)
 *   target = (1 - regularization) * target + regularization / num_states      if regularization != 0
 *   new    = (1 - learning_rate) * old + learning_rate * target               if learning_rate >= 0
 *            target                                                          otherwise
 * computed in double, written in the plan's dtype to new_cpts_dev (flat CPT layout, network
 * order; may alias old_cpts_dev).  count_sums_dev is what bn_em_accumulate / bn_em_statistics
 * wrote (any positive common scale: sums or means); old_cpts_dev is read only when
 * learning_rate >= 0.  Does not touch the plan's own CPT snapshot: pass the result to
 * bn_set_cpts (a new machine does).  Replaces EmBatchOptimizer._m_step
 * (optimizers/em_batch_optimizer.py:70-84) and EmOptimizer._m_step (optimizers/em_optimizer.py:57-63).
 */
int bn_m_step(bn_plan* plan, const double* count_sums_dev, const void* old_cpts_dev, double learning_rate,
              double regularization, void* new_cpts_dev, void* stream);

/*
 * Convergence monitor for loopy networks: the largest |c[node][n] - c_before[node][n]| over all
 * nodes and observations, where c are the normalisers bn_local_log_likelihoods returns and
 * c_before their values at the previous call of this function on the plan (zero after
 * bn_reset_messages / creation).  Written as one double to out_dev.  The reference offers no
 * stopping rule; its loops experiment watches the messages and the sum of the local
 * log-likelihoods per iteration through the callback (dev_experiments/loops/loops.py:93-120),
 * which is what this reduces to one number.  The first call on a plan allocates the c_before array
 * (hipMalloc: it synchronises the device and cannot run under stream capture); later calls only
 * enqueue work on the stream.
 */
int bn_normaliser_change(bn_plan* plan, double* out_dev, void* stream);

/*
 * Inspection: copy one message out as [N, S].
 *   owner      node whose factor is one end of the edge
 *   slot       -1 = edge to the owner's own variable, p >= 0 = edge to its p-th parent,
 *              -2 = the observation slot of `owner` (must be observed; direction ignored)
 *   direction  0 = variable -> factor, 1 = factor -> variable
 * Corresponds to reading FactorNodeGroup/VariableNodeGroup.get_input_tensor
 * (spa_v3/factor_graph.py:156-164, :310-318).
 */
int bn_get_message(bn_plan* plan, int32_t owner, int32_t slot, int32_t direction, void* out_dev,
                   void* stream);

/*
 * Exact inference by enumeration of the joint distribution, for checking sum-product results on
 * small networks (trees, polytrees, loopy networks).  Replaces the reference's brute-force machine
 * (inference_machines/naive/naive_inference_machine.py:11-122): that one materialises
 * p(all nodes) * evidence as an [N, d_1..d_n] tensor; this one sums the nodes that are neither
 * queried nor observed once per query, then walks the observed nodes' joint states per observation.
 * Sums are carried in double for both dtypes.  Limits: 32 nodes, 2^30 joint states, 2^24 joint
 * states over (query + observed) nodes of one query.
 *
 * Unlike the reference, which reshapes each CPT assuming its parents precede it in ascending node
 * order (naive_inference_machine.py:45-55), CPT axes are matched to parents by parent_indices, so
 * any node order gives the right joint.
 */
typedef struct bn_exact bn_exact;

/* Topology arguments as bn_plan_create.  Bound to the current HIP device. */
int bn_exact_create(int dtype, int32_t num_nodes, const int32_t* num_states, const int32_t* parent_offsets,
                    const int32_t* parent_indices, const int32_t* observed_nodes, int32_t num_observed,
                    bn_exact** out_machine);
int bn_exact_destroy(bn_exact* machine);

/* Snapshot the flat CPT buffer (device pointer, machine dtype); the reference builds its joint
 * table at construction (naive_inference_machine.py:30, :37-57). */
int bn_exact_set_cpts(bn_exact* machine, const void* cpts_dev, void* stream);

/*
 * Copy the evidence (addresses as bn_enter_evidence; num_observations may change from call to
 * call) and compute the probability of the evidence c[n] of every observation
 * (naive_inference_machine.py:59-88).  evidence_dev == NULL enters all-ones evidence, the
 * reference's state before enter_evidence (:31): queries then return prior marginals.
 */
int bn_exact_enter_evidence(bn_exact* machine, int64_t num_observations, const void* const* evidence_dev,
                            void* stream);

/*
 * log c[n]: their sum (mean when average != 0) as one double to total_out_dev, and/or each of
 * them, in the machine dtype, to each_out_dev[num_observations]; either may be NULL
 * (naive_inference_machine.py:88, :112-122).
 */
int bn_exact_log_likelihood(bn_exact* machine, int32_t average, double* total_out_dev, void* each_out_dev,
                            void* stream);

/*
 * Posterior over a set of distinct nodes given the evidence:
 * out_dev[N, states(nodes[0]), .., states(nodes[k-1])], machine dtype, normalised by c[n]
 * (naive_inference_machine.py:90-110; the reference returns the axes in ascending node order,
 * which a caller gets by passing the nodes sorted).
 */
int bn_exact_marginal(bn_exact* machine, const int32_t* nodes, int32_t num_query_nodes, void* out_dev,
                      void* stream);

/*
 * Ancestral sampling on the device.  Replaces TorchBayesianNetworkSampler
 * (samplers/torch_sampler.py:10-60), which draws one joint sample per Python loop iteration with one
 * torch.multinomial call per node: here one thread draws one joint sample, walking the requested
 * nodes and their ancestors parents-first (the reference's recursion order, :43-52) and inverting
 * the selected CPT row's cumulative distribution at a counter-based uniform number (Philox-4x32-10:
 * a draw is a function of seed, sample index and node only).  Topology arguments as
 * bn_plan_create; bound to the current HIP device.
 */
typedef struct bn_sampler bn_sampler;
int bn_sampler_create(int dtype, int32_t num_nodes, const int32_t* num_states, const int32_t* parent_offsets,
                      const int32_t* parent_indices, bn_sampler** out_sampler);
int bn_sampler_destroy(bn_sampler* sampler);
/* Element count of the flat CPT buffer the sampler reads (all CPTs in node order). */
int64_t bn_sampler_cpt_elements(const bn_sampler* sampler);
/*
 * The order in which a request's nodes and their ancestors are drawn (parents before children, the
 * reference's recursion order, samplers/torch_sampler.py:43-52): node ids to order_out (room for
 * num_nodes entries), their number as the return value; BN_ERR_INVALID when the parent relation has
 * a cycle or an index is out of range.  Host-only: needs no sampler object and no GPU.
 */
int32_t bn_sampler_walk_order(int32_t num_nodes, const int32_t* parent_offsets, const int32_t* parent_indices,
                              const int32_t* nodes, int32_t num_requested, int32_t* order_out);
/* Bytes of device workspace bn_sampler_sample needs for this request (drawn states of the requested
 * nodes and their ancestors, int32 [nodes walked][num_samples]); -1 on a bad argument. */
int64_t bn_sampler_workspace_bytes(bn_sampler* sampler, const int32_t* nodes, int32_t num_requested, int64_t num_samples);
/*
 * out_dev[num_samples][num_requested] (int32, row-major, the reference's layout, :20-24) = jointly
 * sampled states of nodes[0..num_requested), from the flat CPT buffer cpts_dev (sampler dtype).
 * The call allocates nothing once the request's tables exist (they are rebuilt when `nodes`
 * differs from the previous call's).
 */
int bn_sampler_sample(bn_sampler* sampler, const void* cpts_dev, const int32_t* nodes, int32_t num_requested,
                      int64_t num_samples, uint64_t seed, void* workspace_dev, int32_t* out_dev, void* stream);

/*
 * Per-kernel timing for benchmarks (process-wide switch): while enabled, every kernel any
 * plan launches carries a start/stop hipEvent pair attached to its dispatch (no marker
 * packets are put between kernels).
 * bn_profile_report waits for the recorded events, writes a JSON array
 *   [{"kernel": "...", "launches": n, "total_ms": x, "alg_bytes": b}, ...]
 * (NUL-terminated; "[]" if it does not fit) and clears the records.  alg_bytes is the
 * compulsory traffic of ONE launch: every live input message read once and every used
 * output written once, CPT traffic ignored.  No counterpart in the reference (it has no
 * profiling hooks).
 */
int bn_profile_enable(int32_t enabled);
int bn_profile_report(char* buffer, int64_t capacity);
/*
 * Restrict the timing to kernels whose report label starts with label_prefix (NULL or "" = all
 * kernels again).  A timed dispatch costs a few microseconds more than an untimed one, so a
 * benchmark that wants one kernel's durations from inside its timed region limits the timing to it.
 */
int bn_profile_only(const char* label_prefix);
/*
 * Of the launches that qualify, time only every `every`-th (1 = all of them).  A timed dispatch is
 * fenced off from its neighbours, so timing every launch of a kernel inside a benchmark's timed
 * region slows the region down by a few per cent; a stride that is coprime to the number of such
 * launches per step still samples every kind of launch evenly.
 */
int bn_profile_stride(int32_t every);

#ifdef __cplusplus
}
#endif
#endif /* BN_HIP_H */
