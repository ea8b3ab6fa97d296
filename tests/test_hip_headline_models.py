"""GPU parity on the benchmarked models at sizes the golden fixtures do not reach, against the CPU
oracle run in float64 on the same inputs (the oracle is pinned to the reference by the fixtures,
tests/test_oracle_golden.py; nothing here has a reference-generated answer of its own):

* the 784-leaf models (one hidden node / the loopy two-feature form) at 1 025 observations of
  DISTINCT rows, soft and hard mixed: five observation tiles, every chunk of families, a ragged last
  tile; through the query route, the one-call statistics route and compact hard evidence;
* the deep tree of 64-state nodes at its benchmarked depth (6 levels, 95 nodes, 24 iterations).

Tolerance (float32): 1e-5 relative + 1e-7 of the tensor's scale on trees; a node's normaliser
c = log sum_s prod_i in_i[s] gets 2e-6 + 2e-7 per input message on top (every leaf family works from
the wide parent's product rounded to float32 once, up to 6e-8 of each state, and that shared
rounding moves all 784 messages to the parent the same way instead of averaging out: 1.2e-4 seen on
the 785-input node with soft evidence, where the reference's own float32 run is off by 4e-4; the
log-likelihood, in which these shifts cancel against the leaves' normalisers, stays within 1e-5
relative).  The loopy model's ten iterations feed rounding back into themselves, so two correct
float32 implementations drift apart: there the yardstick is the oracle run in float32 as well (the
reference's own float32 arithmetic).  Ten iterations amplify the rounding of a few ill-conditioned
observations by four orders of magnitude (errors up to 3e-3 on single rows in either float32 run,
tools/diag_c3b.py), so on that model 99 % of the entries of a tensor must be within ten times the
bound above, and the largest error within ten times the largest error of the float32 oracle.
float64: 1e-9 relative.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_cpt(gen, shape):
    p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.02
    return p / p.sum(dim=-1, keepdim=True)


def _leaf_model(kind, dtype, device):
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device=device, dtype=dtype)
    gen = torch.Generator().manual_seed(11)

    def cpt(shape):
        return _rand_cpt(gen, shape).to(ts.dtype).to(ts.device)

    # "c3a" + digits: the tree model with that many latent classes instead of 10
    classes = int(kind[3:]) if kind[3:].isdigit() else 10
    q = Node(cpt((classes,)), name="Q")
    builder = BayesianNetworkBuilder().add_node(q)
    ys = []
    if kind == "c3b":
        f1, f2 = Node(cpt((10, 2)), name="F1"), Node(cpt((10, 2)), name="F2")
        builder.add_node(f1, parents=q).add_node(f2, parents=q)
        for i in range(784):
            ys.append(Node(cpt((2, 2, 2)), name=f"Y{i}"))
            builder.add_node(ys[-1], parents=[f1, f2])
    else:
        for i in range(784):
            ys.append(Node(cpt((classes, 2)), name=f"Y{i}"))
            builder.add_node(ys[-1], parents=q)
    return ts, builder.build(), ys


def _mixed_evidence(n, count, seed, torch_dtype):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(count):
        soft = torch.rand((n, 2), generator=gen, dtype=torch.float64) + 1e-3
        soft = soft / soft.sum(-1, keepdim=True)
        x = (torch.rand((n,), generator=gen) < 0.3).double() * 0.999999 + 5e-7
        hard = torch.stack([1 - x, x], dim=1)
        pick = (torch.rand((n, 1), generator=gen) < 0.6).double()
        out.append((pick * hard + (1 - pick) * soft).to(torch_dtype))
    return out


def _check(got, want, what, f32, slack=1.0, extra_atol=0.0, yardstick=None):
    """yardstick: the same quantity from the float32 run of the oracle (loopy models only)."""
    got = got.detach().cpu().double().numpy()
    want = want.detach().double().numpy()
    assert np.isfinite(got).all(), what
    if f32:
        scale = float(np.max(np.abs(want))) if want.size else 1.0
        bound = slack * (1e-5 * np.abs(want) + 1e-7 * scale + extra_atol)
    else:
        bound = slack * (1e-9 * np.abs(want) + 1e-12)
    ratio = np.abs(got - want) / bound
    worst = float(np.max(ratio))
    if f32 and yardstick is not None:
        theirs = float(np.max(np.abs(yardstick.detach().double().numpy() - want) / bound))
        if ratio.size >= 1000:  # per-observation tensors; sums over observations only have the second rule
            typical = float(np.quantile(ratio, 0.99))
            assert typical <= 1.0, f"{what}: 99 % of the entries are within {typical:.2f}x the bound"
        assert worst <= max(1.0, 10.0 * theirs), f"{what}: {worst:.2f}x the bound (float32 oracle: {theirs:.2f}x)"
        return worst
    assert worst <= 1.0, f"{what}: {worst:.2f}x the bound"
    return worst


def _oracle_for(kind, network_dev, ys_dev, dtype, n, iterations, loops, evidence, oracle_dtype=torch.float64):
    from oracle.spa_oracle import SpaOracle

    _, network64, ys64 = _leaf_model(kind, "float64", "cpu")
    for node64, node in zip(network64.nodes, network_dev.nodes):  # same rounded CPT values as the device network
        node64.cpt = node.cpt.detach().cpu().double()
    oracle = SpaOracle(network64, ys64, n, dtype=oracle_dtype, num_iterations=iterations, allow_loops=loops)
    oracle.enter_evidence([e.to(oracle_dtype) for e in evidence])
    return oracle


class _NoYardstick:
    """Stands in for the float32 oracle where none is needed: every query answers None."""

    def __getattr__(self, name):
        return lambda *a, **k: None


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("kind", ["c3a", "c3b", "c3a6", "c3a7", "c3a8", "c3a9", "c3a12", "c3a15", "c3a16", "c3a20"])
def test_784_leaf_models_on_distinct_rows_across_tiles(kind, dtype):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    n = 1025
    loops, iterations = (True, 10) if kind == "c3b" else (False, None)
    f32 = dtype == "float32"
    slack = 10.0 if loops else 1.0
    ts, network, ys = _leaf_model(kind, dtype, "cuda")
    evidence = _mixed_evidence(n, len(ys), 5, ts.dtype)
    oracle = _oracle_for(kind, network, ys, dtype, n, iterations, loops, evidence)
    o32 = (_oracle_for(kind, network, ys, dtype, n, iterations, loops, evidence, torch.float32)
           if loops and f32 else _NoYardstick())
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops)
    nodes = list(network.nodes)
    picks = sorted({0, 1, 2, 3, len(nodes) // 2, len(nodes) - 1})

    def family_sum(o, j):
        fam = o.infer_family(j)
        return None if fam is None else fam.sum(dim=0)

    def ll_ok(got):
        limit = slack * (1e-5 if f32 else 1e-9) * abs(want_ll)
        if loops and f32:
            limit = max(limit, 10.0 * abs(o32.log_likelihood(False) - want_ll))
        return abs(got - want_ll) <= limit

    # query route: bn_iterate, then marginals / normalisers / log-likelihood / counts
    machine = HipSpaInferenceMachine(settings, network, ys, n)
    machine.enter_evidence(Evidence([e.cuda() for e in evidence], ts))
    for j in picks:
        _check(machine.infer_single_nodes([nodes[j]])[0], oracle.infer_single(j), f"single {nodes[j].name}", f32, slack,
               yardstick=o32.infer_single(j))
        _check(machine.infer_nodes_with_parents([nodes[j]])[0], oracle.infer_family(j), f"family {nodes[j].name}", f32, slack,
               yardstick=o32.infer_family(j))
    observed = set(ys)
    for j in picks:
        inputs = len(list(network.children_of(nodes[j]))) + 1 + (nodes[j] in observed)
        g, k = oracle.v_of[j]
        c32 = None
        if loops and f32:
            o32.log_likelihood(False)  # makes the float32 oracle iterate
            c32 = o32.vgroups[g]["c"][k]
        _check(machine.local_log_likelihoods(nodes[j]), oracle.vgroups[g]["c"][k], f"c {nodes[j].name}", f32, slack,
               extra_atol=2e-6 + 2e-7 * inputs, yardstick=c32)
    want_ll = oracle.log_likelihood(False)
    assert ll_ok(machine.log_likelihood())
    want_counts = [oracle.infer_family(j).sum(dim=0) for j in range(len(nodes))]
    counts32 = [family_sum(o32, j) for j in range(len(nodes))]
    for j, got in enumerate(machine.split_counts(machine.expected_count_sums())):
        _check(got, want_counts[j], f"counts {nodes[j].name}", f32, slack, yardstick=counts32[j])

    # one-call statistics route (evidence read by the first iteration where the model allows it;
    # 1 025 is odd, 1 026 below is even: both sides of that condition)
    one_call = HipSpaInferenceMachine(settings, network, ys, n)
    ll, flat = one_call.em_statistics(Evidence([e.cuda() for e in evidence], ts))
    assert ll_ok(ll.item())
    for j, got in enumerate(one_call.split_counts(flat)):
        _check(got, want_counts[j], f"counts via statistics {nodes[j].name}", f32, slack, yardstick=counts32[j])
    # and the queries made afterwards see the completed last iteration
    _check(one_call.infer_single_nodes([nodes[0]])[0], oracle.infer_single(0), "single Q after statistics", f32, slack,
           yardstick=o32.infer_single(0))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("kind", ["c3a", "c3b", "c3a7", "c3a8", "c3a16"])
def test_784_leaf_models_statistics_from_float_and_hard_evidence(kind, dtype):
    """An even number of observations, hard rows only: the float route ingests the caller's arrays in
    its first iteration, the uint8 route the state bytes; both must equal the oracle on the tensors
    the hard evidence stands for."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    n = 1026
    loops, iterations = (True, 10) if kind == "c3b" else (False, None)
    f32 = dtype == "float32"
    slack = 10.0 if loops else 1.0
    ts, network, ys = _leaf_model(kind, dtype, "cuda")
    gen = torch.Generator().manual_seed(9)
    states = (torch.rand((len(ys), n), generator=gen) < 0.25).to(torch.uint8)
    hard = HardEvidence(states.cuda(), [2] * len(ys), ts, on_value=0.999999 + 5e-7, off_value=5e-7)
    tensors = [t.clone() for t in hard.data]
    oracle = _oracle_for(kind, network, ys, dtype, n, iterations, loops, [t.cpu() for t in tensors])
    o32 = (_oracle_for(kind, network, ys, dtype, n, iterations, loops, [t.cpu() for t in tensors], torch.float32)
           if loops and f32 else _NoYardstick())
    want_ll = oracle.log_likelihood(False)
    nodes = list(network.nodes)
    want_counts = [oracle.infer_family(j).sum(dim=0) for j in range(len(nodes))]
    counts32 = [None if o32.infer_family(j) is None else o32.infer_family(j).sum(dim=0) for j in range(len(nodes))]
    ll_limit = slack * (1e-5 if f32 else 1e-9) * abs(want_ll)
    if loops and f32:
        ll_limit = max(ll_limit, 10.0 * abs(o32.log_likelihood(False) - want_ll))
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops)
    for route, evidence in (("float", Evidence(tensors, ts)), ("uint8", hard)):
        machine = HipSpaInferenceMachine(settings, network, ys, n)
        ll, flat = machine.em_statistics(evidence)
        assert abs(ll.item() - want_ll) <= ll_limit, route
        for j, got in enumerate(machine.split_counts(flat)):
            _check(got, want_counts[j], f"{route}: counts {nodes[j].name}", f32, slack, yardstick=counts32[j])
        for node, want in zip(ys[:3], tensors[:3]):
            assert torch.equal(machine.observation(node), want), f"{route}: observation slot of {node.name}"
        _check(machine.infer_single_nodes([nodes[0]])[0], oracle.infer_single(0), f"{route}: single Q", f32, slack,
               yardstick=o32.infer_single(0))


def _deep_tree(dtype, device, depth=6, states=64):
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device=device, dtype=dtype)
    gen = torch.Generator().manual_seed(3)

    def cpt(shape):
        return _rand_cpt(gen, shape).to(ts.dtype).to(ts.device)

    root = Node(cpt((states,)), name="H0")
    builder = BayesianNetworkBuilder().add_node(root)
    frontier, count = [root], 1
    for _ in range(depth - 1):
        nxt = []
        for parent in frontier:
            for _ in range(2):
                node = Node(cpt((states, states)), name=f"H{count}")
                count += 1
                builder.add_node(node, parents=parent)
                nxt.append(node)
        frontier = nxt
    leaves = []
    for i, parent in enumerate(frontier):
        leaves.append(Node(cpt((states, states)), name=f"Y{i}"))
        builder.add_node(leaves[-1], parents=parent)
    return ts, builder.build(), leaves


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_deep_tree_of_64_state_nodes_at_benchmark_depth(dtype):
    """63 hidden + 32 observed 64-state nodes, 24 iterations (the factor-graph diameter): the dense
    matrix-core factor pass, the split-state variable pass and the count GEMM on the shape that is
    benchmarked, 300 observations (two tiles, the second ragged)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from oracle.spa_oracle import SpaOracle

    n = 300
    f32 = dtype == "float32"
    ts, network, leaves = _deep_tree(dtype, "cuda")
    gen = torch.Generator().manual_seed(4)
    evidence = []
    for _ in leaves:
        idx = torch.randint(0, 64, (n,), generator=gen)
        hard = torch.nn.functional.one_hot(idx, 64).double() * (1 - 1e-6) + 5e-7
        soft = torch.rand((n, 64), generator=gen, dtype=torch.float64) + 1e-3
        pick = (torch.rand((n, 1), generator=gen) < 0.5).double()
        evidence.append((pick * hard + (1 - pick) * soft / soft.sum(-1, keepdim=True)).to(ts.dtype))
    machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), network, leaves, n)
    assert machine.num_iterations == 24
    machine.enter_evidence(Evidence([e.cuda() for e in evidence], ts))

    _, network64, leaves64 = _deep_tree("float64", "cpu")
    for node64, node in zip(network64.nodes, network.nodes):
        node64.cpt = node.cpt.detach().cpu().double()
    oracle = SpaOracle(network64, leaves64, n, dtype=torch.float64)
    assert oracle.num_iterations == 24
    oracle.enter_evidence([e.double() for e in evidence])

    nodes = list(network.nodes)
    index = {node: j for j, node in enumerate(nodes)}
    want_ll = oracle.log_likelihood(False)
    assert abs(machine.log_likelihood() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    by_name = {node.name: node for node in nodes}
    for name in ("H0", "H5", "H40", "Y0", "Y31"):  # root, inner nodes, first and last leaf
        j = index[by_name[name]]
        _check(machine.infer_single_nodes([nodes[j]])[0], oracle.infer_single(j), f"single {name}", f32)
    for name in ("H5", "Y31"):
        j = index[by_name[name]]
        _check(machine.infer_nodes_with_parents([nodes[j]])[0], oracle.infer_family(j), f"family {name}", f32)
    counts = machine.split_counts(machine.expected_count_sums())
    second = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), network, leaves, n)
    ll2, flat2 = second.em_statistics(Evidence([e.cuda() for e in evidence], ts))
    assert abs(ll2.item() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    counts2 = second.split_counts(flat2)
    for name in ("H0", "H1", "H5", "H40", "Y0", "Y31"):
        j = index[by_name[name]]
        want = oracle.infer_family(j).sum(dim=0)
        _check(counts[j], want, f"counts {name}", f32)
        _check(counts2[j], want, f"counts via statistics {name}", f32)


def _mixed_tree(device, dtype="float32"):
    """A tree of 32- and 64-state one-parent families whose variables receive at most two other
    messages: every (parent states, own states) combination of the dense factor kernels, a chain
    node with a single child, an observed inner node and observed leaves."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device=device, dtype=dtype)
    gen = torch.Generator().manual_seed(17)
    made = {}
    builder = BayesianNetworkBuilder()

    def add(name, states, parent=None):
        shape = (states,) if parent is None else (made[parent].num_states, states)
        made[name] = Node(_rand_cpt(gen, shape).to(ts.dtype).to(ts.device), name=name)
        builder.add_node(made[name], parents=None if parent is None else made[parent])

    add("R", 64)
    add("A", 32, "R"); add("B", 64, "R")          # 64x32, 64x64
    add("C", 32, "A"); add("D", 64, "A")          # 32x32, 32x64
    add("E", 64, "B")                             # B: one child only
    add("F", 32, "E"); add("G", 64, "E")
    add("O", 64, "C")                             # observed inner node with a child of its own
    add("Y0", 32, "O"); add("Y1", 64, "C"); add("Y2", 32, "D"); add("Y3", 64, "D")
    add("Y4", 64, "F"); add("Y5", 32, "G"); add("Y6", 32, "G")
    observed = [made[k] for k in ("O", "Y0", "Y1", "Y2", "Y3", "Y4", "Y5", "Y6")]
    return ts, builder.build(), observed


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("iterations", [4, 5, 6, 9, None])
def test_iterations_of_a_dense_tree_in_one_launch_per_factor_group(iterations, dtype):
    """Trees of dense one-parent families run their middle iterations as one launch per factor group
    that reads and writes factor -> variable messages only.  Iteration counts below the diameter
    leave the messages unconverged, so every message, normaliser and marginal depends on each
    iteration having been carried out exactly as a variable pass followed by a factor pass;
    compared with the float64 oracle after the same number of iterations (1e-5 relative in float32,
    1e-9 in float64), and with
    the same machine driven one iteration per call (the per-call route never fuses)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from oracle.spa_oracle import SpaOracle

    n = 333
    f32 = dtype == "float32"
    ts, network, observed = _mixed_tree("cuda", dtype)
    gen = torch.Generator().manual_seed(23)
    evidence = []
    for node in observed:
        s = node.num_states
        idx = torch.randint(0, s, (n,), generator=gen)
        hard = torch.nn.functional.one_hot(idx, s).double() * (1 - 1e-6) + 1e-6 / s
        soft = torch.rand((n, s), generator=gen, dtype=torch.float64) + 1e-3
        pick = (torch.rand((n, 1), generator=gen) < 0.5).double()
        evidence.append((pick * hard + (1 - pick) * soft / soft.sum(-1, keepdim=True)).to(ts.dtype))
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations)
    machine = HipSpaInferenceMachine(settings, network, observed, n)
    machine.enter_evidence(Evidence([e.cuda() for e in evidence], ts))
    # the route under test is the one taken: all but the last one or two iterations
    from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import enable_profiling, profile_report

    enable_profiling(True)
    try:
        profile_report()
        machine.log_likelihood()
        report = profile_report()
    finally:
        enable_profiling(False)
    fused = sum(r["launches"] for r in report if r["kernel"].startswith("tree_iteration"))
    groups = len({r["kernel"] for r in report if r["kernel"].startswith("factor_pass[") and "x" in r["kernel"]})
    fresh = sum(r["launches"] for r in report if r["kernel"].startswith("tree_iteration_fresh"))
    assert groups == 4 and fresh == groups and fused - fresh == groups * ((machine.num_iterations - 2) & ~1), report

    _, network64, observed64 = _mixed_tree("cpu", "float64")
    for node64, node in zip(network64.nodes, network.nodes):
        node64.cpt = node.cpt.detach().cpu().double()
    oracle = SpaOracle(network64, observed64, n, dtype=torch.float64, num_iterations=iterations)
    oracle.enter_evidence([e.double() for e in evidence])
    assert machine.num_iterations == oracle.num_iterations

    nodes = list(network.nodes)
    index = {node: j for j, node in enumerate(nodes)}
    want_ll = oracle.log_likelihood(False)
    assert abs(machine.log_likelihood() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    for node in nodes:
        j = index[node]
        _check(machine.infer_single_nodes([node])[0], oracle.infer_single(j), f"single {node.name}", f32)
        _check(machine.infer_nodes_with_parents([node])[0], oracle.infer_family(j), f"family {node.name}", f32)
        for owner in [node] + list(network.children_of(node)):
            for kind, fn in (("f2v", oracle.message_f2v), ("v2f", oracle.message_v2f)):
                _check(machine.message(node, owner, kind), fn(j, index[owner]),
                       f"{kind} {node.name} / factor of {owner.name}", f32)

    # one iteration per call: the callback route
    seen = []
    stepwise = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=iterations, callback=lambda *a: seen.append(1)),
        network, observed, n)
    stepwise.enter_evidence(Evidence([e.cuda() for e in evidence], ts))
    assert abs(stepwise.log_likelihood() - machine.log_likelihood()) <= (2e-6 if f32 else 1e-12) * abs(want_ll)
    assert len(seen) == machine.num_iterations
    for node in nodes:
        for owner in [node] + list(network.children_of(node)):
            a = machine.message(node, owner, "f2v").double()
            b = stepwise.message(node, owner, "f2v").double()
            assert torch.allclose(a, b, rtol=2e-5 if f32 else 1e-11, atol=1e-30), (node.name, owner.name)

    want_counts = [oracle.infer_family(j).sum(dim=0) for j in range(len(nodes))]

    # new evidence on the same machine: the iterations continue from the stored messages (no fresh
    # first iteration), in the machine and in the oracle alike
    again = []
    for node in observed:
        idx = torch.randint(0, node.num_states, (n,), generator=gen)
        again.append((torch.nn.functional.one_hot(idx, node.num_states).double() * (1 - 1e-6)
                      + 1e-6 / node.num_states).to(ts.dtype))
    machine.enter_evidence(Evidence([e.cuda() for e in again], ts))
    oracle.enter_evidence([e.double() for e in again])
    want_again = oracle.log_likelihood(False)
    assert abs(want_again - want_ll) > 1e-3 * abs(want_ll)  # a different answer, a hundred tolerances away
    assert abs(machine.log_likelihood() - want_again) <= (1e-5 if f32 else 1e-9) * abs(want_again)
    for node in nodes[:4]:
        _check(machine.infer_single_nodes([node])[0], oracle.infer_single(index[node]), f"again: single {node.name}", f32)

    # the one-call statistics route (last iteration in reduce mode) on a second machine
    second = HipSpaInferenceMachine(settings, network, observed, n)
    ll2, flat2 = second.em_statistics(Evidence([e.cuda() for e in evidence], ts))
    assert abs(ll2.item() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    for j, got in enumerate(second.split_counts(flat2)):
        _check(got, want_counts[j], f"counts {nodes[j].name}", f32)


@pytest.mark.parametrize("n", [1026, 16386])
@pytest.mark.parametrize("layout", ["slices of one buffer", "separate arrays"])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("kind", ["c3a", "c3a7", "c3a16", "c3a20"])
def test_whole_step_calls_on_several_tiles_equal_the_separate_calls(kind, dtype, layout, n):
    """``em_step`` and ``em_step_statistics`` on the 784-leaf model at 1 026 and 16 386 distinct soft +
    hard rows (five observation tiles, where the waves of a block share a chunk of families in every
    loop launch, and 65 tiles, where only the later loop launches do; a ragged last tile; evidence
    arrays as equally spaced slices of one buffer,
    which the engine then reads where they lie in every launch, and as separately allocated arrays,
    which it copies into the plan): the same log-likelihood, count sums and new parameters, bit for
    bit, as ``em_statistics(evidence)`` + ``m_step`` on a machine of its own, and the same answers
    when the machines are queried afterwards."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel

    ts, network, ys = _leaf_model(kind, dtype, "cuda")
    rows = _mixed_evidence(n, len(ys), 21, ts.dtype)
    if layout == "slices of one buffer":
        tensors = list(torch.stack(rows).cuda().unbind(0))
    else:
        tensors = [r.cuda().clone() for r in rows]
    settings = SpaInferenceMachineSettings(ts, True)
    model = FlatModel.of(network)
    model.assign(model.gather())
    old = model.gather().clone()

    def machine_for():
        machine = HipSpaInferenceMachine(settings, network, ys, n)
        machine.enable_star_step(False)
        return machine

    plain = machine_for()
    want_ll, want_counts = plain.em_statistics(Evidence(tensors, ts))
    want_new = plain.m_step(want_counts, old, 0.2, 0.01)

    summed = machine_for()
    packed = torch.full((want_counts.numel() + 2,), -1.0, dtype=torch.float64, device="cuda")
    ll, counts = summed.em_step_statistics(Evidence(tensors, ts), out=packed)
    assert torch.equal(ll, want_ll) and torch.equal(counts, want_counts)
    assert counts.data_ptr() == packed.data_ptr() and packed[-1].item() == -1.0
    q = list(network.nodes)[0]
    for a, b in zip(summed.infer_single_nodes([q, ys[5]]), plain.infer_single_nodes([q, ys[5]])):
        assert torch.equal(a, b)
    assert torch.equal(summed.observation(ys[700]), tensors[700])

    stepped = machine_for()
    ll, new = stepped.em_step(Evidence(tensors, ts), 0.2, 0.01)
    assert torch.equal(ll, want_ll) and torch.equal(new, want_new)
    # asked after its step, the machine answers for the batch under the NEW parameters
    model.assign(new)
    fresh = machine_for()
    fresh.enter_evidence(Evidence(tensors, ts))
    for a, b in zip(stepped.infer_nodes_with_parents([ys[3]]), fresh.infer_nodes_with_parents([ys[3]])):
        assert torch.equal(a, b)
    assert stepped.log_likelihood() == fresh.log_likelihood()


@pytest.mark.parametrize("n", [1026, 16386])
@pytest.mark.parametrize("kind", ["c3a", "c3a7", "c3a16", "c3a20"])
def test_whole_step_on_state_bytes_read_in_place(kind, n):
    """``em_step`` / ``em_step_statistics`` on compact hard evidence (float32 plans): every launch reads
    the caller's state bytes where they lie and looks q = cpt . ev and 1 / q up in per-family tables
    (kernels_hard.h).  Against the same step on the float tensors the state bytes stand for: the tables
    are built with the float kernels' operations, so the routes differ only in how they cut the sums
    over the observations (2e-6 relative on the log-likelihood and the count sums, the tolerance of
    the one-launch step's agreement test); a state byte beyond the leaf's states counts as "both
    rows at the off value"; and the machine answers queries afterwards as the float route's does."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel

    ts, network, ys = _leaf_model(kind, "float32", "cuda")
    gen = torch.Generator().manual_seed(12)
    states = (torch.rand((len(ys), n), generator=gen) < 0.3).to(torch.uint8)
    hard = HardEvidence(states.cuda(), [2] * len(ys), ts, on_value=0.999999 + 5e-7, off_value=5e-7)
    # a byte beyond the leaf's states (the constructor refuses them; the C ABI reads them as "both rows
    # at the off value", which is what the expansion below gives)
    hard.states[5, 7] = 2
    hard.states[700, n - 1] = 255
    tensors = [t.clone() for t in hard.data]
    settings = SpaInferenceMachineSettings(ts, True)
    model = FlatModel.of(network)
    model.assign(model.gather())

    def machine_for():
        machine = HipSpaInferenceMachine(settings, network, ys, n)
        machine.enable_star_step(False)
        return machine

    ll_f, counts_f = machine_for().em_step_statistics(Evidence(tensors, ts))
    ll_h, counts_h = machine_for().em_step_statistics(hard)
    assert abs(ll_h.item() - ll_f.item()) <= 2e-6 * abs(ll_f.item())
    np.testing.assert_allclose(counts_h.cpu().numpy(), counts_f.cpu().numpy(), rtol=2e-6, atol=2e-6)

    by_float = machine_for()
    ll_a, new_a = by_float.em_step(Evidence(tensors, ts), 0.2, 0.01)
    new_a = new_a.clone()
    model.assign(model.gather())
    by_bytes = machine_for()
    ll_b, new_b = by_bytes.em_step(hard, 0.2, 0.01)
    assert abs(ll_b.item() - ll_a.item()) <= 2e-6 * abs(ll_a.item())
    np.testing.assert_allclose(new_b.cpu().numpy(), new_a.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # asked after its step, the machine enters the batch again and answers under the new parameters
    model.assign(new_b)
    fresh = machine_for()
    fresh.enter_evidence(hard)
    assert torch.equal(by_bytes.infer_single_nodes([ys[5]])[0], fresh.infer_single_nodes([ys[5]])[0])
    assert by_bytes.log_likelihood() == fresh.log_likelihood()


def test_float_evidence_that_is_hard_evidence_runs_on_state_bytes():
    """``em_step`` / ``em_step_statistics`` look at float evidence of large batches first: when every row is
    one of two rows the step runs on state bytes made from it (one pass, one look at its verdict) and
    agrees with the float route to rounding; evidence of another kind is taken as before, bit for bit, and
    the look is not repeated for a while."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel

    n = 32770
    ts, network, ys = _leaf_model("c3a", "float32", "cuda")
    gen = torch.Generator().manual_seed(21)
    states = (torch.rand((len(ys), n), generator=gen) < 0.25).to(torch.uint8)
    x = states.to(ts.dtype).cuda() * 0.999999 + 5e-7
    stacked = torch.stack([1 - x, x], dim=2).contiguous()  # rows (1 - 5e-7, 5e-7) and (1 - 0.9999995, 0.9999995): four values
    two_valued = Evidence(list(stacked.unbind(0)), ts)
    settings = SpaInferenceMachineSettings(ts, True)
    model = FlatModel.of(network)
    model.assign(model.gather())

    def machine_for(compaction):
        machine = HipSpaInferenceMachine(settings, network, ys, n)
        machine.enable_star_step(False)
        machine.enable_evidence_compaction(compaction)
        return machine

    plain = machine_for(False)
    ll_f, counts_f = plain.em_step_statistics(two_valued)
    assert plain.evidence_compactions() == 0
    compacting = machine_for(True)
    ll_c, counts_c = compacting.em_step_statistics(two_valued)
    assert compacting.evidence_compactions() == 1
    assert abs(ll_c.item() - ll_f.item()) <= 2e-6 * abs(ll_f.item())
    np.testing.assert_allclose(counts_c.cpu().numpy(), counts_f.cpu().numpy(), rtol=2e-6, atol=2e-6)
    # the whole step, and the same machine again on the next batch
    ll_a, new_a = machine_for(False).em_step(two_valued, 0.2, 0.01)
    new_a = new_a.clone()
    model.assign(model.gather())
    ll_b, new_b = compacting.em_step(two_valued, 0.2, 0.01)
    assert compacting.evidence_compactions() == 2
    assert abs(ll_b.item() - ll_a.item()) <= 2e-6 * abs(ll_a.item())
    np.testing.assert_allclose(new_b.cpu().numpy(), new_a.cpu().numpy(), rtol=2e-5, atol=2e-6)
    model.assign(model.gather())

    # one row of another kind anywhere: the float route, bit for bit
    soft = stacked.clone()
    soft[700, n - 3] = torch.tensor([0.25, 0.75], dtype=soft.dtype, device=soft.device)
    soft_evidence = Evidence(list(soft.unbind(0)), ts)
    ll_p, counts_p = machine_for(False).em_step_statistics(soft_evidence)
    looking = machine_for(True)
    ll_q, counts_q = looking.em_step_statistics(soft_evidence)
    assert looking.evidence_compactions() == 0
    assert ll_q.item() == ll_p.item() and torch.equal(counts_q, counts_p)
    # ... and the next batches are taken as they come for a while, whatever they are
    ll_r, counts_r = looking.em_step_statistics(two_valued)
    assert looking.evidence_compactions() == 0
    assert ll_r.item() == ll_f.item() and torch.equal(counts_r, counts_f)

    # The batch after one of that kind runs on the guess that it is of that kind too (the verdict of the pass
    # over it is read when the step is enqueued): the same batch again - the guess holds, the same results;
    guessing = machine_for(True)
    ll_1, counts_1 = guessing.em_step_statistics(two_valued)
    ll_2, counts_2 = guessing.em_step_statistics(two_valued)
    assert guessing.evidence_compactions() == 2
    assert ll_2.item() == ll_1.item() and torch.equal(counts_2, counts_1)
    # ... a batch of another kind - the guess is wrong, the sums are formed again on the float arrays;
    ll_3, counts_3 = guessing.em_step_statistics(soft_evidence)
    assert guessing.evidence_compactions() == 2
    assert ll_3.item() == ll_p.item() and torch.equal(counts_3, counts_p)
    # ... and in a whole step the parameters are put back before it runs again: as if nothing had been guessed
    guessed, straight = machine_for(True), machine_for(True)
    for machine in (guessed, straight):
        ll_g1, new_g1 = machine.em_step(two_valued, 0.2, 0.01)
        model.assign(model.gather())
    straight.enable_evidence_compaction(False)
    ll_s, new_s = straight.em_step(soft_evidence, 0.2, 0.01)
    new_s = new_s.clone()
    model.assign(model.gather())
    ll_g, new_g = guessed.em_step(soft_evidence, 0.2, 0.01)
    assert guessed.evidence_compactions() == 1
    assert ll_g.item() == ll_s.item() and torch.equal(new_g, new_s)
    model.assign(model.gather())
    # other rows of the same kind: the guess is wrong as well, the step runs again on the right ones
    other_rows = Evidence(list(torch.stack([1 - (states.to(ts.dtype).cuda() * 0.99 + 0.005), states.to(ts.dtype).cuda() * 0.99 + 0.005],
                                           dim=2).contiguous().unbind(0)), ts)
    ll_o, counts_o = machine_for(False).em_step_statistics(other_rows)
    switching = machine_for(True)
    switching.em_step_statistics(two_valued)
    ll_w, counts_w = switching.em_step_statistics(other_rows)
    assert switching.evidence_compactions() == 2
    assert abs(ll_w.item() - ll_o.item()) <= 2e-6 * abs(ll_o.item())
    np.testing.assert_allclose(counts_w.cpu().numpy(), counts_o.cpu().numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("iterations", [1, 2, 3, 4, 5, 6])
def test_state_bytes_step_at_iteration_counts_around_the_diameter(iterations):
    """The launches of a step on state bytes depend on the number of iterations: the reduce launch
    takes the leaves' normalisers from the last light loop launch, so with three iterations the loop
    launch before it is not left out, with one or two the float kernels serve the step.  Against the
    step on the float tensors the bytes stand for, run for the same number of iterations."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel

    n = 2050
    ts, network, ys = _leaf_model("c3a", "float32", "cuda")
    gen = torch.Generator().manual_seed(5)
    states = (torch.rand((len(ys), n), generator=gen) < 0.4).to(torch.uint8)
    hard = HardEvidence(states.cuda(), [2] * len(ys), ts, on_value=0.999999 + 5e-7, off_value=5e-7)
    hard.states[3, 2049] = 9
    tensors = [t.clone() for t in hard.data]
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations)
    model = FlatModel.of(network)
    model.assign(model.gather())

    def machine_for():
        machine = HipSpaInferenceMachine(settings, network, ys, n)
        machine.enable_star_step(False)
        return machine

    ll_f, counts_f = machine_for().em_step_statistics(Evidence(tensors, ts))
    ll_h, counts_h = machine_for().em_step_statistics(hard)
    assert abs(ll_h.item() - ll_f.item()) <= 2e-6 * abs(ll_f.item())
    np.testing.assert_allclose(counts_h.cpu().numpy(), counts_f.cpu().numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("iterations", [1, 2, 3, 5, 6])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_784_leaf_model_at_iteration_counts_around_the_diameter(iterations, dtype):
    """The launch sequence of the fused leaf families depends on the number of iterations (the first
    loop launch forms the per-state products, later ones only the normalisers' chunk products, the
    one before a reduce launch is folded into it from three iterations on, a full launch is fed
    per-family normalisers): 1, 2, 3, 5 and 6 iterations (the diameter is 4) against the float64
    oracle run for the same number, through the three statistics routes and the query route, at
    514 observations (an even number, three tiles)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    n = 514
    f32 = dtype == "float32"
    ts, network, ys = _leaf_model("c3a", dtype, "cuda")
    rows = _mixed_evidence(n, len(ys), 33, ts.dtype)
    tensors = list(torch.stack(rows).cuda().unbind(0))
    oracle = _oracle_for("c3a", network, ys, dtype, n, iterations, False, rows)
    want_ll = oracle.log_likelihood(False)
    nodes = list(network.nodes)
    want_counts = [oracle.infer_family(j).sum(dim=0) for j in range(len(nodes))]
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations)
    limit = (1e-5 if f32 else 1e-9) * abs(want_ll)

    def check_statistics(machine, ll, flat, route):
        assert abs(ll.item() - want_ll) <= limit, (route, ll.item(), want_ll)
        for j, got in enumerate(machine.split_counts(flat)):
            _check(got, want_counts[j], f"{route}: counts {nodes[j].name}", f32)

    machine = HipSpaInferenceMachine(settings, network, ys, n)
    machine.enable_star_step(False)
    check_statistics(machine, *machine.em_statistics(Evidence(tensors, ts)), "statistics")
    _check(machine.infer_single_nodes([nodes[0]])[0], oracle.infer_single(0), "single Q after statistics", f32)
    _check(machine.infer_nodes_with_parents([ys[9]])[0], oracle.infer_family(10), "family after statistics", f32)

    summed = HipSpaInferenceMachine(settings, network, ys, n)
    summed.enable_star_step(False)
    check_statistics(summed, *summed.em_step_statistics(Evidence(tensors, ts)), "step statistics")
    _check(summed.infer_single_nodes([nodes[0]])[0], oracle.infer_single(0), "single Q after step statistics", f32)

    queried = HipSpaInferenceMachine(settings, network, ys, n)
    queried.enter_evidence(Evidence(tensors, ts))
    assert abs(queried.log_likelihood() - want_ll) <= limit
    _check(queried.infer_single_nodes([nodes[0]])[0], oracle.infer_single(0), "single Q", f32)
    for j, got in enumerate(queried.split_counts(queried.expected_count_sums())):
        _check(got, want_counts[j], f"query route: counts {nodes[j].name}", f32)


@pytest.mark.parametrize("n", [1, 100])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("kind", ["c3a12", "c3a15", "c3a16", "c3a20"])
def test_many_class_784_leaf_models_at_small_batches(kind, dtype, n):
    """Few observations make many chunks of families: the combine of a many-state parent then works
    with its largest LDS footprint."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    f32 = dtype == "float32"
    ts, network, ys = _leaf_model(kind, dtype, "cuda")
    evidence = _mixed_evidence(n, len(ys), 13, ts.dtype)
    oracle = _oracle_for(kind, network, ys, dtype, n, None, False, evidence)
    settings = SpaInferenceMachineSettings(ts, False)
    nodes = list(network.nodes)
    want_ll = oracle.log_likelihood(False)
    machine = HipSpaInferenceMachine(settings, network, ys, n)
    machine.enter_evidence(Evidence([e.cuda() for e in evidence], ts))
    for j in (0, 1, len(nodes) - 1):
        _check(machine.infer_single_nodes([nodes[j]])[0], oracle.infer_single(j), f"single {nodes[j].name}", f32)
    assert abs(machine.log_likelihood() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    one_call = HipSpaInferenceMachine(settings, network, ys, n)
    ll, flat = one_call.em_statistics(Evidence([e.cuda() for e in evidence], ts))
    assert abs(ll.item() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    for j, got in enumerate(one_call.split_counts(flat)):
        _check(got, oracle.infer_family(j).sum(dim=0), f"counts {nodes[j].name}", f32)
