"""GPU parity on generated networks that put the specialised kernels into corners the golden
fixtures do not reach (partly observed leaf families, odd observation counts, several family
shapes under one wide parent, two-parent families, 32-state nodes), against the CPU oracle run in
float64 on the same inputs.  The oracle itself is pinned to the reference by the golden fixtures
(test_oracle_golden.py); nothing here has a reference-generated fixture, so these cases are
checked against the oracle only.

Tolerance: float64 1e-9 relative; float32 as in test_hip_parity.py (1e-5 relative + 1e-7 of the
tensor's scale; normalisers 2e-6 + 2e-7 per input message as in test_hip_headline_models.py -- the
root of a 60-leaf tree deviates by up to 1.3e-5 at any number of classes, tools/diag_root_c.py; 4x on
loopy networks).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand_cpt(gen, shape):
    p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
    return p / p.sum(dim=-1, keepdim=True)


def _wide_tree(gen):
    """Q(10) with 40 (10,2) leaves, 20 (10,3) leaves and 3 (10,4) leaves; every fourth leaf unobserved."""
    spec = [("Q", (10,), [])]
    for i in range(40):
        spec.append((f"A{i}", (10, 2), ["Q"]))
    for i in range(20):
        spec.append((f"B{i}", (10, 3), ["Q"]))
    for i in range(3):
        spec.append((f"C{i}", (10, 4), ["Q"]))
    observed = [name for k, (name, _, parents) in enumerate(spec) if parents and k % 4 != 0]
    return spec, observed, None, False


def _wide_tree_of(classes):
    """Q(classes) with 40 (classes,2) leaves and 20 (classes,3) leaves; every fourth leaf unobserved."""
    def case(gen):
        spec = [("Q", (classes,), [])]
        for i in range(40):
            spec.append((f"A{i}", (classes, 2), ["Q"]))
        for i in range(20):
            spec.append((f"B{i}", (classes, 3), ["Q"]))
        observed = [name for k, (name, _, parents) in enumerate(spec) if parents and k % 4 != 0]
        return spec, observed, None, False
    return case


def _wide_parent_of_inner_nodes(classes):
    """Q(classes) -> 30 hidden H_i(3 | Q) -> one observed leaf Y_i(2 | H_i) each: the many-input node's
    children are not leaf families, so its variable pass runs on the standalone wide-node kernels
    (float32) or the generic one (float64)."""
    def case(gen):
        spec = [("Q", (classes,), [])]
        for i in range(30):
            spec.append((f"H{i}", (classes, 3), ["Q"]))
        for i in range(30):
            spec.append((f"Y{i}", (3, 2), [f"H{i}"]))
        return spec, [f"Y{i}" for i in range(30) if i % 6 != 0], None, False
    return case


def _two_parent_loopy(gen):
    """Q(3) -> F1(2), F2(2) -> 30 leaves (2 | F1, F2), every fifth leaf unobserved; T = 6."""
    spec = [("Q", (3,), []), ("F1", (3, 2), ["Q"]), ("F2", (3, 2), ["Q"])]
    for i in range(30):
        spec.append((f"Y{i}", (2, 2, 2), ["F1", "F2"]))
    observed = [f"Y{i}" for i in range(30) if i % 5 != 0]
    return spec, observed, 6, True


# (the last four: shapes served by a shape pack, a shared object of its own per CPT shape)
TWO_PARENT_SHAPES = [(2, 2, 3), (2, 2, 4), (2, 3, 2), (3, 2, 2), (2, 3, 3), (3, 2, 3), (3, 3, 2), (3, 3, 3), (4, 4, 2),
                     (3, 3, 4), (2, 4, 2), (4, 2, 2), (2, 2, 5)]


def _two_parent_shape(d0, d1, s):
    """Q(3) -> F1(d0), F2(d1) -> 28 leaves (s | F1, F2), every fifth leaf unobserved; T = 5."""
    def case(gen):
        spec = [("Q", (3,), []), ("F1", (3, d0), ["Q"]), ("F2", (3, d1), ["Q"])]
        for i in range(28):
            spec.append((f"Y{i}", (d0, d1, s), ["F1", "F2"]))
        observed = [f"Y{i}" for i in range(28) if i % 5 != 0]
        return spec, observed, 5, True
    return case


def _many_state_tree(gen):
    """Root(32) -> two hidden 32-state nodes -> one observed 32-state leaf each, plus a 5-state observed child of the root."""
    spec = [("R", (32,), []), ("H0", (32, 32), ["R"]), ("H1", (32, 32), ["R"]), ("Y0", (32, 32), ["H0"]),
            ("Y1", (32, 32), ["H1"]), ("Z", (32, 5), ["R"])]
    return spec, ["Y0", "Y1", "Z"], None, False


def _deep_chain_with_wide_node(gen):
    """A(4) -> B(5) -> 30 leaves (5,2) all observed, and A -> 26 leaves (4,3), half observed: two wide parents in one tree."""
    spec = [("A", (4,), []), ("B", (4, 5), ["A"])]
    for i in range(30):
        spec.append((f"L{i}", (5, 2), ["B"]))
    for i in range(26):
        spec.append((f"M{i}", (4, 3), ["A"]))
    observed = [f"L{i}" for i in range(30)] + [f"M{i}" for i in range(0, 26, 2)]
    return spec, observed, None, False


CASES = {
    "wide_tree_partly_observed": _wide_tree,
    "two_parent_loopy_partly_observed": _two_parent_loopy,
    "many_state_tree": _many_state_tree,
    "two_wide_parents": _deep_chain_with_wide_node,
}
# (11, 13, 24, 32: cardinalities served by a cardinality pack, a shared object of its own per cardinality)
CASES.update({f"wide_tree_{classes}_classes": _wide_tree_of(classes) for classes in (6, 7, 8, 9, 11, 12, 13, 15, 16, 20, 24, 32)})
CASES.update({f"wide_parent_{classes}_of_inner_nodes": _wide_parent_of_inner_nodes(classes) for classes in (7, 10, 13, 15, 16, 20, 24)})
CASES.update({f"two_parent_{d0}x{d1}x{s}": _two_parent_shape(d0, d1, s) for d0, d1, s in TWO_PARENT_SHAPES})


def _build(name, dtype, n):
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.torch_settings import TorchSettings

    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    spec, observed_names, iterations, loops = CASES[name](gen)
    cpts = {node: _rand_cpt(gen, shape) for node, shape, _ in spec}
    evidence = {}
    for node in observed_names:
        states = cpts[node].shape[-1]
        soft = torch.rand((n, states), generator=gen, dtype=torch.float64) + 1e-3
        hard = torch.nn.functional.one_hot(torch.randint(0, states, (n,), generator=gen), states).double() * (1 - 1e-6) + 5e-7
        pick = (torch.rand((n, 1), generator=gen) < 0.5).double()
        evidence[node] = pick * hard + (1 - pick) * soft / soft.sum(-1, keepdim=True)

    def make(device, torch_dtype):
        ts = TorchSettings(device=device, dtype=torch_dtype)
        by_name = {node: Node(cpts[node].to(ts.dtype).to(ts.device), name=node) for node, _, _ in spec}
        builder = BayesianNetworkBuilder()
        for node, _, parents in spec:
            builder.add_node(by_name[node], parents=[by_name[p] for p in parents])
        return ts, builder.build(), by_name

    return spec, observed_names, iterations, loops, cpts, evidence, make


@pytest.mark.parametrize("n", [1, 77, 300])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_generated_network_matches_oracle(name, dtype, n):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from oracle.spa_oracle import SpaOracle

    spec, observed_names, iterations, loops, cpts, evidence, make = _build(name, dtype, n)
    ts, network, by_name = make("cuda", dtype)
    observed = [by_name[o] for o in observed_names]
    machine = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops), network, observed, n)
    # the float32 machine sees float32 inputs; the yardstick is the oracle in float64 on those same values
    ev_dev = [evidence[o].to(ts.dtype) for o in observed_names]
    machine.enter_evidence(Evidence([e.to("cuda") for e in ev_dev], ts))

    if n == 77:
        # reading a message before anything ran makes the engine write out the initial uniform
        # messages; the first iteration then starts from stored arrays instead of the "fresh" shortcut
        first_leaf = by_name[spec[-1][0]]
        initial = machine.message(first_leaf, first_leaf, "f2v")
        assert torch.equal(initial, torch.full_like(initial, 1.0 / first_leaf.num_states))

    ts64, network64, by_name64 = make("cpu", "float64")
    for node, _, _ in spec:  # same rounded CPT values as the device network
        by_name64[node].cpt = by_name[node].cpt.detach().cpu().double()
    oracle = SpaOracle(network64, [by_name64[o] for o in observed_names], n, dtype=torch.float64,
                       num_iterations=iterations, allow_loops=loops)
    oracle.enter_evidence([e.double() for e in ev_dev])

    slack = 4.0 if loops else 1.0
    f32 = dtype == "float32"

    def check(got, want, what, extra_atol=0.0):
        got = got.detach().cpu().double().numpy()
        want = want.detach().double().numpy()
        assert np.isfinite(got).all(), what
        if f32:
            scale = float(np.max(np.abs(want))) if want.size else 1.0
            bound = slack * (1e-5 * np.abs(want) + 1e-7 * scale + extra_atol)
        else:
            bound = 1e-9 * np.abs(want) + 1e-11
        worst = float(np.max(np.abs(got - want) / bound))
        assert worst <= 1.0, f"{what}: {worst:.2f}x the bound"

    nodes = list(network.nodes)
    # log-likelihood first: it runs through the mode that does not store the last iteration
    ll = machine.log_likelihood()
    want_ll = oracle.log_likelihood(False)
    assert abs(ll - want_ll) <= (1e-5 if f32 else 1e-9) * slack * abs(want_ll) + 1e-9
    observed_set = set(observed)
    for j, node in enumerate(nodes):
        check(machine.infer_single_nodes([node])[0], oracle.infer_single(j), f"single {node.name}")
    for j in (0, 1, len(nodes) // 2, len(nodes) - 1):
        check(machine.infer_nodes_with_parents([nodes[j]])[0], oracle.infer_family(j), f"family {nodes[j].name}")
    for j, node in enumerate(nodes):
        inputs = len(list(network.children_of(node))) + 1 + (node in observed_set)
        check(machine.local_log_likelihoods(node), oracle.vgroups[oracle.v_of[j][0]]["c"][oracle.v_of[j][1]],
              f"c {node.name}", extra_atol=2e-6 + 2e-7 * inputs)
    # expected counts through both routes: statistics call on a second machine, accumulate on the first
    counts = machine.split_counts(machine.expected_count_sums())
    second = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops), network, observed, n)
    second.enter_evidence(Evidence([e.to("cuda") for e in ev_dev], ts))
    ll2, flat2 = second.em_statistics()
    assert abs(ll2.item() - want_ll) <= (1e-5 if f32 else 1e-9) * slack * abs(want_ll) + 1e-9
    for j, (node, got, got2) in enumerate(zip(nodes, counts, second.split_counts(flat2))):
        want = oracle.infer_family(j).sum(dim=0)
        check(got, want, f"counts {node.name}")
        check(got2, want, f"counts via statistics {node.name}")


@pytest.mark.parametrize("n", [10_000, 60_000])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("name", ["wide_tree_partly_observed", "two_wide_parents", "wide_tree_7_classes", "wide_tree_16_classes",
                                  "wide_tree_11_classes", "wide_tree_32_classes"])
def test_several_fused_groups_at_large_batches(name, dtype, n):
    """Plans with more than one fused leaf-family group share the partial-sum slots of the
    log-likelihood reduction; at these batch sizes one group alone could fill all of them.  The
    one-call statistics route must agree with iterate + log_likelihood + accumulate and with the
    float64 oracle."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from oracle.spa_oracle import SpaOracle

    spec, observed_names, iterations, loops, cpts, evidence, make = _build(name, dtype, n)
    ts, network, by_name = make("cuda", dtype)
    observed = [by_name[o] for o in observed_names]
    ev_dev = [evidence[o].to(ts.dtype) for o in observed_names]
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops)

    one_call = HipSpaInferenceMachine(settings, network, observed, n)
    one_call.enter_evidence(Evidence([e.to("cuda") for e in ev_dev], ts))
    ll_a, counts_a = one_call.em_statistics()
    # a later query re-reads the normalisers as stored in the plan dtype
    assert abs(one_call.log_likelihood() - ll_a.item()) <= (1e-6 if dtype == "float32" else 1e-12) * abs(ll_a.item())

    separate = HipSpaInferenceMachine(settings, network, observed, n)
    separate.enter_evidence(Evidence([e.to("cuda") for e in ev_dev], ts))
    separate.infer_single_nodes([observed[0]])  # bn_iterate
    ll_b = separate.log_likelihood_tensor()     # bn_log_likelihood
    counts_b = separate.expected_count_sums()   # bn_em_accumulate

    f32 = dtype == "float32"
    ll_rtol = 1e-6 if f32 else 1e-12
    assert abs(ll_a.item() - ll_b.item()) <= ll_rtol * abs(ll_b.item())
    np.testing.assert_allclose(counts_a.cpu().numpy(), counts_b.cpu().numpy(), rtol=2e-5 if f32 else 1e-10,
                               atol=(1e-7 if f32 else 1e-12) * n)

    ts64, network64, by_name64 = make("cpu", "float64")
    for node, _, _ in spec:
        by_name64[node].cpt = by_name[node].cpt.detach().cpu().double()
    oracle = SpaOracle(network64, [by_name64[o] for o in observed_names], n, dtype=torch.float64,
                       num_iterations=iterations, allow_loops=loops)
    oracle.enter_evidence([e.double() for e in ev_dev])
    want_ll = oracle.log_likelihood(False)
    assert abs(ll_a.item() - want_ll) <= (1e-5 if f32 else 1e-9) * abs(want_ll)
    for j, got in enumerate(one_call.split_counts(counts_a)):
        want = oracle.infer_family(j).sum(dim=0).numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5 if f32 else 1e-9,
                                   atol=(1e-7 if f32 else 1e-11) * float(np.max(np.abs(want))))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("shape", [(2, 2, 2)] + TWO_PARENT_SHAPES)
def test_two_parent_shapes_run_on_the_fused_kernel(shape, dtype):
    """The oracle comparisons above pass on whichever kernels serve a network; this one checks that
    every two-parent leaf-family shape listed has the fused kernel as its route."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import enable_profiling, profile_report

    d0, d1, s = shape
    name = f"two_parent_{d0}x{d1}x{s}"
    CASES.setdefault(name, _two_parent_shape(d0, d1, s))
    n = 300
    spec, observed_names, iterations, loops, cpts, evidence, make = _build(name, dtype, n)
    ts, network, by_name = make("cuda", dtype)
    observed = [by_name[o] for o in observed_names]
    machine = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops), network, observed, n)
    machine.enter_evidence(Evidence([evidence[o].to(ts.dtype).to("cuda") for o in observed_names], ts))
    enable_profiling(True)
    try:
        profile_report()
        machine.log_likelihood()
        report = profile_report()
    finally:
        enable_profiling(False)
    label = f"leaf_family[{d0}x{d1}x{s},K=28]"
    assert sum(r["launches"] for r in report if r["kernel"] == label) >= 1, report


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("classes", [6, 7, 8, 9, 11, 12, 13, 15, 16, 20, 24, 32])
def test_every_parent_cardinality_up_to_ten_runs_on_the_fused_kernels(classes, dtype):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import enable_profiling, profile_report

    n = 300
    name = f"wide_tree_{classes}_classes"
    spec, observed_names, iterations, loops, cpts, evidence, make = _build(name, dtype, n)
    ts, network, by_name = make("cuda", dtype)
    observed = [by_name[o] for o in observed_names]
    machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), network, observed, n)
    machine.enter_evidence(Evidence([evidence[o].to(ts.dtype).to("cuda") for o in observed_names], ts))
    enable_profiling(True)
    try:
        profile_report()
        machine.log_likelihood()
        report = profile_report()
    finally:
        enable_profiling(False)
    for label in (f"leaf_family[{classes}x2,K=40]", f"leaf_family[{classes}x3,K=20]"):
        assert sum(r["launches"] for r in report if r["kernel"] == label) >= 1, report
