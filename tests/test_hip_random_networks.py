"""Differential test on randomly drawn networks: the HIP machine against the CPU oracle in float64
on the same inputs.  The draws (fixed seeds) mix what the hand-made cases keep apart: cardinalities
1..7 (so shapes with and without a specialised kernel sit in one network), up to three parents,
trees and loopy graphs, observed inner nodes, soft / hard / near-contradicting evidence, batch
sizes around the tile boundaries.  The oracle is pinned to the reference by the golden fixtures;
nothing here has a reference-generated answer.

Tolerance as in test_hip_generated_networks.py (float64 1e-9 relative; float32 1e-5 relative +
1e-7 of the tensor's scale, normalisers 2e-6 + 1e-7 per input; 4x on loopy networks)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _draw(seed):
    gen = torch.Generator().manual_seed(1000 + seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=gen).item())

    num_nodes = ri(3, 14)
    loopy = seed % 3 == 0
    cards = [ri(1 if seed % 5 == 0 else 2, 7) for _ in range(num_nodes)]
    spec = []
    for j in range(num_nodes):
        max_parents = 0 if j == 0 else (min(j, 3) if loopy else 1)
        k = ri(1, max_parents) if max_parents else 0
        parents = sorted(torch.randperm(j, generator=gen)[:k].tolist()) if k else []
        shape = tuple(cards[p] for p in parents) + (cards[j],)
        cpt = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.02
        if seed % 4 == 1:  # some nearly deterministic rows
            cpt = cpt ** 6
        spec.append((f"X{j}", parents, cpt / cpt.sum(-1, keepdim=True)))
    if seed % 7 in (3, 5):
        # a crowd of leaves under one node (two of them for seed % 7 == 5, loopy draws give the leaves
        # a second parent): the wide-parent and fused leaf-family paths with arbitrary neighbours
        hubs = [ri(0, num_nodes - 1) for _ in range(2 if seed % 7 == 5 else 1)]
        leaf_card = ri(2, 4)
        for _ in range(ri(26, 60)):
            parents = sorted(set(hubs))
            shape = tuple(cards[p] for p in parents) + (leaf_card,)
            cpt = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.02
            cards.append(leaf_card)
            spec.append((f"X{len(spec)}", parents, cpt / cpt.sum(-1, keepdim=True)))
        num_nodes = len(spec)
    observed = [j for j in range(num_nodes) if torch.rand((1,), generator=gen).item() < 0.45]
    if not observed:
        observed = [num_nodes - 1]
    n = [1, 63, 64, 129, 257, 600][seed % 6]
    evidence = []
    for j in observed:
        s = cards[j]
        soft = torch.rand((n, s), generator=gen, dtype=torch.float64) + 1e-3
        hard = torch.nn.functional.one_hot(torch.randint(0, s, (n,), generator=gen), s).double()
        if seed % 4 != 2:  # soft-clamped like the reference's experiments; exact zeros and ones otherwise (its docs)
            hard = hard * (1 - 1e-6) + 1e-6 / max(s, 1)
        pick = (torch.rand((n, 1), generator=gen) < 0.5).double()
        evidence.append(pick * hard + (1 - pick) * soft / soft.sum(-1, keepdim=True))
    return spec, observed, evidence, n, loopy


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("seed", range(56))
def test_random_network_matches_oracle(seed, dtype):
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings
    from oracle.spa_oracle import SpaOracle

    spec, observed_idx, evidence, n, loopy = _draw(seed)
    iterations = 7 if loopy else None

    def make(device, torch_dtype, values=None):
        ts = TorchSettings(device=device, dtype=torch_dtype)
        nodes = []
        for j, (name, parents, cpt) in enumerate(spec):
            if values is not None:
                c = values[j]
            else:  # rows re-normalised in the element type, as Node's validation wants them
                c = cpt.to(ts.dtype)
                c = c / c.sum(-1, keepdim=True)
            nodes.append(Node(c.to(ts.device), name=name))
        builder = BayesianNetworkBuilder()
        for node, (_, parents, _) in zip(nodes, spec):
            builder.add_node(node, parents=[nodes[p] for p in parents])
        return ts, builder.build(), nodes

    ts, network, nodes = make("cuda", dtype)
    # is_tree decides whether explicit iterations are needed: ask the engine the way a user would
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=True)
    try:
        machine = HipSpaInferenceMachine(settings, network, [nodes[j] for j in observed_idx], n)
    except ValueError:  # a loopy draw without explicit iterations
        iterations = 7
        settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=True)
        machine = HipSpaInferenceMachine(settings, network, [nodes[j] for j in observed_idx], n)
    ev_dev = [e.to(ts.dtype) for e in evidence]
    machine.enter_evidence(Evidence([e.to("cuda") for e in ev_dev], ts))

    _, network64, nodes64 = make("cpu", "float64")
    for node64, node in zip(nodes64, nodes):  # the very values the device network holds
        node64.cpt = node.cpt.detach().cpu().double()
    oracle = SpaOracle(network64, [nodes64[j] for j in observed_idx], n, dtype=torch.float64,
                       num_iterations=iterations, allow_loops=True)
    assert oracle.num_iterations == machine.num_iterations
    oracle.enter_evidence([e.double() for e in ev_dev])

    f32 = dtype == "float32"
    slack = 1.0 if oracle.num_iterations and not iterations else 4.0

    def check(got, want, what, extra_atol=0.0):
        got = got.detach().cpu().double().numpy()
        want = want.detach().double().numpy()
        assert np.isfinite(got).all(), what
        if f32:
            scale = float(np.max(np.abs(want))) if want.size else 1.0
            bound = slack * (1e-5 * np.abs(want) + 1e-7 * scale + extra_atol)
        else:
            bound = slack * (1e-9 * np.abs(want) + 1e-11)
        worst = float(np.max(np.abs(got - want) / bound))
        assert worst <= 1.0, f"seed {seed} {what}: {worst:.2f}x the bound"

    order = list(network.nodes)
    index64 = {node.name: j for j, node in enumerate(network64.nodes)}
    ll, want_ll = machine.log_likelihood(), oracle.log_likelihood(False)
    assert abs(ll - want_ll) <= slack * (1e-5 if f32 else 1e-9) * abs(want_ll) + 1e-9, (seed, ll, want_ll)
    observed_set = {nodes[j] for j in observed_idx}
    for node in order:
        j = index64[node.name]
        check(machine.infer_single_nodes([node])[0], oracle.infer_single(j), f"single {node.name}")
        check(machine.infer_nodes_with_parents([node])[0], oracle.infer_family(j), f"family {node.name}")
        inputs = len(list(network.children_of(node))) + 1 + (node in observed_set)
        check(machine.local_log_likelihoods(node), oracle.vgroups[oracle.v_of[j][0]]["c"][oracle.v_of[j][1]],
              f"c {node.name}", extra_atol=2e-6 + 1e-7 * inputs)
    counts = machine.split_counts(machine.expected_count_sums())
    for node, got in zip(order, counts):
        check(got, oracle.infer_family(index64[node.name]).sum(dim=0), f"counts {node.name}")
    # the one-call route (evidence + iterations + statistics; last iteration in reduce mode where a
    # fused kernel serves the family) against the same oracle
    second = HipSpaInferenceMachine(settings, network, [nodes[j] for j in observed_idx], n)
    ll2, flat2 = second.em_statistics(Evidence([e.to("cuda") for e in ev_dev], ts))
    assert abs(ll2.item() - want_ll) <= slack * (1e-5 if f32 else 1e-9) * abs(want_ll) + 1e-9
    for node, got in zip(order, second.split_counts(flat2)):
        check(got, oracle.infer_family(index64[node.name]).sum(dim=0), f"counts via statistics {node.name}")
    check(second.infer_single_nodes([order[0]])[0], oracle.infer_single(index64[order[0].name]), "single after statistics")
