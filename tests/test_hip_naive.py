"""GPU parity of the exact (enumeration) machine, through the C ABI (bn_exact_*).

Checked against (1) what the reference's NaiveInferenceMachine returned (tests/golden/naive_*.npz),
(2) the CPU restatement in float64 on generated networks the fixtures do not cover, and (3) the
sum-product machine on the same device, which must agree with it on trees.

Tolerance: float64 1e-12 absolute on probabilities (sums of at most a few thousand double terms)
and 1e-12 relative on log-likelihoods; float32 outputs are rounded once from double sums, so
2e-7 absolute against the float64 restatement on the same float32 inputs, and 1e-5 against the
reference's float32 outputs (whose own sums ran in float32).
"""

import numpy as np
import pytest
import torch

from conftest import Fixture, naive_fixture_names

pytestmark = pytest.mark.gpu


def _machine(fx, dtype_name, average=False):
    from bayesiannetwork_amd.inference_machines.common import InferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_naive import HipNaiveInferenceMachine
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype=dtype_name)
    network, observed, by_name = fx.build(ts)
    machine = HipNaiveInferenceMachine(InferenceMachineSettings(ts, average), network, observed)
    return ts, network, observed, by_name, machine


@pytest.mark.parametrize("name", naive_fixture_names())
def test_exact_machine_matches_reference_fixture(name):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from test_naive_oracle_golden import build_oracle

    fx = Fixture(name)
    f32 = fx.dtype == torch.float32
    ts, network, observed, by_name, machine = _machine(fx, fx.meta["dtype"])
    machine.enter_evidence(Evidence(fx.evidence_tensors("cuda"), ts))
    nodes = list(network.nodes)
    assert [n.name for n in nodes] == fx.meta["network_order"]
    tol_ref = 1e-5 if f32 else 1e-12
    singles = machine.infer_single_nodes(nodes)
    families = machine.infer_nodes_with_parents(nodes)
    for node, s, f in zip(nodes, singles, families):
        assert s.dtype == fx.dtype and f.dtype == fx.dtype
        np.testing.assert_allclose(s.cpu().numpy(), fx[f"naive/single/{node.name}"], rtol=0, atol=tol_ref)
        assert f.shape == fx[f"naive/family/{node.name}"].shape
        np.testing.assert_allclose(f.cpu().numpy(), fx[f"naive/family/{node.name}"], rtol=0, atol=tol_ref)
    np.testing.assert_allclose(machine.log_likelihoods().cpu().numpy(), fx["naive/log_likelihoods"],
                               rtol=tol_ref, atol=tol_ref)
    want = float(fx["naive/ll_sum"])
    assert abs(machine.log_likelihood() - want) <= tol_ref * abs(want) + tol_ref
    _, _, _, _, averaging = _machine(fx, fx.meta["dtype"], average=True)
    averaging.enter_evidence(Evidence(fx.evidence_tensors("cuda"), ts))
    want = float(fx["naive/ll_avg"])
    assert abs(averaging.log_likelihood() - want) <= tol_ref * abs(want) + tol_ref

    if f32:  # against float64 arithmetic on the same float32 inputs: only the final rounding differs
        oracle, order, parents = build_oracle(fx, torch.float64)
        oracle.enter_evidence(fx.evidence_tensors())
        for j, (s, f) in enumerate(zip(singles, families)):
            np.testing.assert_allclose(s.cpu().double().numpy(), oracle.infer([j]).numpy(), rtol=0, atol=2e-7)
            np.testing.assert_allclose(f.cpu().double().numpy(), oracle.infer(parents[j] + [j]).numpy(), rtol=0, atol=2e-7)


def test_exact_machine_before_evidence_and_errors():
    from bayesiannetwork_amd.inference_machines.evidence import Evidence

    fx = Fixture("c2_docs2_f64")
    ts, network, observed, by_name, machine = _machine(fx, "float64")
    # prior marginals, shape [1, S], before any evidence (naive_inference_machine.py:31)
    q2 = machine.infer_single_nodes([by_name["Q2"]])[0]
    assert q2.shape == (1, 2)
    prior_q1 = torch.from_numpy(fx["cpt/Q1"])
    want = prior_q1 @ torch.from_numpy(fx["cpt/Q2"])
    np.testing.assert_allclose(q2.cpu().numpy()[0], want.numpy(), rtol=0, atol=1e-14)
    with pytest.raises(RuntimeError, match="Did you enter evidence"):
        machine.log_likelihood()
    with pytest.raises(Exception, match="Length of evidence"):
        machine.enter_evidence(Evidence(fx.evidence_tensors("cuda")[:2], ts))
    # a machine without observed nodes answers queries but has no log-likelihood
    from bayesiannetwork_amd.inference_machines.common import InferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_naive import HipNaiveInferenceMachine

    blind = HipNaiveInferenceMachine(InferenceMachineSettings(ts, False), network, [])
    np.testing.assert_allclose(blind.infer_single_nodes([by_name["Q2"]])[0].cpu().numpy()[0], want.numpy(), atol=1e-14)
    with pytest.raises(Exception, match="0 observed nodes"):
        blind.log_likelihood()
    # evidence batches of different sizes one after the other
    rows = fx.evidence_tensors("cuda")
    machine.enter_evidence(Evidence(rows, ts))
    full = machine.infer_single_nodes([by_name["Q1"]])[0].clone()
    machine.enter_evidence(Evidence([r[3:10] for r in rows], ts))
    part = machine.infer_single_nodes([by_name["Q1"]])[0]
    assert part.shape == (7, 2)
    assert torch.equal(part, full[3:10])


def test_exact_machine_rejects_networks_too_large_to_enumerate():
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.common import InferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_naive import HipNaiveInferenceMachine
    from bayesiannetwork_amd.inference_machines.hip_spa._native import BnError
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype="float32")
    builder = BayesianNetworkBuilder()
    root = Node(torch.full((10,), 0.1, device="cuda"), name="Q")
    builder.add_node(root)
    leaves = []
    for i in range(12):  # 10 * 10^12 joint states
        leaf = Node(torch.full((10, 10), 0.1, device="cuda"), name=f"Y{i}")
        builder.add_node(leaf, parents=[root])
        leaves.append(leaf)
    with pytest.raises(BnError, match="too large"):
        HipNaiveInferenceMachine(InferenceMachineSettings(ts, False), builder.build(), leaves)


def _generated(seed, spec, observed_names, n, dtype_name):
    """Network from (name, cpt shape, parent names) rows with random CPTs and soft evidence."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.torch_settings import TorchSettings

    gen = torch.Generator().manual_seed(seed)
    ts = TorchSettings(device="cuda", dtype=dtype_name)
    cpts = {}
    for name, shape, _ in spec:
        p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
        cpts[name] = (p / p.sum(-1, keepdim=True)).to(ts.dtype)
    by_name = {name: Node(cpts[name].to("cuda"), name=name) for name, _, _ in spec}
    builder = BayesianNetworkBuilder()
    for name, _, parents in spec:
        builder.add_node(by_name[name], parents=[by_name[p] for p in parents])
    network = builder.build()
    evidence = []
    for name in observed_names:
        rows = torch.rand((n, cpts[name].shape[-1]), generator=gen, dtype=torch.float64) + 1e-3
        evidence.append(rows.to(ts.dtype))
    return ts, network, by_name, cpts, evidence


LOOPY_SPEC = [("A", (3,), []), ("B", (3, 2), ["A"]), ("C", (3, 2, 4), ["A", "B"]), ("D", (2, 4, 3), ["B", "C"]),
              ("E", (4, 2), ["C"]), ("F", (3, 2, 5), ["D", "E"]), ("G", (5, 2), ["F"]), ("H", (3, 2, 2), ["A", "G"])]


@pytest.mark.parametrize("dtype_name", ["float64", "float32"])
@pytest.mark.parametrize("n", [1, 257, 5000])
def test_exact_machine_on_generated_loopy_network(dtype_name, n):
    """Eight nodes, 17 280 joint states, four of them observed; every single and family marginal
    and every per-observation log-likelihood against the float64 CPU restatement."""
    from bayesiannetwork_amd.inference_machines.common import InferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_naive import HipNaiveInferenceMachine
    from oracle.naive_oracle import NaiveOracle

    observed_names = ["H", "D", "E", "B"]
    ts, network, by_name, cpts, evidence = _generated(5, LOOPY_SPEC, observed_names, n, dtype_name)
    nodes = list(network.nodes)
    order = [node.name for node in nodes]
    machine = HipNaiveInferenceMachine(InferenceMachineSettings(ts, False), network, [by_name[o] for o in observed_names])
    machine.enter_evidence(Evidence([e.to("cuda") for e in evidence], ts))

    parents = [[order.index(p.name) for p in network.parents_of(node)] for node in nodes]
    oracle = NaiveOracle([cpts[name] for name in order], parents, [order.index(o) for o in observed_names])
    oracle.enter_evidence([e.double() for e in evidence])
    atol = 1e-12 if dtype_name == "float64" else 2e-7
    for j, node in enumerate(nodes):
        got = machine.infer_single_nodes([node])[0]
        np.testing.assert_allclose(got.cpu().double().numpy(), oracle.infer([j]).numpy(), rtol=0, atol=atol)
        fam = machine.infer_nodes_with_parents([node])[0]
        want = oracle.infer(parents[j] + [j])
        assert fam.shape == want.shape
        np.testing.assert_allclose(fam.cpu().double().numpy(), want.numpy(), rtol=0, atol=atol)
        np.testing.assert_allclose(fam.double().sum(dim=tuple(range(1, fam.dim()))).cpu().numpy(), 1.0, atol=5e-6)
    ll_tol = 1e-12 if dtype_name == "float64" else 1e-6
    np.testing.assert_allclose(machine.log_likelihoods().cpu().double().numpy(), oracle.log_likelihoods.numpy(),
                               rtol=ll_tol, atol=ll_tol)
    want_ll = oracle.log_likelihood(False)
    assert abs(machine.log_likelihood() - want_ll) <= 1e-12 * abs(want_ll) * (1 if dtype_name == "float64" else 1e6)


@pytest.mark.parametrize("name", ["tree8_f64", "mixed_group_f64", "polytree_f64", "c3a_mini16_f64"])
def test_sum_product_agrees_with_exact_machine_on_trees(name):
    """Self-check on the device, no stored answers involved: on trees and polytrees the sum-product
    single marginals are exact, so the two machines must agree.  mixed_group lists a child before
    its parent (an order the reference's brute-force machine gets wrong); c3a_mini16 has 17 nodes
    and 655 360 joint states."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    fx = Fixture(name)
    ts, network, observed, by_name, exact = _machine(fx, "float64")
    rows = fx.evidence_tensors("cuda")
    n = rows[0].shape[0]
    exact.enter_evidence(Evidence(rows, ts))
    spa = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), network, observed, n)
    spa.enter_evidence(Evidence(rows, ts))
    nodes = list(network.nodes)
    for node, a, b in zip(nodes, spa.infer_single_nodes(nodes), exact.infer_single_nodes(nodes)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-9, atol=1e-12, err_msg=node.name)
    for node, a, b in zip(nodes, spa.infer_nodes_with_parents(nodes), exact.infer_nodes_with_parents(nodes)):
        # the exact machine returns axes in network order, the sum-product machine in CPT order
        axes = [nodes.index(p) for p in network.parents_of(node)] + [nodes.index(node)]
        perm = [0] + [1 + sorted(axes).index(x) for x in axes]
        np.testing.assert_allclose(a.cpu().numpy(), b.permute(perm).cpu().numpy(), rtol=1e-9, atol=1e-12,
                                   err_msg=node.name)
