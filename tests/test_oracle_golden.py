"""The CPU oracle against the reference's own outputs (CPU only).

Golden files were produced by ``oracle/make_fixtures.py`` running the real
reference package; the printed numbers of the reference's docs are checked
literally as well."""

import numpy as np
import pytest
import torch

from conftest import Fixture, em_fixture_names, fixture_names
from bayesiannetwork_amd.torch_settings import TorchSettings
from oracle.spa_oracle import SpaOracle, m_step_batch, m_step_full


def make_oracle(fx, cpts=None):
    ts = TorchSettings(device="cpu", dtype=fx.meta["dtype"])
    network, observed, by_name = fx.build(ts)
    oracle = SpaOracle(
        network,
        observed,
        fx.meta["N"],
        dtype=fx.dtype,
        num_iterations=fx.meta["num_iterations"],
        allow_loops=fx.meta["allow_loops"],
        cpts=cpts,
    )
    return oracle, network, observed, by_name


@pytest.mark.parametrize("name", fixture_names())
def test_oracle_matches_reference_outputs(name):
    fx = Fixture(name)
    oracle, network, observed, _ = make_oracle(fx)
    assert oracle.num_iterations == fx.meta["T_used"]
    order = [n.name for n in network.nodes]
    assert order == fx.meta["network_order"]
    if observed:
        oracle.enter_evidence(fx.evidence_tensors())
    tol = dict(rtol=1e-12, atol=0) if fx.dtype == torch.float64 else dict(rtol=1e-6, atol=1e-30)
    for j, node_name in enumerate(order):
        if f"out/single/{node_name}" in fx:
            np.testing.assert_allclose(oracle.infer_single(j).numpy(), fx[f"out/single/{node_name}"], **tol)
        if f"out/family/{node_name}" in fx:
            np.testing.assert_allclose(oracle.infer_family(j).numpy(), fx[f"out/family/{node_name}"], **tol)
        if f"out/counts/{node_name}" in fx:
            np.testing.assert_allclose(
                oracle.infer_family(j).mean(dim=0).numpy(), fx[f"out/counts/{node_name}"], **tol
            )
    if observed:
        np.testing.assert_allclose(oracle.log_likelihood(False), float(fx["out/ll_sum"]), rtol=1e-12 if fx.dtype == torch.float64 else 1e-6)
        np.testing.assert_allclose(oracle.log_likelihood(True), float(fx["out/ll_avg"]), rtol=1e-12 if fx.dtype == torch.float64 else 1e-6)
        np.testing.assert_allclose(oracle.local_log_likelihoods().numpy(), fx["out/c"], **tol)
    for key in fx.keys("out/f2v/") + fx.keys("out/v2f/"):
        _, kind, var, owner = key.split("/")
        fn = oracle.message_f2v if kind == "f2v" else oracle.message_v2f
        np.testing.assert_allclose(fn(order.index(var), order.index(owner)).numpy(), fx[key], **tol)


def test_docs1_printed_posteriors():
    # docs/1. Simple Bayesian network.md:70-75
    fx = Fixture("c1_docs1_f64")
    oracle, network, _, _ = make_oracle(fx)
    oracle.enter_evidence(fx.evidence_tensors())
    q = [n.name for n in network.nodes].index("Q")
    expected = np.array([[0.11764706, 0.88235294], [0.8, 0.2], [0.33333333, 0.66666667]])
    np.testing.assert_allclose(oracle.infer_single(q).numpy(), expected, atol=5e-9)


def test_docs2_printed_posteriors_and_ll():
    # docs/2. Complex Bayesian Network.md:909-914 ; first evidence row is the docs' observation
    fx = Fixture("c2_docs2_f64")
    oracle, network, _, _ = make_oracle(fx)
    oracle.enter_evidence(fx.evidence_tensors())
    order = [n.name for n in network.nodes]
    np.testing.assert_allclose(oracle.infer_single(order.index("Q1"))[0].numpy(), [0.78409091, 0.21590909], atol=5e-9)
    np.testing.assert_allclose(oracle.infer_single(order.index("Q2"))[0].numpy(), [0.07954545, 0.92045455], atol=5e-9)
    ll_row0 = oracle.local_log_likelihoods()[:, 0].sum().item()
    assert abs(ll_row0 - (-2.4712404590241857)) < 1e-13


@pytest.mark.parametrize("name", em_fixture_names())
def test_oracle_em_steps(name):
    fx = Fixture(name)
    import json

    em = json.loads(str(fx["em/meta"]))
    ts = TorchSettings(device="cpu", dtype=fx.meta["dtype"])
    network, observed, by_name = fx.build(ts)
    order = [n.name for n in network.nodes]
    cpts = [n.cpt for n in network.nodes]
    evidence = fx.evidence_tensors()
    tol = dict(rtol=1e-10, atol=1e-14) if fx.dtype == torch.float64 else dict(rtol=2e-5, atol=1e-7)
    lls = []
    step = 0
    if em["kind"] == "full":
        batches = [(0, fx.meta["N"])] * em["steps"]
    else:
        b = em["batch"]
        batches = [(0, b), (b, 2 * b)] * em["num_epochs"]
    for lo, hi in batches:
        oracle = SpaOracle(network, observed, hi - lo, dtype=fx.dtype,
                           num_iterations=fx.meta["num_iterations"], allow_loops=fx.meta["allow_loops"], cpts=cpts)
        oracle.enter_evidence([e[lo:hi] for e in evidence])
        lls.append(oracle.log_likelihood(True))
        counts = oracle.expected_counts()
        if em["kind"] == "full":
            cpts = m_step_full(counts)
        else:
            cpts = m_step_batch(cpts, counts, em["learning_rate"], em["regularization"])
        for nm, c in zip(order, cpts):
            np.testing.assert_allclose(c.numpy(), fx[f"em/step{step}/{nm}"], **tol)
        step += 1
    np.testing.assert_allclose(lls, fx["em/ll"], rtol=1e-10 if fx.dtype == torch.float64 else 1e-5)
