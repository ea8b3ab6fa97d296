"""The CPU restatement of the brute-force machine (oracle/naive_oracle.py) against what the
reference's own NaiveInferenceMachine returned (tests/golden/naive_*.npz, written by
oracle/make_naive_fixtures.py).  Tolerance: float64 1e-12 absolute (probabilities <= 1, same
operation order), float32 1e-6 absolute."""

import numpy as np
import pytest
import torch

from conftest import Fixture, naive_fixture_names
from oracle.naive_oracle import NaiveOracle


def build_oracle(fx, dtype):
    order = fx.meta["network_order"]
    by_parents = dict((n, p) for n, p in fx.meta["nodes"])
    cpts = [torch.from_numpy(fx["cpt/" + n]) for n in order]
    parents = [[order.index(p) for p in by_parents[n]] for n in order]
    oracle = NaiveOracle(cpts, parents, [order.index(o) for o in fx.meta["observed"]], dtype=dtype)
    return oracle, order, parents


@pytest.mark.parametrize("name", naive_fixture_names())
def test_naive_oracle_matches_reference(name):
    fx = Fixture(name)
    oracle, order, parents = build_oracle(fx, fx.dtype)
    oracle.enter_evidence(fx.evidence_tensors())
    tol = 1e-12 if fx.dtype == torch.float64 else 1e-6
    for j, node in enumerate(order):
        np.testing.assert_allclose(oracle.infer([j]).numpy(), fx[f"naive/single/{node}"], rtol=0, atol=tol)
        np.testing.assert_allclose(oracle.infer(parents[j] + [j]).numpy(), fx[f"naive/family/{node}"], rtol=0, atol=tol)
    ll_tol = 1e-12 if fx.dtype == torch.float64 else 1e-5
    np.testing.assert_allclose(oracle.log_likelihoods.numpy(), fx["naive/log_likelihoods"], rtol=ll_tol, atol=ll_tol)
    assert abs(oracle.log_likelihood(False) - float(fx["naive/ll_sum"])) <= ll_tol * abs(float(fx["naive/ll_sum"])) + ll_tol
    assert abs(oracle.log_likelihood(True) - float(fx["naive/ll_avg"])) <= ll_tol * abs(float(fx["naive/ll_avg"])) + ll_tol


def test_naive_oracle_handles_children_listed_before_parents():
    """mixed_group lists a child before its parent; the exact single marginals must still equal
    the (exact on trees) sum-product marginals the reference returned for it."""
    fx = Fixture("mixed_group_f64")
    oracle, order, parents = build_oracle(fx, torch.float64)
    oracle.enter_evidence(fx.evidence_tensors())
    for j, node in enumerate(order):
        np.testing.assert_allclose(oracle.infer([j]).numpy(), fx[f"out/single/{node}"], rtol=1e-9, atol=1e-12)
