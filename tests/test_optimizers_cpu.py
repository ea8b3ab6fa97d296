"""Host-side EM logic on CPU: the optimizers against the reference's post-step CPTs (golden
fixtures), the flat whole-network M-step against the per-node one, evaluator bookkeeping,
and data-parallel EM over two gloo ranks against the single-process run."""

import json
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import Fixture, em_fixture_names
from oracle_machine import OracleMachine
from bayesiannetwork_amd.inference_machines.evidence import Evidence
from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import SpaInferenceMachineSettings
from bayesiannetwork_amd.optimizers import (
    BatchEvaluator,
    EmBatchOptimizer,
    EmBatchOptimizerSettings,
    EmOptimizer,
    EmOptimizerSettings,
    Evaluator,
    OptimizerLogger,
)
from bayesiannetwork_amd.optimizers.em_step import FlatModel
from bayesiannetwork_amd.torch_settings import TorchSettings


class Loader:
    def __init__(self, batches, total):
        self._batches, self.num_observations = batches, total

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        return iter(self._batches)


def _setup(fx, fast_surface):
    ts = TorchSettings(device="cpu", dtype=fx.meta["dtype"])
    network, observed, by_name = fx.build(ts)
    settings = SpaInferenceMachineSettings(ts, True, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])

    def factory_for(n):
        return lambda net: OracleMachine(settings, net, observed, n, with_count_sums=fast_surface)

    return ts, network, observed, factory_for


@pytest.mark.parametrize("fast_surface", [False, True])
@pytest.mark.parametrize("name", em_fixture_names())
def test_optimizers_reproduce_reference_steps(name, fast_surface):
    fx = Fixture(name)
    em = json.loads(str(fx["em/meta"]))
    ts, network, observed, factory_for = _setup(fx, fast_surface)
    evidence = Evidence(fx.evidence_tensors(), ts)
    logger = OptimizerLogger()
    if em["kind"] == "full":
        EmOptimizer(network, factory_for(fx.meta["N"]), EmOptimizerSettings(em["steps"]), logger).optimize(evidence)
        last = em["steps"] - 1
    else:
        b = em["batch"]
        loader = Loader([evidence[0:b], evidence[b:2 * b]], fx.meta["N"])
        EmBatchOptimizer(network, factory_for(b),
                         EmBatchOptimizerSettings(em["learning_rate"], em["num_epochs"], em["regularization"]),
                         logger).optimize(loader)
        last = 2 * em["num_epochs"] - 1
    tol = dict(rtol=1e-9, atol=1e-13) if fx.dtype == torch.float64 else dict(rtol=3e-5, atol=1e-7)
    for node in network.nodes:
        np.testing.assert_allclose(node.cpt.numpy(), fx[f"em/step{last}/{node.name}"], **tol)
    np.testing.assert_allclose([log.ll for log in logger.logs], fx["em/ll"],
                               rtol=1e-9 if fx.dtype == torch.float64 else 1e-5)


def test_flat_m_step_equals_per_node_formulas():
    fx = Fixture("em_batch_c3b_mini16_f64")
    ts = TorchSettings(device="cpu", dtype="float64")
    network, _, _ = fx.build(ts)
    model = FlatModel(network)
    gen = torch.Generator().manual_seed(3)
    sums = torch.rand(sum(model.sizes), generator=gen, dtype=torch.float64) + 0.1
    old = model.gather()
    new = model.m_step(sums, old, 0.1, 0.01)
    for node, s, o, n in zip(model.nodes, model.views(sums), model.views(old), model.views(new)):
        cpt = s / s.sum(dim=-1, keepdim=True)
        cpt = 0.99 * cpt + 0.01 / cpt.shape[-1]
        np.testing.assert_allclose(n.numpy(), (0.9 * o + 0.1 * cpt).numpy(), rtol=1e-14)
    model.assign(new)
    assert all(node.cpt._base is new for node in model.nodes)


def test_evaluators_weight_batches_like_the_reference():
    fx = Fixture("em_batch_docs2_f64")
    ts, network, observed, factory_for = _setup(fx, False)
    evidence = Evidence(fx.evidence_tensors(), ts)
    whole = Evaluator(factory_for(fx.meta["N"]), evidence, lambda e, i: True)
    whole.evaluate(0, 0, network)
    b = fx.meta["N"] // 4
    batched = BatchEvaluator(factory_for(b), Loader([evidence[i * b:(i + 1) * b] for i in range(4)], fx.meta["N"]))
    batched.evaluate(0, 0, network)
    # averaged batch lls weighted by len/num_observations add up to the full-batch average
    np.testing.assert_allclose(batched.log_likelihoods[(0, 0)], whole.log_likelihoods[(0, 0)], rtol=1e-12)


def _sharded_worker(rank, world, port, name, fast_surface, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fx = Fixture(name)
    ts, network, observed, factory_for = _setup(fx, fast_surface)
    evidence = Evidence(fx.evidence_tensors(), ts)
    n = fx.meta["N"]
    per = n // world
    shard = evidence[rank * per:(rank + 1) * per]
    logger = OptimizerLogger()
    EmBatchOptimizer(network, factory_for(per),
                     EmBatchOptimizerSettings(0.1, 2, 0.01, sharded=True), logger).optimize(Loader([shard], per))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ll=np.array([l.ll for l in logger.logs]),
             **{node.name: node.cpt.numpy() for node in network.nodes})
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("fast_surface", [False, True, "step"])
def test_two_rank_sharded_em_equals_single_process(tmp_path, fast_surface):
    name = "em_batch_c3a_mini16_f64"
    port = 29500 + (os.getpid() % 2000) + {False: 0, True: 1, "step": 2}[fast_surface]
    mp.spawn(_sharded_worker, args=(2, port, name, fast_surface, str(tmp_path)), nprocs=2, join=True)

    fx = Fixture(name)
    ts, network, observed, factory_for = _setup(fx, fast_surface)
    evidence = Evidence(fx.evidence_tensors(), ts)
    n = fx.meta["N"]
    logger = OptimizerLogger()
    EmBatchOptimizer(network, factory_for(n), EmBatchOptimizerSettings(0.1, 2, 0.01), logger).optimize(
        Loader([evidence], n))
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for node in network.nodes:
        np.testing.assert_allclose(r0[node.name], node.cpt.numpy(), rtol=1e-11)
        np.testing.assert_array_equal(r0[node.name], r1[node.name])  # replicas stay bit-identical
    np.testing.assert_allclose(r0["ll"], [l.ll for l in logger.logs], rtol=1e-11)


def test_shared_cpt_buffer_follows_assignments_and_direct_writes():
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node

    ts = TorchSettings(device="cpu", dtype="float64")
    q = Node.random((3,), ts, "Q")
    ys = [Node.random((3, 2), ts, f"Y{i}") for i in range(4)]
    net = BayesianNetworkBuilder().add_node(q).add_nodes(ys, q).build()
    before = [n.cpt.clone() for n in net.nodes]

    model = FlatModel(net)
    flat = model.gather()
    model.assign(flat)
    for n, b in zip(net.nodes, before):
        assert torch.equal(n.cpt, b) and n.cpt.shape == b.shape
    assert model.gather() is flat

    flat2 = flat.flip(0).contiguous()
    model.assign(flat2)  # one assignment re-parameterises every node
    assert torch.equal(torch.cat([n.cpt.reshape(-1) for n in net.nodes]), flat2)

    # a direct write to one node takes it out of the shared buffer; the others stay
    ys[1].cpt = before[2]
    assert torch.equal(ys[1].cpt, before[2])
    gathered = model.gather()
    assert gathered is not flat2
    assert torch.equal(gathered, torch.cat([n.cpt.reshape(-1) for n in net.nodes]))
    model.assign(gathered)  # re-attaches all nodes
    assert model.gather() is gathered
