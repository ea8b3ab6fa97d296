"""GPU: the EM optimizers driving the HIP machine against the reference's post-step CPTs, and
size-independent properties at the benchmark's full batch size."""

import json

import numpy as np
import pytest
import torch

from conftest import Fixture, em_fixture_names

pytestmark = pytest.mark.gpu


class Loader:
    def __init__(self, batches, total):
        self._batches, self.num_observations = batches, total

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        return iter(self._batches)


def _factory(fx, ts, observed, n):
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    settings = SpaInferenceMachineSettings(ts, True, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])
    return lambda net: HipSpaInferenceMachine(settings, net, observed, n)


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("name", em_fixture_names())
def test_em_optimizers_on_gpu_reproduce_reference(name, deferred):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.optimizers import (
        EmBatchOptimizer,
        EmBatchOptimizerSettings,
        EmOptimizer,
        EmOptimizerSettings,
        OptimizerLogger,
    )
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    em = json.loads(str(fx["em/meta"]))
    if deferred and em["kind"] == "full":
        pytest.skip("deferred logging is a mini-batch optimizer option")
    ts = TorchSettings(device="cuda", dtype=fx.meta["dtype"])
    network, observed, _ = fx.build(ts)
    evidence = Evidence(fx.evidence_tensors("cuda"), ts)
    logger = OptimizerLogger()
    if em["kind"] == "full":
        EmOptimizer(network, _factory(fx, ts, observed, fx.meta["N"]), EmOptimizerSettings(em["steps"]), logger).optimize(
            evidence)
        last = em["steps"] - 1
    else:
        b = em["batch"]
        loader = Loader([evidence[0:b], evidence[b:2 * b]], fx.meta["N"])
        EmBatchOptimizer(network, _factory(fx, ts, observed, b),
                         EmBatchOptimizerSettings(em["learning_rate"], em["num_epochs"], em["regularization"],
                                                  deferred_logging=deferred), logger).optimize(loader)
        last = 2 * em["num_epochs"] - 1
    if fx.dtype == torch.float64:
        for node in network.nodes:
            np.testing.assert_allclose(node.cpt.cpu().numpy(), fx[f"em/step{last}/{node.name}"], err_msg=node.name,
                                       rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose([log.ll for log in logger.logs], fx["em/ll"], rtol=1e-9)
        return
    # float32.  The yardstick is the reference's float64 run of the same optimisation (the sibling
    # fixture: same network, same evidence, same settings), not its float32 run, which carries rounding
    # of its own: the parameters after the last step must be within 1e-5 relative (the tolerance of the
    # reference's own float32 assertions, common/testcase_extensions.py:7-14: five decimal places;
    # 2e-7 absolute covers entries near zero) of that truth, on tree models and loopy ones alike
    # (measured: 1.6e-7 on the trees, 4.2e-6 at worst on the loopy 16-pixel model, where the
    # reference's own float32 run is 8e-6 off; gpurun_out/em_margins.json).
    truth = Fixture(name[:-4] + "_f64")
    for a, b in zip(fx.evidence_tensors(), truth.evidence_tensors()):
        assert torch.equal(a.double(), b.double().float().double()), "the two fixtures describe different runs"
    ours_worst, theirs_worst = 0.0, 0.0
    for node in network.nodes:
        want = truth[f"em/step{last}/{node.name}"].astype(np.float64)
        ours = node.cpt.cpu().numpy().astype(np.float64)
        theirs = fx[f"em/step{last}/{node.name}"].astype(np.float64)
        scale = np.abs(want) + 2e-7 / 1e-5
        ours_worst = max(ours_worst, float((np.abs(ours - want) / scale).max()))
        theirs_worst = max(theirs_worst, float((np.abs(theirs - want) / scale).max()))
    bound = 1e-5
    _record_margin(name + ("/deferred" if deferred else ""), ours_worst, theirs_worst, bound)
    assert ours_worst <= bound, (ours_worst, theirs_worst, bound)
    np.testing.assert_allclose([log.ll for log in logger.logs], truth["em/ll"], rtol=1e-5)


def _record_margin(key, ours, theirs, bound):
    """gpurun_out/em_margins.json: per float32 EM fixture, the largest deviation (relative, with the
    absolute floor folded in) of the engine's and of the reference's float32 parameters from the
    reference's float64 ones, and the bound that was applied."""
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "em_margins.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = dict(ours=ours, reference_float32=theirs, bound=bound)
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("name", ["c3a_784_f32", "c3b_784_f32"])
def test_full_batch_size_properties(name):
    """About 60 000 observations (the benchmark batch) made of the fixture's rows repeated: every
    copy must reproduce the fixture row's answer, marginals must be normalised, the
    log-likelihood must equal the sum of the local normalisers and scale with the copies, and
    EM counts must be the fixture's counts (means are invariant under replication)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    ts = TorchSettings(device="cuda", dtype="float32")
    network, observed, by_name = fx.build(ts)
    base = fx.meta["N"]
    reps = 60000 // base
    n = base * reps
    ev = [e.cuda().repeat(reps, 1).contiguous() for e in fx.evidence_tensors()]
    machine = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=fx.meta["num_iterations"], allow_loops=fx.meta["allow_loops"]),
        network, observed, n)
    machine.enter_evidence(Evidence(ev, ts))

    small = HipSpaInferenceMachine(
        SpaInferenceMachineSettings(ts, False, num_iterations=fx.meta["num_iterations"], allow_loops=fx.meta["allow_loops"]),
        network, observed, base)
    small.enter_evidence(Evidence([e.cuda() for e in fx.evidence_tensors()], ts))

    q = by_name["Q"]
    big_q = machine.infer_single_nodes([q])[0]
    small_q = small.infer_single_nodes([q])[0]
    assert torch.equal(big_q.view(reps, base, -1), small_q.expand(reps, -1, -1)), "replicated rows must be bit-identical"
    assert torch.allclose(big_q.sum(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    y = by_name["Y391"]
    fam = machine.infer_nodes_with_parents([y])[0]
    assert torch.equal(fam[:base], small.infer_nodes_with_parents([y])[0])
    if not fx.meta["allow_loops"]:
        assert torch.allclose(fam.reshape(n, -1).sum(dim=1), torch.ones(n, device="cuda"), atol=1e-5)

    ll_big, ll_small = machine.log_likelihood(), small.log_likelihood()
    np.testing.assert_allclose(ll_big, reps * ll_small, rtol=1e-9)
    c_total = sum(machine.local_log_likelihoods(node).double().sum().item() for node in (q, y))
    assert np.isfinite(c_total)

    counts_big = machine.expected_count_sums() / n
    counts_small = small.expected_count_sums() / base
    np.testing.assert_allclose(counts_big.cpu().numpy(), counts_small.cpu().numpy(), rtol=1e-6, atol=1e-12)
    for node, sums in zip(network.nodes, machine.split_counts(counts_big)):
        key = f"truth/counts/{node.name}"
        if key in fx:
            # loopy iteration amplifies float32 rounding: there the reference's own float32 counts set
            # the scale (off by 1.5e-4 to 3e-4 from the float64 answer, depending on the node)
            rtol = 4e-5
            if fx.meta["allow_loops"]:
                truth = fx[key].astype(np.float64)
                theirs = np.abs(fx[f"out/counts/{node.name}"].astype(np.float64) - truth) / (np.abs(truth) + 1e-7)
                rtol = max(2e-4, 2.0 * float(theirs.max()))
            np.testing.assert_allclose(sums.cpu().numpy(), fx[key], rtol=rtol, atol=1e-7)


@pytest.mark.parametrize("name,n", [("c2_docs2_f32", 1 << 20), ("c4_loops_f32", 1 << 20), ("deep tree", 131072)])
def test_full_batch_size_properties_of_the_other_workloads(name, n):
    """The docs network and the loops network at 2^20 observations and the depth-6 tree of 64-state
    nodes at 131 072 (the sizes they are benchmarked at), made of a few hundred base rows repeated:
    every copy reproduces the base row's marginals bit for bit, marginals are normalised, the
    log-likelihood scales with the copies, the expected-count means are those of the base batch, and
    (fixture networks) the base rows' marginals are the fixture's."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype="float32")
    if name == "deep tree":
        import os
        import sys

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench

        network, observed = bench.build_deep_tree(ts)
        base = 256
        rows = bench.one_hot_evidence(ts, base, observed, 5, "cuda").data
        iterations, loops, fx = None, False, None
        probes = [list(network.nodes)[0], list(network.nodes)[17], observed[3]]
    else:
        fx = Fixture(name)
        network, observed, by_name = fx.build(ts)
        rows = fx.evidence_tensors("cuda")
        base = fx.meta["N"]
        iterations, loops = fx.meta["num_iterations"], fx.meta["allow_loops"]
        probes = list(network.nodes)
    reps = n // base
    n = base * reps
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=iterations, allow_loops=loops)
    small = HipSpaInferenceMachine(settings, network, observed, base)
    small.enter_evidence(Evidence(rows, ts))
    big = HipSpaInferenceMachine(settings, network, observed, n)
    big.enter_evidence(Evidence([r.repeat(reps, 1).contiguous() for r in rows], ts))

    for node, a, b in zip(probes, big.infer_single_nodes(probes), small.infer_single_nodes(probes)):
        assert torch.equal(a.view(reps, base, -1), b.expand(reps, -1, -1)), f"replicated rows of {node.name} differ"
        assert torch.allclose(a.sum(dim=1), torch.ones(n, device="cuda"), atol=2e-5)
        if fx is not None:
            np.testing.assert_allclose(b.cpu().numpy(), fx[f"truth/single/{node.name}"], rtol=4e-5 if loops else 1e-5,
                                       atol=1e-7)
    fam_big = big.infer_nodes_with_parents(probes[-1:])[0]
    fam_small = small.infer_nodes_with_parents(probes[-1:])[0]
    assert torch.equal(fam_big[:base], fam_small) and torch.equal(fam_big[n - base:], fam_small)
    np.testing.assert_allclose(big.log_likelihood(), reps * small.log_likelihood(), rtol=1e-9)
    counts_big = (big.expected_count_sums() / n).cpu().numpy()
    counts_small = (small.expected_count_sums() / base).cpu().numpy()
    np.testing.assert_allclose(counts_big, counts_small, rtol=2e-6, atol=1e-12)
    # the one-call statistics route (what an EM step runs) on the big batch: the same numbers
    again = HipSpaInferenceMachine(settings, network, observed, n)
    ll, sums = again.em_statistics(Evidence([r.repeat(reps, 1).contiguous() for r in rows], ts))
    np.testing.assert_allclose(ll.item(), reps * small.log_likelihood(), rtol=1e-7)
    np.testing.assert_allclose((sums / n).cpu().numpy(), counts_small, rtol=4e-6, atol=1e-12)


@pytest.mark.parametrize("name", ["c3a_784_f32", "c3a_mini16_f32", "c2_docs2_f32", "c3b_784_f32"])
def test_em_statistics_equals_separate_calls_and_leaves_queries_intact(name):
    """bn_em_statistics may skip storing the last iteration's messages; its statistics must equal
    bn_iterate + bn_log_likelihood + bn_em_accumulate, and every query made afterwards must give
    what it gives after a plain bn_iterate (the engine completes the iteration on demand)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    ts = TorchSettings(device="cuda", dtype="float32")
    network, observed, by_name = fx.build(ts)
    settings = SpaInferenceMachineSettings(ts, True, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])
    ev = Evidence(fx.evidence_tensors("cuda"), ts)
    plain = HipSpaInferenceMachine(settings, network, observed, fx.meta["N"])
    plain.enter_evidence(ev)
    plain._iterate()  # bn_iterate: every message of the last iteration stored
    ll_plain = plain.log_likelihood_tensor()
    counts_plain = plain.expected_count_sums()

    # log-likelihood asked first: iterations + ll in one call, nothing of the last iteration stored
    ll_only = HipSpaInferenceMachine(settings, network, observed, fx.meta["N"])
    ll_only.enter_evidence(ev)
    np.testing.assert_allclose(ll_only.log_likelihood_tensor().item(), ll_plain.item(), rtol=1e-7)
    np.testing.assert_allclose(ll_only.expected_count_sums().cpu().numpy(), counts_plain.cpu().numpy(), rtol=2e-6,
                               atol=1e-12)
    for a, b in zip(ll_only.infer_single_nodes(list(network.nodes)), plain.infer_single_nodes(list(network.nodes))):
        assert torch.equal(a, b)

    fused = HipSpaInferenceMachine(settings, network, observed, fx.meta["N"])
    fused.enter_evidence(ev)
    ll_fused, counts_fused = fused.em_statistics()
    np.testing.assert_allclose(ll_fused.item(), ll_plain.item(), rtol=1e-7)
    np.testing.assert_allclose(counts_fused.cpu().numpy(), counts_plain.cpu().numpy(), rtol=2e-6, atol=1e-12)

    nodes = list(network.nodes)
    for a, b in zip(fused.infer_single_nodes(nodes), plain.infer_single_nodes(nodes)):
        assert torch.equal(a, b)
    probe = [nodes[0], nodes[-1]]
    for a, b in zip(fused.infer_nodes_with_parents(probe), plain.infer_nodes_with_parents(probe)):
        assert torch.equal(a, b)
    for node in probe:
        assert torch.equal(fused.local_log_likelihoods(node), plain.local_log_likelihoods(node))
    # and the state keeps evolving identically: a second round of evidence on both
    ev2 = Evidence([torch.flip(e, dims=[0]).contiguous() for e in fx.evidence_tensors("cuda")], ts)
    plain.enter_evidence(ev2)
    plain._iterate()
    fused.enter_evidence(ev2)
    l2, c2 = fused.em_statistics()
    np.testing.assert_allclose(l2.item(), plain.log_likelihood_tensor().item(), rtol=1e-7)
    np.testing.assert_allclose(c2.cpu().numpy(), plain.expected_count_sums().cpu().numpy(), rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n,offset", [(24, 0), (77, 0), (300, 0), (1025, 0), (300, 1)])
def test_em_statistics_with_evidence_equals_enter_evidence_then_statistics(dtype, n, offset):
    """bn_em_statistics_from = bn_enter_evidence + bn_em_statistics: on the 784-leaf model the first
    iteration reads the caller's evidence arrays itself.  Log-likelihood and counts must be bit-identical
    to the two-call route (same arithmetic on the same values), the observation slots must hold the
    evidence bit for bit afterwards, and later queries must agree.  Odd N and N not a multiple of the
    tile exercise the partly filled lanes; offset 1 makes the arrays 8-byte aligned only, which must
    take the two-call route inside the engine (same results)."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture("c3a_784_f64" if dtype == "float64" else "c3a_784_f32")
    ts = TorchSettings(device="cuda", dtype=dtype)
    network, observed, by_name = fx.build(ts)
    gen = torch.Generator().manual_seed(n)
    rows = []
    for node in observed:
        soft = torch.rand((n + offset, 2), generator=gen, dtype=torch.float64) + 1e-3
        rows.append(soft.to(ts.dtype).to("cuda")[offset:])
    assert all(r.is_contiguous() for r in rows)
    ev = Evidence(rows, ts)
    assert (ev.data[0].data_ptr() % 32 == 0) == (offset == 0)
    settings = SpaInferenceMachineSettings(ts, False)

    two_calls = HipSpaInferenceMachine(settings, network, observed, n)
    two_calls.enter_evidence(ev)
    ll_a, counts_a = two_calls.em_statistics()
    one_call = HipSpaInferenceMachine(settings, network, observed, n)
    ll_b, counts_b = one_call.em_statistics(ev)
    assert ll_a.item() == ll_b.item()
    assert torch.equal(counts_a, counts_b)
    assert not one_call.must_iterate
    for node, src in list(zip(observed, ev.data))[::97]:
        assert torch.equal(one_call.observation(node), src), node.name
    probe = [by_name["Q"], observed[0], observed[391], observed[-1]]
    for a, b in zip(one_call.infer_single_nodes(probe), two_calls.infer_single_nodes(probe)):
        assert torch.equal(a, b)
    for a, b in zip(one_call.infer_nodes_with_parents(probe), two_calls.infer_nodes_with_parents(probe)):
        assert torch.equal(a, b)
    # a second batch on the same machines: messages are no longer fresh, both take the copying route
    ev2 = Evidence([torch.flip(r, dims=[0]).contiguous() for r in rows], ts)
    two_calls.enter_evidence(ev2)
    ll_a, counts_a = two_calls.em_statistics()
    ll_b, counts_b = one_call.em_statistics(ev2)
    assert ll_a.item() == ll_b.item() and torch.equal(counts_a, counts_b)


def test_em_statistics_with_evidence_on_networks_without_the_fused_path():
    """Networks whose observed nodes are not all two-state leaves of a fused family (docs network,
    loopy 784-leaf model) and a one-iteration run take the two-call route; results unchanged."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype="float32")
    for name, iterations in (("c2_docs2_f32", None), ("c3b_784_f32", 10), ("c3a_784_f32", 1)):
        fx = Fixture(name)
        network, observed, by_name = fx.build(ts)
        settings = SpaInferenceMachineSettings(ts, True, num_iterations=iterations, allow_loops=fx.meta["allow_loops"])
        ev = Evidence(fx.evidence_tensors("cuda"), ts)
        a = HipSpaInferenceMachine(settings, network, observed, fx.meta["N"])
        a.enter_evidence(ev)
        ll_a, counts_a = a.em_statistics()
        b = HipSpaInferenceMachine(settings, network, observed, fx.meta["N"])
        ll_b, counts_b = b.em_statistics(ev)
        assert ll_a.item() == ll_b.item() and torch.equal(counts_a, counts_b), name
        nodes = list(network.nodes)[:5]
        for x, y in zip(a.infer_single_nodes(nodes), b.infer_single_nodes(nodes)):
            assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n", [24, 77, 300, 1026])
def test_em_statistics_with_hard_evidence_equals_the_two_call_route(dtype, n):
    """bn_em_statistics_from_states = bn_enter_evidence_states + bn_em_statistics.  With an even N the
    first iteration of the 784-leaf model reads the state bytes itself; N = 77 must take the two-call
    route inside the engine.  Results bit-identical either way, the observation slots afterwards hold
    exactly the tensors the hard evidence stands for."""
    from bayesiannetwork_amd.inference_machines.evidence import HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture("c3a_784_f64" if dtype == "float64" else "c3a_784_f32")
    ts = TorchSettings(device="cuda", dtype=dtype)
    network, observed, by_name = fx.build(ts)
    gen = torch.Generator().manual_seed(n)
    states = (torch.rand((len(observed), n), generator=gen) < 0.2).to(torch.uint8).to("cuda")
    ev = HardEvidence(states, [2] * len(observed), ts, on_value=0.9999995, off_value=5e-7)
    settings = SpaInferenceMachineSettings(ts, False)

    two_calls = HipSpaInferenceMachine(settings, network, observed, n)
    two_calls.enter_evidence(ev)
    ll_a, counts_a = two_calls.em_statistics()
    one_call = HipSpaInferenceMachine(settings, network, observed, n)
    ll_b, counts_b = one_call.em_statistics(ev)
    assert ll_a.item() == ll_b.item()
    assert torch.equal(counts_a, counts_b)
    for node, want in list(zip(observed, ev.data))[::131]:
        assert torch.equal(one_call.observation(node), want), node.name
    probe = [by_name["Q"], observed[0], observed[-1]]
    for a, b in zip(one_call.infer_single_nodes(probe), two_calls.infer_single_nodes(probe)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("lr,reg", [(None, None), (0.1, 0.01), (0.3, None), (None, 0.05)])
def test_m_step_kernel_equals_the_tensor_formulas(dtype, lr, reg):
    """bn_m_step against the same M-step written with torch tensor operations (FlatModel.m_step,
    itself checked on the CPU against the per-node reference formulas).  Rows of two states sum in
    one order only, so they must agree bit for bit; longer rows within 1e-15 relative in float64
    (summation order), one float32 ulp otherwise."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype=dtype)
    torch.manual_seed(3)
    q = Node.random((5,), ts, "Q")
    mids = [Node.random((5, 3), ts, f"M{i}") for i in range(3)]
    ys = [Node.random((3, 2), ts, f"Y{i}") for i in range(6)]
    b = BayesianNetworkBuilder().add_node(q).add_nodes(mids, q)
    for i, y in enumerate(ys):
        b.add_node(y, mids[i % 3])
    net = b.build()
    machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, ys, 8)
    model = FlatModel(net)
    old = model.gather()
    counts = torch.rand(old.numel(), dtype=torch.float64, device=old.device) * 7 + 0.01
    want = model.m_step(counts, old, lr, reg)
    got = machine.m_step(counts, old, lr, reg)
    assert got.dtype == want.dtype and got.shape == want.shape
    pos = 0
    for shape in model.shapes:
        size = int(np.prod(shape))
        w, g = want[pos:pos + size], got[pos:pos + size]
        if shape[-1] == 2:
            assert torch.equal(w, g), shape
        else:
            torch.testing.assert_close(g, w, rtol=1e-15 if dtype == "float64" else 1.2e-7, atol=0)
        pos += size
    rows = torch.cat([v.reshape(-1, v.shape[-1]).sum(-1) for v in model.views(got)])
    torch.testing.assert_close(rows, torch.ones_like(rows), rtol=0, atol=1e-6 if dtype == "float32" else 1e-14)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n", [5, 300, 1027])
def test_hard_evidence_fills_the_observation_slots_like_its_tensors(dtype, n):
    """bn_enter_evidence_states: the slots must hold, bit for bit, what entering HardEvidence.data
    through the tensor path puts there, and every result downstream must be identical."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype=dtype)
    torch.manual_seed(11)
    q = Node.random((4,), ts, "Q")
    kids = [Node.random((4, s), ts, f"Y{i}") for i, s in enumerate([2, 2, 3, 5, 2, 7])]
    net = BayesianNetworkBuilder().add_node(q).add_nodes(kids, q).build()
    gen = torch.Generator().manual_seed(n)
    states = torch.stack([torch.randint(0, k.num_states, (n,), generator=gen) for k in kids])
    hard = HardEvidence(states, [k.num_states for k in kids], ts, on_value=0.999999 + 5e-7, off_value=5e-7)

    compact = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, kids, n)
    compact.enter_evidence(hard)
    dense = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, kids, n)
    dense.enter_evidence(Evidence(hard.data, ts))
    for node, want in zip(kids, hard.data):
        assert torch.equal(compact.observation(node), want), node.name
        assert torch.equal(dense.observation(node), want), node.name
    for a, b in zip(compact.infer_single_nodes(list(net.nodes)), dense.infer_single_nodes(list(net.nodes))):
        assert torch.equal(a, b)
    assert compact.log_likelihood() == dense.log_likelihood()
    with pytest.raises(RuntimeError):
        compact.enter_evidence(hard[0:2] if n > 2 else HardEvidence(states[:, :1].repeat(1, n + 1), hard.num_states, ts))


@pytest.mark.parametrize("deferred", [False, True])
def test_prefetching_loader_feeds_the_batch_optimizer_from_host_memory(deferred):
    """Host-resident uint8 batches through PrefetchingLoader give the same EM trajectory as the same
    state bytes resident on the device (bit for bit) and as the float tensors they stand for.  With deferred logging no step waits for the
    host, so a device batch is dropped while the kernels that read it may still be queued: the
    loader has to keep its memory from being reused by a later copy until they have run."""
    import copy
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence, PrefetchingLoader
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers import EmBatchOptimizer, EmBatchOptimizerSettings, OptimizerLogger
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype="float32")
    host = TorchSettings(device="cpu", dtype="float32")
    n, batches = (20000, 12) if deferred else (200, 4)

    def build():
        torch.manual_seed(2)
        q = Node.random((5,), ts, "Q")
        kids = [Node.random((5, 2), ts, f"Y{i}") for i in range(30)]
        return BayesianNetworkBuilder().add_node(q).add_nodes(kids, q).build(), kids

    gen = torch.Generator().manual_seed(9)
    host_batches = [HardEvidence(torch.randint(0, 2, (30, n), generator=gen), [2] * 30, host, 0.9999995, 5e-7)
                    for _ in range(batches)]
    results = []
    for mode in ("prefetched", "resident states", "resident"):
        net, kids = build()
        logger = OptimizerLogger()
        opt = EmBatchOptimizer(
            net, lambda network: HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), network, kids, n),
            EmBatchOptimizerSettings(learning_rate=0.2, num_epochs=2, regularization=0.01,
                                     deferred_logging=deferred), logger)
        if mode == "prefetched":
            loader = PrefetchingLoader(host_batches, "cuda")
        elif mode == "resident states":
            loader = [HardEvidence(b.states.to("cuda"), b.num_states, ts, b.on_value, b.off_value) for b in host_batches]
        else:
            loader = [Evidence([d.to("cuda") for d in b.data], ts) for b in host_batches]
        opt.optimize(loader)
        results.append(([log.ll for log in logger.logs], [node.cpt.clone() for node in net.nodes]))
    # prefetched or resident, the same state bytes take the same route: bit for bit
    assert results[0][0] == results[1][0]
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)
    # the state bytes are read in place with per-family tables, the float tensors multiplied out per
    # observation: the same sums cut differently (2e-6 relative, as for a single step), over 8 / 24 steps
    for ll_a, ll_b in zip(results[0][0], results[2][0]):
        assert abs(ll_a - ll_b) <= 2e-6 * abs(ll_b)
    for a, b in zip(results[0][1], results[2][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-6)


def test_full_batch_em_never_lowers_the_likelihood_on_sampled_data():
    """Property of exact EM on a tree (the 784-leaf model's shape at 60 leaves): on data sampled from a
    ground-truth network with the batched sampler, every full-batch EM iteration raises the
    log-likelihood (up to float32 rounding of the mean)."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.evidence import HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers import EmOptimizer, EmOptimizerSettings, OptimizerLogger
    from bayesiannetwork_amd.samplers import BatchedBayesianNetworkSampler
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype="float32")
    n, leaves = 20000, 60

    def build(seed):
        torch.manual_seed(seed)
        q = Node.random((10,), ts, "Q")
        ys = [Node.random((10, 2), ts, f"Y{i}") for i in range(leaves)]
        return BayesianNetworkBuilder().add_node(q).add_nodes(ys, q).build(), ys

    truth, truth_ys = build(1)
    gen = torch.Generator(device="cuda").manual_seed(3)
    states = BatchedBayesianNetworkSampler(truth, ts, gen).sample(n, truth_ys)  # [n, leaves]
    evidence = HardEvidence(states.t().contiguous(), [2] * leaves, ts, on_value=0.9999995, off_value=5e-7)

    model, ys = build(2)
    logger = OptimizerLogger()
    EmOptimizer(model, lambda net: HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, True), net, ys, n),
                EmOptimizerSettings(num_iterations=8), logger).optimize(evidence)
    lls = [log.ll for log in logger.logs]
    assert len(lls) == 8
    for before, after in zip(lls, lls[1:]):
        assert after >= before - 1e-5 * abs(before), lls
    assert lls[-1] > lls[0] + 0.1


@pytest.mark.parametrize("name", ["c3a_784_f32", "c3b_784_f32", "c3a_784_f64"])
def test_message_arrays_of_fused_families_are_backed_on_first_use(name):
    """A statistics-only run never touches the message arrays of the fused leaf families, so they
    exist as addresses only; the first query backs them with memory and must see exactly what a
    plan that had them from the start sees."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    ts = TorchSettings(device="cuda", dtype=fx.meta["dtype"])
    network, observed, by_name = fx.build(ts)
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])
    n = fx.meta["N"]
    evidence = Evidence([e.cuda() for e in fx.evidence_tensors()], ts)

    lazy = HipSpaInferenceMachine(settings, network, observed, n)
    held_back = lazy.unmapped_bytes
    # all four (six) messages of the 784 families, each states x observations elements
    assert held_back >= 784 * 2 * 4 * 256 * (4 if fx.dtype == torch.float32 else 8) // 2
    ll_lazy, counts_lazy = lazy.em_statistics(evidence)
    torch.cuda.synchronize()
    assert lazy.unmapped_bytes == held_back, "a statistics call must not need the message arrays"

    eager = HipSpaInferenceMachine(settings, network, observed, n)
    eager.enter_evidence(evidence)
    eager.infer_single_nodes([observed[0]])  # a query backs the arrays right away
    assert eager.unmapped_bytes == 0
    eager.reset_messages()  # back to the state of a new machine, arrays in place
    ll_eager, counts_eager = eager.em_statistics(evidence)
    assert ll_lazy.item() == ll_eager.item() and torch.equal(counts_lazy, counts_eager)

    probe = [by_name["Q"], observed[0], observed[391], observed[-1]]
    singles = lazy.infer_single_nodes(probe)
    assert lazy.unmapped_bytes == 0 and lazy.device_bytes >= held_back
    for a, b in zip(singles, eager.infer_single_nodes(probe)):
        assert torch.equal(a, b)
    for a, b in zip(lazy.infer_nodes_with_parents(probe), eager.infer_nodes_with_parents(probe)):
        assert torch.equal(a, b)
    parent = list(network.parents_of(observed[5]))[0]
    assert torch.equal(lazy.message(parent, observed[5], "v2f"), eager.message(parent, observed[5], "v2f"))
    # and a second batch on the now fully backed plan
    ll2_lazy, c2_lazy = lazy.em_statistics(evidence)
    ll2_eager, c2_eager = eager.em_statistics(evidence)
    assert ll2_lazy.item() == ll2_eager.item() and torch.equal(c2_lazy, c2_eager)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3a_784_f32", "c3b_784_f32", "c1_docs1_f32", "c3a_784_f64"])
def test_captured_launch_sequence_equals_kernel_by_kernel(name):
    """Repeated statistics calls on a small batch replay one captured graph; everything they return,
    and everything queried afterwards, is bit-identical to the kernel-by-kernel launches."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    ts = TorchSettings(device="cuda", dtype=fx.meta["dtype"])
    network, observed, _ = fx.build(ts)
    settings = SpaInferenceMachineSettings(ts, False, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])
    n = fx.meta["N"]
    rows = [e.cuda() for e in fx.evidence_tensors()]
    batches = [Evidence(rows, ts), Evidence([r.flip(0).contiguous() for r in rows], ts)]

    def run(graphs):
        machine = HipSpaInferenceMachine(settings, network, observed, n)
        machine.enable_step_graphs(graphs)
        base = machine.step_graph_launches
        out = []
        cpts = None
        for step in range(7):
            machine.reset_messages()
            ll, counts = machine.em_statistics(batches[step % 2])
            out += [ll.clone(), counts.clone()]
            if step == 3:  # a query in between: the plan leaves and re-enters the state the graph was made for
                out += machine.infer_single_nodes([observed[0], list(network.nodes)[0]])
            if step == 4:  # new parameters
                cpts = machine.m_step(counts, None, None, 0.05)
                machine.set_cpts(cpts)
        out += machine.infer_nodes_with_parents([observed[-1]])
        out.append(machine.log_likelihood_tensor().clone())
        launches = machine.step_graph_launches - base
        machine.enable_step_graphs(False)  # the plan goes back to the pool as plans are made
        return out, launches

    captured, launches = run(True)
    plain, none = run(False)
    assert none == 0 and launches >= 4, (none, launches)
    assert len(captured) == len(plain)
    for a, b in zip(captured, plain):
        assert torch.equal(a, b)


@pytest.mark.parametrize("graphs", [False, True])
@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("name", ["c3a_784_f32", "c3b_784_f32", "c2_docs2_f32", "c3a_784_f64"])
def test_one_call_em_step_equals_the_separate_calls(name, hard, graphs):
    """``em_step`` (statistics, M-step and parameter hand-over in one engine call, replayed as a
    captured graph after the second step) against ``em_statistics`` + ``m_step`` + a new machine
    per batch: same log-likelihoods, same parameters after every step, same answers when the machine
    is queried after its step."""
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel
    from bayesiannetwork_amd.torch_settings import TorchSettings

    fx = Fixture(name)
    ts = TorchSettings(device="cuda", dtype=fx.meta["dtype"])
    settings = SpaInferenceMachineSettings(ts, True, num_iterations=fx.meta["num_iterations"],
                                           allow_loops=fx.meta["allow_loops"])
    n = fx.meta["N"]

    def batches_for(observed):
        if hard:
            gen = torch.Generator().manual_seed(5)
            out = []
            for _ in range(2):
                states = torch.stack([torch.randint(0, o.num_states, (n,), generator=gen) for o in observed])
                out.append(HardEvidence(states.cuda(), [o.num_states for o in observed], ts, 1.0 - 1e-6, 1e-6))
            return out
        rows = [e.cuda() for e in fx.evidence_tensors()]
        return [Evidence(rows, ts), Evidence([r.flip(0).contiguous() for r in rows], ts)]

    def run(one_call):
        network, observed, _ = fx.build(ts)
        batches = batches_for(observed)
        model = FlatModel.of(network)
        model.assign(model.gather())
        trace = []
        for step in range(6):
            machine = HipSpaInferenceMachine(settings, network, observed, n)
            machine.enable_step_graphs(graphs and one_call)
            machine.enable_star_step(False)  # one launch per pass: bit-identical to the separate calls
            if one_call:
                ll, new = machine.em_step(batches[step % 2], 0.2, 0.01)
            else:
                ll, counts = machine.em_statistics(batches[step % 2])
                new = machine.m_step(counts, model.gather(), 0.2, 0.01)
            trace += [ll.clone(), new.clone()]
            if step == 3:
                # asked after its step, a machine answers for the batch's evidence under the new parameters
                if not one_call:
                    model.assign(new)
                    machine = HipSpaInferenceMachine(settings, network, observed, n)
                    machine.enter_evidence(batches[step % 2])
                trace += machine.infer_single_nodes([list(network.nodes)[0], observed[0]])
                trace.append(machine.log_likelihood_tensor().clone())
            model.assign(new)
            machine.enable_step_graphs(False)
            machine.enable_star_step(True)
            del machine
        return trace

    one, separate = run(True), run(False)
    assert len(one) == len(separate)
    # State bytes of a float32 plan whose families all sit under one wide parent are read in place by the
    # one-call step (per-family tables, sums by evidence form: kernels_hard.h), and expanded by the
    # separate calls: the same sums cut differently, 2e-6 relative per step as in the headline-model
    # test, over six steps.  Every other combination launches the same kernels: bit for bit.
    in_place = hard and name == "c3a_784_f32"
    for a, b in zip(one, separate):
        if in_place:
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), (a - b).abs().max()
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("n", [1, 64, 100, 129, 1000, 1024, 1025, 2050, 2560, 3072])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("classes", [10, 7])
def test_one_launch_star_step_agrees_with_one_launch_per_pass(classes, dtype, n, hard):
    _star_step_agreement(classes, dtype, n, hard)


def test_streaming_one_launch_step_beyond_the_batch_sizes_it_is_used_for():
    """The streaming one-launch step is used up to 3 072 observations because one launch per pass is
    faster beyond; the kernel itself takes up to 256 slices of 128.  With the limit lifted
    (BN_STREAM_MAX_SLICES, read once per process: hence a process of its own) it must still agree with
    the launch-per-pass route at 4 096, 7 500 and 16 384 observations, on float evidence and on state
    bytes."""
    import os
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, 'tests'); import test_hip_em_and_scale as t\n"
            "for n in (4096, 7500, 16384):\n"
            "    for hard in (False, True):\n"
            "        t._star_step_agreement(10, 'float32', n, hard)\n"
            "t._star_step_agreement(7, 'float64', 7500, False)\n"
            "print('agreed')\n")
    env = dict(os.environ, BN_STREAM_MAX_SLICES="256")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert done.returncode == 0 and "agreed" in done.stdout, done.stdout[-2000:] + done.stderr[-4000:]


def _star_step_agreement(classes, dtype, n, hard):
    """The one-launch EM step of the 784-leaf model against the launch-per-pass route (itself checked
    against the oracle above): log-likelihoods and parameters after each of four steps, to
    rounding.  Up to 1 024 observations the kernel that keeps the evidence in registers, above that
    (to 3 072, where one launch per pass catches up) the one that streams it per iteration (16, 14,
    12 and 10 chunks of families per slice; an odd and a ragged batch size)."""
    import bench
    from bayesiannetwork_amd.inference_machines.evidence import HardEvidence
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.optimizers.em_step import FlatModel
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cuda", dtype=dtype)
    device = torch.device("cuda", 0)
    settings = SpaInferenceMachineSettings(ts, True)

    def run(star):
        network, ys = bench.build_model(ts, "c3a", classes=classes)
        batches = [bench.synthetic_evidence(ts, n, len(ys), 11 + b, device) for b in range(2)]
        if hard:
            batches = [HardEvidence(torch.stack([d.argmax(dim=1) for d in b.data]), [2] * len(ys), ts,
                                    on_value=1.0 - 1e-6, off_value=1e-6) for b in batches]
        model = FlatModel.of(network)
        model.assign(model.gather())
        trace = []
        for step in range(4):
            machine = HipSpaInferenceMachine(settings, network, ys, n)
            machine.enable_star_step(star)
            before = machine.star_step_launches  # (per plan, and machines of one network share plans)
            ll, new = machine.em_step(batches[step % 2], 0.3, 0.01)
            trace.append((ll.clone(), new.clone()))
            # the route under test is the one taken
            assert machine.star_step_launches - before == (1 if star else 0)
            model.assign(new)
            if step == 2:
                # asked after its step, the machine answers for the batch's evidence under the new parameters
                # (the one-launch step leaves no evidence in the plan: the machine enters it again)
                trace.append((machine.log_likelihood_tensor().clone(),
                              machine.infer_single_nodes([list(network.nodes)[0]])[0].reshape(-1).clone()))
            machine.enable_star_step(True)
            del machine
        return trace

    tol = 2e-6 if dtype == "float32" else 1e-11
    for (ll_a, new_a), (ll_b, new_b) in zip(run(True), run(False)):
        assert torch.isfinite(ll_a).all()
        assert abs(ll_a.item() - ll_b.item()) <= tol * abs(ll_b.item())
        # parameters: 10 tol.  The root's posterior of the query (700 or 1000 values, each a normalised
        # product of 784 factors) multiplies the parameters' rounding differences up: 50 tol
        rtol = 10 * tol if new_a.numel() > n * classes else 50 * tol
        np.testing.assert_allclose(new_a.cpu().double().numpy(), new_b.cpu().double().numpy(), rtol=rtol, atol=tol)
