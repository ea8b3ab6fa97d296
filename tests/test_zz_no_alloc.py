"""bn_em_statistics_static (the entry point for callers that cannot allocate device memory) against
bn_em_statistics, through the C ABI: two calls in a row on one GPU, the second on a smaller batch,
so that the second call finds the first one's partial sums in the shared static scratch.

Tolerance: the two entry points run the same kernels; only the number of chunks the expected-count
partial sums are cut into may differ (it depends on the room in the scratch), which changes the
order of a double-precision summation: 1e-12 relative on the log-likelihood, 1e-11 relative (+1e-13
of the largest count) on the counts, for float32 and float64 plans alike.
"""

import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _network(ts, leaves_a=40, leaves_b=20):
    """Q(10) with fused (10,2) and (10,3) leaf families, plus a chain R(3) -> M(4) -> Z(5) on generic-sized
    kernels; every fourth leaf unobserved."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node

    gen = torch.Generator().manual_seed(11)

    def cpt(shape):
        p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
        return (p / p.sum(-1, keepdim=True)).to(ts.dtype).to(ts.device)

    q = Node(cpt((10,)), name="Q")
    builder = BayesianNetworkBuilder().add_node(q)
    observed = []
    for i in range(leaves_a):
        node = Node(cpt((10, 2)), name=f"A{i}")
        builder.add_node(node, parents=q)
        if i % 4:
            observed.append(node)
    for i in range(leaves_b):
        node = Node(cpt((10, 3)), name=f"B{i}")
        builder.add_node(node, parents=q)
        if i % 4:
            observed.append(node)
    m = Node(cpt((10, 4)), name="M")
    z = Node(cpt((4, 5)), name="Z")
    builder.add_node(m, parents=q).add_node(z, parents=m)
    observed.append(z)
    return builder.build(), observed


def _evidence(ts, observed, n, seed):
    from bayesiannetwork_amd.inference_machines.evidence import Evidence

    gen = torch.Generator().manual_seed(seed)
    data = []
    for node in observed:
        soft = torch.rand((n, node.num_states), generator=gen, dtype=torch.float64) + 1e-3
        data.append((soft / soft.sum(-1, keepdim=True)).to(ts.dtype).to(ts.device))
    return Evidence(data, ts)


def _statistics(entry, machine, expect=0, warm=False):
    ll = torch.full((1,), float("nan"), device="cuda", dtype=torch.float64)
    counts = torch.full((machine._lib.bn_plan_cpt_elements(machine._plan),), float("nan"), device="cuda",
                        dtype=torch.float64)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    unmapped_before = machine.unmapped_bytes
    rc = entry(machine._plan, machine.num_iterations, 0, ctypes.c_void_p(ll.data_ptr()),
               ctypes.c_void_p(counts.data_ptr()), machine._stream())
    assert rc == expect, machine._lib.bn_last_error().decode()
    torch.cuda.synchronize()
    if entry is machine._lib.bn_em_statistics_static:
        # the call backed no message array, whatever it answered; and once the HIP runtime has loaded
        # the kernels involved (their first launch in a process takes device memory of the runtime's
        # own), it neither takes nor returns device memory
        assert machine.unmapped_bytes == unmapped_before
        if warm:
            assert torch.cuda.mem_get_info()[0] == free_before
    return ll.item(), counts.cpu().numpy()


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_static_scratch_entry_point_equals_bn_em_statistics(dtype):
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_spa import _native
    from bayesiannetwork_amd.torch_settings import TorchSettings

    lib = _native.load()
    ts = TorchSettings(device="cuda", dtype=dtype)
    network, observed = _network(ts)
    settings = SpaInferenceMachineSettings(ts, False)

    last = None
    for n, seed in ((1000, 5), (300, 6)):  # the second call is the smaller one
        evidence = _evidence(ts, observed, n, seed)
        pair = []
        for entry in (lib.bn_em_statistics_static, lib.bn_em_statistics):
            machine = HipSpaInferenceMachine(settings, network, observed, n)
            machine.enter_evidence(evidence)
            if entry is lib.bn_em_statistics_static and machine.unmapped_bytes > 0:
                # This network has nodes on the unfused kernels, whose launches touch message arrays that
                # may have no memory yet.  The static entry point then refuses before launching anything
                # (BN_ERR_STATE) instead of backing them; set-up code backs them once.
                ll_probe = torch.zeros((1,), device="cuda", dtype=torch.float64)
                rc = entry(machine._plan, machine.num_iterations, 0, ctypes.c_void_p(ll_probe.data_ptr()), None,
                           machine._stream())
                if rc != 0:
                    assert rc == -3 and "bn_plan_map_message_arrays" in lib.bn_last_error().decode()
                    assert lib.bn_plan_map_message_arrays(machine._plan) == 0
                    assert machine.unmapped_bytes == 0
                else:
                    machine.reset_messages()  # (forgets the evidence too)
                    machine.enter_evidence(evidence)
            pair.append((machine,) + _statistics(entry, machine))
        (static_machine, ll_s, counts_s), (plain_machine, ll_p, counts_p) = pair
        assert np.isfinite(ll_s) and np.isfinite(counts_s).all()
        assert abs(ll_s - ll_p) <= 1e-12 * abs(ll_p)
        np.testing.assert_allclose(counts_s, counts_p, rtol=1e-11, atol=1e-13 * float(np.abs(counts_p).max()))
        # the state it leaves behind serves later queries like bn_em_statistics' does
        for node in (observed[0], list(network.nodes)[0]):
            a = static_machine.infer_single_nodes([node])[0]
            b = plain_machine.infer_single_nodes([node])[0]
            assert torch.equal(a, b)
        last = static_machine

    # a repeated call neither takes nor returns device memory
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    last.reset_messages()
    last.enter_evidence(evidence)
    _statistics(lib.bn_em_statistics_static, last, warm=True)
    assert torch.cuda.mem_get_info()[0] == free_before


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_static_scratch_entry_point_never_backs_the_lazy_message_arrays(dtype):
    """A plan whose observed families all run on the fused kernels keeps its message arrays without
    memory through any number of statistics calls on reset messages; a call that would need them (a
    second one without a reset: it has a reduced iteration to complete) is refused before anything is
    launched, and served once the caller has backed the arrays."""
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_spa import _native
    from bayesiannetwork_amd.torch_settings import TorchSettings

    lib = _native.load()
    ts = TorchSettings(device="cuda", dtype=dtype)
    gen = torch.Generator().manual_seed(3)

    def cpt(shape):
        p = torch.rand(shape, generator=gen, dtype=torch.float64) + 0.05
        return (p / p.sum(-1, keepdim=True)).to(ts.dtype).to(ts.device)

    q = Node(cpt((10,)), name="Q")
    builder = BayesianNetworkBuilder().add_node(q)
    leaves = []
    for i in range(784):
        leaf = Node(cpt((10, 2)), name=f"Y{i}")
        builder.add_node(leaf, parents=q)
        leaves.append(leaf)
    network = builder.build()
    settings = SpaInferenceMachineSettings(ts, False)
    n = 700
    evidence = _evidence(ts, leaves, n, 9)
    static_machine = HipSpaInferenceMachine(settings, network, leaves, n)
    plain_machine = HipSpaInferenceMachine(settings, network, leaves, n)
    for machine in (static_machine, plain_machine):
        machine.enter_evidence(evidence)
    lazy = static_machine.unmapped_bytes
    assert lazy > 0
    ll_s, counts_s = _statistics(lib.bn_em_statistics_static, static_machine)
    ll_p, counts_p = _statistics(lib.bn_em_statistics, plain_machine)
    assert abs(ll_s - ll_p) <= 1e-12 * abs(ll_p)
    np.testing.assert_allclose(counts_s, counts_p, rtol=1e-11, atol=1e-13 * float(np.abs(counts_p).max()))
    assert static_machine.unmapped_bytes == lazy
    # a second call without a reset would have to complete the reduced iteration into the arrays
    _statistics(lib.bn_em_statistics_static, static_machine, expect=-3, warm=True)
    assert "bn_reset_messages" in lib.bn_last_error().decode()
    assert static_machine.unmapped_bytes == lazy
    # after a reset the plan is as new: served again, still nothing backed
    static_machine.reset_messages()
    static_machine.enter_evidence(evidence)
    ll_again, counts_again = _statistics(lib.bn_em_statistics_static, static_machine, warm=True)
    assert ll_again == ll_s and np.array_equal(counts_again, counts_s)
    assert static_machine.unmapped_bytes == lazy
    # with the arrays backed by the caller, calls that continue from stored messages are served too
    assert lib.bn_plan_map_message_arrays(static_machine._plan) == 0
    assert static_machine.unmapped_bytes == 0
    ll_2, counts_2 = _statistics(lib.bn_em_statistics_static, static_machine)
    ll_2p, counts_2p = _statistics(lib.bn_em_statistics, plain_machine)
    assert abs(ll_2 - ll_2p) <= 1e-12 * abs(ll_2p)
    np.testing.assert_allclose(counts_2, counts_2p, rtol=1e-11, atol=1e-13 * float(np.abs(counts_2p).max()))


def test_static_scratch_entry_point_rejects_networks_beyond_its_scratch():
    from bayesiannetwork_amd.bayesian_network import BayesianNetworkBuilder, Node
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings
    from bayesiannetwork_amd.inference_machines.hip_spa import _native
    from bayesiannetwork_amd.torch_settings import TorchSettings

    lib = _native.load()
    ts = TorchSettings(device="cuda", dtype="float32")
    parents = [Node(torch.full((32,), 1 / 32, device="cuda"), name=f"P{i}") for i in range(3)]
    child = Node(torch.full((32, 32, 32, 32), 1 / 32, device="cuda"), name="C")  # 2^20 CPT entries
    builder = BayesianNetworkBuilder()
    for parent in parents:
        builder.add_node(parent)
    network = builder.add_node(child, parents=parents).build()
    machine = HipSpaInferenceMachine(SpaInferenceMachineSettings(ts, False), network, [child], 2)
    ll = torch.zeros((1,), device="cuda", dtype=torch.float64)
    counts = torch.zeros((lib.bn_plan_cpt_elements(machine._plan),), device="cuda", dtype=torch.float64)
    rc = lib.bn_em_statistics_static(machine._plan, 2, 0, ctypes.c_void_p(ll.data_ptr()),
                                     ctypes.c_void_p(counts.data_ptr()), machine._stream())
    assert rc == -1  # BN_ERR_INVALID
    assert "static scratch" in lib.bn_last_error().decode()
    # without the expected counts only the log-likelihood partial sums need room
    rc = lib.bn_em_statistics_static(machine._plan, 2, 0, ctypes.c_void_p(ll.data_ptr()), None, machine._stream())
    assert rc == 0, lib.bn_last_error().decode()
