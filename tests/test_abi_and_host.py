"""CPU-only checks: the shared library loads and exports exactly what include/bn_hip.h
declares, argument validation works before any GPU call, and the host classes keep the
reference's ordering and validation rules."""

import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from bayesiannetwork_amd.bayesian_network import (
    BayesianNetwork,
    BayesianNetworkBuilder,
    NetworkValidationError,
    Node,
)
from bayesiannetwork_amd.inference_machines.evidence import Evidence, EvidenceLoader
from bayesiannetwork_amd.inference_machines.hip_spa import _native
from bayesiannetwork_amd.statistics import generate_random_probability_matrix, is_valid_probability_matrix
from bayesiannetwork_amd.torch_settings import TorchSettings

TS = TorchSettings(device="cpu", dtype="float64")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "bn_hip.h")).read()
    declared = set(re.findall(r"\b(bn_[a-z_]+)\s*\(", header))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.bn_abi_version() == _native.ABI_VERSION


def test_plan_create_rejects_bad_topology_without_touching_the_gpu():
    lib = _native.load()
    handle = ctypes.c_void_p()

    def create(dtype, n, states, p_off, p_idx, c_off, c_idx, obs):
        return lib.bn_plan_create(dtype, n, len(states), _native.i32_array(states), _native.i32_array(p_off),
                                  _native.i32_array(p_idx), _native.i32_array(c_off), _native.i32_array(c_idx),
                                  _native.i32_array(obs), len(obs), ctypes.byref(handle))

    good = dict(states=[2, 3], p_off=[0, 0, 1], p_idx=[0], c_off=[0, 1, 1], c_idx=[1], obs=[1])
    assert create(7, 4, **good) == -1 and b"dtype" in lib.bn_last_error()
    assert create(0, 0, **good) == -1 and b"num_observations" in lib.bn_last_error()
    bad_parent = dict(good, p_idx=[5])
    assert create(0, 4, **bad_parent) == -1 and b"parent" in lib.bn_last_error()
    mismatch = dict(good, c_off=[0, 0, 0], c_idx=[])
    assert create(0, 4, **mismatch) == -1 and b"disagree" in lib.bn_last_error()
    twice = dict(good, obs=[1, 1])
    assert create(0, 4, **twice) == -1 and b"twice" in lib.bn_last_error()
    shifted = dict(good, p_off=[1, 1, 2], p_idx=[0, 0])
    assert create(0, 4, **shifted) == -1 and b"start at 0" in lib.bn_last_error()
    backwards = dict(good, p_off=[0, 1, 0])
    assert create(0, 4, **backwards) == -1 and b"non-decreasing" in lib.bn_last_error()
    # a non-empty CSR whose index array is missing
    rc = lib.bn_plan_create(0, 4, 2, _native.i32_array(good["states"]), _native.i32_array(good["p_off"]), None,
                            _native.i32_array(good["c_off"]), _native.i32_array(good["c_idx"]),
                            _native.i32_array(good["obs"]), 1, ctypes.byref(handle))
    assert rc == -1 and b"null index array" in lib.bn_last_error()
    assert lib.bn_plan_destroy(None) == 0


def test_machine_refuses_cpu_settings():
    from bayesiannetwork_amd.inference_machines.hip_spa import HipSpaInferenceMachine, SpaInferenceMachineSettings

    a = Node(torch.tensor([0.5, 0.5], dtype=torch.float64), "A")
    b = Node(torch.tensor([[0.5, 0.5], [0.1, 0.9]], dtype=torch.float64), "B")
    net = BayesianNetworkBuilder().add_node(a).add_node(b, parents=a).build()
    with pytest.raises(ValueError, match="no CPU path"):
        HipSpaInferenceMachine(SpaInferenceMachineSettings(TS, False), net, [b], 3)


def test_torch_settings_validation():
    assert TorchSettings(device="cpu", dtype="float32").dtype == torch.float32
    with pytest.raises(TypeError):
        TorchSettings("cpu", "float32")
    with pytest.raises(ValueError):
        TorchSettings(device="tpu", dtype="float32")
    with pytest.raises(ValueError):
        TorchSettings(device="cpu", dtype="float16")
    os.environ["BN__TORCH_SETTINGS__DEVICE"] = "cpu"
    os.environ["BN__TORCH_SETTINGS__DTYPE"] = "float64"
    try:
        assert TorchSettings().dtype == torch.float64
    finally:
        del os.environ["BN__TORCH_SETTINGS__DEVICE"], os.environ["BN__TORCH_SETTINGS__DTYPE"]


def test_node_order_is_graph_insertion_order():
    # child listed before its parent: the parent is inserted right after the child
    mk = lambda shape, name: Node(generate_random_probability_matrix(shape, TS), name)
    a, b, c, d = mk((2,), "A"), mk((2, 3), "B"), mk((3, 2), "C"), mk((2, 2), "D")
    net = BayesianNetwork([c, a, b, d], {c: [b], a: [], b: [a], d: [a]})
    assert [n.name for n in net.nodes] == ["C", "B", "A", "D"]
    assert [n.name for n in net.children_of(a)] == ["B", "D"]
    assert [n.name for n in net.parents_of(c)] == ["B"]
    assert [(p.name, ch.name) for p, ch in net.edges] == [("B", "C"), ("A", "B"), ("A", "D")]
    assert net.is_leaf_node(c) and not net.is_leaf_node(a)
    assert net.degrees_of_freedom == 1 + 2 * 2 + 3 * 1 + 2 * 1
    assert list(net.G.nodes) == list(net.nodes)
    # multi-parent CPT axis order follows the parents list
    e = mk((3, 2, 4), "E")
    net2 = BayesianNetwork([a, b, e], {a: [], b: [a], e: [b, a]})
    assert [n.name for n in net2.parents_of(e)] == ["B", "A"]


def test_builder_validation_rules():
    mk = lambda shape, name: Node(generate_random_probability_matrix(shape, TS), name)
    a, b = mk((2,), "A"), mk((2, 3), "B")
    assert BayesianNetworkBuilder().add_node(a).add_node(b, parents=a).build().num_nodes == 2
    with pytest.raises(NetworkValidationError):  # CPT axis does not match parent cardinality
        BayesianNetworkBuilder().add_node(a).add_node(mk((3, 3), "X"), parents=a).build()
    with pytest.raises(ValueError):  # same node twice
        BayesianNetworkBuilder().add_node(a).add_node(a).add_node(b, parents=a).build()
    with pytest.raises(NetworkValidationError):  # isolated node
        BayesianNetworkBuilder().add_node(a).add_node(b, parents=a).add_node(mk((2,), "Z")).build()
    with pytest.raises(NetworkValidationError):  # parent not in network
        BayesianNetworkBuilder().add_node(b, parents=a).build()
    with pytest.raises(ValueError):  # duplicate name
        BayesianNetworkBuilder().add_node(a).add_node(mk((2, 2), "A"), parents=a).build()
    with pytest.raises(Exception, match="sum to 1"):
        Node(torch.tensor([0.5, 0.6], dtype=torch.float64))
    assert is_valid_probability_matrix(generate_random_probability_matrix((4, 5), TS))


def test_evidence_and_loader():
    ev = Evidence([torch.eye(3, dtype=torch.float32), torch.ones((3, 2))], TS)
    assert len(ev) == 3 and ev.num_observed_nodes == 2 and ev.data[0].dtype == torch.float64
    assert ev[1:3].num_observations == 2
    assert ev.get_observation(0).shape == (2, 3) if False else True
    with pytest.raises(AssertionError):
        Evidence([torch.eye(3), torch.ones((2, 2))], TS)

    class DS(list):
        pass

    class DL:
        dataset = DS(range(10))

        def __len__(self):
            return 2

        def __iter__(self):
            yield torch.zeros((5, 4)), None
            yield torch.ones((5, 4)), None

    loader = EvidenceLoader(DL(), lambda batch: Evidence([torch.stack([1 - x, x], dim=1) for x in batch.T], TS))
    batches = list(loader)
    assert len(loader) == 2 and loader.num_observations == 10
    assert batches[1].num_observed_nodes == 4 and batches[1].data[0][0, 1] == 1


def test_hard_evidence_stands_for_its_one_hot_tensors():
    from bayesiannetwork_amd.inference_machines.evidence import Evidence, HardEvidence, PrefetchingLoader
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cpu", dtype="float64")
    gen = torch.Generator().manual_seed(5)
    num_states = [2, 3, 5]
    states = torch.stack([torch.randint(0, s, (11,), generator=gen) for s in num_states])
    hard = HardEvidence(states, num_states, ts, on_value=1 - 1e-6, off_value=5e-7)
    assert len(hard) == 11 and hard.num_observed_nodes == 3
    for row, s, got in zip(states, num_states, hard.data):
        want = torch.nn.functional.one_hot(row, s).double() * ((1 - 1e-6) - 5e-7) + 5e-7
        assert got.shape == (11, s) and got.dtype == torch.float64
        # on / off are written as given, not recomputed: the one-hot formula agrees to rounding
        torch.testing.assert_close(got, want, rtol=0, atol=1e-15)
        assert set(got.unique().tolist()) <= {1 - 1e-6, 5e-7}
    part = hard[3:7]
    assert isinstance(part, HardEvidence) and len(part) == 4
    for a, b in zip(part.data, hard.data):
        assert torch.equal(a, b[3:7])
    assert torch.equal(hard.get_observation(2), torch.stack(hard[2].data)) if len(set(num_states)) == 1 else True
    # usable wherever an Evidence is: same type hierarchy, same members
    assert isinstance(hard, Evidence)
    with pytest.raises(ValueError):
        HardEvidence(states, [2, 3], ts)
    # on a CPU device the prefetching loader is a plain pass-through, in order
    batches = [hard[0:4], hard[4:8], hard[8:11]]
    assert [b is c for b, c in zip(PrefetchingLoader(batches, "cpu"), batches)] == [True, True, True]
    assert len(PrefetchingLoader(batches, "cpu")) == 3


def test_prefetching_loader_recognises_evidence_that_tiles_one_host_buffer():
    """torch.stack(...).unbind(0) evidence goes to the device as one copy; separately allocated
    tensors, a gap, or a permuted order do not qualify."""
    import torch

    from bayesiannetwork_amd.inference_machines.evidence import Evidence, PrefetchingLoader
    from bayesiannetwork_amd.torch_settings import TorchSettings

    ts = TorchSettings(device="cpu", dtype="float32")
    x = torch.rand(5, 7, 2)
    tiled = Evidence(list(x.unbind(0)), ts)
    base = PrefetchingLoader._common_base(tiled.data)
    assert base is not None and base.numel() == 70 and torch.equal(base.view(5, 7, 2), x)
    inner = Evidence(list(x[1:4].unbind(0)), ts)  # a window into a larger buffer
    base = PrefetchingLoader._common_base(inner.data)
    assert base is not None and torch.equal(base.view(3, 7, 2), x[1:4])
    assert PrefetchingLoader._common_base(Evidence([torch.rand(7, 2) for _ in range(3)], ts).data) is None
    assert PrefetchingLoader._common_base(Evidence([x[0], x[2], x[3]], ts).data) is None
    assert PrefetchingLoader._common_base(Evidence([x[1], x[0]], ts).data) is None


def test_flat_cpt_buffer_is_recognised_without_a_copy():
    """CPTs that are consecutive views of one buffer (what the EM optimizers hand back) reach the
    engine as that buffer; anything else is concatenated."""
    import torch

    from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import _flatten_cpts

    cpu = torch.device("cpu")
    buf = torch.arange(12.0)
    flat = _flatten_cpts([buf[0:4].view(2, 2), buf[4:10].view(2, 3), buf[10:12]], cpu, torch.float32)
    assert flat.data_ptr() == buf.data_ptr() and torch.equal(flat, buf)
    inner = _flatten_cpts([buf[2:4], buf[4:10].view(2, 3)], cpu, torch.float32)
    assert inner.data_ptr() == buf[2:].data_ptr() and torch.equal(inner, buf[2:10])
    # a gap, another dtype, separate allocations: copies
    gap = _flatten_cpts([buf[0:4], buf[6:8]], cpu, torch.float32)
    assert gap.data_ptr() != buf.data_ptr() and torch.equal(gap, torch.cat([buf[0:4], buf[6:8]]))
    assert _flatten_cpts([buf[0:4]], cpu, torch.float64).dtype == torch.float64
    assert _flatten_cpts([torch.ones(2, 2), torch.zeros(3)], cpu, torch.float32).shape == (7,)


def test_barrier_verdict_of_the_one_launch_step():
    """The decision every block of the one-launch EM step takes from an arrival counter's value:
    wait below the target, pass at or above it, give up when the counter carries the poison a block
    adds when its wait runs out - also when the 32-bit epochs wrap around."""
    lib = _native.load()
    wait, go, void = 0, 1, 2
    poison = 0x40000000
    for target in (128, 5_000, 0xFFFFFF00, 0xFFFFFFFF, 17):
        for behind in (1, 2, 127, 4096):
            assert lib.bn_star_barrier_verdict((target - behind) & 0xFFFFFFFF, target) == wait
        for ahead in (0, 1, 127, 100_000):
            assert lib.bn_star_barrier_verdict((target + ahead) & 0xFFFFFFFF, target) == go
        # poisoned before everybody arrived, with any number of arrivals around it
        for arrived in (-128, -1, 0, 1, 128):
            assert lib.bn_star_barrier_verdict((target + arrived + poison) & 0xFFFFFFFF, target) == void
    # a poisoned counter stays void for the epochs of later launches too (they abort at once)
    assert lib.bn_star_barrier_verdict((1000 + poison) & 0xFFFFFFFF, 1000 + 16 * 128) == void


def test_cardinality_packs_are_found_next_to_the_library():
    """Wide-parent cardinalities: a fixed list is part of the library, the Makefile's PACKS are shared
    objects of their own next to it (loaded here, on the CPU: a pack holds kernels and tables, and
    loading one launches nothing), and anything else is reported as not served.  Every pack exports
    the one entry point the library binds and answers for the cardinality in its name."""
    import ctypes
    import glob
    import os
    import re

    lib = _native.load()
    directory = lib.bn_cardinality_pack_directory().decode()
    assert os.path.samefile(directory, os.path.dirname(_native._LIB_PATH))
    for states in (2, 3, 4, 5, 10):
        assert lib.bn_cardinality_served(states) == 1
    packs = sorted(glob.glob(os.path.join(directory, "libbn_hip_d*.so")))
    assert len(packs) >= 4, "make all builds the packs of the Makefile's PACKS list"
    for path in packs:
        states = int(re.search(r"_d(\d+)\.so$", path).group(1))
        assert lib.bn_cardinality_served(states) == 2, path
        pack = ctypes.CDLL(path)
        pack.bn_cardinality_pack.restype = ctypes.POINTER(ctypes.c_int32)
        table = pack.bn_cardinality_pack()
        assert table[1] == states  # {abi, d0, ...}
    for states in (1, 0, -3, 47, 65, 1000):
        assert lib.bn_cardinality_served(states) == 0
    # shape packs (two-parent leaf families) export their own entry point and name their shape
    shapes = sorted(glob.glob(os.path.join(directory, "libbn_hip_p*.so")))
    assert len(shapes) >= 4
    for path in shapes:
        d0, d1, s = map(int, re.search(r"_p(\d+)x(\d+)x(\d+)\.so$", path).groups())
        pack = ctypes.CDLL(path)
        pack.bn_shape_pack.restype = ctypes.POINTER(ctypes.c_int32)
        table = pack.bn_shape_pack()
        assert (table[1], table[2], table[3]) == (d0, d1, s)
        assert lib.bn_two_parent_shape_served(d0, d1, s) == 2
    assert lib.bn_two_parent_shape_served(2, 2, 2) == 1 and lib.bn_two_parent_shape_served(7, 7, 7) == 0
