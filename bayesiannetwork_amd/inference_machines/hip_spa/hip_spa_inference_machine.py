"""Sum-product inference machine whose arithmetic runs in gfx950 HIP kernels.

Drop-in for the reference's ``SpaInferenceMachine``
(``inference_machines/spa_v3/spa_inference_machine.py:23-137``): same
constructor ``(settings, bayesian_network, observed_nodes, num_observations)``,
same settings dataclass, same five ``IInferenceMachine`` members, same lazy
evaluation (``enter_evidence`` only stores; the first query runs all
iterations), same exceptions.  Everything below the interface — message
buffers, both message passes, normalisers, marginals, EM statistics — lives in
``libbn_hip.so`` behind the C ABI of ``include/bn_hip.h``.
"""

from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from ...bayesian_network import BayesianNetwork, Node
from ..abstractions import IInferenceMachine
from ..common import InferenceMachineSettings
from ..evidence import Evidence, HardEvidence
from . import _native


@dataclass
class SpaInferenceMachineSettings(InferenceMachineSettings):
    num_iterations: Optional[int] = None
    allow_loops: bool = False
    callback: Optional[Callable[[int], None]] = None
    # Not in the reference: stop iterating as soon as no normaliser c[node][n] moved by more than
    # this between two iterations (``num_iterations`` stays the upper bound).  None = always run
    # ``num_iterations`` iterations, as the reference does.
    convergence_tolerance: Optional[float] = None


def _stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _flatten_cpts(cpts, device, dtype) -> torch.Tensor:
    """Concatenate CPT tensors; free when they already are consecutive views of one buffer
    (what the EM optimizers hand back), one ``torch.cat`` otherwise."""
    first = cpts[0]
    if first.device == device and first.dtype == dtype:
        storage = first.untyped_storage()
        start = pos = first.storage_offset()
        for c in cpts:
            if (c.dtype != dtype or c.device != device or not c.is_contiguous() or c.storage_offset() != pos
                    or c.untyped_storage().data_ptr() != storage.data_ptr()):
                break
            pos += c.numel()
        else:
            return torch.empty(0, device=device, dtype=dtype).set_(storage, start, (pos - start,))
    return torch.cat([c.detach().to(device, dtype).reshape(-1) for c in cpts])


class _FactorGraphView:
    """``machine.factor_graph.G``: the undirected variable/factor graph
    (``V_<name>`` / ``F_<name>`` vertices, as ``spa_v3/factor_graph.py:475-506``)."""

    def __init__(self, network: BayesianNetwork):
        self._network = network
        self._G = None

    @property
    def G(self):
        if self._G is None:
            import networkx as nx

            g = nx.Graph()
            for node in self._network.nodes:
                g.add_node(f"V_{node.name}", node=node)
                g.add_node(f"F_{node.name}", node=node)
            for node in self._network.nodes:
                g.add_edge(f"V_{node.name}", f"F_{node.name}")
            for parent, child in self._network.edges:
                g.add_edge(f"V_{parent.name}", f"F_{child.name}")
            self._G = g
        return self._G


def enable_profiling(enabled: bool = True, only: Optional[str] = None, stride: int = 1) -> None:
    """Process-wide: time every kernel launch with HIP events attached to its dispatch (benchmark
    use); ``only`` restricts the timing to kernels whose report label starts with it, ``stride`` to
    every stride-th of those launches."""
    lib = _native.load()
    _native.check(lib.bn_profile_only(only.encode() if only else None))
    _native.check(lib.bn_profile_stride(int(stride)))
    _native.check(lib.bn_profile_enable(1 if enabled else 0))


def profile_report():
    """``[{"kernel", "launches", "total_ms", "alg_bytes"}, ...]`` since the last report."""
    import json

    buf = ctypes.create_string_buffer(1 << 16)
    _native.check(_native.load().bn_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())


class _PlanHandle:
    """Owns one ``bn_plan``; destroys it when the last reference goes away."""

    def __init__(self, lib, handle):
        self.lib = lib
        self.handle = handle
        # the flat CPT tensor whose values the plan's snapshot is known to hold (weak reference and
        # the tensor's version counter at that time), set by ``em_step``
        self.snapshot_of = None
        self.last_stream = None  # (raw handle, torch stream) of the plan's latest call
        self.star_failures_reported = 0  # one-launch EM steps found void that a warning has been given for

    def __del__(self):
        if self.handle:
            try:
                self.lib.bn_plan_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


class _Topology:
    """Everything derived from (network, observed nodes, N, dtype, device) that does not
    depend on CPT values; shared by all machines built for that combination, together
    with a small pool of idle plans.  Optimizers build a new machine every EM step
    (``optimizers/em_batch_optimizer.py:41`` of the reference); with the pool that costs a
    message reset and a CPT upload instead of a plan compilation and a device allocation."""

    MAX_IDLE = 2

    def __init__(self, network: BayesianNetwork, observed_nodes: List[Node]):
        self.nodes = list(network.nodes)
        self.index = {node: i for i, node in enumerate(self.nodes)}
        self.states, self.p_off, self.p_idx, self.c_off, self.c_idx = [], [0], [], [0], []
        for node in self.nodes:
            self.states.append(int(node.num_states))
            self.p_idx.extend(self.index[q] for q in network.parents_of(node))
            self.p_off.append(len(self.p_idx))
            self.c_idx.extend(self.index[q] for q in network.children_of(node))
            self.c_off.append(len(self.c_idx))
        self.observed = [self.index[o] for o in observed_nodes]
        self.cpt_shapes = [tuple(node.cpt.shape) for node in self.nodes]
        self.idle: List[_PlanHandle] = []
        self.is_tree = None
        self.diameter = None


class HipSpaInferenceMachine(IInferenceMachine):
    #: children from which on a node counts as a "wide parent" for the engine (csrc/plan.h: kWideFanIn)
    _WIDE_FAN_IN = 24

    def _serve_wide_cardinalities(self, topo) -> None:
        """With ``BN_AUTO_PACKS=1`` in the environment, a many-children node whose number of states has
        neither kernels in the library nor a cardinality pack gets its pack built before the plan is
        compiled (``make pack``: a minute or two of hipcc, once per cardinality and installation);
        without it such a node runs on the unfused kernels and a warning says what would help."""
        import os
        import warnings

        for i, states in enumerate(topo.states):
            children = topo.c_off[i + 1] - topo.c_off[i]
            if children <= self._WIDE_FAN_IN or not 2 <= states <= 64 or self._lib.bn_cardinality_served(states):
                continue
            if os.environ.get("BN_AUTO_PACKS") == "1":
                _native.build_cardinality_pack(states)
            else:
                warnings.warn(
                    f"node {i} has {children} children and {states} states: no fused kernels for that many states "
                    f"are installed, so it runs on the unfused ones (much slower).  `make pack D0={states}` "
                    f"(or BN_AUTO_PACKS=1) builds them.", RuntimeWarning, stacklevel=3)

    def __init__(
        self,
        settings: SpaInferenceMachineSettings,
        bayesian_network: BayesianNetwork,
        observed_nodes: List[Node],
        num_observations: int,
    ):
        ts = settings.torch_settings
        if ts.device.type != "cuda":
            raise ValueError(
                "HipSpaInferenceMachine runs on an AMD GPU only (TorchSettings(device='cuda')); "
                "there is no CPU path"
            )
        self._lib = _native.load()
        self._settings = settings
        self._network = bayesian_network
        self.num_observations = num_observations
        self.observed_nodes = list(observed_nodes)
        self._device = torch.device("cuda", torch.cuda.current_device()) if ts.device.index is None else ts.device
        self._dtype = ts.dtype
        self._n = max(int(num_observations), 1)

        pool = bayesian_network.__dict__.setdefault("_hip_plan_pool", {})
        key = (tuple(map(id, self.observed_nodes)), self._n, self._dtype, self._device)
        topo = pool.get(key)
        if topo is None:
            topo = pool[key] = _Topology(bayesian_network, self.observed_nodes)
        self._topo = topo
        self._nodes = topo.nodes
        self._index = topo.index
        self._cpt_shapes = topo.cpt_shapes

        if topo.idle:
            self._owner = topo.idle.pop()
            self._plan = self._owner.handle
            with torch.cuda.device(self._device):
                _native.check(self._lib.bn_reset_messages(self._plan, self._stream()))
        else:
            self._serve_wide_cardinalities(topo)
            handle = ctypes.c_void_p()
            with torch.cuda.device(self._device):
                _native.check(
                    self._lib.bn_plan_create(
                        0 if self._dtype == torch.float32 else 1,
                        self._n,
                        len(topo.nodes),
                        _native.i32_array(topo.states),
                        _native.i32_array(topo.p_off),
                        _native.i32_array(topo.p_idx),
                        _native.i32_array(topo.c_off),
                        _native.i32_array(topo.c_idx),
                        _native.i32_array(topo.observed),
                        len(topo.observed),
                        ctypes.byref(handle),
                    )
                )
            self._owner = _PlanHandle(self._lib, handle)
            self._plan = handle
            if topo.is_tree is None:
                is_tree, diameter = ctypes.c_int32(), ctypes.c_int32()
                _native.check(self._lib.bn_plan_topology(self._plan, ctypes.byref(is_tree), ctypes.byref(diameter)))
                topo.is_tree, topo.diameter = bool(is_tree.value), diameter.value

        if topo.is_tree:
            self._num_iterations = settings.num_iterations if settings.num_iterations else topo.diameter
        else:
            if not settings.allow_loops:
                raise ValueError("Network contains loops, but the specified settings disallow them.")
            if not settings.num_iterations:
                raise ValueError("Networks containers loops; number of iterations must be set explicitly.")
            self._num_iterations = settings.num_iterations

        store = getattr(self._nodes[0], "_store", None) if self._nodes else None
        if (store is not None and store.network is bayesian_network and store.complete
                and store.count == len(self._nodes)):
            known = self._owner.snapshot_of
            if known is None or known[0]() is not store.flat or known[1] != store.flat._version:
                self.set_cpts(store.flat)  # all CPTs already sit back to back in network order
        else:
            self.set_cpts([node.cpt for node in self._nodes])
        self.factor_graph = _FactorGraphView(bayesian_network)
        self.must_iterate: bool = True
        self.iterations_run = 0
        self._evidence_keepalive = None
        self._evidence_to_restore = None

    def __del__(self):
        owner = getattr(self, "_owner", None)
        topo = getattr(self, "_topo", None)
        if owner is not None and topo is not None and len(topo.idle) < _Topology.MAX_IDLE:
            topo.idle.append(owner)
        self._owner = None
        self._plan = None

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        """The current torch stream as the engine's ``void*``.  A plan's calls are ordered by the stream
        they are issued on; when the stream changes between two calls on one plan (a pooled plan taken
        over by a machine that works on another stream, a ``torch.cuda.stream`` block), the new stream
        first waits for everything the old one was given."""
        stream = torch.cuda.current_stream(self._device)
        handle = stream.cuda_stream
        last = self._owner.last_stream
        if last is not None and last[0] != handle:
            stream.wait_stream(last[1])
        if last is None or last[0] != handle:
            self._owner.last_stream = (handle, stream)
        return ctypes.c_void_p(handle)

    def _restore_evidence(self) -> None:
        """After ``em_step`` the engine holds new parameters and no evidence; the batch's evidence is
        entered again when something is asked of this machine afterwards."""
        evidence, self._evidence_to_restore = self._evidence_to_restore, None
        if evidence is not None:
            self.enter_evidence(evidence)

    def set_cpts(self, cpts) -> None:
        """Snapshot CPTs into the plan: one tensor per node in network order, or one flat
        tensor holding their concatenation."""
        if isinstance(cpts, torch.Tensor):
            flat = cpts.detach().to(self._device, self._dtype).contiguous().reshape(-1)
        else:
            flat = _flatten_cpts(cpts, self._device, self._dtype)
        if flat.numel() != self._lib.bn_plan_cpt_elements(self._plan):
            raise ValueError("CPT sizes do not match the network the machine was built for")
        self._owner.snapshot_of = None
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_set_cpts(self._plan, ctypes.c_void_p(flat.data_ptr()), self._stream()))
        if isinstance(cpts, torch.Tensor) and flat.data_ptr() == cpts.data_ptr() and flat.numel() == cpts.numel():
            # the caller's own buffer: the plan now holds its values as of this version
            self._owner.snapshot_of = (weakref.ref(cpts), cpts._version)
        self._cpt_keepalive = flat
        self.must_iterate = True

    @property
    def settings(self) -> InferenceMachineSettings:
        return self._settings

    @property
    def num_iterations(self) -> int:
        return self._num_iterations

    # ------------------------------------------------------------------ interface
    def enter_evidence(self, evidence: Evidence) -> None:
        if isinstance(evidence, HardEvidence):
            return self._enter_hard_evidence(evidence)
        cached = self._evidence_table(evidence)
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_enter_evidence(self._plan, cached[2], self._stream()))
        self._evidence_keepalive = cached
        self.must_iterate = True

    def _evidence_table(self, evidence: Evidence):
        """``(key, tensors, table of their device addresses)`` for the engine, cached on the
        evidence object; tensors are checked against the observed nodes."""
        key = (self._device, self._dtype, self._n)
        cached = getattr(evidence, "_hip_table", None)
        if cached is None or cached[0] != key:
            tensors = []
            for t in evidence.data:
                if t.device != self._device or t.dtype != self._dtype or not t.is_contiguous():
                    t = t.detach().to(self._device, self._dtype).contiguous()
                tensors.append(t)
            if len(tensors) != len(self.observed_nodes):
                raise ValueError("evidence must hold one tensor per observed node")
            for t, node in zip(tensors, self.observed_nodes):
                if tuple(t.shape) != (self._n, node.num_states):
                    raise RuntimeError(
                        f"evidence for node {node} has shape {tuple(t.shape)}, "
                        f"expected {(self._n, node.num_states)}"
                    )
            table = (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])
            cached = (key, tensors, table)
            try:
                evidence._hip_table = cached
            except AttributeError:
                pass
        return cached

    def _enter_hard_evidence(self, evidence: HardEvidence) -> None:
        """State indices go to the engine as they are (one byte per observed node and observation);
        the observation slots end up holding exactly ``evidence.data``."""
        states = self._hard_states(evidence)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_enter_evidence_states(self._plan, ctypes.c_void_p(states.data_ptr()),
                                                   evidence.on_value, evidence.off_value, self._stream())
            )
        self._evidence_keepalive = states
        self.must_iterate = True

    def _hard_states(self, evidence: HardEvidence) -> torch.Tensor:
        """The evidence's state matrix on this machine's device, checked against the observed nodes."""
        states = evidence.states
        if states.device != self._device:
            states = states.to(self._device)
        if tuple(states.shape) != (len(self.observed_nodes), self._n):
            raise RuntimeError(f"hard evidence has shape {tuple(states.shape)}, expected "
                               f"{(len(self.observed_nodes), self._n)}")
        if evidence.num_states != [int(node.num_states) for node in self.observed_nodes]:
            raise RuntimeError("hard evidence was built for other state counts than the observed nodes have")
        if not states.is_contiguous():
            states = states.contiguous()
        return states

    def _iterate(self) -> None:
        self._restore_evidence()
        callback = self._settings.callback
        tolerance = self._settings.convergence_tolerance
        self.iterations_run = 0
        # queries made from inside the callback see the state after the iterations run so far
        self.must_iterate = False
        try:
            with torch.cuda.device(self._device):
                if callback or tolerance is not None:
                    for iteration in range(self._num_iterations):
                        _native.check(self._lib.bn_iterate(self._plan, 1, self._stream()))
                        self.iterations_run = iteration + 1
                        if callback:
                            callback(iteration)
                        if tolerance is not None and self.normaliser_change() <= tolerance:
                            break
                else:
                    _native.check(self._lib.bn_iterate(self._plan, self._num_iterations, self._stream()))
                    self.iterations_run = self._num_iterations
        except BaseException:
            self.must_iterate = True
            raise

    def normaliser_change(self) -> float:
        """Largest change of any node's local log-likelihood term since this was last asked (since
        the reset for the first call): the convergence monitor for loopy networks."""
        out = torch.empty((1,), device=self._device, dtype=torch.float64)
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_normaliser_change(self._plan, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out.item()

    def infer_single_nodes(self, nodes: List[Node]) -> List[torch.Tensor]:
        if self.must_iterate:
            self._iterate()
        out = []
        with torch.cuda.device(self._device):
            for node in nodes:
                p = torch.empty((self._n, node.num_states), device=self._device, dtype=self._dtype)
                _native.check(
                    self._lib.bn_marginal_single(self._plan, self._index[node], ctypes.c_void_p(p.data_ptr()),
                                                 self._stream())
                )
                out.append(p)
        return out

    def infer_nodes_with_parents(self, child_nodes: List[Node]) -> List[torch.Tensor]:
        if self.must_iterate:
            self._iterate()
        out = []
        with torch.cuda.device(self._device):
            for node in child_nodes:
                j = self._index[node]
                p = torch.empty((self._n,) + self._cpt_shapes[j], device=self._device, dtype=self._dtype)
                _native.check(
                    self._lib.bn_marginal_family(self._plan, j, ctypes.c_void_p(p.data_ptr()), self._stream())
                )
                out.append(p)
        return out

    def log_likelihood(self) -> float:
        if not self.observed_nodes:
            raise Exception("Log likelihood can't be calculated with 0 observed nodes")
        return self.log_likelihood_tensor().item()

    # ------------------------------------------------------------------ extensions
    def log_likelihood_tensor(self) -> torch.Tensor:
        """Log-likelihood as a one-element float64 device tensor (no host sync)."""
        self._restore_evidence()
        out = torch.empty((1,), device=self._device, dtype=torch.float64)
        if self.must_iterate and not self._settings.callback and self._settings.convergence_tolerance is None:
            # iterations and log-likelihood in one engine call; the last iteration's messages are
            # completed by the engine if a later query needs them
            with torch.cuda.device(self._device):
                _native.check(
                    self._lib.bn_em_statistics(self._plan, self._num_iterations,
                                               1 if self._settings.average_log_likelihood else 0,
                                               ctypes.c_void_p(out.data_ptr()), None, self._stream())
                )
            self.must_iterate = False
            self.iterations_run = self._num_iterations
            return out
        if self.must_iterate:
            self._iterate()
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_log_likelihood(self._plan, 1 if self._settings.average_log_likelihood else 0,
                                            ctypes.c_void_p(out.data_ptr()), self._stream())
            )
        return out

    def expected_count_sums(self) -> torch.Tensor:
        """EM statistic: for every node the sum over observations of its family marginal, as one
        flat float64 device tensor laid out like the concatenated CPTs (network order).
        ``split_counts`` cuts it into CPT-shaped views."""
        if self.must_iterate:
            self._iterate()
        total = self._lib.bn_plan_cpt_elements(self._plan)
        out = torch.empty((total,), device=self._device, dtype=torch.float64)
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_em_accumulate(self._plan, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def em_statistics(self, evidence: Optional[Evidence] = None):
        """``(log_likelihood_tensor, expected_count_sums)`` of the entered evidence in one engine call.
        When the iterations have not run yet (and no per-iteration callback is set) the last
        iteration is evaluated on the fly for the statistics; message queries made afterwards
        bring the stored messages up to date first, so results are the same either way.

        With ``evidence`` the call is ``enter_evidence(evidence)`` followed by ``em_statistics()``;
        the engine then gets both in one call and may read the evidence while it runs the first
        iteration instead of copying it first (same results, same state afterwards)."""
        if evidence is None:
            self._restore_evidence()
        else:
            self._evidence_to_restore = None
        plain = self._settings.callback is None and self._settings.convergence_tolerance is None
        if evidence is not None and not plain:
            self.enter_evidence(evidence)
            evidence = None
        if evidence is None and (not self.must_iterate or not plain):
            return self.log_likelihood_tensor(), self.expected_count_sums()
        ll = torch.empty((1,), device=self._device, dtype=torch.float64)
        counts = torch.empty((self._lib.bn_plan_cpt_elements(self._plan),), device=self._device, dtype=torch.float64)
        average = 1 if self._settings.average_log_likelihood else 0
        with torch.cuda.device(self._device):
            if isinstance(evidence, HardEvidence):
                states = self._hard_states(evidence)
                _native.check(
                    self._lib.bn_em_statistics_from_states(
                        self._plan, ctypes.c_void_p(states.data_ptr()), evidence.on_value, evidence.off_value,
                        self._num_iterations, average, ctypes.c_void_p(ll.data_ptr()),
                        ctypes.c_void_p(counts.data_ptr()), self._stream())
                )
                self._evidence_keepalive = states
            elif evidence is not None:
                cached = self._evidence_table(evidence)
                _native.check(
                    self._lib.bn_em_statistics_from(self._plan, cached[2], self._num_iterations, average,
                                                    ctypes.c_void_p(ll.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                                    self._stream())
                )
                self._evidence_keepalive = cached
            else:
                _native.check(
                    self._lib.bn_em_statistics(self._plan, self._num_iterations, average,
                                               ctypes.c_void_p(ll.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                               self._stream())
                )
        self.must_iterate = False
        self.iterations_run = self._num_iterations
        return ll, counts

    @property
    def can_em_step(self) -> bool:
        return self._settings.callback is None and self._settings.convergence_tolerance is None

    def em_step(self, evidence: Evidence, learning_rate: Optional[float], regularization: Optional[float]):
        """One mini-batch EM step in a single engine call: ``em_statistics(evidence)`` followed by
        ``m_step`` with the engine's current parameters as the old ones.  Returns
        ``(log_likelihood_tensor, new flat CPT buffer)``; the engine then already holds the new
        parameters (a machine built next from a network that carries exactly this buffer does not
        copy them again).  The reference's loop body, ``optimizers/em_batch_optimizer.py:37-59``."""
        if not self.can_em_step:
            raise RuntimeError("em_step needs plain settings (no callback, no convergence tolerance)")
        total = self._lib.bn_plan_cpt_elements(self._plan)
        ll = torch.empty((1,), device=self._device, dtype=torch.float64)
        new = torch.empty((total,), device=self._device, dtype=self._dtype)
        average = 1 if self._settings.average_log_likelihood else 0
        rate = -1.0 if learning_rate is None else float(learning_rate)
        gamma = float(regularization or 0.0)
        with torch.cuda.device(self._device):
            if isinstance(evidence, HardEvidence):
                states = self._hard_states(evidence)
                keep = states
                rc = self._lib.bn_em_step(self._plan, None, ctypes.c_void_p(states.data_ptr()), evidence.on_value,
                                          evidence.off_value, self._num_iterations, average, rate, gamma,
                                          ctypes.c_void_p(ll.data_ptr()), ctypes.c_void_p(new.data_ptr()), self._stream())
            else:
                keep = self._evidence_table(evidence)
                rc = self._lib.bn_em_step(self._plan, keep[2], None, 1.0, 0.0, self._num_iterations, average, rate,
                                          gamma, ctypes.c_void_p(ll.data_ptr()), ctypes.c_void_p(new.data_ptr()),
                                          self._stream())
            _native.check(rc)
        self._evidence_keepalive = keep
        self._owner.snapshot_of = (weakref.ref(new), new._version)
        self._evidence_to_restore = evidence
        self.must_iterate = True
        self.iterations_run = self._num_iterations
        self.report_void_steps()  # (of earlier calls: this one may still be running)
        return ll, new

    def report_void_steps(self, synchronize: bool = False) -> int:
        """One-launch EM steps of this machine's plan that were found void so far (their blocks could
        not meet at a barrier because the GPU was shared with other work).  Such a step changes
        nothing: the parameters it returns are the old ones, its log-likelihood is NaN, and the
        engine launches per pass from then on.  A ``RuntimeWarning`` is given once per failure.
        ``synchronize=True`` waits for the plan's stream first, which makes the answer include the
        step that was submitted last."""
        if synchronize:
            torch.cuda.current_stream(self._device).synchronize()
        failures = int(self._lib.bn_plan_star_step_failures(self._plan))
        if failures > self._owner.star_failures_reported:
            import warnings

            warnings.warn(
                f"{failures - self._owner.star_failures_reported} one-launch EM step(s) could not synchronise their "
                "blocks (GPU shared with other work) and were skipped: parameters unchanged, log-likelihood NaN; "
                "the plan launches one kernel per pass from now on", RuntimeWarning, stacklevel=2)
            self._owner.star_failures_reported = failures
        return failures

    def em_step_statistics(self, evidence: Evidence, out: Optional[torch.Tensor] = None):
        """The statistics half of ``em_step`` for runs that add them up over several devices before the
        M-step: ``(log_likelihood_tensor, expected_count_sums)`` of ``evidence`` on fresh messages,
        the same values as ``em_statistics(evidence)`` on a new machine.  The engine keeps no copy
        of the evidence (it is reset when the call ends; the machine re-enters the batch if it is
        queried afterwards).  ``out``: a float64 buffer of at least ``cpt elements + 1`` entries to
        write into (count sums first, then the log-likelihood), e.g. the front of the buffer that
        is all-reduced next, so that nothing has to be copied there."""
        if not self.can_em_step:
            raise RuntimeError("em_step_statistics needs plain settings (no callback, no convergence tolerance)")
        total = self._lib.bn_plan_cpt_elements(self._plan)
        if out is None:
            out = torch.empty((total + 1,), device=self._device, dtype=torch.float64)
        if out.dtype != torch.float64 or out.device != self._device or not out.is_contiguous() or out.numel() < total + 1:
            raise ValueError("out must be a contiguous float64 tensor on the machine's device with cpt elements + 1 entries")
        counts, ll = out[:total], out[total : total + 1]
        average = 1 if self._settings.average_log_likelihood else 0
        with torch.cuda.device(self._device):
            if isinstance(evidence, HardEvidence):
                states = self._hard_states(evidence)
                keep = states
                rc = self._lib.bn_em_step_statistics(self._plan, None, ctypes.c_void_p(states.data_ptr()),
                                                     evidence.on_value, evidence.off_value, self._num_iterations,
                                                     average, ctypes.c_void_p(ll.data_ptr()),
                                                     ctypes.c_void_p(counts.data_ptr()), self._stream())
            else:
                keep = self._evidence_table(evidence)
                rc = self._lib.bn_em_step_statistics(self._plan, keep[2], None, 1.0, 0.0, self._num_iterations, average,
                                                     ctypes.c_void_p(ll.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                                     self._stream())
            _native.check(rc)
        self._evidence_keepalive = keep
        self._evidence_to_restore = evidence
        self.must_iterate = True
        self.iterations_run = self._num_iterations
        return ll, counts

    def m_step(self, count_sums: torch.Tensor, old_flat: Optional[torch.Tensor],
               learning_rate: Optional[float], regularization: Optional[float]) -> torch.Tensor:
        """New flat CPT buffer (network order, machine dtype) from flat float64 expected counts:
        rows normalised, optionally blended with the uniform distribution (``regularization``) and
        with ``old_flat`` (``learning_rate``) — the reference's ``_m_step`` formulas
        (``optimizers/em_batch_optimizer.py:70-84``, ``optimizers/em_optimizer.py:57-63``) for all
        nodes in one kernel."""
        total = self._lib.bn_plan_cpt_elements(self._plan)
        counts = count_sums.detach().to(self._device, torch.float64).contiguous().reshape(-1)
        if counts.numel() != total:
            raise ValueError("count_sums does not match the network the machine was built for")
        old_ptr = None
        if learning_rate is not None:
            old = old_flat.detach().to(self._device, self._dtype).contiguous().reshape(-1)
            if old.numel() != total:
                raise ValueError("old_flat does not match the network the machine was built for")
            old_ptr = ctypes.c_void_p(old.data_ptr())
        out = torch.empty((total,), device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_m_step(self._plan, ctypes.c_void_p(counts.data_ptr()), old_ptr,
                                    -1.0 if learning_rate is None else float(learning_rate),
                                    float(regularization or 0.0), ctypes.c_void_p(out.data_ptr()), self._stream())
            )
        return out

    def split_counts(self, flat: torch.Tensor) -> List[torch.Tensor]:
        views, pos = [], 0
        for shape in self._cpt_shapes:
            size = 1
            for d in shape:
                size *= d
            views.append(flat[pos : pos + size].reshape(shape))
            pos += size
        return views

    def local_log_likelihoods(self, node: Node) -> torch.Tensor:
        if self.must_iterate:
            self._iterate()
        out = torch.empty((self._n,), device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_local_log_likelihoods(self._plan, self._index[node], ctypes.c_void_p(out.data_ptr()),
                                                   self._stream())
            )
        return out

    def message(self, variable: Node, factor_owner: Node, direction: str) -> torch.Tensor:
        """``[N, S]`` copy of the message on the edge between ``variable`` and the factor of
        ``factor_owner``; ``direction`` is ``"v2f"`` or ``"f2v"``.  Does not trigger iteration."""
        owner = self._index[factor_owner]
        if variable is factor_owner:
            slot = -1
        else:
            slot = list(self._network.parents_of(factor_owner)).index(variable)
        out = torch.empty((self._n, variable.num_states), device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_get_message(self._plan, owner, slot, 0 if direction == "v2f" else 1,
                                         ctypes.c_void_p(out.data_ptr()), self._stream())
            )
        return out

    def observation(self, node: Node) -> torch.Tensor:
        """``[N, S]`` copy of what the observation slot of an observed node currently holds."""
        out = torch.empty((self._n, node.num_states), device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_get_message(self._plan, self._index[node], -2, 0, ctypes.c_void_p(out.data_ptr()),
                                         self._stream())
            )
        return out

    def reset_messages(self) -> None:
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_reset_messages(self._plan, self._stream()))
        self.must_iterate = True

    @property
    def device_bytes(self) -> int:
        return self._lib.bn_plan_device_bytes(self._plan)

    @property
    def unmapped_bytes(self) -> int:
        """Bytes of message arrays that exist as addresses only (no query has needed them yet)."""
        return self._lib.bn_plan_unmapped_bytes(self._plan)

    @property
    def step_graph_launches(self) -> int:
        """How often the engine ran an ``em_statistics`` launch sequence as one captured graph."""
        return self._lib.bn_plan_step_graph_launches(self._plan)

    def enable_step_graphs(self, enabled: bool = True) -> None:
        """Small batches replay repeated ``em_statistics`` calls as a captured graph; ``False`` makes
        the engine launch kernel by kernel (same results)."""
        _native.check(self._lib.bn_plan_enable_step_graphs(self._plan, 1 if enabled else 0))

    @property
    def star_step_launches(self) -> int:
        """How many ``em_step`` calls of this machine ran as one cooperative launch."""
        return self._lib.bn_plan_star_step_launches(self._plan)

    def enable_evidence_compaction(self, enabled: bool = True) -> None:
        """``em_step`` / ``em_step_statistics`` on float evidence whose rows are two rows only -
        one-hot evidence and its clamped variants - run on one state byte per leaf and observation
        from batches of 28 672 observations on (results agree with the float route to rounding);
        ``False`` makes the engine read the float arrays in every launch."""
        _native.check(self._lib.bn_plan_enable_evidence_compaction(self._plan, 1 if enabled else 0))

    def evidence_compactions(self) -> int:
        """How many whole-step calls of this machine ran on state bytes made from float evidence."""
        return self._lib.bn_plan_evidence_compactions(self._plan)

    def enable_star_step(self, enabled: bool = True) -> None:
        """``em_step`` on a small batch of a star network (one root, two-state leaves) is one
        cooperative launch; ``False`` makes the engine take one launch per pass (results agree to
        rounding)."""
        _native.check(self._lib.bn_plan_enable_star_step(self._plan, 1 if enabled else 0))
