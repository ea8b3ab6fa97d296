"""Exact inference by enumeration of the joint distribution, on the GPU.

Drop-in for the reference's ``NaiveInferenceMachine``
(``inference_machines/naive/naive_inference_machine.py:11-122``): same constructor
``(settings, bayesian_network, observed_nodes)``, same ``IInferenceMachine`` members, same
exceptions, results normalised by the probability of the evidence.  It exists for the same reason
as there: a yardstick for the sum-product machine on small networks (exact on any topology,
exponential in the number of nodes).  The arithmetic runs in ``libbn_hip.so``
(``bn_exact_*`` in ``include/bn_hip.h``); there is no CPU path.

Two deliberate differences from the reference, both documented where they occur:

* the reference reshapes every CPT assuming the node's parents precede it in ascending network
  order (``:45-55``) and so returns a wrong joint for other orders; here CPT axes are matched to
  the parents by identity;
* sums are carried in float64 also for a float32 machine.
"""

from __future__ import annotations

import ctypes
from typing import List

import torch

from ...bayesian_network import BayesianNetwork, Node
from ..abstractions import IInferenceMachine
from ..common import InferenceMachineSettings
from ..evidence import Evidence
from ..hip_spa import _native


class HipNaiveInferenceMachine(IInferenceMachine):
    def __init__(
        self,
        settings: InferenceMachineSettings,
        bayesian_network: BayesianNetwork,
        observed_nodes: List[Node],
    ):
        ts = settings.torch_settings
        if ts.device.type != "cuda":
            raise ValueError(
                "HipNaiveInferenceMachine runs on an AMD GPU only (TorchSettings(device='cuda')); "
                "there is no CPU path"
            )
        self._lib = _native.load()
        self._settings = settings
        self._device = torch.device("cuda", torch.cuda.current_device()) if ts.device.index is None else ts.device
        self._dtype = ts.dtype
        self.bayesian_network = bayesian_network
        self._nodes = list(bayesian_network.nodes)
        self.dims = [int(node.num_states) for node in self._nodes]
        self.num_nodes = len(self._nodes)
        self.node_to_index = {node: i for i, node in enumerate(self._nodes)}
        self.num_observed_nodes = len(observed_nodes)
        self.observed_nodes_indices: List[int] = [self.node_to_index[node] for node in observed_nodes]
        self._observed = list(observed_nodes)

        p_off, p_idx = [0], []
        for node in self._nodes:
            p_idx.extend(self.node_to_index[q] for q in bayesian_network.parents_of(node))
            p_off.append(len(p_idx))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_exact_create(
                    0 if self._dtype == torch.float32 else 1,
                    self.num_nodes,
                    _native.i32_array(self.dims),
                    _native.i32_array(p_off),
                    _native.i32_array(p_idx),
                    _native.i32_array(self.observed_nodes_indices),
                    self.num_observed_nodes,
                    ctypes.byref(handle),
                )
            )
        self._handle = handle
        flat = torch.cat([node.cpt.detach().to(self._device, self._dtype).reshape(-1) for node in self._nodes])
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_exact_set_cpts(self._handle, ctypes.c_void_p(flat.data_ptr()), self._stream()))
            # before any evidence the reference answers with prior marginals of shape [1, ...]
            # (its evidence factor starts as all ones, :31)
            _native.check(self._lib.bn_exact_enter_evidence(self._handle, 1, None, self._stream()))
        self._n = 1
        self._have_evidence = False

    def __del__(self):
        handle = getattr(self, "_handle", None)
        if handle:
            try:
                self._lib.bn_exact_destroy(handle)
            except Exception:
                pass
            self._handle = None

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    @property
    def settings(self) -> InferenceMachineSettings:
        return self._settings

    def enter_evidence(self, evidence: Evidence) -> None:
        if evidence.num_observed_nodes != self.num_observed_nodes:
            raise Exception(
                f"Length of evidence must match number of observed nodes: {len(self.observed_nodes_indices)}"
            )
        tensors = []
        for t, node in zip(evidence.data, self._observed):
            t = t.detach().to(self._device, self._dtype).contiguous()
            if t.dim() != 2 or t.shape[1] != node.num_states or t.shape[0] != evidence.num_observations:
                raise RuntimeError(
                    f"evidence for node {node} has shape {tuple(t.shape)}, "
                    f"expected {(evidence.num_observations, node.num_states)}"
                )
            tensors.append(t)
        n = int(evidence.num_observations)
        table = (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_exact_enter_evidence(self._handle, n, table, self._stream()))
        self._n = n
        self._have_evidence = True

    def _infer(self, nodes: List[Node]) -> torch.Tensor:
        # axes in ascending network order whatever the order of `nodes`, as the reference's
        # sum over the other axes leaves them (:90-97)
        indices = sorted(self.node_to_index[node] for node in nodes)
        out = torch.empty([self._n] + [self.dims[i] for i in indices], device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_exact_marginal(self._handle, _native.i32_array(indices), len(indices),
                                            ctypes.c_void_p(out.data_ptr()), self._stream())
            )
        return out

    def infer_nodes_with_parents(self, child_nodes: List[Node]) -> List[torch.Tensor]:
        return [self._infer(list(self.bayesian_network.parents_of(node)) + [node]) for node in child_nodes]

    def infer_single_nodes(self, nodes: List[Node]) -> List[torch.Tensor]:
        return [self._infer([node]) for node in nodes]

    def log_likelihoods(self) -> torch.Tensor:
        """log p(evidence) of every observation, ``[N]`` (the reference keeps it as
        ``_log_likelihoods``, :88)."""
        if not self._have_evidence:
            raise RuntimeError("No log-likelihood calculated. Did you enter evidence?")
        out = torch.empty((self._n,), device=self._device, dtype=self._dtype)
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_exact_log_likelihood(self._handle, 0, None, ctypes.c_void_p(out.data_ptr()),
                                                            self._stream()))
        return out

    def log_likelihood(self) -> float:
        if self.num_observed_nodes == 0:
            raise Exception("Log likelihood can't be calculated with 0 observed nodes")
        if not self._have_evidence:
            raise RuntimeError("No log-likelihood calculated. Did you enter evidence?")
        out = torch.empty((1,), device=self._device, dtype=torch.float64)
        with torch.cuda.device(self._device):
            _native.check(
                self._lib.bn_exact_log_likelihood(self._handle, 1 if self._settings.average_log_likelihood else 0,
                                                  ctypes.c_void_p(out.data_ptr()), None, self._stream())
            )
        return out.item()
