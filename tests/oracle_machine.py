"""Test-only adapter: the CPU oracle behind the IInferenceMachine interface, so that host
logic (optimizers, evaluators, sharded statistics) can be exercised without a GPU."""

from typing import List

import torch

from bayesiannetwork_amd.inference_machines.abstractions import IInferenceMachine
from bayesiannetwork_amd.inference_machines.hip_spa.hip_spa_inference_machine import SpaInferenceMachineSettings
from oracle.spa_oracle import SpaOracle


class OracleMachine(IInferenceMachine):
    def __init__(self, settings: SpaInferenceMachineSettings, network, observed_nodes, num_observations,
                 with_count_sums: bool = False):
        self._settings = settings
        self._network = network
        self.num_observations = num_observations
        self.observed_nodes = observed_nodes
        self._index = {n: i for i, n in enumerate(network.nodes)}
        self._oracle = SpaOracle(network, observed_nodes, num_observations, dtype=settings.torch_settings.dtype,
                                 num_iterations=settings.num_iterations, allow_loops=settings.allow_loops)
        if with_count_sums:
            # mimic the HIP machine's fast-path surface
            self.expected_count_sums = self._expected_count_sums
            self.split_counts = self._split_counts
            self.log_likelihood_tensor = lambda: torch.tensor([self.log_likelihood()], dtype=torch.float64)
        if with_count_sums == "step":
            # ... and its whole-step surface (statistics written into the caller's packed buffer)
            self.can_em_step = True
            self.em_step_statistics = self._em_step_statistics
            self.em_step = self._em_step
            self.m_step = self._m_step

    @property
    def settings(self):
        return self._settings

    def enter_evidence(self, evidence):
        self._oracle.enter_evidence(evidence.data)

    def infer_single_nodes(self, nodes) -> List[torch.Tensor]:
        return [self._oracle.infer_single(self._index[n]) for n in nodes]

    def infer_nodes_with_parents(self, child_nodes) -> List[torch.Tensor]:
        return [self._oracle.infer_family(self._index[n]) for n in child_nodes]

    def log_likelihood(self) -> float:
        return self._oracle.log_likelihood(self._settings.average_log_likelihood)

    def _expected_count_sums(self):
        n = max(self.num_observations, 1)
        return torch.cat([(c.double() * n).reshape(-1) for c in self._oracle.expected_counts()])

    def _split_counts(self, flat):
        out, pos = [], 0
        for node in self._network.nodes:
            size = node.cpt.numel()
            out.append(flat[pos:pos + size].reshape(node.cpt.shape))
            pos += size
        return out

    def _em_step_statistics(self, evidence, out=None):
        self.enter_evidence(evidence)
        counts = self._expected_count_sums()
        if out is None:
            out = torch.empty((counts.numel() + 1,), dtype=torch.float64)
        out[:counts.numel()] = counts
        out[counts.numel()] = self.log_likelihood()
        return out[counts.numel():counts.numel() + 1], out[:counts.numel()]

    def _m_step(self, count_sums, old_flat, learning_rate, regularization):
        from bayesiannetwork_amd.optimizers.em_step import FlatModel

        return FlatModel.of(self._network).m_step(count_sums, old_flat, learning_rate, regularization)

    def _em_step(self, evidence, learning_rate, regularization):
        from bayesiannetwork_amd.optimizers.em_step import FlatModel

        ll, counts = self._em_step_statistics(evidence)
        old = FlatModel.of(self._network).gather()
        return ll.clone(), self._m_step(counts, old, learning_rate, regularization)
