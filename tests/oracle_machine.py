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
