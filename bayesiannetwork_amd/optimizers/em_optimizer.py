"""Full-batch EM (reference: ``optimizers/em_optimizer.py:13-63``): every iteration builds a
machine from the factory, enters the whole evidence, reads the log-likelihood, and replaces
every CPT by the normalised expected counts."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

from ..bayesian_network import BayesianNetwork
from ..inference_machines.evidence import Evidence
from .abstractions import IOptimizer
from .common import MachineFactory, OptimizerLogger
from .em_step import FlatModel, expected_counts, normalise, reduce_statistics


@dataclass(frozen=True)
class EmOptimizerSettings:
    num_iterations: int
    sharded: bool = False  # all-reduce the statistics over torch.distributed ranks


class EmOptimizer(IOptimizer):
    def __init__(self, bayesian_network: BayesianNetwork, inference_machine_factory: MachineFactory,
                 settings: EmOptimizerSettings, logger: Optional[OptimizerLogger] = None):
        self._network = bayesian_network
        self._factory = inference_machine_factory
        self._settings = settings
        self._logger = logger

    def optimize(self, evidence: Evidence) -> None:
        flat_model = None
        for iteration in range(self._settings.num_iterations):
            machine = self._factory(self._network)
            count_sums = None
            if hasattr(machine, "em_statistics"):
                ll, count_sums = machine.em_statistics(evidence)
                ll = ll.item()
            else:
                machine.enter_evidence(evidence)
                ll = machine.log_likelihood()
            if hasattr(machine, "expected_count_sums"):
                if flat_model is None:
                    flat_model = FlatModel.of(self._network)
                if count_sums is None:
                    count_sums = machine.expected_count_sums()
                sums, ll, _ = reduce_statistics(machine, count_sums, ll, self._settings.sharded)
                step = machine.m_step if hasattr(machine, "m_step") else flat_model.m_step
                flat_model.assign(step(sums, None, None, None))
            else:
                counts, ll = expected_counts(machine, self._network, self._settings.sharded, ll)
                for node, p in zip(self._network.nodes, counts):
                    node.cpt = normalise(p)
            if self._logger:
                self._logger.log(0, iteration, ll)
