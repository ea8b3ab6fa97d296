"""Bookkeeping around the EM loops: a log of per-step log-likelihoods and two
evaluators that score the current network on held-out evidence.

Semantics follow the reference (``optimizers/common.py:14-121``), including the
batch evaluator's rule that an *averaged* batch log-likelihood is weighted by
``len(batch) / num_observations`` before the batches are added up (``:109-110``).
"""

from __future__ import annotations

import logging
from dataclasses import dataclass
from datetime import datetime
from typing import Callable, Dict, List, Optional, Tuple

from ..bayesian_network import BayesianNetwork
from ..inference_machines.abstractions import IInferenceMachine
from ..inference_machines.evidence import Evidence, EvidenceLoader
from .abstractions import IEvaluator

MachineFactory = Callable[[BayesianNetwork], IInferenceMachine]


@dataclass(frozen=True)
class Log:
    ts: datetime
    epoch: int
    iteration: int
    ll: float


class OptimizerLogger:
    def __init__(self, should_log: Optional[Callable[[int, int], bool]] = None):
        self._should_log = should_log
        self._logs: List[Log] = []

    def log(self, epoch: int, iteration: int, ll: float):
        if self._should_log and not self._should_log(epoch, iteration):
            return
        entry = Log(datetime.now(), epoch, iteration, ll)
        self._logs.append(entry)
        logging.info("%s", entry)

    @property
    def logs(self) -> List[Log]:
        return self._logs


class Evaluator(IEvaluator):
    def __init__(self, inference_machine_factory: MachineFactory, evidence: Evidence,
                 should_evaluate: Callable[[int, int], bool]):
        self._factory = inference_machine_factory
        self._evidence = evidence
        self._should_evaluate = should_evaluate
        self._log_likelihoods: Dict[Tuple[int, int], float] = {}

    def evaluate(self, epoch: int, iteration: int, network: BayesianNetwork):
        if not self._should_evaluate(epoch, iteration):
            return
        machine = self._factory(network)
        machine.enter_evidence(self._evidence)
        ll = machine.log_likelihood()
        self._log_likelihoods[(epoch, iteration)] = ll
        logging.info("Evaluated for epoch %s, iteration: %s - ll: %s", epoch, iteration, ll)

    @property
    def log_likelihoods(self):
        return self._log_likelihoods


class BatchEvaluator(IEvaluator):
    def __init__(self, inference_machine_factory: MachineFactory, evidence_loader: EvidenceLoader,
                 should_evaluate: Optional[Callable[[int, int], bool]] = None):
        self._factory = inference_machine_factory
        self._loader = evidence_loader
        self._should_evaluate = should_evaluate
        self._log_likelihoods: Dict[Tuple[int, int], float] = {}

    def evaluate(self, epoch: int, iteration: int, network: BayesianNetwork):
        if self._should_evaluate and not self._should_evaluate(epoch, iteration):
            return
        machine = self._factory(network)
        total = 0.0
        for evidence in self._loader:
            machine.enter_evidence(evidence)
            ll = machine.log_likelihood()
            if machine.settings.average_log_likelihood:
                ll *= len(evidence) / self._loader.num_observations
            total += ll
        self._log_likelihoods[(epoch, iteration)] = total
        logging.info("Evaluated for epoch %s, iteration: %s, ll: %s", epoch, iteration, total)

    @property
    def log_likelihoods(self):
        return self._log_likelihoods
