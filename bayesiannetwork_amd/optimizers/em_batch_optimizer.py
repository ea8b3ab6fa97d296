"""Mini-batch EM (reference: ``optimizers/em_batch_optimizer.py:15-84``): per batch a fresh
machine, log-likelihood, expected counts, then
``cpt <- (1-lr) * cpt + lr * ((1-gamma) * normalised_counts + gamma * uniform)``."""

from __future__ import annotations

from dataclasses import dataclass
import torch
from typing import Optional

from ..bayesian_network import BayesianNetwork
from ..inference_machines.evidence import EvidenceLoader
from .abstractions import IBatchOptimizer, IEvaluator
from .common import MachineFactory, OptimizerLogger
from .em_step import FlatModel, blend, expected_counts, normalise, reduce_packed, reduce_statistics, regularise


@dataclass(frozen=True)
class EmBatchOptimizerSettings:
    learning_rate: float
    num_epochs: int
    regularization: Optional[float] = None
    sharded: bool = False  # all-reduce the statistics over torch.distributed ranks
    # keep the per-step log-likelihood on the device and hand it to the logger at the end of
    # each epoch, so that no EM step waits for the host (needs log_likelihood_tensor())
    deferred_logging: bool = False


class EmBatchOptimizer(IBatchOptimizer):
    def __init__(self, bayesian_network: BayesianNetwork, inference_machine_factory: MachineFactory,
                 settings: EmBatchOptimizerSettings, logger: Optional[OptimizerLogger] = None,
                 evaluator: Optional[IEvaluator] = None):
        self._network = bayesian_network
        self._factory = inference_machine_factory
        self._settings = settings
        self._logger = logger
        self._evaluator = evaluator

    def optimize(self, evidence_loader: EvidenceLoader) -> None:
        cfg = self._settings
        flat_model = None
        stepped = False  # some step ran as one engine call (em_step)
        for epoch in range(cfg.num_epochs):
            pending = []
            for iteration, evidence in enumerate(evidence_loader):
                machine = self._factory(self._network)
                deferred = cfg.deferred_logging and hasattr(machine, "log_likelihood_tensor")
                count_sums = None
                if (not cfg.sharded and getattr(machine, "can_em_step", False)):
                    # the whole step in one engine call; the parameters it returns become the network's
                    if flat_model is None:
                        flat_model = FlatModel.of(self._network)
                        flat_model.assign(flat_model.gather())  # nodes read one shared buffer from here on
                        machine = None  # back to the pool first: the rebuilt machine gets the same plan
                        machine = self._factory(self._network)
                    ll, flat = machine.em_step(evidence, cfg.learning_rate, cfg.regularization)
                    flat_model.assign(flat)
                    stepped = True
                    # back to the pool before the next machine is built: that one then gets this very
                    # plan, which already holds the new parameters
                    machine = None
                    if not deferred:
                        ll = ll.item()
                    if deferred:
                        pending.append((iteration, ll))
                    elif self._logger:
                        self._logger.log(epoch, iteration, ll)
                    if self._evaluator:
                        self._evaluator.evaluate(epoch, iteration, self._network)
                    continue
                if cfg.sharded and getattr(machine, "can_em_step", False) and hasattr(machine, "em_step_statistics"):
                    # sharded run: this rank's statistics in one engine call, written straight into the
                    # buffer that is all-reduced; then the same M-step on every rank
                    if flat_model is None:
                        flat_model = FlatModel.of(self._network)
                    flat = flat_model.gather()
                    packed = torch.empty((flat.numel() + 2,), dtype=torch.float64, device=flat.device)
                    machine.em_step_statistics(evidence, out=packed)
                    sums, ll, _ = reduce_packed(machine, packed, cfg.sharded)
                    flat = machine.m_step(sums, flat, cfg.learning_rate, cfg.regularization)
                    flat_model.assign(flat)
                    machine = None
                    if not deferred:
                        ll = ll.item()
                    if deferred:
                        pending.append((iteration, ll))
                    elif self._logger:
                        self._logger.log(epoch, iteration, ll)
                    if self._evaluator:
                        self._evaluator.evaluate(epoch, iteration, self._network)
                    continue
                if hasattr(machine, "em_statistics"):
                    # evidence, iterations, log-likelihood and expected counts in one engine call
                    ll, count_sums = machine.em_statistics(evidence)
                    if not deferred:
                        ll = ll.item()
                else:
                    machine.enter_evidence(evidence)
                    ll = machine.log_likelihood_tensor() if deferred else machine.log_likelihood()
                if hasattr(machine, "expected_count_sums"):
                    # whole-network M-step on one flat buffer; nodes get views of it
                    if flat_model is None:
                        flat_model = FlatModel.of(self._network)
                    flat = flat_model.gather()
                    if count_sums is None:
                        count_sums = machine.expected_count_sums()
                    sums, ll, _ = reduce_statistics(machine, count_sums, ll, cfg.sharded)
                    step = machine.m_step if hasattr(machine, "m_step") else flat_model.m_step
                    flat = step(sums, flat, cfg.learning_rate, cfg.regularization)
                    flat_model.assign(flat)
                else:
                    counts, ll = expected_counts(machine, self._network, cfg.sharded, ll)
                    for node, p in zip(self._network.nodes, counts):
                        target = regularise(normalise(p), cfg.regularization)
                        node.cpt = blend(node.cpt, target, cfg.learning_rate)
                if deferred:
                    pending.append((iteration, ll))
                elif self._logger:
                    self._logger.log(epoch, iteration, ll)
                if self._evaluator:
                    self._evaluator.evaluate(epoch, iteration, self._network)
            if self._logger:
                for iteration, ll in pending:
                    self._logger.log(epoch, iteration, float(ll))
        if stepped:
            # a one-launch step that could not run is a skipped step: say so before returning (the
            # pooled plan of the last step comes back with this machine)
            machine = self._factory(self._network)
            if hasattr(machine, "report_void_steps"):
                machine.report_void_steps(synchronize=True)
