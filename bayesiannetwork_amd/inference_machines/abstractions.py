"""The inference-machine contract every caller programs against.

Five members, as in the reference ABC
(``packages/bayesian_network/src/bayesian_network/inference_machines/abstractions.py:13-33``):
``settings``, ``enter_evidence``, ``infer_single_nodes``,
``infer_nodes_with_parents`` and ``log_likelihood``.  Optimizers and evaluators
only ever see a ``Callable[[BayesianNetwork], IInferenceMachine]`` factory.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List

import torch

from ..bayesian_network import Node
from .common import InferenceMachineSettings
from .evidence import Evidence


class IInferenceMachine(ABC):
    @property
    @abstractmethod
    def settings(self) -> InferenceMachineSettings:
        raise NotImplementedError()

    @abstractmethod
    def enter_evidence(self, evidence: Evidence) -> None:
        raise NotImplementedError()

    @abstractmethod
    def infer_single_nodes(self, nodes: List[Node]) -> List[torch.Tensor]:
        raise NotImplementedError()

    @abstractmethod
    def infer_nodes_with_parents(self, child_nodes: List[Node]) -> List[torch.Tensor]:
        raise NotImplementedError()

    @abstractmethod
    def log_likelihood(self) -> float:
        raise NotImplementedError()
