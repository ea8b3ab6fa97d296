"""Settings shared by all inference machines
(reference: ``inference_machines/common.py:6-9``)."""

from __future__ import annotations

from dataclasses import dataclass

from ..torch_settings import TorchSettings


@dataclass
class InferenceMachineSettings:
    torch_settings: TorchSettings
    average_log_likelihood: bool
