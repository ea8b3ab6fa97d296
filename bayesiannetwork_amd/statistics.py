"""Probability-table helpers.

Same contracts as ``packages/common/src/common/statistics.py:6-25`` of the
reference: a random table is uniform noise normalised along the last axis, and
a table is valid when every last-axis slice sums to one within 1e-6 (float32)
or 1e-15 (float64).
"""

from __future__ import annotations

import torch

from .torch_settings import TorchSettings


def generate_random_probability_matrix(size, torch_settings: TorchSettings) -> torch.Tensor:
    raw = torch.rand(size, dtype=torch_settings.dtype, device=torch_settings.device)
    return raw / raw.sum(dim=-1, keepdim=True)


def is_valid_probability_matrix(p: torch.Tensor, tolerance: float | None = None):
    if tolerance is None:
        tolerance = 1e-6 if p.dtype == torch.float32 else 1e-15
    row_sums = p.sum(dim=-1)
    return torch.all((row_sums - 1).abs() < tolerance)


def rescale_tensor(t: torch.Tensor, gamma: float = 0.999999999) -> torch.Tensor:
    """Pull a one-hot/probability tensor away from exact zeros
    (``packages/common/src/common/tensor_helpers.py:6-11``)."""
    return t * gamma + (1 - gamma) / 2


def rescale_tensors(tensors, gamma: float = 0.999999999):
    return [rescale_tensor(t, gamma) for t in tensors]
