"""Device / dtype selection shared by every component.

Mirrors the keyword-only ``TorchSettings(device=..., dtype=...)`` object of the
reference (``packages/common/src/common/torch_settings.py:6-36``): the same two
fields, the same accepted spellings (``"cpu" | "cuda" | "mps"`` and
``"float32" | "float64"``), ``ValueError`` on anything else, and the same
environment fallback (``BN__TORCH_SETTINGS__DEVICE`` / ``BN__TORCH_SETTINGS__DTYPE``).
It is a plain class: the reference's pydantic-settings dependency is not needed
for two validated fields.  PyTorch-ROCm exposes an MI355X as ``"cuda"``.
"""

from __future__ import annotations

import os

import torch

_ENV_PREFIX = "BN__TORCH_SETTINGS__"
_DEVICES = ("cpu", "cuda", "mps")
_DTYPES = {"float32": torch.float32, "float64": torch.float64}
_MISSING = object()


class TorchSettings:
    __slots__ = ("device", "dtype")

    def __init__(self, *args, device=_MISSING, dtype=_MISSING):
        if args:
            raise TypeError(f"Positional arguments are not allowed. Got: {args}")

        if device is _MISSING:
            device = os.environ.get(_ENV_PREFIX + "DEVICE", _MISSING)
        if dtype is _MISSING:
            dtype = os.environ.get(_ENV_PREFIX + "DTYPE", _MISSING)
        if device is _MISSING or dtype is _MISSING:
            raise ValueError("TorchSettings needs both 'device' and 'dtype'")

        if not (isinstance(device, str) and device in _DEVICES):
            raise ValueError(f"Invalid device '{device}'. Must be one of {set(_DEVICES)}.")
        if not (isinstance(dtype, str) and dtype in _DTYPES):
            raise ValueError(f"Invalid dtype '{dtype}'. Must be one of {list(_DTYPES)}.")

        object.__setattr__(self, "device", torch.device(device))
        object.__setattr__(self, "dtype", _DTYPES[dtype])

    def __setattr__(self, key, value):
        raise AttributeError("TorchSettings is immutable")

    def __repr__(self):
        return f"TorchSettings(device={self.device!r}, dtype={self.dtype!r})"

    def __eq__(self, other):
        return (
            isinstance(other, TorchSettings)
            and self.device == other.device
            and self.dtype == other.dtype
        )

    def __hash__(self):
        return hash((str(self.device), self.dtype))
