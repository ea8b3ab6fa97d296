"""Evidence containers.

``Evidence`` is a list with one ``[num_observations, num_states_i]`` tensor per
observed node (soft or one-hot rows), cast to the configured device and dtype
on construction; indexing slices the observation axis of every tensor.
``EvidenceLoader`` adapts a ``(batch, label)`` iterable through a user
transform.  Behaviour follows the reference
(``inference_machines/evidence.py:8-45`` and ``:48-68``).
"""

from __future__ import annotations

from typing import Callable, Iterator, List, Sequence

import torch

from ..torch_settings import TorchSettings


class Evidence:
    def __init__(self, data: Sequence[torch.Tensor], torch_settings: TorchSettings):
        self.torch_settings = torch_settings
        self.num_observed_nodes = len(data)
        self.num_observations = int(data[0].shape[0])
        for i, d in enumerate(data):
            assert d.shape[0] == self.num_observations, (
                "Number of observations not equal for each observed node"
            )
        self._data: List[torch.Tensor] = [
            d.to(torch_settings.device, torch_settings.dtype) for d in data
        ]

    @property
    def data(self) -> List[torch.Tensor]:
        return self._data

    def __len__(self) -> int:
        return self.num_observations

    def __getitem__(self, index) -> "Evidence":
        return Evidence([d[index] for d in self._data], self.torch_settings)

    def get_observation(self, index: int) -> torch.Tensor:
        return torch.stack(self[index].data)

    def __repr__(self) -> str:
        return (
            f"\nnum_observed_nodes: {self.num_observed_nodes},\n"
            f"num_observations: {self.num_observations}\n"
        )


class EvidenceLoader:
    def __init__(self, data_loader, transform: Callable[[torch.Tensor], Evidence]):
        self._data_loader = data_loader
        self._transform = transform

    def __len__(self) -> int:
        return len(self._data_loader)

    def __iter__(self) -> Iterator[Evidence]:
        for batch, _labels in self._data_loader:
            yield self._transform(batch)

    @property
    def num_observations(self) -> int:
        return len(self._data_loader.dataset)
