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


def _common_base(tensors):
    """The flat host tensor that ``tensors`` tile back to back, or None."""
    first = tensors[0]
    if first.is_cuda or not first.is_contiguous():
        return None
    storage_ptr = first.untyped_storage().data_ptr()
    start, pos = first.storage_offset(), first.storage_offset()
    for t in tensors:
        if (t.is_cuda or t.dtype != first.dtype or not t.is_contiguous() or t.storage_offset() != pos
                or t.untyped_storage().data_ptr() != storage_ptr):
            return None
        pos += t.numel()
    if len(tensors) < 2:
        return None
    return torch.empty(0, dtype=first.dtype).set_(first.untyped_storage(), start, (pos - start,))


def _to_device(tensors, device, dtype, non_blocking=False, pin=False):
    """``[t.to(device, dtype) for t in tensors]``; host tensors that tile one buffer (the product of
    ``torch.stack(...).unbind(0)`` in an evidence transform) cross to the device as one copy."""
    device = torch.device(device)
    base = _common_base(tensors) if device.type == "cuda" else None
    if base is not None:
        if pin and not base.is_pinned():
            base = base.pin_memory()
        whole = base.to(device, dtype, non_blocking=non_blocking)
        out, pos = [], 0
        for t in tensors:
            out.append(whole[pos : pos + t.numel()].view(t.shape))
            pos += t.numel()
        return out
    if pin:
        tensors = [t if t.is_pinned() or t.is_cuda else t.pin_memory() for t in tensors]
    return [t.to(device, dtype, non_blocking=non_blocking) for t in tensors]


class Evidence:
    def __init__(self, data: Sequence[torch.Tensor], torch_settings: TorchSettings):
        self.torch_settings = torch_settings
        self.num_observed_nodes = len(data)
        self.num_observations = int(data[0].shape[0])
        for i, d in enumerate(data):
            assert d.shape[0] == self.num_observations, (
                "Number of observations not equal for each observed node"
            )
        self._data: List[torch.Tensor] = _to_device(list(data), torch_settings.device, torch_settings.dtype)

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


class HardEvidence(Evidence):
    """Hard (one-hot) evidence kept as one ``uint8`` state index per observed node and observation
    instead of ``num_states`` floats.  It stands for the tensors
    ``one_hot(state) * (on_value - off_value) + off_value`` (the soft-clamped one-hot rows the
    reference's experiments feed in) and hands out exactly those through ``data`` to any
    ``IInferenceMachine``; a machine that knows the type can consume the indices directly.

    ``states``: integer tensor ``[num_observed_nodes, num_observations]``;
    ``num_states``: number of states of each observed node, in the same order."""

    def __init__(self, states: torch.Tensor, num_states: Sequence[int], torch_settings: TorchSettings,
                 on_value: float = 1.0, off_value: float = 0.0):
        if states.dim() != 2 or states.shape[0] != len(num_states):
            raise ValueError("states must be [num_observed_nodes, num_observations] with one num_states entry per row")
        if max(num_states) > 256:
            raise ValueError("state indices are stored in 8 bits")
        if states.numel() and not states.dtype.is_floating_point:
            # before the 8-bit cast: a wrapped value would silently become another valid state
            limits = torch.tensor([int(s) for s in num_states], device=states.device).unsqueeze(1)
            if bool(((states < 0) | (states >= limits)).any()):
                raise ValueError("every state index must satisfy 0 <= state < num_states of its row")
        elif states.numel():
            raise ValueError("states must be an integer tensor")
        self.torch_settings = torch_settings
        self.num_observed_nodes = int(states.shape[0])
        self.num_observations = int(states.shape[1])
        self.num_states = [int(s) for s in num_states]
        self.on_value, self.off_value = float(on_value), float(off_value)
        self.states = states.to(torch_settings.device, torch.uint8).contiguous()
        self._expanded = None

    @property
    def data(self) -> List[torch.Tensor]:
        if self._expanded is None:
            dtype = self.torch_settings.dtype
            on = torch.tensor(self.on_value, dtype=dtype, device=self.states.device)
            off = torch.tensor(self.off_value, dtype=dtype, device=self.states.device)
            self._expanded = [
                torch.where(row.long().unsqueeze(1) == torch.arange(s, device=row.device).unsqueeze(0), on, off)
                for row, s in zip(self.states, self.num_states)
            ]
        return self._expanded

    def __getitem__(self, index) -> "HardEvidence":
        if isinstance(index, int):
            index = slice(index, index + 1)
        return HardEvidence(self.states[:, index], self.num_states, self.torch_settings, self.on_value, self.off_value)


class PrefetchingLoader:
    _common_base = staticmethod(_common_base)
    """Wraps an iterable of evidence batches that live in host memory and moves batch k+1 to the
    device (pinned staging buffer, side stream) while batch k is being used, so that a training
    loop never waits for a host-to-device copy it could have overlapped.  ``to_device`` maps a host
    batch to its device form with ``non_blocking`` copies; the default handles ``Evidence`` and
    ``HardEvidence``."""

    def __init__(self, batches, device: torch.device, to_device: Callable = None):
        self._batches = batches
        self._device = torch.device(device)
        self._to_device = to_device or self._move
        self.num_observations = getattr(batches, "num_observations", None)

    def __len__(self) -> int:
        return len(self._batches)

    def _move(self, batch):
        ts = TorchSettings(device=self._device.type, dtype=str(batch.torch_settings.dtype).replace("torch.", ""))
        if isinstance(batch, HardEvidence):
            states = batch.states if batch.states.is_pinned() or batch.states.is_cuda else batch.states.pin_memory()
            moved = HardEvidence.__new__(HardEvidence)
            moved.torch_settings = ts
            moved.num_observed_nodes, moved.num_observations = batch.num_observed_nodes, batch.num_observations
            moved.num_states, moved.on_value, moved.off_value = batch.num_states, batch.on_value, batch.off_value
            moved.states = states.to(self._device, non_blocking=True)
            moved._expanded = None
            return moved
        moved = Evidence.__new__(Evidence)
        moved.torch_settings = ts
        moved.num_observed_nodes, moved.num_observations = batch.num_observed_nodes, batch.num_observations
        moved._data = _to_device(batch.data, self._device, ts.dtype, non_blocking=True, pin=True)
        return moved

    @staticmethod
    def _device_tensors(moved):
        """The device tensors a moved batch owns, one per storage."""
        if isinstance(moved, HardEvidence):
            tensors = [moved.states]
        elif isinstance(moved, Evidence):
            tensors = list(moved._data)
        elif isinstance(moved, torch.Tensor):
            tensors = [moved]
        elif isinstance(moved, (list, tuple)):
            tensors = [t for t in moved if isinstance(t, torch.Tensor)]
        else:
            tensors = []
        seen, out = set(), []
        for t in tensors:
            key = t.untyped_storage().data_ptr()
            if t.is_cuda and key not in seen:
                seen.add(key)
                out.append(t)
        return out

    def _hand_over(self, pending):
        """Make the consumer's stream wait for the copy, and tell the caching allocator that the
        batch is used on that stream: its memory was allocated on the copy stream, and without this
        a later copy could reuse it while the consumer's kernels are still queued."""
        moved, ready = pending
        consumer = torch.cuda.current_stream(self._device)
        consumer.wait_event(ready)
        for t in self._device_tensors(moved):
            t.record_stream(consumer)
        return moved

    def __iter__(self):
        if self._device.type != "cuda":
            yield from self._batches
            return
        copy_stream = torch.cuda.Stream(self._device)
        pending = None
        for batch in self._batches:
            with torch.cuda.stream(copy_stream):
                moved = self._to_device(batch)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            if pending is not None:
                yield self._hand_over(pending)
            pending = (moved, ready)
        if pending is not None:
            yield self._hand_over(pending)


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
