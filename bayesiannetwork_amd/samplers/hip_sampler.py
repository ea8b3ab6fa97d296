"""Ancestral sampling in a gfx950 HIP kernel.

Same contract as the reference's ``TorchBayesianNetworkSampler``
(``samplers/torch_sampler.py:10-60``): ``sample(num_samples, nodes)`` returns an ``int32`` tensor
``[num_samples, len(nodes)]`` of jointly sampled states, parents sampled before their children
whether or not they were asked for.  The reference draws one trial per Python loop iteration; here
one GPU thread draws one trial (``bn_sampler_sample`` of ``include/bn_hip.h``), with a
counter-based generator: the same ``seed`` gives the same samples whatever the batch is cut into.
There is no CPU path (``BatchedBayesianNetworkSampler`` covers that).
"""

from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from ..bayesian_network import BayesianNetwork, Node
from ..inference_machines.hip_spa import _native
from ..torch_settings import TorchSettings
from .abstractions import IBayesianNetworkSampler


class HipBayesianNetworkSampler(IBayesianNetworkSampler):
    def __init__(self, bayesian_network: BayesianNetwork, torch_settings: TorchSettings, seed: Optional[int] = None):
        if torch_settings.device.type != "cuda":
            raise ValueError("HipBayesianNetworkSampler runs on an AMD GPU only (TorchSettings(device='cuda'))")
        self.torch_settings = torch_settings
        self.bayesian_network = bayesian_network
        self._lib = _native.load()
        self._device = (torch.device("cuda", torch.cuda.current_device()) if torch_settings.device.index is None
                        else torch_settings.device)
        self._nodes = list(bayesian_network.nodes)
        self._index = {node: i for i, node in enumerate(self._nodes)}
        states, p_off, p_idx = [], [0], []
        for node in self._nodes:
            states.append(int(node.num_states))
            p_idx.extend(self._index[q] for q in bayesian_network.parents_of(node))
            p_off.append(len(p_idx))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self._device):
            _native.check(self._lib.bn_sampler_create(
                0 if torch_settings.dtype == torch.float32 else 1, len(self._nodes), _native.i32_array(states),
                _native.i32_array(p_off), _native.i32_array(p_idx), ctypes.byref(handle)))
        self._handle = handle
        # successive calls continue one stream of seeds, so that two calls never repeat each other
        self._seed = int(seed) if seed is not None else int(torch.randint(0, 2**62, (1,)).item())
        self._calls = 0

    def __del__(self):
        handle = getattr(self, "_handle", None)
        if handle:
            try:
                self._lib.bn_sampler_destroy(handle)
            except Exception:
                pass
            self._handle = None

    def sample(self, num_samples: int, nodes: List[Node]) -> torch.Tensor:
        out = torch.empty((num_samples, len(nodes)), device=self._device, dtype=torch.int32)
        if not nodes or num_samples == 0:
            return out
        wanted = _native.i32_array([self._index[node] for node in nodes])
        flat = torch.cat([node.cpt.detach().to(self._device, self.torch_settings.dtype).reshape(-1)
                          for node in self._nodes])
        if flat.numel() != self._lib.bn_sampler_cpt_elements(self._handle):
            raise ValueError("CPT sizes do not match the network the sampler was built for")
        with torch.cuda.device(self._device):
            need = self._lib.bn_sampler_workspace_bytes(self._handle, wanted, len(nodes), num_samples)
            if need < 0:
                raise _native.BnError("bad sampling request")
            workspace = torch.empty((need,), device=self._device, dtype=torch.uint8)
            seed = (self._seed + 0x9E3779B97F4A7C15 * self._calls) % (1 << 64)
            self._calls += 1
            _native.check(self._lib.bn_sampler_sample(
                self._handle, ctypes.c_void_p(flat.data_ptr()), wanted, len(nodes), num_samples, seed,
                ctypes.c_void_p(workspace.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)))
        return out
