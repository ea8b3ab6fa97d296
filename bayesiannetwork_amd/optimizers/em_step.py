"""E-step statistic and M-step arithmetic shared by both EM optimizers.

E-step (reference ``em_optimizer.py:48-55``, ``em_batch_optimizer.py:61-68``):
for every node the observation-mean of its family marginal.  A machine that
offers ``expected_count_sums()`` (the HIP machine) hands back the *sums* in one
flat float64 buffer without materialising any ``[N, ...]`` tensor; any other
``IInferenceMachine`` is served through ``infer_nodes_with_parents``.

When ``torch.distributed`` is initialised and ``sharded=True`` the statistic is
all-reduced (SUM) over ranks together with the log-likelihood sum and the
observation count — the single collective of a data-parallel EM step — so every
rank then runs the identical M-step.

M-step (``em_optimizer.py:57-63``, ``em_batch_optimizer.py:70-84``): normalise
along the last axis; optionally blend with the uniform distribution
(regularisation) and with the previous CPT (learning rate).
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from ..bayesian_network import BayesianNetwork


def all_reduce_sum(buffer: torch.Tensor) -> None:
    """In-place SUM over the ranks of torch.distributed: the one collective of a data-parallel EM step.
    On RCCL (backend "nccl") the device buffer goes into the collective as it is.  gloo reduces in
    host memory: a device buffer is taken there and back explicitly, in this thread and on the current
    stream, rather than by gloo's own staging threads (which were seen to stall after a few dozen
    steps when two ranks share one GPU, the set-up in which gloo stands in for RCCL)."""
    dist = torch.distributed
    if buffer.is_cuda and dist.get_backend() == "gloo":
        host = buffer.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buffer.copy_(host)
    else:
        dist.all_reduce(buffer, op=dist.ReduceOp.SUM)


def expected_counts(machine, network: BayesianNetwork, sharded: bool = False,
                    ll: Optional[float] = None) -> Tuple[List[torch.Tensor], Optional[float]]:
    """Returns (per-node mean family marginals in network order, log-likelihood).

    ``ll`` is passed through, and when ``sharded`` it is combined across ranks with the
    machine's averaging convention (mean over all observations, or sum)."""
    nodes = list(network.nodes)
    dist = torch.distributed if sharded and torch.distributed.is_available() and torch.distributed.is_initialized() else None

    if hasattr(machine, "expected_count_sums"):
        flat = machine.expected_count_sums()
        n_local = float(max(machine.num_observations, 1))
        if dist is not None:
            avg = machine.settings.average_log_likelihood
            ll_sum = (ll * n_local) if (ll is not None and avg) else (ll if ll is not None else 0.0)
            tail = torch.tensor([ll_sum, n_local], dtype=torch.float64, device=flat.device)
            packed = torch.cat([flat, tail])
            all_reduce_sum(packed)
            flat, ll_total, n_total = packed[:-2], packed[-2].item(), packed[-1].item()
            if ll is not None:
                ll = ll_total / n_total if avg else ll_total
            n_local = n_total
        dtype = nodes[0].cpt.dtype
        means = [(v / n_local).to(dtype) for v in machine.split_counts(flat)]
        return means, ll

    families = machine.infer_nodes_with_parents(nodes)
    means = [p.mean(dim=0) for p in families]
    if dist is not None:
        n_local = float(families[0].shape[0])
        sizes = [m.numel() for m in means]
        flat = torch.cat([(m.double() * n_local).reshape(-1) for m in means])
        avg = machine.settings.average_log_likelihood
        ll_sum = (ll * n_local) if (ll is not None and avg) else (ll if ll is not None else 0.0)
        packed = torch.cat([flat, torch.tensor([ll_sum, n_local], dtype=torch.float64, device=flat.device)])
        all_reduce_sum(packed)
        n_total = packed[-1].item()
        if ll is not None:
            ll = packed[-2].item() / n_total if avg else packed[-2].item()
        pieces = torch.split(packed[:-2] / n_total, sizes)
        means = [p.reshape(m.shape).to(m.dtype) for p, m in zip(pieces, means)]
    return means, ll


class FlatModel:
    """All CPTs of a network as one flat device buffer plus the index vectors that let the
    M-step run as a handful of whole-network tensor operations instead of a few per node
    (785 nodes in the 784-leaf model).  Nodes keep CPT-shaped *views* of the buffer."""

    def __init__(self, network: BayesianNetwork):
        self.nodes = list(network.nodes)
        first = self.nodes[0].cpt
        self.device, self.dtype = first.device, first.dtype
        self.shapes = [tuple(n.cpt.shape) for n in self.nodes]
        self.sizes = [n.cpt.numel() for n in self.nodes]
        self._network = network
        self._store = None
        self._index = None  # (num_rows, row_ids, inv_states), built when the tensor M-step is first used

    @classmethod
    def of(cls, network: BayesianNetwork) -> "FlatModel":
        """The network's flat model, kept on the network between optimizer runs (building it walks
        every node; a second ``optimize`` call on the same network finds it ready)."""
        model = network.__dict__.get("_flat_model")
        if model is not None:
            store = model._store
            if (store is not None and store.complete and store.network is network
                    and store.flat.device == model.device and store.flat.dtype == model.dtype):
                return model  # every node still reads the shared buffer: nothing can have changed shape
            cpts = [n.cpt for n in network.nodes]
            if (model.shapes != [tuple(c.shape) for c in cpts] or cpts[0].device != model.device
                    or cpts[0].dtype != model.dtype):
                model = None
        if model is None:
            model = network.__dict__["_flat_model"] = cls(network)
        return model

    def _row_index(self):
        if self._index is None:
            rows, inv = [], []
            row_base = 0
            for shape, size in zip(self.shapes, self.sizes):
                s_last = shape[-1]
                n_rows = size // s_last
                rows.append(torch.arange(row_base, row_base + n_rows).repeat_interleave(s_last))
                inv.append(torch.full((size,), 1.0 / s_last, dtype=torch.float64))
                row_base += n_rows
            self._index = (row_base, torch.cat(rows).to(self.device), torch.cat(inv).to(self.device))
        return self._index

    @property
    def num_rows(self):
        return self._row_index()[0]

    @property
    def row_ids(self):
        return self._row_index()[1]

    @property
    def inv_states(self):
        return self._row_index()[2]

    def views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        out, pos = [], 0
        for shape, size in zip(self.shapes, self.sizes):
            out.append(flat[pos : pos + size].view(shape))
            pos += size
        return out

    def gather(self) -> torch.Tensor:
        """Current CPTs as one flat tensor in the model dtype (the shared buffer itself when
        every node still reads from it)."""
        store = self._store
        if store is not None and store.complete and store.flat.dtype == self.dtype:
            return store.flat
        return torch.cat([n.cpt.detach().to(self.device, self.dtype).reshape(-1) for n in self.nodes])

    def m_step(self, count_sums: torch.Tensor, old_flat: Optional[torch.Tensor],
               learning_rate: Optional[float], regularization: Optional[float]) -> torch.Tensor:
        """count_sums: flat float64 expected counts (any positive scale).  Returns the new flat
        CPT buffer in the model dtype."""
        row_sums = torch.zeros(self.num_rows, dtype=torch.float64, device=self.device)
        row_sums.index_add_(0, self.row_ids, count_sums)
        cpt = count_sums / row_sums[self.row_ids]
        if regularization:
            cpt = (1 - regularization) * cpt + regularization * self.inv_states
        if learning_rate is not None:
            cpt = (1 - learning_rate) * old_flat.double() + learning_rate * cpt
        return cpt.to(self.dtype)

    def assign(self, flat: torch.Tensor) -> None:
        """Make ``flat`` the network's parameters.  Nodes that can read from a shared buffer
        (this package's ``Node``) are switched in one assignment; others get a view each."""
        store = self._store
        if store is not None and store.complete:
            store.flat = flat
            return
        if all(hasattr(n, "attach") for n in self.nodes):
            from ..bayesian_network import CptStore

            store = CptStore(self._network, flat)
            pos = 0
            for node, shape, size in zip(self.nodes, self.shapes, self.sizes):
                node.attach(store, pos, pos + size, shape)
                pos += size
            self._store = store
            return
        for node, view in zip(self.nodes, self.views(flat)):
            node.cpt = view


def reduce_statistics(machine, flat_sums: torch.Tensor, ll, sharded: bool):
    """All-reduce (SUM) of [expected-count sums, log-likelihood sum, observation count] over the
    ranks of torch.distributed; identity when not sharded.  Returns (sums, ll, total N).
    Nothing here makes the host wait for the device when ``ll`` is a tensor: the packed buffer
    is assembled with device-side copies and fills only."""
    n_local = float(max(machine.num_observations, 1))
    dist = torch.distributed
    if not (sharded and dist.is_available() and dist.is_initialized()):
        return flat_sums, ll, n_local
    avg = machine.settings.average_log_likelihood
    total = flat_sums.numel()
    packed = torch.empty(total + 2, dtype=torch.float64, device=flat_sums.device)
    packed[:total].copy_(flat_sums)
    if isinstance(ll, torch.Tensor):
        packed[total : total + 1].copy_(ll.reshape(1))
        if avg:
            packed[total : total + 1].mul_(n_local)
    else:
        packed[total : total + 1].fill_((ll * n_local) if avg else ll)
    packed[total + 1 :].fill_(n_local)
    all_reduce_sum(packed)
    n_total = packed[total + 1]
    ll_out = packed[total] / n_total if avg else packed[total]
    if not isinstance(ll, torch.Tensor):
        ll_out = ll_out.item()
        n_total = n_total.item()
    return packed[:total], ll_out, n_total


def reduce_packed(machine, packed: torch.Tensor, sharded: bool):
    """``reduce_statistics`` for statistics that already lie in the buffer that is all-reduced:
    ``packed`` = [expected-count sums..., log-likelihood, (free slot)] as
    ``machine.em_step_statistics(evidence, out=packed)`` leaves it.  Returns (sums, ll tensor, total N
    tensor); device-side operations only."""
    total = packed.numel() - 2
    n_local = float(max(machine.num_observations, 1))
    dist = torch.distributed
    if not (sharded and dist.is_available() and dist.is_initialized()):
        return packed[:total], packed[total : total + 1], n_local
    avg = machine.settings.average_log_likelihood
    if avg:
        packed[total : total + 1].mul_(n_local)
    packed[total + 1 :].fill_(n_local)
    all_reduce_sum(packed)
    ll = packed[total : total + 1] / packed[total + 1] if avg else packed[total : total + 1]
    return packed[:total], ll, packed[total + 1]


def normalise(p_conditional: torch.Tensor) -> torch.Tensor:
    return p_conditional / p_conditional.sum(dim=-1, keepdim=True)


def regularise(cpt: torch.Tensor, gamma: Optional[float]) -> torch.Tensor:
    if not gamma:
        return cpt
    return (1 - gamma) * cpt + gamma / cpt.shape[-1]


def blend(old: torch.Tensor, new: torch.Tensor, learning_rate: float) -> torch.Tensor:
    return (1 - learning_rate) * old + learning_rate * new
