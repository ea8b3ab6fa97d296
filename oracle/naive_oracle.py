"""CPU restatement of the reference's brute-force machine — TEST INFRASTRUCTURE ONLY.

Follows ``inference_machines/naive/naive_inference_machine.py`` of the reference:

* joint table ``p = prod_nodes cpt_node`` broadcast over one axis per node (``:37-57``);
* evidence factor ``prod_i e_i[n, x_i]``, ``c[n] = sum p * evidence``, log-likelihoods ``log c``
  (``:59-88``);
* marginal over a node set = sum of ``p * evidence / c`` over all other axes, remaining axes in
  ascending network order (``:90-110``).

One difference, on purpose: each CPT axis is matched to its parent by identity (an einsum
subscript per node) instead of by ``reshape`` into ascending-order axes, so the joint is right for
any node order; for networks whose parents precede their children in CPT-axis order the two are
the same array.  Pinned by ``tests/golden/naive_*.npz`` (outputs of the reference's own machine).

Only ``tests/`` may import this module.
"""

from __future__ import annotations

from typing import List, Sequence

import torch


class NaiveOracle:
    def __init__(self, cpts: Sequence[torch.Tensor], parents: Sequence[Sequence[int]], observed: Sequence[int],
                 dtype=torch.float64):
        """``cpts[j]`` is node j's CPT ``[parent axes.., states]``, ``parents[j]`` its parents' node
        indices in CPT axis order, ``observed`` the observed node indices in evidence order."""
        self.dtype = dtype
        self.num_nodes = len(cpts)
        self.dims = [int(c.shape[-1]) for c in cpts]
        self.observed = list(observed)
        letters = "abcdefghijklmnopqrstuvwxyz"
        assert self.num_nodes <= len(letters)
        p = torch.ones(self.dims, dtype=dtype)
        everything = letters[: self.num_nodes]
        for j, cpt in enumerate(cpts):
            sub = "".join(letters[q] for q in parents[j]) + letters[j]
            p = torch.einsum(f"{everything},{sub}->{everything}", p, cpt.to(dtype))
        self.p = p[None]
        self.p_evidence = torch.ones_like(self.p)
        self.log_likelihoods = None

    def enter_evidence(self, evidence: List[torch.Tensor]):
        assert len(evidence) == len(self.observed)
        n = evidence[0].shape[0]
        e = torch.ones([n] + self.dims, dtype=self.dtype)
        for rows, j in zip(evidence, self.observed):
            shape = [n] + [1] * self.num_nodes
            shape[1 + j] = self.dims[j]
            e = e * rows.to(self.dtype).reshape(shape)
        c = (self.p * e).sum(dim=tuple(range(1, self.num_nodes + 1)), keepdim=True)
        self.p_evidence = e / c
        self.log_likelihoods = torch.log(c).reshape(n)

    def infer(self, nodes: Sequence[int]) -> torch.Tensor:
        keep = set(nodes)
        away = [1 + j for j in range(self.num_nodes) if j not in keep]
        full = self.p * self.p_evidence
        return full.sum(dim=away) if away else full

    def log_likelihood(self, average: bool) -> float:
        return (self.log_likelihoods.mean() if average else self.log_likelihoods.sum()).item()
