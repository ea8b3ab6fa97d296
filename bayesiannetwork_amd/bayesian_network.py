"""Discrete Bayesian network definition: nodes holding CPT tensors, and a DAG.

Public surface and ordering rules follow the reference
(``packages/bayesian_network/src/bayesian_network/bayesian_network.py:13-181``):

* ``Node(cpt, name)`` — ``cpt`` has shape ``[parent_0, ..., parent_{P-1}, states]``
  and must sum to one along its last axis (``:17-23``).
* ``BayesianNetwork(nodes, parents)`` — ``nodes`` iterates in *graph insertion
  order*: walking the constructor's list, a node is inserted when first seen
  and each of its parents is inserted right after it if still unseen (the
  reference gets this from ``DiGraph.add_node`` / ``add_edge``, ``:36-40``).
  ``parents_of`` keeps the order of ``parents[node]`` (= CPT axis order);
  ``children_of`` keeps the order in which the children were walked
  (= variable-node input slot order in the sum-product machine).
* ``BayesianNetworkBuilder`` with the same five validation rules (``:88-138``).

The structure is kept in plain insertion-ordered dicts so that index tables
for the GPU plan can be produced without graph-library calls; ``G`` builds a
``networkx.DiGraph`` view on demand for callers that want one.
"""

from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .statistics import generate_random_probability_matrix, is_valid_probability_matrix
from .torch_settings import TorchSettings


class CptStore:
    """The CPTs of all nodes of one network back to back in one flat tensor (network order).
    Nodes attached to a store read their ``cpt`` as a view of ``flat``; replacing ``flat``
    therefore re-parameterises the whole network in one assignment.  A node whose ``cpt`` is
    assigned directly leaves the store (``attached`` drops below ``count``)."""

    __slots__ = ("network", "flat", "count", "attached")

    def __init__(self, network: "BayesianNetwork", flat: torch.Tensor):
        self.network = network
        self.flat = flat
        self.count = 0
        self.attached = 0

    @property
    def complete(self) -> bool:
        return self.attached == self.count


class Node:
    def __init__(self, cpt: torch.Tensor, name: Optional[str] = None):
        if not is_valid_probability_matrix(cpt):
            raise Exception("The CPT should sum to 1 along the last dimension.")

        self._store: Optional[CptStore] = None
        self._cpt = cpt
        self.num_states: int = int(cpt.shape[-1])
        self.name = name

    @property
    def cpt(self) -> torch.Tensor:
        store = self._store
        if store is None:
            return self._cpt
        flat = store.flat
        if self._view_of is not flat:
            self._view = flat[self._lo : self._hi].view(self._shape)
            self._view_of = flat
        return self._view

    @cpt.setter
    def cpt(self, value: torch.Tensor) -> None:
        if self._store is not None:
            self._store.attached -= 1
            self._store = None
            self._view = self._view_of = None
        self._cpt = value

    def attach(self, store: CptStore, lo: int, hi: int, shape) -> None:
        """Read ``cpt`` from ``store.flat[lo:hi]`` (viewed as ``shape``) from now on."""
        self.cpt = None  # leaves a previous store, if any
        self._store, self._lo, self._hi, self._shape = store, lo, hi, tuple(shape)
        self._view = self._view_of = None
        store.count += 1
        store.attached += 1

    @classmethod
    def random(cls, size, torch_settings: TorchSettings, name: Optional[str] = None) -> "Node":
        return cls(generate_random_probability_matrix(size, torch_settings), name)

    def __repr__(self):
        return self.name if self.name else super().__repr__()


class BayesianNetwork:
    def __init__(self, nodes: Sequence[Node], parents: Dict[Node, List[Node]]):
        # node -> ordered parents / ordered children; dict order == insertion order
        self._parents: Dict[Node, List[Node]] = {}
        self._children: Dict[Node, List[Node]] = {}

        def touch(n: Node):
            if n not in self._parents:
                self._parents[n] = []
                self._children[n] = []

        seen_edges = set()
        for node in nodes:
            touch(node)
            for parent in parents[node]:
                touch(parent)
                edge = (id(parent), id(node))
                if edge not in seen_edges:
                    seen_edges.add(edge)
                    self._children[parent].append(node)
                    self._parents[node].append(parent)

        self._index: Dict[Node, int] = {n: i for i, n in enumerate(self._parents)}
        self._nx = None

    # --- iteration -------------------------------------------------------
    @property
    def nodes(self) -> Iterator[Node]:
        return iter(self._parents)

    @property
    def edges(self) -> Iterator[Tuple[Node, Node]]:
        return ((parent, child) for parent, kids in self._children.items() for child in kids)

    @property
    def num_nodes(self) -> int:
        return len(self._parents)

    def parents_of(self, node: Node) -> Iterator[Node]:
        return iter(self._parents[node])

    def children_of(self, node: Node) -> Iterator[Node]:
        return iter(self._children[node])

    def is_leaf_node(self, node: Node) -> bool:
        return not self._children[node]

    def index_of(self, node: Node) -> int:
        """Position of ``node`` in ``nodes`` order (constant time)."""
        return self._index[node]

    @property
    def degrees_of_freedom(self) -> int:
        total = 0
        for node in self._parents:
            free = 1
            for d in node.cpt.shape[:-1]:
                free *= int(d)
            total += free * (int(node.cpt.shape[-1]) - 1)
        return total

    # --- graph-library view ---------------------------------------------
    @property
    def G(self):
        """``networkx.DiGraph`` with the same node and edge insertion order."""
        if self._nx is None:
            import networkx as nx

            g = nx.DiGraph()
            g.add_nodes_from(self._parents)
            g.add_edges_from(self.edges)
            self._nx = g
        return self._nx


class NetworkValidationError(Exception):
    pass


def _check_cpts_match_parents(nodes, parents):
    for node in nodes:
        for axis, parent in enumerate(parents[node]):
            if node.cpt.shape[axis] != parent.num_states:
                raise NetworkValidationError(
                    f"Node \"{node}\"'s CPT's {axis}'th dimension should match with "
                    f"corresponding parent \"{parent}\"'s num_states."
                )


def _check_no_duplicate_nodes(nodes, parents):
    seen = set()
    for node in nodes:
        if node in seen:
            raise ValueError(f'Node "{node}" is added more than once.')
        seen.add(node)


def _check_no_isolated_nodes(nodes, parents):
    has_child = set()
    for node in nodes:
        has_child.update(parents[node])
    for node in nodes:
        if not parents[node] and node not in has_child:
            raise NetworkValidationError(
                f'Node "{node}" is isolated, i.e. has no children nor parents.'
            )


def _check_parents_exist(nodes, parents):
    known = set(nodes)
    for node in known:
        for parent in parents[node]:
            if parent not in known:
                raise NetworkValidationError(f'Parent "{parent}" for node "{node}" is not added.')


def _check_no_duplicate_names(nodes, parents):
    names = set()
    for node in nodes:
        if node.name in names:
            raise ValueError(f'A node with name "{node.name}" was already added.')
        names.add(node.name)


_VALIDATORS = (
    _check_no_duplicate_nodes,
    _check_parents_exist,
    _check_cpts_match_parents,
    _check_no_isolated_nodes,
    _check_no_duplicate_names,
)


class BayesianNetworkBuilder:
    def __init__(self):
        self.nodes: List[Node] = []
        self.parents: Dict[Node, List[Node]] = {}

    def add_node(self, node: Node, parents: Union[None, Node, List[Node]] = None):
        self.nodes.append(node)
        self.set_parents(node, parents)
        return self

    def add_nodes(self, nodes: Iterable[Node], parents: Union[None, Node, List[Node]] = None):
        for node in nodes:
            self.add_node(node, parents)
        return self

    def set_parents(self, node: Node, parents: Union[None, Node, List[Node]] = None):
        if isinstance(parents, Node):
            parents = [parents]
        self.parents[node] = list(parents) if parents else []

    def build(self) -> BayesianNetwork:
        for check in _VALIDATORS:
            check(self.nodes, self.parents)
        return BayesianNetwork(self.nodes, self.parents)
