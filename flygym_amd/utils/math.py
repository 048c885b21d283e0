"""``Rotation3D`` and small helpers (mirror of reference ``src/flygym/utils/math.py:114-175``)."""

from dataclasses import dataclass
from numbers import Number
from typing import Sequence

import numpy as np

__all__ = ["Rotation3D", "Tree", "orderedset"]

_DIMS = {"quat": 4, "axisangle": 4, "xyaxes": 6, "zaxis": 3, "euler": 3}


def orderedset(items):
    """The items without repeats, first occurrences in order."""
    return list(dict.fromkeys(items))


class Tree:
    """An undirected tree over hashable nodes (reference ``utils/math.py``: ``Tree(nodes, edges)``, ``dfs_edges(root)``).

    Construction checks that the graph really is a tree — no repeated nodes, no self loops, no parallel edges, every
    edge between known nodes, connected, acyclic — and raises ``ValueError`` otherwise.  ``dfs_edges(root)`` yields
    ``(parent, child)`` pairs in depth-first order, children in the order their edges were given."""

    def __init__(self, nodes, edges):
        self.nodes = list(nodes)
        self.edges = [tuple(e) for e in edges]
        if len(set(self.nodes)) != len(self.nodes):
            raise ValueError("Tree nodes must be unique.")
        known = set(self.nodes)
        self._adj = {n: [] for n in self.nodes}
        seen_pairs = set()
        for a, b in self.edges:
            if a not in known or b not in known:
                raise ValueError(f"Edge ({a}, {b}) refers to a node that is not in the tree.")
            if a == b or frozenset((a, b)) in seen_pairs:
                raise ValueError("Self loops and parallel edges are not allowed in a tree.")
            seen_pairs.add(frozenset((a, b)))
            self._adj[a].append(b)
            self._adj[b].append(a)
        if self.nodes:
            if len(self.edges) != len(self.nodes) - 1:
                raise ValueError("A tree over n nodes has exactly n - 1 edges (cycle or disconnected graph).")
            reached, todo = set(), [self.nodes[0]]
            while todo:
                n = todo.pop()
                if n not in reached:
                    reached.add(n)
                    todo.extend(self._adj[n])
            if len(reached) != len(self.nodes):
                raise ValueError("The graph is not connected.")

    def dfs_edges(self, root):
        if root not in self._adj:
            raise ValueError(f"Root '{root}' not in tree")
        seen = {root}
        stack = [(root, iter(self._adj[root]))]
        while stack:
            node, it = stack[-1]
            for nb in it:
                if nb not in seen:
                    seen.add(nb)
                    yield node, nb
                    stack.append((nb, iter(self._adj[nb])))
                    break
            else:
                stack.pop()


@dataclass(frozen=True)
class Rotation3D:
    format: str
    values: Sequence[Number]

    def __post_init__(self):
        ok = (
            self.format in _DIMS
            and isinstance(self.values, Sequence)
            and all(isinstance(v, Number) for v in self.values)
        )
        if not ok:
            raise ValueError(
                f"Invalid rotation spec: format={self.format}, values={self.values}. "
                f"Format must be one of {list(_DIMS)} and values must be a sequence of numbers."
            )
        if len(self.values) != _DIMS[self.format]:
            raise ValueError(
                f"Invalid rotation spec: format={self.format}, values={self.values}. "
                f"Format {self.format} should be {_DIMS[self.format]}-dimensional, got {len(self.values)}."
            )

    def as_kwargs(self):
        return {self.format: self.values}

    def as_quat(self) -> np.ndarray:
        """(w, x, y, z); only the formats the engine can spawn from."""
        if self.format == "quat":
            q = np.asarray(self.values, dtype=np.float64)
            return q / np.linalg.norm(q)
        if self.format == "axisangle":
            ax = np.asarray(self.values[:3], dtype=np.float64)
            ax = ax / np.linalg.norm(ax)
            h = 0.5 * float(self.values[3])
            return np.array([np.cos(h), *(ax * np.sin(h))])
        raise ValueError(f"cannot convert rotation format '{self.format}' to a quaternion")
