"""Drop-in for the reference's ``caretta/neighbor_joining.py`` (:19-157): host-side C++ in
libcaretta_hip, row sums hoisted out of the pair loop without changing any rounded value."""
from __future__ import annotations

import numpy as np

from . import _capi
from ._capi import check, f64, ptr


def neighbor_joining(distance_matrix):
    """-> (tree uint64 (2P-3, 2) rows (child, parent), branch_lengths float64 (2P-3, 1))."""
    d = f64(distance_matrix)
    if d.ndim != 2 or d.shape[0] != d.shape[1]:
        raise ValueError("distance_matrix must be square")
    p = d.shape[0]
    tree = np.zeros((2 * p - 3, 2), dtype=np.uint64)
    bl = np.zeros((2 * p - 3, 1), dtype=np.float64)
    check(_capi.load().cr_neighbor_joining(ptr(d), p, ptr(tree), ptr(bl)))
    return tree, bl


def bipartitions(tree, num_leaves: int):
    """Unrooted bipartition set of a guide tree (the topology; child order within a join is not)."""
    tree = np.asarray(tree, dtype=np.int64)
    children = {}
    for child, parent in tree:
        children.setdefault(int(parent), []).append(int(child))
    memo = {}

    def leaves(node):
        if node < num_leaves:
            return frozenset([node])
        if node not in memo:
            acc = frozenset()
            for c in children.get(node, []):
                acc |= leaves(c)
            memo[node] = acc
        return memo[node]

    full = frozenset(range(num_leaves))
    out = set()
    for kids in children.values():
        for c in kids:
            side = leaves(c)
            if 1 < len(side) < num_leaves - 1:
                out.add(min(side, full - side, key=lambda s: (len(s), sorted(s))))
    return out
