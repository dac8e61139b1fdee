"""Drop-in for the reference's ``caretta/neighbor_joining.py`` (:19-157).  Two implementations in libcaretta_hip
with bit-identical results: host C++ (``cr_neighbor_joining``) and one persistent launch of up to 64 workgroups on
the GPU (``cr_neighbor_joining_device``, symmetric matrices of 256 - 2048 nodes; a launch whose workgroups cannot all
be resident falls back to the host implementation); both form each row sum once per iteration in the reference's
order, so no rounded value and no tie changes."""
from __future__ import annotations

import os

import numpy as np

from . import _capi
from ._capi import check, f64, ptr

# from this many nodes on the device kernel is the faster one (tools/nj_time.py on an MI355X box)
DEVICE_MIN_NODES = int(os.environ.get("CARETTA_NJ_DEVICE_MIN_NODES", "256"))


def neighbor_joining(distance_matrix, device=None, ctx=None):
    """-> (tree uint64 (2P-3, 2) rows (child, parent), branch_lengths float64 (2P-3, 1)).

    ``device``: True = GPU kernel (error without a GPU), False = host C++, None = the GPU kernel when there is a GPU
    and the matrix has at least ``DEVICE_MIN_NODES`` rows.  The result does not depend on the choice."""
    d = f64(distance_matrix)
    if d.ndim != 2 or d.shape[0] != d.shape[1]:
        raise ValueError("distance_matrix must be square")
    p = d.shape[0]
    tree = np.zeros((2 * p - 3, 2), dtype=np.uint64)
    bl = np.zeros((2 * p - 3, 1), dtype=np.float64)
    lib = _capi.load()
    if device is None:
        from . import engine
        device = ctx is not None or (p >= DEVICE_MIN_NODES and engine.device_count() > 0)
    if device:
        if ctx is None:
            from . import engine
            ctx = engine.default_context()
        check(lib.cr_neighbor_joining_device(ctx._h, ptr(d), p, ptr(tree), ptr(bl)))
    else:
        check(lib.cr_neighbor_joining(ptr(d), p, ptr(tree), ptr(bl)))
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
