"""Numeric helpers of the reference's ``caretta/helper.py`` that sit on the pairwise path
(helper.py:13-42).  Integer work, done on the host inside libcaretta_hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, i64, ptr


def get_common_positions(aln_array_1, aln_array_2):
    """Positions where neither alignment row has a gap (-1)."""
    a1, a2 = i64(aln_array_1), i64(aln_array_2)
    if a1.shape != a2.shape:
        raise ValueError("alignment rows must have equal length")
    p1 = np.empty(max(len(a1), 1), np.int64)
    p2 = np.empty(max(len(a1), 1), np.int64)
    k = C.c_int64(0)
    check(_capi.load().cr_get_common_positions(ptr(a1), ptr(a2), len(a1), ptr(p1), ptr(p2), C.byref(k)))
    return p1[:k.value].copy(), p2[:k.value].copy()
