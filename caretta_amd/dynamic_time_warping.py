"""Drop-ins for the reference's ``caretta/dynamic_time_warping.py`` -- same names, arguments and
return values, computed by the gfx950 sweep kernels through the C ABI.

Reference: dynamic_time_warping.py:148-184 (dtw_align), :188-201 (dtw_align_score),
:205-222 (smith_waterman_score), :226-278 (smith_waterman).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, f64, i64, ptr
from .engine import default_context

MIN_FLOAT64 = np.finfo(np.float64).min


def _prep(seq1, seq2, score_matrix):
    s1, s2, s = i64(seq1), i64(seq2), f64(score_matrix)
    if s.ndim != 2:
        raise ValueError("score_matrix must be 2-D")
    return s1, s2, s


def dtw_align(seq1, seq2, score_matrix, gap_open_penalty: float = 0.0, gap_extend_penalty: float = 0.0):
    """-> (aligned_indices_1, aligned_indices_2, score); -1 marks a gap."""
    s1, s2, s = _prep(seq1, seq2, score_matrix)
    n, m = len(s1), len(s2)
    a1 = np.empty(n + m, np.int64)
    a2 = np.empty(n + m, np.int64)
    ln, sc = C.c_int64(0), C.c_double(0.0)
    check(_capi.load().cr_dtw_align(default_context()._h, ptr(s1), n, ptr(s2), m, ptr(s), s.shape[0], s.shape[1],
                                    float(gap_open_penalty), float(gap_extend_penalty), ptr(a1), ptr(a2),
                                    C.byref(ln), C.byref(sc)))
    return a1[:ln.value].copy(), a2[:ln.value].copy(), sc.value


def dtw_align_score(seq1, seq2, score_matrix, gap_open_penalty: float = 0.0, gap_extend_penalty: float = 0.0):
    s1, s2, s = _prep(seq1, seq2, score_matrix)
    sc = C.c_double(0.0)
    check(_capi.load().cr_dtw_align(default_context()._h, ptr(s1), len(s1), ptr(s2), len(s2), ptr(s), s.shape[0],
                                    s.shape[1], float(gap_open_penalty), float(gap_extend_penalty), None, None, None,
                                    C.byref(sc)))
    return sc.value


def smith_waterman_score(seq1, seq2, matrix, gap: float = 0.0):
    s1, s2, s = _prep(seq1, seq2, matrix)
    sc = C.c_double(0.0)
    check(_capi.load().cr_smith_waterman_score(default_context()._h, ptr(s1), len(s1), ptr(s2), len(s2), ptr(s),
                                               s.shape[0], s.shape[1], float(gap), C.byref(sc)))
    return sc.value


def smith_waterman(seq1, seq2, score_matrix, gap: float = 0.0):
    """-> (align1, align2, max_score).  Where the reference fails with ``TypeError`` (no positive
    cell, ``max_pos`` is None at dynamic_time_warping.py:249) this raises ``TypeError`` too."""
    s1, s2, s = _prep(seq1, seq2, score_matrix)
    n, m = len(s1), len(s2)
    a1 = np.empty(n + m, np.int64)
    a2 = np.empty(n + m, np.int64)
    ln, sc, az = C.c_int64(0), C.c_double(0.0), C.c_int(0)
    check(_capi.load().cr_smith_waterman(default_context()._h, ptr(s1), n, ptr(s2), m, ptr(s), s.shape[0], s.shape[1],
                                         float(gap), ptr(a1), ptr(a2), C.byref(ln), C.byref(sc), C.byref(az)))
    if az.value:
        raise TypeError("cannot unpack non-iterable NoneType object (score matrix has no positive local alignment)")
    return a1[:ln.value].copy(), a2[:ln.value].copy(), sc.value
