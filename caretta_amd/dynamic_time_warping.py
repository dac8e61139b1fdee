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


class ExplicitBatch:
    """A list of (seq1, seq2, score_matrix) problems resident in HBM: ``smith_waterman_scores`` / ``dtw_align`` run the
    reference's function over the whole list in one launch sequence (C ABI: cr_explicit_batch).  This is how
    ``MultipleAlignment.make_pairwise_matrix`` serves third-party ``SequenceBase`` plugins (multiple_alignment.py:158-170):
    the plugin's score matrices are computed by its own Python code, the O(P^2) smith_waterman_score calls are not."""

    def __init__(self, problems, context=None):
        problems = list(problems)
        if not problems:
            raise ValueError("need at least one problem")
        self._lib = _capi.load()
        self._ctx = context or default_context()
        mats, seqs = [], []
        desc = np.zeros(len(problems), dtype=_capi.EXPLICIT_PROBLEM_DTYPE)
        s_off = q_off = 0
        for k, (seq1, seq2, matrix) in enumerate(problems):
            s1, s2, s = _prep(seq1, seq2, matrix)
            desc[k] = (s_off, q_off, q_off + len(s1), s.shape[0], s.shape[1], len(s1), len(s2))
            mats.append(s.ravel())
            seqs.extend((s1, s2))
            s_off += s.size
            q_off += len(s1) + len(s2)
        self.shapes = [(int(d["n"]), int(d["m"])) for d in desc]
        self.cells = int(sum(int(d["s_rows"]) * int(d["s_cols"]) for d in desc))
        S = np.concatenate(mats) if len(mats) > 1 else np.ascontiguousarray(mats[0])
        Q = np.concatenate(seqs)
        self._h = C.c_void_p()
        check(self._lib.cr_explicit_batch_create(self._ctx._h, ptr(S), S.size, ptr(Q), Q.size, ptr(desc), len(desc),
                                                 C.byref(self._h)))

    def __len__(self):
        return len(self.shapes)

    def smith_waterman_scores(self, gap: float = 0.0) -> np.ndarray:
        """smith_waterman_score (dynamic_time_warping.py:205-222) of every problem."""
        out = np.zeros(len(self))
        check(self._lib.cr_smith_waterman_score_batch(self._h, float(gap), ptr(out)))
        return out

    def dtw_align(self, gap_open_penalty: float = 0.0, gap_extend_penalty: float = 0.0, want_alignments: bool = True):
        """dtw_align (dynamic_time_warping.py:148-184) of every problem -> list of (aln_1, aln_2, score), or the scores."""
        scores = np.zeros(len(self))
        if not want_alignments:
            check(self._lib.cr_dtw_align_batch(self._h, float(gap_open_penalty), float(gap_extend_penalty), None, 0, None,
                                               ptr(scores)))
            return scores
        stride = max(n + m for n, m in self.shapes)
        aln = np.empty((len(self), 2, stride), dtype=np.int64)
        lens = np.zeros(len(self), dtype=np.int64)
        check(self._lib.cr_dtw_align_batch(self._h, float(gap_open_penalty), float(gap_extend_penalty), ptr(aln), stride,
                                           ptr(lens), ptr(scores)))
        return [(aln[k, 0, :lens[k]].copy(), aln[k, 1, :lens[k]].copy(), float(scores[k])) for k in range(len(self))]

    def smith_waterman(self, gap: float = 0.0):
        """smith_waterman (dynamic_time_warping.py:226-278) of every problem -> list of (aln_1, aln_2, score); a problem
        without a positive cell raises the reference's TypeError (its ``max_pos`` stays None)."""
        stride = max(n + m for n, m in self.shapes)
        aln = np.empty((len(self), 2, stride), dtype=np.int64)
        lens = np.zeros(len(self), dtype=np.int64)
        scores = np.zeros(len(self))
        zero = np.zeros(len(self), dtype=np.int32)
        check(self._lib.cr_smith_waterman_batch(self._h, float(gap), ptr(aln), stride, ptr(lens), ptr(scores), ptr(zero)))
        if zero.any():
            raise TypeError(f"cannot unpack non-iterable NoneType object (score matrix of problem {int(np.nonzero(zero)[0][0])} "
                            "has no positive local alignment)")
        return [(aln[k, 0, :lens[k]].copy(), aln[k, 1, :lens[k]].copy(), float(scores[k])) for k in range(len(self))]

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0.0)
        check(self._lib.cr_explicit_batch_last_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self._h:
            self._lib.cr_explicit_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def smith_waterman_score_batch(problems, gap: float = 0.0) -> np.ndarray:
    """[smith_waterman_score(seq1, seq2, matrix, gap) for (seq1, seq2, matrix) in problems] in one launch."""
    batch = ExplicitBatch(problems)
    try:
        return batch.smith_waterman_scores(gap)
    finally:
        batch.close()


def smith_waterman_batch(problems, gap: float = 0.0):
    """[smith_waterman(seq1, seq2, matrix, gap) for (seq1, seq2, matrix) in problems] in one launch sequence."""
    batch = ExplicitBatch(problems)
    try:
        return batch.smith_waterman(gap)
    finally:
        batch.close()


def dtw_align_batch(problems, gap_open_penalty: float = 0.0, gap_extend_penalty: float = 0.0):
    """[dtw_align(seq1, seq2, matrix, gap_open_penalty, gap_extend_penalty) for ... in problems] in one launch sequence."""
    batch = ExplicitBatch(problems)
    try:
        return batch.dtw_align(gap_open_penalty, gap_extend_penalty)
    finally:
        batch.close()
