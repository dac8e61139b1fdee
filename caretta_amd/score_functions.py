"""Drop-ins for the reference's ``caretta/score_functions.py`` (score_functions.py:7-51)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, f64, ptr
from .engine import default_context


def get_gaussian_score(coord_1, coord_2, gamma: float = 0.03):
    """exp(-gamma * sum((c1 - c2)**2)) for one pair of feature vectors (score_functions.py:7-11)."""
    a = f64(coord_1).reshape(1, -1)
    b = f64(coord_2).reshape(1, -1)
    return float(make_score_matrix(a, b, get_gaussian_score, gamma)[0, 0])


def get_rmsd(coords_1, coords_2) -> float:
    x1, x2 = f64(coords_1), f64(coords_2)
    out = C.c_double(0.0)
    check(_capi.load().cr_get_rmsd(default_context()._h, ptr(x1), ptr(x2), x1.shape[0], C.byref(out)))
    return out.value


def make_score_matrix(coords_1, coords_2, score_function, gamma, normalized: bool = False) -> np.ndarray:
    """(n, m) matrix of ``score_function`` between all rows (score_functions.py:23-51).

    ``get_gaussian_score`` -- the only function the reference ever passes -- runs as a HIP kernel.  Any other
    callable (a third-party plugin's cell score) is applied cell by cell on the host exactly as the reference's loop
    does (score_functions.py:48-50): that is the plugin's own code, not a fallback of the accelerated path.
    ``normalized=True`` (no caller in the reference) z-scores both inputs with the mean / standard deviation of their
    concatenation first (score_functions.py:43-47).
    """
    a, b = f64(coords_1), f64(coords_2)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("coords_1 and coords_2 must be 2-D with equal width")
    if normalized:
        from . import helper
        both = np.concatenate((a, b))
        mean, std = helper.nb_mean_axis_0(both), helper.nb_std_axis_0(both)
        a, b = f64((a - mean) / std), f64((b - mean) / std)
    if score_function is not get_gaussian_score:
        if not callable(score_function):
            raise TypeError("score_function must be callable")
        return np.array([[score_function(row, col, gamma) for col in b] for row in a], dtype=np.float64).reshape(len(a), len(b))
    s = np.zeros((a.shape[0], b.shape[0]))
    check(_capi.load().cr_make_score_matrix(default_context()._h, ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1],
                                            float(gamma), ptr(s)))
    return s
