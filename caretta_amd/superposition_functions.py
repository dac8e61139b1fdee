"""Drop-ins for the reference's ``caretta/superposition_functions.py`` (:7-80)."""
from __future__ import annotations

import numpy as np

from . import _capi
from ._capi import check, f64, ptr
from .engine import default_context


def paired_svd_superpose(coords_1, coords_2):
    """Kabsch: rotation (3,3) and translation (3,) with coords_2 @ R + t ~ coords_1."""
    x1, x2 = f64(coords_1), f64(coords_2)
    if x1.shape != x2.shape or x1.ndim != 2 or x1.shape[1] != 3:
        raise ValueError("paired coordinates must both have shape (k, 3)")
    r, t = np.empty((3, 3)), np.empty(3)
    check(_capi.load().cr_paired_svd_superpose(default_context()._h, ptr(x1), ptr(x2), x1.shape[0], ptr(r), ptr(t)))
    return r, t


def paired_svd_superpose_with_subset(coords_1, coords_2, common_coords_1, common_coords_2):
    c1, c2, s1, s2 = f64(coords_1), f64(coords_2), f64(common_coords_1), f64(common_coords_2)
    o1, o2, o3 = np.empty_like(c1), np.empty_like(c2), np.empty_like(s2)
    check(_capi.load().cr_paired_svd_superpose_with_subset(default_context()._h, ptr(c1), c1.shape[0], ptr(c2),
                                                           c2.shape[0], ptr(s1), ptr(s2), s1.shape[0], ptr(o1),
                                                           ptr(o2), ptr(o3)))
    return o1, o2, o3


def apply_rotran(coords, rotation_matrix, translation_matrix) -> np.ndarray:
    x, r, t = f64(coords), f64(rotation_matrix), f64(translation_matrix)
    out = np.empty_like(x)
    check(_capi.load().cr_apply_rotran(default_context()._h, ptr(x), x.shape[0], ptr(r), ptr(t), ptr(out)))
    return out
