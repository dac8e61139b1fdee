"""Numeric helpers of the reference's ``caretta/helper.py`` that sit on the pairwise path
(helper.py:13-42).  Integer work, done on the host inside libcaretta_hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from pathlib import Path

from . import _capi
from ._capi import check, i64, ptr

THREE_TO_ONE = {"ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H",
                "ILE": "I", "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W",
                "TYR": "Y", "VAL": "V", "MSE": "M", "SEC": "U", "PYL": "O"}


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


def nb_mean_axis_0(array) -> np.ndarray:
    """Column means accumulated sequentially, as numba evaluates the reference's helper.nb_mean_axis_0
    (helper.py:46-53)."""
    x = np.ascontiguousarray(array, dtype=np.float64)
    out = np.empty(x.shape[1])
    check(_capi.load().cr_mean_axis0(ptr(x), x.shape[0], x.shape[1], ptr(out)))
    return out


def write_distance_matrix(names, distance_matrix, filename):
    """CLUSTAL-style text matrix: first line = count, then ``name v v v ...`` with 4 decimals
    (format of the reference's helper.write_distance_matrix, helper.py:183-202)."""
    matrix = np.asarray(distance_matrix, dtype=np.float64)
    if matrix.shape != (len(names), len(names)):
        raise ValueError("matrix shape does not match the number of names")
    with open(filename, "w") as handle:
        handle.write(f"{len(names)}\n")
        for name, row in zip(names, matrix):
            handle.write(name + " " + " ".join("%.4f" % value for value in row) + "\n")


def read_distance_matrix(filename):
    """Inverse of write_distance_matrix (helper.py:205-229): names (anything after a ``/`` dropped) and matrix."""
    with open(filename) as handle:
        count = int(handle.readline().strip())
        names, rows = [], []
        for line in handle:
            fields = line.split()
            if not fields:
                continue
            names.append(fields[0].strip().split("/")[0].strip())
            rows.append([float(v) for v in fields[1:count + 1]])
    if len(names) != count:
        raise ValueError(f"{filename}: header says {count} entries, found {len(names)}")
    return names, np.array(rows, dtype=np.float64).reshape(count, count)


def read_calpha_pdb(filename):
    """Minimal C-alpha reader standing in for parse_protein_files_and_clean (helper.py:161-180): ATOM/HETATM(MSE)
    records named CA of the FIRST chain of the FIRST model, first alternate location.
    Returns (coordinates float64 (L, 3), one-letter sequence)."""
    coords, seq = [], []
    chain = None
    seen = set()
    with open(filename) as handle:
        for line in handle:
            record = line[:6]
            if record.startswith("ENDMDL"):
                break
            if record not in ("ATOM  ", "HETATM") or line[12:16].strip() != "CA":
                continue
            resname = line[17:20].strip()
            if record == "HETATM" and resname != "MSE":
                continue
            if line[16] not in (" ", "A"):
                continue
            if chain is None:
                chain = line[21]
            elif line[21] != chain:
                break
            key = (line[22:27])
            if key in seen:
                continue
            seen.add(key)
            coords.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
            seq.append(THREE_TO_ONE.get(resname, "X"))
    if not coords:
        raise ValueError(f"{filename}: no C-alpha atoms found")
    return np.array(coords, dtype=np.float64), "".join(seq)


def local_shape_descriptor(coordinates, width: int = 10):
    """Stand-in per-residue descriptor in [0, 1]^width for real structures when geometricus' learned
    embedding (the reference's tensor source, multiple_alignment.py:479-488) is not available:
    C-alpha distances to the residues at sequence offsets -k..+k (k = 2..), squashed with d / (d + 10 A).
    NOT the reference's tensors: results obtained with it are outside the parity contract."""
    x = np.asarray(coordinates, dtype=np.float64)
    length = x.shape[0]
    offsets = []
    k = 2
    while len(offsets) < width:
        offsets += [-k, k]
        k += 1
    out = np.zeros((length, width))
    for col, off in enumerate(offsets[:width]):
        idx = np.clip(np.arange(length) + off, 0, length - 1)
        dist = np.linalg.norm(x - x[idx], axis=1)
        out[:, col] = dist / (dist + 10.0)
    return out


def nb_std_axis_0(array) -> np.ndarray:
    """Column standard deviations (helper.py:57-64)."""
    return np.std(np.asarray(array, dtype=np.float64), axis=0)


def normalize(numbers) -> np.ndarray:
    """Min-max scaling to [0, 1] (helper.py:67-70)."""
    numbers = np.asarray(numbers, dtype=np.float64)
    lo, hi = np.min(numbers), np.max(numbers)
    return (numbers - lo) / (hi - lo)
