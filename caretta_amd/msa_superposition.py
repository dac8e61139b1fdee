"""Post-alignment products of a finished multiple alignment, mirroring the reference's
``caretta/multiple_alignment.py``: ``make_coverage_gap_distance_matrix`` (:45-56), ``get_reference_structures``
(:741-783), ``superpose`` / ``superpose_core`` / ``superpose_reference`` / ``superpose_references`` (:896-997) and
``make_rmsd_coverage_tm_matrix`` (:1000-1055).  Kabsch fits, rigid moves and the pairwise RMSD / coverage / TM
matrices run on the GPU through the C ABI (the matrices as one batched launch, one wave per pair); the set
logic around them is integer work on the host.  As in the reference, ``superpose*`` replace each protein's
``coordinates``.
"""
from __future__ import annotations

import typing

import numpy as np

from . import _capi, helper
from . import superposition_functions as sup
from ._capi import check, f64, ptr
from .engine import default_context

GAP = -1


def make_coverage_gap_distance_matrix(alignment_array):
    """For rows i, j: fraction of i's residues that face a gap in j, and the number that face a residue."""
    present = np.asarray(alignment_array) != GAP
    counts = present.sum(axis=1)
    aligning = (present[:, None, :] & present[None, :, :]).sum(axis=2).astype(np.int32)
    distance = (counts[:, None] - aligning) / counts[:, None]
    return distance.astype(np.float64), aligning


def get_reference_structures(alignment, minimum_coverage=50, gap=GAP):
    """Greedy cover: a first reference (smallest median gap distance), then further references until every
    structure has one that covers at least ``minimum_coverage`` percent of its residues."""
    names = list(alignment.keys())
    rows = np.array([alignment[name] for name in names])
    distance, aligning = make_coverage_gap_distance_matrix(rows)
    needed = np.array([minimum_coverage * np.count_nonzero(np.asarray(alignment[name]) != gap) / 100 for name in names])
    first = int(np.argmin(np.median(distance, axis=0)))
    ok = aligning[:, first] >= needed
    covered = list(np.where(ok)[0])
    pending = np.where(~ok)[0]
    groups = {first: [names[c] for c in covered]}
    stuck: typing.List[int] = []
    while len(pending) > 0:
        block = distance[pending, :][:, covered]
        ref = covered[int(np.argmin(np.median(block, axis=0)))] if len(pending) > 1 else covered[int(np.argmin(block))]
        takes = aligning[pending, ref] >= needed[pending]
        if not takes.any():
            stuck += list(pending)
            break
        groups[ref] = [names[c] for c in pending[takes]]
        covered += list(pending[takes])
        pending = pending[~takes]
    orphans = []
    for i in stuck:
        home = next((j for j in covered if aligning[i, j] >= needed[i]), None)
        if home is None:
            orphans.append(names[i])
        else:
            groups[home].append(names[i])
    return names[first], {names[k]: v for k, v in groups.items()}, orphans


def _index_of(proteins, name):
    return next(i for i, p in enumerate(proteins) if p.name == name)


def superpose_core(alignment, proteins, reference_name, core_indices=None, gap=GAP):
    """Fit every structure onto the reference over the gap-free columns (multiple_alignment.py:914-950)."""
    if core_indices is None:
        rows = np.array([alignment[n] for n in alignment])
        core_indices = np.where((rows != gap).all(axis=0))[0]
    core_indices = np.asarray(core_indices, dtype=np.int64)
    ref = _index_of(proteins, reference_name)
    # all structures in one launch (cr_superpose_core): paired_svd_superpose on the core columns + apply_rotran
    lens = [len(p) for p in proteins]
    offsets = np.zeros(len(proteins) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    coords = np.ascontiguousarray(np.vstack([f64(p.coordinates) for p in proteins]))
    msa = np.ascontiguousarray(np.array([alignment[p.name] for p in proteins]), dtype=np.int32)
    core = np.ascontiguousarray(core_indices, dtype=np.int32)
    moved = np.empty_like(coords)
    check(_capi.load().cr_superpose_core(default_context()._h, ptr(coords), ptr(offsets), len(proteins), ptr(msa), msa.shape[1],
                                         ptr(core), len(core), ref, ptr(moved)))
    for i, protein in enumerate(proteins):
        protein.coordinates = moved[offsets[i]:offsets[i + 1]].copy()
    return proteins


def superpose_reference(alignment, proteins, reference_name):
    """Fit every structure onto the reference over the positions the two share (multiple_alignment.py:953-972).
    The reference itself is refitted in turn, exactly as the reference's loop does."""
    ref = _index_of(proteins, reference_name)
    lens = [len(p) for p in proteins]
    offsets = np.zeros(len(proteins) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    coords = np.ascontiguousarray(np.vstack([f64(p.coordinates) for p in proteins]))
    msa = np.ascontiguousarray(np.array([alignment[p.name] for p in proteins]), dtype=np.int32)
    moved = np.empty_like(coords)
    try:
        check(_capi.load().cr_superpose_reference(default_context()._h, ptr(coords), ptr(offsets), len(proteins), ptr(msa),
                                                  msa.shape[1], ref, ptr(moved)))
    except ValueError as e:
        if "3 or fewer" in str(e):
            raise AssertionError(str(e)) from None
        raise
    for i, protein in enumerate(proteins):
        protein.coordinates = moved[offsets[i]:offsets[i + 1]].copy()
    return proteins


def superpose_references(alignment, proteins, minimum_coverage=50):
    """Fit every structure onto the reference structure chosen for it (multiple_alignment.py:975-997)."""
    order = [p.name for p in proteins]
    index = {name: i for i, name in enumerate(order)}
    _, groups, _ = get_reference_structures(alignment, minimum_coverage)
    lens = [len(p) for p in proteins]
    offsets = np.zeros(len(proteins) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    coords = np.ascontiguousarray(np.vstack([f64(p.coordinates) for p in proteins]))
    msa = np.ascontiguousarray(np.array([alignment[name] for name in order]), dtype=np.int32)
    lib, ctx = _capi.load(), default_context()._h

    def fit(ref, members):
        which = np.ascontiguousarray([index[name] for name in members], dtype=np.int32)
        if len(which):
            try:
                check(lib.cr_superpose_members(ctx, ptr(coords), ptr(offsets), len(proteins), ptr(msa), msa.shape[1], ref,
                                               ptr(which), len(which)))
            except ValueError as e:
                if "3 or fewer" in str(e):
                    raise AssertionError(str(e)) from None
                raise

    # the groups in order; inside a group the members are independent, except that a reference listed among its own
    # members is refitted onto itself when its turn comes and the members after it see the refitted copy (:985-995)
    for reference_name, members in groups.items():
        ref = index[reference_name]
        if reference_name in members:
            at = members.index(reference_name)
            fit(ref, members[:at])
            fit_self = np.ascontiguousarray([ref], dtype=np.int32)
            check(lib.cr_superpose_members(ctx, ptr(coords), ptr(offsets), len(proteins), ptr(msa), msa.shape[1], ref,
                                           ptr(fit_self), 1))
            fit(ref, members[at + 1:])
        else:
            fit(ref, members)
    for i, protein in enumerate(proteins):
        protein.coordinates = coords[offsets[i]:offsets[i + 1]].copy()
    return proteins


def superpose(alignment, proteins, gap=GAP, verbose=False):
    """Reference = the row with most residues (first of equals); superpose on the gap-free core when it spans
    at least half of the reference's row, otherwise pairwise on the reference (multiple_alignment.py:896-911)."""
    names = list(alignment.keys())
    counts = [np.count_nonzero(np.asarray(alignment[n]) != gap) for n in names]
    reference_name = names[int(np.argmax(counts))]
    rows = np.array([alignment[n] for n in names])
    core = np.where((rows != gap).all(axis=0))[0]
    if verbose:
        print("Core indices", len(core))
    if len(core) < len(alignment[reference_name]) // 2:
        return superpose_reference(alignment, proteins, reference_name)
    return superpose_core(alignment, proteins, reference_name, core)


def make_rmsd_coverage_tm_matrix(alignment, proteins, superpose_first: bool = True):
    """RMSD, coverage and TM matrices of every pair of rows (multiple_alignment.py:1000-1055), one batched GPU
    launch.  ``superpose_first=True`` moves the proteins with ``superpose`` first (in place, as the reference
    does) and compares them as they are; ``False`` fits each pair on its own common positions."""
    if superpose_first:
        proteins = superpose(alignment, proteins)
    names = [p.name for p in proteins]
    lens = [len(p) for p in proteins]
    offsets = np.zeros(len(proteins) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    coords = np.ascontiguousarray(np.vstack([f64(p.coordinates) for p in proteins]))
    msa = np.ascontiguousarray(np.array([alignment[n] for n in names]), dtype=np.int32)
    num = len(proteins)
    rmsd, coverage, tm = np.empty((num, num)), np.empty((num, num)), np.empty((num, num))
    check(_capi.load().cr_msa_metrics(default_context()._h, ptr(coords), ptr(offsets), num, ptr(msa), msa.shape[1],
                                      0 if superpose_first else 1, ptr(rmsd), ptr(coverage), ptr(tm)))
    if np.isnan(rmsd).any():
        raise AssertionError("a pair of rows shares fewer than 3 positions (reference: assert len(pos_1) >= 3)")
    return rmsd, coverage, tm
