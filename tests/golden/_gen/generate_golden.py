#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own source.

Run only in the build container (the reference tree does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/_gen/generate_golden.py

How the reference is executed: ``/root/reference/caretta`` is imported with the
name-only stand-ins in ``_gen/stubs`` first on ``sys.path`` (``numba.njit`` -> identity,
empty ``prody``/``Bio``/``geometricus`` names).  The hot-path modules are plain numpy code
under ``@nb.njit``, so CPython runs the reference's statements unchanged.  Only inputs
and outputs are written; no reference source text is stored.

Deltas of this CPython execution against a real numba run:
``np.exp`` is numpy's SIMD exp instead of libm's (<=1 ulp apart), ``np.sum``/``np.mean`` use
numpy's pairwise order instead of numba's sequential order, BLAS/LAPACK builds differ.
Integer semantics (argmax tie-breaks, loop order, traceback) are the reference's own.
"""
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[2]
sys.dont_write_bytecode = True
sys.path.insert(0, str(HERE / "stubs"))
sys.path.insert(1, os.environ.get("CARETTA_REFERENCE", "/root/reference"))
sys.path.insert(2, str(REPO))

import numpy as np  # noqa: E402

from caretta import dynamic_time_warping as dtw  # noqa: E402
from caretta import helper, multiple_alignment, neighbor_joining, score_functions  # noqa: E402
from caretta import superposition_functions as sup  # noqa: E402

from caretta_amd import synthetic  # noqa: E402

OUT = HERE.parent
SF_PARAMS = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03)


def save(name, arrays):
    arrays = dict(arrays)
    arrays["_numpy_version"] = np.array(np.__version__)
    path = OUT / name
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path.name}: {len(arrays)} arrays, {path.stat().st_size / 1024:.0f} KiB")


# --------------------------------------------------------------------------- F1
def gen_score_matrix(rng):
    out = {}
    cases = [(1, 1, 3, 0.03), (2, 3, 3, 0.03), (5, 17, 3, 0.03), (37, 64, 3, 0.03), (65, 100, 3, 0.03),
             (17, 5, 10, 7.0), (64, 65, 10, 7.0), (100, 37, 10, 7.0), (3, 2, 1, 1.0), (64, 17, 1, 1.0),
             (40, 40, 16, 2.5), (150, 150, 3, 0.03)]
    for c, (n, m, k, gamma) in enumerate(cases):
        if k == 3:
            a = rng.normal(scale=15.0, size=(n, k))
            b = rng.normal(scale=15.0, size=(m, k))
        else:
            a = rng.uniform(size=(n, k))
            b = rng.uniform(size=(m, k))
        s = score_functions.make_score_matrix(a, b, score_functions.get_gaussian_score, gamma)
        out[f"c{c}_a"], out[f"c{c}_b"], out[f"c{c}_gamma"], out[f"c{c}_S"] = a, b, np.float64(gamma), s
    # far-apart chains: RBF underflows to subnormals / exact zero
    a = rng.normal(scale=5.0, size=(20, 3))
    b = rng.normal(scale=5.0, size=(23, 3)) + np.array([150.0, 20.0, 0.0])
    c = len(cases)
    out[f"c{c}_a"], out[f"c{c}_b"], out[f"c{c}_gamma"] = a, b, np.float64(0.03)
    out[f"c{c}_S"] = score_functions.make_score_matrix(a, b, score_functions.get_gaussian_score, 0.03)
    out["ncases"] = np.int64(c + 1)
    save("f1_score_matrix.npz", out)


def rbf_pair(rng, n, m, noise=1.5):
    """A related pair of C-alpha traces already roughly superposed."""
    base = synthetic._walk(rng, max(n, m) + 8)
    a = base[2:2 + n] + rng.normal(scale=noise, size=(n, 3))
    b = base[5:5 + m] + rng.normal(scale=noise, size=(m, 3))
    return score_functions.make_score_matrix(a, b, score_functions.get_gaussian_score, 0.03)


def dp_inputs(rng):
    """Score matrices shared by the dtw_align and smith_waterman fixtures."""
    mats = []
    for n, m in [(1, 1), (1, 5), (5, 1), (2, 3), (3, 3), (5, 17), (17, 5), (17, 17), (37, 64), (64, 37),
                 (64, 65), (65, 64), (100, 100), (100, 65), (150, 150), (129, 150)]:
        mats.append(("rbf", rbf_pair(rng, n, m)))
    mats.append(("ones", np.ones((7, 9))))
    mats.append(("ones", np.ones((16, 16))))
    mats.append(("identity", np.eye(12)))
    mats.append(("identity", np.eye(9, 14)))
    mats.append(("checker", (np.add.outer(np.arange(13), np.arange(11)) % 2).astype(np.float64)))
    mats.append(("small_ints", rng.integers(0, 3, size=(15, 14)).astype(np.float64)))
    mats.append(("small_ints", rng.integers(0, 2, size=(33, 70)).astype(np.float64)))
    far_a = rng.normal(scale=5.0, size=(20, 3))
    far_b = rng.normal(scale=5.0, size=(23, 3)) + np.array([148.0, 20.0, 0.0])
    mats.append(("underflow", score_functions.make_score_matrix(far_a, far_b, score_functions.get_gaussian_score, 0.03)))
    mats.append(("uniform", rng.uniform(size=(70, 66))))
    mats.append(("negative", rng.normal(size=(24, 31))))
    return mats


def gen_dtw(rng, mats):
    out = {}
    penalties = [(0.0, 0.0), (1.0, 0.01), (0.5, 0.5), (3.0, 0.1)]
    c = 0
    for kind, s in mats:
        n, m = s.shape
        for (go, ge) in penalties:
            a1, a2, score = dtw.dtw_align(np.arange(n), np.arange(m), s, go, ge)
            out[f"c{c}_S"], out[f"c{c}_open"], out[f"c{c}_extend"] = s, np.float64(go), np.float64(ge)
            out[f"c{c}_aln1"], out[f"c{c}_aln2"], out[f"c{c}_score"] = a1.astype(np.int64), a2.astype(np.int64), np.float64(score)
            out[f"c{c}_score2"] = np.float64(dtw.dtw_align_score(np.arange(n), np.arange(m), s, go, ge))
            if n <= 17 and m <= 17:
                mat, bt = dtw._make_dtw_matrix(np.arange(n), np.arange(m), s, go, ge)
                out[f"c{c}_matrix"], out[f"c{c}_backtrack"] = mat, bt
            c += 1
    # alphabet mode: score matrix over an alphabet, sequences are index strings
    for (la, n, m) in [(4, 11, 14), (20, 40, 33)]:
        s = rng.normal(size=(la, la))
        s = s + s.T + 2.0 * np.eye(la)
        s1 = rng.integers(0, la, size=n).astype(np.int64)
        s2 = rng.integers(0, la, size=m).astype(np.int64)
        for (go, ge) in penalties[1:3]:
            a1, a2, score = dtw.dtw_align(s1, s2, s, go, ge)
            out[f"c{c}_S"], out[f"c{c}_open"], out[f"c{c}_extend"] = s, np.float64(go), np.float64(ge)
            out[f"c{c}_seq1"], out[f"c{c}_seq2"] = s1, s2
            out[f"c{c}_aln1"], out[f"c{c}_aln2"], out[f"c{c}_score"] = a1.astype(np.int64), a2.astype(np.int64), np.float64(score)
            out[f"c{c}_score2"] = np.float64(dtw.dtw_align_score(s1, s2, s, go, ge))
            c += 1
    out["ncases"] = np.int64(c)
    save("f1_dtw.npz", out)


def gen_sw(rng, mats):
    out = {}
    c = 0
    for kind, s in mats:
        n, m = s.shape
        for gap in (0.0, 0.25):
            if not (s > 0).any():
                continue
            try:
                a1, a2, score = dtw.smith_waterman(np.arange(n), np.arange(m), s, gap)
            except TypeError:
                continue  # all-zero H: the reference crashes (max_pos None)
            score2 = dtw.smith_waterman_score(np.arange(n), np.arange(m), s, gap)
            out[f"c{c}_S"], out[f"c{c}_gap"] = s, np.float64(gap)
            out[f"c{c}_aln1"], out[f"c{c}_aln2"] = a1.astype(np.int64), a2.astype(np.int64)
            out[f"c{c}_score"], out[f"c{c}_score_only"] = np.float64(score), np.float64(score2)
            c += 1
    out["ncases"] = np.int64(c)
    save("f1_sw.npz", out)


def gen_kabsch(rng):
    out = {}
    c = 0

    def add(x1, x2, tag):
        nonlocal c
        rot, tran = sup.paired_svd_superpose(x1, x2)
        moved = sup.apply_rotran(x2, rot, tran)
        out[f"c{c}_x1"], out[f"c{c}_x2"], out[f"c{c}_R"], out[f"c{c}_t"] = x1, x2, rot, tran
        out[f"c{c}_moved"] = moved
        out[f"c{c}_rmsd"] = np.float64(score_functions.get_rmsd(x1, moved))
        out[f"c{c}_tag"] = np.array(tag)
        c += 1

    for k in (4, 5, 17, 64, 150, 300):
        x1 = synthetic._walk(rng, k)
        x2 = (x1 + rng.normal(scale=1.0, size=x1.shape)) @ synthetic._random_rotation(rng) + rng.uniform(-50, 50, 3)
        add(x1, x2, "noisy")
    x1 = synthetic._walk(rng, 40)
    add(x1, x1 @ synthetic._random_rotation(rng) + np.array([3.0, -7.0, 11.0]), "rigid")
    add(x1, x1 * np.array([1.0, 1.0, -1.0]), "mirror")
    add(x1, rng.normal(scale=20.0, size=x1.shape), "unrelated")
    add(x1[:3], x1[:3] @ synthetic._random_rotation(rng), "three_points")
    out["ncases"] = np.int64(c)
    # with_subset
    for s, (n, m, k) in enumerate([(30, 34, 12), (150, 150, 90)]):
        a = synthetic._walk(rng, n)
        b = synthetic._walk(rng, m) @ synthetic._random_rotation(rng) + 9.0
        p1 = np.sort(rng.choice(n, size=k, replace=False))
        p2 = np.sort(rng.choice(m, size=k, replace=False))
        o1, o2, o3 = sup.paired_svd_superpose_with_subset(a, b, a[p1], b[p2])
        out[f"s{s}_a"], out[f"s{s}_b"], out[f"s{s}_p1"], out[f"s{s}_p2"] = a, b, p1.astype(np.int64), p2.astype(np.int64)
        out[f"s{s}_o1"], out[f"s{s}_o2"], out[f"s{s}_o3"] = o1, o2, o3
    out["nsubset"] = np.int64(2)
    save("f1_kabsch.npz", out)


def gen_kabsch_degenerate():
    """Rank-deficient correlation matrices through the reference's own paired_svd_superpose (LAPACK dgesdd): collinear
    positions (rank 1), coincident positions (rank 0), planar (rank 2), the same inputs tests/test_oracle_golden.py builds
    in degenerate_kabsch_cases().  LAPACK's choice of the null-space basis is not the Jacobi SVD's: R is stored so that the
    tests can RECORD the distance, the superposed coordinates / RMSD so that they can compare what is determined."""
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
    from test_oracle_golden import degenerate_kabsch_cases
    out = {}
    c = 0
    for tag, x1, x2 in degenerate_kabsch_cases():
        rot, tran = sup.paired_svd_superpose(x1, x2)
        moved = sup.apply_rotran(x2, rot, tran)
        out[f"c{c}_x1"], out[f"c{c}_x2"], out[f"c{c}_R"], out[f"c{c}_t"], out[f"c{c}_moved"] = x1, x2, rot, tran, moved
        out[f"c{c}_rmsd"] = np.float64(score_functions.get_rmsd(x1, moved))
        out[f"c{c}_tag"] = np.array(tag)
        c += 1
    out["ncases"] = np.int64(c)
    save("f8_kabsch_degenerate.npz", out)


def gen_misc(rng):
    out = {}
    # get_common_positions
    alns = [(np.array([0, -1, 1]), np.array([0, 1, -1])),
            (np.array([-1, -1, 0, 1, 2, -1, 3]), np.array([0, 1, 2, -1, 3, 4, 5])),
            (np.array([-1, -1]), np.array([0, 1])),
            (np.arange(10), np.arange(10))]
    for c, (a1, a2) in enumerate(alns):
        p1, p2 = helper.get_common_positions(a1, a2)
        out[f"cp{c}_a1"], out[f"cp{c}_a2"] = a1.astype(np.int64), a2.astype(np.int64)
        out[f"cp{c}_p1"], out[f"cp{c}_p2"] = np.asarray(p1, dtype=np.int64), np.asarray(p2, dtype=np.int64)
    out["ncp"] = np.int64(len(alns))
    # tm_score and get_rmsd
    c = 0
    for k, l1, l2 in [(3, 10, 12), (5, 14, 16), (20, 15, 30), (50, 60, 80), (100, 100, 100), (140, 150, 300)]:
        x1 = rng.normal(scale=10.0, size=(k, 3))
        x2 = x1 + rng.normal(scale=1.5, size=(k, 3))
        out[f"tm{c}_x1"], out[f"tm{c}_x2"], out[f"tm{c}_l1"], out[f"tm{c}_l2"] = x1, x2, np.int64(l1), np.int64(l2)
        out[f"tm{c}_tm"] = np.float64(multiple_alignment.tm_score(x1, x2, l1, l2))
        out[f"tm{c}_rmsd"] = np.float64(score_functions.get_rmsd(x1, x2))
        c += 1
    out["ntm"] = np.int64(c)
    # apply_rotran
    x = rng.normal(size=(9, 3))
    rot = synthetic._random_rotation(rng)
    tr = rng.normal(size=3)
    out["ar_x"], out["ar_R"], out["ar_t"], out["ar_out"] = x, rot, tr, sup.apply_rotran(x, rot, tr)
    save("f1_misc.npz", out)


# --------------------------------------------------------------------------- F2
def to_proteins(family):
    return [multiple_alignment.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in family]


def pipeline_h(pi, pj, out, key):
    """One pair through pipeline H using only reference functions."""
    n, m = len(pi), len(pj)
    s_t = score_functions.make_score_matrix(pi.tensors, pj.tensors, score_functions.get_gaussian_score, 7.0)
    flags = 0
    if (s_t > 0).any():
        sa1, sa2, sw_t = dtw.smith_waterman(np.arange(n), np.arange(m), s_t, 0.0)
        pos1, pos2 = helper.get_common_positions(sa1, sa2)
    else:
        flags |= 4
        sa1 = sa2 = pos1 = pos2 = np.zeros(0, dtype=np.int64)
        sw_t = 0.0
    if flags & 4:
        s_c = None
    else:
        s_c = pi.score_function(pj, verbose=False, **SF_PARAMS)
        if len(pos1) <= 3:
            flags |= 1
            chk = score_functions.make_score_matrix(np.array(pi.coordinates), np.array(pj.coordinates),
                                                    score_functions.get_gaussian_score, 0.03)
        else:
            c1, c2, _ = sup.paired_svd_superpose_with_subset(pi.coordinates, pj.coordinates,
                                                             pi.coordinates[pos1], pj.coordinates[pos2])
            chk = score_functions.make_score_matrix(c1, c2, score_functions.get_gaussian_score, 0.03)
        assert np.array_equal(chk, s_c), "step-by-step pipeline differs from Protein.score_function"
    out[f"{key}_seed_aln1"], out[f"{key}_seed_aln2"] = np.asarray(sa1, np.int64), np.asarray(sa2, np.int64)
    out[f"{key}_seed_score"] = np.float64(sw_t)
    if s_c is None:
        out[f"{key}_flags"] = np.int64(flags)
        return
    sw = dtw.smith_waterman_score(np.arange(n), np.arange(m), s_c, 0.0)
    a1, a2, dscore = dtw.dtw_align(np.arange(n), np.arange(m), s_c, 1.0, 0.01)
    q1, q2 = helper.get_common_positions(a1, a2)
    out[f"{key}_sw"], out[f"{key}_dtw_score"] = np.float64(sw), np.float64(dscore)
    out[f"{key}_aln1"], out[f"{key}_aln2"] = np.asarray(a1, np.int64), np.asarray(a2, np.int64)
    if len(q1) >= 3:
        x1, x2 = pi.coordinates[q1], pj.coordinates[q2]
        rot, tran = sup.paired_svd_superpose(x1, x2)
        x2m = sup.apply_rotran(x2, rot, tran)
        out[f"{key}_R"], out[f"{key}_t"] = rot, tran
        out[f"{key}_rmsd"] = np.float64(score_functions.get_rmsd(x1, x2m))
        out[f"{key}_coverage"] = np.float64(x1.shape[0] / len(a1))
        out[f"{key}_tm"] = np.float64(multiple_alignment.tm_score(x1, x2m, n, m))
    else:
        flags |= 2
    out[f"{key}_flags"] = np.int64(flags)


def store_family(out, prefix, family):
    coords, tensors, offsets = synthetic.pack(family)
    out[f"{prefix}_coords"], out[f"{prefix}_tensors"], out[f"{prefix}_offsets"] = coords, tensors, offsets


def gen_pipeline():
    out = {}
    fams = {
        "A": synthetic.make_family(8, 60, seed=20231, ragged=True),
        "B": synthetic.make_family(6, 150, seed=20232),
        "C": synthetic.make_family(2, 300, seed=20233, clades=1),
        "D": synthetic.make_family(3, 310, seed=20234, ragged=True, clades=1),
    }
    # family E: includes a 3-residue member (k<=3 branch) and a 5-residue one
    e = synthetic.make_family(4, 40, seed=20235, clades=1)
    e[1] = synthetic.Structure("tiny3", e[1].tensors[:3].copy(), e[1].coordinates[:3].copy(), "AAA")
    e[3] = synthetic.Structure("tiny5", e[3].tensors[10:15].copy(), e[3].coordinates[10:15].copy(), "AAAAA")
    fams["E"] = e
    for tag, fam in fams.items():
        t0 = time.time()
        store_family(out, f"fam{tag}", fam)
        prots = to_proteins(fam)
        pairs = [(i, j) for i in range(len(prots)) for j in range(i + 1, len(prots))]
        out[f"fam{tag}_pairs"] = np.array(pairs, dtype=np.int32)
        for p, (i, j) in enumerate(pairs):
            pipeline_h(prots[i], prots[j], out, f"fam{tag}_p{p}")
        print(f"  family {tag}: {len(pairs)} pairs in {time.time() - t0:.1f}s")
    out["families"] = np.array(list(fams.keys()))
    save("f2_pipeline.npz", out)


def gen_flexible():
    """flexible=True (multiple_alignment.py:323-326): the reference's own make_pairwise_matrix on two small families -- the
    P x P matrix of smith_waterman_score over the TENSOR score matrices alone -- and one score matrix / alignment of its
    two-sequence branch (smith_waterman on the flexible score function's matrix, :226-278)."""
    out = {}
    for tag, fam in (("FA", synthetic.make_family(8, 60, seed=20241, ragged=True)), ("FB", synthetic.make_family(3, 340, seed=20242, clades=1))):
        store_family(out, f"fam{tag}", fam)
        msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
        out[f"fam{tag}_M"] = msa.make_pairwise_matrix(score_function_params=dict(flexible=True, gamma_tensor=SF_PARAMS["gamma_tensor"]))
        prots = to_proteins(fam)
        s = prots[0].score_function(prots[1], flexible=True, gamma_tensor=SF_PARAMS["gamma_tensor"])
        a1, a2, score = dtw.smith_waterman(np.arange(s.shape[0]), np.arange(s.shape[1]), s, gap=0.0)
        if tag == "FA":                      # (the 340 x 340 matrix of FB would be 0.9 MB: its alignment and score are kept)
            out[f"fam{tag}_S01"] = s
        out[f"fam{tag}_sw_aln1"], out[f"fam{tag}_sw_aln2"] = a1.astype(np.int64), a2.astype(np.int64)
        out[f"fam{tag}_sw_score"] = np.float64(score)
    out["families"] = np.array(["FA", "FB"])
    save("f9_flexible.npz", out)


def gen_pipeline_long():
    out = {}
    fam = synthetic.make_family(2, 1200, seed=20236, clades=1)
    store_family(out, "famL", fam)
    prots = to_proteins(fam)
    out["famL_pairs"] = np.array([(0, 1)], dtype=np.int32)
    t0 = time.time()
    pipeline_h(prots[0], prots[1], out, "famL_p0")
    print(f"  family L (1200 residues): {time.time() - t0:.1f}s")
    out["families"] = np.array(["L"])
    save("f2_pipeline_long.npz", out)


# --------------------------------------------------------------------------- F3 / F4
def gen_tree(rng):
    out = {}
    for tag, fam in (("T8", synthetic.make_family(8, 60, seed=20231, ragged=True)),
                     ("T16", synthetic.make_family(16, 80, seed=20237, ragged=True))):
        store_family(out, f"fam{tag}", fam)
        msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
        m = msa.make_pairwise_matrix(score_function_params=dict(SF_PARAMS, verbose=False))
        d = m.max() - m
        tree, bl = neighbor_joining.neighbor_joining(d.copy())
        out[f"fam{tag}_M"], out[f"fam{tag}_D"] = m, d
        out[f"fam{tag}_tree"], out[f"fam{tag}_branch_lengths"] = tree, bl
        if tag == "T8":
            aln = msa.multiple_align(d, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0,
                                     gamma_weight=1.0, score_function_params=dict(SF_PARAMS, verbose=False),
                                     mean_function_params=dict(flexible=False, verbose=False))
            out["famT8_msa"] = np.array([aln[s.name] for s in fam], dtype=np.int64)
    c = 0
    for p in (3, 4, 5, 8, 16, 40, 64):
        for rep in range(2 if p >= 8 else 1):
            x = rng.uniform(0.5, 10.0, size=(p, p))
            d = (x + x.T) / 2
            np.fill_diagonal(d, rng.uniform(0.0, 3.0))  # non-zero diagonal, as on the real path
            tree, bl = neighbor_joining.neighbor_joining(d.copy())
            out[f"nj{c}_D"], out[f"nj{c}_tree"], out[f"nj{c}_branch_lengths"] = d, tree, bl
            c += 1
    out["nnj"] = np.int64(c)
    save("f3_tree.npz", out)


def gen_tree64():
    """Pairwise matrix and neighbor-joining tree of a 64-structure family, by the reference's own
    make_pairwise_matrix (multiple_alignment.py:158-170) and neighbor_joining (neighbor_joining.py:19-157)."""
    out = {}
    fam = synthetic.make_family(64, 48, seed=20238, ragged=True, clades=6)
    store_family(out, "famT64", fam)
    msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
    m = msa.make_pairwise_matrix(score_function_params=dict(SF_PARAMS, verbose=False))
    d = m.max() - m
    tree, bl = neighbor_joining.neighbor_joining(d.copy())
    out["famT64_M"], out["famT64_D"] = m, d
    out["famT64_tree"], out["famT64_branch_lengths"] = tree, bl
    save("f3_tree64.npz", out)


def gen_formats(rng):
    """The text the reference's writers produce (helper.write_distance_matrix, helper.py:183-202;
    MultipleAlignment.write_alignment, multiple_alignment.py:299-309; to_sequence_alignment :287-297) and what its
    reader returns (helper.read_distance_matrix, :205-229), for fixed inputs."""
    import tempfile
    out = {}
    names = ["1kdu", "1pk4/A", "kringle_domain_3", "x"]
    d = rng.uniform(0.0, 25.0, size=(4, 4))
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0.0)
    d[0, 1] = d[1, 0] = 1234567.891234          # wide numbers, rounding at the 4th decimal
    d[2, 3] = d[3, 2] = 0.00005
    d[1, 3] = d[3, 1] = 2.00005
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "matrix.mat"
        helper.write_distance_matrix(names, d, path)
        out["matrix_text"] = np.frombuffer(path.read_bytes(), dtype=np.uint8)
        back_names, back = helper.read_distance_matrix(path)
        out["matrix_names"], out["matrix_D"] = np.array(names), d
        out["matrix_read_names"], out["matrix_read_D"] = np.array(back_names), back
        fam = synthetic.make_family(5, 30, seed=20239, ragged=True, clades=2)
        letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
        fam = [synthetic.Structure(s.name, s.tensors, s.coordinates, "".join(rng.choice(letters, size=len(s.sequence))))
               for s in fam]
        store_family(out, "fasta_fam", fam)
        out["fasta_names"] = np.array([s.name for s in fam])
        out["fasta_sequences"] = np.array([s.sequence for s in fam])
        msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
        # a hand-made alignment (gaps at both ends and inside), independent of any DP
        width = max(len(s.sequence) for s in fam) + 4
        aln = {}
        for k, s in enumerate(fam):
            row = np.full(width, -1, dtype=np.int64)
            start = k % 3
            idx = np.arange(len(s.sequence))
            cols = start + idx + (idx > len(s.sequence) // 2)      # one internal gap
            row[cols] = idx
            aln[s.name] = row
        out["fasta_alignment"] = np.array([aln[s.name] for s in fam])
        fasta = Path(tmp) / "aln.fasta"
        msa.write_alignment(fasta, aln)
        out["fasta_text"] = np.frombuffer(fasta.read_bytes(), dtype=np.uint8)
        seq_aln = msa.to_sequence_alignment(aln)
        out["fasta_rows"] = np.array([seq_aln[s.name] for s in fam])
    save("f7_formats.npz", out)


# --------------------------------------------------------------------------- F4: progressive alignment internals
def gen_progressive():
    """Every intermediate node of progressive_align (multiple_alignment.py:172-253) for two small families:
    the node's mean tensors / coordinates / consensus weights and the final alignment."""
    out = {}
    for tag, fam in (("P8", synthetic.make_family(8, 60, seed=20231, ragged=True)),
                     ("P5", synthetic.make_family(5, 40, seed=20238, ragged=True, clades=2))):
        store_family(out, f"fam{tag}", fam)
        msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
        m = msa.make_pairwise_matrix(score_function_params=dict(SF_PARAMS, verbose=False))
        d = m.max() - m
        aln = msa.multiple_align(d, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0,
                                 gamma_weight=1.0, score_function_params=dict(SF_PARAMS, verbose=False),
                                 mean_function_params=dict(flexible=False, verbose=False))
        out[f"fam{tag}_D"] = d
        out[f"fam{tag}_tree"] = msa.tree
        out[f"fam{tag}_msa"] = np.array([aln[s.name] for s in fam], dtype=np.int64)
        nleaf = len(fam)
        out[f"fam{tag}_nnodes"] = np.int64(len(msa.final_sequences) - nleaf)
        for k, node in enumerate(msa.final_sequences[nleaf:]):
            out[f"fam{tag}_n{k}_tensors"] = node.tensors
            out[f"fam{tag}_n{k}_coords"] = node.coordinates
            out[f"fam{tag}_n{k}_weights"] = msa.final_consensus_weights[nleaf + k]
            out[f"fam{tag}_n{k}_name"] = np.array(node.name)
    save("f4_progressive.npz", out)


def gen_flexible_progressive():
    """flexible=True all the way (multiple_alignment.py:323-326 score function, :351-362 mean function): the reference's own
    make_pairwise_matrix -> neighbor_joining -> progressive_align on two small families; every intermediate node carries
    mean tensors and consensus weights only (no coordinates)."""
    out = {}
    sf = dict(flexible=True, gamma_tensor=SF_PARAMS["gamma_tensor"])
    for tag, fam in (("G8", synthetic.make_family(8, 60, seed=20251, ragged=True)),
                     ("G5", synthetic.make_family(5, 90, seed=20252, ragged=True, clades=2))):
        store_family(out, f"fam{tag}", fam)
        msa = multiple_alignment.MultipleAlignment(to_proteins(fam))
        m = msa.make_pairwise_matrix(score_function_params=sf)
        d = m.max() - m
        aln = msa.multiple_align(d, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0, gamma_weight=1.0,
                                 score_function_params=sf, mean_function_params=dict(flexible=True))
        out[f"fam{tag}_D"] = d
        out[f"fam{tag}_tree"] = msa.tree
        out[f"fam{tag}_msa"] = np.array([aln[s.name] for s in fam], dtype=np.int64)
        nleaf = len(fam)
        out[f"fam{tag}_nnodes"] = np.int64(len(msa.final_sequences) - nleaf)
        for k, node in enumerate(msa.final_sequences[nleaf:]):
            assert node.coordinates is None
            out[f"fam{tag}_n{k}_tensors"] = node.tensors
            out[f"fam{tag}_n{k}_weights"] = msa.final_consensus_weights[nleaf + k]
            out[f"fam{tag}_n{k}_name"] = np.array(node.name)
    out["families"] = np.array(["G8", "G5"])
    save("f10_flexible_progressive.npz", out)


def gen_post_msa():
    """Post-alignment products on the 8-structure family's MSA: superpose() (multiple_alignment.py:896-997),
    make_coverage_gap_distance_matrix (:45-56), get_reference_structures (:741-783),
    make_rmsd_coverage_tm_matrix with and without superpose_first (:1000-1055)."""
    import copy
    import contextlib
    import io
    out = {}
    g = np.load(OUT / "f4_progressive.npz")
    for tag in ("P8", "P5"):
        fam_coords, fam_tensors, off = g[f"fam{tag}_coords"], g[f"fam{tag}_tensors"], g[f"fam{tag}_offsets"]
        num = len(off) - 1
        names = [f"s{i:04d}" for i in range(num)]
        prots = [multiple_alignment.Protein(names[i], fam_tensors[off[i]:off[i + 1]].copy(), fam_coords[off[i]:off[i + 1]].copy(), "")
                 for i in range(num)]
        aln = {names[i]: g[f"fam{tag}_msa"][i] for i in range(num)}
        arr = np.array([aln[n] for n in names])
        dist, aligning = multiple_alignment.make_coverage_gap_distance_matrix(arr)
        out[f"{tag}_cov_dist"], out[f"{tag}_cov_aligning"] = dist, aligning
        first, refs, none = multiple_alignment.get_reference_structures(aln, 50)
        out[f"{tag}_ref_first"] = np.array(first)
        out[f"{tag}_ref_keys"] = np.array(list(refs.keys()))
        for k, v in refs.items():
            out[f"{tag}_ref_{k}"] = np.array(v)
        out[f"{tag}_ref_none"] = np.array(none, dtype=str)
        r0, c0, t0 = multiple_alignment.make_rmsd_coverage_tm_matrix(aln, copy.deepcopy(prots), superpose_first=False)
        out[f"{tag}_rmsd"], out[f"{tag}_coverage"], out[f"{tag}_tm"] = r0, c0, t0
        moved = copy.deepcopy(prots)
        with contextlib.redirect_stdout(io.StringIO()):
            r1, c1, t1 = multiple_alignment.make_rmsd_coverage_tm_matrix(aln, moved, superpose_first=True)
        out[f"{tag}_rmsd_sf"], out[f"{tag}_coverage_sf"], out[f"{tag}_tm_sf"] = r1, c1, t1
        for i in range(num):
            out[f"{tag}_superposed_{i}"] = moved[i].coordinates
        ref_moved = multiple_alignment.superpose_reference(aln, copy.deepcopy(prots), names[1])
        for i in range(num):
            out[f"{tag}_superposed_ref1_{i}"] = ref_moved[i].coordinates
    save("f5_post_msa.npz", out)


def gen_c1_inputs():
    """BASELINE config 1 inputs: C-alpha coordinates and sequences of the three kringle-domain PDB files the
    reference ships as its README example (test_data/), read with the product's own minimal reader."""
    from caretta_amd import helper as amd_helper
    out = {}
    ref_root = Path(sys.modules["caretta"].__path__[0]).parent
    names = []
    for path in sorted((ref_root / "test_data").glob("*.pdb")):
        xyz, seq = amd_helper.read_calpha_pdb(path)
        out[f"{path.stem}_coords"], out[f"{path.stem}_sequence"] = xyz, np.array(seq)
        names.append(path.stem)
    out["names"] = np.array(names)
    save("c1_kringle_calpha.npz", out)


def gen_extras(rng):
    """Small helpers next to the path: the normalized=True branch of make_score_matrix (no caller in the reference),
    alignment_to_numpy, helper.normalize / nb_std_axis_0."""
    out = {}
    c = 0
    for n, m, k, gamma in [(7, 5, 3, 0.03), (20, 31, 10, 7.0), (1, 4, 1, 1.0)]:
        a = rng.normal(scale=3.0, size=(n, k)) + 2.0
        b = rng.normal(scale=3.0, size=(m, k)) - 1.0
        out[f"ns{c}_a"], out[f"ns{c}_b"], out[f"ns{c}_gamma"] = a, b, np.float64(gamma)
        out[f"ns{c}_S"] = score_functions.make_score_matrix(a, b, score_functions.get_gaussian_score, gamma, normalized=True)
        c += 1
    out["nns"] = np.int64(c)
    gapped = {"a": "AC-D-", "b": "--XYZ", "c": "-----", "d": "MKV"}
    got = multiple_alignment.alignment_to_numpy(gapped)
    for key, seq in gapped.items():
        out[f"a2n_{key}_in"] = np.frombuffer(seq.encode(), dtype=np.uint8)
        out[f"a2n_{key}_out"] = np.asarray(got[key], dtype=np.int64)
    x = rng.normal(size=(11, 4))
    out["std_x"], out["std_out"] = x, helper.nb_std_axis_0(x)
    v = rng.uniform(-3, 9, size=13)
    out["norm_x"], out["norm_out"] = v, helper.normalize(v)
    save("f6_extras.npz", out)


def main():
    only = set(sys.argv[1:])
    rng = np.random.default_rng(20230)
    print("generating golden vectors from", sys.modules["caretta"].__path__)
    steps = [("score", lambda: gen_score_matrix(rng))]
    mats = dp_inputs(np.random.default_rng(20229))
    steps += [("dtw", lambda: gen_dtw(np.random.default_rng(20228), mats)),
              ("sw", lambda: gen_sw(np.random.default_rng(20227), mats)),
              ("kabsch", lambda: gen_kabsch(np.random.default_rng(20226))),
              ("kabsch_degenerate", gen_kabsch_degenerate),
              ("misc", lambda: gen_misc(np.random.default_rng(20225))),
              ("pipeline", gen_pipeline),
              ("long", gen_pipeline_long),
              ("flexible", gen_flexible),
              ("tree", lambda: gen_tree(np.random.default_rng(20224))),
              ("tree64", gen_tree64),
              ("formats", lambda: gen_formats(np.random.default_rng(20222))),
              ("progressive", gen_progressive),
              ("flexprog", gen_flexible_progressive),
              ("c1", gen_c1_inputs),
              ("postmsa", gen_post_msa),
              ("extras", lambda: gen_extras(np.random.default_rng(20223)))]
    for name, fn in steps:
        if only and name not in only:
            continue
        t0 = time.time()
        print(f"[{name}]")
        fn()
        print(f"  {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
