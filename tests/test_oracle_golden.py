"""Pin the C oracle against golden vectors produced by the reference's own source.

Golden vectors: tests/golden/*.npz (generator: tests/golden/_gen/generate_golden.py).
Bar: integers exact; floats within the tolerances of SURVEY.md section 8(c).
"""
import numpy as np
import pytest

from oracle.pyoracle import tree_bipartitions

RTOL_S = 1e-12      # exp differs from numpy's by <= 1 ulp, d^2 summation order by a few ulp


def test_exp_accuracy(oracle, oracle_libm):
    rng = np.random.default_rng(7)
    x = np.concatenate([-rng.uniform(0, 750, 20000), rng.uniform(0, 709, 2000), -10 ** rng.uniform(-20, 0, 2000),
                        [0.0, -0.0, -745.2, -745.13, -745.14, -708.4, -708.39, 709.78, 709.79, -1e-300, 1e-300]])
    ours = oracle.exp(x)
    libm = oracle_libm.exp(x)
    with np.errstate(over="ignore"):
        ref = np.exp(x.astype(np.longdouble)).astype(np.float64)  # 64-bit mantissa reference, rounded
    fin = np.isfinite(ref)
    assert np.array_equal(np.isinf(ours), ~fin) and np.array_equal(np.isinf(libm), ~fin)
    ours, libm, ref = ours[fin], libm[fin], ref[fin]
    ulp = np.spacing(np.maximum(np.abs(ref), 5e-324))
    assert np.all(np.abs(ours - ref) <= ulp), "own exp is more than 1 ulp off"
    assert np.all(np.abs(ours - libm) <= ulp)
    frac = np.mean(ours != libm)
    assert frac < 0.02, f"own exp differs from libm on {frac:.3%} of arguments"
    assert oracle.exp(np.array([-746.0, -1000.0, -1e308]))[0] == 0.0
    assert np.isinf(oracle.exp(np.array([710.0]))[0])   # (NaN arguments never reach exp: inputs are validated finite)
    assert oracle.exp(np.array([0.0]))[0] == 1.0


def test_make_score_matrix(oracle, golden):
    g = golden("f1_score_matrix.npz")
    for c in range(int(g["ncases"])):
        for mode in (0, 1):
            oracle.set_sum_mode(mode)
            s = oracle.make_score_matrix(g[f"c{c}_a"], g[f"c{c}_b"], float(g[f"c{c}_gamma"]))
            ref = g[f"c{c}_S"]
            assert s.shape == ref.shape
            np.testing.assert_allclose(s, ref, rtol=RTOL_S, atol=5e-324 * 4)
            assert np.array_equal(s == 0, ref == 0) or np.max(np.abs(s - ref)) < 1e-320
    oracle.set_sum_mode(0)


def _dtw_case(g, c):
    s = g[f"c{c}_S"]
    if f"c{c}_seq1" in g:
        return g[f"c{c}_seq1"], g[f"c{c}_seq2"], s
    return np.arange(s.shape[0]), np.arange(s.shape[1]), s


def test_dtw_align(oracle, golden):
    g = golden("f1_dtw.npz")
    n = int(g["ncases"])
    assert n > 100
    for c in range(n):
        s1, s2, s = _dtw_case(g, c)
        go, ge = float(g[f"c{c}_open"]), float(g[f"c{c}_extend"])
        small = f"c{c}_matrix" in g
        res = oracle.dtw_align(s1, s2, s, go, ge, want_matrices=small)
        assert np.array_equal(res[0], g[f"c{c}_aln1"]), c
        assert np.array_equal(res[1], g[f"c{c}_aln2"]), c
        assert res[2] == float(g[f"c{c}_score"]), c          # same S in, same adds: bit-exact
        assert oracle.dtw_align_score(s1, s2, s, go, ge) == float(g[f"c{c}_score2"])
        if small:
            assert np.array_equal(res[4], g[f"c{c}_backtrack"]), c
            assert np.array_equal(res[3], g[f"c{c}_matrix"]), c


def test_smith_waterman(oracle, golden):
    g = golden("f1_sw.npz")
    n = int(g["ncases"])
    assert n > 30
    for c in range(n):
        s = g[f"c{c}_S"]
        a, b = np.arange(s.shape[0]), np.arange(s.shape[1])
        gap = float(g[f"c{c}_gap"])
        a1, a2, sc, rc = oracle.smith_waterman(a, b, s, gap)
        assert rc == 0
        assert np.array_equal(a1, g[f"c{c}_aln1"]) and np.array_equal(a2, g[f"c{c}_aln2"]), c
        assert sc == float(g[f"c{c}_score"])
        assert oracle.smith_waterman_score(a, b, s, gap) == float(g[f"c{c}_score_only"])


def test_smith_waterman_all_zero_flag(oracle):
    a1, a2, sc, rc = oracle.smith_waterman(np.arange(4), np.arange(5), np.zeros((4, 5)), 0.0)
    assert rc == 1 and len(a1) == 0 and sc == 0.0


def degenerate_kabsch_cases():
    """Rank-deficient correlation matrices: collinear positions (rank 1), coincident positions (rank 0), planar (rank 2)."""
    rng = np.random.default_rng(77)
    cases = []
    for _ in range(6):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ts = rng.normal(size=(12, 1)) * 5
        line1 = rng.normal(size=3) + ts * d
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        line2 = (line1 - line1.mean(0)) @ q + rng.normal(size=3)
        cases.append(("collinear", line1, line2))
    axis = np.zeros((9, 3))
    axis[:, 0] = np.arange(9.0)
    cases.append(("collinear on x", axis, axis[::-1].copy()))
    cases.append(("coincident", np.ones((7, 3)) * 2.5, np.ones((7, 3)) * -1.0))
    plane = rng.normal(size=(15, 3))
    plane[:, 2] = 0.0
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    cases.append(("planar", plane, plane @ q + 1.0))
    return cases


def check_degenerate_kabsch_against_lapack(g, superpose):
    """f8_kabsch_degenerate.npz: what the reference's LAPACK SVD returns for rank-deficient correlation matrices.  A rotation
    about a line of collinear positions is not determined by the data, so R may differ (the distance is returned, per case,
    for the record); what IS determined -- where the positions land, the RMSD, R a proper rotation -- must agree."""
    dist = {}
    for c in range(int(g["ncases"])):
        tag, x1, x2 = str(g[f"c{c}_tag"]), g[f"c{c}_x1"], g[f"c{c}_x2"]
        r, t = superpose(x1, x2)
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-12, err_msg=tag)
        assert abs(np.linalg.det(r) - 1.0) < 1e-12 and abs(np.linalg.det(g[f"c{c}_R"]) - 1.0) < 1e-12, tag
        moved = x2 @ r + t
        np.testing.assert_allclose(moved, g[f"c{c}_moved"], atol=1e-9, err_msg=tag)
        rmsd = float(np.sqrt(((x1 - moved) ** 2).sum() / len(x1)))
        assert abs(rmsd - float(g[f"c{c}_rmsd"])) < 1e-5, tag
        dist.setdefault(tag, []).append(float(np.linalg.norm(r - g[f"c{c}_R"])))
        if tag in ("coincident", "planar"):               # rank 0: both return the identity; rank 2: R is determined
            np.testing.assert_allclose(r, g[f"c{c}_R"], atol=1e-9, err_msg=tag)
    return dist


def test_kabsch_degenerate_against_lapack(oracle, golden):
    dist = check_degenerate_kabsch_against_lapack(golden("f8_kabsch_degenerate.npz"), oracle.paired_svd_superpose)
    # recorded: collinear positions leave the rotation about the line free -- LAPACK's R and the Jacobi SVD's are 1.5 .. 2.8
    # apart (Frobenius) while the superposed positions agree to 1e-14
    assert max(dist["collinear"]) > 1.0 and max(dist["coincident"]) == 0.0 and max(dist["planar"]) < 1e-12


def test_kabsch_rank_deficient(oracle):
    """The reference's LAPACK SVD returns an orthonormal U for any input; so must the Jacobi SVD (collinear seeds)."""
    for tag, x1, x2 in degenerate_kabsch_cases():
        r, t = oracle.paired_svd_superpose(x1, x2)
        assert np.all(np.isfinite(r)) and np.all(np.isfinite(t)), tag
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-12, err_msg=tag)
        assert abs(np.linalg.det(r) - 1.0) < 1e-12, tag
        moved = oracle.apply_rotran(x2, r, t)
        # the superposition is optimal: same RMSD as numpy's LAPACK-based Kabsch
        c1, c2 = x1.mean(0), x2.mean(0)
        u, sv, vh = np.linalg.svd((x2 - c2).T @ (x1 - c1))
        if np.linalg.det(u) * np.linalg.det(vh) < 0:
            u[:, -1] = -u[:, -1]
        ref = (x2 - c2) @ (u @ vh) + c1
        assert abs(oracle.get_rmsd(x1, moved) - oracle.get_rmsd(x1, ref)) < 1e-9, tag


def test_kabsch(oracle, golden):
    g = golden("f1_kabsch.npz")
    for c in range(int(g["ncases"])):
        x1, x2 = g[f"c{c}_x1"], g[f"c{c}_x2"]
        r, t = oracle.paired_svd_superpose(x1, x2)
        tag = str(g[f"c{c}_tag"])
        assert abs(np.linalg.det(r) - 1.0) < 1e-12
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(r, g[f"c{c}_R"], atol=1e-9, err_msg=tag)
        np.testing.assert_allclose(t, g[f"c{c}_t"], atol=1e-8, err_msg=tag)
        moved = oracle.apply_rotran(x2, r, t)
        np.testing.assert_allclose(moved, g[f"c{c}_moved"], atol=1e-8)
        assert abs(oracle.get_rmsd(x1, moved) - float(g[f"c{c}_rmsd"])) < 1e-9
    for s in range(int(g["nsubset"])):
        a, b = g[f"s{s}_a"], g[f"s{s}_b"]
        o1, o2, o3 = oracle.paired_svd_superpose_with_subset(a, b, a[g[f"s{s}_p1"]], b[g[f"s{s}_p2"]])
        np.testing.assert_allclose(o1, g[f"s{s}_o1"], atol=1e-9)
        np.testing.assert_allclose(o2, g[f"s{s}_o2"], atol=1e-8)
        np.testing.assert_allclose(o3, g[f"s{s}_o3"], atol=1e-8)


def test_svd3_properties(oracle):
    rng = np.random.default_rng(3)
    mats = [rng.normal(size=(3, 3)) * 10 ** rng.uniform(-3, 3) for _ in range(200)]
    mats += [np.outer(rng.normal(size=3), rng.normal(size=3)), np.zeros((3, 3)), np.eye(3), np.diag([3.0, 3.0, 1.0]),
             np.diag([2.0, 1.0, 0.0])]
    for c in mats:
        u, s, vt = oracle.svd3(c)
        scale = max(np.abs(c).max(), 1e-300)
        np.testing.assert_allclose(u @ np.diag(s) @ vt, c, atol=1e-13 * scale)
        assert s[0] >= s[1] >= s[2] >= 0
        np.testing.assert_allclose(s, np.linalg.svd(c, compute_uv=False), atol=1e-13 * scale)
        if s[1] > 1e-9 * max(s[0], 1e-300):
            np.testing.assert_allclose(u.T @ u, np.eye(3), atol=1e-12)
            np.testing.assert_allclose(vt @ vt.T, np.eye(3), atol=1e-12)


def test_misc(oracle, golden):
    g = golden("f1_misc.npz")
    for c in range(int(g["ncp"])):
        p1, p2 = oracle.get_common_positions(g[f"cp{c}_a1"], g[f"cp{c}_a2"])
        assert np.array_equal(p1, g[f"cp{c}_p1"]) and np.array_equal(p2, g[f"cp{c}_p2"])
    for c in range(int(g["ntm"])):
        x1, x2 = g[f"tm{c}_x1"], g[f"tm{c}_x2"]
        tm = oracle.tm_score(x1, x2, int(g[f"tm{c}_l1"]), int(g[f"tm{c}_l2"]))
        assert abs(tm - float(g[f"tm{c}_tm"])) <= 1e-13 * max(1.0, abs(tm))
        assert abs(oracle.get_rmsd(x1, x2) - float(g[f"tm{c}_rmsd"])) < 1e-13
        oracle.set_sum_mode(1)
        assert oracle.get_rmsd(x1, x2) == float(g[f"tm{c}_rmsd"])  # numpy summation order: bit-exact
        oracle.set_sum_mode(0)
    np.testing.assert_allclose(oracle.apply_rotran(g["ar_x"], g["ar_R"], g["ar_t"]), g["ar_out"], atol=1e-14)


def _check_pipeline(orc, g, fam):
    coords, tensors, offsets = g[f"fam{fam}_coords"], g[f"fam{fam}_tensors"], g[f"fam{fam}_offsets"]
    pairs = g[f"fam{fam}_pairs"]
    outs, aln = orc.pairwise_batch(coords, tensors, offsets, pairs)
    for p in range(len(pairs)):
        key = f"fam{fam}_p{p}"
        o = outs[p]
        assert int(o["flags"]) == int(g[key + "_flags"]), key
        assert int(o["seed_len"]) == len(g[key + "_seed_aln1"]), key
        assert abs(o["seed_score"] - float(g[key + "_seed_score"])) <= 1e-11 * max(1.0, abs(o["seed_score"]))
        ln = int(o["aln_len"])
        assert np.array_equal(aln[p, 0, :ln], g[key + "_aln1"]), key
        assert np.array_equal(aln[p, 1, :ln], g[key + "_aln2"]), key
        assert np.all(aln[p, :, ln:] == -2)
        sw_ref = float(g[key + "_sw"])
        assert abs(o["sw"] - sw_ref) <= 1e-9 * max(1.0, abs(sw_ref)), key
        d_ref = float(g[key + "_dtw_score"])
        assert abs(o["dtw_score"] - d_ref) <= 1e-9 * max(1.0, abs(d_ref)), key
        if not int(o["flags"]) & 2:
            np.testing.assert_allclose(o["R"].reshape(3, 3), g[key + "_R"], atol=1e-9)
            np.testing.assert_allclose(o["t"], g[key + "_t"], atol=1e-7)
            assert abs(o["rmsd"] - float(g[key + "_rmsd"])) < 1e-9
            assert o["coverage"] == float(g[key + "_coverage"])
            assert abs(o["tm"] - float(g[key + "_tm"])) < 1e-9
    return outs


@pytest.mark.parametrize("fam", ["A", "B", "C", "D", "E"])
def test_pipeline_h(oracle, golden, fam):
    _check_pipeline(oracle, golden("f2_pipeline.npz"), fam)


def test_pipeline_h_libm_exp_same_integers(oracle, oracle_libm, golden):
    """libm exp (what numba calls) and the shared exp give the same integer outputs."""
    g = golden("f2_pipeline.npz")
    for fam in ("A", "E"):
        a = _check_pipeline(oracle_libm, g, fam)
        b = _check_pipeline(oracle, g, fam)
        assert np.array_equal(a["aln_len"], b["aln_len"])
        np.testing.assert_allclose(a["sw"], b["sw"], rtol=1e-13)


def test_pipeline_h_long(oracle, golden):
    _check_pipeline(oracle, golden("f2_pipeline_long.npz"), "L")


def test_pairwise_matrix_and_tree(oracle, golden):
    g = golden("f3_tree.npz")
    for fam in ("T8", "T16"):
        offsets = g[f"fam{fam}_offsets"]
        p = len(offsets) - 1
        pairs = np.array([(i, j) for i in range(p) for j in range(i + 1, p)], dtype=np.int32)
        outs, _ = oracle.pairwise_batch(g[f"fam{fam}_coords"], g[f"fam{fam}_tensors"], offsets, pairs, want_aln=False,
                                        nthreads=4)
        m = np.zeros((p, p))
        m[pairs[:, 0], pairs[:, 1]] = outs["sw"]
        m[pairs[:, 1], pairs[:, 0]] = outs["sw"]
        np.testing.assert_allclose(m, g[f"fam{fam}_M"], rtol=1e-9)
        d = m.max() - m                                   # multiple_alignment.py:501
        tree, bl = oracle.neighbor_joining(d)
        assert tree_bipartitions(tree, p) == tree_bipartitions(g[f"fam{fam}_tree"], p)
        np.testing.assert_allclose(np.sort(bl.ravel()), np.sort(g[f"fam{fam}_branch_lengths"].ravel()), atol=1e-7)


def test_pairwise_matrix_and_tree_64(oracle, oracle_libm, golden):
    """A 64-structure family through the reference's make_pairwise_matrix + neighbor_joining (f3_tree64.npz): the oracle's
    matrix to 1e-9, the same bipartitions -- with the shared exp and with libm's exp (what numba calls)."""
    g = golden("f3_tree64.npz")
    offsets = g["famT64_offsets"]
    p = len(offsets) - 1
    pairs = np.array([(i, j) for i in range(p) for j in range(i + 1, p)], dtype=np.int32)
    want = tree_bipartitions(g["famT64_tree"], p)
    for orc in (oracle, oracle_libm):
        outs, _ = orc.pairwise_batch(g["famT64_coords"], g["famT64_tensors"], offsets, pairs, want_aln=False, nthreads=8)
        m = np.zeros((p, p))
        m[pairs[:, 0], pairs[:, 1]] = outs["sw"]
        m[pairs[:, 1], pairs[:, 0]] = outs["sw"]
        np.testing.assert_allclose(m, g["famT64_M"], rtol=1e-9)
        tree, _ = orc.neighbor_joining(m.max() - m)
        assert tree_bipartitions(tree, p) == want


def test_neighbor_joining(oracle, golden):
    g = golden("f3_tree.npz")
    for c in range(int(g["nnj"])):
        d = g[f"nj{c}_D"]
        p = d.shape[0]
        oracle.set_sum_mode(1)                            # numpy's summation order -> byte-exact tree
        for hoist in (False, True) if p <= 40 else (True,):
            tree, bl = oracle.neighbor_joining(d, hoist=hoist)
            assert np.array_equal(tree, g[f"nj{c}_tree"]), (c, hoist)
            np.testing.assert_allclose(bl, g[f"nj{c}_branch_lengths"], rtol=0, atol=1e-12)
        oracle.set_sum_mode(0)                            # numba's order: same topology
        tree, bl = oracle.neighbor_joining(d)
        assert tree_bipartitions(tree, p) == tree_bipartitions(g[f"nj{c}_tree"], p)


@pytest.mark.parametrize("tag", ["P8", "P5"])
def test_progressive_alignment(oracle, golden, tag):
    """Oracle restatement of make_intermediate_node / mean_function / get_mean_weights driven over the
    golden guide tree: identical MSA, node tensors and weights exact, node coordinates to 1e-9."""
    g = golden("f4_progressive.npz")
    coords, tensors, off = g[f"fam{tag}_coords"], g[f"fam{tag}_tensors"], g[f"fam{tag}_offsets"]
    p = len(off) - 1
    nodes = [(coords[off[i]:off[i + 1]], tensors[off[i]:off[i + 1]], np.full((off[i + 1] - off[i], 1), 1.0)) for i in range(p)]
    alns = [{i: np.arange(off[i + 1] - off[i])} for i in range(p)]
    tree = g[f"fam{tag}_tree"].astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]
    for n1, n2 in joins:
        tot = len(alns[n1]) + len(alns[n2])            # multiple_alignment.py:199-202
        a1, a2, xn, tn, wn, _ = oracle.progressive_node(*nodes[n1], *nodes[n2], len(alns[n2]) / (2 * tot), len(alns[n1]) / (2 * tot))
        merged = {k: np.array([v[i] if i != -1 else -1 for i in a1]) for k, v in alns[n1].items()}
        merged.update({k: np.array([v[i] if i != -1 else -1 for i in a2]) for k, v in alns[n2].items()})
        nodes.append((xn, tn, wn))
        alns.append(merged)
    assert np.array_equal(np.array([alns[-1][i] for i in range(p)]), g[f"fam{tag}_msa"])
    for k in range(int(g[f"fam{tag}_nnodes"])):
        xn, tn, wn = nodes[p + k]
        np.testing.assert_allclose(xn, g[f"fam{tag}_n{k}_coords"], atol=1e-9)
        assert np.array_equal(tn, g[f"fam{tag}_n{k}_tensors"]) and np.array_equal(wn, g[f"fam{tag}_n{k}_weights"])


def tree_joins(tree):
    tree = np.asarray(tree).astype(np.int64)
    return [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]


@pytest.mark.parametrize("tag", ["G8", "G5"])
def test_flexible_progressive_alignment(oracle, golden, tag):
    """f10_flexible_progressive.npz -- the reference's own multiple_align with flexible=True in score AND mean function -- replayed
    by the oracle's flexible node (multiple_alignment.py:323-326, :193-217, :351-362, :73-82) over the golden guide tree:
    identical MSA, node tensors and consensus weights exact (the mean of two tensors rounds the same way everywhere)."""
    g = golden("f10_flexible_progressive.npz")
    tensors, off = g[f"fam{tag}_tensors"], g[f"fam{tag}_offsets"]
    p = len(off) - 1
    nodes = [(tensors[off[i]:off[i + 1]], np.full((off[i + 1] - off[i], 1), 1.0)) for i in range(p)]
    alns = [{i: np.arange(off[i + 1] - off[i])} for i in range(p)]
    for n1, n2 in tree_joins(g[f"fam{tag}_tree"]):
        tot = len(alns[n1]) + len(alns[n2])            # multiple_alignment.py:199-202
        a1, a2, tn, wn = oracle.progressive_node_flexible(*nodes[n1], *nodes[n2], len(alns[n2]) / (2 * tot), len(alns[n1]) / (2 * tot),
                                                          gamma_tensor=7.0, gamma_weight=1.0, gap_open=1.0, gap_extend=0.01)
        merged = {k: np.array([v[i] if i != -1 else -1 for i in a1]) for k, v in alns[n1].items()}
        merged.update({k: np.array([v[i] if i != -1 else -1 for i in a2]) for k, v in alns[n2].items()})
        nodes.append((tn, wn))
        alns.append(merged)
    assert np.array_equal(np.array([alns[-1][i] for i in range(p)]), g[f"fam{tag}_msa"])
    for k in range(int(g[f"fam{tag}_nnodes"])):
        tn, wn = nodes[p + k]
        assert np.array_equal(tn, g[f"fam{tag}_n{k}_tensors"]) and np.array_equal(wn, g[f"fam{tag}_n{k}_weights"])


def flexible_reference(oracle, g, tag, gamma_tensor=7.0):
    """The oracle's restatement of the flexible=True matrix (multiple_alignment.py:323-326, :158-170) on a stored family."""
    offs = g[f"fam{tag}_offsets"]
    tens = g[f"fam{tag}_tensors"]
    num = len(offs) - 1
    m = np.zeros((num, num))
    for i in range(num - 1):
        for j in range(i + 1, num):
            s = oracle.make_score_matrix(tens[offs[i]:offs[i + 1]], tens[offs[j]:offs[j + 1]], gamma_tensor)
            m[i, j] = m[j, i] = oracle.smith_waterman_score(np.arange(s.shape[0]), np.arange(s.shape[1]), s, 0.0)
    return m


def test_flexible_matrix_golden(oracle, golden):
    """f9_flexible.npz: the reference's own make_pairwise_matrix(flexible=True) and smith_waterman on a flexible score matrix."""
    g = golden("f9_flexible.npz")
    for tag in ("FA", "FB"):
        np.testing.assert_allclose(flexible_reference(oracle, g, tag), g[f"fam{tag}_M"], rtol=1e-9, atol=1e-12)
        offs, tens = g[f"fam{tag}_offsets"], g[f"fam{tag}_tensors"]
        s = oracle.make_score_matrix(tens[offs[0]:offs[1]], tens[offs[1]:offs[2]], 7.0)
        a1, a2, score, _none = oracle.smith_waterman(np.arange(s.shape[0]), np.arange(s.shape[1]), s, 0.0)
        assert np.array_equal(a1, g[f"fam{tag}_sw_aln1"]) and np.array_equal(a2, g[f"fam{tag}_sw_aln2"])
        assert abs(score - float(g[f"fam{tag}_sw_score"])) <= 1e-9 * abs(score)
