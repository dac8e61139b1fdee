"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI
(``caretta_amd`` -> ctypes -> libcaretta_hip.so), against
  * the golden vectors produced by the reference's own source (integers exact, floats to tolerance),
  * the C oracle on fresh seeded inputs (bit-identical: same FP64 operation order on both sides),
  * size-independent properties at BASELINE.json's full sizes.
"""
import os

import numpy as np
import pytest

from caretta_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from caretta_amd import engine
    c = engine.Context(0)
    yield c
    c.close()


def run_batch(ctx, coords, tensors, offsets, pairs, **params):
    from caretta_amd import engine
    b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    b.run(engine.make_params(**params) if params else None)
    res, aln = b.fetch()
    b.close()
    return res, aln


def assert_bit_identical(res, aln, ref, ref_aln):
    assert np.array_equal(res["flags"], ref["flags"])
    assert np.array_equal(res["aln_len"], ref["aln_len"])
    assert np.array_equal(res["seed_len"], ref["seed_len"])
    for p in range(len(res)):
        ln = int(ref["aln_len"][p])
        assert np.array_equal(aln[p, :, :ln], ref_aln[p, :, :ln]), f"pair {p}: alignment differs"
        assert np.all(aln[p, :, ln:] == -2)
    for key in ("sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm", "R", "t"):
        assert np.array_equal(res[key], ref[key]), f"{key}: not bit-identical to the oracle"


# ------------------------------------------------------------------------------- device math
def test_exp_bit_identical_to_oracle(oracle):
    from caretta_amd import score_functions as sf
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 27.4, 4000), 10 ** rng.uniform(-12, 0, 1000), [0.0, 27.29, 27.3, 26.6, 26.61]])
    a = x.reshape(-1, 1)
    b = np.zeros((3, 1))
    for gamma in (1.0, 0.03, 7.0, -0.9):          # exp(-gamma * x^2): covers [-746, 0] and positive arguments
        s = sf.make_score_matrix(a, b, sf.get_gaussian_score, gamma)
        assert np.array_equal(s, oracle.make_score_matrix(a, b, gamma))
    assert sf.get_gaussian_score(np.zeros(3), np.zeros(3)) == 1.0
    with pytest.raises(ValueError):
        sf.make_score_matrix(np.array([[np.nan, 0.0, 0.0]]), np.zeros((2, 3)), sf.get_gaussian_score, 0.03)


def test_make_score_matrix_golden(oracle, golden):
    from caretta_amd import score_functions as sf
    g = golden("f1_score_matrix.npz")
    for c in range(int(g["ncases"])):
        a, b, gamma = g[f"c{c}_a"], g[f"c{c}_b"], float(g[f"c{c}_gamma"])
        s = sf.make_score_matrix(a, b, sf.get_gaussian_score, gamma)
        assert np.array_equal(s, oracle.make_score_matrix(a, b, gamma)), c
        np.testing.assert_allclose(s, g[f"c{c}_S"], rtol=1e-12, atol=2e-323)


# ------------------------------------------------------------------------------- DP drop-ins
def test_make_score_matrix_normalized_golden(golden):
    """The normalized=True branch (score_functions.py:43-47; no caller in the reference) against the reference's output."""
    from caretta_amd import score_functions as sf
    g = golden("f6_extras.npz")
    for c in range(int(g["nns"])):
        s = sf.make_score_matrix(g[f"ns{c}_a"], g[f"ns{c}_b"], sf.get_gaussian_score, float(g[f"ns{c}_gamma"]), normalized=True)
        np.testing.assert_allclose(s, g[f"ns{c}_S"], rtol=1e-11, atol=1e-300)


def test_dtw_align_golden(golden):
    from caretta_amd import dynamic_time_warping as dtw
    g = golden("f1_dtw.npz")
    for c in range(int(g["ncases"])):
        s = g[f"c{c}_S"]
        if f"c{c}_seq1" in g:
            s1, s2 = g[f"c{c}_seq1"], g[f"c{c}_seq2"]
        else:
            s1, s2 = np.arange(s.shape[0]), np.arange(s.shape[1])
        go, ge = float(g[f"c{c}_open"]), float(g[f"c{c}_extend"])
        a1, a2, score = dtw.dtw_align(s1, s2, s, go, ge)
        assert np.array_equal(a1, g[f"c{c}_aln1"]) and np.array_equal(a2, g[f"c{c}_aln2"]), c
        assert score == float(g[f"c{c}_score"]), c
        assert dtw.dtw_align_score(s1, s2, s, go, ge) == float(g[f"c{c}_score2"])


def test_smith_waterman_golden(golden):
    from caretta_amd import dynamic_time_warping as dtw
    g = golden("f1_sw.npz")
    for c in range(int(g["ncases"])):
        s = g[f"c{c}_S"]
        s1, s2 = np.arange(s.shape[0]), np.arange(s.shape[1])
        gap = float(g[f"c{c}_gap"])
        a1, a2, score = dtw.smith_waterman(s1, s2, s, gap)
        assert np.array_equal(a1, g[f"c{c}_aln1"]) and np.array_equal(a2, g[f"c{c}_aln2"]), c
        assert score == float(g[f"c{c}_score"])
        assert dtw.smith_waterman_score(s1, s2, s, gap) == float(g[f"c{c}_score_only"])


def test_smith_waterman_edge_cases():
    from caretta_amd import dynamic_time_warping as dtw
    with pytest.raises(TypeError):
        dtw.smith_waterman(np.arange(4), np.arange(5), np.zeros((4, 5)))
    assert dtw.smith_waterman_score(np.arange(4), np.arange(5), np.zeros((4, 5))) == 0.0
    # a row stops at the first -1 of seq2 (dynamic_time_warping.py:214-215)
    s = np.ones((3, 6))
    seq2 = np.array([0, 1, 2, -1, 4, 5])
    assert dtw.smith_waterman_score(np.arange(3), seq2, s) == 3.0
    with pytest.raises(ValueError):
        dtw.dtw_align(np.array([0, 7]), np.arange(3), np.ones((2, 3)))


def _explicit_problems(rng):
    """Ragged list of (seq1, seq2, S): identity sequences, alphabet mode, negative scores, a -1 in seq2, columns
    beyond one strip of every width, a 1 x 1 matrix."""
    problems = []
    for n, m in [(1, 1), (5, 64), (64, 65), (37, 128), (150, 150), (300, 300), (90, 321), (33, 700), (257, 513), (12, 1100)]:
        problems.append((np.arange(n), np.arange(m), rng.uniform(size=(n, m)) ** 3 - 0.2))
    sub = rng.normal(size=(20, 20))                      # alphabet mode: a substitution matrix indexed by residue type
    for n, m in [(40, 55), (200, 333), (7, 600)]:
        problems.append((rng.integers(0, 20, size=n), rng.integers(0, 20, size=m), sub))
    big = rng.uniform(size=(50, 400))                    # a window of a larger matrix: contiguous columns from 17 on
    problems.append((np.arange(10, 40), np.arange(17, 317), big))
    problems.append((np.arange(50)[::-1].copy(), np.arange(400)[::2].copy(), big))     # strided columns: gathers
    return problems


def test_smith_waterman_score_batch_vs_oracle(oracle):
    """Many matrices per launch: the row sweep (gap 0) and the skewed sweep (gap != 0) against the oracle, cell-exact."""
    from caretta_amd import dynamic_time_warping as dtw
    rng = np.random.default_rng(11)
    problems = _explicit_problems(rng)
    batch = dtw.ExplicitBatch(problems)
    for gap in (0.0, 0.3):
        got = batch.smith_waterman_scores(gap)
        want = np.array([oracle.smith_waterman_score(a, b, s, gap) for a, b, s in problems])
        assert np.array_equal(got, want), (gap, np.nonzero(got != want)[0])
    batch.close()
    # a row stops at the first -1 of seq2 (dynamic_time_warping.py:214-215), also in a batch; an all-(-1) row scores 0
    s = np.ones((3, 6))
    cut = [(np.arange(3), np.array([0, 1, 2, -1, 4, 5]), s), (np.arange(3), np.arange(6), s),
           (np.arange(3), np.array([-1, 1, 2, 3, 4, 5]), s)]
    assert np.array_equal(dtw.smith_waterman_score_batch(cut), [3.0, 3.0, 0.0])
    assert np.array_equal(dtw.smith_waterman_score_batch(cut, 0.25), [oracle.smith_waterman_score(*c, 0.25) for c in cut[:2]] + [0.0])
    # equal to the single-call drop-in
    one = problems[5]
    assert dtw.smith_waterman_score_batch([one])[0] == dtw.smith_waterman_score(*one)


def test_dtw_align_batch_vs_oracle(oracle):
    from caretta_amd import dynamic_time_warping as dtw
    rng = np.random.default_rng(12)
    problems = _explicit_problems(rng)
    got = dtw.dtw_align_batch(problems, 1.0, 0.01)
    for (a, b, s), (a1, a2, score) in zip(problems, got):
        o1, o2, osc = oracle.dtw_align(a, b, s, 1.0, 0.01)
        assert np.array_equal(a1, o1) and np.array_equal(a2, o2) and score == osc, (len(a), len(b))
    batch = dtw.ExplicitBatch(problems[:6])
    assert np.array_equal(batch.dtw_align(0.5, 0.1, want_alignments=False),
                          [oracle.dtw_align_score(a, b, s, 0.5, 0.1) for a, b, s in problems[:6]])
    batch.close()
    with pytest.raises(ValueError):
        dtw.dtw_align_batch([(np.arange(3), np.array([0, -1, 2]), np.ones((3, 3)))])


def test_explicit_batch_streaming_kernels_vs_oracle(oracle):
    """Lists in which EVERY problem has contiguous columns take the streaming sweep (per-lane rings of aligned 64-byte
    blocks): ragged shapes, several strips of every rows-per-lane choice, windows of larger matrices (row starts at every
    alignment), negative scores -- smith_waterman_score with gap != 0 and dtw_align with alignments, cell-exact."""
    from caretta_amd import dynamic_time_warping as dtw
    rng = np.random.default_rng(13)
    problems = []
    for n, m in [(1, 1), (2, 9), (63, 64), (64, 63), (129, 300), (300, 300), (257, 77), (500, 513), (7, 1100), (333, 8)]:
        problems.append((np.arange(n), np.arange(m), rng.uniform(size=(n, m)) ** 3 - 0.15))
    for shift in range(9):                                # windows: every alignment of the row starts
        big = rng.normal(size=(70, 101 + shift))
        problems.append((np.arange(5, 65), np.arange(shift, 90 + shift), big))
    batch = dtw.ExplicitBatch(problems)
    for gap in (0.3, 0.0):
        got = batch.smith_waterman_scores(gap)
        want = np.array([oracle.smith_waterman_score(a, b, s, gap) for a, b, s in problems])
        assert np.array_equal(got, want), (gap, np.nonzero(got != want)[0])
    for go, ge in ((1.0, 0.01), (0.0, 0.0), (0.5, 0.5)):
        for (a, b, s), (a1, a2, score) in zip(problems, batch.dtw_align(go, ge)):
            o1, o2, osc = oracle.dtw_align(a, b, s, go, ge)
            assert np.array_equal(a1, o1) and np.array_equal(a2, o2) and score == osc, (len(a), len(b), go, ge)
    batch.close()


def test_fetch_variants_agree(ctx):
    """cr_batch_fetch (int64 rows) and cr_batch_fetch_i32 into page-locked arrays: same records, same rows, also when the
    launch order differs from the caller's pair order (ragged batch, both orientations)."""
    from caretta_amd import engine
    fam = synthetic.make_family(9, 260, seed=515, ragged=True, clades=2)
    for k, s in enumerate(fam):
        cut = [260, 30, 100, 200, 64, 150, 90, 256, 5][k]
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.vstack([engine.all_pairs(9), engine.all_pairs(9)[:, ::-1]])
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    batch.run(engine.make_params())
    res, aln = batch.fetch()
    res32, aln32 = batch.fetch(pinned=True)
    assert aln32.dtype == np.int32 and res.tobytes() == res32.tobytes() and np.array_equal(aln, aln32)
    res_only, none = batch.fetch(want_alignments=False, pinned=True)
    assert none is None and res_only.tobytes() == res.tobytes()
    again, _ = batch.fetch(pinned=True)                   # the page-locked arrays are reused
    assert again is res32 or np.shares_memory(again, res32)
    batch.close()
    assert res32.tobytes() == res.tobytes()               # ... and outlive the batch


def test_transfers_through_the_ring_match_direct_ones(ctx):
    """Pageable caller memory goes through the context's page-locked ring in 8 MiB slots (upload_async / download), page-
    locked memory goes straight: same bytes either way, for sizes below one slot, across a slot boundary and not a
    multiple of the slot."""
    from caretta_amd import engine, score_functions as sf
    rng = np.random.default_rng(77)
    # download: a 1100 x 1100 score matrix (9.7 MB: two slots, the second partial) into pageable and into page-locked memory
    a, b = rng.normal(size=(1100, 3)), rng.normal(size=(1100, 3))
    want = np.exp(-0.03 * ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))
    got = sf.make_score_matrix(a, b, sf.get_gaussian_score, 0.03)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    # upload: 64 structures x 2000 residues x (3 + 10) doubles = 3 + 10 MB of pageable arrays (the tensors cross a slot
    # boundary) against the same data in page-locked arrays; fetch into pageable and into page-locked arrays
    fam = synthetic.make_family(64, 2000, seed=99, clades=3)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(64)[::90]
    pinned_coords, pinned_tensors = engine.pinned_empty(coords.shape, np.float64), engine.pinned_empty(tensors.shape, np.float64)
    pinned_coords[...] = coords
    pinned_tensors[...] = tensors
    out = []
    for c, t in ((coords, tensors), (pinned_coords, pinned_tensors)):
        batch = engine.PairBatch(ctx, c, t, offsets).set_pairs(pairs)
        batch.run(engine.make_params())
        out.append((batch.fetch(), batch.fetch(pinned=True)))
        batch.close()
    (res_a, aln_a), (res_a32, aln_a32) = out[0]
    (res_b, aln_b), (res_b32, aln_b32) = out[1]
    assert res_a.tobytes() == res_b.tobytes() == res_a32.tobytes() == res_b32.tobytes()
    assert np.array_equal(aln_a, aln_b) and np.array_equal(aln_a, aln_a32) and np.array_equal(aln_a32, aln_b32)
    assert tensors.nbytes > 8 << 20 and got.nbytes > 8 << 20


def test_plugin_pairwise_matrix_runs_batched(oracle):
    """A third-party SequenceBase plugin (multiple_alignment.py:109-127): make_pairwise_matrix = the reference's loop of
    smith_waterman_score over the plugin's own score matrices, computed many matrices per launch."""
    from caretta_amd import multiple_alignment as ma

    class Letters(ma.SequenceBase):
        def __init__(self, name, codes):
            self.name, self.codes = name, np.asarray(codes)

        def score_function(self, other, match=1.0, mismatch=-0.5):
            return np.where(self.codes[:, None] == other.codes[None, :], match, mismatch)

        def __len__(self):
            return len(self.codes)

        def __str__(self):
            return "".join(chr(65 + c) for c in self.codes)

    rng = np.random.default_rng(3)
    seqs = [Letters(f"s{k}", rng.integers(0, 4, size=int(rng.integers(30, 400)))) for k in range(9)]
    got = ma.MultipleAlignment(seqs).make_pairwise_matrix({"match": 2.0, "mismatch": -1.0})
    want = np.zeros((9, 9))
    for i in range(8):
        for j in range(i + 1, 9):
            want[i, j] = want[j, i] = oracle.smith_waterman_score(
                np.arange(len(seqs[i])), np.arange(len(seqs[j])), seqs[i].score_function(seqs[j], 2.0, -1.0), 0.0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("staged", ["1", "0"])
def test_dp_multistrip_vs_oracle(oracle, monkeypatch, staged):
    """n > 64 rows: several strips.  Up to 1024 rows the single-call drop-ins gather the matrix into the staged sweep's
    step order and run one wave per strip (one or two rows per lane; cr_staged.h), beyond that -- and with
    CARETTA_STAGED=0 -- one wave takes the strips in turn and hands the last row over through LDS."""
    from caretta_amd import dynamic_time_warping as dtw
    monkeypatch.setenv("CARETTA_STAGED", staged)
    rng = np.random.default_rng(5)
    for n, m in [(321, 50), (700, 333), (1000, 64), (64, 1000), (1024, 7), (513, 200), (1025, 90), (130, 3), (1537, 100), (2048, 33),
                 (2049, 20)]:
        s = rng.uniform(size=(n, m)) ** 3
        a, b = np.arange(n), np.arange(m)
        r1 = dtw.dtw_align(a, b, s, 1.0, 0.01)
        o1 = oracle.dtw_align(a, b, s, 1.0, 0.01)
        assert np.array_equal(r1[0], o1[0]) and np.array_equal(r1[1], o1[1]) and r1[2] == o1[2]
        r2 = dtw.smith_waterman(a, b, s - 0.3, 0.1)
        o2 = oracle.smith_waterman(a, b, s - 0.3, 0.1)
        assert np.array_equal(r2[0], o2[0]) and np.array_equal(r2[1], o2[1]) and r2[2] == o2[2]
        assert dtw.smith_waterman_score(a, b, s - 0.3, 0.1) == oracle.smith_waterman_score(a, b, s - 0.3, 0.1)
        assert dtw.dtw_align_score(a, b, s, 0.5, 0.5) == oracle.dtw_align_score(a, b, s, 0.5, 0.5)
    # alphabet mode: index sequences into a small substitution matrix (gathered by the staging launch)
    sub = rng.normal(size=(21, 23))
    for n, m in [(400, 380), (90, 700)]:
        a, b = rng.integers(0, 21, size=n), rng.integers(0, 23, size=m)
        r1, o1 = dtw.dtw_align(a, b, sub, 1.0, 0.1), oracle.dtw_align(a, b, sub, 1.0, 0.1)
        assert np.array_equal(r1[0], o1[0]) and np.array_equal(r1[1], o1[1]) and r1[2] == o1[2]
        r2, o2 = dtw.smith_waterman(a, b, sub, 0.5), oracle.smith_waterman(a, b, sub, 0.5)
        assert np.array_equal(r2[0], o2[0]) and np.array_equal(r2[1], o2[1]) and r2[2] == o2[2]
        assert dtw.smith_waterman_score(a, b, sub, 0.0) == oracle.smith_waterman_score(a, b, sub, 0.0)


# ------------------------------------------------------------------------------- Kabsch & metrics
def test_kabsch_rank_deficient(oracle):
    """Collinear / coincident / planar positions: the device SVD completes U to a rotation exactly as the oracle does."""
    from caretta_amd import superposition_functions as sup
    from test_oracle_golden import degenerate_kabsch_cases
    for tag, x1, x2 in degenerate_kabsch_cases():
        r, t = sup.paired_svd_superpose(x1, x2)
        ro, to = oracle.paired_svd_superpose(x1, x2)
        assert np.array_equal(r, ro) and np.array_equal(t, to), tag
        np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-12, err_msg=tag)


def test_small_dropins_host_and_kernel_paths_agree(oracle, monkeypatch):
    """paired_svd_superpose / get_rmsd / tm_score run the same CR_HD code on the host for small inputs and in a kernel
    for large ones: both paths, and the oracle, bit for bit."""
    from caretta_amd import multiple_alignment as ma
    from caretta_amd import score_functions as sf
    from caretta_amd import superposition_functions as sup
    rng = np.random.default_rng(5)
    for k in (3, 17, 300):
        x1 = rng.normal(size=(k, 3)) * 10
        x2 = rng.normal(size=(k, 3)) * 10
        got = {}
        for limit in ("4096", "0"):
            monkeypatch.setenv("CARETTA_HOST_SMALL_K", limit)
            r, t = sup.paired_svd_superpose(x1, x2)
            got[limit] = (r, t, sf.get_rmsd(x1, x2), ma.tm_score(x1, x2, k + 20, k + 40))
        for a, b in zip(got["4096"], got["0"]):
            assert np.array_equal(a, b)
        ro, to = oracle.paired_svd_superpose(x1, x2)
        assert np.array_equal(got["0"][0], ro) and np.array_equal(got["0"][1], to)
        assert got["0"][2] == oracle.get_rmsd(x1, x2) and got["0"][3] == oracle.tm_score(x1, x2, k + 20, k + 40)


def test_kabsch_golden(oracle, golden):
    from caretta_amd import score_functions as sf
    from caretta_amd import superposition_functions as sup
    g = golden("f1_kabsch.npz")
    for c in range(int(g["ncases"])):
        x1, x2 = g[f"c{c}_x1"], g[f"c{c}_x2"]
        r, t = sup.paired_svd_superpose(x1, x2)
        ro, to = oracle.paired_svd_superpose(x1, x2)
        assert np.array_equal(r, ro) and np.array_equal(t, to), "SVD/div/sqrt differ between GPU and CPU"
        np.testing.assert_allclose(r, g[f"c{c}_R"], atol=1e-9)
        np.testing.assert_allclose(t, g[f"c{c}_t"], atol=1e-8)
        moved = sup.apply_rotran(x2, r, t)
        assert np.array_equal(moved, oracle.apply_rotran(x2, r, t))
        assert abs(sf.get_rmsd(x1, moved) - float(g[f"c{c}_rmsd"])) < 1e-5
        assert sf.get_rmsd(x1, moved) == oracle.get_rmsd(x1, moved)
    for s in range(int(g["nsubset"])):
        a, b = g[f"s{s}_a"], g[f"s{s}_b"]
        o = sup.paired_svd_superpose_with_subset(a, b, a[g[f"s{s}_p1"]], b[g[f"s{s}_p2"]])
        ref = oracle.paired_svd_superpose_with_subset(a, b, a[g[f"s{s}_p1"]], b[g[f"s{s}_p2"]])
        for x, y, key in zip(o, ref, ("o1", "o2", "o3")):
            assert np.array_equal(x, y)
            np.testing.assert_allclose(x, g[f"s{s}_{key}"], atol=1e-8)


def test_misc_golden(oracle, golden):
    from caretta_amd import helper, multiple_alignment as ma, score_functions as sf
    g = golden("f1_misc.npz")
    for c in range(int(g["ncp"])):
        p1, p2 = helper.get_common_positions(g[f"cp{c}_a1"], g[f"cp{c}_a2"])
        assert np.array_equal(p1, g[f"cp{c}_p1"]) and np.array_equal(p2, g[f"cp{c}_p2"])
    for c in range(int(g["ntm"])):
        x1, x2, l1, l2 = g[f"tm{c}_x1"], g[f"tm{c}_x2"], int(g[f"tm{c}_l1"]), int(g[f"tm{c}_l2"])
        assert abs(ma.tm_score(x1, x2, l1, l2) - float(g[f"tm{c}_tm"])) < 1e-5
        assert ma.tm_score(x1, x2, l1, l2) == oracle.tm_score(x1, x2, l1, l2)
        assert abs(sf.get_rmsd(x1, x2) - float(g[f"tm{c}_rmsd"])) < 1e-5


# ------------------------------------------------------------------------------- pipeline H
def check_against_golden(g, fam, res, aln):
    pairs = g[f"fam{fam}_pairs"]
    for p in range(len(pairs)):
        key = f"fam{fam}_p{p}"
        assert int(res["flags"][p]) == int(g[key + "_flags"]), key
        assert int(res["seed_len"][p]) == len(g[key + "_seed_aln1"]), key
        ln = int(res["aln_len"][p])
        assert np.array_equal(aln[p, 0, :ln], g[key + "_aln1"]), key       # bit-exact DTW traceback indices
        assert np.array_equal(aln[p, 1, :ln], g[key + "_aln2"]), key
        assert abs(res["sw"][p] - float(g[key + "_sw"])) <= 1e-9 * max(1.0, abs(res["sw"][p]))
        assert abs(res["dtw_score"][p] - float(g[key + "_dtw_score"])) <= 1e-9 * max(1.0, abs(res["dtw_score"][p]))
        if not int(res["flags"][p]) & 2:
            assert abs(res["rmsd"][p] - float(g[key + "_rmsd"])) < 1e-5     # north-star tolerance
            assert abs(res["tm"][p] - float(g[key + "_tm"])) < 1e-5
            assert abs(res["coverage"][p] - float(g[key + "_coverage"])) < 1e-12
            np.testing.assert_allclose(res["R"][p].reshape(3, 3), g[key + "_R"], atol=1e-8)


@pytest.mark.parametrize("fam", ["A", "B", "C", "D", "E"])
def test_pipeline_golden(ctx, oracle, golden, fam):
    g = golden("f2_pipeline.npz")
    coords, tensors, offsets = g[f"fam{fam}_coords"], g[f"fam{fam}_tensors"], g[f"fam{fam}_offsets"]
    pairs = g[f"fam{fam}_pairs"]
    res, aln = run_batch(ctx, coords, tensors, offsets, pairs)
    check_against_golden(g, fam, res, aln)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs)
    assert_bit_identical(res, aln, ref, ref_aln)


def test_pipeline_golden_long(ctx, oracle, golden):
    g = golden("f2_pipeline_long.npz")
    coords, tensors, offsets, pairs = g["famL_coords"], g["famL_tensors"], g["famL_offsets"], g["famL_pairs"]
    res, aln = run_batch(ctx, coords, tensors, offsets, pairs)
    check_against_golden(g, "L", res, aln)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs)
    assert_bit_identical(res, aln, ref, ref_aln)


@pytest.mark.parametrize("num,length,dim,ragged,seed", [
    (32, 150, 10, False, 20241),      # BASELINE config 2, all 496 pairs
    (12, 200, 10, True, 31),          # ragged, 160..200 rows: R=3/R=5 boundary
    (6, 330, 10, True, 32),           # spills into a second strip for some pairs
    (5, 90, 3, True, 33), (5, 90, 4, False, 34), (5, 90, 7, True, 35), (5, 90, 16, False, 36), (4, 70, 1, True, 37),
])
def test_pipeline_vs_oracle(ctx, oracle, num, length, dim, ragged, seed):
    from caretta_amd import engine
    fam = synthetic.make_family(num, length, dim=dim, seed=seed, ragged=ragged)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    res, aln = run_batch(ctx, coords, tensors, offsets, pairs)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, nthreads=8)
    assert_bit_identical(res, aln, ref, ref_aln)


def test_pipeline_other_parameters_and_orientation(ctx, oracle):
    from caretta_amd import engine
    fam = synthetic.make_family(6, 80, seed=77, ragged=True)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.array([(i, j) for i in range(6) for j in range(6) if i != j], dtype=np.int32)  # both orientations
    prm = dict(gamma_tensor=3.0, gamma_coords=0.05, gap_open=0.5, gap_extend=0.5, sw_gap=0.2)
    res, aln = run_batch(ctx, coords, tensors, offsets, pairs, **prm)
    from oracle.pyoracle import default_params
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, default_params(**prm))
    assert_bit_identical(res, aln, ref, ref_aln)


def test_chunked_scratch_is_invisible(ctx, monkeypatch):
    """A pair list longer than the decision-scratch budget runs chunk after chunk with identical results."""
    from caretta_amd import engine
    fam = synthetic.make_family(10, 120, seed=91, ragged=True)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(10)
    res1, aln1 = run_batch(ctx, coords, tensors, offsets, pairs)
    monkeypatch.setenv("CARETTA_SCRATCH_MB", "1")            # ~20 pairs of 120x120 per chunk
    b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    ctx.set_profiling(2)
    b.run()
    b.run()
    res2, aln2 = b.fetch()
    ms, runs = b.stage_ms()
    ctx.set_profiling(0)
    b.close()
    assert runs == 2 and np.all(ms > 0)
    assert res1.tobytes() == res2.tobytes() and np.array_equal(aln1, aln2)


def test_pipeline_tiny_and_degenerate(ctx, oracle):
    """1..5-residue structures, identical structures, and a tensor score matrix that underflows to all
    zeros (the reference raises there; the batch reports CR_FLAG_SEED_ALL_ZERO and carries on unsuperposed)."""
    from caretta_amd import engine
    rng = np.random.default_rng(123)
    lens = [1, 2, 3, 4, 5, 9, 40, 40]
    structs = []
    for k, ln in enumerate(lens):
        structs.append(synthetic.Structure(f"t{k}", rng.uniform(size=(ln, 10)), synthetic._walk(rng, ln), "A" * ln))
    structs[7] = synthetic.Structure("copy", structs[6].tensors.copy(), structs[6].coordinates.copy(), "A" * 40)
    far = synthetic.Structure("far", structs[6].tensors + 40.0, structs[6].coordinates + 5.0, "A" * 40)  # exp(-7*16000) = 0
    structs.append(far)
    coords, tensors, offsets = synthetic.pack(structs)
    num = len(structs)
    pairs = np.array([(i, j) for i in range(num) for j in range(num) if i != j], dtype=np.int32)
    res, aln = run_batch(ctx, coords, tensors, offsets, pairs)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs)
    assert_bit_identical(res, aln, ref, ref_aln)
    flags = {tuple(p): int(f) for p, f in zip(pairs, res["flags"])}
    assert flags[(6, 8)] & 4 and flags[(6, 8)] & 1          # all-zero seed -> no superposition
    assert flags[(0, 1)] & 2                                 # fewer than 3 aligned positions -> no metrics
    p67 = int(np.nonzero((pairs[:, 0] == 6) & (pairs[:, 1] == 7))[0][0])
    assert res["rmsd"][p67] < 1e-12 and res["aln_len"][p67] == 40 and res["coverage"][p67] == 1.0


@pytest.mark.parametrize("num,length,seed,stride", [(128, 300, 20242, 50), (512, 300, 20243, 1400), (64, 1200, 20244, 130)],
                         ids=["config3_128x300", "config4_512x300", "config5_64x1200"])
def test_full_size_configs_sample_and_properties(ctx, oracle, num, length, seed, stride):
    """BASELINE configs 3, 4 and 5 at full size on the GPU (all pairs): a deterministic sample re-done by the oracle
    (bit-identical), and size-independent properties on every pair."""
    from caretta_amd import engine
    fam = synthetic.make_family(num, length, seed=seed)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    batch.run(engine.make_params())
    res, _ = batch.fetch(want_alignments=False)
    sw_only, flags_only = batch.fetch_scores()
    assert np.array_equal(sw_only, res["sw"]) and np.array_equal(flags_only, res["flags"])
    batch.close()
    n = length
    ln = res["aln_len"]
    assert np.all((ln >= n) & (ln <= 2 * n)) and np.all(res["flags"] == 0)
    assert np.all(res["sw"] > 0) and np.all(np.isfinite(res["dtw_score"])) and np.all(res["rmsd"] >= 0)
    assert np.all((res["coverage"] > 0) & (res["coverage"] <= 1)) and np.all((res["tm"] > 0) & (res["tm"] <= 1))
    matched = np.rint(res["coverage"] * ln).astype(np.int64)
    assert np.all(ln == 2 * n - matched)                       # every residue of both structures exactly once
    np.testing.assert_allclose(np.linalg.det(res["R"].reshape(-1, 3, 3)), 1.0, atol=1e-12)
    # the P x P matrix of multiple_alignment.py:158-170: symmetric, zero diagonal
    m = engine.assemble_matrix(pairs, res["sw"], num)
    assert np.array_equal(m, m.T) and np.all(np.diag(m) == 0)
    # alignments and the oracle on a deterministic sample of pairs
    sample = np.arange(0, len(pairs), stride)
    sub = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs[sample])
    sub.run(engine.make_params())
    sres, saln = sub.fetch(want_alignments=True)
    sub.close()
    assert sres.tobytes() == res[sample].tobytes()             # a pair's result does not depend on its batch
    for row in (0, 1):
        a = saln[:, row, :]
        present = np.sort(np.where(a >= 0, a, 10 ** 6), axis=1)[:, :n]
        assert np.array_equal(present, np.tile(np.arange(n), (len(sample), 1)))
        for p in range(len(sample)):
            x = a[p][a[p] >= 0]
            assert np.all(np.diff(x) == 1)
    both = ((saln[:, 0, :] >= 0) & (saln[:, 1, :] >= 0)).sum(axis=1)
    np.testing.assert_allclose(sres["coverage"], both / sres["aln_len"], rtol=0, atol=0)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs[sample], nthreads=8)
    assert_bit_identical(sres, saln, ref, ref_aln)


def test_guide_tree_from_gpu_matrix(ctx, oracle, golden):
    """P x P matrix -> max - M -> neighbor joining: identical topology to the golden tree and to the oracle."""
    from caretta_amd import engine, multiple_alignment as ma, neighbor_joining as nj
    g = golden("f3_tree.npz")
    for fam in ("T8", "T16"):
        coords, tensors, offsets = g[f"fam{fam}_coords"], g[f"fam{fam}_tensors"], g[f"fam{fam}_offsets"]
        p = len(offsets) - 1
        prots = [ma.Protein(f"s{i}", tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]], "")
                 for i in range(p)]
        msa = ma.MultipleAlignment(prots)
        m = msa.make_pairwise_matrix(dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03))
        np.testing.assert_allclose(m, g[f"fam{fam}_M"], rtol=1e-9)
        d = m.max() - m
        tree, bl = nj.neighbor_joining(d)
        assert nj.bipartitions(tree, p) == nj.bipartitions(g[f"fam{fam}_tree"], p)
        otree, obl = oracle.neighbor_joining(d)
        assert np.array_equal(tree, otree) and np.array_equal(bl, obl)


@pytest.mark.parametrize("dim", [17, 20, 24, 29, 32])
def test_tensor_widths_up_to_32(ctx, oracle, dim):
    """The tensor width is geometricus' output_dimension (multiple_alignment.py:479-488), a run-time parameter: widths
    above 16 (padded to 24 / 32 in registers) through the single-wave kernels of every rows-per-lane group, the team
    kernels (few long pairs), both SW gap settings, and a progressive alignment -- bit-identical to the oracle."""
    from caretta_amd import engine, multiple_alignment as ma, neighbor_joining as nj
    fam = synthetic.make_family(10, 330, dim=dim, seed=400 + dim, ragged=True, clades=2)
    for k, s in enumerate(fam):                         # lengths across the R = 2 .. 5 groups and two strips
        cut = [330, 60, 128, 200, 256, 300, 330, 90, 150, 40][k]
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(len(fam))
    for gap in (0.0, 0.05):
        prm = engine.make_params(sw_gap=gap)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(prm)
        res, aln = batch.fetch()
        batch.close()
        from oracle.pyoracle import default_params
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=8)
        assert_bit_identical(res, aln, ref, ref_aln)
    long_pairs = np.array([[0, 6], [5, 6]], dtype=np.int32)          # two pairs of ~300 rows: the team kernels
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(long_pairs)
    batch.run(engine.make_params())
    res, aln = batch.fetch()
    batch.close()
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, long_pairs, nthreads=2)
    assert_bit_identical(res, aln, ref, ref_aln)
    # progressive alignment on the wide tensors: finishes, and its MSA rows index every residue once
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam[:6]]
    msa = ma.MultipleAlignment(prots)
    m = msa.make_pairwise_matrix(dict(gamma_tensor=7.0, gamma_coords=0.03))
    tree, _ = nj.neighbor_joining(m.max() - m)
    out = msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, dict(gamma_tensor=7.0, gamma_coords=0.03), {})
    for p in prots:
        row = out[p.name]
        assert np.array_equal(row[row >= 0], np.arange(len(p)))


def test_scores_only_run_matches_the_full_pipeline(ctx):
    """cr_batch_run_scores (what make_pairwise_matrix needs: multiple_alignment.py:164) gives the same sw / flags as the
    full pipeline, bit for bit, on every kernel family: one wave per pair (all rows-per-lane groups, several strips),
    the four-wave teams and the wide layouts (few long pairs)."""
    from caretta_amd import engine
    cases = []
    fam = synthetic.make_family(14, 420, seed=808, ragged=True, clades=3)
    for k, s in enumerate(fam):
        cut = [420, 40, 120, 200, 64, 150, 90, 330, 300, 256, 257, 5, 3, 400][k]
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    both = np.vstack([engine.all_pairs(14), engine.all_pairs(14)[:, ::-1]])
    cases.append((fam, both))                                             # 182 pairs: single-wave kernels, grouped
    cases.append((fam, np.array([[0, 7], [8, 13], [13, 8]], dtype=np.int32)))       # three pairs of 300-420 rows: teams
    long = synthetic.make_family(3, 900, seed=809, clades=1)
    cases.append((long, engine.all_pairs(3)))                             # 900 rows: wide kernels, R = 3
    mid = synthetic.make_family(4, 600, seed=810, ragged=True, clades=1)
    cases.append((mid, engine.all_pairs(4)))                              # ~600 rows: wide kernels, R = 2
    for fam, pairs in cases:
        coords, tensors, offsets = synthetic.pack(fam)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(engine.make_params())
        sw_full, flags_full = batch.fetch_scores()
        batch.run(engine.make_params(), scores_only=True)
        sw, flags = batch.fetch_scores()
        assert np.array_equal(sw, sw_full) and np.array_equal(flags, flags_full)
        with pytest.raises(Exception):
            batch.fetch()                                                 # nothing but the scores after a scores-only run
        batch.run(engine.make_params(sw_gap=0.1), scores_only=True)        # gap != 0: the full pipeline answers
        sw_gap, _ = batch.fetch_scores()
        batch.run(engine.make_params(sw_gap=0.1))
        assert np.array_equal(sw_gap, batch.fetch_scores()[0])
        batch.close()


def test_guide_tree_64_from_gpu_matrix(oracle, golden):
    """The 64-structure family of f3_tree64.npz (matrix and tree by the reference's own make_pairwise_matrix and
    neighbor_joining): GPU matrix to 1e-9, identical bipartitions, tree identical to the oracle's."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj
    g = golden("f3_tree64.npz")
    coords, tensors, offsets = g["famT64_coords"], g["famT64_tensors"], g["famT64_offsets"]
    p = len(offsets) - 1
    prots = [ma.Protein(f"s{i}", tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]], "") for i in range(p)]
    m = ma.MultipleAlignment(prots).make_pairwise_matrix(dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03))
    np.testing.assert_allclose(m, g["famT64_M"], rtol=1e-9)
    tree, bl = nj.neighbor_joining(m.max() - m)
    assert nj.bipartitions(tree, p) == nj.bipartitions(g["famT64_tree"], p)
    otree, obl = oracle.neighbor_joining(m.max() - m)
    assert np.array_equal(tree, otree) and np.array_equal(bl, obl)


def _count_integer_differences(res, aln, ref, ref_aln):
    """Pairs whose INTEGER outputs differ (alignment rows, lengths, seed length, flags), and the largest relative
    difference of the float outputs."""
    bad = 0
    for p in range(len(res)):
        ln = int(ref["aln_len"][p])
        same = (int(res["aln_len"][p]) == ln and int(res["seed_len"][p]) == int(ref["seed_len"][p])
                and int(res["flags"][p]) == int(ref["flags"][p]) and np.array_equal(aln[p, :, :ln], ref_aln[p, :, :ln]))
        bad += 0 if same else 1
    rel = 0.0
    for key in ("sw", "dtw_score", "rmsd", "tm", "coverage"):
        denom = np.maximum(np.abs(ref[key]), 1e-300)
        rel = max(rel, float(np.max(np.abs(res[key] - ref[key]) / denom)))
    return bad, rel


def test_libm_exp_path_at_scale(ctx, oracle_libm):
    """The reference's numba path calls libm's exp; the kernels their own (<= 1 ulp apart on < 2 % of arguments,
    test_exp_accuracy).  At BASELINE sizes -- config 2 (all 496 pairs), samples of configs 3, 5 and 4 (999 of its 130 816
    pairs) -- the GPU against the libm-exp oracle: ZERO pairs with a differing traceback / seed index, floats within 1e-9
    (north star: bit-exact indices, RMSD / TM within 1e-5), and the same neighbor-joining bipartitions at P = 32, 128
    and 512 (whole matrices on both sides)."""
    from caretta_amd import engine, neighbor_joining as nj
    threads = max(8, int(oracle_libm.max_threads()))
    for num, length, seed, stride in ((32, 150, 20241, 1), (128, 300, 20242, 41), (64, 1200, 20244, 168), (512, 300, 20243, 131)):
        fam = synthetic.make_family(num, length, seed=seed)
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(num)
        sample = np.arange(0, len(pairs), stride)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs[sample])
        batch.run(engine.make_params())
        res, aln = batch.fetch(want_alignments=True)
        batch.close()
        ref, ref_aln = oracle_libm.pairwise_batch(coords, tensors, offsets, pairs[sample], nthreads=threads)
        bad, rel = _count_integer_differences(res, aln, ref, ref_aln)
        assert bad == 0, f"{num} x {length}: {bad} of {len(sample)} pairs differ in an alignment / seed index"
        assert rel < 1e-9, f"{num} x {length}: floats differ by {rel:.3e}"
        if length <= 300:                                   # tree topology: the whole matrix on both sides
            full = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
            full.run(engine.make_params())
            sw, _ = full.fetch_scores()
            full.close()
            cpu, _ = oracle_libm.pairwise_batch(coords, tensors, offsets, pairs, want_aln=False, nthreads=threads)
            m_gpu, m_cpu = engine.assemble_matrix(pairs, sw, num), engine.assemble_matrix(pairs, cpu["sw"], num)
            np.testing.assert_allclose(m_gpu, m_cpu, rtol=1e-12)
            t_gpu, _ = nj.neighbor_joining(m_gpu.max() - m_gpu)
            t_cpu, _ = oracle_libm.neighbor_joining(m_cpu.max() - m_cpu)
            assert nj.bipartitions(t_gpu, num) == nj.bipartitions(t_cpu, num), f"P = {num}: tree topologies differ"


def test_multiple_align_golden(golden):
    """Progressive alignment on the same kernels (a 'next' row): the 8-structure family's MSA."""
    from caretta_amd import multiple_alignment as ma
    g = golden("f3_tree.npz")
    coords, tensors, offsets = g["famT8_coords"], g["famT8_tensors"], g["famT8_offsets"]
    p = len(offsets) - 1
    prots = [ma.Protein(f"s{i:04d}", tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]], "")
             for i in range(p)]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03)
    m = msa.make_pairwise_matrix(prm)
    aln = msa.multiple_align(m.max() - m, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0,
                             gamma_weight=1.0, score_function_params=dict(prm, verbose=False),
                             mean_function_params=dict(flexible=False, verbose=False))
    got = np.array([aln[q.name] for q in prots], dtype=np.int64)
    assert np.array_equal(got, g["famT8_msa"])


@pytest.mark.parametrize("tag", ["P8", "P5"])
def test_progressive_nodes_golden_and_oracle(oracle, golden, tag):
    """Every intermediate node of progressive_align: alignment identical to the reference's, node tensors /
    coordinates / weights bit-identical to the oracle and within 1e-9 of the reference's."""
    from caretta_amd import multiple_alignment as ma
    g = golden("f4_progressive.npz")
    coords, tensors, offsets = g[f"fam{tag}_coords"], g[f"fam{tag}_tensors"], g[f"fam{tag}_offsets"]
    p = len(offsets) - 1
    prots = [ma.Protein(f"s{i:04d}", tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]], "")
             for i in range(p)]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    aln = msa.multiple_align(g[f"fam{tag}_D"], gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0,
                             gamma_weight=1.0, score_function_params=prm,
                             mean_function_params=dict(flexible=False, verbose=False))
    assert np.array_equal(np.array([aln[q.name] for q in prots]), g[f"fam{tag}_msa"])
    assert np.array_equal(msa.tree, g[f"fam{tag}_tree"])
    nn = int(g[f"fam{tag}_nnodes"])
    assert len(msa.final_sequences) == p + nn
    for k in range(nn):
        node, w = msa.final_sequences[p + k], msa.final_consensus_weights[p + k]
        np.testing.assert_allclose(node.coordinates, g[f"fam{tag}_n{k}_coords"], atol=1e-9)
        assert np.array_equal(node.tensors, g[f"fam{tag}_n{k}_tensors"]) and np.array_equal(w, g[f"fam{tag}_n{k}_weights"])
    # oracle: replay every join on the GPU's own child nodes
    tree = msa.tree.astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]
    sizes = [1] * p
    for k, (n1, n2) in enumerate(joins):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        a1, a2, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, msa.final_consensus_weights[n1],
                                                        s2.coordinates, s2.tensors, msa.final_consensus_weights[n2],
                                                        sizes[n2] / (2 * tot), sizes[n1] / (2 * tot))
        node = msa.final_sequences[p + k]
        assert np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors)
        assert np.array_equal(wn, msa.final_consensus_weights[p + k])
        sizes.append(tot)


def _replay_flexible_tree(oracle, msa, p, gamma_tensor, gamma_weight, gap_open, gap_extend):
    """Every join of a flexible progressive alignment replayed by the oracle on the device's own child nodes: bit-identical."""
    sizes = [1] * p
    for k, (n1, n2) in enumerate(_replay_tree(msa, msa.tree, p)):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        a1, a2, tn, wn = oracle.progressive_node_flexible(s1.tensors, msa.final_consensus_weights[n1], s2.tensors, msa.final_consensus_weights[n2],
                                                          sizes[n2] / (2 * tot), sizes[n1] / (2 * tot), gamma_tensor, gamma_weight, gap_open, gap_extend)
        node = msa.final_sequences[p + k]
        assert node.coordinates is None
        assert np.array_equal(tn, node.tensors) and np.array_equal(wn, msa.final_consensus_weights[p + k]), f"node {k}"
        sizes.append(tot)


@pytest.mark.parametrize("tag", ["G8", "G5"])
def test_flexible_progressive_golden_and_oracle(oracle, golden, tag):
    """flexible=True in score AND mean function (multiple_alignment.py:323-326, :351-362) with every tree node on the device
    (cr_progressive_align_flexible: node scores = tensor RBF + consensus-weight RBF by their own launches, DTW sweep, mean
    tensors and weights): the reference's own multiple_align (f10_flexible_progressive.npz) -- pairwise matrix, tree, MSA exact,
    node tensors and weights exact -- and the oracle replaying every join on the device's own children."""
    from caretta_amd import multiple_alignment as ma
    g = golden("f10_flexible_progressive.npz")
    tensors, offsets = g[f"fam{tag}_tensors"], g[f"fam{tag}_offsets"]
    p = len(offsets) - 1
    prots = [ma.Protein(f"s{i:04d}", tensors[offsets[i]:offsets[i + 1]]) for i in range(p)]       # tensors only
    msa = ma.MultipleAlignment(prots)
    sf = dict(flexible=True, gamma_tensor=7.0)
    m = msa.make_pairwise_matrix(sf)
    np.testing.assert_allclose(m.max() - m, g[f"fam{tag}_D"], rtol=1e-9, atol=1e-12)
    aln = msa.multiple_align(g[f"fam{tag}_D"], gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0, gamma_weight=1.0,
                             score_function_params=sf, mean_function_params=dict(flexible=True))
    assert getattr(msa, "node_table", None) is not None                  # (the resident path ran, not the host walk)
    assert np.array_equal(np.array([aln[q.name] for q in prots]), g[f"fam{tag}_msa"])
    assert np.array_equal(msa.tree, g[f"fam{tag}_tree"])
    nn = int(g[f"fam{tag}_nnodes"])
    assert len(msa.final_sequences) == p + nn
    for k in range(nn):
        node, w = msa.final_sequences[p + k], msa.final_consensus_weights[p + k]
        assert node.coordinates is None
        assert np.array_equal(node.tensors, g[f"fam{tag}_n{k}_tensors"]) and np.array_equal(w, g[f"fam{tag}_n{k}_weights"])
    _replay_flexible_tree(oracle, msa, p, 7.0, 1.0, 1.0, 0.01)


@pytest.mark.parametrize("num,length,dim,seed", [(24, 300, 10, 21), (7, 700, 7, 22), (40, 90, 16, 23)])
def test_flexible_progressive_resident_vs_oracle_and_host_walk(oracle, monkeypatch, num, length, dim, seed):
    """The resident flexible tree on larger families (several strips per node, two rows per lane, padded tensor widths, ragged
    lengths): every join against the oracle, and the same MSA as the generic host walk (score_function / dtw_align /
    mean_function per node)."""
    from caretta_amd import multiple_alignment as ma
    fam = synthetic.make_family(num, length, dim=dim, seed=seed, ragged=True, clades=3)
    prots = [ma.Protein(s.name, s.tensors) for s in fam]
    msa = ma.MultipleAlignment(prots)
    sf = dict(flexible=True, gamma_tensor=2.5)
    m = msa.make_pairwise_matrix(sf)
    aln = msa.multiple_align(m.max() - m, gap_open_penalty=0.8, gap_extend_penalty=0.02, consensus_weight=1.0, gamma_weight=0.7,
                             score_function_params=sf, mean_function_params=dict(flexible=True))
    assert getattr(msa, "node_table", None) is not None
    _replay_flexible_tree(oracle, msa, num, 2.5, 0.7, 0.8, 0.02)
    walk = ma.MultipleAlignment([ma.Protein(s.name, s.tensors) for s in fam])

    def refuse(*args, **kwargs):                  # what cr_progressive_align_flexible answers when a node outgrows its launch bound
        err = ma._capi.CarettaHipError("a tree node outgrew the launch bound")
        err.code = ma._capi.CR_ERR_STATE
        raise err

    monkeypatch.setattr(ma.MultipleAlignment, "_progressive_align_resident", refuse)
    walk.multiple_align(m.max() - m, gap_open_penalty=0.8, gap_extend_penalty=0.02, consensus_weight=1.0, gamma_weight=0.7,
                        score_function_params=sf, mean_function_params=dict(flexible=True))
    assert getattr(walk, "node_table", None) is None
    for q in prots:
        assert np.array_equal(walk.alignment[q.name], aln[q.name])


def _replay_tree(msa, tree, p):
    tree = np.asarray(tree).astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)]
    return joins + [(int(tree[-1, 0]), int(tree[-1, 1]))]


@pytest.mark.parametrize("num,length,ragged,seed", [(6, 250, True, 11), (9, 70, True, 12), (4, 330, False, 13)])
def test_progressive_resident_vs_oracle_and_single_node(oracle, num, length, ragged, seed):
    """cr_progressive_align (whole tree resident, level by level) against (a) the oracle replaying every join on the
    GPU's own children and (b) the single-node entry point cr_progressive_node; rows > 192 use the 5-rows-per-lane
    kernels, 330-residue leaves make nodes that need two strips."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic
    fam = synthetic.make_family(num, length, seed=seed, ragged=ragged, clades=2)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    m = msa.make_pairwise_matrix(prm)
    tree, _ = nj.neighbor_joining(m.max() - m)
    aln = msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
    assert msa.node_table.shape == (num - 1, 6) and msa.node_table[-1, 5] == num
    width = len(aln[prots[0].name])
    for q in prots:                                   # every residue exactly once, in order
        row = aln[q.name]
        assert len(row) == width and np.array_equal(row[row != -1], np.arange(len(q)))
    sizes, rows = [1] * num, [np.arange(len(q))[None, :] for q in prots]
    for k, (n1, n2) in enumerate(_replay_tree(msa, tree, num)):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        w1, w2 = msa.final_consensus_weights[n1], msa.final_consensus_weights[n2]
        m1, m2 = sizes[n2] / (2 * tot), sizes[n1] / (2 * tot)
        a1, a2, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, w1, s2.coordinates, s2.tensors, w2, m1, m2)
        node, w = msa.final_sequences[num + k], msa.final_consensus_weights[num + k]
        assert np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors) and np.array_equal(wn, w)
        b1, b2, node2, w2n = ma._progressive_node(s1, s2, w1, w2, m1, m2, "x", 1.0, 0.01, 1.0, prm, dict(verbose=False))
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2)
        assert np.array_equal(node2.coordinates, node.coordinates) and np.array_equal(node2.tensors, node.tensors)
        assert np.array_equal(w2n, w)
        rows[n1] = np.where(a1 != -1, rows[n1][:, a1], -1)
        rows[n2] = np.where(a2 != -1, rows[n2][:, a2], -1)
        rows.append(np.vstack([rows[n1], rows[n2]]))
        sizes.append(tot)
    order = [q for q in msa.final_alignments["int-final"]]
    assert np.array_equal(np.array([aln[q] for q in order]), rows[-1])


def test_progressive_levels_of_more_than_64_nodes(oracle):
    """A tree whose first levels hold more than 64 nodes: the one-workgroup planning kernel (k_plan_level: lengths, arena offsets and
    decision-scratch offsets of a level from the lengths the level before produced) then spreads a level over all four of
    its waves -- its phases are separated by workgroup barriers, not wave barriers.  360 short structures (178 cherries in
    the first level), every join replayed by the oracle on the device's own children."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic
    num = 360
    fam = synthetic.make_family(num, 36, seed=4242, ragged=True, clades=8)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    m = msa.make_pairwise_matrix(prm)
    tree, _ = nj.neighbor_joining(m.max() - m)
    for rep in range(3):                                   # (a race would not show every time)
        aln = msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
        levels = msa.node_table[:, 3]
        assert np.bincount(levels).max() > 64, np.bincount(levels)
        width = len(aln[prots[0].name])
        for q in prots:
            row = aln[q.name]
            assert len(row) == width and np.array_equal(row[row != -1], np.arange(len(q)))
        sizes = [1] * num
        for k, (n1, n2) in enumerate(_replay_tree(msa, tree, num)):
            tot = sizes[n1] + sizes[n2]
            s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
            _, _, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, msa.final_consensus_weights[n1], s2.coordinates, s2.tensors,
                                                          msa.final_consensus_weights[n2], sizes[n2] / (2 * tot), sizes[n1] / (2 * tot))
            node = msa.final_sequences[num + k]
            assert np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors), f"node {k} (level {levels[k]})"
            assert np.array_equal(wn, msa.final_consensus_weights[num + k])
            sizes.append(tot)


@pytest.mark.parametrize("num,length,ragged,seed", [(12, 300, False, 21), (5, 335, False, 22), (7, 200, True, 23), (9, 140, True, 24),
                                                    (5, 600, True, 25), (3, 675, False, 26), (4, 400, False, 27),
                                                    (3, 900, False, 28), (3, 1300, True, 29)])
def test_progressive_staged_scores_equal_fused(oracle, monkeypatch, num, length, ragged, seed):
    """The tree levels whose scores are formed by their own launches (cr_staged.h) against the fused kernels
    (CARETTA_STAGED=0): alignments, node coordinates / tensors / weights and flags bit for bit; 335-residue leaves size the
    launches for 511 rows (all 8 waves of the staged sweep with one row per lane), ragged 140-residue ones for 218 (4 waves,
    most nodes fewer); 400 / 600 / 675-residue leaves for 608 / 908 / 1021 rows: TWO rows per lane, 5 / 8 / 8 waves; 900 /
    1300-residue leaves for 1358 / 1958 rows: three and four rows per lane (blocks of 8 steps, skewed seed sweep; the
    comparison run then takes the level-by-level single-wave kernels).  The root join also against the oracle."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj
    fam = synthetic.make_family(num, length, seed=seed, ragged=ragged, clades=2)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    runs = []
    for staged in ("1", "0"):
        monkeypatch.setenv("CARETTA_STAGED", staged)
        prots = [ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam]
        msa = ma.MultipleAlignment(prots)
        if not runs:
            m = msa.make_pairwise_matrix(prm)
            tree, _ = nj.neighbor_joining(m.max() - m)
        aln = msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
        runs.append((msa, aln))
    (a, aln_a), (b, aln_b) = runs
    assert np.array_equal(a.node_table, b.node_table)
    for q in aln_a:
        assert np.array_equal(aln_a[q], aln_b[q])
    for k in range(num, 2 * num - 1):
        assert np.array_equal(a.final_sequences[k].coordinates, b.final_sequences[k].coordinates)
        assert np.array_equal(a.final_sequences[k].tensors, b.final_sequences[k].tensors)
        assert np.array_equal(a.final_consensus_weights[k], b.final_consensus_weights[k])
    n1, n2 = _replay_tree(a, tree, num)[-1]
    members = a.node_table[:, 5]
    size = lambda x: 1 if x < num else int(members[x - num])
    tot = size(n1) + size(n2)
    s1, s2 = a.final_sequences[n1], a.final_sequences[n2]
    _, _, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, a.final_consensus_weights[n1], s2.coordinates, s2.tensors,
                                                  a.final_consensus_weights[n2], size(n2) / (2 * tot), size(n1) / (2 * tot))
    root = a.final_sequences[2 * num - 2]
    assert np.array_equal(xn, root.coordinates) and np.array_equal(tn, root.tensors)
    assert np.array_equal(wn, a.final_consensus_weights[2 * num - 2])


def test_gamma_too_small_is_rejected(ctx):
    """gamma = 0 would make the scores of the padding rows 1.0 instead of 0.0: the fused kernels refuse it."""
    from caretta_amd import engine, synthetic as syn
    fam = syn.make_family(3, 30, seed=2, clades=1)
    batch = engine.PairBatch(ctx, *syn.pack(fam)).set_pairs(engine.all_pairs(3))
    for bad in (dict(gamma_coords=0.0), dict(gamma_tensor=1e-300)):
        with pytest.raises(ValueError):
            batch.run(engine.make_params(**bad))
    batch.run(engine.make_params(gamma_coords=1e-4))
    batch.close()


def test_progressive_unrelated_structures_outgrow_the_bound(oracle):
    """With rewarded gaps every alignment is all gaps, so tree nodes grow past 1.5 x the longest leaf: the planned
    (host-round-trip-free) launch sequence reports the overflow and the level-by-level path takes over.  Every node
    still has to match the oracle."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj
    fam = [synthetic.make_family(1, ln, seed=500 + k, clades=1)[0] for k, ln in enumerate((300, 280, 260, 310, 295))]
    prots = [ma.Protein(f"u{k}", s.tensors, s.coordinates, "") for k, s in enumerate(fam)]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    m = msa.make_pairwise_matrix(prm)
    tree, _ = nj.neighbor_joining(m.max() - m)
    aln = msa.progressive_align(tree, -0.5, -0.5, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
    width = len(aln["u0"])
    assert width > 1.5 * 310                                 # the case this test is about
    from oracle.pyoracle import default_params
    oprm = default_params(gap_open=-0.5, gap_extend=-0.5)
    sizes = [1] * 5
    for k, (n1, n2) in enumerate(_replay_tree(msa, tree, 5)):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        _, _, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, msa.final_consensus_weights[n1], s2.coordinates,
                                                      s2.tensors, msa.final_consensus_weights[n2], sizes[n2] / (2 * tot),
                                                      sizes[n1] / (2 * tot), oprm)
        node = msa.final_sequences[5 + k]
        assert np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors)
        assert np.array_equal(wn, msa.final_consensus_weights[5 + k])
        sizes.append(tot)


def test_neighbor_joining_device_matches_oracle(ctx, oracle, golden):
    """cr_neighbor_joining_device (one workgroup) against the oracle and the host implementation: trees and branch lengths
    bit for bit on the reference's golden matrices, random symmetric matrices, tie-heavy integer matrices, the 3-node
    case, and an asymmetric matrix (handed to the host implementation)."""
    from caretta_amd import neighbor_joining as nj
    g = golden("f3_tree.npz")
    cases = [(f"golden{c}", g[f"nj{c}_D"]) for c in range(int(g["nnj"]))] + [(f"fam{t}", g[f"fam{t}_D"]) for t in ("T8", "T16")]
    for tag, d in cases[:]:
        tree, _ = nj.neighbor_joining(d, device=True, ctx=ctx)
        key = tag.replace("golden", "nj") + "_tree"
        assert nj.bipartitions(tree, d.shape[0]) == nj.bipartitions(g[key], d.shape[0]), tag
    rng = np.random.default_rng(11)
    for p in (3, 4, 5, 17, 64, 65, 130, 256, 300, 1025, 1100, 2048, 2049):      # 2048: the kernel's largest; 2049: host
        a = rng.uniform(0.5, 40.0, size=(p, p))
        d = a + a.T
        d[np.diag_indices(p)] = rng.uniform(0.0, 80.0)           # constant non-zero diagonal, as max(M) - M has
        cases.append((f"uniform{p}", d))
    for p in (8, 33, 96):                                          # many equal Q values: the first in row-major order wins
        a = rng.integers(1, 4, size=(p, p)).astype(np.float64)
        d = np.triu(a, 1) + np.triu(a, 1).T
        cases.append((f"ties{p}", d))
    cases.append(("zeros12", np.zeros((12, 12))))
    asym = rng.uniform(1.0, 9.0, size=(40, 40))
    cases.append(("asymmetric40", asym))
    huge = rng.uniform(1.0, 9.0, size=(20, 20))
    huge = huge + huge.T
    huge[3, 7] = huge[7, 3] = np.inf                               # not finite: the host implementation's, no hang
    cases.append(("inf20", huge))
    for tag, d in cases:
        p = d.shape[0]
        tree, bl = nj.neighbor_joining(d, device=True, ctx=ctx)
        htree, hbl = nj.neighbor_joining(d, device=False)
        assert np.array_equal(tree, htree) and np.array_equal(bl, hbl), tag
        if p <= 300 and np.all(np.isfinite(d)):
            otree, obl = oracle.neighbor_joining(d, hoist=(p > 40))
            assert np.array_equal(tree.astype(np.int64), np.asarray(otree).astype(np.int64)), tag
            assert np.array_equal(bl.ravel(), np.asarray(obl).ravel()), tag


def test_progressive_tree_validation(ctx):
    from caretta_amd import multiple_alignment as ma, synthetic
    fam = synthetic.make_family(4, 40, seed=5, clades=2)
    msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam])
    bad = np.array([[0, 4], [1, 4], [0, 5], [2, 5], [5, 3]], dtype=np.uint64)      # leaf 0 joined twice
    with pytest.raises(ValueError):
        msa.progressive_align(bad, 1.0, 0.01, 1.0, 1.0, dict(verbose=False), dict(verbose=False))


def test_two_sequence_alignment_and_metrics(ctx, golden):
    """multiple_align's 2-sequence branch and make_rmsd_coverage_tm_matrix through the drop-ins."""
    from caretta_amd import multiple_alignment as ma
    g = golden("f2_pipeline.npz")
    coords, tensors, offsets = g["famB_coords"], g["famB_tensors"], g["famB_offsets"]
    prots = [ma.Protein(f"s{i}", tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]], "A" * 150)
             for i in (0, 1)]
    msa = ma.MultipleAlignment(prots)
    aln = msa.multiple_align(None, 1.0, 0.01, 1.0, 1.0,
                             score_function_params=dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03))
    assert np.array_equal(aln["s0"], g["famB_p0_aln1"]) and np.array_equal(aln["s1"], g["famB_p0_aln2"])
    rmsd, cov, tm = ma.make_rmsd_coverage_tm_matrix(aln, prots, superpose_first=False)
    assert abs(rmsd[0, 1] - float(g["famB_p0_rmsd"])) < 1e-5 and abs(tm[0, 1] - float(g["famB_p0_tm"])) < 1e-5
    assert cov[0, 1] == float(g["famB_p0_coverage"])


def test_config1_three_kringle_domains(golden, tmp_path):
    """BASELINE config 1 (the reference's README example, 3 PDB files) as plumbing through the GPU path:
    pairwise matrix -> max - M -> neighbor joining -> progressive alignment -> FASTA and matrix files.
    Tensors come from the documented stand-in descriptor (geometricus is not available): non-parity, so the
    checks are structural."""
    from caretta_amd import helper, multiple_alignment as ma
    g = golden("c1_kringle_calpha.npz")
    prots = []
    for name in g["names"]:
        xyz = g[f"{name}_coords"]
        prots.append(ma.Protein(str(name), helper.local_shape_descriptor(xyz, 10), xyz, str(g[f"{name}_sequence"])))
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    m = msa.make_pairwise_matrix(prm)
    assert m.shape == (3, 3) and np.array_equal(m, m.T) and np.all(m[np.triu_indices(3, 1)] > 0)
    d = m.max() - m
    aln = msa.multiple_align(d, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0, gamma_weight=1.0,
                             score_function_params=prm, mean_function_params=dict(flexible=False, verbose=False))
    rows = [aln[p.name] for p in prots]
    assert len({len(r) for r in rows}) == 1
    for r, p in zip(rows, prots):
        assert np.array_equal(r[r >= 0], np.arange(len(p)))          # every residue once, in order
    msa.write_alignment(tmp_path / "result.fasta")
    fasta = (tmp_path / "result.fasta").read_text().split("\n")
    assert fasta[0] == ">1kdu" and len(fasta[1]) == len(rows[0]) and fasta[1].replace("-", "") == str(prots[0])
    helper.write_distance_matrix([p.name for p in prots], d, tmp_path / "distance_matrix_guide_tree.txt")
    names, back = helper.read_distance_matrix(tmp_path / "distance_matrix_guide_tree.txt")
    assert names == [p.name for p in prots] and np.allclose(back, d, atol=5e-5)
    rmsd, cov, tm = ma.make_rmsd_coverage_tm_matrix(aln, prots, superpose_first=False)
    assert np.all(rmsd[np.triu_indices(3, 1)] > 0) and np.all(rmsd < 15) and np.all((cov > 0.5) & (cov <= 1))


@pytest.mark.parametrize("tag", ["P8", "P5"])
def test_post_msa_products_golden(golden, tag):
    """superpose(), coverage/gap matrices, reference-structure selection and the RMSD / coverage / TM matrices of
    a finished MSA against the reference's outputs (tests/golden/f5_post_msa.npz)."""
    import copy
    from caretta_amd import msa_superposition as post, multiple_alignment as ma
    g, f = golden("f5_post_msa.npz"), golden("f4_progressive.npz")
    coords, tensors, off = f[f"fam{tag}_coords"], f[f"fam{tag}_tensors"], f[f"fam{tag}_offsets"]
    num = len(off) - 1
    names = [f"s{i:04d}" for i in range(num)]
    prots = [ma.Protein(names[i], tensors[off[i]:off[i + 1]].copy(), coords[off[i]:off[i + 1]].copy(), "") for i in range(num)]
    aln = {names[i]: f[f"fam{tag}_msa"][i] for i in range(num)}
    dist, aligning = post.make_coverage_gap_distance_matrix(np.array([aln[n] for n in names]))
    assert np.array_equal(aligning, g[f"{tag}_cov_aligning"]) and np.allclose(dist, g[f"{tag}_cov_dist"], rtol=0, atol=1e-15)
    first, refs, none = post.get_reference_structures(aln, 50)
    assert first == str(g[f"{tag}_ref_first"]) and list(refs) == list(g[f"{tag}_ref_keys"]) and list(none) == list(g[f"{tag}_ref_none"])
    for k, v in refs.items():
        assert list(v) == list(g[f"{tag}_ref_{k}"])
    r0, c0, t0 = post.make_rmsd_coverage_tm_matrix(aln, copy.deepcopy(prots), superpose_first=False)
    np.testing.assert_allclose(r0, g[f"{tag}_rmsd"], atol=1e-9)
    np.testing.assert_allclose(t0, g[f"{tag}_tm"], atol=1e-9)
    assert np.array_equal(c0, g[f"{tag}_coverage"])
    moved = copy.deepcopy(prots)
    r1, c1, t1 = post.make_rmsd_coverage_tm_matrix(aln, moved, superpose_first=True)
    np.testing.assert_allclose(r1, g[f"{tag}_rmsd_sf"], atol=1e-8)
    np.testing.assert_allclose(t1, g[f"{tag}_tm_sf"], atol=1e-8)
    assert np.array_equal(c1, g[f"{tag}_coverage_sf"])
    for i in range(num):
        np.testing.assert_allclose(moved[i].coordinates, g[f"{tag}_superposed_{i}"], atol=1e-8)
    ref_moved = post.superpose_reference(aln, copy.deepcopy(prots), names[1])
    for i in range(num):
        np.testing.assert_allclose(ref_moved[i].coordinates, g[f"{tag}_superposed_ref1_{i}"], atol=1e-8)
    assert ma.superpose is post.superpose                      # reachable under the reference's module name too


def test_superpose_core_batched_equals_the_per_structure_calls(golden):
    """superpose_core as one launch (cr_superpose_core) against the reference's own formulation with the single-call
    drop-ins: paired_svd_superpose on the core columns, then apply_rotran, structure by structure -- bit-identical."""
    from caretta_amd import helper, multiple_alignment as ma, msa_superposition as msup, superposition_functions as sup
    f = golden("f4_progressive.npz")
    coords, tensors, offsets = f["famP8_coords"], f["famP8_tensors"], f["famP8_offsets"]
    p = len(offsets) - 1
    names = [f"s{i:04d}" for i in range(p)]
    aln = {n: f["famP8_msa"][i] for i, n in enumerate(names)}
    prots = [ma.Protein(n, tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]].copy(), "") for i, n in enumerate(names)]
    rows = np.array([aln[n] for n in names])
    core = np.where((rows != -1).all(axis=0))[0]
    assert len(core) > 3
    ref = 2
    moved = msup.superpose_core(aln, [ma.Protein(q.name, q.tensors, q.coordinates.copy(), "") for q in prots], names[ref], core)
    ref_core = prots[ref].coordinates[aln[names[ref]][core]]
    centroid = helper.nb_mean_axis_0(ref_core)
    ref_core = ref_core - centroid
    for i, q in enumerate(prots):
        if i == ref:
            want = q.coordinates - centroid
        else:
            rot, tran = sup.paired_svd_superpose(ref_core, q.coordinates[aln[q.name][core]])
            want = sup.apply_rotran(q.coordinates, rot, tran)
        assert np.array_equal(moved[i].coordinates, want), i


def test_superpose_reference_batched_equals_the_loop(golden):
    """superpose_reference as three launches (cr_superpose_reference) against the reference's loop written with the
    single-call drop-ins, incl. the refit of the reference onto itself half way through -- bit-identical."""
    from caretta_amd import helper, multiple_alignment as ma, msa_superposition as msup, superposition_functions as sup
    f = golden("f4_progressive.npz")
    coords, tensors, offsets = f["famP8_coords"], f["famP8_tensors"], f["famP8_offsets"]
    p = len(offsets) - 1
    names = [f"s{i:04d}" for i in range(p)]
    aln = {n: f["famP8_msa"][i] for i, n in enumerate(names)}
    make = lambda: [ma.Protein(n, tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]].copy(), "")  # noqa: E731
                    for i, n in enumerate(names)]
    ref = 3
    moved = msup.superpose_reference(aln, make(), names[ref])
    want = make()
    for q in want:
        pos_1, pos_2 = helper.get_common_positions(aln[names[ref]], aln[q.name])
        rot, tran = sup.paired_svd_superpose(want[ref].coordinates[pos_1], q.coordinates[pos_2])
        q.coordinates = sup.apply_rotran(q.coordinates, rot, tran)
    for i in range(p):
        assert np.array_equal(moved[i].coordinates, want[i].coordinates), i


def test_superpose_references_batched_equals_the_loop(golden):
    """superpose_references group by group (cr_superpose_members) against the reference's loop written with the
    single-call drop-ins -- bit-identical, for two coverage thresholds (one and several groups)."""
    from caretta_amd import helper, multiple_alignment as ma, msa_superposition as msup, superposition_functions as sup
    f = golden("f4_progressive.npz")
    coords, tensors, offsets = f["famP8_coords"], f["famP8_tensors"], f["famP8_offsets"]
    p = len(offsets) - 1
    names = [f"s{i:04d}" for i in range(p)]
    aln = {n: f["famP8_msa"][i] for i, n in enumerate(names)}
    make = lambda: [ma.Protein(n, tensors[offsets[i]:offsets[i + 1]], coords[offsets[i]:offsets[i + 1]].copy(), "")  # noqa: E731
                    for i, n in enumerate(names)]
    for coverage in (50, 97):
        moved = msup.superpose_references(aln, make(), coverage)
        want = {q.name: q for q in make()}
        _, groups, _ = msup.get_reference_structures(aln, coverage)
        for reference_name, members in groups.items():
            for name in members:
                pos_1, pos_2 = helper.get_common_positions(aln[reference_name], aln[name])
                rot, tran = sup.paired_svd_superpose(want[reference_name].coordinates[pos_1], want[name].coordinates[pos_2])
                want[name].coordinates = sup.apply_rotran(want[name].coordinates, rot, tran)
        for q in moved:
            assert np.array_equal(q.coordinates, want[q.name].coordinates), (coverage, q.name)


def test_integration_md_binding_snippet(golden):
    """The hand-written ctypes binding shown in INTEGRATION.md section 2 runs as printed and reproduces the reference's
    pairwise matrix (golden family A)."""
    import re
    import types
    from pathlib import Path
    from caretta_amd import _capi
    text = (Path(__file__).resolve().parents[1] / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(# caretta/_hip\.py.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libcaretta_hip.so")', f'C.CDLL(r"{_capi.LIB_PATH}")')
    mod = types.ModuleType("caretta_hip_binding")
    exec(compile(code, "INTEGRATION.md", "exec"), mod.__dict__)
    g = golden("f2_pipeline.npz")
    coords, tensors, offsets = g["famA_coords"], g["famA_tensors"], g["famA_offsets"]

    class Seq:                                     # what the snippet needs of a Protein
        def __init__(self, x, t):
            self.coordinates, self.tensors = x, t

        def __len__(self):
            return len(self.coordinates)

    seqs = [Seq(coords[offsets[i]:offsets[i + 1]], tensors[offsets[i]:offsets[i + 1]]) for i in range(len(offsets) - 1)]
    m = mod.make_pairwise_matrix(seqs, gamma_tensor=7.0, gamma_coords=0.03)
    pairs = g["famA_pairs"]
    for p, (i, j) in enumerate(pairs):
        assert abs(m[i, j] - float(g[f"famA_p{p}_sw"])) <= 1e-9 * max(1.0, abs(m[i, j])) and m[i, j] == m[j, i]


def test_sharded_matrix_single_rank_ragged():
    """caretta_amd.distributed.pairwise_matrix_sharded on one rank: the kernels write the scores straight into a torch
    tensor; a ragged family sends them through the scatter kernel (tests/sharded_single_rank_check.py; its own process,
    because torch has to initialise its HIP runtime before libcaretta_hip does)."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "sharded_single_rank_check.py"
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "sharded matrix ok" in out.stdout


def test_very_long_pairs_match_the_oracle():
    """3000 x 2500 residues: ten 320-row strips per sweep with their hand-off rows through HBM (tests/long_pair_oracle_check.py)."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "long_pair_oracle_check.py"
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "long pairs ok" in out.stdout


def test_randomised_parity_run():
    """tests/fuzz_parity.py for a minute: random ragged batches, all parameter settings, the resident progressive
    alignment and the explicit-matrix drop-ins, everything bit-identical to the oracle (longer runs: profiles/)."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "fuzz_parity.py"
    out = subprocess.run([sys.executable, str(script), "60", "99"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "all bit-identical to the oracle" in out.stdout


def test_rccl_all_gather_single_rank():
    """The RCCL code path on a one-GPU box: one rank under torch.distributed.run (backend nccl), all-gather forced,
    matrix identical to the non-distributed one; the gather is ordered behind the kernels without a host sync."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "rccl_single_rank_check.py"
    env = dict(os.environ, CARETTA_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl single-rank matrix ok" in out.stdout


def test_bench_line_of_an_n_rank_run_proves_itself():
    """Every branch an N-rank `bench.py` line runs, on the one-GPU box: ONE rank under torch.distributed.run with
    CARETTA_FORCE_DIST=1 (process group over RCCL, all-gather, barrier, max-reduce).  The line must carry, for BASELINE configs
    3, 4 and 5, `multi_gpu_gate` (gathered score vector = the one-GPU vector bit for bit, identical neighbor-joining trees),
    the oracle's `pair_gate_own_share`, the all-gather's own time, the rank -> device records with the RCCL version, `value` =
    the FIXED 128 x 300 pair set (scaling "strong"), no failed gate, exit code 0 -- and `shares` as its LAST key."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, CARETTA_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", str(root / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(root))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert list(rec)[-1] == "shares" and rec["gates_failed"] == []
    assert rec["scaling"] == "strong" and rec["config"]["pairs"] == 8128 and rec["config"]["structures"] == 128 and rec["value_weak"] == rec["value"]
    assert rec["ranks"]["ranks"][0]["rank"] == 0 and rec["ranks"]["ranks"][0]["pci_bus_id"] and rec["ranks"]["collective"]["backend"] == "nccl"
    assert rec["ranks"]["collective"]["rccl_version"]
    for key, taxa in (("c3", 128), ("c4", 512), ("c5", 64)):
        sh = rec[f"{key}_sharded"]
        gate = sh["multi_gpu_gate"]
        assert gate["matrix_equal"] is True and gate["trees_identical"] is True and gate["nan_scores"] == 0 and gate["tree_rows"] == 2 * taxa - 3
        assert sh["pair_gate_own_share"]["mismatches"] == 0 and sh["pair_gate_own_share"]["fraction"] >= 0.01
        assert isinstance(sh["all_gather_ms"], float) and sh["all_gather_ms"] > 0
        assert sh["nj_gate"]["trees_identical"] and sh["pair_gate"]["mismatches"] == 0            # (the N = 1 gates stay)
        row = rec["shares"][key]
        assert row["measured_1"][3] is True and row["measured_1"][4] is True and row["measured_1"][5] == 0 and len(row["8"]) == 2


def test_single_process_multi_device_path_equals_the_batch(ctx):
    """cr_multi_pairwise_scores (one process, a context per listed GPU, cr_partition_pairs, one grouped RCCL all-gather)
    with the device list [0]: scores and flags equal cr_batch_run_scores bit for bit, ragged (scatter kernel) and equal
    lengths; MultipleAlignment._pairwise_matrix_multi gives make_pairwise_matrix's matrix.  (More devices than one are
    not available on the test box: the deal itself is pinned on the CPU, tests/test_capi_cpu.py.)"""
    from caretta_amd import engine
    from caretta_amd import multiple_alignment as ma
    multi = engine.MultiDevice([0])
    assert multi.num_devices == 1
    for fam in (synthetic.make_family(11, 70, seed=4041, ragged=True, clades=2), synthetic.make_family(24, 150, seed=4042)):
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(len(fam))
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(engine.make_params(), scores_only=True)
        sw_ref, flags_ref = batch.fetch_scores()
        batch.close()
        for _ in range(2):                                 # the second call reuses the communicator and the buffers
            sw, flags = multi.pairwise_scores(coords, tensors, offsets, engine.make_params())
            assert np.array_equal(sw, sw_ref) and np.array_equal(flags, flags_ref)
        msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam])
        prm = dict(gamma_tensor=7.0, gamma_coords=0.03)
        assert np.array_equal(msa._pairwise_matrix_multi(multi, prm), msa.make_pairwise_matrix(prm))
    # a structure without any positive local alignment of the tensor scores: the reference's TypeError, not a NaN fault
    far = synthetic.make_family(3, 30, seed=4043, clades=1)
    far[1].tensors = far[1].tensors + 1e3
    msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in far])
    with pytest.raises(TypeError):
        msa._pairwise_matrix_multi(multi, dict(gamma_tensor=7.0, gamma_coords=0.03))
    multi.close()
    # More than one share on the one GPU of the box: the device list may name a device twice when
    # CARETTA_MULTI_ALLOW_DUPLICATES is set; the shares are then gathered with device copies (RCCL refuses such a
    # communicator), everything else -- the deal, a host thread and a context per share, the share layout, the scatter back
    # to pair order -- is the product path.  3 and 4 shares, ragged and equal lengths, pair counts that do not divide.
    import os
    os.environ["CARETTA_MULTI_ALLOW_DUPLICATES"] = "1"
    engine.reload_config()
    try:
        for shares in (3, 4):
            multi = engine.MultiDevice([0] * shares)
            assert multi.num_devices == shares
            for fam in (synthetic.make_family(11, 70, seed=4041, ragged=True, clades=2), synthetic.make_family(23, 150, seed=4044)):
                coords, tensors, offsets = synthetic.pack(fam)
                pairs = engine.all_pairs(len(fam))
                batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
                batch.run(engine.make_params(), scores_only=True)
                sw_ref, flags_ref = batch.fetch_scores()
                batch.close()
                sw, flags = multi.pairwise_scores(coords, tensors, offsets, engine.make_params())
                assert np.array_equal(sw, sw_ref) and np.array_equal(flags, flags_ref)
            multi.close()
    finally:
        del os.environ["CARETTA_MULTI_ALLOW_DUPLICATES"]
        engine.reload_config()


def test_streamed_run_writes_what_fetch_copies(ctx):
    """cr_batch_run_stream_i32: the alignment kernels store rows and records straight into page-locked host arrays.  Same
    bytes as cr_batch_fetch_i32 afterwards, for every kernel family: one wave per pair (ragged: launch order differs from
    the caller's), the staged sweeps (one and two rows per lane), the wide layout; pageable arrays are refused."""
    from caretta_amd import _capi, engine
    cases = [(synthetic.make_family(12, 90, seed=5051, ragged=True, clades=2), None),
             (synthetic.make_family(20, 150, seed=5052), None),
             (synthetic.make_family(4, 300, seed=5053, clades=1), None),               # 6 pairs of 300 rows: teams
             (synthetic.make_family(3, 700, seed=5054, clades=1), None),               # 700 rows: staged, two rows per lane
             (synthetic.make_family(3, 1100, seed=5055, clades=1), None),              # 1100 rows: staged, three rows per lane
             (synthetic.make_family(3, 2100, seed=5056, clades=1), None)]              # 2100 rows: wide kernels
    for fam, _ in cases:
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(len(fam))
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        res_s, aln_s = batch.run_streamed(engine.make_params())
        ctx.synchronize()
        res_s, aln_s = res_s.copy(), aln_s.copy()
        res_f, aln_f = batch.fetch(want_alignments=True, pinned=False)
        assert res_s.tobytes() == res_f.tobytes()
        for p in range(len(pairs)):                        # (the streamed rows carry no -2 padding behind aln_len)
            ln = int(res_f["aln_len"][p])
            assert np.array_equal(aln_s[p, :, :ln], aln_f[p, :, :ln])
        # records only
        r_only, none = batch.run_streamed(engine.make_params(), want_alignments=False)
        ctx.synchronize()
        assert none is None and r_only.tobytes() == res_f.tobytes()
        with pytest.raises(ValueError):
            import ctypes as C
            bad = np.zeros(len(pairs), dtype=_capi.PAIR_RESULT_DTYPE)
            _capi.check(batch._lib.cr_batch_run_stream_i32(batch._h, C.byref(engine.make_params()), _capi.ptr(bad), None, 0, None))
        batch.close()


def test_wide_strip_plans_vs_oracle(ctx, oracle, monkeypatch):
    """The wide layout (one workgroup per pair, one wave per strip, both stages in one launch) with strips of UNEQUAL rows
    per lane -- 3 in the first nA strips, 2 in the others, CARETTA_WIDE=RA,RB,nA,B -- on ragged lengths, so that pairs of one
    launch end in different strips and zones; every output bit-identical to the oracle, full pipeline and scores only,
    with and without a Smith-Waterman gap (skewed seed sweep)."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(5, 700, seed=6061, ragged=True, clades=1)
    cuts = [700, 130, 450, 577, 193]
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.vstack([engine.all_pairs(5), engine.all_pairs(5)[:, ::-1]])
    ref = {gap: oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=8) for gap in (0.0, 0.05)}
    for plan in ("3,2,1,8", "3,2,2,8", "3,2,3,2", "2,2,0,8", "3,3,0,4"):
        monkeypatch.setenv("CARETTA_WIDE", plan)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        for gap in (0.0, 0.05):
            prm = engine.make_params(sw_gap=gap)
            batch.run(prm)
            res, aln = batch.fetch()
            assert_bit_identical(res, aln, *ref[gap])
            batch.run(prm, scores_only=True)
            sw, flags = batch.fetch_scores()
            assert np.array_equal(sw, ref[gap][0]["sw"])
        batch.close()


@pytest.mark.parametrize("dim,seed", [(10, 8081), (4, 8082), (16, 8083)])
def test_staged_pair_batches_vs_oracle_and_fused(ctx, oracle, monkeypatch, dim, seed):
    """Short pair lists run on STAGED scores (cr_staged.h: the RBF scores formed by their own launches, the sweeps with one
    row per lane).  Ragged lengths from 1 to 512 rows (1 .. 8 waves per pair, partial last strips, fewer columns than a
    block), both orientations of every pair, with and without a Smith-Waterman gap (column sweep / skewed sweep of the
    seed), full pipeline, scores only and the streamed run: bit-identical to the oracle and to the fused kernels
    (CARETTA_STAGED=0: teams and wide layout)."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(8, 512, dim=dim, seed=seed, ragged=True, clades=2)
    cuts = [512, 1, 449, 64, 65, 300, 7, 130]
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.vstack([engine.all_pairs(8), engine.all_pairs(8)[:, ::-1]])
    for gap in (0.0, 0.05):
        prm = engine.make_params(sw_gap=gap)
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=8)
        got = {}
        for staged in ("1", "0"):
            monkeypatch.setenv("CARETTA_STAGED", staged)
            batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
            batch.run(prm)
            res, aln = batch.fetch()
            assert_bit_identical(res, aln, ref, ref_aln)
            batch.run(prm, scores_only=True)
            sw, flags = batch.fetch_scores()
            assert np.array_equal(sw, ref["sw"])
            res_s, aln_s = batch.run_streamed(prm)
            ctx.synchronize()
            assert res_s.tobytes() == res.tobytes()
            for p in range(len(pairs)):
                ln = int(res["aln_len"][p])
                assert np.array_equal(aln_s[p, :, :ln], aln[p, :, :ln])
            got[staged] = (res.tobytes(), aln.copy(), sw.copy(), flags.copy())
            batch.close()
        assert got["1"][0] == got["0"][0] and np.array_equal(got["1"][1], got["0"][1])
        assert np.array_equal(got["1"][2], got["0"][2]) and np.array_equal(got["1"][3], got["0"][3])


@pytest.mark.parametrize("longest,cuts", [(1024, [1024, 513, 700, 90]), (2048, [2048, 1100, 1537, 300])])
def test_staged_pair_batches_two_rows_per_lane(ctx, oracle, monkeypatch, longest, cuts):
    """The staged sweeps with TWO rows per lane (513 .. 1024 rows), three and four (.. 2048 rows; blocks of 8 steps, skewed
    seed sweep): ragged pairs, both orientations, gap 0 and 0.05, full pipeline and scores only, bit-identical to the
    oracle."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(4, longest, seed=8181, ragged=True, clades=1)
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.vstack([engine.all_pairs(4), engine.all_pairs(4)[:, ::-1]])
    for gap in (0.0, 0.05):
        prm = engine.make_params(sw_gap=gap)
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=8)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(prm)
        res, aln = batch.fetch()
        assert_bit_identical(res, aln, ref, ref_aln)
        batch.run(prm, scores_only=True)
        sw, flags = batch.fetch_scores()
        assert np.array_equal(sw, ref["sw"])
        batch.close()


def test_short_list_with_a_very_long_structure_leaves_the_staged_path(ctx, oracle):
    """One pair of 64 x 22 000 residues: a single strip, so the list qualifies for staged scores -- but the workgroup-wide
    sums of that path share the LDS with the n + m alignment columns, which do not fit beside the term tile; the fused
    kernel (whose limit is n + m <= 39 000) has to take it.  Both orientations, bit-identical to the oracle."""
    from caretta_amd import engine
    a = synthetic.make_family(1, 64, seed=9091, clades=1)[0]
    b = synthetic.make_family(1, 22000, seed=9092, clades=1)[0]
    coords, tensors, offsets = synthetic.pack([a, b])
    for pairs in (np.array([[0, 1]], dtype=np.int32), np.array([[1, 0]], dtype=np.int32)):
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(engine.make_params())
        res, aln = batch.fetch()
        batch.close()
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs)
        assert_bit_identical(res, aln, ref, ref_aln)


def test_wide_layout_at_the_lds_limit(ctx, oracle, monkeypatch):
    """A pair list whose resident tensor columns (Smith-Waterman gap != 0: the skewed seed sweep keeps all m columns of
    width 16 in LDS) fill the CU's 160 KB to the last byte: the fused wide kernel's static LDS must still fit (found by
    tests/fuzz_parity.py), results bit-identical to the oracle.  (CARETTA_STAGED=0: one pair of 500 rows would otherwise run
    on staged scores.)"""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    monkeypatch.setenv("CARETTA_STAGED", "0")
    for m in (1245, 1246, 1247, 1260):
        a = synthetic.make_family(1, 500, dim=16, seed=7071, clades=1)[0]
        b = synthetic.make_family(1, m, dim=16, seed=7072, clades=1)[0]
        coords, tensors, offsets = synthetic.pack([a, b])
        pairs = np.array([[0, 1]], dtype=np.int32)
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(engine.make_params(sw_gap=0.05))
        res, aln = batch.fetch()
        batch.close()
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=0.05))
        assert_bit_identical(res, aln, ref, ref_aln)
