"""CPU-side checks of the C ABI library: it loads, exports every symbol the header declares, its
host-side pieces (neighbor joining, common positions, matrix assembly) match the golden vectors,
and compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from caretta_amd import _capi
    return _capi.load()


def test_exports_match_header(lib):
    header = (ROOT / "include" / "caretta_hip.h").read_text()
    declared = set(re.findall(r"\b(cr_[a-z0-9_]+)\s*\(", header))
    declared -= {"cr_status"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/caretta_hip.h but not exported"
    from caretta_amd import _capi
    assert declared == set(_capi.SIGNATURES) | {"cr_last_error"}


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from caretta_amd import _capi, engine, score_functions
    assert engine.device_count() == 0
    with pytest.raises(_capi.CarettaHipError):
        engine.Context(0)
    with pytest.raises(_capi.CarettaHipError):
        score_functions.make_score_matrix(np.zeros((2, 3)), np.zeros((2, 3)), score_functions.get_gaussian_score, 1.0)
    # the reference's own entry points: every compute path of the mirror needs the device
    from caretta_amd import dynamic_time_warping as dtw, multiple_alignment as ma, superposition_functions as sup, synthetic
    fam = synthetic.make_family(3, 20, seed=1, clades=1)
    msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam])
    tree = np.array([[0, 3], [1, 3], [3, 2]], dtype=np.uint64)
    for call in (lambda: msa.make_pairwise_matrix(dict(gamma_tensor=7.0, gamma_coords=0.03)),
                 lambda: msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, dict(verbose=False), dict(verbose=False)),
                 lambda: msa.sequences[0].score_function(msa.sequences[1], verbose=False),
                 lambda: dtw.dtw_align(np.arange(3), np.arange(3), np.ones((3, 3)), 1.0, 0.01),
                 lambda: dtw.smith_waterman_score(np.arange(3), np.arange(3), np.ones((3, 3))),
                 lambda: sup.paired_svd_superpose(np.eye(3), np.eye(3)),
                 lambda: ma.tm_score(np.eye(3), np.eye(3), 3, 3)):
        with pytest.raises(_capi.CarettaHipError):
            call()


def test_product_does_not_touch_oracle():
    for path in (ROOT / "caretta_amd").rglob("*"):
        if path.suffix in {".py", ".h", ".hip", ".cpp"}:
            text = path.read_text()
            assert "pyoracle" not in text and "caretta_oracle" not in text.replace("oracle/caretta_oracle.c", ""), path


def test_neighbor_joining_host(golden):
    from caretta_amd import neighbor_joining as nj
    g = golden("f3_tree.npz")
    for c in range(int(g["nnj"])):
        d = g[f"nj{c}_D"]
        p = d.shape[0]
        tree, bl = nj.neighbor_joining(d)
        assert tree.shape == (2 * p - 3, 2) and tree.dtype == np.uint64 and bl.shape == (2 * p - 3, 1)
        assert nj.bipartitions(tree, p) == nj.bipartitions(g[f"nj{c}_tree"], p)
        np.testing.assert_allclose(np.sort(bl.ravel()), np.sort(g[f"nj{c}_branch_lengths"].ravel()), atol=1e-9)
        # rows come in sibling pairs sharing a parent; the last row is the root join (multiple_alignment.py:236-245)
        assert np.all(tree[0:-1:2, 1] == tree[1::2, 1]) and tree[-1, 1] == tree[-2, 1]
    for fam in ("T8", "T16"):
        p = g[f"fam{fam}_D"].shape[0]
        tree, _ = nj.neighbor_joining(g[f"fam{fam}_D"])
        assert nj.bipartitions(tree, p) == nj.bipartitions(g[f"fam{fam}_tree"], p)
    with pytest.raises(ValueError):
        nj.neighbor_joining(np.zeros((2, 2)))


def test_neighbor_joining_matches_oracle_sequential(oracle):
    from caretta_amd import neighbor_joining as nj
    rng = np.random.default_rng(9)
    for p in (3, 4, 7, 33, 128):
        x = rng.uniform(1, 9, size=(p, p))
        d = (x + x.T) / 2
        np.fill_diagonal(d, 2.5)
        tree, bl = nj.neighbor_joining(d)
        otree, obl = oracle.neighbor_joining(d, hoist=(p > 40))
        assert np.array_equal(tree, otree) and np.array_equal(bl, obl)


def test_neighbor_joining_threaded_matches_oracle(oracle, monkeypatch):
    """Above 640 taxa the two O(n^2) passes are shared between helper threads; the tree must not depend on the thread
    count (every row is summed and searched by one thread in the reference's order), incl. tie-heavy matrices and the
    compaction / switch back to one thread below 640 live nodes."""
    from caretta_amd import neighbor_joining as nj
    rng = np.random.default_rng(4)
    x = rng.normal(size=(700, 4))
    d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
    ties = np.round(d * 2) / 2
    for mat in (d, ties):
        otree, obl = oracle.neighbor_joining(mat, hoist=True)
        for threads in ("1", "3", "8"):
            monkeypatch.setenv("CARETTA_NJ_THREADS", threads)
            tree, bl = nj.neighbor_joining(mat)
            assert np.array_equal(tree, otree) and np.array_equal(bl, obl), threads


def test_common_positions_and_assembly(golden):
    from caretta_amd import engine, helper
    g = golden("f1_misc.npz")
    for c in range(int(g["ncp"])):
        p1, p2 = helper.get_common_positions(g[f"cp{c}_a1"], g[f"cp{c}_a2"])
        assert np.array_equal(p1, g[f"cp{c}_p1"]) and np.array_equal(p2, g[f"cp{c}_p2"])
    pairs = engine.all_pairs(5)
    assert pairs.shape == (10, 2) and tuple(pairs[0]) == (0, 1) and tuple(pairs[-1]) == (3, 4)
    m = engine.assemble_matrix(pairs, np.arange(10.0) + 1, 5)
    assert np.array_equal(m, m.T) and np.all(np.diag(m) == 0) and m[0, 1] == 1 and m[3, 4] == 10


def test_synthetic_family_shapes():
    from caretta_amd import synthetic
    fam = synthetic.make_family(32, 150, seed=20241)
    assert len(fam) == 32 and all(s.coordinates.shape == (150, 3) and s.tensors.shape == (150, 10) for s in fam)
    steps = np.linalg.norm(np.diff(fam[0].coordinates, axis=0), axis=1)
    assert np.median(steps) > 2.0
    rag = synthetic.make_family(8, 100, seed=3, ragged=True)
    assert all(80 <= len(s.sequence) <= 100 for s in rag)
    c, t, o = synthetic.pack(rag)
    assert o[-1] == c.shape[0] == t.shape[0]
    again = synthetic.make_family(8, 100, seed=3, ragged=True)
    assert all(np.array_equal(a.coordinates, b.coordinates) for a, b in zip(rag, again))


def test_on_disk_formats(tmp_path):
    from caretta_amd import helper
    names = ["a", "b/chainA", "c"]
    m = np.array([[0.0, 1.23456, 2.5], [1.23456, 0.0, 3.0], [2.5, 3.0, 0.0]])
    helper.write_distance_matrix(names, m, tmp_path / "m.txt")
    text = (tmp_path / "m.txt").read_text().splitlines()
    assert text[0] == "3" and text[1] == "a 0.0000 1.2346 2.5000"       # helper.py:183-202: "%.4f", count first
    rnames, rm = helper.read_distance_matrix(tmp_path / "m.txt")
    assert rnames == ["a", "b", "c"] and np.allclose(rm, m, atol=5e-5)
    pdb = tmp_path / "x.pdb"
    pdb.write_text(
        "ATOM      1  N   THR A   1      37.078  -8.422  -5.315  1.00  0.00           N\n"
        "ATOM      2  CA  THR A   1      37.419  -8.016  -3.919  1.00  0.00           C\n"
        "ATOM      3  CA AGLY A   2      38.000  -7.000  -2.000  0.50  0.00           C\n"
        "ATOM      4  CA BGLY A   2      39.000  -7.000  -2.000  0.50  0.00           C\n"
        "HETATM    5  CA  MSE A   3      40.000  -6.000  -1.000  1.00  0.00           C\n"
        "ATOM      6  CA  ALA B   1      50.000   0.000   0.000  1.00  0.00           C\n"
        "ENDMDL\n"
        "ATOM      7  CA  ALA A   4      60.000   0.000   0.000  1.00  0.00           C\n")
    xyz, seq = helper.read_calpha_pdb(pdb)
    assert seq == "TGM" and xyz.shape == (3, 3) and xyz[1, 0] == 38.0
    desc = helper.local_shape_descriptor(xyz, 10)
    assert desc.shape == (3, 10) and np.all((desc >= 0) & (desc < 1))


def test_c1_fixture(golden):
    g = golden("c1_kringle_calpha.npz")
    assert list(g["names"]) == ["1kdu", "1pk4", "1pkr"]
    assert [g[f"{n}_coords"].shape[0] for n in g["names"]] == [85, 79, 80]
    assert all(len(str(g[f"{n}_sequence"])) == g[f"{n}_coords"].shape[0] for n in g["names"])


def test_small_host_helpers(golden):
    """alignment_to_numpy, helper.normalize / nb_std_axis_0 against vectors from the reference's own source."""
    from caretta_amd import helper, multiple_alignment as ma
    g = golden("f6_extras.npz")
    for key in "abcd":
        gapped = bytes(g[f"a2n_{key}_in"]).decode()
        assert np.array_equal(ma.alignment_to_numpy({key: gapped})[key], g[f"a2n_{key}_out"])
    assert len(ma.alignment_to_numpy({"e": ""})["e"]) == 0
    np.testing.assert_allclose(helper.nb_std_axis_0(g["std_x"]), g["std_out"], rtol=1e-14)
    np.testing.assert_allclose(helper.normalize(g["norm_x"]), g["norm_out"], rtol=0, atol=1e-15)


def test_make_score_matrix_applies_a_plugin_score_function_on_the_host():
    """score_functions.py:48-50: any callable is applied cell by cell (a third-party SequenceBase plugin's own code)."""
    from caretta_amd import score_functions as sf
    a = np.arange(6.0).reshape(3, 2)
    b = np.arange(8.0).reshape(4, 2) / 3
    got = sf.make_score_matrix(a, b, lambda u, v, gamma: gamma * float(np.abs(u - v).sum()), 0.5)
    want = np.array([[0.5 * np.abs(u - v).sum() for v in b] for u in a])
    assert got.shape == (3, 4) and np.array_equal(got, want)
    with pytest.raises(TypeError):
        sf.make_score_matrix(a, b, "not callable", 0.5)


def test_formats_match_the_reference_writers(tmp_path):
    """tests/golden/f7_formats.npz holds the text the reference's own writers produced (helper.write_distance_matrix,
    helper.py:183-202; MultipleAlignment.write_alignment / to_sequence_alignment, multiple_alignment.py:287-309) and what
    its reader returned (helper.read_distance_matrix, :205-229): the product's writers byte for byte, its reader value
    for value."""
    from caretta_amd import helper
    from caretta_amd import multiple_alignment as ma
    g = np.load(Path(__file__).resolve().parent / "golden" / "f7_formats.npz", allow_pickle=False)
    names = [str(x) for x in g["matrix_names"]]
    path = tmp_path / "matrix.mat"
    helper.write_distance_matrix(names, g["matrix_D"], path)
    assert path.read_bytes() == bytes(g["matrix_text"])
    back_names, back = helper.read_distance_matrix(path)
    assert back_names == [str(x) for x in g["matrix_read_names"]]
    assert np.array_equal(back, g["matrix_read_D"])
    offsets, coords, tensors = g["fasta_fam_offsets"], g["fasta_fam_coords"], g["fasta_fam_tensors"]
    prots = [ma.Protein(str(name), tensors[offsets[k]:offsets[k + 1]], coords[offsets[k]:offsets[k + 1]], str(seq))
             for k, (name, seq) in enumerate(zip(g["fasta_names"], g["fasta_sequences"]))]
    msa = ma.MultipleAlignment(prots)
    aln = {p.name: g["fasta_alignment"][k] for k, p in enumerate(prots)}
    fasta = tmp_path / "aln.fasta"
    msa.write_alignment(fasta, aln)
    assert fasta.read_bytes() == bytes(g["fasta_text"])
    rows = msa.to_sequence_alignment(aln)
    assert [rows[p.name] for p in prots] == [str(x) for x in g["fasta_rows"]]
    # and back: gapped sequences -> index rows
    again = ma.alignment_to_numpy(rows)
    for k, p in enumerate(prots):
        assert np.array_equal(again[p.name], g["fasta_alignment"][k])


def test_partition_pairs_is_the_deal_of_distributed_py(lib):
    """cr_partition_pairs (the deal the single-process multi-GPU path makes, cr_multi_pairwise_scores) = the deal of
    caretta_amd.distributed.partition_pairs (one process per GPU), ragged and equal lengths, every world size."""
    from caretta_amd import distributed as cdist
    from caretta_amd import engine
    rng = np.random.default_rng(3)
    for lengths in (rng.integers(20, 400, size=23), np.full(17, 300), rng.integers(5, 9, size=40), np.array([7, 9])):
        pairs = engine.all_pairs(len(lengths))
        for world in (1, 2, 3, 4, 8):
            got = [engine.partition_pairs(lengths, world, r) for r in range(world)]
            for r in range(world):
                assert np.array_equal(got[r], cdist.partition_pairs(pairs, lengths, world, r))
            assert np.array_equal(np.sort(np.concatenate(got)), np.arange(len(pairs)))
            assert max(len(g) for g in got) <= cdist.shard_size(len(pairs), world)


def test_multi_device_needs_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from caretta_amd import _capi, engine
    assert engine.multi_device() is None
    with pytest.raises(_capi.CarettaHipError):
        engine.MultiDevice([0])


# ------------------------------------------------------------------- the layout table and the size classes, host only
def _equal_family(rows, npairs):
    """(offsets, pairs) of an equal-length family with `npairs` pairs (both orientations), as test_path_selection_boundaries builds it."""
    num = 2
    while num * (num - 1) < npairs:
        num += 1
    offsets = np.arange(num + 1, dtype=np.int64) * rows
    i, j = np.triu_indices(num, 1)
    fwd = np.stack([i, j], 1).astype(np.int32)
    return offsets, np.vstack([fwd, fwd[:, ::-1]])[:npairs]


@pytest.mark.parametrize("npairs,rows,expect", [
    (256, 193, "trio"), (257, 257, "trio"), (257, 64, "staged"), (257, 65, "trio"), (160, 300, "staged"), (161, 300, "trio"),
    (160, 250, "staged"), (161, 250, "trio"), (64, 150, "staged"), (65, 150, "trio"), (1300, 300, "trio"), (1301, 300, "single"),
    (257, 321, "duo"), (256, 321, "wide"), (1024, 360, "duo"), (1025, 360, "single"), (170, 330, "staged"), (171, 330, "wide"),
    (300, 1200, "duo"), (1024, 900, "duo"), (1025, 900, "single"), (252, 1200, "wide"), (504, 1200, "duo"), (2016, 1200, "single"),
    (8128, 300, "single"), (496, 150, "trio"), (1, 300, "staged"), (3, 2049, "single"), (3, 2048, "staged")])
def test_layout_table_boundaries_on_the_host(lib, npairs, rows, expect):
    """kLayoutTable / choose_layout (cr_layout.h) through cr_plan_layout, which needs no device: the pair-count and row-count limits
    at their boundary values (the GPU twin, tests/test_gpu_midsize.py::test_path_selection_boundaries, also runs both sides)."""
    from caretta_amd import engine
    offsets, pairs = _equal_family(rows, npairs)
    parts, cls = engine.plan_layout(offsets, 10, pairs)
    assert len(parts) == 1 and parts[0][0] == expect and parts[0][4] == npairs, (npairs, rows, parts)
    assert not cls.any()


def test_layout_table_tensor_widths_on_the_host(lib):
    from caretta_amd import engine
    offsets, pairs = _equal_family(300, 316)
    assert engine.plan_layout(offsets, 10, pairs)[0][0][0] == "trio"
    assert engine.plan_layout(offsets, 12, pairs)[0][0][0] == "trio"             # (k_pair_trio: score waves for widths 12 and 16 since round 6)
    assert engine.plan_layout(offsets, 16, pairs)[0][0][0] == "trio"
    assert engine.plan_layout(offsets, 17, pairs)[0][0][0] == "single"           # (no one-workgroup layout above width 16)
    with pytest.raises(ValueError):
        engine.plan_layout(offsets, 33, pairs)


def test_size_classes_on_the_host(lib, monkeypatch):
    """A ragged list is up to three lists (cr_batch_set_pairs): 20 domains of 150 residues with chains of 600, 600 and 1 300
    in it; a ragged family whose classes would all run one wave per pair stays one list; CARETTA_CLASSES=0 switches the split off
    (the switches are read once: cr_config_reload)."""
    from caretta_amd import engine
    lengths = np.array([150] * 20 + [600, 600, 1300], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)])
    i, j = np.triu_indices(len(lengths), 1)
    fwd = np.stack([i, j], 1).astype(np.int32)
    pairs = np.vstack([fwd, fwd[:, ::-1]])
    parts, cls = engine.plan_layout(offsets, 10, pairs)
    assert [p[0] for p in parts] == ["trio", "staged"] and [p[4] for p in parts] == [420, 86], parts
    n, m = lengths[pairs[:, 0]], lengths[pairs[:, 1]]
    assert np.array_equal(cls, np.where((n <= 320) & (m <= 1280), 0, 1))
    monkeypatch.setenv("CARETTA_CLASSES", "0")
    parts, cls = engine.plan_layout(offsets, 10, pairs)
    assert len(parts) == 1 and parts[0][4] == len(pairs) and not cls.any()
    monkeypatch.delenv("CARETTA_CLASSES")
    assert len(engine.plan_layout(offsets, 10, pairs)[0]) == 2
    # 160 structures of 80 .. 520 residues, 12 720 pairs: more than the 4 096 pairs a split is made for
    rng = np.random.default_rng(3)
    lengths = rng.integers(80, 521, size=160).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)])
    i, j = np.triu_indices(160, 1)
    parts, _ = engine.plan_layout(offsets, 10, np.stack([i, j], 1).astype(np.int32))
    assert len(parts) == 1 and parts[0][0] == "single"
    # a pair index out of range is an argument error, as in cr_batch_set_pairs
    with pytest.raises(ValueError):
        engine.plan_layout(offsets, 10, np.array([[0, 160]], np.int32))
