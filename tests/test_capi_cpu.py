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
