"""GPU parity tests of the mid-size pair-list layout (caretta_amd/csrc/cr_duo.h: one small workgroup per pair, one wave per
strip, strips paced by LDS progress words) and of the path-selection boundaries of cr_batch_set_pairs.  Everything is compared
with the C oracle bit for bit, through the C ABI."""
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


def layout_of(batch):
    """(kernel family, rows per lane A, rows per lane B, strips with A) the library chose for the batch's pair list."""
    return batch.layout()


def run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, expect, sw_gaps=(0.0,), threads=8, parts=None):
    """Full pipeline, matrix entries only and the streamed run of one pair list against the oracle; `expect`: the kernel
    family cr_batch_set_pairs must have chosen."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    expect = (expect,) if isinstance(expect, str) else tuple(expect)
    assert layout_of(batch)[0] in expect, f"layout {layout_of(batch)}, expected one of {expect}"
    if parts is not None:                                         # the families of the size classes of a split list
        assert [x[0] for x in batch.part_layouts()] == list(parts), batch.part_layouts()
        assert sum(x[4] for x in batch.part_layouts()) == len(pairs)
    for gap in sw_gaps:
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=threads)
        prm = engine.make_params(sw_gap=gap)
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
    batch.close()


def test_one_of_eight_share_of_the_headline(ctx, oracle, monkeypatch):
    """BASELINE config 3 sharded over 8 GPUs: every 8th of the 8 128 pairs of 128 x 300 (1 016 pairs) is what ONE GPU runs
    (north star: ">= 6x further scaling at 8 GPUs").  That list runs three waves per pair -- one of recurrences, two of
    scores (k_pair_trio) --; all 1 016 pairs, every output, against the oracle; and the same list on the row-split layout
    (k_pair_duo: CARETTA_TRIO=0)."""
    from caretta_amd import engine
    fam = synthetic.make_family(128, 300, seed=20242)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(128)[::8]
    assert len(pairs) == 1016
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "trio", sw_gaps=(0.0, 0.05))
    monkeypatch.setenv("CARETTA_TRIO", "0")
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "duo")


@pytest.mark.parametrize("dim,seed", [(10, 9101), (4, 9102), (16, 9103), (7, 9104)])
def test_midsize_ragged_lists_vs_oracle(ctx, oracle, monkeypatch, dim, seed):
    """Ragged lengths 40 .. 330 rows, both orientations of every pair (290 .. 400 pairs: beyond the one-pair-per-CU layouts):
    pairs of one launch end in strip 0 or strip 1, some have fewer rows than one strip; tensor widths that are padded (7)
    or not; with a Smith-Waterman gap the list is laid out again for the kernels that have a skewed seed sweep."""
    from caretta_amd import engine
    fam = synthetic.make_family(15, 330, dim=dim, seed=seed, clades=3)
    cuts = [330, 300, 257, 320, 321, 193, 192, 64, 65, 40, 288, 129, 310, 256, 191]
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(15)
    pairs = np.vstack([fwd, fwd[:, ::-1], fwd[::2]])             # 105 + 105 + 53 = 263 > 256
    # as ONE list (the row split: the longest structure has 330 rows) ...
    monkeypatch.setenv("CARETTA_CLASSES", "0")
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "duo", sw_gaps=(0.0, 0.05) if dim == 10 else (0.0,))
    # ... and as the library lays it out: two size classes (rows <= 320: one strip, by function -- no instance for widths
    # above 10: staged scores or one wave per pair --; the 321- and 330-row structures' pairs: staged scores)
    monkeypatch.delenv("CARETTA_CLASSES")
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "classes", sw_gaps=(0.0, 0.05) if dim == 10 else (0.0,),
                 parts=("trio", "staged") if dim <= 10 else None)


@pytest.mark.parametrize("dim,seed,waves,longest", [(10, 9111, None, 320), (4, 9112, "5", 320), (8, 9113, "2", 320), (7, 9114, "4", 320),
                                                    (10, 9115, None, 256), (8, 9116, "3", 255), (10, 9117, None, 192), (4, 9118, "5", 150),
                                                    (10, 9119, None, 128), (7, 9120, "2", 100)])
def test_midsize_lists_one_strip_by_function(ctx, oracle, monkeypatch, dim, seed, waves, longest):
    """k_pair_trio (cr_trio.h): lists of more than 256 pairs whose longest structure has 65 .. 320 rows (one strip of two to
    five rows per lane) -- one wave of recurrences and one to four waves of scores per pair (the library's choice, or
    CARETTA_TRIO_WAVES).  Ragged lengths 1 .. 320 (fewer rows than lanes, fewer columns than a batch or than the ring), both
    orientations, widths that are padded or not; with a Smith-Waterman gap the same layout runs the single-wave kernels."""
    from caretta_amd import engine
    fam = synthetic.make_family(15, 320, dim=dim, seed=seed, clades=3)
    cuts = [min(c, longest) for c in [320, 300, 257, 319, 1, 193, 192, 64, 65, 3, 288, 129, 7, 256, 9]]
    cuts[0] = longest
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(15)
    pairs = np.vstack([fwd, fwd[:, ::-1], fwd[::2]])             # 263 pairs
    if waves:
        monkeypatch.setenv("CARETTA_TRIO_WAVES", waves)
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "trio", sw_gaps=(0.0, 0.05) if dim == 10 else (0.0,))


def test_few_pairs_by_function_and_again_with_a_gap(ctx, oracle):
    """Lists the split by function takes from the one-pair-per-CU layouts (more than 160 / 160 / 64 pairs, at most 256): 182 pairs of
    up to 300 rows run on k_pair_trio; with a Smith-Waterman gap the SAME batch object is laid out again for the layouts that
    have a skewed seed sweep (cr_batch_run), and goes on giving the oracle's results -- also back at gap 0."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(14, 300, seed=9301, ragged=True, clades=2)
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(14)
    pairs = np.vstack([fwd, fwd[:, ::-1]])                       # 182 pairs
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    assert layout_of(batch)[0] == "trio", layout_of(batch)
    for gap in (0.0, 0.05, 0.0):
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=gap), nthreads=8)
        batch.run(engine.make_params(sw_gap=gap))
        res, aln = batch.fetch()
        assert_bit_identical(res, aln, ref, ref_aln)
        if gap != 0.0:
            assert layout_of(batch)[0] != "trio", layout_of(batch)
    batch.close()


@pytest.mark.parametrize("family", ["trio_few", "duo"])
def test_streamed_run_with_a_gap_first_on_a_fresh_batch(ctx, oracle, monkeypatch, family):
    """The gap-driven re-layout of a duo / few-pair trio list happens inside the FIRST run.  When that first run is the
    streamed one (the alignment kernel writes into the caller's page-locked arrays through the batch's order map), the map
    must be the one of the NEW layout: the list below is already sorted by cells (so the gap-0 layout keeps the caller's
    order and has no map at all) but spans several rows-per-lane groups (so the layout for the gap reorders it)."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    rows = [300, 290, 260, 230, 200, 170, 150, 120, 100, 90, 80, 70] if family == "trio_few" else \
           [500, 480, 450, 400, 360, 330, 300, 250, 200, 150, 100, 90, 80, 70, 66, 65, 50, 40]
    fam = synthetic.make_family(len(rows), max(rows), seed=9501 + len(rows), clades=2)
    for s, cut in zip(fam, rows):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(len(rows))
    pairs = np.vstack([fwd, fwd[:, ::-1]])
    if family == "trio_few":
        pairs = np.vstack([pairs, fwd[:60]])                     # 192 pairs: more than 160, at most 256
    cells = np.diff(offsets)[pairs[:, 0]] * np.diff(offsets)[pairs[:, 1]]
    pairs = np.ascontiguousarray(pairs[np.argsort(-cells, kind="stable")])
    monkeypatch.setenv("CARETTA_CLASSES", "0")                   # (ONE list: the re-layout of the whole list is what is tested)
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    assert layout_of(batch)[0] == ("trio" if family == "trio_few" else "duo"), layout_of(batch)
    prm = engine.make_params(sw_gap=0.05)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(sw_gap=0.05), nthreads=8)
    res_s, aln_s = batch.run_streamed(prm)                       # the first run of the batch
    ctx.synchronize()
    assert layout_of(batch)[0] not in ("trio", "duo"), layout_of(batch)
    for key in ("sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm", "aln_len", "seed_len", "flags"):
        assert np.array_equal(res_s[key], ref[key]), f"{key}: streamed results landed on the wrong pairs"
    for p in range(len(pairs)):
        ln = int(ref["aln_len"][p])
        assert np.array_equal(aln_s[p, :, :ln], ref_aln[p, :, :ln]), f"pair {p}: alignment differs"
    batch.close()


def test_midsize_three_and_more_strips(ctx, oracle, monkeypatch):
    """321 .. 600 rows: three to five waves per pair (3 rows per lane in strip 0, 2 in the others), ragged."""
    from caretta_amd import engine
    fam = synthetic.make_family(18, 600, seed=9201, clades=2)
    cuts = [600, 450, 321, 577, 448, 449, 320, 576, 300, 130, 512, 360, 333, 599, 64, 400, 585, 470]
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(18)
    pairs = np.vstack([fwd, fwd[:, ::-1]])                      # 306 pairs
    monkeypatch.setenv("CARETTA_CLASSES", "0")                  # one list: the row split with three to five waves per pair
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "duo")
    monkeypatch.delenv("CARETTA_CLASSES")                       # size classes: up to 320 rows / 321 .. 600 rows
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "classes")


def test_size_classes_of_a_mixed_list(ctx, oracle):
    """A family of 150-residue domains with two 600-residue chains in it (and one of 1 300): the list is split into size
    classes by rows, each laid out as a list of its own (cr_batch_set_pairs) -- full pipeline, matrix entries, the streamed
    run and a Smith-Waterman gap (which lays the few-pair classes out again, each keeping its place in the caller's order)
    against the oracle; the flexible=True matrix entries through the same classes."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_mixed_family(20, 150, 3, 1300, seed=9601)
    for s, cut in zip(fam[20:], (600, 600, 1300)):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(len(fam))
    pairs = np.vstack([fwd, fwd[:, ::-1]])                       # 506 pairs: rows 150 (440, 20 of them with 1 300 columns), 600 (44), 1 300 (22)
    run_all_ways(ctx, oracle, coords, tensors, offsets, pairs, "classes", sw_gaps=(0.0, 0.05), parts=("trio", "staged"))
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    batch.run(engine.make_params(gamma_tensor=1.3), scores_only=True, flexible=True)
    sw, _ = batch.fetch_scores()
    batch.close()
    for p in (0, 19, 21, 250, 505):
        i, j = pairs[p]
        s_ij = oracle.make_score_matrix(fam[i].tensors, fam[j].tensors, 1.3)
        assert sw[p] == oracle.smith_waterman_score(np.arange(len(fam[i].tensors)), np.arange(len(fam[j].tensors)), s_ij, 0.0)


@pytest.mark.parametrize("npairs,rows,expect", [
    (256, 193, ("trio",)), (257, 193, ("trio",)), (257, 256, ("trio",)), (257, 257, ("trio",)), (256, 257, ("trio",)),
    (257, 64, ("single", "staged")), (257, 65, ("trio",)), (300, 128, ("trio",)), (300, 129, ("trio",)),
    (160, 300, ("wide", "staged")), (161, 300, ("trio",)), (160, 250, ("wide", "staged")), (161, 250, ("trio",)), (64, 150, ("single", "staged")), (65, 150, ("trio",)),
    (700, 320, ("trio",)), (701, 320, ("trio",)), (257, 321, ("duo",)), (256, 321, ("wide", "staged")), (1300, 300, ("trio",)), (1301, 300, ("single",)), (1024, 360, ("duo",)),
    (1025, 360, ("single",)),
    (256, 64, ("single", "staged")), (170, 330, ("staged",)), (171, 330, ("wide",)),
    # long chains (seven or eight strips): the row split in up to two rounds of 512 resident pairs
    (300, 1200, ("duo",)), (1024, 900, ("duo",)), (1025, 900, ("single",))])
def test_path_selection_boundaries(ctx, oracle, npairs, rows, expect):
    """The pair-count and row-count limits of cr_batch_set_pairs at their boundary values: which kernel family runs on either
    side, and that both sides give the oracle's results (a sample of the pairs is compared: the lists differ by one pair)."""
    from caretta_amd import engine
    num = 2
    while num * (num - 1) < npairs:
        num += 1
    fam = synthetic.make_family(num, rows, seed=7000 + npairs + rows, clades=3)
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(num)
    pairs = np.vstack([fwd, fwd[:, ::-1]])[:npairs]
    assert len(pairs) == npairs
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    assert layout_of(batch)[0] in expect, f"{npairs} pairs of {rows}: layout {layout_of(batch)}, expected one of {expect}"
    batch.run()
    res, aln = batch.fetch()
    batch.close()
    sample = np.unique(np.concatenate([np.arange(0, npairs, max(1, npairs // 40)), [npairs - 1]]))
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs[sample], nthreads=8)
    assert_bit_identical(res[sample], aln[sample], ref, ref_aln)


@pytest.mark.parametrize("rows", [512, 513, 1024, 1025, 2048, 2049])
def test_row_limits_of_the_staged_sweeps(ctx, oracle, rows):
    """One, two, three / four rows per lane of the staged sweeps change at 512 and 1024 rows, the staged path ends at 2048:
    three pairs on either side of each limit."""
    from caretta_amd import engine
    fam = synthetic.make_family(3, rows, seed=7100 + rows, clades=1)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(3)
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    fam_name, ra, rb, na = layout_of(batch)
    # (beyond 2048 rows: the wide layout while the columns of the widest sweep fit the LDS -- 10 features x 2049 columns do
    # not --, else one wave per pair)
    assert fam_name == "staged" if rows <= 2048 else fam_name in ("wide", "single"), (rows, fam_name)
    if rows <= 2048:
        assert ra == (1 if rows <= 512 else 2 if rows <= 1024 else 3 if rows <= 1536 else 4), (rows, ra)
    batch.run()
    res, aln = batch.fetch()
    batch.close()
    assert_bit_identical(res, aln, *oracle.pairwise_batch(coords, tensors, offsets, pairs, nthreads=3))


@pytest.mark.parametrize("go,ge,gap", [(1.0, 0.01, 0.0), (0.0, 0.0, 0.0), (0.0, 0.5, 0.0), (3.0, 0.0, 0.1), (-0.5, 0.01, 0.0), (1.0, -0.01, 0.0)])
def test_staged_sweeps_ramps_without_masks(ctx, oracle, go, ge, gap):
    """The sweeps on staged scores run the ramps of their strips without EXEC masks (cr_sweep_wide.h, sweep_staged): staged zeros
    outside [0, m), a fixed point before a lane's column 0 that needs non-negative penalties, the last block of every strip
    masked.  Short lists (the staged family) whose ramps are most of the sweep: fewer columns than a wave has lanes, one to six
    strips, single rows; penalties of 0 (the fixed point with equal candidates), a Smith-Waterman gap (the seed keeps its
    masks) and NEGATIVE penalties (every mask back) -- all against the oracle."""
    from caretta_amd import engine
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(9, 330, seed=9301, clades=2)
    cuts = [330, 5, 40, 64, 65, 1, 129, 200, 321]
    for s, cut in zip(fam, cuts):
        s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
    coords, tensors, offsets = synthetic.pack(fam)
    fwd = engine.all_pairs(9)
    pairs = np.vstack([fwd, fwd[:, ::-1]])
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    assert all(x[0] == "staged" for x in batch.part_layouts()) or layout_of(batch)[0] == "staged", (layout_of(batch), batch.part_layouts())
    prm = dict(gap_open=go, gap_extend=ge, sw_gap=gap)
    batch.run(engine.make_params(**prm))
    res, aln = batch.fetch()
    batch.close()
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(**prm), nthreads=8)
    assert_bit_identical(res, aln, ref, ref_aln)


@pytest.mark.parametrize("dim", [16, 17])
def test_tensor_width_limit_of_the_one_workgroup_layouts(ctx, oracle, dim):
    """Widths up to 16 have wide / mid-size instances (the split by function included, since round 6), 17 and more run one
    wave per pair (or four-wave teams)."""
    from caretta_amd import engine
    fam = synthetic.make_family(24, 300, dim=dim, seed=7200 + dim, clades=2)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = np.vstack([engine.all_pairs(24), engine.all_pairs(24)[:40, ::-1]])       # 316 pairs
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    assert layout_of(batch)[0] == ("trio" if dim <= 16 else "single")
    batch.run()
    res, aln = batch.fetch()
    batch.close()
    assert_bit_identical(res, aln, *oracle.pairwise_batch(coords, tensors, offsets, pairs, nthreads=8))


def test_kabsch_degenerate_against_lapack(golden):
    """The device's paired_svd_superpose on the reference's own outputs for collinear / coincident / planar positions
    (tests/golden/f8_kabsch_degenerate.npz, LAPACK dgesdd): same landing positions and RMSD; R itself is only determined
    for the planar and coincident classes (tests/test_oracle_golden.py records the distances)."""
    from caretta_amd import superposition_functions as sup
    from test_oracle_golden import check_degenerate_kabsch_against_lapack
    check_degenerate_kabsch_against_lapack(golden("f8_kabsch_degenerate.npz"), sup.paired_svd_superpose)


# ------------------------------------------------------------------------------- the two paths that used to run pair by pair
def test_flexible_pairwise_matrix_runs_batched(oracle, golden):
    """flexible=True (multiple_alignment.py:323-326): the P x P matrix is smith_waterman_score of the TENSOR score matrices
    alone -- one launch over the pair list (cr_batch_run_tensor_scores) -- against the reference's own matrices
    (f9_flexible.npz), against the oracle bit for bit, for one and for several strips per pair and tensor widths that are
    padded, and against the per-pair calls of the score function + smith_waterman_score."""
    from caretta_amd import dynamic_time_warping as dtw, multiple_alignment as ma
    from test_oracle_golden import flexible_reference
    g = golden("f9_flexible.npz")
    for tag in ("FA", "FB"):
        offs, tens, xyz = g[f"fam{tag}_offsets"], g[f"fam{tag}_tensors"], g[f"fam{tag}_coords"]
        prots = [ma.Protein(f"s{k}", tens[offs[k]:offs[k + 1]], xyz[offs[k]:offs[k + 1]]) for k in range(len(offs) - 1)]
        m = ma.MultipleAlignment(prots).make_pairwise_matrix(dict(flexible=True, gamma_tensor=7.0))
        np.testing.assert_allclose(m, g[f"fam{tag}_M"], rtol=1e-9, atol=1e-12)
        assert np.array_equal(m, flexible_reference(oracle, g, tag))
        # tensors only (what mean_function(flexible=True) leaves of a node): same matrix
        bare = [ma.Protein(p.name, p.tensors) for p in prots]
        assert np.array_equal(ma.MultipleAlignment(bare).make_pairwise_matrix(dict(flexible=True, gamma_tensor=7.0)), m)
    for dim, lengths, seed in [(7, [700, 40, 333, 321, 64], 9301), (16, [150] * 7, 9302), (24, [90, 500, 257], 9303)]:
        fam = synthetic.make_family(len(lengths), max(lengths), dim=dim, seed=seed, ragged=True, clades=2)
        for s, cut in zip(fam, lengths):
            s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
        prots = [ma.Protein(s.name, s.tensors, s.coordinates) for s in fam]
        m = ma.MultipleAlignment(prots).make_pairwise_matrix(dict(flexible=True, gamma_tensor=1.3))
        for i in range(len(prots) - 1):
            for j in range(i + 1, len(prots)):
                s_ij = oracle.make_score_matrix(prots[i].tensors, prots[j].tensors, 1.3)
                assert m[i, j] == m[j, i] == oracle.smith_waterman_score(np.arange(len(prots[i])), np.arange(len(prots[j])), s_ij, 0.0)
        s01 = prots[0].score_function(prots[1], flexible=True, gamma_tensor=1.3)        # the per-pair route of the product
        assert m[0, 1] == dtw.smith_waterman_score(np.arange(s01.shape[0]), np.arange(s01.shape[1]), s01)


def test_smith_waterman_batch_vs_oracle(oracle, golden):
    """smith_waterman WITH its traceback over a list (cr_smith_waterman_batch; dynamic_time_warping.py:226-278): ragged
    problems, alphabet mode, windows of larger matrices, negative scores, gap 0 and 0.25, the streaming kernels (contiguous
    columns everywhere) and the tile kernels; the reference's own alignment on a flexible score matrix; an all-zero matrix
    raises the reference's TypeError."""
    from caretta_amd import dynamic_time_warping as dtw
    from test_gpu_parity import _explicit_problems
    rng = np.random.default_rng(9401)
    mixed = [p for p in _explicit_problems(rng) if not np.any(np.asarray(p[1]) < 0)]
    ident = [(np.arange(n), np.arange(m), rng.uniform(size=(n, m)) ** 2 - 0.15) for n, m in [(300, 300), (64, 65), (1, 9), (257, 130), (90, 400), (513, 77)]]
    small = [(np.arange(n), np.arange(m), rng.uniform(size=(n, m)) ** 2 - 0.15) for n, m in [(64, 65), (1, 9), (90, 40), (130, 77)]]
    for problems in (mixed, ident, small * 800):                 # (3 200 problems: one row per lane in the streaming kernels)
        for gap in (0.0, 0.25):
            got = dtw.smith_waterman_batch(problems, gap)
            for (s1, s2, mat), (a1, a2, score) in zip(problems[:len(mixed) + len(ident)], got):
                r1, r2, rs, none = oracle.smith_waterman(s1, s2, mat, gap)
                assert not none and np.array_equal(a1, r1) and np.array_equal(a2, r2) and score == rs
    g = golden("f9_flexible.npz")
    s = g["famFA_S01"]
    (a1, a2, score), = dtw.smith_waterman_batch([(np.arange(s.shape[0]), np.arange(s.shape[1]), s)], 0.0)
    assert np.array_equal(a1, g["famFA_sw_aln1"]) and np.array_equal(a2, g["famFA_sw_aln2"])
    assert abs(score - float(g["famFA_sw_score"])) <= 1e-12 * abs(score)
    with pytest.raises(TypeError):
        dtw.smith_waterman_batch([(np.arange(4), np.arange(5), np.ones((4, 5))), (np.arange(3), np.arange(3), -np.ones((3, 3)))], 0.0)


def test_smith_waterman_gap0_traceback_on_the_row_sweep(oracle, monkeypatch):
    """smith_waterman's default call (gap 0, dynamic_time_warping.py:226-278; multiple_alignment.py:332-334) over a list runs
    fill, decisions, first maximum and walk in ONE launch of the row sweep (k_sw_trace_rows: a lane owns columns, the walk reads
    the transposed blocks).  Tie-heavy matrices (constant, 0/1, small integers: the diag -> left -> up priority and the
    row-major FIRST maximum decide every step), column strips (m > 64 x 8), single rows / columns, zero borders, negative
    scores -- against the oracle, and against the skewed sweep of the gap != 0 lists (CARETTA_NO_SW_ROWS=1)."""
    from caretta_amd import dynamic_time_warping as dtw, engine
    rng = np.random.default_rng(9501)
    ar = np.arange
    problems = []
    for n, m in [(1, 1), (1, 70), (70, 1), (17, 17), (64, 64), (65, 320), (300, 300), (33, 513), (300, 1030), (129, 1500)]:
        problems.append((ar(n), ar(m), np.ones((n, m))))                                     # every cell ties
        problems.append((ar(n), ar(m), rng.integers(-1, 2, size=(n, m)).astype(np.float64)))   # -1 / 0 / 1
        problems.append((ar(n), ar(m), (rng.uniform(size=(n, m)) < 0.1).astype(np.float64)))   # sparse ones: long flat runs
        problems.append((ar(n), ar(m), rng.uniform(size=(n, m)) ** 2 - 0.3))
    z = np.zeros((40, 50))
    z[20:30, 10:20] = np.eye(10)                                                             # a maximum reached first in the interior
    problems.append((ar(40), ar(50), z))
    z2 = np.zeros((90, 700))
    z2[5, 600] = 2.0                                                                         # first maximum in the second column strip, early row
    z2[50, 3] = 2.0
    problems.append((ar(90), ar(700), z2))
    sub = rng.integers(-2, 3, size=(20, 20)).astype(np.float64)                              # alphabet mode (gathered columns), ties
    for n, m in [(40, 55), (200, 333), (7, 600)]:
        problems.append((rng.integers(0, 20, size=n), rng.integers(0, 20, size=m), sub))
    want = [oracle.smith_waterman(s1, s2, mat, 0.0) for s1, s2, mat in problems]
    keep = [k for k, w in enumerate(want) if not w[3]]                                       # (all-zero matrices raise: tested elsewhere)
    problems, want = [problems[k] for k in keep], [want[k] for k in keep]

    def check(got):
        for k, ((a1, a2, score), (r1, r2, rs, _none)) in enumerate(zip(got, want)):
            assert np.array_equal(a1, r1) and np.array_equal(a2, r2) and score == rs, (k, problems[k][2].shape)
    check(dtw.smith_waterman_batch(problems, 0.0))
    # many problems per launch (one per wave, several waves per CU walking while others stream)
    got = dtw.smith_waterman_batch(problems * 30, 0.0)
    check(got[:len(problems)])
    check(got[-len(problems):])
    monkeypatch.setenv("CARETTA_NO_SW_ROWS", "1")
    engine.reload_config()
    try:
        check(dtw.smith_waterman_batch(problems, 0.0))
    finally:
        monkeypatch.delenv("CARETTA_NO_SW_ROWS")
        engine.reload_config()


@pytest.mark.parametrize("num,length,seed,limit", [(512, 300, 20243, 1.5), (64, 1200, 20244, None)])
def test_multi_device_loopback_eight_shares(ctx, num, length, seed, limit, monkeypatch):
    """cr_multi_* with EIGHT shares on the one GPU of the box (loopback: the gather is device copies; the deal, the parked host
    threads, the kept batches and the scatter back to pair order are the product path) on BASELINE configs 4 and 5:
    bit-identical to one batch, a second layout through the same object and back.  Times are printed (config 4: eight
    shares on one device take 1.03 - 1.10 x the one-batch call, the repeated call -- kept layout: structures re-uploaded
    into the kept batches, no new pair lists -- 0.7 x the first); the assertions on them are deliberately loose (a test
    must not fail on a busy box): the repeated call at most 1.25 x the first, config 4 at most 1.5 x one batch."""
    import time
    from caretta_amd import engine
    monkeypatch.setenv("CARETTA_MULTI_ALLOW_DUPLICATES", "1")
    fam = synthetic.make_family(num, length, seed=seed)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    prm = engine.make_params()
    t_one = None
    for it in range(3):
        t0 = time.perf_counter()
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(prm, scores_only=True)
        sw_one, flags_one = batch.fetch_scores()
        batch.close()
        t_one = time.perf_counter() - t0
    multi = engine.MultiDevice([0] * 8)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        sw, flags = multi.pairwise_scores(coords, tensors, offsets, prm)
        times.append(time.perf_counter() - t0)
        assert np.array_equal(sw, sw_one) and np.array_equal(flags, flags_one)
    phases = multi.last_ms()
    assert all(p >= 0.0 for p in phases) and phases[0] > 0.0          # per phase, from events, without CARETTA_MULTI_TIMING
    # another layout through the same object, then back
    small = synthetic.make_family(9, 80, seed=5, ragged=True)
    c2, t2, o2 = synthetic.pack(small)
    b2 = engine.PairBatch(ctx, c2, t2, o2).set_pairs(engine.all_pairs(9))
    b2.run(prm, scores_only=True)
    sw2_ref, fl2_ref = b2.fetch_scores()
    b2.close()
    sw2, fl2 = multi.pairwise_scores(c2, t2, o2, prm)
    assert np.array_equal(sw2, sw2_ref) and np.array_equal(fl2, fl2_ref)
    sw, flags = multi.pairwise_scores(coords, tensors, offsets, prm)
    assert np.array_equal(sw, sw_one) and np.array_equal(flags, flags_one)
    multi.close()
    import torch
    assert torch.cuda.current_device() == 0
    print(f"{num} x {length}: one batch {t_one * 1e3:.1f} ms, eight shares on one device {min(times[1:]) * 1e3:.1f} ms (first call {times[0] * 1e3:.1f})")
    assert min(times[1:]) <= times[0] * 1.25
    if limit is not None:
        assert min(times[1:]) <= t_one * limit, (min(times[1:]), t_one)


@pytest.mark.parametrize("dim", [12, 16])
def test_by_function_layout_serves_tensor_widths_up_to_16(ctx, oracle, dim):
    """Protein accepts any (L, d) tensor array (multiple_alignment.py:312-319, :328-331).  Lists of 65 .. 320 rows with
    d = 11 .. 16 used to fall back to one wave per pair / the row split: k_pair_trio is instantiated for the padded widths 12
    and 16 (its score waves pad the STORED width, not the batch's).  One, three and five rows per lane, every output."""
    from caretta_amd import engine
    for num, length in ((13, 90), (24, 150), (20, 300)):
        fam = synthetic.make_family(num, length, dim=dim, seed=6100 + dim + length)
        coords, tensors, offsets = synthetic.pack(fam)
        run_all_ways(ctx, oracle, coords, tensors, offsets, engine.all_pairs(num), "trio")
    # a ragged list in the same regime (rows past n, unequal columns)
    fam = synthetic.make_ragged_family(16, 120, 250, dim=dim, seed=6200 + dim)
    coords, tensors, offsets = synthetic.pack(fam)
    run_all_ways(ctx, oracle, coords, tensors, offsets, engine.all_pairs(16), ("trio", "staged", "single"))


@pytest.mark.parametrize("dim", [33, 48, 200])
def test_tensors_wider_than_the_fused_kernels(oracle, dim):
    """d > 32 (multiple_alignment.py:312-331 takes any width): up to 192 the pairwise matrix, the score function, the
    two-structure alignment and the progressive alignment run on STAGED scores (the run-time-width staging kernel
    k_stage_tensor_any; the whole tree resident), beyond that through the per-function drop-ins -- make_score_matrix (any
    width) -> smith_waterman -> Kabsch on the seed -> make_score_matrix -> smith_waterman_score over the list / dtw_align --, never
    an error; the oracle's values bit for bit either way."""
    from caretta_amd import engine
    from caretta_amd import multiple_alignment as ma
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(6, 64, dim=dim, seed=7000 + dim, ragged=True)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(len(fam))
    prm = dict(gamma_tensor=7.0 / dim * 10, gamma_coords=0.03, verbose=False)
    oprm = default_params(gamma_tensor=prm["gamma_tensor"], gamma_coords=0.03)
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=oprm)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
    msa = ma.MultipleAlignment(prots)
    matrix = msa.make_pairwise_matrix(prm)
    assert np.array_equal(matrix, engine.assemble_matrix(pairs, ref["sw"], len(fam)))
    # the two-structure alignment (multiple_alignment.py:263-275): pipeline H's dtw_align of the pair
    two = ma.MultipleAlignment(prots[:2])
    aln = two.multiple_align(None, 1.0, 0.01, 1.0, 1.0, prm, dict(verbose=False))
    ln = int(ref["aln_len"][0])
    assert np.array_equal(aln[prots[0].name], ref_aln[0, 0, :ln]) and np.array_equal(aln[prots[1].name], ref_aln[0, 1, :ln])
    # the progressive alignment: every node replayed by the oracle on the walk's own children
    full = msa.multiple_align(matrix.max() - matrix, 1.0, 0.01, 1.0, 1.0, prm, dict(verbose=False))
    width = {len(v) for v in full.values()}
    assert len(width) == 1
    for s in fam:                                              # every residue exactly once, in order
        row = np.asarray(full[s.name])
        assert np.array_equal(row[row >= 0], np.arange(len(s.coordinates)))
    tree = np.asarray(msa.tree).astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]
    num = len(fam)
    sizes = [1] * num
    for k, (n1, n2) in enumerate(joins):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        w1, w2 = np.ravel(msa.final_consensus_weights[n1]), np.ravel(msa.final_consensus_weights[n2])
        _a1, _a2, xn, tn, wn, _f = oracle.progressive_node(s1.coordinates, s1.tensors, w1, s2.coordinates, s2.tensors, w2,
                                                          sizes[n2] / (2 * tot), sizes[n1] / (2 * tot), params=oprm)
        node = msa.final_sequences[num + k]
        assert np.array_equal(tn, node.tensors) and np.array_equal(np.ravel(wn), np.ravel(msa.final_consensus_weights[num + k]))
        if dim <= 192:
            assert np.array_equal(xn, node.coordinates)                      # (the resident tree: the device's own merge)
        else:
            assert np.allclose(xn, node.coordinates, rtol=0, atol=1e-9)      # (host mean_function: numpy's matmul in the frame change)
        sizes.append(tot)


@pytest.mark.parametrize("dim,num,length,gap", [(40, 30, 300, 0.0), (64, 9, 700, 0.0), (33, 14, 150, 0.05), (192, 5, 120, 0.0)])
def test_wide_tensor_lists_in_pieces_vs_oracle(ctx, oracle, dim, num, length, gap):
    """Tensors of 33 ... 192 features on the batched engine: the staged family with the run-time-width staging kernel, a list
    longer than its 1 024 strips handed over in pieces by `MultipleAlignment.pairwise` (30 x 300: 435 pairs x 5 strips = three
    pieces; 9 x 700: two rows per lane), every output of pipeline H against the oracle bit for bit; a Smith-Waterman gap; the
    C ABI itself refuses the list that is too long for one piece with an argument error."""
    from caretta_amd import engine
    from caretta_amd import multiple_alignment as ma
    from caretta_amd._capi import CarettaHipError
    from oracle.pyoracle import default_params
    fam = synthetic.make_family(num, length, dim=dim, seed=7300 + dim, ragged=(num % 2 == 1), clades=3)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    gt = 70.0 / dim
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, params=default_params(gamma_tensor=gt, sw_gap=gap), nthreads=8)
    if gap == 0.0:
        msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam])
        out = msa.pairwise(dict(gamma_tensor=gt, gamma_coords=0.03, verbose=False))
        assert_bit_identical(out.results, out.alignments, ref, ref_aln)
        sc = msa.pairwise(dict(gamma_tensor=gt, gamma_coords=0.03, verbose=False), scores_only=True, want_alignments=False)
        assert np.array_equal(sc.results["sw"], ref["sw"])
    # one piece through the engine directly (layout "staged"), with the gap
    piece = pairs[:min(len(pairs), 1024 // (-(-length // 64)) if length <= 512 else 40)]
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(piece)
    assert layout_of(batch)[0] == "staged"
    batch.run(engine.make_params(gamma_tensor=gt, sw_gap=gap))
    res, aln = batch.fetch()
    for p in range(len(piece)):
        ln = int(ref["aln_len"][p])
        assert int(res["aln_len"][p]) == ln and np.array_equal(aln[p, :, :ln], ref_aln[p, :, :ln])
    for key in ("sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm"):
        assert np.array_equal(res[key], ref[key][:len(piece)]), key
    if len(pairs) * (-(-length // 64)) > 1024:
        with pytest.raises((CarettaHipError, ValueError)):
            batch.set_pairs(pairs)
    batch.close()
