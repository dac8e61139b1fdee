#!/usr/bin/env python3
"""Randomised parity run on one MI355X: the HIP path against the C oracle (TEST infrastructure, oracle/) on many small
random batches -- ragged lengths from 1 to 1400 residues, tensor widths 1..32, all gap / gamma settings, the batched
pipeline on every kernel family (single-wave, team and wide kernels; column and skewed sweeps), the device-resident
progressive alignment, the explicit-score-matrix drop-ins and their batched forms.  Every output must be bit-identical.

    python tests/fuzz_parity.py [seconds] [seed]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from caretta_amd import engine, multiple_alignment as ma, neighbor_joining as nj, synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402


def random_family(rng):
    kind = rng.integers(0, 7)
    dim = int(rng.choice([1, 2, 3, 4, 5, 8, 10, 13, 16, 10, 10, 17, 24, 29, 32]))
    if kind == 5:      # a MIXED list (round 5: size classes of cr_batch_set_pairs): a family of short domains with a few long
        # chains in it -- the list is split by rows (<= 320 / <= 1 472 / longer), every class laid out as a list of its own
        short, long_len = int(rng.choice([70, 120, 150, 200, 300])), int(rng.choice([400, 600, 900, 1300, 1500]))
        fam = synthetic.make_mixed_family(int(rng.integers(6, 22)), short, int(rng.integers(1, 4)), long_len, dim=min(dim, 16),
                                          seed=int(rng.integers(1 << 30)))
        if rng.integers(0, 2):                                    # ... ragged on top
            for st in fam:
                cut = int(rng.integers(max(1, int(0.7 * len(st.coordinates))), len(st.coordinates) + 1))
                st.coordinates, st.tensors = st.coordinates[:cut].copy(), st.tensors[:cut].copy()
        return fam, min(dim, 16)
    if kind == 6:      # LONG chains, few structures with both orientations: the row split beyond 1 088 rows (round 5), the wide layout
        num, length = int(rng.integers(3, 7)), int(rng.choice([850, 900, 1100, 1200, 1400, 1472]))
        fam = synthetic.make_family(num, length, dim=min(dim, 16), seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(0, 2)), clades=2)
        return fam, min(dim, 16)
    if kind == 4:      # a MID-SIZE list: 70 .. 600 rows, 36 .. 380 pairs (cr_trio.h: one strip of two to five rows per lane split by
        # function, from 65 / 111 / 161 pairs on; cr_duo.h: two to five waves per pair beyond 320 rows and 256 pairs)
        num, length = int(rng.integers(9, 21)), int(rng.choice([70, 100, 128, 129, 150, 192, 193, 220, 256, 257, 300, 320, 321, 384, 450, 600]))
        fam = synthetic.make_family(num, length, dim=min(dim, 16), seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(0, 2)), clades=int(rng.integers(1, 4)))
        return fam, min(dim, 16)
    if kind == 0:      # related structures, ragged
        # up to 18 structures: with both orientations more than 128 pairs, i.e. the grouped single-wave kernels too
        num, length = int(rng.integers(2, 19)), int(rng.choice([12, 40, 90, 150, 200, 260, 330, 450, 700, 1000]))
        fam = synthetic.make_family(num, length, dim=dim, seed=int(rng.integers(1 << 30)), ragged=True, clades=int(rng.integers(1, 4)))
    elif kind == 1:    # unrelated random walks of very different lengths
        fam = []
        for k in range(int(rng.integers(2, 8))):
            ln = int(rng.choice([1, 2, 3, 4, 5, 7, 17, 63, 64, 65, 128, 129, 192, 193, 257, 320, 321, 400]))
            one = synthetic.make_family(1, ln, dim=dim, seed=int(rng.integers(1 << 30)), clades=1)[0]
            fam.append(one)
    elif kind == 2:    # tie-heavy: coordinates and tensors on a coarse grid
        num, length = int(rng.integers(2, 7)), int(rng.choice([20, 70, 140]))
        fam = synthetic.make_family(num, length, dim=dim, seed=int(rng.integers(1 << 30)), ragged=True, clades=1)
        for s in fam:
            s.coordinates[:] = np.round(s.coordinates / 4.0) * 4.0
            s.tensors[:] = np.round(s.tensors * 2.0) / 2.0
    elif kind == 3 and rng.integers(0, 2):    # a few long structures: the wide kernels (one wave per strip, up to 16 waves)
        num, length = int(rng.integers(2, 5)), int(rng.choice([400, 520, 770, 1000, 1400]))
        fam = synthetic.make_family(num, length, dim=min(dim, 16), seed=int(rng.integers(1 << 30)), ragged=True, clades=1)
        dim = min(dim, 16)
    else:              # far apart: RBF underflows, seeds of <= 3 positions
        num, length = int(rng.integers(2, 6)), int(rng.choice([5, 30, 100]))
        fam = synthetic.make_family(num, length, dim=dim, seed=int(rng.integers(1 << 30)), ragged=True, clades=num)
        for s in fam:
            s.tensors[:] = s.tensors * float(rng.choice([1.0, 3.0, 8.0]))
    return fam, dim


def check_batch(ctx, oracle, fam, rng):
    coords, tensors, offsets = synthetic.pack(fam)
    num = len(fam)
    pairs = engine.all_pairs(num)
    if rng.integers(0, 2):
        pairs = np.vstack([pairs, pairs[:, ::-1]])                 # both orientations
    prm = dict(gamma_tensor=float(rng.choice([7.0, 1.0, 0.3])), gamma_coords=float(rng.choice([0.03, 0.1, 1e-4])),
               gap_open=float(rng.choice([1.0, 0.0, 0.5, 3.0])), gap_extend=float(rng.choice([0.01, 0.0, 0.5])),
               sw_gap=float(rng.choice([0.0, 0.0, 0.0, 0.1])))
    if len(pairs) < 300 and max(len(st.coordinates) for st in fam) > 832 and rng.integers(0, 2):
        pairs = np.vstack([pairs] * int(np.ceil(300 / len(pairs))))    # (more than 256 pairs of long chains: k_pair_duo in rounds)
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    if rng.integers(0, 4) == 0:                                    # the streamed run FIRST (the order maps of a fresh batch)
        res, aln32 = batch.run_streamed(engine.make_params(**prm))
        ctx.synchronize()
        res, aln = res.copy(), aln32.astype(np.int64)
    else:
        batch.run(engine.make_params(**prm))
        res, aln = batch.fetch()
    if rng.integers(0, 3) == 0:                                    # the matrix entries alone (cr_batch_run_scores)
        batch.run(engine.make_params(**prm), scores_only=True)
        sw_only, flags_only = batch.fetch_scores()
        # (the scores-only run reports the seed conditions only: no alignment, hence no "fewer than 3 aligned positions")
        if not (np.array_equal(sw_only, res["sw"]) and np.array_equal(flags_only & 5, res["flags"] & 5)):
            raise AssertionError(f"scores-only run differs: params {prm}, lengths {np.diff(offsets)}")
    batch.close()
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, pyoracle.default_params(**prm), nthreads=8)
    for key in ("flags", "aln_len", "seed_len", "sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm", "R", "t"):
        if not np.array_equal(res[key], ref[key]):
            raise AssertionError(f"{key} differs: params {prm}, lengths {np.diff(offsets)}")
    for p in range(len(pairs)):
        ln = int(ref["aln_len"][p])
        if not np.array_equal(aln[p, :, :ln], ref_aln[p, :, :ln]):
            raise AssertionError(f"alignment of pair {pairs[p]} differs: params {prm}, lengths {np.diff(offsets)}")
    return len(pairs), res


def check_flexible_progressive(oracle, fam, rng):
    """flexible=True in score and mean function with every node on the device (cr_progressive_align_flexible) against the oracle's
    flexible node replayed on the device's own children."""
    num = len(fam)
    if num < 3 or min(len(s.coordinates) for s in fam) < 5 or fam[0].tensors.shape[1] > 32:
        return 0
    prots = [ma.Protein(s.name, s.tensors) for s in fam]
    msa = ma.MultipleAlignment(prots)
    gt = float(rng.choice([7.0, 1.0]))
    sf = dict(flexible=True, gamma_tensor=gt)
    m = msa.make_pairwise_matrix(sf)
    tree, _ = nj.neighbor_joining(m.max() - m)
    go, ge, cw, gw = float(rng.choice([1.0, 0.5])), float(rng.choice([0.01, 0.1])), float(rng.choice([1.0, 0.5])), float(rng.choice([1.0, 0.2]))
    msa.progressive_align(tree, go, ge, cw, gw, sf, dict(flexible=True))
    tree = np.asarray(tree).astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]
    sizes = [1] * num
    for k, (n1, n2) in enumerate(joins):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        _, _, tn, wn = oracle.progressive_node_flexible(s1.tensors, msa.final_consensus_weights[n1], s2.tensors, msa.final_consensus_weights[n2],
                                                        sizes[n2] / (2 * tot), sizes[n1] / (2 * tot), gt, gw, go, ge)
        node = msa.final_sequences[num + k]
        if not (np.array_equal(tn, node.tensors) and np.array_equal(wn, msa.final_consensus_weights[num + k])):
            raise AssertionError(f"flexible progressive node {k} differs: lengths {[len(p) for p in prots]}")
        sizes.append(tot)
    return len(joins)


def check_progressive(oracle, fam, rng):
    num = len(fam)
    if num < 3 or min(len(s.coordinates) for s in fam) < 5:
        return 0
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    try:
        m = msa.make_pairwise_matrix(prm)
    except TypeError:
        return 0                                                   # a pair without any positive local alignment
    tree, _ = nj.neighbor_joining(m.max() - m)
    go, ge, cw, gw = float(rng.choice([1.0, 0.5])), float(rng.choice([0.01, 0.1])), float(rng.choice([1.0, 0.5])), float(rng.choice([1.0, 0.2]))
    try:
        msa.progressive_align(tree, go, ge, cw, gw, prm, dict(flexible=False, verbose=False))
    except TypeError:
        return 0                                                   # a node without any positive local alignment
    tree = np.asarray(tree).astype(np.int64)
    joins = [(int(tree[x, 0]), int(tree[x + 1, 0])) for x in range(0, tree.shape[0] - 1, 2)] + [(int(tree[-1, 0]), int(tree[-1, 1]))]
    sizes = [1] * num
    oprm = pyoracle.default_params(gap_open=go, gap_extend=ge)
    for k, (n1, n2) in enumerate(joins):
        tot = sizes[n1] + sizes[n2]
        s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
        _, _, xn, tn, wn, _ = oracle.progressive_node(s1.coordinates, s1.tensors, msa.final_consensus_weights[n1], s2.coordinates,
                                                      s2.tensors, msa.final_consensus_weights[n2], sizes[n2] / (2 * tot),
                                                      sizes[n1] / (2 * tot), oprm, gw)
        node = msa.final_sequences[num + k]
        if not (np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors)
                and np.array_equal(wn, msa.final_consensus_weights[num + k])):
            raise AssertionError(f"progressive node {k} differs: lengths {[len(p) for p in prots]}")
        sizes.append(tot)
    return len(joins)


def check_dropins(oracle, rng):
    """dtw_align / smith_waterman(_score) on an explicit score matrix, incl. alphabet mode and tie-heavy matrices."""
    from caretta_amd import dynamic_time_warping as dtw
    n, m = (int(rng.choice([1, 2, 3, 7, 31, 64, 65, 127, 128, 129, 200, 257, 300, 390])) for _ in range(2))
    rows, cols = (n, m) if rng.integers(0, 2) else (int(rng.integers(1, 30)), int(rng.integers(1, 30)))   # alphabet mode
    kind = rng.integers(0, 4)
    if kind == 0:
        s = rng.uniform(0.0, 1.0, size=(rows, cols))
    elif kind == 1:
        s = np.round(rng.uniform(-1.0, 2.0, size=(rows, cols)))                    # ties, negative scores
    elif kind == 2:
        s = np.exp(-rng.uniform(0.0, 800.0, size=(rows, cols)))                    # underflow to 0
    else:
        s = np.ones((rows, cols)) * float(rng.choice([0.0, 1.0, 0.25]))            # constant
    seq1 = np.arange(n) if (rows, cols) == (n, m) else rng.integers(0, rows, size=n)
    seq2 = np.arange(m) if (rows, cols) == (n, m) else rng.integers(0, cols, size=m)
    go, ge = float(rng.choice([0.0, 1.0, 0.5, 3.0])), float(rng.choice([0.0, 0.01, 0.5]))
    a1, a2, sc = dtw.dtw_align(seq1, seq2, s, go, ge)
    b1, b2, sr = oracle.dtw_align(seq1, seq2, s, go, ge)
    if not (np.array_equal(a1, b1) and np.array_equal(a2, b2) and sc == sr and dtw.dtw_align_score(seq1, seq2, s, go, ge) == sr):
        raise AssertionError(f"dtw_align differs: n {n} m {m} kind {kind} gaps {go} {ge} matrix {s.shape}")
    gap = float(rng.choice([0.0, 0.0, 0.1, 1.0]))
    if dtw.smith_waterman_score(seq1, seq2, s, gap) != oracle.smith_waterman_score(seq1, seq2, s, gap):
        raise AssertionError(f"smith_waterman_score differs: n {n} m {m} kind {kind} gap {gap} matrix {s.shape}")
    b1, b2, sr, none = oracle.smith_waterman(seq1, seq2, s, gap)
    try:
        a1, a2, sc = dtw.smith_waterman(seq1, seq2, s, gap)
        ok = not none and np.array_equal(a1, b1) and np.array_equal(a2, b2) and sc == sr
    except TypeError:
        ok = bool(none)
    if not ok:
        raise AssertionError(f"smith_waterman differs: n {n} m {m} kind {kind} gap {gap} matrix {s.shape}")
    return 1


def check_explicit_batch(oracle, rng):
    """smith_waterman_score_batch / dtw_align_batch on a ragged list of matrices (row sweep, gathers, column strips)."""
    from caretta_amd import dynamic_time_warping as dtw
    problems = []
    for _ in range(int(rng.integers(1, 9))):
        n = int(rng.choice([1, 2, 9, 63, 64, 65, 150, 300, 513]))
        m = int(rng.choice([1, 3, 64, 65, 128, 129, 300, 321, 512, 513, 700, 1030]))
        kind = rng.integers(0, 4)
        if kind == 0:
            problems.append((np.arange(n), np.arange(m), rng.uniform(-0.3, 1.0, size=(n, m))))
        elif kind == 1:                                                             # ties
            problems.append((np.arange(n), np.arange(m), np.round(rng.uniform(-1.0, 2.0, size=(n, m)))))
        elif kind == 2:                                                             # alphabet mode
            sub = rng.normal(size=(int(rng.integers(1, 25)), int(rng.integers(1, 25))))
            problems.append((rng.integers(0, sub.shape[0], size=n), rng.integers(0, sub.shape[1], size=m), sub))
        else:                                                                       # a window of a larger matrix
            big = rng.uniform(size=(n + 5, m + 9)) ** 4
            problems.append((np.arange(3, n + 3), np.arange(7, m + 7), big))
    gap = float(rng.choice([0.0, 0.0, 0.0, 0.2]))
    got = dtw.smith_waterman_score_batch(problems, gap)
    want = np.array([oracle.smith_waterman_score(a, b, s, gap) for a, b, s in problems])
    if not np.array_equal(got, want):
        raise AssertionError(f"smith_waterman_score_batch differs: gap {gap}, shapes {[(len(a), len(b)) for a, b, _ in problems]}")
    if rng.integers(0, 2) == 0:
        # smith_waterman WITH its traceback over the list (gap 0: the row sweep with decisions, first maximum and walk in one
        # launch, round 6; else the skewed sweep with its walk); an all-zero matrix raises, as the reference does
        want_sw = [oracle.smith_waterman(a, b, s, gap) for a, b, s in problems]
        if any(w[3] for w in want_sw):
            try:
                dtw.smith_waterman_batch(problems, gap)
                raise AssertionError("smith_waterman_batch did not raise on a matrix without a positive cell")
            except TypeError:
                pass
        else:
            for (a, b, s), (a1, a2, sc), (o1, o2, osc, _none) in zip(problems, dtw.smith_waterman_batch(problems, gap), want_sw):
                if not (np.array_equal(a1, o1) and np.array_equal(a2, o2) and sc == osc):
                    raise AssertionError(f"smith_waterman_batch differs: n {len(a)} m {len(b)} gap {gap}")
    if rng.integers(0, 3) == 0:
        go, ge = float(rng.choice([0.0, 1.0, 0.5])), float(rng.choice([0.0, 0.01, 0.5]))
        for (a, b, s), (a1, a2, sc) in zip(problems, dtw.dtw_align_batch(problems, go, ge)):
            o1, o2, osc = oracle.dtw_align(a, b, s, go, ge)
            if not (np.array_equal(a1, o1) and np.array_equal(a2, o2) and sc == osc):
                raise AssertionError(f"dtw_align_batch differs: n {len(a)} m {len(b)} gaps {go} {ge}")
    return len(problems)


def check_sw_gap_stream(oracle, rng):
    """smith_waterman with a gap and its traceback over lists with contiguous columns everywhere: the streaming sweep with 2-bit
    decisions and the walk in one launch (sweep_stream<kSwTrace>), at every rows-per-lane the list length selects."""
    from caretta_amd import dynamic_time_warping as dtw
    base = []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.choice([1, 2, 9, 63, 64, 65, 129, 150, 300, 321]))
        m = int(rng.choice([1, 3, 8, 64, 65, 129, 300, 321, 513]))
        if rng.integers(0, 2) == 0:
            base.append((np.arange(n), np.arange(m), rng.uniform(-0.3, 1.0, size=(n, m))))
        else:                                                                       # ties
            base.append((np.arange(n), np.arange(m), np.round(rng.uniform(-1.0, 2.0, size=(n, m)) * 2) / 2))
    gap = float(rng.choice([0.2, 0.5, 1.0, 0.05]))
    want = [oracle.smith_waterman(a, b, s, gap) for a, b, s in base]
    if any(w[3] for w in want):
        return 0
    reps = int(rng.choice([1, 1, 40, 700]))                                         # (list length: five ... one row per lane)
    got = dtw.smith_waterman_batch(base * reps, gap)
    for k, (a1, a2, sc) in enumerate(got):
        o1, o2, osc, _none = want[k % len(base)]
        if not (np.array_equal(a1, o1) and np.array_equal(a2, o2) and sc == osc):
            raise AssertionError(f"smith_waterman_batch (gap {gap}, streaming) differs: problem {k} of {len(got)}, n {len(base[k % len(base)][0])} m {len(base[k % len(base)][1])}")
    return len(got)


def check_wide_tensors(oracle, rng):
    """Tensors of 33 ... 192 features (round 6): MultipleAlignment.pairwise on the staged family with the run-time-width staging
    kernel, the list in pieces when it is longer than 1 024 strips; every output against the oracle."""
    dim = int(rng.choice([33, 40, 48, 64, 100, 192]))
    num, length = int(rng.integers(2, 14)), int(rng.choice([20, 64, 65, 150, 300, 520]))
    fam = synthetic.make_family(num, length, dim=dim, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(0, 2)), clades=int(rng.integers(1, 4)))
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    gt = float(rng.choice([70.0, 10.0])) / dim
    msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam])
    ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs, pyoracle.default_params(gamma_tensor=gt), nthreads=8)
    try:
        out = msa.pairwise(dict(gamma_tensor=gt, gamma_coords=0.03, verbose=False))
    except TypeError:
        if np.any(ref["flags"] & 4):
            return 0                                   # (a pair without a positive local alignment: the reference raises too)
        raise
    for key in ("flags", "aln_len", "seed_len", "sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm", "R", "t"):
        if not np.array_equal(out.results[key], ref[key]):
            raise AssertionError(f"wide tensors: {key} differs: d {dim}, lengths {np.diff(offsets)}")
    for p in range(len(pairs)):
        ln = int(ref["aln_len"][p])
        if not np.array_equal(out.alignments[p, :, :ln], ref_aln[p, :, :ln]):
            raise AssertionError(f"wide tensors: alignment of pair {pairs[p]} differs: d {dim}, lengths {np.diff(offsets)}")
    return len(pairs)


def check_neighbor_joining(ctx, oracle, rng):
    """The persistent device kernel against the host implementation (any size) and the oracle (small sizes): uniform,
    tie-heavy, clustered and near-degenerate symmetric matrices."""
    p = int(rng.choice([3, 4, 5, 8, 31, 64, 65, 100, 257, 300, 420, 600]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        a = rng.uniform(0.1, 50.0, size=(p, p))
        d = a + a.T
    elif kind == 1:                                                                 # few distinct values: ties everywhere
        a = rng.integers(0, 4, size=(p, p)).astype(np.float64)
        d = np.triu(a, 1) + np.triu(a, 1).T
    elif kind == 2:                                                                 # points in space, some coincident
        x = np.round(rng.normal(size=(p, 3)) * float(rng.choice([1.0, 3.0])), int(rng.integers(0, 3)))
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
    else:                                                                           # shaped like max(M) - M
        a = rng.uniform(0.0, 300.0, size=(p, p))
        m = np.triu(a, 1) + np.triu(a, 1).T
        d = m.max() - m
    tree, bl = nj.neighbor_joining(d, device=True, ctx=ctx)
    htree, hbl = nj.neighbor_joining(d, device=False)
    if not (np.array_equal(tree, htree) and np.array_equal(bl, hbl)):
        raise AssertionError(f"device neighbor joining differs from the host implementation: P {p} kind {kind}")
    if p <= 100:
        otree, obl = oracle.neighbor_joining(d, hoist=(p > 40))
        if not (np.array_equal(tree, otree) and np.array_equal(bl, obl)):
            raise AssertionError(f"neighbor joining differs from the oracle: P {p} kind {kind}")
    return 1


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx, oracle = engine.Context(0), pyoracle.Oracle()
    t0, batches, pairs, nodes, flagged, dropins, batched, trees, flex_nodes, wide = time.time(), 0, 0, 0, 0, 0, 0, 0, 0, 0
    while time.time() - t0 < seconds:
        fam, _ = random_family(rng)
        n, res = check_batch(ctx, oracle, fam, rng)
        pairs += n
        flagged += int(np.count_nonzero(res["flags"]))
        if rng.integers(0, 3) == 0:
            nodes += check_progressive(oracle, fam, rng)
        if rng.integers(0, 5) == 0 and len(fam) <= 12:
            flex_nodes += check_flexible_progressive(oracle, fam, rng)
        dropins += check_dropins(oracle, rng)
        if rng.integers(0, 4) == 0:
            batched += check_explicit_batch(oracle, rng)
        if rng.integers(0, 4) == 0:
            trees += check_neighbor_joining(ctx, oracle, rng)
        if rng.integers(0, 8) == 0:
            wide += check_wide_tensors(oracle, rng)
        if rng.integers(0, 6) == 0:
            batched += check_sw_gap_stream(oracle, rng)
        batches += 1
    print(f"fuzz_parity: seed {seed}, {batches} batches, {pairs} pairs ({flagged} with a soft-condition flag), "
          f"{nodes} progressive nodes, {flex_nodes} flexible progressive nodes, {dropins} explicit-matrix drop-in cases, {batched} matrices in batched explicit calls, "
          f"{wide} pairs with tensors of 33 ... 192 features, {trees} device neighbor joinings: all bit-identical to the oracle")


if __name__ == "__main__":
    main()
