#!/usr/bin/env python3
"""One very long pair (3000 x 2500 residues: 10 strips of 320 rows, hand-off rows through HBM) against the oracle,
bit for bit.  python tests/long_pair_oracle_check.py   (needs ~1 GB of host memory for the oracle's dense matrices)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import engine, synthetic  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

a = synthetic.make_family(2, 3000, seed=9, clades=1)
b = synthetic.make_family(1, 2500, seed=10, clades=1)
fam = [a[0], b[0], a[1]]
coords, tensors, offsets = synthetic.pack(fam)
pairs = np.array([[0, 1], [1, 0], [0, 2]], np.int32)
ctx = engine.Context(0)
batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
batch.run(engine.make_params())
res, aln = batch.fetch()
ref, ref_aln = Oracle().pairwise_batch(coords, tensors, offsets, pairs, nthreads=3)
for key in ("flags", "aln_len", "seed_len", "sw", "dtw_score", "seed_score", "rmsd", "coverage", "tm", "R", "t"):
    assert np.array_equal(res[key], ref[key]), key
for p in range(len(pairs)):
    ln = int(ref["aln_len"][p])
    assert np.array_equal(aln[p, :, :ln], ref_aln[p, :, :ln])
print("long pairs ok:", res["aln_len"], res["sw"])
