#!/usr/bin/env python3
"""pairwise_matrix_sharded on a single rank against fetched results and the oracle (run by test_gpu_parity.py)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
torch.cuda.init()                                   # torch's HIP runtime first
from caretta_amd import distributed as cdist, engine, synthetic  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

fam = synthetic.make_family(14, 230, seed=77, ragged=True, clades=3)
for k, s in enumerate(fam):                         # very different lengths: several rows-per-lane groups
    cut = [230, 40, 120, 200, 64, 150, 90][k % 7]
    s.coordinates, s.tensors = s.coordinates[:cut].copy(), s.tensors[:cut].copy()
coords, tensors, offsets = synthetic.pack(fam)
m = cdist.pairwise_matrix_sharded(coords, tensors, offsets, engine.make_params())
torch.cuda.synchronize()
pairs = engine.all_pairs(len(fam))
ctx = engine.Context(0)
batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
batch.run(engine.make_params())
res, _ = batch.fetch(want_alignments=False)
batch.close()
assert np.array_equal(m, engine.assemble_matrix(pairs, res["sw"], len(fam)))
ref, _ = Oracle().pairwise_batch(coords, tensors, offsets, pairs, nthreads=8)
assert np.array_equal(res["sw"], ref["sw"])
print("sharded matrix ok", m.shape)
