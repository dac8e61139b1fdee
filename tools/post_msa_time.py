#!/usr/bin/env python3
"""Wall time of the post-MSA products (make_rmsd_coverage_tm_matrix, superpose, coverage-gap matrix) for P x L.  python tools/post_msa_time.py [P] [L]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import multiple_alignment as ma, msa_superposition as sup, synthetic  # noqa: E402

num = int(sys.argv[1]) if len(sys.argv) > 1 else 128
length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fam = synthetic.make_family(num, length, seed=20242)
prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
msa = ma.MultipleAlignment(prots)
prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
m = msa.make_pairwise_matrix(prm)
aln = msa.multiple_align(m.max() - m, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
for name, fn in (("make_rmsd_coverage_tm_matrix(superpose_first=False)", lambda: ma.make_rmsd_coverage_tm_matrix(aln, prots, superpose_first=False)),
                 ("make_rmsd_coverage_tm_matrix(superpose_first=True)", lambda: ma.make_rmsd_coverage_tm_matrix(aln, prots, superpose_first=True)),
                 ("make_coverage_gap_distance_matrix", lambda: sup.make_coverage_gap_distance_matrix(np.array([aln[p.name] for p in prots])))):
    fn()
    t0 = time.perf_counter()
    fn()
    print(f"P={num} L={length}: {name}: {1e3 * (time.perf_counter() - t0):.2f} ms")
