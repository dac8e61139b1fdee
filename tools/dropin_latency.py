#!/usr/bin/env python3
"""Wall time per call of the single-call drop-ins (host arrays in, host arrays out) on one MI355X.

    python tools/dropin_latency.py [L]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import (dynamic_time_warping as dtw, helper, multiple_alignment as ma, score_functions as sf,  # noqa: E402
                         superposition_functions as sup, synthetic)


def timeit(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e3, out


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    fam = synthetic.make_family(2, length, seed=3, clades=1)
    p, q = (ma.Protein(s.name, s.tensors, s.coordinates, "") for s in fam)
    rows = []
    ms, s_t = timeit(lambda: sf.make_score_matrix(p.tensors, q.tensors, sf.get_gaussian_score, 7.0))
    rows.append(("make_score_matrix (tensors, d=10)", ms))
    ms, s_c = timeit(lambda: p.score_function(q, flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False))
    rows.append(("Protein.score_function", ms))
    ar = np.arange(length)
    ms, (a1, a2, _) = timeit(lambda: dtw.dtw_align(ar, ar, s_c, 1.0, 0.01))
    rows.append(("dtw_align", ms))
    ms, _ = timeit(lambda: dtw.dtw_align_score(ar, ar, s_c, 1.0, 0.01))
    rows.append(("dtw_align_score", ms))
    ms, _ = timeit(lambda: dtw.smith_waterman(ar, ar, s_t, 0.0))
    rows.append(("smith_waterman", ms))
    ms, _ = timeit(lambda: dtw.smith_waterman_score(ar, ar, s_c, 0.0))
    rows.append(("smith_waterman_score", ms))
    c1, c2 = helper.get_common_positions(a1, a2)
    x1, x2 = p.coordinates[c1], q.coordinates[c2]
    ms, (rot, tran) = timeit(lambda: sup.paired_svd_superpose(x1, x2))
    rows.append(("paired_svd_superpose", ms))
    ms, _ = timeit(lambda: sup.apply_rotran(x2, rot, tran))
    rows.append(("apply_rotran", ms))
    ms, _ = timeit(lambda: sf.get_rmsd(x1, x2))
    rows.append(("get_rmsd", ms))
    ms, _ = timeit(lambda: ma.tm_score(x1, x2, length, length))
    rows.append(("tm_score", ms))
    msa = ma.MultipleAlignment([p, q])
    ms, _ = timeit(lambda: msa.multiple_align(None, 1.0, 0.01, 1.0, 1.0, dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)))
    rows.append(("multiple_align of 2 structures", ms))
    ms, _ = timeit(lambda: msa.pairwise(dict(gamma_tensor=7.0, gamma_coords=0.03)))
    rows.append(("pairwise() of 1 pair (batched engine)", ms))
    print(f"L = {length}")
    for name, v in rows:
        print(f"  {name:40s} {v:8.3f} ms")


if __name__ == "__main__":
    main()
