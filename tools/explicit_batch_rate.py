#!/usr/bin/env python3
"""Rates of the batched explicit-score-matrix kernels: COUNT matrices of N x N doubles resident in HBM.

    python tools/explicit_batch_rate.py [COUNT [N]]        default 8128 x 300 x 300 = 5.85 GB (BASELINE config 3's pair count)

For every function of the reference that takes an explicit score matrix (dynamic_time_warping.py:148-184 dtw_align -- which
ALWAYS traces back --, :188-201 dtw_align_score, :205-222 smith_waterman_score, :226-278 smith_waterman): the device time of
the launch sequence (HIP events on the launch stream: fill kernel + traceback kernel where there is one), the algorithmic
bytes of SURVEY.md 8(d)'s explicit mode -- 8 n m (the matrix, read once) + n m / 2 (4-bit DTW decisions; n m / 4 for the 2-bit
SW ones; none for a score alone) + 24 (n + m) + 8 -- and their rate against the 8 TB/s HBM peak.
bench.py imports `explicit_record`.
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import dynamic_time_warping as dtw  # noqa: E402

HBM_PEAK_GBS = 8000.0


def family_matrices(n, how_many=64):
    """Score matrices as the reference forms them for a plugin (multiple_alignment.py:323-335: make_score_matrix of two related
    structures' per-residue tensors, gamma 7): `how_many` pairs of a synthetic family of n-residue structures."""
    from caretta_amd import score_functions as sf, synthetic
    fam = synthetic.make_family(12, n, seed=20242)
    out = []
    for i in range(len(fam)):
        for j in range(i + 1, len(fam)):
            if len(out) < how_many:
                out.append(sf.make_score_matrix(fam[i].tensors, fam[j].tensors, sf.get_gaussian_score, 7.0))
    return np.stack(out)


def explicit_record(count=8128, n=300, reps=5, with_tracebacks=True, matrices="random"):
    rng = np.random.default_rng(1)
    # 64 distinct matrices, tiled: the kernels still read every byte.  "random": uniform^3 (the scores-only rates do not depend on
    # the values; a traceback through random scores with free gaps is the WORST case for the walks: runs of one or two cells);
    # "family": RBF score matrices of related structures, as the reference forms them
    base = rng.uniform(size=(64, n, n)) ** 3 if matrices == "random" else family_matrices(n)
    idx = np.arange(n)
    t0 = time.perf_counter()
    batch = dtw.ExplicitBatch([(idx, idx, base[k % 64]) for k in range(count)])
    t_up = time.perf_counter() - t0
    cells = float(batch.cells)
    rows = float(count) * 2 * n                              # sum of n + m
    small = 24.0 * rows + 8.0 * count                        # index sequences in, two int64 rows out, the score
    small_score = 8.0 * rows + 8.0 * count                   # (a score alone writes no rows)
    modes = [("smith_waterman_score gap 0 (row sweep)", lambda: batch.smith_waterman_scores(0.0), 8.0 * cells + small_score),
             ("smith_waterman_score gap 0.1 (skewed sweep)", lambda: batch.smith_waterman_scores(0.1), 8.0 * cells + small_score),
             ("dtw_align_score (skewed sweep, no decisions)", lambda: batch.dtw_align(1.0, 0.01, want_alignments=False), 8.0 * cells + small_score)]
    if with_tracebacks:
        modes += [("dtw_align WITH traceback (4-bit decisions + walk)", lambda: batch.dtw_align(1.0, 0.01, want_alignments=True), 8.5 * cells + small),
                  ("smith_waterman WITH traceback gap 0 (2-bit decisions + walk)", lambda: batch.smith_waterman(0.0), 8.25 * cells + small),
                  ("smith_waterman WITH traceback gap 0.1", lambda: batch.smith_waterman(0.1), 8.25 * cells + small)]
    out = {"matrices": count, "rows": n, "values": matrices, "matrix_bytes": 8.0 * cells, "upload_s": t_up, "functions": {}}
    for name, fn, nbytes in modes:
        heavy = "WITH" in name
        fn()
        ms = []
        for _ in range(4 if heavy else reps):
            res = fn()
            ms.append(batch.last_kernel_ms())
        best = min(ms)
        check = float(np.sum(res)) if not heavy else float(sum(r[2] for r in res))
        out["functions"][name] = {"ms": best, "ms_median": sorted(ms)[len(ms) // 2], "algorithmic_bytes": nbytes,
                                  "gb_per_s": nbytes / best / 1e6, "frac_of_hbm_peak": nbytes / best / 1e6 / HBM_PEAK_GBS, "checksum": check}
        del res
    batch.close()
    return out


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 8128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    for matrices in ("random", "family"):
        rec = explicit_record(count, n, matrices=matrices)
        print(f"-- {matrices} score matrices" + (" (uniform^3)" if matrices == "random" else " (tensor RBF of two related structures, as the reference forms them)"))
        for name, r in rec["functions"].items():
            if matrices == "family" and "WITH" not in name:
                continue                                      # (the scores-only rates do not depend on the values)
            print(f"{name}: {count} x {n} x {n}: {r['ms']:.3f} ms (median {r['ms_median']:.3f}) -> {r['gb_per_s']:.0f} GB/s "
                  f"= {r['frac_of_hbm_peak']:.3f} of the 8 TB/s peak; checksum {r['checksum']:.6f}")
        print(f"packing + upload of {rec['matrix_bytes'] / 1e9:.2f} GB: {rec['upload_s']:.2f} s")
